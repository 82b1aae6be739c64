// the periodic-phase fused kernel at output : source rows = 1 : 2 (4K -> 1080p (the interpolation shader at exactly 50 %)): see vp_fused_period.h
#include "vp_fused_period.h"

namespace mpcvr {
template hipError_t LaunchFusedPeriodPQ<1, 2>(const FusedArgs &, const PeriodArgs &, int, int, int, int, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
}
