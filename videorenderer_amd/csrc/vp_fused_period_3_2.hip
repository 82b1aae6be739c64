// the periodic-phase fused kernel at output : source rows = 3 : 2 (720p -> 1080p, 1440p -> 4K): see vp_fused_period.h
#include "vp_fused_period.h"

namespace mpcvr {
template hipError_t LaunchFusedPeriodPQ<3, 2>(const FusedArgs &, const PeriodArgs &, int, int, int, int, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
}
