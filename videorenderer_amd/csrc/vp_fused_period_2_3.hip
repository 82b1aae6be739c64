// the periodic-phase fused kernel at output : source rows = 2 : 3 (4K -> 1440p, 1080p -> 720p (the interpolation shader below 2x)): see vp_fused_period.h
#include "vp_fused_period.h"

namespace mpcvr {
template hipError_t LaunchFusedPeriodPQ<2, 3>(const FusedArgs &, const PeriodArgs &, int, int, int, int, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
}
