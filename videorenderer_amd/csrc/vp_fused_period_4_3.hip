// the periodic-phase fused kernel at output : source rows = 4 : 3 (1080p -> 1440p, 720p -> 960p): see vp_fused_period.h
#include "vp_fused_period.h"

namespace mpcvr {
template hipError_t LaunchFusedPeriodPQ<4, 3>(const FusedArgs &, const PeriodArgs &, int, int, int, int, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
}
