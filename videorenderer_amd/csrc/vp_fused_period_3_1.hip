// the periodic-phase fused kernel at output : source rows = 3 : 1 (720p -> 2160p, 480p -> 1440p, 360p -> 1080p): see vp_fused_period.h
// (every third output row sits exactly on a texel centre: period_centre)
#include "vp_fused_period.h"

namespace mpcvr {
template hipError_t LaunchFusedPeriodPQ<3, 1>(const FusedArgs &, const PeriodArgs &, int, int, int, int, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
}
