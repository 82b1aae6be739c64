// the fused exact-2x kernel with 6 taps per axis (Lanczos3 with the tap fix, Spline36): see vp_fused_up2x.h
#include "vp_fused_up2x.h"

namespace mpcvr {
template hipError_t LaunchFusedUp2xNT<6>(const FusedParams &, const FusedArgs &, int, int, const FusedFrame *, FusedFrame, int, hipStream_t);
}
