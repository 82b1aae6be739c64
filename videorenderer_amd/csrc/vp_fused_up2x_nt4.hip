// the fused exact-2x kernel with 4 taps per axis (Mitchell / Catmull-Rom / Lanczos2): see vp_fused_up2x.h
#include "vp_fused_up2x.h"

namespace mpcvr {
template hipError_t LaunchFusedUp2xNT<4>(const FusedParams &, const FusedArgs &, int, int, const FusedFrame *, FusedFrame, int, hipStream_t);
}
