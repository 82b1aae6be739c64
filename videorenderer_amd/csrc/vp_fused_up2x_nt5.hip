// the fused exact-2x kernel with 5 taps per axis (Lanczos3 as Direct3D 11 draws it: taps 0 and 1 share a texel (quirk Q1)): see vp_fused_up2x.h
#include "vp_fused_up2x.h"

namespace mpcvr {
template hipError_t LaunchFusedUp2xNT<5>(const FusedParams &, const FusedArgs &, int, int, const FusedFrame *, FusedFrame, int, hipStream_t);
}
