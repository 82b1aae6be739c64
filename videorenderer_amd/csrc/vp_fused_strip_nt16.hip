// the arbitrary-ratio fused kernel with 9..16 taps: bicubic / Lanczos downscales beyond ~2x (ps_convolution.hlsl:30-47 with support * scale), one pixel per lane, a 32-row ring: see vp_fused_strip.h
#include "vp_fused_strip.h"

namespace mpcvr {
template hipError_t LaunchFusedStripNT<16>(const FusedArgs &, const StripArgs &, const StoreParams &, int, int, int, int, bool, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
}
