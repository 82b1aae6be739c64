// the arbitrary-ratio fused kernel with 7..8 taps: Hamming / bilinear downscales up to ~2.6x: see vp_fused_strip.h
#include "vp_fused_strip.h"

namespace mpcvr {
template hipError_t LaunchFusedStripNT<8>(const FusedArgs &, const StripArgs &, const StoreParams &, int, int, int, int, bool, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
}
