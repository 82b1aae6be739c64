// the arbitrary-ratio fused kernel with 4 taps: Mitchell / Catmull-Rom / Lanczos2, bilinear and box downscales: see vp_fused_strip.h
#include "vp_fused_strip.h"

namespace mpcvr {
template hipError_t LaunchFusedStripNT<4>(const FusedArgs &, const StripArgs &, const StoreParams &, int, int, int, int, bool, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
}
