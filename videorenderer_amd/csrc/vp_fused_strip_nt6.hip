// the arbitrary-ratio fused kernel with 5..6 taps: Lanczos3, Spline36 (extension): see vp_fused_strip.h
#include "vp_fused_strip.h"

namespace mpcvr {
template hipError_t LaunchFusedStripNT<6>(const FusedArgs &, const StripArgs &, const StoreParams &, int, int, int, int, bool, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
}
