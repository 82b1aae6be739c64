// vp_launch.h — host-callable launchers of the HIP kernels (vp_kernels.hip, vp_fused.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include "vp_params.h"

namespace mpcvr {

// pass-per-kernel path (vp_kernels.hip)
// generic = true: the one-kernel-fits-all version only (MPCVR_FLAG_NO_FUSED keeps the whole path on the plain kernels)
hipError_t LaunchConvert(const ConvertParams &P, const Surface &out, hipStream_t s, bool generic = false);
hipError_t LaunchConvertDirect(const ConvertParams &P, const StoreParams &st, hipStream_t s);
// CopyFrameRGB24 / R210 / RGB48 / BGR48 / BGRA64 / B64A / CopyPlaneAsIs: interleaved RGB sample -> its texture
// srcs != nullptr: a batch of n samples in one launch per 32 frames — frame z reads srcs[z] and writes dst + z * dst_stride (src unused)
struct SrcTable32 { const uint8_t *p[32]; int n; };
hipError_t LaunchRepackRgb(int kind, const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch, int width, int lines, hipStream_t s,
                           const void *const *srcs = nullptr, int n = 1, size_t dst_stride = 0);
// CopyFrameV210 (Helper.cpp:709-748): v210 sample -> Y210-layout texture
hipError_t LaunchRepackV210(const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch, int lines, hipStream_t s,
                            const void *const *srcs = nullptr, int n = 1, size_t dst_stride = 0);
// axis = screen axis the tap table runs along; swap = rotation 90/270 (taps address the other texture axis)
// batch: n frames per launch — frame z reads in.ptr + z * in_stride and writes frames[z].dst when a
// frame table is given, else st.dst + z * dst_stride (the folded kernels and the one-kernel-fits-all versions alike).
struct FusedFrame;
struct ResizeBatch { int n = 1; size_t in_stride = 0, dst_stride = 0; const FusedFrame *frames = nullptr; int dst_aligned8 = 1; /* every frames[z].dst on an 8-byte boundary */ };
hipError_t LaunchResize(int axis, bool swap, const Surface &in, const AxisTaps &taps, const int32_t *other,
                        int out_w, int out_h, const StoreParams &st, hipStream_t s, bool generic = false,
                        const ResizeBatch *batch = nullptr);
bool ResizeHasFoldedKernel(int axis, bool swap, const Surface &in, const AxisTaps &taps, const StoreParams &st);
// both draws of an unrotated two-pass resize in one LDS-tiled kernel (no m_TexResize in memory); st = the second draw's epilogue
bool Resize2DSupported(const Surface &in, const AxisTaps &tx, const AxisTaps &ty, const StoreParams &st);
// mid_h = rows of the first draw's result (entries of `other`): the source rect's extent along screen y
hipError_t LaunchResize2D(const Surface &in, const AxisTaps &tx, const AxisTaps &ty, const int32_t *other, int mid_h, int out_w, int out_h,
                          const StoreParams &st, hipStream_t s, const ResizeBatch *batch = nullptr);
hipError_t LaunchCopy(const Surface &in, int out_w, int out_h, const StoreParams &st, hipStream_t s);
// m_pPSCorrection shaders as a same-size pass (kind = MPCVR_CORR_*); fix16 = the kind's 4x4 fix-up matrix (unused for 5, 6)
hipError_t LaunchCorrection(int kind, const Surface &in, const Surface &out, const float fix16[16], const float gamut9[9], float lum_scale, hipStream_t s);
// ps_hdr10_tonemap.hlsl: HDR10 local tone mapping as a post-scale step
hipError_t LaunchHdr10ToneMap(const Surface &in, const HdrToneMapParams &tm, int out_w, int out_h, const StoreParams &st, hipStream_t s,
                              const ResizeBatch *batch = nullptr);
// ps_resize_onepass_jinc2.hlsl: the 2-D Jinc2m draw
// phases_dev: device copy of the table BuildJincPhases filled (dyadic, unrotated draws: weights per phase instead of per pixel)
// fast: the default tier may take the quad kernel (exact 2x; FMA contraction) instead of the phase-table kernel
// centres_dev: device copy of BuildDrawCentres' table (n_x + n_y floats) — what the plain kernel (no phase table) reads its texcoords from
hipError_t LaunchJinc2(const Surface &in, const DrawCoords &dc, int out_w, int out_h, const StoreParams &st, hipStream_t s,
                       const void *phases_dev = nullptr, bool fast = false, const ResizeBatch *batch = nullptr, const float *centres_dev = nullptr);
void BuildDrawCentres(const DrawCoords &dc, float *out);       // out: dc.n_x + dc.n_y floats of host memory
// vp_jinc.hip: Jinc2m at exactly 2x on both axes, one 2x2 output quad per lane (25 LDS texel reads for 4 pixels instead of 64)
bool Jinc2QuadSupported(const Surface &in, const DrawCoords &dc, int out_w, int out_h, const StoreParams &st);
hipError_t LaunchJinc2Quad(const Surface &in, const DrawCoords &dc, int out_w, int out_h, const StoreParams &st, hipStream_t s, const void *phases_dev,
                           const ResizeBatch *batch = nullptr);
bool BuildJincPhases(const DrawCoords &dc, void *out_table);      // out_table: JincPhasesBytes() bytes of host memory
size_t JincPhasesBytes();

// fused 2x path (vp_fused.hip): convert + X pass + Y pass + final pass in one kernel, n frames per launch.
struct FusedFrame {
    const uint8_t *src;   // sample base (planes back to back)
    void *dst;            // render target base
};
struct FusedParams {
    ConvertParams conv;       // plane pointers are ignored; derived per frame from FusedFrame::src
    size_t plane_off[3];      // byte offsets of the planes inside a sample
    Up2xWeights wx, wy;
    StoreParams store;        // dst ignored; per frame
    int out_w, out_h;         // 2*conv.out_w, 2*conv.out_h
    const float *pq_lut;      // device, kPqLutSize floats: x -> Hable(ST2084ToLinear(x)*scale)/hable(4.8); null => ALU
    int fast_convert;         // layout/alignments allow the vectorised convert
    int dst_aligned16;        // every render target of the launch starts on a 16-byte boundary
    int src_aligned16;        // every sample of the launch starts on a 16-byte boundary (wide block convert)
    int literal_tail;         // MPCVR_FLAG_NO_LUT: evaluate the HDR tails literally in ALU (no LUT, no algebraic shortcut)
    const float *hlg_lut;     // device, kPqLutSize floats (BuildHlgInverseLut): HLG -> SDR tail of the fused kernels; null => literal chain
    const float *eotf_lut;    // device, kEotfLutSize + 1 floats: log2 ST2084ToLinear(x, 1) at x = (i/N)^2 — the Dolby Vision variants of the block convert decode PQ from it
    int dovi_l2;              // the frame's Dolby Vision metadata carries level-2 trims for this display (DoviParams::l2_enabled)
    // a batch with one RPU per frame: conv.dovi points at n DoviParams and dovi_cm at n colour matrices (12 floats each, the layout of
    // ConvertParams::cm); the block convert's Dolby Vision variants index both by the frame.  Null: one RPU for the launch (conv.dovi, conv.cm)
    const float *dovi_cm;
    const float *jinc_tab;    // fused 2x route with the 2-D Jinc2m filter (PassPlan::fused_jinc): the device copy of BuildFusedJincTable's table; null: separable taps (wx, wy)
    int exact_wide;           // an HDR10 tone-mapping operator follows (PassPlan::hdr_tonemap): its curve multiplies a code of the 10-bit internal format by up
                              // to ~5, so such plans take the exact form of the convert stage for 10-bit internal formats as well
    int exact_convert;        // a resize reads this launch's convert output: 8-bit internal formats then take the exact form of the convert stage
                              // (FusedArgs::exact_cv).  LaunchFusedUp2x / LaunchFusedStrip set it themselves; the block convert's callers say so.
    int inflight;             // single-frame launches: frames the host keeps running side by side (the context's frame lanes), 0 / 1 = none.  The
                              // segment rules count them like frames of a batch: four overlapping 4K frames fill the chip with long segments
};
// vp_errdiff.hip — the error-diffusion final pass (EXTENSION, bUseDither = 2; definition in vp_errdiff_core.h): frame z's R10G10B10A2 image
// (frames[z].src, window geometry, src_pitch) -> its B8G8R8A8 render target (frames[z].dst, dst_pitch) inside [x0, x1) x [y0, y1)
struct ErrDiffParams {
    int x0, y0, x1, y1;        // video rect ∩ window, window coordinates
    int src_pitch, dst_pitch;  // bytes; src is readable from two pixels in front of a row of the region to three behind it (the caller's intermediates have the margins);
                               // rows of both surfaces on dword boundaries (the kernel moves 16-byte pieces that need no more)
    int order;                 // ticket order: 1 = band-major (a band of every frame, then the next band), 0 = frame-major (MPCVR_ERRDIFF_ORDER, A/B)
    int gen;                   // generation the launch tags its hand-off words with, 1 .. 4095: no word of an earlier launch still in the rows may carry
                               // it (the owner counts launches and passes 0 — "clear the rows first" — when the count wraps or the layout changed)
    int spin_limit;            // polls a band grants the band above before it gives up and sets *status; 0 = the default (2^21, ~2 s)
    int test_stall;            // tests only: frame 0's first band never publishes, so that the band below gives up
    uint32_t *handoff;         // device: ErrorDiffusionHandoffBytes(P, n_frames) — the bands' bottom rows for the bands below and the ticket counter
    int *status;               // host memory the device can write: set to 1 when a band gave up waiting for the band above (never, unless a launch is broken)
};
size_t ErrorDiffusionHandoffBytes(const ErrDiffParams &P, int n_frames);
hipError_t LaunchErrorDiffusion(const ErrDiffParams &P, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s);
bool FusedUp2xSupported(const FusedParams &P);
// vp_fused_jinc.hip: the weight table of the fused Jinc2m kernel, from BuildJincPhases' table of a 2x draw
hipError_t LaunchEvalDoviTail(const float *rgb_dev, float *out_dev, size_t n, const float lms[9], const float k[5], const float gamut[9], int l2, float lum_scale, int stage, hipStream_t s);
size_t FusedJincTableBytes();
size_t FusedJincLdsBytes(const FusedParams &P);        // dynamic LDS of the fused Jinc2m kernel this plan would launch (vs DeviceLdsLimit())
void BuildFusedJincTable(const void *phases, float *out);
bool BlockConvertLayout(const FusedParams &P, bool catmull_420);       // source layout + chroma filter convert_block serves
// the fused kernel's convert stage as a kernel of its own: 2x2 blocks, shared chroma fetch, table tone map.  P.store describes
// the destination: texels of the internal format (m_TexConvertOutput, or the render target when nothing follows: to_rt), or
// the final pass into a B8G8R8A8 render target (store.mode == ST_FINAL).  out_w / out_h / wx / wy of P are not used.
bool ConvertBlocksSupported(const FusedParams &P, bool to_rt);
// batch_stride != 0: frame z is stored at P.store.dst + z * batch_stride (a batched intermediate) instead of frames[z].dst
// frames_host != nullptr and n_frames <= 32 (<= kHostTableMax where the streaming kernel takes the launch; hipErrorInvalidValue otherwise: the
// caller then uploads the table): the frame table travels by value in the kernel arguments (frames_dev is not read)
struct FrameTable32 { FusedFrame f[32]; int n; };
// k_convert_stream: up to 128 frames (a step of 1080p frames is 128 of them: no table upload in front of an 80-300 us launch)
struct FrameTable128 { FusedFrame f[128]; int n; };
enum { kHostTableMax = 128 };
hipError_t LaunchConvertBlocks(const FusedParams &P, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s,
                               size_t batch_stride = 0, const FusedFrame *frames_host = nullptr);
// frames_dev == nullptr: n_frames must be 1 and `single` is used (no device-side table needed)
hipError_t LaunchFusedUp2x(const FusedParams &P, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s);

// arbitrary-ratio fused path (vp_fused_strip.hip): block convert + both draws of an unrotated two-pass resize + final pass in one
// kernel, driven by the tap tables of BuildAxisTaps; wave-autonomous strips like the 2x kernel, the vertical window in LDS.
struct FusedStripParams {
    FusedParams fp;            // conv / plane_off / store / pq_lut / alignment flags (wx, wy, out_w, out_h of fp are not used)
    // device copies of PlanFusedStrip's tables: nt taps per output, zero-weight padding, normalisation folded in
    const void *xi_t, *xw_t;   // X taps, tap-major: int32 / float [nt][out_w]
    const void *yi, *yw;       // Y taps, row-major: int32 / float [out_h][nt]
    const void *yrange;        // int32[out_h][2]: {smallest, largest} source row of every output row's taps
    const void *xstrip;        // int32[n_strips][2]: {smallest, largest} source column of every strip's taps
    int out_w, out_h;
    int nt, pxl, strip_w, ring, acols;   // PlanFusedStrip's choices
    // surface mode (surf.ptr != nullptr or surf_batch): no convert stage — the X draw samples `surf` (m_TexConvertOutput of any
    // convert kernel, or the source texture of an interleaved RGB sample); fp.store is the epilogue, fp.conv is not used.
    // other: the draw's row map (device; null = identity); mid_h: rows of the X draw's result; surf_stride: bytes between the
    // frames of a batch (frame z reads surf.ptr + z * surf_stride)
    Surface surf;
    const int32_t *other;
    int mid_h;
    size_t surf_stride;
    int surface_mode;
    // periodic-phase variant (vp_fused_period.h; per_P != 0): device copies of PlanFusedPeriod's tables.  LaunchFusedStrip takes it
    // whenever the launch meets its preconditions (fast epilogue, 8-byte aligned rows) and falls back to k_fused_strip otherwise.
    int per_P, per_Q, per_nt, per_acols, per_strip_w, per_own;
    int *ran_period;           // host, may be null: LaunchFusedStrip notes which kernel it launched (1 = k_fused_period, 0 = k_fused_strip)
    int per_force;             // MPCVR_FLAG_FORCE_PERIOD: also where the planner would prefer k_fused_strip
    const void *per_xi_t, *per_xw_t, *per_yw, *per_xstrip;
};
bool FusedStripSupported(const FusedStripParams &S);
hipError_t LaunchFusedStrip(const FusedStripParams &S, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s);
size_t FusedStripLdsBytes(const FusedStripParams &S);
size_t DeviceLdsLimit();       // LDS bytes one workgroup may claim on the current device (queried, not assumed)
// true when LaunchFusedStrip would run the periodic-phase kernel for this launch (GetVPInfo reports it)
bool FusedPeriodTakes(const FusedStripParams &S);

}  // namespace mpcvr
