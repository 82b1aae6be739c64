// vp_params.h — plain-old-data shared between the host planner (vp_plan.cpp) and the HIP kernels
// (vp_kernels.hip).  Everything a launch needs travels by value as a kernel argument.
#pragma once
#include <stdint.h>

namespace mpcvr {

// entries of the fused path's PQ->SDR per-channel table (linear interpolation; worst case 0.17 LSB of the
// 10-bit convert output on saturated colours, where the gamut matrix cancels to near zero)
constexpr int kPqLutSize = 4096;
// PQ EOTF table of the Dolby Vision block convert: log2 of the EOTF sampled in sqrt(x), read as adjacent entries (ds_read2_b32). 8192
// intervals in the 32 KiB the {value, slope} table of 4096 took: interpolation error 1.2e-6 instead of 5e-6 — round 4 found the coarser
// table (built through an fp32 pow chain on top) behind three quarters of the channels beyond 1 LSB on Dolby Vision frames
constexpr int kEotfLutSize = 8192;          // intervals; the table holds kEotfLutSize + 1 values
// PQ ENCODE table of the Dolby Vision level-2 variant (round 5): LinearToST2084(x, 1) over t = log2 x in [-kPqEncLog2Range, 0], kPqEncSize intervals
// (the function is an S-curve of slope <= 0.06 per stop there: linear interpolation is good to 3e-6).  Replaces two pow() and a division per
// channel in front of the trims — 15 of the 40 transcendentals a pixel of that variant costs.  It rides behind the EOTF table in the same
// device buffer (kPqEncOffset floats in) and behind the tone-map table in LDS.
constexpr int kPqEncSize = 1024;            // intervals; kPqEncSize + 1 values
constexpr int kPqEncLog2Range = 48;
constexpr int kPqEncOffset = (kEotfLutSize + 1 + 3) & ~3;

// surface formats of the intermediate / output textures
// (m_InternalTexFmt — DX11VideoProcessor.cpp:1143-1155; m_TexResize is always fp16 — :3155)
// SF_RGBA16 (R16G16B16A16_UNORM) only occurs as the source texture of the 16-bit interleaved RGB formats
enum SurfFmt : int { SF_BGRA8 = 8, SF_RGB10A2 = 10, SF_RGBA16F = 16, SF_RGBA16 = 17 };

// what the generated convert shader appends after "//convert color" (Shaders.cpp:861-923)
enum TailMode : int {
    TAIL_NONE = 0,
    TAIL_PQ_TO_SDR = 1,      // saturate, ST2084ToLinear*scale, Hable, 2020->709, saturate, pow 1/2.2
    TAIL_HLG_TO_SDR = 2,     // saturate, HLGtoLinear, LinearToST2084(/1000), then as PQ
    TAIL_GAMMA_GAMUT = 3,    // saturate, pow(gamma), 2020->709, saturate, pow 1/2.2
    TAIL_HLG_TO_PQ = 4       // HDR output: saturate, HLGtoLinear, LinearToST2084(/1000)   (SHADER_CONVERT_TO_PQ, :885-891)
};

// constants of ps_hdr10_tonemap.hlsl as SetHDR10ShaderParams sanitises them (DX11VideoProcessor.cpp:907-917)
struct HdrToneMapParams {
    float min_mastering, max_mastering, max_cll, max_fall, display_max;
    int selection;           // 1 ACES, 2 Reinhard, 3 Habel, 4 Moebius, 5 BT.2390, 6 ST 2094-10
    int l2_enabled;          // DolbyConstants at b1 (DX11VideoProcessor.cpp:3362-3364)
    float l2k[5];
};

// Dolby Vision constants of the convert shader: PS_DOVI_CURVE x3 (cbuffer b2, Shaders.cpp:715-727), the LMS matrix baked into
// the shader text (Shaders.cpp:826-842) and DolbyConstants (cbuffer b3, :751-760).  Lives in device memory, read through
// ConvertParams::dovi.
struct DoviCurve {
    float pivots[7]; float pad0;
    float coeffs[8][4];
    float mmr[48][4];
    uint32_t methods, mmr_single, min_order, max_order;
};
struct DoviParams {
    DoviCurve curves[3];
    float lms[9];
    int has_mmr;            // which reshape shader variant was generated (:2305-2318)
    int l2_enabled;         // L2Enabled
    float l2k[5];           // ChromaWeight, SaturationGain, TrimSlope, TrimOffset, TrimPower
};
enum { DOVI_RESHAPE_POLY = 1, DOVI_RESHAPE_MMR = 2 };

enum ChromaLoc : int { CLOC_MPEG2 = 0, CLOC_MPEG1 = 1, CLOC_COSITED = 2 };

// organisation of the source texture(s) — Helper.cpp:295-307 (DX11PlaneConfig_t) and ShaderGetPixels' switch on it
enum SrcLayout : int {
    LAY_PLANAR = 0,      // planes 2/3 (also planar RGB: G,B,R sampled as Y,U,V)
    LAY_PACKED422 = 1,   // one RGBA8/RGBA16 texel = two pixels (YUY2, UYVY, Y210, Y216, v210 after CopyFrameV210)
    LAY_PACKED444 = 2,   // one texel = one pixel (AYUV, Y410, Y416)
    LAY_GRAY = 3,        // R8/R16: Sample() returns (Y,0,0,1)
    LAY_RGB = 4          // interleaved RGB in a B8G8R8X8 / R10G10B10A2 / R16G16B16A16 texture
};
// GetCopyPlaneFunction (Helper.cpp:377-412) for the formats whose upload is not a plain copy
enum Repack : int { RPK_NONE = 0, RPK_RGB24, RPK_R210, RPK_RGB48, RPK_BGR48, RPK_BGRA64, RPK_B64A };
enum ColorSystem : int { CST_YUV = 0, CST_RGB = 1, CST_GRAY = 2 };   // Helper.h:129-133

struct SrcFormat {
    int planes;      // 1: single texture ; 2: Y + interleaved UV ; 3: Y,U,V
    int bytes;       // 1 or 2 bytes per sample
    int div_w, div_h;
    int shift;       // CopyPlane10to16 (<<6) applied on load for 10-bit planar (Helper.cpp:789-803)
    int v_first;     // YV12 family: 2nd plane holds V
    int subsampling; // 420 / 422 / 444 / 400
    int cdepth;
    int layout;      // SrcLayout
    int ci[4];       // packed 4:2:2: texel components of Y0,U,Y1,V ; packed 4:4:4: components of Y,U,V
    int bits10;      // texel is an R10G10B10A2 dword (Y410)
};

struct ConvertParams {
    const uint8_t *plane[3];
    int pitch[3];
    int tex_w, tex_h;        // luma texture size
    int cw, ch;              // chroma texture size
    int rect_l, rect_t;      // source rect origin inside the texture
    int out_w, out_h;        // rect size == convert-output size
    SrcFormat fmt;
    int blend_deint;         // blendDeint420 variant of the shader (Shaders.cpp:115,232-237)
    int chroma_scaling;      // CHROMA_*
    int chroma_loc;          // ChromaLoc
    int tail;                // TailMode
    float gamma;             // for TAIL_GAMMA_GAMUT (1.0 => skip pow)
    float cm[12];            // cm_r, cm_g, cm_b, cm_c  (PS_COLOR_TRANSFORM, Shaders.h:25-30)
    float lum_scale;         // PS_PARAMETERS.LuminanceScale
    float gamut[9];          // matrix_conv_prim
    int out_fmt;             // SurfFmt of m_TexConvertOutput
    const DoviParams *dovi;  // device pointer, null unless m_Dovi.bValid
    const float *pq_lut;     // device, kPqLutSize floats (BuildPqSdrLut) for the folded convert kernel's PQ->SDR tail; null => ALU chain
};

struct Surface {
    void *ptr;
    int pitch;      // bytes
    int w, h;
    int fmt;        // SurfFmt
};

// epilogue of a resize / copy draw
enum StoreMode : int {
    ST_SURFACE = 0,    // round to the destination surface format (fp16, internal, or RT without final pass)
    ST_FINAL = 1       // round to `mid_fmt` (m_TexsPostScale), then ps_final_pass -> RT
};

struct StoreParams {
    void *dst;          // destination base (window-sized for RT stores, image-sized otherwise)
    int dst_pitch;
    int dst_fmt;        // SurfFmt of dst
    int mode;           // StoreMode
    int mid_fmt;        // SurfFmt the value passes through before the final pass (ST_FINAL)
    int quant;          // 255 or 1023 (ps_final_pass QUANTIZATION)
    int off_x, off_y;   // image (0,0) lands at window pixel (off_x, off_y)
    int clip_w, clip_h; // window size for clipping (0 => no clipping, dst is image-sized)
    const uint16_t *dither;  // 32x32 fp16 table (device)
};

// Tex[AXIS] * wh[AXIS] of output i of n_out, as the reference's shaders see it.  FillVertices (DX11VideoProcessor.cpp:133-138)
// puts fp32 texture coordinates on the quad's corners (src_dx = 1.0f / texLen; src_l = src_dx * rect.left; src_r = src_dx *
// rect.right), the rasteriser interpolates them to the pixel centre (i + .5) / n_out (taken as exact, rounded once to fp32) and the
// shader multiplies by wh[AXIS] = (float)texLen (ps_interpolation_*.hlsl:25, ps_convolution.hlsl:30).  `rev`: the coordinate
// starts at the far edge of the source range (rotation / flip, FillVertices :140-169).  Every tap table and the plain Jinc kernel
// go through this one function; the oracle restates it the same way (oracle/mpcvr_oracle.c axis_center) and is bit-identical to
// the reference's shader text with it — the fp32 roundings of the corner values move t = frac(Tex * wh - .5) by up to 2^-13 at
// 3840 -> 7680, enough to move one 8-bit channel in 2,000 by one code.
__host__ __device__ inline float TexCenter(int org, int len, int tex_len, int i, int n_out, int rev)
{
    const float src_d = 1.0f / (float)tex_len;
    const float c_lo = src_d * (float)org, c_hi = src_d * (float)(org + len);
    const double ua = rev ? c_hi : c_lo, ub = rev ? c_lo : c_hi;
    const double a = ((double)i + 0.5) / (double)n_out;
    const float tex = (float)(ua + (ub - ua) * a);
    return tex * (float)tex_len;
}

// texture coordinate of an output pixel of a (possibly rotated / flipped) draw, per screen axis: TexCenter(org, len, tex, i, n, rev);
// step = len / n serves the phase-table kernels (dyadic ratios), whose tap bases do not depend on the last ulp
struct DrawCoords {
    int org_x, len_x, rev_x; float step_x;     // run through by screen x
    int org_y, len_y, rev_y; float step_y;     // run through by screen y
    int swap;                                  // rotation 90/270: screen x runs along texture Y
    int tex_x, tex_y;                          // texture extent along the axis screen x / y runs through (wh[AXIS])
    int n_x, n_y;                              // outputs along screen x / y (the viewport)
};

// Jinc2m at dyadic ratios: the 16 weights of an output pixel per phase (BuildJincPhases, vp_kernels.hip)
struct JincPhases { float w[4][4][16]; float wsum[4][4]; int px, py; };     // [phase y][phase x][j * 4 + i]

// per-output-index tap tables for one axis (built on the host, vp_plan.cpp)
struct AxisTaps {
    const int32_t *idx;   // [n_out * ntaps] clamped source indices
    const float *w;       // [n_out * ntaps]
    const float *wsum;    // [n_out] (only when normalise)
    int ntaps;
    int normalise;        // ps_convolution: avg /= ww
    // hints for the folded kernels (0 / null: not available, the plain kernel runs)
    const int32_t *blk_lo;   // [ceil(n_out / 64)] smallest source index any tap of outputs 64b .. 64b+63 touches
    const int32_t *idx_t;    // [ntaps][n_out] the same tables tap-major: lanes of a wave read consecutive words
    const float *w_t;
    int n_out;
    const int32_t *blk8_lo;  // the same per block of 8 outputs (row-tap kernel: 8 output rows per workgroup)
    int blk8_span;
    const int32_t *blk32_lo; // ... and per block of 32 outputs (tiled two-draw kernel: 64 x 32 output tiles)
    int blk32_span;
    int blk_span;            // max over blocks of (largest - smallest index + 1); 0 = unknown
    int other_identity;      // the `other` map of this draw is x -> x (no flip / rotation / source offset)
};
enum { kResizeRowSpanMax = 14 };
enum { kResizeTileRowsMax = 72 };  // source rows the tiled two-draw kernel keeps per 32 output rows   // source rows the row-tap kernel keeps per lane in LDS for its 8 output rows
enum { kResizeSpanMax = 192 };     // source texels per row one wave stages in LDS for the column-tap kernel (4 rows x 4 waves x 16 B)

// 2x fast path: the two phase-weight sets per axis (t = 0.75 for even outputs, 0.25 for odd)
struct Up2xWeights {
    int ntaps;            // 4 or 6
    float w_even[6];      // taps at base-1.. (4) or base-2.. (6), base = k-1 for output 2k
    float w_odd[6];       // base = k for output 2k+1
    int q1_quirk;         // Lanczos3 D3D11: tap 1 re-reads tap 0's texel
};

}  // namespace mpcvr
