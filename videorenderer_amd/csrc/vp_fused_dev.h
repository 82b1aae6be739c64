// vp_fused_dev.h — device helpers shared by the fused exact-2x kernels (vp_fused.hip: packed-fp32 taps; vp_fused_mx.hip:
// the taps on the matrix cores) and by the block convert: the kernel argument block, raw-code loads, the 2x2-block convert
// (ShaderGetPixels CHROMA_Bilinear + colour matrix + HDR tail).  Included by .hip files only.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "vp_device.h"
#include "vp_launch.h"
#include "vp_plan.h"

namespace mpcvr {

// everything the kernel needs, flattened (kernel argument => SGPRs)
struct FusedArgs {
    uint32_t off_u, off_v;         // byte offsets of the chroma plane(s) inside a sample (u = interleaved UV when biplanar)
    int pitch_y, pitch_c;
    int tex_w, cw, ch;             // luma width, chroma size
    int rect_l, rect_t, W, H;      // source rect origin and size (== convert-output size)
    int bytes, planes;
    int center_h;                  // MPEG-1 siting: chroma sample centred between luma columns
    int v_off4;                    // vertical chroma offset in quarter chroma rows: 1 for co-sited (+0.25), else 0
    int vk1, vk2, vk3;             // chroma_v4(sy) = vk1 * sy + vk2 * (sy & ~1) + vk3: the three siting rules as one branch-free scalar expression
    int sub422;                    // 4:2:2 planar / bi-planar (P210, P216, YV16, YUV422P10...): chroma subsampled horizontally only
    int sub444;                    // 4:4:4 planar (YV24, YUV444P8/10/16): a chroma sample per pixel, no interpolation at all
    int packed422;                 // one plane of (Y0,U,Y1,V) texels (YUY2, UYVY, Y210, Y216, v210 after the unpack): implies sub422
    int ci[4];                     // packed422: position of Y0, U, Y1, V inside a texel; packed444: of Y, U, V
    int packed444;                 // one texel per pixel: 1 = four bytes (AYUV), 2 = 10:10:10:2 (Y410), 3 = four words (Y416); implies sub444
    int gray;                      // one plane, no chroma (Y8, Y10, Y16): U = V = 0; implies sub444
    int nearest;                   // CHROMA_Nearest on 4:2:0 / planar 4:2:2: chroma texel (sx / div_w, sy / div_h), no filter (Shaders.cpp:239-241)
    float cw_own, cw_next;         // the odd luma column's chroma = cw_own * texel c0 + cw_next * texel c0 + 1: {.5, .5}; {0, 1} at 4:4:4; {1, 0} nearest
    float m[9], c[3];              // colour matrix with the UNORM scale (and CopyPlane10to16 shift) folded in
    int tail; float gamma, lum_scale;
    float gamut[9];
    const float *lut;              // LUT_N floats (device) for TAILK_PQ_LUT
    float maxv, inv_maxv;          // internal UNORM format
    float q_over_maxv;             // ps_final_pass QUANTIZATION / maxv
    uint32_t epi_mul;              // FASTEPI: ceil(QUANTIZATION * 2^24 / maxv), see the final-pass epilogue
    float we[6], wo[6];            // phase weights (even/odd outputs); Q1-folded by the launcher
    int dst_pitch, off_x, off_y;
    int final_pass, out10;
    float quant;
    const uint16_t *dither;
    int seg_rows;
    int dbg;                       // ablation switches of the matrix-core kernel (MPCVR_MX_DBG; 0 in normal use)
    // Dolby Vision (DV template argument of convert_block): reshaping curves / LMS matrix / L2 trims (device DoviParams, copied
    // into LDS by the kernel), the PQ EOTF table, and the UNORM scales the matrix does NOT carry then (the curves want 0..1 values)
    // CHROMA_CatmullRom (4:2:0): catmull_weights(t) of even / odd luma columns and rows for the stream's chroma siting
    float crx[2][4], cry[2][4];
    const DoviParams *dovi;
    const float *eotf_lut;         // kEotfLutSize + 1 floats (device): log2 ST2084ToLinear((i / kEotfLutSize)^2, 1)
    float sy, sc;
    // one RPU per frame of a batch (mpcvr_process_batch_dovi): frame z reads dovi[z] and the colour matrix dovi_cm[12 z .. 12 z + 11]
    // (ycc_to_rgb_matrix / offset of ITS RPU, rows then constants, as ConvertParams::cm) instead of dovi[0] and m / c above
    const float *dovi_cm;
    int dovi_per_frame;
    // exact_cv (8-bit internal format in front of a resize, no tail, no Dolby Vision): the convert stage runs convert_block_exact — the
    // reference's own expression shapes on 0..1 values, nothing contracted — so that every texel of m_TexConvertOutput carries the
    // oracle's code (the fast form is one code off on 1e-4 of the texels; a negative-lobe filter behind it can make that two).
    // xm / xc: the colour matrix WITHOUT the UNORM scale; code / maxv = unorm_div with (xd, xr) = (maxv / 2^shift, 2^shift / maxv) per plane kind
    int exact_cv;
    float xm[9], xc[3];
    float xdy, xry, xdc, xrc;
};

namespace {

constexpr int S = 120;             // source pixels per strip
constexpr int AW = 128;            // LDS A row width: rect columns x0-4 .. x0+123
constexpr int WAVES = 4;           // strips per workgroup
constexpr int A_FLOATS = 3 * AW * 2;   // [ch][col][row a | row a+1]
constexpr int LUT_N = kPqLutSize;  // PQ->SDR per-channel table (vp_params.h)
constexpr int LDS_A = WAVES * A_FLOATS * 4;
constexpr int LDS_D = 32 * 32 * 2;      // dither table, fp16 bits (generic epilogue)
constexpr int LDS_DB = 32 * 32 * 4;     // dither table as integers j << 14 (d = j/1024) for the FASTEPI epilogue
constexpr int LDS_T = LUT_N * 8;   // {value, delta-to-next} pairs

typedef const __attribute__((address_space(1))) uint8_t *gcptr;
typedef __attribute__((address_space(1))) uint8_t *gptr;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum { TAILK_NONE = 0, TAILK_PQ_LUT = 1, TAILK_ALU = 2, TAILK_HLG = 3 };
// tails that read a 4096-entry {value, slope} table out of LDS (P.lut): PQ -> SDR = hable(ST2084ToLinear(x) * scale) / hable(4.8);
// HLG -> SDR = the per-channel part of HLGtoLinear (inverse_HLG, hlg.hlsl:1-9)
__host__ __device__ constexpr bool tail_has_table(int t) { return t == TAILK_PQ_LUT || t == TAILK_HLG; }
// Dolby Vision variants of convert_block: DV_SDR = PQ -> SDR tail without level-2 trims (the LMS step's PQ encode and the tail's PQ
// decode cancel and are elided, Hable in ALU); DV_SDR_L2 = PQ -> SDR with level-2 trims (PQ decode from the table, encode and trims
// in ALU, tone map from the HDR10 path's table); DV_GENERAL = everything else (HDR output, no tone mapping): literal chain
enum { DV_NONE = 0, DV_SDR = 1, DV_GENERAL = 2, DV_SDR_L2 = 3 };
constexpr int EOTF_N = kEotfLutSize;
constexpr int LDS_E = (EOTF_N + 1 + 3) / 4 * 16;   // PQ EOTF table: EOTF_N + 1 values, adjacent pairs read with one ds_read2_b32
constexpr int LDS_V = (sizeof(DoviParams) + 15) & ~15;
constexpr int LDS_PE = kPqEncSize * 8;         // PQ encode table of DV_SDR_L2: {value, slope} pairs behind the tone-map table
// source specialisation: GENERIC reads planes / bytes / siting at run time; P01X = bi-planar 16-bit (P010/P016), NV12 =
// bi-planar 8-bit, PLANAR16 / PLANAR8 = three planes of 16- / 8-bit samples (YUV420P10/16, YV12 / I420: what software decoders
// hand over), all with MPEG-2 or co-sited chroma (not horizontally centred)
enum { SRC_GENERIC = 0, SRC_P01X = 1, SRC_NV12 = 2, SRC_PLANAR16 = 3, SRC_PLANAR8 = 4,
       SRC_SURFACE = 5 };      // k_fused_strip only: no convert stage, the source is a B8G8R8A8 / R10G10B10A2 / fp16 surface
// The exact form of the convert stage (convert_block_exact, asked for by FusedArgs::exact_cv) is a compile-time property of a KERNEL, its
// last template argument: XC_NEVER = the fast form, XC_ALWAYS = the exact one.  An instantiation that can meet an 8-bit internal format in
// front of a resize (exact_capable() == XC_RUNTIME: no tail, no Dolby Vision, no 10 -> 8 final pass behind it, an 8-bit loader or the
// run-time one — FusedSourceKind sends 16-bit samples behind a forced 8-bit format there) is built TWICE and the launcher picks by
// exact_cv (fused_*_kernel()); every other one exists in the fast form alone.  How it got there (round 5, 1080p NV12 -> 1440p, frames/s):
//   123 k  round 4, the fast form alone (a class of frames two codes off the reference behind negative-lobe filters);
//    90 k  a wave-uniform branch on exact_cv inside convert_block of EVERY tail-less instantiation: 70 -> 81 VGPRs in the strip kernels,
//          61 -> 69 in the streaming convert — C1 lost 15 %, HDR passthrough 22 %, with the branch never taken there;
//    90 k  the branch in the capable instantiations only (the others back to their registers and speed): still 81 - 85 VGPRs;
//    88 k  two bodies behind one branch at the top of the kernel: 76 VGPRs but 72 spilled SGPRs (the arguments of both bodies are loaded in
//          the entry block);
//   106 k  two kernels.
enum { XC_NEVER = 0, XC_RUNTIME = 1, XC_ALWAYS = 2 };
template <int TAIL, int SRC, bool FINAL10>
__host__ __device__ constexpr int exact_capable() { return (TAIL == 0 /* TAILK_NONE */ && !FINAL10 && (SRC == 0 /* GENERIC */ || SRC == 2 /* NV12 */ || SRC == 4 /* PLANAR8 */)) ? XC_RUNTIME : XC_NEVER; }
// epilogue specialisation: DITHER8 = B8G8R8A8 target behind a final pass (integer form); DIRECT8 = B8G8R8A8 or R10G10B10A2 target written
// straight from the Y pass (no post-scale step: 8-bit sources, HDR passthrough to a 10-bit swap chain); both require 16-byte aligned rows and off_x % 4 == 0
enum { EPI_GENERIC = 0, EPI_DITHER8 = 1, EPI_DIRECT8 = 2 };


template <int SRC> __device__ __forceinline__ bool src_wide(const FusedArgs &P) { return (SRC == SRC_P01X || SRC == SRC_PLANAR16) ? true : (SRC == SRC_NV12 || SRC == SRC_PLANAR8) ? false : P.bytes == 2; }
template <int SRC> __device__ __forceinline__ bool src_biplanar(const FusedArgs &P) { return (SRC == SRC_P01X || SRC == SRC_NV12) ? true : SRC != SRC_GENERIC ? false : P.planes == 2; }
template <int SRC> __device__ __forceinline__ bool src_center(const FusedArgs &P) { return SRC != SRC_GENERIC ? false : P.center_h != 0; }

// tap offsets relative to `base` (ps_interpolation_*.hlsl).  NT = 5 is the D3D11 Lanczos3 as written (quirk Q1,
// ps_interpolation_lanczos3.hlsl:33-34: the second tap re-reads the first tap's texel): taps {-2, 0, 1, 2, 3}
// with the first weight = w0 + w1 (folded by the launcher).
template <int NT>
__host__ __device__ constexpr int tap_off(int t) { return NT == 4 ? (t - 1) : NT == 6 ? (t - 2) : (t == 0 ? -2 : t - 1); }

__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
// UNORM store rounding floor(x*maxv + 0.5) for x in [0,1] without v_floor (which has no packed form):
// x*maxv + 2^23 rounds to an integer in the FMA itself (nearest-even; x*maxv can only tie at x = 0.5, where both
// conventions give (maxv+1)/2), then 2^23 comes off again — two packed instructions for two values.
// `big` = splat(2^23) held in a VGPR pair for the whole kernel (the other two operands are VGPR + SGPR; a third
// constant would be re-materialised with v_mov_b64 at every use).
__device__ __forceinline__ f2 unorm_round2(f2 x, f2 maxv2, f2 big)
{
    return pk_fma(x, maxv2, big) - big;
}
// fp32 -> fp16 (RNE) -> fp32 for a pair: v_cvt_pk_f16_f32 + two v_cvt_f32_f16
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 half_round2(f2 v) { return __builtin_convertvector(__builtin_convertvector(v, h2v), f2); }

// Wave-uniform coefficients live two to an SGPR pair; VOP3P op_sel broadcasts either half to both lanes, so a
// coefficient costs one SGPR instead of a splatted pair (the kernel is SGPR-bound otherwise: spills cost v_readlane).
//   r = w.{x|y} * b + c   [saturated to 0..1 when CLAMP]
template <int HALF, bool CLAMP>
__device__ __forceinline__ f2 pk_fma_w(f2 w, f2 b, f2 c)
{
    f2 r;
#ifdef MPCVR_NO_PK      // experiment (tools/build_nopk.sh): the same FMAs one at a time, so that nothing packed sits beside the MFMAs
    const float ws = HALF == 0 ? w.x : w.y;
    r = f2{__builtin_fmaf(ws, b.x, c.x), __builtin_fmaf(ws, b.y, c.y)};
    if (CLAMP) r = f2{__builtin_amdgcn_fmed3f(r.x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(r.y, 0.0f, 1.0f)};
    return r;
#endif
    if (HALF == 0) {
        if (CLAMP) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] clamp" : "=v"(r) : "s"(w), "v"(b), "v"(c));
        else       asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "s"(w), "v"(b), "v"(c));
    } else {
        if (CLAMP) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] clamp" : "=v"(r) : "s"(w), "v"(b), "v"(c));
        else       asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "s"(w), "v"(b), "v"(c));
    }
    return r;
}
template <int HALF>
__device__ __forceinline__ f2 pk_mul_w(f2 w, f2 b)
{
    f2 r;
#ifdef MPCVR_NO_PK
    return f2{(HALF == 0 ? w.x : w.y) * b.x, (HALF == 0 ? w.x : w.y) * b.y};
#endif
    if (HALF == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "s"(w), "v"(b));
    else           asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "s"(w), "v"(b));
    return r;
}
// tap chain: sum_t w[t] * x[t] with the weights in pairs; the last tap saturates when CLAMP
template <int NT, bool CLAMP, typename F>
__device__ __forceinline__ f2 taps(const f2 (&wp)[3], F x)
{
    f2 acc = pk_mul_w<0>(wp[0], x(0));
    acc = pk_fma_w<1, false>(wp[0], x(1), acc);
    acc = pk_fma_w<0, false>(wp[1], x(2), acc);
    if (NT == 4) return pk_fma_w<1, CLAMP>(wp[1], x(3), acc);
    acc = pk_fma_w<1, false>(wp[1], x(3), acc);
    if (NT == 5) return pk_fma_w<0, CLAMP>(wp[2], x(4), acc);
    acc = pk_fma_w<0, false>(wp[2], x(4), acc);
    return pk_fma_w<1, CLAMP>(wp[2], x(5), acc);
}

// two independent outputs in lockstep: back-to-back dependent v_pk_fma_f32 cost a wait state each (s_nop)
template <int NT, bool CLAMP, typename FA, typename FB>
__device__ __forceinline__ void taps2(const f2 (&wp)[3], FA xa, FB xb, f2 &ra, f2 &rb)
{
    f2 a = pk_mul_w<0>(wp[0], xa(0)), b = pk_mul_w<0>(wp[0], xb(0));
    a = pk_fma_w<1, false>(wp[0], xa(1), a); b = pk_fma_w<1, false>(wp[0], xb(1), b);
    a = pk_fma_w<0, false>(wp[1], xa(2), a); b = pk_fma_w<0, false>(wp[1], xb(2), b);
    if (NT == 4) { ra = pk_fma_w<1, CLAMP>(wp[1], xa(3), a); rb = pk_fma_w<1, CLAMP>(wp[1], xb(3), b); return; }
    a = pk_fma_w<1, false>(wp[1], xa(3), a); b = pk_fma_w<1, false>(wp[1], xb(3), b);
    if (NT == 5) { ra = pk_fma_w<0, CLAMP>(wp[2], xa(4), a); rb = pk_fma_w<0, CLAMP>(wp[2], xb(4), b); return; }
    a = pk_fma_w<0, false>(wp[2], xa(4), a); b = pk_fma_w<0, false>(wp[2], xb(4), b);
    ra = pk_fma_w<1, CLAMP>(wp[2], xa(5), a); rb = pk_fma_w<1, CLAMP>(wp[2], xb(5), b);
}


// N independent tap chains in lockstep (tap by tap across the chains): a dependent v_pk_fma_f32 right behind its producer costs
// wait states, N >= 4 chains in between cost none.  w(i) = the weight pairs of chain i (SGPR pairs), x(i, t) = tap t's operand.
template <int TT, int NT, bool CLAMP, int N, typename WF, typename XF>
__device__ __forceinline__ void taps_step(WF w, XF x, f2 (&acc)[N])
{
    constexpr bool last = TT == NT - 1;
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (TT == 0) acc[i] = pk_mul_w<0>(w(i)[0], x(i, 0));
        else acc[i] = pk_fma_w<TT & 1, CLAMP && last>(w(i)[TT >> 1], x(i, TT), acc[i]);
    }
    if constexpr (!last) taps_step<TT + 1, NT, CLAMP, N>(w, x, acc);
}
template <int NT, bool CLAMP, int N, typename WF, typename XF>
__device__ __forceinline__ void tapsN(WF w, XF x, f2 (&acc)[N]) { taps_step<0, NT, CLAMP, N>(w, x, acc); }

// coefficient i of a table packed two to an SGPR pair (i is a constant after unrolling)
template <bool CLAMP>
__device__ __forceinline__ f2 fma_k(const f2 *K, int i, f2 b, f2 c)
{
    return (i & 1) ? pk_fma_w<1, CLAMP>(K[i >> 1], b, c) : pk_fma_w<0, CLAMP>(K[i >> 1], b, c);
}
__device__ __forceinline__ f2 mul_k(const f2 *K, int i, f2 b) { return (i & 1) ? pk_mul_w<1>(K[i >> 1], b) : pk_mul_w<0>(K[i >> 1], b); }
// (Round 4 tried the Dolby Vision variants' three matrices — ycc_to_rgb, the LMS step, 2020 -> 709 — as uncontracted products summed
// left to right, the way mat3_mul and the oracle write them: the count of channels beyond 1 LSB on the whole-frame cases did not move
// (16 / 12 / 10 per 2 M pixels, profiles/r04/parity_identical_channels.jsonl of call 1), and the kernels lost 10-15 %.  What moved it was the
// PQ EOTF table — see EOTF_N.  The fused chains stay.)

// raw codes of one 2x2 block (cols Xg, Xg+1; two source rows), prefetched one iteration ahead
struct Raw {
    uint32_t y[2];           // luma of the two rows: 2 px each (16-bit: one dword; 8-bit: low 16 bits)
    uint32_t c[2][3];        // [chroma row n, n+1][cols c0-1, c0, c0+1]: packed (U | V<<16) codes, shared by both luma rows
};

__device__ __forceinline__ uint32_t ld_u8(gcptr p) { return *p; }
__device__ __forceinline__ uint32_t ld_u16(gcptr p) { return *(const __attribute__((address_space(1))) uint16_t *)p; }
__device__ __forceinline__ uint32_t ld_u32(gcptr p) { return *(const __attribute__((address_space(1))) uint32_t *)p; }

// Addressing: every access is (wave-uniform row base, SGPR pair) + (per-lane 32-bit byte offset that does not change
// over the rows), so the loads/stores take the saddr form and the loop spends no VALU on 64-bit pointer arithmetic.
// Row offsets are 32-bit products (the launcher refuses surfaces of 4 GiB and more).
// `opaque` hides a loop-invariant 32-bit lane offset from LICM: the zero-extension then stays next to the access and
// instruction selection folds (uniform base + zext(offset)) into the saddr form instead of a 64-bit VALU add per access.
__device__ __forceinline__ uint32_t opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }

struct RawAddr {
    uint32_t yoff;            // luma: byte offset of column Xg inside a row
    uint32_t coff[3];         // chroma columns c0-1, c0, c0+1 (clamp addressing), byte offset inside a chroma row
};

template <int SRC>
__device__ __forceinline__ void make_raw_addr(const FusedArgs &P, int Xg, RawAddr &ra)
{
    const int sx0 = P.rect_l + Xg, c0 = P.sub444 ? sx0 : sx0 >> 1;        // 4:4:4: texels c0, c0+1 are the block's own two columns
    if (SRC == SRC_GENERIC && P.packed422) {       // texel c0 carries the block's two luma samples and its own chroma; c0+1 the neighbour's chroma
        const int tb = P.bytes == 2 ? 8 : 4;
        ra.yoff = (uint32_t)(tb * c0); ra.coff[0] = 0; ra.coff[1] = ra.yoff; ra.coff[2] = (uint32_t)(tb * clampi(c0 + 1, 0, P.cw - 1));
        return;
    }
    if (SRC == SRC_GENERIC && P.packed444) {       // the block's two columns are two consecutive texels
        ra.yoff = (uint32_t)((P.packed444 == 3 ? 8 : 4) * sx0); ra.coff[0] = ra.coff[1] = ra.coff[2] = 0;
        return;
    }
    const int yb = src_wide<SRC>(P) ? 2 : 1;
    const int cb = src_biplanar<SRC>(P) ? 2 * yb : yb;
    ra.yoff = (uint32_t)(yb * sx0);
#pragma unroll
    for (int i = 0; i < 3; i++) ra.coff[i] = (uint32_t)(cb * clampi(c0 - 1 + i, 0, P.cw - 1));
}

// chroma texel as U | V << 16 (raw codes); pu/pv = row bases
template <int SRC>
__device__ __forceinline__ uint32_t ld_uv(const FusedArgs &P, gcptr pu, gcptr pv, uint32_t off)
{
    if (src_biplanar<SRC>(P)) {
        if (src_wide<SRC>(P)) return ld_u32(pu + off);
        const uint32_t d = ld_u16(pu + off);
        return (d & 0xffu) | ((d >> 8) << 16);
    }
    if (src_wide<SRC>(P)) return ld_u16(pu + off) | (ld_u16(pv + off) << 16);
    return ld_u8(pu + off) | (ld_u8(pv + off) << 16);
}

// vertical chroma position of source row sy (Shaders.cpp:118-138): v' = (sy+0.5)/2 [+0.25 co-sited] - 0.5, kept in
// QUARTER chroma rows as an integer (4v' = 2sy - 1 [+1]) so that the whole siting computation stays on the scalar unit
// (4:2:2: chroma rows are luma rows — v' = sy exactly, so a row pair takes row 0 from chroma row sy0 and row 1 from sy0 + 1)
// (CHROMA_Nearest at 4:2:0: row sy reads chroma row sy >> 1 whole — v' = sy >> 1, so of an (odd, odd + 1) pair row 0 takes row n, row 1 row n + 1)
// (FillFusedArgs folds the three cases into coefficients: {4, 0, 0}, {0, 2, 0}, {2, 0, v_off4 - 1} — the nested selects on kernel
// arguments compiled to a chain of scalar branches, four times per iteration of every fused kernel)
__device__ __forceinline__ int chroma_v4(const FusedArgs &P, int sy) { return P.vk1 * sy + P.vk2 * (sy & ~1) + P.vk3; }
// fr/4 for fr = 0..4 as a float built from integer selects (wave-uniform => SGPR; no v_cvt/v_mul per iteration)
__device__ __forceinline__ float quarter(int fr)
{
    // float bits of fr/4 = (one byte of a 40-bit table) << 22: 0.25 = 0xFA<<22, 0.5 = 0xFC<<22, 0.75 = 0xFD<<22, 1 = 0xFE<<22
    const uint64_t table = 0xFEFDFCFA00ull;
    return __builtin_bit_cast(float, (uint32_t)((table >> (8 * fr)) & 0xffu) << 22);
}

// y0,y1: the two (clamped) rect rows of the block.
// The two luma rows of an iteration are (odd, odd+1) source rows — or the same row twice where the rect clamps —
// (rect top and segment starts are even, host-checked), so for every siting both take their chroma from the same
// two chroma rows n = floor(v'(row 0)) and n+1.
template <int SRC>
__device__ __forceinline__ void load_raw(const FusedArgs &P, gcptr py, const RawAddr &ra, int y0, int y1, Raw &r)
{
    const int sy0 = P.rect_t + y0, sy1 = P.rect_t + y1;
    const gcptr ry0 = py + (uint32_t)sy0 * (uint32_t)P.pitch_y, ry1 = py + (uint32_t)sy1 * (uint32_t)P.pitch_y;
    if (SRC == SRC_GENERIC && P.packed422) {
        // packed 4:2:2 (Shaders.cpp:195-229): the even pixel takes the texel's own chroma, the odd pixel the mean with the next
        // texel's (clamp addressing) — convert_block's 4:2:2 rule with c[.][1] = own, c[.][2] = next; chroma rows = luma rows
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const gcptr row = rr ? ry1 : ry0;
            uint32_t own[4], nxt[4];
            if (P.bytes == 2) {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));      // rows and samples are dword aligned, no more is promised
                const u32x2 t = *(const __attribute__((address_space(1))) u32x2 *)(row + opaque(ra.yoff));
                const u32x2 n = *(const __attribute__((address_space(1))) u32x2 *)(row + opaque(ra.coff[2]));
                own[0] = t.x & 0xffffu; own[1] = t.x >> 16; own[2] = t.y & 0xffffu; own[3] = t.y >> 16;
                nxt[0] = n.x & 0xffffu; nxt[1] = n.x >> 16; nxt[2] = n.y & 0xffffu; nxt[3] = n.y >> 16;
            } else {
                const uint32_t t = ld_u32(row + opaque(ra.yoff)), n = ld_u32(row + opaque(ra.coff[2]));
#pragma unroll
                for (int k = 0; k < 4; k++) { own[k] = (t >> (8 * k)) & 0xffu; nxt[k] = (n >> (8 * k)) & 0xffu; }
            }
            // wave-uniform component positions: selects, not indexed registers
            auto pick = [](const uint32_t (&v)[4], int k) { return k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : v[3]; };
            r.y[rr] = pick(own, P.ci[0]) | (pick(own, P.ci[2]) << (P.bytes == 2 ? 16 : 8));
            r.c[rr][0] = 0;
            r.c[rr][1] = pick(own, P.ci[1]) | (pick(own, P.ci[3]) << 16);
            r.c[rr][2] = pick(nxt, P.ci[1]) | (pick(nxt, P.ci[3]) << 16);
        }
        return;
    }
    if (SRC == SRC_GENERIC && P.packed444) {
        // packed 4:4:4 (Shaders.cpp:186-193: color.zyxw for AYUV, .yxzw for Y410 / Y416): a texel per pixel, no chroma filter
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const gcptr row = rr ? ry1 : ry0;
            uint32_t t[2][4];
            if (P.packed444 == 3) {
                const u32x4 q = *(const __attribute__((address_space(1))) u32x4 *)(row + opaque(ra.yoff));
                t[0][0] = q.x & 0xffffu; t[0][1] = q.x >> 16; t[0][2] = q.y & 0xffffu; t[0][3] = q.y >> 16;
                t[1][0] = q.z & 0xffffu; t[1][1] = q.z >> 16; t[1][2] = q.w & 0xffffu; t[1][3] = q.w >> 16;
            } else {
                const u32x2 q = *(const __attribute__((address_space(1))) u32x2 *)(row + opaque(ra.yoff));
                const int bits = P.packed444 == 2 ? 10 : 8;
                const uint32_t mask = (1u << bits) - 1u;
#pragma unroll
                for (int k = 0; k < 4; k++) { t[0][k] = (q.x >> (bits * k)) & mask; t[1][k] = (q.y >> (bits * k)) & mask; }
            }
            auto pick = [](const uint32_t (&v)[4], int k) { return k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : v[3]; };
            r.y[rr] = pick(t[0], P.ci[0]) | (pick(t[1], P.ci[0]) << (P.bytes == 2 ? 16 : 8));
            r.c[rr][0] = 0;
            r.c[rr][1] = pick(t[0], P.ci[1]) | (pick(t[0], P.ci[2]) << 16);
            r.c[rr][2] = pick(t[1], P.ci[1]) | (pick(t[1], P.ci[2]) << 16);
        }
        return;
    }
    r.y[0] = src_wide<SRC>(P) ? ld_u32(ry0 + opaque(ra.yoff)) : ld_u16(ry0 + opaque(ra.yoff));
    r.y[1] = src_wide<SRC>(P) ? ld_u32(ry1 + opaque(ra.yoff)) : ld_u16(ry1 + opaque(ra.yoff));
    if (SRC == SRC_GENERIC && P.gray) {            // float4 color = tex.Sample of an R8 / R16 texture (Shaders.cpp:184): no chroma
#pragma unroll
        for (int i = 0; i < 3; i++) r.c[0][i] = r.c[1][i] = 0;
        return;
    }
    const int n = chroma_v4(P, sy0) >> 2;
    const uint32_t oA = (uint32_t)clampi(n, 0, P.ch - 1) * (uint32_t)P.pitch_c, oB = (uint32_t)clampi(n + 1, 0, P.ch - 1) * (uint32_t)P.pitch_c;
    const gcptr pu = py + P.off_u, pv = src_biplanar<SRC>(P) ? pu : py + P.off_v;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (i == 0 && !src_center<SRC>(P)) { r.c[0][0] = r.c[1][0] = 0; continue; }
        r.c[0][i] = ld_uv<SRC>(P, pu + oA, pv + oA, opaque(ra.coff[i]));
        r.c[1][i] = ld_uv<SRC>(P, pu + oB, pv + oB, opaque(ra.coff[i]));
    }
}


// ---- Dolby Vision reshaping for a 2x2 block (ShaderDoviReshape / ShaderDoviReshapePoly, Shaders.cpp:531-589) ----
// dovi_reshape (vp_device.h) serves one pixel and reads every field of the curves where it needs it; run four times per block
// out of the LDS copy that is ~30 dependent LDS round trips per pixel.  Here the wave-uniform parts (pivots, method flags) are
// loaded ONCE per wave by scalar loads, the piece search of the 12 (pixel, component) values runs on SGPR pivots, and the 12
// coefficient reads are issued together.  Same arithmetic: pieces by `s < pivot`, (c2*s + c1)*s + c0, reshape_mmr for MMR pieces.
struct DoviRegs {
    float pv[3][7];
    uint32_t methods[3], mmr_single[3], min_order[3], max_order[3];
    int has_mmr;
};
// Workgroups are dealt round-robin to the 8 XCDs in launch order (x fastest, then z here: the grids are (items, 1, frames)), and each
// XCD has its own L2.  The waves of neighbouring strips read the same 128-byte lines at their seams and neighbouring segments the same
// source rows: give XCD k the k-th contiguous range of the launch's (frame, item) sequence instead of every 8th workgroup of it, so a
// seam is fetched once (round 4: the periodic kernel's FETCH_SIZE was 1.11x the algorithmic bytes with the plain mapping).
// A bijection on [0, gridDim.x * gridDim.z) for any size; wave-uniform scalar arithmetic.
__device__ __forceinline__ void xcd_contiguous_block(int &bx, int &bz)
{
    const int gx = (int)gridDim.x, total = gx * (int)gridDim.z;
    const int g = (int)blockIdx.x + gx * (int)blockIdx.z;
    const int per = total >> 3, rem = total & 7, xcd = g & 7, idx = g >> 3;
    const int gp = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
    bz = gp / gx;
    bx = gp - bz * gx;
}

template <typename T> using dv_cptr = const __attribute__((address_space(4))) T *;
__device__ __forceinline__ void load_dovi_regs(const DoviParams *g, DoviRegs &R)
{
    const dv_cptr<DoviParams> c = (dv_cptr<DoviParams>)(uintptr_t)g;       // read-only for the launch: scalar loads
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int i = 0; i < 7; i++) R.pv[k][i] = c->curves[k].pivots[i];
        R.methods[k] = c->curves[k].methods; R.mmr_single[k] = c->curves[k].mmr_single;
        R.min_order[k] = c->curves[k].min_order; R.max_order[k] = c->curves[k].max_order;
    }
    R.has_mmr = c->has_mmr;
}
// reshape_mmr (Shaders.cpp:734-762) for a component whose MMR pieces all share ONE weight set of ONE order (mmr_single, min_order ==
// max_order: what a chroma curve with a single MMR piece — the usual profile 5 / 8 stream — packs to).  The weights are wave-uniform:
// broadcast reads of the LDS copy, a float4 row at a time, either half of a register pair broadcast to both pixels by op_sel; the
// pixels go two to a packed FMA as the block's (row 0, row 1) pairs, one block column at a time.  Only the seven base monomials
// {x, y, z, xy, xz, yz, xyz} of one column are kept — squares and cubes are formed where they are summed — instead of 2 x 21 terms
// (which cost the kernel half its occupancy when tried; SGPR-held weights from scalar loads were tried too: the kernel has no SGPRs to
// spare and the loads' latency stayed exposed).  Same terms and weights as the shader;
// the summation order differs (rounding noise only).
// r = w.{x|y} * b + c with the weight pair in VGPRs (a broadcast LDS read), either half broadcast to both pixels by op_sel
template <int HALF>
__device__ __forceinline__ f2 pk_fma_wv(f2 w, f2 b, f2 c)
{
    f2 r;
#ifdef MPCVR_NO_PK
    return f2{__builtin_fmaf(HALF == 0 ? w.x : w.y, b.x, c.x), __builtin_fmaf(HALF == 0 ? w.x : w.y, b.y, c.y)};
#endif
    if (HALF == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(w), "v"(b), "v"(c));
    else           asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(w), "v"(b), "v"(c));
    return r;
}
// (x, y, z) = the block column's (Y, U, V) pairs; acc starts as the pieces' constant coefficient.  Weight rows (one float4 each, read
// where they are used — all lanes the same LDS address — so that only a row or two is live): [0] 3 linear, [1] 4 cross, [2] 3 squares,
// [3] 4 squared cross, [4] 3 cubes, [5] 4 cubed cross
template <int LV>
__device__ __forceinline__ f2 mmr_level(const DoviParams *DL, int k, const f2 (&b)[7], f2 acc)
{
    const float4 w3 = *reinterpret_cast<const float4 *>(DL->curves[k].mmr[2 * LV]), w4 = *reinterpret_cast<const float4 *>(DL->curves[k].mmr[2 * LV + 1]);
    const f2 W[4] = {f2{w3.x, w3.y}, f2{w3.z, w3.w}, f2{w4.x, w4.y}, f2{w4.z, w4.w}};
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const f2 p = LV == 0 ? b[i] : LV == 1 ? b[i] * b[i] : (b[i] * b[i]) * b[i];
        const int j = i < 3 ? i : i + 1;             // position in the two rows laid end to end (w3.w is padding)
        acc = (j & 1) ? pk_fma_wv<1>(W[j >> 1], p, acc) : pk_fma_wv<0>(W[j >> 1], p, acc);    // (two chains were no faster)
    }
    return acc;
}
__device__ __forceinline__ f2 mmr_component(const DoviParams *DL, int k, uint32_t order, f2 x, f2 y, f2 z, f2 acc)
{
    const f2 xy = x * y;
    const f2 b[7] = {x, y, z, xy, x * z, y * z, xy * z};
    if (order == 3)         // the usual order: one basic block, so that the weight reads of all three levels can be issued ahead of the sums
        return mmr_level<2>(DL, k, b, mmr_level<1>(DL, k, b, mmr_level<0>(DL, k, b, acc)));
    acc = mmr_level<0>(DL, k, b, acc);
    if (order >= 2) acc = mmr_level<1>(DL, k, b, acc);
    return acc;
}

// U and V together — the usual stream: each chroma curve is ONE order-N MMR piece over the whole range (no pivots to search, no
// polynomial pieces, the constant term wave-uniform).  Every monomial is formed once and goes straight into both sums (two independent
// FMA chains, nothing stored): 4 + 14 multiplies + 2 x 21 FMAs per block column instead of 2 x (4 + 14 + 21).
template <int LV>
__device__ __forceinline__ void mmr_level_uv(const DoviParams *DL, const f2 (&b)[7], f2 &au, f2 &av)
{
    const float4 u3 = *reinterpret_cast<const float4 *>(DL->curves[1].mmr[2 * LV]), u4 = *reinterpret_cast<const float4 *>(DL->curves[1].mmr[2 * LV + 1]);
    const float4 v3 = *reinterpret_cast<const float4 *>(DL->curves[2].mmr[2 * LV]), v4 = *reinterpret_cast<const float4 *>(DL->curves[2].mmr[2 * LV + 1]);
    const f2 WU[4] = {f2{u3.x, u3.y}, f2{u3.z, u3.w}, f2{u4.x, u4.y}, f2{u4.z, u4.w}};
    const f2 WV[4] = {f2{v3.x, v3.y}, f2{v3.z, v3.w}, f2{v4.x, v4.y}, f2{v4.z, v4.w}};
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const f2 p = LV == 0 ? b[i] : LV == 1 ? b[i] * b[i] : (b[i] * b[i]) * b[i];
        const int j = i < 3 ? i : i + 1;
        au = (j & 1) ? pk_fma_wv<1>(WU[j >> 1], p, au) : pk_fma_wv<0>(WU[j >> 1], p, au);
        av = (j & 1) ? pk_fma_wv<1>(WV[j >> 1], p, av) : pk_fma_wv<0>(WV[j >> 1], p, av);
    }
}
__device__ __forceinline__ void mmr_pair(const DoviParams *DL, uint32_t order, f2 x, f2 y, f2 z, f2 &au, f2 &av)
{
    const f2 xy = x * y;
    const f2 b[7] = {x, y, z, xy, x * z, y * z, xy * z};
    if (order == 3) { mmr_level_uv<0>(DL, b, au, av); mmr_level_uv<1>(DL, b, au, av); mmr_level_uv<2>(DL, b, au, av); return; }
    mmr_level_uv<0>(DL, b, au, av);
    if (order >= 2) mmr_level_uv<1>(DL, b, au, av);
}

// Y, U, V: [column] as (row 0, row 1) pairs of 0..1 values; reshaped in place.  DL = the LDS copy (coefficients, MMR weights)
__device__ __forceinline__ void dovi_reshape_block(const DoviRegs &R, const DoviParams *DL, f2 (&Y)[2], f2 (&U)[2], f2 (&V)[2])
{
    float s[3][4];
#pragma unroll
    for (int p = 0; p < 4; p++) { s[0][p] = Y[p >> 1][p & 1]; s[1][p] = U[p >> 1][p & 1]; s[2][p] = V[p >> 1][p & 1]; }
    // the shared-weight MMR components (see mmr_component): wave-uniform conditions
    bool fast_mmr[3], any_fast = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        fast_mmr[k] = R.has_mmr && (R.methods[k] & DOVI_RESHAPE_MMR) && R.mmr_single[k] && R.min_order[k] == R.max_order[k] &&
                      R.max_order[k] >= 1 && R.max_order[k] <= 3;
        any_fast = any_fast || fast_mmr[k];
    }
    if (!any_fast) {
        // polynomial curves (and MMR pieces with weight sets of their own): the 12 coefficient reads are issued together
        float4 co[3][4];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            int piece[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 7; i++) {
                if (R.pv[k][i] > 2.0f) break;              // unused pivots are 1e9 (PackDoviCurves): wave-uniform exit
#pragma unroll
                for (int p = 0; p < 4; p++) piece[p] += (s[k][p] >= R.pv[k][i]) ? 1 : 0;      // == the nested `s < pivot` search on sorted pivots
            }
#pragma unroll
            for (int p = 0; p < 4; p++) co[k][p] = *reinterpret_cast<const float4 *>(DL->curves[k].coeffs[piece[p]]);
        }
        float out[3][4];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const bool any_mmr = R.has_mmr && (R.methods[k] & DOVI_RESHAPE_MMR);       // wave-uniform
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const float x = s[k][p];
                float r = (co[k][p].z * x + co[k][p].y) * x + co[k][p].x;
                if (any_mmr) {
                    const bool poly = R.methods[k] == DOVI_RESHAPE_POLY + DOVI_RESHAPE_MMR && co[k][p].w == 0.0f;
                    if (!poly) {
                        const float c4[4] = {co[k][p].x, co[k][p].y, co[k][p].z, co[k][p].w};
                        r = dovi_reshape_mmr(DL->curves[k], c4, f3{s[0][p], s[1][p], s[2][p]});
                    }
                }
                out[k][p] = saturate(r);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) { Y[p >> 1][p & 1] = out[0][p]; U[p >> 1][p & 1] = out[1][p]; V[p >> 1][p & 1] = out[2][p]; }
        return;
    }
    // at least one shared-weight MMR component: one component at a time (its four coefficient sets are the only ones in registers)
    float out[3][4];
    // both chroma curves a single MMR piece of the same order and nothing else: U and V in one pass over the monomials
    const bool uv_pair = fast_mmr[1] && fast_mmr[2] && R.methods[1] == DOVI_RESHAPE_MMR && R.methods[2] == DOVI_RESHAPE_MMR &&
                         R.pv[1][0] > 2.0f && R.pv[2][0] > 2.0f && R.max_order[1] == R.max_order[2];
    if (uv_pair) {
        const float cu = DL->curves[1].coeffs[0][0], cv = DL->curves[2].coeffs[0][0];
#pragma unroll
        for (int col = 0; col < 2; col++) {
            f2 au = splat(cu), av = splat(cv);
            mmr_pair(DL, R.max_order[1], f2{s[0][2 * col], s[0][2 * col + 1]}, f2{s[1][2 * col], s[1][2 * col + 1]}, f2{s[2][2 * col], s[2][2 * col + 1]}, au, av);
            out[1][2 * col] = saturate(au.x); out[1][2 * col + 1] = saturate(au.y);
            out[2][2 * col] = saturate(av.x); out[2][2 * col + 1] = saturate(av.y);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (uv_pair && k > 0) break;
        int piece[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 7; i++) {
            if (R.pv[k][i] > 2.0f) break;
#pragma unroll
            for (int p = 0; p < 4; p++) piece[p] += (s[k][p] >= R.pv[k][i]) ? 1 : 0;
        }
        float4 co[4];
#pragma unroll
        for (int p = 0; p < 4; p++) co[p] = *reinterpret_cast<const float4 *>(DL->curves[k].coeffs[piece[p]]);
        const bool any_mmr = R.has_mmr && (R.methods[k] & DOVI_RESHAPE_MMR);
        f2 fast[2];
        if (fast_mmr[k]) {
#pragma unroll
            for (int col = 0; col < 2; col++)
                fast[col] = mmr_component(DL, k, R.max_order[k], f2{s[0][2 * col], s[0][2 * col + 1]}, f2{s[1][2 * col], s[1][2 * col + 1]},
                                          f2{s[2][2 * col], s[2][2 * col + 1]}, f2{co[2 * col].x, co[2 * col + 1].x});
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const float x = s[k][p];
            float r = (co[p].z * x + co[p].y) * x + co[p].x;
            const bool poly = !any_mmr || (R.methods[k] == DOVI_RESHAPE_POLY + DOVI_RESHAPE_MMR && co[p].w == 0.0f);
            if (fast_mmr[k]) r = poly ? r : fast[p >> 1][p & 1];
            else if (!poly) {
                const float c4[4] = {co[p].x, co[p].y, co[p].z, co[p].w};
                r = dovi_reshape_mmr(DL->curves[k], c4, f3{s[0][p], s[1][p], s[2][p]});
            }
            out[k][p] = saturate(r);
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the next component's coefficient reads behind this one's arithmetic
    }
#pragma unroll
    for (int p = 0; p < 4; p++) { Y[p >> 1][p & 1] = out[0][p]; U[p >> 1][p & 1] = out[1][p]; V[p >> 1][p & 1] = out[2][p]; }
}

// ---- the exact form of the block convert (FusedArgs::exact_cv) ----
// What the generated shader computes (Shaders.cpp:231-325 sampling, :819-820 matrix) and the store into m_TexConvertOutput does, in the
// shader's own expression shapes: texels as 0..1 values (correctly rounded code / maxv), the bilinear chroma sample horizontally first and
// then vertically with both products rounded, the matrix as three rounded products summed left to right plus the constant, the UNORM
// store as floor(saturate(x) * maxv + 0.5).  Nothing may contract here: out[column][channel] is code / maxv of EXACTLY the code the
// oracle stores, as (row 0, row 1) pairs, so a kernel's own `x * maxv + 2^23` reads the same code back.
#pragma clang fp contract(off)
__device__ __forceinline__ f2 xnorm2(f2 code, float d, float r)
{
    // unorm_div (vp_device.h) on a pair; (d, r) = (maxv / 2^shift, 2^shift / maxv) divides (code << shift) by maxv: the power of two
    // scales every intermediate exactly
    const f2 q = code * splat(r);
    return pk_fma(pk_fma(-q, splat(d), code), splat(r), q);
}
// the same quotient for an 8-bit code in two operations: 1/255 as a float pair (hi + lo to 2^-56), code * hi + fl(code * lo) rounded once —
// the correctly rounded code / 255 for every code 0 .. 255 (checked exhaustively: tests/test_host_logic.py::test_two_step_quotient_of_8bit_codes)
constexpr float kInv255Hi = 0x1.010102p-8f, kInv255Lo = -0x1.fdfdfep-33f;
__device__ __forceinline__ f2 xnorm2_u8(f2 code) { return pk_fma(code, splat(kInv255Hi), code * splat(kInv255Lo)); }
template <int SRC>
__device__ __forceinline__ f2 xnorm2_src(f2 code, float d, float r)
{
    if constexpr (SRC == SRC_NV12 || SRC == SRC_PLANAR8) return xnorm2_u8(code);
    else return xnorm2(code, d, r);
}
// what convert_block leaves in out[][]: the 0..1 value, or — the exact form only, which rounds to the internal format's code itself — that
// code as a float or as an integer's bits: a consumer that wants the code does not multiply it back (OUT_NORM: floor, * 1/maxv there and
// * maxv + 2^23 here per value)
enum { OUT_NORM = 0, OUT_CODE_F = 1, OUT_CODE_I = 2 };
// (Y, U, V) of the block as 0..1 values, [column] as (row 0, row 1) pairs -> out
template <int OUTK>
__device__ __forceinline__ void exact_matrix_store(const FusedArgs &P, const f2 (&Y)[2], const f2 (&U)[2], const f2 (&V)[2], f2 out[2][3])
{
    const f2 half2 = splat(0.5f), mx = splat(P.maxv), inv = splat(P.inv_maxv);
#pragma unroll
    for (int col = 0; col < 2; col++)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            f2 s = splat(P.xm[3 * ch]) * Y[col] + splat(P.xm[3 * ch + 1]) * U[col];
            s = s + splat(P.xm[3 * ch + 2]) * V[col];
            // + cm_c and saturate: the packed add's clamp modifier (the compiler spends a v_max per value on it)
            const f2 cc = splat(P.xc[ch]);
            asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(s) : "v"(s), "v"(cc));
            const f2 t = s * mx + half2;                       // (>= 0.5: the truncating conversion is the floor)
            if constexpr (OUTK == OUT_CODE_I) out[col][ch] = f2{__uint_as_float((uint32_t)t.x), __uint_as_float((uint32_t)t.y)};
            else if constexpr (OUTK == OUT_CODE_F) out[col][ch] = f2{__builtin_floorf(t.x), __builtin_floorf(t.y)};
            else out[col][ch] = f2{__builtin_floorf(t.x), __builtin_floorf(t.y)} * inv;
        }
}
template <int SRC, int OUTK>
__device__ __forceinline__ void convert_block_exact(const FusedArgs &P, const Raw &r, int sy0, int sy1, f2 out[2][3])
{
    const int n4 = chroma_v4(P, sy0) & ~3;
    const int fr0 = chroma_v4(P, sy0) - n4, fr1 = chroma_v4(P, sy1) - n4;
    const f2 w1 = f2{quarter(fr0), quarter(fr1)}, w0 = f2{quarter(4 - fr0), quarter(4 - fr1)};    // wy and 1 - wy of (row 0, row 1): quarters, exact
    f2 Un[3], Vn[3];                              // chroma columns c0-1, c0, c0+1 as (chroma row n, row n+1) pairs, 0..1
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (i == 0 && !src_center<SRC>(P)) { Un[0] = Vn[0] = splat(0.0f); continue; }
        Un[i] = xnorm2_src<SRC>(f2{(float)(r.c[0][i] & 0xffffu), (float)(r.c[1][i] & 0xffffu)}, P.xdc, P.xrc);
        Vn[i] = xnorm2_src<SRC>(f2{(float)(r.c[0][i] >> 16), (float)(r.c[1][i] >> 16)}, P.xdc, P.xrc);
    }
    f2 Hu[2], Hv[2];                              // c00 * (1 - wx) + c10 * wx of the even / odd luma column, rows (n, n+1)
    if (src_center<SRC>(P)) {                     // MPEG-1: wx = 0.75 (even column, texels c0-1, c0), 0.25 (odd column, texels c0, c0+1)
        Hu[0] = Un[0] * splat(0.25f) + Un[1] * splat(0.75f); Hv[0] = Vn[0] * splat(0.25f) + Vn[1] * splat(0.75f);
        Hu[1] = Un[1] * splat(0.75f) + Un[2] * splat(0.25f); Hv[1] = Vn[1] * splat(0.75f) + Vn[2] * splat(0.25f);
    } else {                                      // wx = 0 (even column: c00 * 1 + c10 * 0 = c00), 0.5 (odd; {0, 1} at 4:4:4, {1, 0} nearest)
        Hu[0] = Un[1]; Hv[0] = Vn[1];
        // (the weights are 0, 0.5 or 1: both products are exact, so the FMA rounds once like the shader's add)
        Hu[1] = pk_fma(Un[2], splat(P.cw_next), Un[1] * splat(P.cw_own));
        Hv[1] = pk_fma(Vn[2], splat(P.cw_next), Vn[1] * splat(P.cw_own));
    }
    f2 Y[2], U[2], V[2];
#pragma unroll
    for (int col = 0; col < 2; col++) {           // top * (1 - wy) + bot * wy for (row 0, row 1)
        U[col] = splat(Hu[col].x) * w0 + splat(Hu[col].y) * w1;
        V[col] = splat(Hv[col].x) * w0 + splat(Hv[col].y) * w1;
    }
    if (src_wide<SRC>(P)) {
        Y[0] = xnorm2_src<SRC>(f2{(float)(r.y[0] & 0xffffu), (float)(r.y[1] & 0xffffu)}, P.xdy, P.xry);
        Y[1] = xnorm2_src<SRC>(f2{(float)(r.y[0] >> 16), (float)(r.y[1] >> 16)}, P.xdy, P.xry);
    } else {
        Y[0] = xnorm2_src<SRC>(f2{(float)(r.y[0] & 0xffu), (float)(r.y[1] & 0xffu)}, P.xdy, P.xry);
        Y[1] = xnorm2_src<SRC>(f2{(float)((r.y[0] >> 8) & 0xffu), (float)((r.y[1] >> 8) & 0xffu)}, P.xdy, P.xry);
    }
    exact_matrix_store<OUTK>(P, Y, U, V, out);
}
#pragma clang fp contract(fast)

// The 2x2 block: 4:2:0 bilinear chroma + matrix (+ tail) for (even, odd column) x (row 0, row 1).
// ShaderGetPixels' CHROMA_Bilinear branch (Shaders.cpp:265-270,319-325): same sample positions and weights,
// evaluated in code units (vertical lerp first), UNORM scale folded into the matrix.  out[column][ch] = the channel as
// a (row 0, row 1) pair — the layout LDS slice A wants — saturated (every continuation, tail or UNORM store,
// saturates first).
template <int TAIL, int SRC, int DV>
__device__ __forceinline__ void convert_block_yuv(const FusedArgs &P, const f2 (&MM)[5], const f2 (&GG)[5], const f2 (&CC)[3], f2 (&Ycol)[2], f2 (&Ucol)[2], f2 (&Vcol)[2],
                                                  const f2 *T, f2 out[2][3], const DoviParams *DL, const float *TE, const DoviRegs *DR);

template <int TAIL, int SRC, int DV = DV_NONE, int XC = XC_NEVER, int OUTK = OUT_NORM>
__device__ __forceinline__ void convert_block(const FusedArgs &P, const f2 (&MM)[5], const f2 (&GG)[5], const f2 (&CC)[3], const Raw &r, int sy0, int sy1, const f2 *T, f2 out[2][3],
                                              const DoviParams *DL = nullptr, const float *TE = nullptr, const DoviRegs *DR = nullptr)
{
    if constexpr (XC == XC_ALWAYS) { convert_block_exact<SRC, OUTK>(P, r, sy0, sy1, out); return; }
    // vertical weights of chroma rows n (w0) and n+1 (w1) for (row 0, row 1): wave-uniform, one SGPR pair each
    const int n4 = chroma_v4(P, sy0) & ~3;                 // 4 * floor(v'(row 0))
    const int fr0 = chroma_v4(P, sy0) - n4, fr1 = chroma_v4(P, sy1) - n4;     // 0..4 quarters
    const f2 w1 = f2{quarter(fr0), quarter(fr1)}, w0 = f2{quarter(4 - fr0), quarter(4 - fr1)};
    f2 Uc[3], Vc[3];                              // U, V at chroma columns c0-1, c0, c0+1 as (row 0, row 1) pairs
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float tu = (float)(r.c[0][i] & 0xffffu), tv = (float)(r.c[0][i] >> 16);
        const float bu = (float)(r.c[1][i] & 0xffffu), bv = (float)(r.c[1][i] >> 16);
        Uc[i] = pk_fma(splat(bu), w1, splat(tu) * w0);
        Vc[i] = pk_fma(splat(bv), w1, splat(tv) * w0);
    }
    f2 Ycol[2], Ucol[2], Vcol[2];                 // even and odd luma column
    if (src_wide<SRC>(P)) {
        Ycol[0] = f2{(float)(r.y[0] & 0xffffu), (float)(r.y[1] & 0xffffu)};
        Ycol[1] = f2{(float)(r.y[0] >> 16), (float)(r.y[1] >> 16)};
    } else {
        Ycol[0] = f2{(float)(r.y[0] & 0xffu), (float)(r.y[1] & 0xffu)};
        Ycol[1] = f2{(float)((r.y[0] >> 8) & 0xffu), (float)((r.y[1] >> 8) & 0xffu)};
    }
    if (src_center<SRC>(P)) {                     // u' = sx/2 - 0.25 (MPEG-1 siting runs through the generic variant)
        Ucol[0] = pk_fma(Uc[1], splat(0.75f), Uc[0] * splat(0.25f)); Vcol[0] = pk_fma(Vc[1], splat(0.75f), Vc[0] * splat(0.25f));
        Ucol[1] = pk_fma(Uc[2], splat(0.25f), Uc[1] * splat(0.75f)); Vcol[1] = pk_fma(Vc[2], splat(0.25f), Vc[1] * splat(0.75f));
    } else {                                      // u' = sx/2; 4:4:4: the odd column has its own sample; nearest: the block's one texel
        Ucol[0] = Uc[1]; Vcol[0] = Vc[1];
        // odd column: (c0 + c1) / 2 — or c1 alone (4:4:4) or c0 alone (nearest): wave-uniform weights {0.5, 0.5} / {0, 1} / {1, 0}
        // instead of selects on the kernel arguments (eight v_cndmask per block in every variant when tried); exact in all three cases
        const f2 cwa = splat(P.cw_own), cwb = splat(P.cw_next);
        Ucol[1] = pk_fma(Uc[2], cwb, Uc[1] * cwa);
        Vcol[1] = pk_fma(Vc[2], cwb, Vc[1] * cwa);
    }
    convert_block_yuv<TAIL, SRC, DV>(P, MM, GG, CC, Ycol, Ucol, Vcol, T, out, DL, TE, DR);
}

// everything behind ShaderGetPixels: (Dolby Vision reshaping,) colour matrix, HDR tail — for the (Y, U, V) of the block's four pixels in
// code units, [column] as (row 0, row 1) pairs, whichever chroma filter produced them
template <int TAIL, int SRC, int DV>
__device__ __forceinline__ void convert_block_yuv(const FusedArgs &P, const f2 (&MM)[5], const f2 (&GG)[5], const f2 (&CC)[3], f2 (&Ycol)[2], f2 (&Ucol)[2], f2 (&Vcol)[2],
                                                  const f2 *T, f2 out[2][3], const DoviParams *DL, const float *TE, const DoviRegs *DR)
{
    if (DV != DV_NONE) {
        // ShaderDoviReshape[Poly] (Shaders.cpp:531-589,786-792) on the sampled (Y, U, V) of every pixel, as 0..1 values; the
        // matrix behind it (ycc_to_rgb, DoviColorMatrix) then carries no UNORM scale.  Curves are read from the LDS copy.
        const f2 sy2 = splat(P.sy), sc2 = splat(P.sc);
#pragma unroll
        for (int rr = 0; rr < 2; rr++) { Ycol[rr] = Ycol[rr] * sy2; Ucol[rr] = Ucol[rr] * sc2; Vcol[rr] = Vcol[rr] * sc2; }
        dovi_reshape_block(*DR, DL, Ycol, Ucol, Vcol);
    }
    f2 rgbc[2][3];
#pragma unroll
    for (int rr = 0; rr < 2; rr++)                // rr = luma column of the block
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            if (DV != DV_NONE)
                rgbc[rr][ch] = fma_k<false>(MM, 3 * ch, Ycol[rr], fma_k<false>(MM, 3 * ch + 1, Ucol[rr], fma_k<false>(MM, 3 * ch + 2, Vcol[rr], CC[ch])));
            else
                rgbc[rr][ch] = fma_k<true>(MM, 3 * ch, Ycol[rr], fma_k<false>(MM, 3 * ch + 1, Ucol[rr], fma_k<false>(MM, 3 * ch + 2, Vcol[rr], CC[ch])));
        }
    if (DV != DV_NONE) {
        // PQ EOTF -> LMS matrix -> PQ OETF (Shaders.cpp:844-859).  EOTF from the LDS table on [0, 1]; the reference clamps at 0
        // only, so a code above 1.0 (possible behind ycc_to_rgb) is decoded literally — a branch no wave takes on ordinary content.
        // DV_SDR (8-bit tone-mapped target) reads the EOTF from the LDS table; DV_GENERAL (HDR output: 10-bit PQ codes leave the
        // kernel, and the reference's own fp32 pow chain is what they are compared with code for code) evaluates it literally —
        // a smooth table cannot follow the rounding noise of exp2(y * log2 x) closer than 1 % of the 10-bit codes.
        f2 ent[2][3][2]; float frc[2][3][2];
        float over = 1.0f;
        if (DV == DV_SDR || DV == DV_SDR_L2) {
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        over = fmaxf(over, rgbc[rr][ch][e]);
                        // the table holds log2 of the EOTF, sampled at x = (i / N)^2: the EOTF itself is a ~x^3 power law at the
                        // dark end, where linear interpolation on a uniform grid is 1e-3 relative; its logarithm over sqrt(x)
                        // interpolates to 1.2e-6 with N = 8192 (adjacent entries: one ds_read2_b32, the slope is their difference)
                        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
                        const float t = __builtin_amdgcn_sqrtf(__builtin_amdgcn_fmed3f(rgbc[rr][ch][e], 0.0f, 1.0f)) * (float)EOTF_N;
                        frc[rr][ch][e] = __builtin_amdgcn_fractf(t);
                        const f2u pe = *(const f2u *)(TE + (int)t);            // (int)t <= N: the table has N + 1 values and a pad
                        ent[rr][ch][e] = f2{pe.x, pe.y - pe.x};
                    }
        }
        float L[9];
#pragma unroll
        for (int i = 0; i < 9; i++) L[i] = DL->lms[i];
        const float A_ = 0.15f, B_ = 0.50f, CB = 0.10f * 0.50f, DE = 0.20f * 0.02f, DF = 0.20f * 0.30f, EF = 0.02f / 0.30f;
        const float inv_div = 1.0f / (((4.8f * (A_ * 4.8f + CB) + DE) / (4.8f * (A_ * 4.8f + B_) + DF)) - EF);
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            f2 lin[3], lms[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
#ifndef MPCVR_DV_EXP
#define MPCVR_DV_EXP 0      // experiment builds only (tools/build_variant.sh): 1 = the PQ EOTF of the Dolby Vision block convert literally instead of from
#endif                      // its table, 2 = Hable's quotient by a true division instead of v_rcp_f32 (round 4: which of them moves the cancelling channels)
                if ((DV == DV_SDR || DV == DV_SDR_L2) && !(MPCVR_DV_EXP & 1))
                    lin[ch] = f2{__builtin_amdgcn_exp2f(__builtin_fmaf(ent[rr][ch][0].y, frc[rr][ch][0], ent[rr][ch][0].x)),
                                 __builtin_amdgcn_exp2f(__builtin_fmaf(ent[rr][ch][1].y, frc[rr][ch][1], ent[rr][ch][1].x))};
                else
                    lin[ch] = f2{st2084_to_linear(fmaxf(rgbc[rr][ch][0], 0.0f), 1.0f), st2084_to_linear(fmaxf(rgbc[rr][ch][1], 0.0f), 1.0f)};
            }
            if ((DV == DV_SDR || DV == DV_SDR_L2) && over > 1.0f) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
#pragma unroll
                    for (int e = 0; e < 2; e++)
                        if (rgbc[rr][ch][e] > 1.0f) lin[ch][e] = st2084_to_linear(rgbc[rr][ch][e], 1.0f);
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const f2 v = pk_fma(splat(L[3 * ch]), lin[0], pk_fma(splat(L[3 * ch + 1]), lin[1], splat(L[3 * ch + 2]) * lin[2]));
                lms[ch] = f2{fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f)};
            }
            if (DV == DV_SDR) {
                // LinearToST2084 -> saturate -> ST2084ToLinear(., scale) is x -> min(x, 1) * scale: elided (like the HLG tail's
                // round trip); Hable / hable(4.8), 2020 -> 709, saturate, pow 1/2.2 in ALU
                f2 tm[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const f2 x = f2{fminf(lms[ch].x, 1.0f), fminf(lms[ch].y, 1.0f)} * splat(P.lum_scale);
                    const f2 num = pk_fma(x, pk_fma(splat(A_), x, splat(CB)), splat(DE));
                    const f2 den = pk_fma(x, pk_fma(splat(A_), x, splat(B_)), splat(DF));
                    const f2 q = (MPCVR_DV_EXP & 2) ? f2{num.x / den.x, num.y / den.y} : f2{num.x * __builtin_amdgcn_rcpf(den.x), num.y * __builtin_amdgcn_rcpf(den.y)};
                    tm[ch] = (q - splat(EF)) * splat(inv_div);
                }
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const f2 g = fma_k<true>(GG, 3 * ch, tm[0], fma_k<false>(GG, 3 * ch + 1, tm[1], mul_k(GG, 3 * ch + 2, tm[2])));
                    out[rr][ch] = f2{hlsl_pow(g.x, 1.0f / 2.2f), hlsl_pow(g.y, 1.0f / 2.2f)};
                }
            } else if (DV == DV_SDR_L2) {
                // LinearToST2084, saturate, DolbyVisionTrims (Shaders.cpp:766-773,873-877) in ALU; then the HDR10 path's table for
                // saturate -> ST2084ToLinear * scale -> Hable / hable(4.8); 2020 -> 709, saturate, pow 1/2.2
                float k5[5];
#pragma unroll
                for (int i = 0; i < 5; i++) k5[i] = DL->l2k[i];
                f2 tm[3];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    // the divisions as multiplications by v_rcp_f32 (1 ulp): the result is an 8-bit code behind a tone map and a dither
                    float c3[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
#ifndef MPCVR_DV_PQENC_ALU      // (experiment builds: the literal chain — two pow() and a division — instead of the table)
                        // saturate(LinearToST2084(x, 1)) from the table over log2 x (vp_params.h kPqEncSize): x >= 1 reads the last entry's end (1.0),
                        // x = 0 (log2 = -inf) the first entry
                        const float u = __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_amdgcn_logf(lms[ch][e]), (float)kPqEncSize / (float)kPqEncLog2Range, (float)kPqEncSize),
                                                                0.0f, (float)kPqEncSize - 0.001f);
                        const f2 pe = (T + LUT_N)[(int)u];
                        const float pq = __builtin_fmaf(pe.y, __builtin_amdgcn_fractf(u), pe.x);
                        c3[ch] = hlsl_pow(pq * k5[2] + k5[3], k5[4]);
#else
                        const float z = hlsl_pow(lms[ch][e], MPCVR_ST2084_m1);
                        const float q = (MPCVR_ST2084_c1 + MPCVR_ST2084_c2 * z) * __builtin_amdgcn_rcpf(1.0f + MPCVR_ST2084_c3 * z);
                        c3[ch] = hlsl_pow(saturate(hlsl_pow(q, MPCVR_ST2084_m2)) * k5[2] + k5[3], k5[4]);
#endif
                    }
                    const float ky = (1.0f + k5[0]) * __builtin_amdgcn_rcpf(0.2627f * c3[0] + 0.6780f * c3[1] + 0.0593f * c3[2]);
                    const float v3[3] = {c3[0] * hlsl_pow(ky * c3[0], k5[1]), c3[1] * hlsl_pow(ky * c3[1], k5[1]), c3[2] * hlsl_pow(ky * c3[2], k5[1])};
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const float t = __builtin_amdgcn_fmed3f(v3[ch], 0.0f, 1.0f) * (float)(LUT_N - 1);
                        const f2 en = T[(int)t];
                        tm[ch][e] = __builtin_fmaf(en.y, __builtin_amdgcn_fractf(t), en.x);
                    }
                }
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const f2 g = fma_k<true>(GG, 3 * ch, tm[0], fma_k<false>(GG, 3 * ch + 1, tm[1], mul_k(GG, 3 * ch + 2, tm[2])));
                    out[rr][ch] = f2{hlsl_pow(g.x, 1.0f / 2.2f), hlsl_pow(g.y, 1.0f / 2.2f)};
                }
            } else {
                const float *l2k = DL->l2_enabled ? DL->l2k : nullptr;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    f3 c = {linear_to_st2084(lms[0][e], 1.0f), linear_to_st2084(lms[1][e], 1.0f), linear_to_st2084(lms[2][e], 1.0f)};
                    c = hdr_tail(c, P.tail, P.gamma, P.lum_scale, make_mat3(P.gamut), l2k, nullptr);
                    out[rr][0][e] = saturate(c.x); out[rr][1][e] = saturate(c.y); out[rr][2][e] = saturate(c.z);
                }
            }
        }
        return;
    }
    // PQ: all twelve table reads of the block are issued together (their addresses depend only on the matrix results),
    // so the wave pays one LDS round trip per iteration instead of six
    f2 linc[2][3];
    if (tail_has_table(TAIL)) {
        f2 ent[2][3][2]; float frc[2][3][2];
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float t = rgbc[rr][ch][e] * (float)(LUT_N - 1);
                    frc[rr][ch][e] = __builtin_amdgcn_fractf(t);
                    ent[rr][ch][e] = T[(int)t];
                }
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    float r;   // {value, slope} entry: plain v_fma_f32 — a packed pair would need three v_mov to line its operands
                               // up, and packing only pays when it is free (tools/ubench/op_rate.hip)
                    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(ent[rr][ch][e].y), "v"(frc[rr][ch][e]), "v"(ent[rr][ch][e].x));
                    linc[rr][ch][e] = r;
                }
    }
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const f2 *rgb = rgbc[rr];
        if (TAIL == TAILK_PQ_LUT) {
            // Shaders.cpp:870-923: per-channel saturate -> ST2084ToLinear*scale -> Hable/hable(4.8) from the LDS table,
            // then the 2020->709 matrix, saturate and pow 1/2.2 in ALU
            const f2 *lin = linc[rr];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const f2 g = fma_k<true>(GG, 3 * ch, lin[0], fma_k<false>(GG, 3 * ch + 1, lin[1], mul_k(GG, 3 * ch + 2, lin[2])));
                out[rr][ch] = f2{hlsl_pow(g.x, 1.0f / 2.2f), hlsl_pow(g.y, 1.0f / 2.2f)};
            }
        } else if (TAIL == TAILK_HLG) {
            // Shaders.cpp:862-923 for HLG: saturate -> HLGtoLinear (hlg.hlsl:1-20) -> LinearToST2084(., 1000) -> saturate ->
            // ST2084ToLinear(., scale) -> Hable -> 2020->709 -> saturate -> pow 1/2.2.  The PQ encode/decode round trip
            // (quirk Q7) is the identity x -> x*scale/1000 on the whole reachable range (x/1000 <= 0.09, never
            // saturated); it is elided here.  The literal chain — kept in the pass-per-kernel path and the oracle —
            // differs from the identity by ~1e-5 relative, the rounding noise of its own four pow() calls.
            // inverse_HLG per channel (x <= 0.5 ? 4x^2 : exp((x - c) / a) + b) from the LDS table, read with the block's other lookups
            const f2 *lin = linc[rr];
            const f2 ys = splat(2000.0f) * pk_fma(splat(0.2627f), lin[0], pk_fma(splat(0.6780f), lin[1], splat(0.0593f) * lin[2]));
            const float ks = P.lum_scale * (1.0f / 1000.0f);
            const f2 gain = f2{hlsl_pow(ys.x, 0.2f) * ks, hlsl_pow(ys.y, 0.2f) * ks};
            const float A_ = 0.15f, B_ = 0.50f, CB = 0.10f * 0.50f, DE = 0.20f * 0.02f, DF = 0.20f * 0.30f, EF = 0.02f / 0.30f;
            const float inv_div = 1.0f / (((4.8f * (A_ * 4.8f + CB) + DE) / (4.8f * (A_ * 4.8f + B_) + DF)) - EF);
            f2 tm[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const f2 x = lin[ch] * gain;
                const f2 num = pk_fma(x, pk_fma(splat(A_), x, splat(CB)), splat(DE));
                const f2 den = pk_fma(x, pk_fma(splat(A_), x, splat(B_)), splat(DF));
                const f2 q = f2{num.x * __builtin_amdgcn_rcpf(den.x), num.y * __builtin_amdgcn_rcpf(den.y)};
                tm[ch] = (q - splat(EF)) * splat(inv_div);
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const f2 g = fma_k<true>(GG, 3 * ch, tm[0], fma_k<false>(GG, 3 * ch + 1, tm[1], mul_k(GG, 3 * ch + 2, tm[2])));
                out[rr][ch] = f2{hlsl_pow(g.x, 1.0f / 2.2f), hlsl_pow(g.y, 1.0f / 2.2f)};
            }
        } else if (TAIL == TAILK_ALU) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                f3 v = {rgb[0][e], rgb[1][e], rgb[2][e]};
                v = hdr_tail(v, P.tail, P.gamma, P.lum_scale, make_mat3(P.gamut));
                out[rr][0][e] = saturate(v.x); out[rr][1][e] = saturate(v.y); out[rr][2][e] = saturate(v.z);
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) out[rr][ch] = rgb[ch];
        }
    }
}


// ---- CHROMA_CatmullRom for 4:2:0 (Shaders.cpp:66-72,242-251,288-299): the 2x2 block's four pixels weigh the same 4 chroma columns
// bx-1 .. bx+2 (two weight sets, by column parity) and — rows (odd, odd+1) straddling two chroma rows — 5 chroma rows base .. base+4
// (base = (sy0 >> 1) - 1; row r uses base + o_r .. + 3 with the weight set of its parity).  P.crx / P.cry hold catmull_weights(t) for
// t(parity) as the shader computes it for the stream's siting, built on the host with the shader's own expressions. ----
struct RawCR {
    uint32_t y[2];
    uint32_t c[5][4];        // [chroma row base + j][column bx - 1 + i]: U | V << 16 raw codes
};
struct RawAddrCR { uint32_t yoff; uint32_t coff[4]; };

template <int SRC>
__device__ __forceinline__ void make_raw_addr_cr(const FusedArgs &P, int Xg, RawAddrCR &ra)
{
    const int sx0 = P.rect_l + Xg, bx = sx0 >> 1;
    const int yb = src_wide<SRC>(P) ? 2 : 1;
    const int cb = src_biplanar<SRC>(P) ? 2 * yb : yb;
    ra.yoff = (uint32_t)(yb * sx0);
#pragma unroll
    for (int i = 0; i < 4; i++) ra.coff[i] = (uint32_t)(cb * clampi(bx - 1 + i, 0, P.cw - 1));
}
template <int SRC>
__device__ __forceinline__ void load_raw_cr(const FusedArgs &P, gcptr py, const RawAddrCR &ra, int y0, int y1, RawCR &r)
{
    const int sy0 = P.rect_t + y0, sy1 = P.rect_t + y1;
    const gcptr ry0 = py + (uint32_t)sy0 * (uint32_t)P.pitch_y, ry1 = py + (uint32_t)sy1 * (uint32_t)P.pitch_y;
    r.y[0] = src_wide<SRC>(P) ? ld_u32(ry0 + opaque(ra.yoff)) : ld_u16(ry0 + opaque(ra.yoff));
    r.y[1] = src_wide<SRC>(P) ? ld_u32(ry1 + opaque(ra.yoff)) : ld_u16(ry1 + opaque(ra.yoff));
    const int base = (sy0 >> 1) - 1;
    const gcptr pu = py + P.off_u, pv = src_biplanar<SRC>(P) ? pu : py + P.off_v;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint32_t o = (uint32_t)clampi(base + j, 0, P.ch - 1) * (uint32_t)P.pitch_c;
#pragma unroll
        for (int i = 0; i < 4; i++) r.c[j][i] = ld_uv<SRC>(P, pu + o, pv + o, opaque(ra.coff[i]));
    }
}
// exact form (FusedArgs::exact_cv, see convert_block_exact): code_Bicubic_UV (Shaders.cpp:74-79) on 0..1 texels, the four products of a row
// summed left to right, then the four rows the same way
#pragma clang fp contract(off)
template <int SRC, int OUTK>
__device__ __forceinline__ void convert_block_cr_exact(const FusedArgs &P, const RawCR &r, int sy0, int sy1, f2 out[2][3])
{
    f2 Q[5][2];                                   // [chroma row base + j][column parity] = (U, V)
#pragma unroll
    for (int j = 0; j < 5; j++) {
        f2 t[4];
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = xnorm2_src<SRC>(f2{(float)(r.c[j][i] & 0xffffu), (float)(r.c[j][i] >> 16)}, P.xdc, P.xrc);
#pragma unroll
        for (int par = 0; par < 2; par++) {
            const float *w = P.crx[par];
            f2 q = t[0] * splat(w[0]) + t[1] * splat(w[1]);
            q = q + t[2] * splat(w[2]);
            Q[j][par] = q + t[3] * splat(w[3]);
        }
    }
    const int base = (sy0 >> 1) - 1;
    f2 uv[2][2];                                  // [column][row] = (U, V)
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int sy = rr ? sy1 : sy0;
        const int o = (sy >> 1) - 1 - base;       // 0 or 1, wave-uniform
        const float *w = P.cry[sy & 1];
#pragma unroll
        for (int col = 0; col < 2; col++) {
            const f2 q0 = o ? Q[1][col] : Q[0][col], q1 = o ? Q[2][col] : Q[1][col], q2 = o ? Q[3][col] : Q[2][col], q3 = o ? Q[4][col] : Q[3][col];
            f2 v = q0 * splat(w[0]) + q1 * splat(w[1]);
            v = v + q2 * splat(w[2]);
            uv[col][rr] = v + q3 * splat(w[3]);
        }
    }
    f2 Y[2], U[2], V[2];
    if (src_wide<SRC>(P)) {
        Y[0] = xnorm2_src<SRC>(f2{(float)(r.y[0] & 0xffffu), (float)(r.y[1] & 0xffffu)}, P.xdy, P.xry);
        Y[1] = xnorm2_src<SRC>(f2{(float)(r.y[0] >> 16), (float)(r.y[1] >> 16)}, P.xdy, P.xry);
    } else {
        Y[0] = xnorm2_src<SRC>(f2{(float)(r.y[0] & 0xffu), (float)(r.y[1] & 0xffu)}, P.xdy, P.xry);
        Y[1] = xnorm2_src<SRC>(f2{(float)((r.y[0] >> 8) & 0xffu), (float)((r.y[1] >> 8) & 0xffu)}, P.xdy, P.xry);
    }
#pragma unroll
    for (int col = 0; col < 2; col++) { U[col] = f2{uv[col][0].x, uv[col][1].x}; V[col] = f2{uv[col][0].y, uv[col][1].y}; }
    exact_matrix_store<OUTK>(P, Y, U, V, out);
}
#pragma clang fp contract(fast)
template <int TAIL, int SRC, int DV = DV_NONE, int XC = XC_NEVER, int OUTK = OUT_NORM>
__device__ __forceinline__ void convert_block_cr(const FusedArgs &P, const f2 (&MM)[5], const f2 (&GG)[5], const f2 (&CC)[3], const RawCR &r, int sy0, int sy1, const f2 *T, f2 out[2][3],
                                                 const DoviParams *DL = nullptr, const float *TE = nullptr, const DoviRegs *DR = nullptr)
{
    if constexpr (XC == XC_ALWAYS) { convert_block_cr_exact<SRC, OUTK>(P, r, sy0, sy1, out); return; }
    // horizontal pass: Q[row j][column parity] = sum_i wx[parity][i] * texel[j][i], as (U, V) pairs
    f2 Q[5][2];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        f2 t[4];
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = f2{(float)(r.c[j][i] & 0xffffu), (float)(r.c[j][i] >> 16)};
#pragma unroll
        for (int par = 0; par < 2; par++) {
            const float *w = P.crx[par];
            Q[j][par] = pk_fma(splat(w[3]), t[3], pk_fma(splat(w[2]), t[2], pk_fma(splat(w[1]), t[1], splat(w[0]) * t[0])));
        }
    }
    // vertical pass per luma row: its 4 chroma rows start at o_r = (sy_r >> 1) - 1 - base, its weights follow its parity
    const int base = (sy0 >> 1) - 1;
    f2 uv[2][2];                                  // [column][row] = (U, V)
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int sy = rr ? sy1 : sy0;
        const int o = (sy >> 1) - 1 - base;       // 0 or 1, wave-uniform
        const float *w = P.cry[sy & 1];
#pragma unroll
        for (int col = 0; col < 2; col++) {
            const f2 q0 = o ? Q[1][col] : Q[0][col], q1 = o ? Q[2][col] : Q[1][col], q2 = o ? Q[3][col] : Q[2][col], q3 = o ? Q[4][col] : Q[3][col];
            uv[col][rr] = pk_fma(splat(w[3]), q3, pk_fma(splat(w[2]), q2, pk_fma(splat(w[1]), q1, splat(w[0]) * q0)));
        }
    }
    f2 Ycol[2], Ucol[2], Vcol[2];
    if (src_wide<SRC>(P)) {
        Ycol[0] = f2{(float)(r.y[0] & 0xffffu), (float)(r.y[1] & 0xffffu)};
        Ycol[1] = f2{(float)(r.y[0] >> 16), (float)(r.y[1] >> 16)};
    } else {
        Ycol[0] = f2{(float)(r.y[0] & 0xffu), (float)(r.y[1] & 0xffu)};
        Ycol[1] = f2{(float)((r.y[0] >> 8) & 0xffu), (float)((r.y[1] >> 8) & 0xffu)};
    }
#pragma unroll
    for (int col = 0; col < 2; col++) { Ucol[col] = f2{uv[col][0].x, uv[col][1].x}; Vcol[col] = f2{uv[col][0].y, uv[col][1].y}; }
    convert_block_yuv<TAIL, SRC, DV>(P, MM, GG, CC, Ycol, Ucol, Vcol, T, out, DL, TE, DR);
}


inline int EnvInt(const char *name, int def)
{
    const char *v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : def;
}

}  // namespace

// host side, vp_fused.hip
// resize_follows: 1 / 0 = a resize reads the convert output / nothing does; -1 = what P.exact_convert says (FusedArgs::exact_cv)
void FillFusedArgs(const FusedParams &P, FusedArgs &a, int resize_follows = -1);
void ChromaCatmullWeights(int chroma_loc, float wx[2][4], float wy[2][4]);
int FusedTailKind(const FusedParams &P);
int FusedSourceKind(const FusedParams &P);
int FusedDoviKind(const FusedParams &P);          // DV_* for the launch
// vp_fused_mx.hip: the same launch as LaunchFusedUp2x with the resize taps on the matrix cores; hipErrorNotSupported when the
// variant does not cover the configuration (the caller then launches the packed-fp32 kernel)
// dynamic LDS above the default limit needs the function attribute once per kernel (and device); remembered per (kernel, device) (vp_fused_strip.hip)
hipError_t AllowLargeLds(const void *kern, size_t lds);
int DeviceCuCount();        // vp_fused.hip: compute units of the current device (256)
// vp_fused_jinc.hip: the fused 2x launch with the 2-D Jinc2m filter in the place of the two separable draws (`a` complete but for seg_rows)
hipError_t LaunchFusedJinc2x(const FusedParams &P, const FusedArgs &a, const float *jtab_dev, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s);
// LDS a workgroup of the current device may claim with that attribute (gfx950: 160 KiB): queried once per device, so a part or
// partition with less makes the planners fall back at plan time instead of failing at launch; 160 KiB where no device answers
size_t DeviceLdsLimit();
// vp_fused_period.hip: the periodic-phase kernel for this strip launch, or hipErrorNotSupported (LaunchFusedStrip then runs k_fused_strip)
hipError_t LaunchFusedPeriod(const FusedStripParams &S, const FusedArgs &a, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s);
// vp_fused_up2x.h, instantiated by vp_fused_up2x_nt{4,5,6}.hip: the packed-fp32 kernel for one tap count
template <int NT>
hipError_t LaunchFusedUp2xNT(const FusedParams &P, const FusedArgs &a, int strips, int seg, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s);

}  // namespace mpcvr
