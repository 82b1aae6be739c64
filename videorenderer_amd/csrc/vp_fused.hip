// vp_fused.hip — the fused exact-2x path: convert -> X pass -> Y pass -> final pass in ONE kernel.
//
// What the reference does in four draws with three HBM-resident intermediates
// (ConvertColorPass -> m_TexConvertOutput, TextureResizeShader X -> fp16 m_TexResize,
//  TextureResizeShader Y -> m_TexsPostScale, FinalPass -> back buffer; DX11VideoProcessor.cpp:3285-3424)
// happens here without any intermediate in HBM: the source sample is read once (+ halo) and the
// BGRA8/RGB10A2 output is written once.  Every intermediate rounding of the reference is kept:
//   convert output -> UNORM8/10 (m_InternalTexFmt), X pass -> fp16 RNE (:3155), Y pass -> UNORM8/10,
//   final pass floor(p*Q + dither) (ps_final_pass.hlsl:29).
//
// Design: WAVE-AUTONOMOUS STRIPS, REGISTER-RESIDENT VERTICAL WINDOW, PACKED FP32 MATH.
//   Exact 2x => two fixed phases per axis (t = 0.75 for even outputs, base = k-1; t = 0.25 for odd, base = k).
//   One wavefront owns a strip of S = 120 source columns (240 output columns = 60 lanes x 4 px = one
//   16-byte store per lane and output row) and marches down a segment of source rows, two rows per
//   iteration, with no workgroup barrier inside the loop:
//     stage C  lane j converts the 2x2 block {cols 2j,2j+1} x {rows a,a+1} of the 128-column window
//              (4-px halo each side) from raw codes prefetched one iteration ahead and writes it, rounded to
//              the internal UNORM format, to this wave's LDS slice A as (row a, row a+1) pairs
//     stage X  lane l reads columns 2l..2l+9 (5 x ds_read_b128 per channel) and produces the 4 output
//              columns it owns for BOTH rows at once (v_pk_fma_f32 on the row pairs); the fp16-rounded
//              results (m_TexResize) enter an 8-row register window — no LDS, no HBM
//     stage Y  from the window: 4 output rows x 4 px per lane with v_pk_fma_f32 on pixel pairs, UNORM
//              rounding (m_TexsPostScale), dither, one 16-byte store per row
//   On gfx950 a wave64 VALU instruction costs ~4 cycles of its SIMD (plain VOP2 fp32 with VGPR operands ~3; measured,
//   tools/ubench/op_rate.hip), so v_pk_{fma,mul}_f32 with an SGPR weight is the cheapest FMA here: two for the price of one.
//   The four waves of a workgroup share only the read-only tables (dither, PQ->SDR LUT).
//   Recomputed: the horizontal halo (8 of 128 columns) and 6 rows per segment.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "vp_device.h"
#include "vp_launch.h"
#include "vp_plan.h"

namespace mpcvr {

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
// source specialisation: GENERIC reads planes / bytes / siting at run time; P01X = bi-planar 16-bit (P010/P016), NV12 =
// bi-planar 8-bit, both with MPEG-2 or co-sited chroma (not horizontally centred)
enum { SRC_GENERIC = 0, SRC_P01X = 1, SRC_NV12 = 2 };
// epilogue specialisation: DITHER8 = B8G8R8A8 target behind a final pass (integer form); DIRECT8 = B8G8R8A8 target written
// straight from the Y pass (8-bit sources: no post-scale step); both require 16-byte aligned rows and off_x % 4 == 0
enum { EPI_GENERIC = 0, EPI_DITHER8 = 1, EPI_DIRECT8 = 2 };


// everything the kernel needs, flattened (kernel argument => SGPRs)
struct FusedArgs {
    uint32_t off_u, off_v;         // byte offsets of the chroma plane(s) inside a sample (u = interleaved UV when biplanar)
    int pitch_y, pitch_c;
    int tex_w, cw, ch;             // luma width, chroma size
    int rect_l, rect_t, W, H;      // source rect origin and size (== convert-output size)
    int bytes, planes;
    int center_h;                  // MPEG-1 siting: chroma sample centred between luma columns
    int v_off4;                    // vertical chroma offset in quarter chroma rows: 1 for co-sited (+0.25), else 0
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
};

template <int SRC> __device__ __forceinline__ bool src_wide(const FusedArgs &P) { return SRC == SRC_P01X ? true : SRC == SRC_NV12 ? false : P.bytes == 2; }
template <int SRC> __device__ __forceinline__ bool src_biplanar(const FusedArgs &P) { return SRC != SRC_GENERIC ? true : P.planes == 2; }
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

// coefficient i of a table packed two to an SGPR pair (i is a constant after unrolling)
template <bool CLAMP>
__device__ __forceinline__ f2 fma_k(const f2 *K, int i, f2 b, f2 c)
{
    return (i & 1) ? pk_fma_w<1, CLAMP>(K[i >> 1], b, c) : pk_fma_w<0, CLAMP>(K[i >> 1], b, c);
}
__device__ __forceinline__ f2 mul_k(const f2 *K, int i, f2 b) { return (i & 1) ? pk_mul_w<1>(K[i >> 1], b) : pk_mul_w<0>(K[i >> 1], b); }

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
    const int sx0 = P.rect_l + Xg, c0 = sx0 >> 1;
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
    if (P.bytes == 2) return ld_u16(pu + off) | (ld_u16(pv + off) << 16);
    return ld_u8(pu + off) | (ld_u8(pv + off) << 16);
}

// vertical chroma position of source row sy (Shaders.cpp:118-138): v' = (sy+0.5)/2 [+0.25 co-sited] - 0.5, kept in
// QUARTER chroma rows as an integer (4v' = 2sy - 1 [+1]) so that the whole siting computation stays on the scalar unit
__device__ __forceinline__ int chroma_v4(const FusedArgs &P, int sy) { return 2 * sy - 1 + P.v_off4; }
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
    r.y[0] = src_wide<SRC>(P) ? ld_u32(ry0 + opaque(ra.yoff)) : ld_u16(ry0 + opaque(ra.yoff));
    r.y[1] = src_wide<SRC>(P) ? ld_u32(ry1 + opaque(ra.yoff)) : ld_u16(ry1 + opaque(ra.yoff));
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

// The 2x2 block: 4:2:0 bilinear chroma + matrix (+ tail) for (even, odd column) x (row 0, row 1).
// ShaderGetPixels' CHROMA_Bilinear branch (Shaders.cpp:265-270,319-325): same sample positions and weights,
// evaluated in code units (vertical lerp first), UNORM scale folded into the matrix.  out[column][ch] = the channel as
// a (row 0, row 1) pair — the layout LDS slice A wants — saturated (every continuation, tail or UNORM store,
// saturates first).
template <int TAIL, int SRC>
__device__ __forceinline__ void convert_block(const FusedArgs &P, const f2 (&MM)[5], const f2 (&GG)[5], const f2 (&CC)[3], const Raw &r, int sy0, int sy1, const f2 *T, f2 out[2][3])
{
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
    } else {                                      // u' = sx/2
        Ucol[0] = Uc[1]; Vcol[0] = Vc[1];
        Ucol[1] = pk_fma(Uc[2], splat(0.5f), Uc[1] * splat(0.5f)); Vcol[1] = pk_fma(Vc[2], splat(0.5f), Vc[1] * splat(0.5f));
    }
    f2 rgbc[2][3];
#pragma unroll
    for (int rr = 0; rr < 2; rr++)                // rr = luma column of the block
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            rgbc[rr][ch] = fma_k<true>(MM, 3 * ch, Ycol[rr], fma_k<false>(MM, 3 * ch + 1, Ucol[rr], fma_k<false>(MM, 3 * ch + 2, Vcol[rr], CC[ch])));
    // PQ: all twelve table reads of the block are issued together (their addresses depend only on the matrix results),
    // so the wave pays one LDS round trip per iteration instead of six
    f2 linc[2][3];
    if (TAIL == TAILK_PQ_LUT) {
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
            f2 lin[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const f2 v = rgb[ch];
                lin[ch] = f2{v.x <= 0.5f ? v.x * v.x * 4.0f : __expf((v.x - 0.55991073f) * (1.0f / 0.17883277f)) + 0.28466892f,
                             v.y <= 0.5f ? v.y * v.y * 4.0f : __expf((v.y - 0.55991073f) * (1.0f / 0.17883277f)) + 0.28466892f};
            }
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

template <int NT, int TAIL, int SRC, int EPI>
__global__ __launch_bounds__(256, 3) void k_fused_up2x(FusedArgs P, const FusedFrame *__restrict__ frames, FusedFrame single)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Aall = (float *)smem;
    unsigned short *D = (unsigned short *)(smem + LDS_A);
    uint32_t *Di = (uint32_t *)(smem + LDS_A + LDS_D);
    f2 *T = (f2 *)(smem + LDS_A + LDS_D + LDS_DB);

    for (int i = threadIdx.x; i < 1024; i += 256) {
        const unsigned short d = P.dither[i];
        D[i] = d;
        Di[i] = (uint32_t)(__half2float(__ushort_as_half(d)) * 1024.0f + 0.5f) << 14;     // d = j/1024 exactly (dither32x32float16.bin)
    }
    if (TAIL == TAILK_PQ_LUT)
        for (int i = threadIdx.x; i < LUT_N; i += 256) {
            const float v = P.lut[i], n = P.lut[min(i + 1, LUT_N - 1)];
            T[i] = f2{v, n - v};
        }
    __syncthreads();                                   // the only workgroup barrier: tables visible

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int W = P.W, H = P.H;
    const int x0 = (blockIdx.x * WAVES + wave) * S;
    const int s0 = blockIdx.y * P.seg_rows;
    if (x0 >= W || s0 >= H) return;
    const int s1 = min(s0 + P.seg_rows, H);
    float *A = Aall + wave * A_FLOATS;

    // the frame table entry is wave-uniform; readfirstlane tells the compiler so (SGPR bases => saddr loads/stores)
    const FusedFrame frame = frames ? frames[blockIdx.z] : single;
    auto uniform_ptr = [](const void *q) {
        const uint64_t v = (uint64_t)q;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const uint64_t src_u = uniform_ptr(frame.src), dst_u = uniform_ptr(frame.dst);
    const gcptr py = (gcptr)src_u;
    const gptr pdst = (gptr)dst_u;

    // stage C role: A columns 2*lane, 2*lane+1 = rect columns X, X+1; the block is fetched at Xg (inside the rect)
    const int X = x0 - 4 + 2 * lane;
    const int Xg = clampi(X, 0, W - 2);
    const bool edge_wave = x0 == 0 || x0 + 2 * 63 - 4 > W - 2;      // wave-uniform: some lane's block hangs over the rect
    // stage X / Y role: output columns ox .. ox+3 (rect-relative); lanes 60..63 idle there
    const bool xy_active = lane < 60;
    const int ox = 2 * x0 + 4 * lane;
    const bool store_ok = xy_active && ox < 2 * W;
    const int wx0 = P.off_x + ox;
    const bool d_aligned = (wx0 & 3) == 0;          // wave-uniform: ox is a multiple of 4
    const bool st_aligned = d_aligned && (((uintptr_t)dst_u | (uintptr_t)P.dst_pitch) & 15) == 0;
    const uint32_t lane_off = (uint32_t)wx0 * 4u;

    // phase weights, two per SGPR pair: WT[parity][pair]
    const f2 WT[2][3] = {{f2{P.we[0], P.we[1]}, f2{P.we[2], P.we[3]}, f2{P.we[4], P.we[5]}},
                        {f2{P.wo[0], P.wo[1]}, f2{P.wo[2], P.wo[3]}, f2{P.wo[4], P.wo[5]}}};
    // colour matrix and gamut matrix, two coefficients per SGPR pair
    const f2 MM[5] = {f2{P.m[0], P.m[1]}, f2{P.m[2], P.m[3]}, f2{P.m[4], P.m[5]}, f2{P.m[6], P.m[7]}, f2{P.m[8], 0.0f}};
    const f2 GG[5] = {f2{P.gamut[0], P.gamut[1]}, f2{P.gamut[2], P.gamut[3]}, f2{P.gamut[4], P.gamut[5]}, f2{P.gamut[6], P.gamut[7]}, f2{P.gamut[8], 0.0f}};
    constexpr bool FASTEPI = EPI == EPI_DITHER8;     // integer final pass; EPI_DIRECT8 shares its alignment preconditions
    const f2 maxv2 = splat((FASTEPI || P.final_pass) ? P.maxv : P.quant);
    const f2 cmax2 = splat(P.maxv), cinv2 = splat(P.inv_maxv);
    f2 big2 = splat(8388608.0f);                     // 2^23, pinned in VGPRs (see unorm_round2)
    asm volatile("" : "+v"(big2));
    f2 CC[3] = {splat(P.c[0]), splat(P.c[1]), splat(P.c[2])};     // matrix offsets: FMA addends must be VGPRs anyway
    asm volatile("" : "+v"(CC[0]), "+v"(CC[1]), "+v"(CC[2]));

    // 8-row window of X-pass results, already rounded through fp16: [row slot][channel][pixel pair]
    f2 win[8][3][2];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int c = 0; c < 3; c++) win[i][c][0] = win[i][c][1] = splat(0.0f);

    // iteration t adds virtual rows a, a+1 with a = s0 - 3 + 2t; from t = 3 on it emits output rows of k = a-3, a-2
    const int n_iter = (s1 - s0 + 1) / 2 + 3;
    Raw raw;
    RawAddr ra;
    make_raw_addr<SRC>(P, Xg, ra);
    load_raw<SRC>(P, py, ra, clampi(s0 - 3, 0, H - 1), clampi(s0 - 2, 0, H - 1), raw);

    // stage C for virtual rows ar, ar+1 (whose raw codes were prefetched): convert, write A, prefetch the next pair
    auto stage_c = [&](int ar) {
        f2 rc[2][3];
        convert_block<TAIL, SRC>(P, MM, GG, CC, raw, P.rect_t + clampi(ar, 0, H - 1), P.rect_t + clampi(ar + 1, 0, H - 1), T, rc);
        load_raw<SRC>(P, py, ra, clampi(ar + 2, 0, H - 1), clampi(ar + 3, 0, H - 1), raw);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // store to m_TexConvertOutput (UNORM: floor(sat(x)*maxv + 0.5)) and read back (q/maxv to 1 ulp)
            f2 qe = unorm_round2(rc[0][c], cmax2, big2) * cinv2;             // even column, rows (a, a+1)
            f2 qo = unorm_round2(rc[1][c], cmax2, big2) * cinv2;             // odd column
            // A[ch][col][row]: columns 2l, 2l+1 as (row a, row a+1) pairs = one 16-byte store
            *(f4 *)(A + (c * AW + 2 * lane) * 2) = f4{qe.x, qe.y, qo.x, qo.y};
            if (edge_wave) {       // clamp-to-edge of the convert texture: patch the column that hangs over (rare wave;
                                   // an LDS store so that the compiler keeps it a branch instead of 12 selects per iteration)
                if (X < 0) *(f2 *)(A + (c * AW + 2 * lane + 1) * 2) = qe;
                else if (X > W - 2) *(f2 *)(A + (c * AW + 2 * lane) * 2) = qo;
            }
        }
    };
    // Software pipeline: iteration t runs  X(t) -> C(t+1) -> Y(t), so the LDS write->read round trip of A (and the
    // global prefetch behind it) is covered by the Y stage instead of stalling the wave.  A is exchanged between
    // lanes of this wave only: LDS operations of one wave execute in order; the fences keep the compiler from
    // reordering the A reads and writes (which look unrelated thread by thread).
    stage_c(s0 - 3);

    for (int tb = 0; tb < n_iter; tb += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = tb + u;
            if (t >= n_iter) break;
            const int a = s0 - 3 + 2 * t;

            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---------------- stage X ----------------
            // lane owns output columns 4l..4l+3 = sources k = 2l (e = 0,1), 2l+1 (e = 2,3); A column of source k is k+4.
            // av[i] = (row a, row a+1) of A column 2l+i, i = 0..9  =>  source 2l + i - 4.
            if (xy_active) {
                f4 abuf[2][5];                                // LDS reads of channel c+1 are in flight while channel c is filtered
#pragma unroll
                for (int i = 0; i < 5; i++) abuf[0][i] = ((const f4 *)(A + (0 * AW + 2 * lane) * 2))[i];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (c < 2) {
#pragma unroll
                        for (int i = 0; i < 5; i++) abuf[(c + 1) & 1][i] = ((const f4 *)(A + ((c + 1) * AW + 2 * lane) * 2))[i];
                    }
                    f2 av[10];
#pragma unroll
                    for (int i = 0; i < 5; i++) { const f4 p4 = abuf[c & 1][i]; av[2 * i] = f2{p4.x, p4.y}; av[2 * i + 1] = f2{p4.z, p4.w}; }
                    f2 o[4];                                  // 4 output columns x (row a, row a+1)
#pragma unroll
                    for (int odd = 0; odd < 2; odd++)         // even output 2k: base = k-1; odd output 2k+1: base = k
                        taps2<NT, false>(WT[odd],             // source k = 2l + kk (kk = 0, 1) -> av index of k is kk + 4
                                         [&](int tt) { return av[4 + (odd ? 0 : -1) + tap_off<NT>(tt)]; },
                                         [&](int tt) { return av[5 + (odd ? 0 : -1) + tap_off<NT>(tt)]; }, o[odd], o[2 + odd]);
                    // m_TexResize is R16G16B16A16_FLOAT (:3155): round to fp16 (RNE), keep the rounded value as fp32
                    const int sa = (2 * u) & 7, sb = (2 * u + 1) & 7;
                    const f2 h0 = half_round2(o[0]), h1 = half_round2(o[1]), h2 = half_round2(o[2]), h3 = half_round2(o[3]);
                    win[sa][c][0] = f2{h0.x, h1.x};
                    win[sa][c][1] = f2{h2.x, h3.x};
                    win[sb][c][0] = f2{h0.y, h1.y};
                    win[sb][c][1] = f2{h2.y, h3.y};
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---------------- stage C of the NEXT iteration ----------------
            if (t + 1 < n_iter) stage_c(a + 2);

            // ---------------- stage Y + final pass ----------------
            // window slot of virtual row r is (r - (s0-3)) & 7; rows a-6 .. a+1 are live: slot(a-6+i) = (2u+2+i) & 7
            if (t >= 3 && store_ok) {
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const int k = a - 3 + kk;                     // source row -> output rows 2k (even), 2k+1 (odd); k < s1
                                                                  // because segments hold an even number of rows (host-checked)
#pragma unroll
                    for (int par = 0; par < 2; par++) {
                        // even: base = k-1 -> row k-1+off = a-6 + (kk+2+off); odd: base = k -> a-6 + (kk+3+off)
                        const int wy = P.off_y + 2 * k + par;
                        uint32_t dj[4] = {0, 0, 0, 0};
                        if (FASTEPI) {  // dither texels of this row first: the LDS round trip hides behind the tap filters.
                            // sampler WRAP+POINT: texel (wx mod 32, wy mod 32); FASTEPI implies off_x % 4 == 0 (launcher),
                            // so the four texels are one aligned 16-byte read
                            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                            const u32x4 dd = *(const u32x4 *)(Di + (wy & 31) * 32 + (wx0 & 31));
                            dj[0] = dd.x; dj[1] = dd.y; dj[2] = dd.z; dj[3] = dd.w;
                        }
                        f2 res[3][2];                             // [channel][pixel pair], saturated
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            taps2<NT, true>(WT[par],
                                            [&](int tt) { return win[(2 * u + 2 + kk + 2 + par + tap_off<NT>(tt)) & 7][c][0]; },
                                            [&](int tt) { return win[(2 * u + 2 + kk + 2 + par + tap_off<NT>(tt)) & 7][c][1]; }, res[c][0], res[c][1]);
                        }
                        uint32_t pk[4];
                        if (FASTEPI) {
                            // m_TexsPostScale store/load: k = floor(x*maxv + 0.5), p = k/maxv; ps_final_pass.hlsl:29:
                            // floor(p*255 + d), d = j/1024.  In integers: (k*M + (j << 14)) >> 24 with M = ceil(255*2^24/maxv)
                            // equals floor(k*255/maxv + j/1024) for every (k, j) (exhaustively checked, tests/test_host_logic.py)
                            // and differs from the fp32 shader arithmetic in 4 of the 2^20 (k, j) pairs, where fp32 rounds
                            // the sum up onto an integer.  x*maxv + 2^23 leaves k in the low mantissa bits, which is all
                            // v_mad_u32_u24 reads; the result byte is the top byte, gathered by two v_perm_b32 per pixel.
                            f2 uq[3][2];
#pragma unroll
                            for (int c = 0; c < 3; c++)
#pragma unroll
                                for (int pp = 0; pp < 2; pp++) uq[c][pp] = pk_fma(res[c][pp], maxv2, big2);
#pragma unroll
                            for (int px = 0; px < 4; px++) {
                                const uint32_t ib = __umul24(__float_as_uint(uq[2][px >> 1][px & 1]), P.epi_mul) + dj[px];
                                const uint32_t ig = __umul24(__float_as_uint(uq[1][px >> 1][px & 1]), P.epi_mul) + dj[px];
                                const uint32_t ir = __umul24(__float_as_uint(uq[0][px >> 1][px & 1]), P.epi_mul) + dj[px];
                                const uint32_t bg = __builtin_amdgcn_perm(ig, ib, 0x0c0c0703u);    // [B, G, 0, 0]
                                pk[px] = __builtin_amdgcn_perm(ir, bg, 0x0d070100u);               // [B, G, R, 0xff]
                            }
                        } else if (EPI == EPI_DIRECT8) {
                            // no post-scale step (8-bit internal format): the Y pass result is stored straight into the
                            // B8G8R8A8 target, floor(x*255 + 0.5).  x*255 + 2^23 leaves the code in the low mantissa byte;
                            // two v_perm_b32 gather B,G,R and the opaque alpha.
                            f2 uq[3][2];
#pragma unroll
                            for (int c = 0; c < 3; c++)
#pragma unroll
                                for (int pp = 0; pp < 2; pp++) uq[c][pp] = pk_fma(res[c][pp], maxv2, big2);
#pragma unroll
                            for (int px = 0; px < 4; px++) {
                                const uint32_t bg = __builtin_amdgcn_perm(__float_as_uint(uq[1][px >> 1][px & 1]), __float_as_uint(uq[2][px >> 1][px & 1]), 0x0c0c0400u);
                                pk[px] = __builtin_amdgcn_perm(__float_as_uint(uq[0][px >> 1][px & 1]), bg, 0x0d040100u);
                            }
                        } else {
                            // generic epilogue: no final pass (straight UNORM store into the RT) and/or R10G10B10A2 target
#pragma unroll
                            for (int px = 0; px < 4; px++) {
                                float c3[3];
#pragma unroll
                                for (int c = 0; c < 3; c++) {
                                    const float q = floorf(fmaf(res[c][px >> 1][px & 1], P.final_pass ? P.maxv : P.quant, 0.5f));
                                    float v = q;
                                    if (P.final_pass) {
                                        const float d = __half2float(__ushort_as_half(D[(wy & 31) * 32 + ((wx0 + px) & 31)]));
                                        v = fminf(fmaxf(floorf(fmaf(q, P.q_over_maxv, d)), 0.0f), P.quant);
                                    }
                                    c3[c] = v;
                                }
                                pk[px] = P.out10 ? pack_rgb10a2(c3[0], c3[1], c3[2]) : pack_bgra8(c3[0], c3[1], c3[2]);
                            }
                        }
                        const gptr rowp = pdst + (uint32_t)wy * (uint32_t)P.dst_pitch;    // wave-uniform row base + per-lane 32-bit offset
                        if (EPI != EPI_GENERIC || st_aligned) {      // specialised epilogues: 16-byte alignment of every row is a launch precondition
                            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                            u32x4 v4 = {pk[0], pk[1], pk[2], pk[3]};
                            *(__attribute__((address_space(1))) u32x4 *)(rowp + opaque(lane_off)) = v4;
                        } else {
                            __attribute__((address_space(1))) uint32_t *dst = (__attribute__((address_space(1))) uint32_t *)(rowp + lane_off);
                            dst[0] = pk[0]; dst[1] = pk[1]; dst[2] = pk[2]; dst[3] = pk[3];
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The convert stage on its own (pass-per-kernel path and same-size frames): the 2x2-block convert of the fused kernel —
// shared chroma fetch, scalar siting, packed matrix, table tone map — without the resize behind it.  A wave owns a strip of
// 128 rect columns and walks `pairs` row pairs (a, a+1), a odd (so both rows take their chroma from the same two chroma
// rows); the first and the last pair of a frame clamp to one useful row.
// FINAL = false: the block is stored as texels of the internal UNORM format (m_TexConvertOutput, or the render target when
// nothing follows); FINAL = true: 10-bit internal -> ps_final_pass in integers -> B8G8R8A8 (see the fused epilogue).
// ------------------------------------------------------------------------------------------------
template <int TAIL, int SRC, bool FINAL>
__global__ __launch_bounds__(256) void k_convert_blocks(FusedArgs P, const FusedFrame *__restrict__ frames, FusedFrame single, int pairs,
                                                       uint8_t *batch_dst, size_t batch_stride, FrameTable32 tab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *Di = (uint32_t *)smem;                                   // dither as j << 14 (FINAL)
    f2 *T = (f2 *)(smem + (FINAL ? 4096 : 0));
    if (FINAL)
        for (int i = threadIdx.x; i < 1024; i += 256)
            Di[i] = (uint32_t)(__half2float(__ushort_as_half(P.dither[i])) * 1024.0f + 0.5f) << 14;
    if (TAIL == TAILK_PQ_LUT)
        for (int i = threadIdx.x; i < LUT_N; i += 256) {
            const float v = P.lut[i], n = P.lut[min(i + 1, LUT_N - 1)];
            T[i] = f2{v, n - v};
        }
    if (FINAL || TAIL == TAILK_PQ_LUT) __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int W = P.W, H = P.H;
    const int X = blockIdx.x * 128 + 2 * lane;                         // rect columns X, X+1
    const int pair0 = (blockIdx.y * 4 + wave) * pairs;                 // pair p covers rect rows 2p-1, 2p
    if (X >= W || 2 * pair0 - 1 >= H) return;
    const FusedFrame frame = tab.n ? tab.f[blockIdx.z] : frames ? frames[blockIdx.z] : single;     // tab: the table in the kernel arguments
    auto uniform_ptr = [](const void *q) {
        const uint64_t v = (uint64_t)q;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const gcptr py = (gcptr)uniform_ptr(frame.src);
    // batch_dst: frame z of the launch goes into an intermediate (batch_dst + z * batch_stride) instead of the table's target
    const gptr pdst = (gptr)uniform_ptr(batch_dst ? (void *)(batch_dst + (size_t)blockIdx.z * batch_stride) : frame.dst);

    const f2 MM[5] = {f2{P.m[0], P.m[1]}, f2{P.m[2], P.m[3]}, f2{P.m[4], P.m[5]}, f2{P.m[6], P.m[7]}, f2{P.m[8], 0.0f}};
    const f2 GG[5] = {f2{P.gamut[0], P.gamut[1]}, f2{P.gamut[2], P.gamut[3]}, f2{P.gamut[4], P.gamut[5]}, f2{P.gamut[6], P.gamut[7]}, f2{P.gamut[8], 0.0f}};
    const f2 cmax2 = splat(P.maxv);
    f2 big2 = splat(8388608.0f);
    asm volatile("" : "+v"(big2));
    f2 CC[3] = {splat(P.c[0]), splat(P.c[1]), splat(P.c[2])};
    asm volatile("" : "+v"(CC[0]), "+v"(CC[1]), "+v"(CC[2]));

    RawAddr ra;
    make_raw_addr<SRC>(P, X, ra);
    const uint32_t lane_off = (uint32_t)(P.off_x + X) * 4u;
    Raw raw;
    {
        const int a = 2 * pair0 - 1;
        load_raw<SRC>(P, py, ra, clampi(a, 0, H - 1), clampi(a + 1, 0, H - 1), raw);
    }
    for (int p = 0; p < pairs; p++) {
        const int a = 2 * (pair0 + p) - 1;                             // rows a, a+1
        if (a >= H) break;
        f2 rc[2][3];
        convert_block<TAIL, SRC>(P, MM, GG, CC, raw, P.rect_t + clampi(a, 0, H - 1), P.rect_t + clampi(a + 1, 0, H - 1), T, rc);
        if (p + 1 < pairs && a + 2 < H)
            load_raw<SRC>(P, py, ra, clampi(a + 2, 0, H - 1), clampi(a + 3, 0, H - 1), raw);
        // UNORM store of m_TexConvertOutput: floor(sat(x)*maxv + 0.5); x*maxv + 2^23 leaves the code in the low mantissa bits
        uint32_t code[2][3][2];                                        // [column][channel][row]
#pragma unroll
        for (int col = 0; col < 2; col++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const f2 q = pk_fma(rc[col][c], cmax2, big2);
                code[col][c][0] = __float_as_uint(q.x) & 0xffffu; code[col][c][1] = __float_as_uint(q.y) & 0xffffu;
            }
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int y = a + r;
            if (y < 0 || y >= H) continue;                             // wave-uniform
            const int wy = P.off_y + y;
            uint32_t px[2];
#pragma unroll
            for (int col = 0; col < 2; col++) {
                const uint32_t cr = code[col][0][r], cg = code[col][1][r], cb = code[col][2][r];
                if (FINAL) {
                    const uint32_t dj = Di[(wy & 31) * 32 + ((P.off_x + X + col) & 31)];
                    const uint32_t ib = __umul24(cb, P.epi_mul) + dj, ig = __umul24(cg, P.epi_mul) + dj, ir = __umul24(cr, P.epi_mul) + dj;
                    const uint32_t bg = __builtin_amdgcn_perm(ig, ib, 0x0c0c0703u);
                    px[col] = __builtin_amdgcn_perm(ir, bg, 0x0d070100u);
                } else if (P.out10) {
                    px[col] = cr | (cg << 10) | (cb << 20) | 0xc0000000u;
                } else {
                    px[col] = cb | (cg << 8) | (cr << 16) | 0xff000000u;
                }
            }
            const gptr rowp = pdst + (uint32_t)wy * (uint32_t)P.dst_pitch;
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            *(__attribute__((address_space(1))) u32x2 *)(rowp + opaque(lane_off)) = u32x2{px[0], px[1]};
        }
    }
}

// The same, four blocks (8 columns x 2 rows) per lane for bi-planar sources whose rows allow it: one 8- / 16-byte load per
// luma row and chroma row instead of four 2- / 4-byte ones, two 16-byte stores per output row.  Same convert_block, same
// results; it only changes how the bytes travel.  SRC is SRC_NV12 or SRC_P01X.
template <int TAIL, int SRC, bool FINAL>
__global__ __launch_bounds__(256) void k_convert_blocks_wide(FusedArgs P, const FusedFrame *__restrict__ frames, FusedFrame single, int pairs,
                                                            uint8_t *batch_dst, size_t batch_stride, FrameTable32 tab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *Di = (uint32_t *)smem;
    f2 *T = (f2 *)(smem + (FINAL ? 4096 : 0));
    if (FINAL)
        for (int i = threadIdx.x; i < 1024; i += 256)
            Di[i] = (uint32_t)(__half2float(__ushort_as_half(P.dither[i])) * 1024.0f + 0.5f) << 14;
    if (TAIL == TAILK_PQ_LUT)
        for (int i = threadIdx.x; i < LUT_N; i += 256) {
            const float v = P.lut[i], n = P.lut[min(i + 1, LUT_N - 1)];
            T[i] = f2{v, n - v};
        }
    if (FINAL || TAIL == TAILK_PQ_LUT) __syncthreads();

    constexpr bool WIDE16 = SRC == SRC_P01X;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int W = P.W, H = P.H;
    const int X = blockIdx.x * 512 + 8 * lane;                         // rect columns X .. X+7
    const int pair0 = (blockIdx.y * 4 + wave) * pairs;
    if (X >= W || 2 * pair0 - 1 >= H) return;
    const FusedFrame frame = tab.n ? tab.f[blockIdx.z] : frames ? frames[blockIdx.z] : single;
    auto uniform_ptr = [](const void *q) {
        const uint64_t v = (uint64_t)q;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const gcptr py = (gcptr)uniform_ptr(frame.src);
    const gptr pdst = (gptr)uniform_ptr(batch_dst ? (void *)(batch_dst + (size_t)blockIdx.z * batch_stride) : frame.dst);

    const f2 MM[5] = {f2{P.m[0], P.m[1]}, f2{P.m[2], P.m[3]}, f2{P.m[4], P.m[5]}, f2{P.m[6], P.m[7]}, f2{P.m[8], 0.0f}};
    const f2 GG[5] = {f2{P.gamut[0], P.gamut[1]}, f2{P.gamut[2], P.gamut[3]}, f2{P.gamut[4], P.gamut[5]}, f2{P.gamut[6], P.gamut[7]}, f2{P.gamut[8], 0.0f}};
    const f2 cmax2 = splat(P.maxv);
    f2 big2 = splat(8388608.0f);
    asm volatile("" : "+v"(big2));
    f2 CC[3] = {splat(P.c[0]), splat(P.c[1]), splat(P.c[2])};
    asm volatile("" : "+v"(CC[0]), "+v"(CC[1]), "+v"(CC[2]));

    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int sx0 = P.rect_l + X, c0 = sx0 >> 1;
    const uint32_t yoff = (uint32_t)(WIDE16 ? 2 * sx0 : sx0);           // luma bytes; the UV row starts at the same byte offset
    const uint32_t xoff = (uint32_t)((WIDE16 ? 4 : 2) * clampi(c0 + 4, 0, P.cw - 1));     // the chroma texel right of the lane's four
    const uint32_t lane_off = (uint32_t)(P.off_x + X) * 4u;
    const gcptr pu = py + P.off_u;
    // luma rows a, a+1 and chroma rows n, n+1 of a pair: 4 wide loads + 2 for the neighbour texel
    uint32_t L[2][4], Cc[2][5];
    auto load_pair = [&](int a) {
        const int sy0 = P.rect_t + clampi(a, 0, H - 1), sy1 = P.rect_t + clampi(a + 1, 0, H - 1);
        const int n = chroma_v4(P, sy0) >> 2;
        const gcptr ry[2] = {py + (uint32_t)sy0 * (uint32_t)P.pitch_y, py + (uint32_t)sy1 * (uint32_t)P.pitch_y};
        const gcptr rc[2] = {pu + (uint32_t)clampi(n, 0, P.ch - 1) * (uint32_t)P.pitch_c, pu + (uint32_t)clampi(n + 1, 0, P.ch - 1) * (uint32_t)P.pitch_c};
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (WIDE16) {
                const u32x4 l = *(const __attribute__((address_space(1))) u32x4 *)(ry[r] + opaque(yoff));
                const u32x4 c = *(const __attribute__((address_space(1))) u32x4 *)(rc[r] + opaque(yoff));
                L[r][0] = l.x; L[r][1] = l.y; L[r][2] = l.z; L[r][3] = l.w;
                Cc[r][0] = c.x; Cc[r][1] = c.y; Cc[r][2] = c.z; Cc[r][3] = c.w;
                Cc[r][4] = ld_u32(rc[r] + opaque(xoff));
            } else {
                const u32x2 l = *(const __attribute__((address_space(1))) u32x2 *)(ry[r] + opaque(yoff));
                const u32x2 c = *(const __attribute__((address_space(1))) u32x2 *)(rc[r] + opaque(yoff));
                L[r][0] = l.x & 0xffffu; L[r][1] = l.x >> 16; L[r][2] = l.y & 0xffffu; L[r][3] = l.y >> 16;
                const uint32_t e = ld_u16(rc[r] + opaque(xoff));
                const uint32_t pr[5] = {c.x & 0xffffu, c.x >> 16, c.y & 0xffffu, c.y >> 16, e};
#pragma unroll
                for (int i = 0; i < 5; i++) Cc[r][i] = (pr[i] & 0xffu) | ((pr[i] >> 8) << 16);       // U | V << 16
            }
        }
    };
    load_pair(2 * pair0 - 1);
    for (int p = 0; p < pairs; p++) {
        const int a = 2 * (pair0 + p) - 1;
        if (a >= H) break;
        uint32_t px[2][8];                                             // [row][column]
#pragma unroll
        for (int b = 0; b < 4; b++) {
            Raw raw;
            raw.y[0] = L[0][b]; raw.y[1] = L[1][b];
            raw.c[0][0] = raw.c[1][0] = 0;
            raw.c[0][1] = Cc[0][b]; raw.c[0][2] = Cc[0][b + 1];
            raw.c[1][1] = Cc[1][b]; raw.c[1][2] = Cc[1][b + 1];
            f2 rc[2][3];
            convert_block<TAIL, SRC>(P, MM, GG, CC, raw, P.rect_t + clampi(a, 0, H - 1), P.rect_t + clampi(a + 1, 0, H - 1), T, rc);
#pragma unroll
            for (int col = 0; col < 2; col++) {
                uint32_t code[3][2];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const f2 q = pk_fma(rc[col][c], cmax2, big2);
                    code[c][0] = __float_as_uint(q.x) & 0xffffu; code[c][1] = __float_as_uint(q.y) & 0xffffu;
                }
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const uint32_t cr = code[0][r], cg = code[1][r], cb = code[2][r];
                    if (FINAL) {
                        const int wy = P.off_y + a + r;
                        const uint32_t dj = Di[(wy & 31) * 32 + ((P.off_x + X + 2 * b + col) & 31)];
                        const uint32_t ib = __umul24(cb, P.epi_mul) + dj, ig = __umul24(cg, P.epi_mul) + dj, ir = __umul24(cr, P.epi_mul) + dj;
                        const uint32_t bg = __builtin_amdgcn_perm(ig, ib, 0x0c0c0703u);
                        px[r][2 * b + col] = __builtin_amdgcn_perm(ir, bg, 0x0d070100u);
                    } else if (P.out10) {
                        px[r][2 * b + col] = cr | (cg << 10) | (cb << 20) | 0xc0000000u;
                    } else {
                        px[r][2 * b + col] = cb | (cg << 8) | (cr << 16) | 0xff000000u;
                    }
                }
            }
        }
        if (p + 1 < pairs && a + 2 < H) load_pair(a + 2);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int y = a + r;
            if (y < 0 || y >= H) continue;
            const gptr rowp = pdst + (uint32_t)(P.off_y + y) * (uint32_t)P.dst_pitch;
            *(__attribute__((address_space(1))) u32x4 *)(rowp + opaque(lane_off)) = u32x4{px[r][0], px[r][1], px[r][2], px[r][3]};
            *(__attribute__((address_space(1))) u32x4 *)(rowp + opaque(lane_off) + 16) = u32x4{px[r][4], px[r][5], px[r][6], px[r][7]};
        }
    }
}

int EnvInt(const char *name, int def)
{
    const char *v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : def;
}

}  // namespace

bool FusedUp2xSupported(const FusedParams &P)
{
    const ConvertParams &c = P.conv;
    if (P.out_w != 2 * c.out_w || P.out_h != 2 * c.out_h) return false;
    if (P.wx.ntaps != 4 && P.wx.ntaps != 6) return false;
    if (P.wy.ntaps != P.wx.ntaps || P.wy.q1_quirk != P.wx.q1_quirk) return false;
    if (std::memcmp(P.wx.w_even, P.wy.w_even, sizeof(P.wx.w_even)) || std::memcmp(P.wx.w_odd, P.wy.w_odd, sizeof(P.wx.w_odd))) return false;
    if (c.out_fmt != SF_BGRA8 && c.out_fmt != SF_RGB10A2) return false;
    if (c.fmt.subsampling != 420 || c.chroma_scaling != 1 || c.blend_deint) return false;
    if (c.out_w < 8 || c.out_h < 8 || (c.out_w & 1) || (c.out_h & 1)) return false;
    if (!P.fast_convert) return false;            // dword loads need aligned rows / rect (host-checked)
    // 32-bit row offsets inside the kernel
    if ((uint64_t)c.pitch[0] * (uint64_t)(c.rect_t + c.out_h + 2) >= (1ull << 32)) return false;
    if ((uint64_t)P.plane_off[1] >= (1ull << 31) || (uint64_t)P.plane_off[2] >= (1ull << 31)) return false;
    if ((uint64_t)P.store.dst_pitch * (uint64_t)(P.store.off_y + P.out_h) >= (1ull << 32)) return false;
    return true;
}

// the convert-side and store-side constants both kernels of this file take
static void FillFusedArgs(const FusedParams &P, FusedArgs &a)
{
    const ConvertParams &c = P.conv;
    std::memset(&a, 0, sizeof(a));
    const bool swap_uv = c.fmt.planes == 3 && c.fmt.v_first;
    a.off_u = (uint32_t)(swap_uv ? P.plane_off[2] : P.plane_off[1]);
    a.off_v = (uint32_t)(swap_uv ? P.plane_off[1] : P.plane_off[2]);
    a.pitch_y = c.pitch[0]; a.pitch_c = c.pitch[1];
    a.tex_w = c.tex_w; a.cw = c.cw; a.ch = c.ch;
    a.rect_l = c.rect_l; a.rect_t = c.rect_t; a.W = c.out_w; a.H = c.out_h;
    a.bytes = c.fmt.bytes; a.planes = c.fmt.planes;
    a.center_h = c.chroma_loc == CLOC_MPEG1;
    a.v_off4 = c.chroma_loc == CLOC_COSITED ? 1 : 0;
    // UNORM scale: v/255, or (v << shift)/65535 for planar data; interleaved UV planes carry no shift
    const float sy = c.fmt.bytes == 1 ? 1.0f / 255.0f : (float)(1 << c.fmt.shift) / 65535.0f;
    const float sc = c.fmt.bytes == 1 ? 1.0f / 255.0f : (float)(1 << (c.fmt.planes == 2 ? 0 : c.fmt.shift)) / 65535.0f;
    for (int i = 0; i < 3; i++) {
        a.m[3 * i + 0] = c.cm[3 * i + 0] * sy;
        a.m[3 * i + 1] = c.cm[3 * i + 1] * sc;
        a.m[3 * i + 2] = c.cm[3 * i + 2] * sc;
        a.c[i] = c.cm[9 + i];
    }
    a.tail = c.tail; a.gamma = c.gamma; a.lum_scale = c.lum_scale;
    std::memcpy(a.gamut, c.gamut, sizeof(a.gamut));
    a.lut = P.pq_lut;
    a.maxv = c.out_fmt == SF_RGB10A2 ? 1023.0f : 255.0f;
    a.inv_maxv = 1.0f / a.maxv;
    a.q_over_maxv = (float)P.store.quant / a.maxv;
    a.epi_mul = FinalPassMultiplier(P.store.quant, (int)a.maxv);
    a.dst_pitch = P.store.dst_pitch; a.off_x = P.store.off_x; a.off_y = P.store.off_y;
    a.final_pass = P.store.mode == ST_FINAL; a.out10 = P.store.dst_fmt == SF_RGB10A2;
    a.quant = (float)P.store.quant;
    a.dither = P.store.dither;
}

static int TailKind(const FusedParams &P)
{
    const ConvertParams &c = P.conv;
    // P.pq_lut is null when MPCVR_FLAG_NO_LUT asks for the literal ALU chains (A/B testing)
    return c.tail == TAIL_NONE ? TAILK_NONE : (c.tail == TAIL_PQ_TO_SDR && P.pq_lut) ? TAILK_PQ_LUT
         : (c.tail == TAIL_HLG_TO_SDR && !P.literal_tail) ? TAILK_HLG : TAILK_ALU;
}
static int SourceKind(const FusedParams &P)
{
    // source specialisations: bi-planar 16-bit (P010/P016) and bi-planar 8-bit (NV12) with MPEG-2 / co-sited chroma;
    // everything else (planar, MPEG-1 siting) runs through the variant that reads these properties at run time
    const ConvertParams &c = P.conv;
    const bool biplanar_fast = c.fmt.planes == 2 && c.chroma_loc != CLOC_MPEG1;
    return (biplanar_fast && c.fmt.bytes == 2) ? SRC_P01X : (biplanar_fast && c.fmt.bytes == 1) ? SRC_NV12 : SRC_GENERIC;
}

bool ConvertBlocksSupported(const FusedParams &P, bool to_rt)
{
    const ConvertParams &c = P.conv;
    if (c.out_fmt != SF_BGRA8 && c.out_fmt != SF_RGB10A2) return false;
    if (c.fmt.layout != LAY_PLANAR || c.fmt.subsampling != 420 || c.chroma_scaling != 1 || c.blend_deint || c.dovi) return false;
    if (c.out_w < 8 || c.out_h < 2 || (c.out_w & 1) || (c.out_h & 1)) return false;
    if (!P.fast_convert) return false;
    if ((uint64_t)c.pitch[0] * (uint64_t)(c.rect_t + c.out_h + 2) >= (1ull << 32)) return false;
    if ((uint64_t)P.plane_off[1] >= (1ull << 31) || (uint64_t)P.plane_off[2] >= (1ull << 31)) return false;
    const StoreParams &st = P.store;
    if ((uint64_t)st.dst_pitch * (uint64_t)(st.off_y + c.out_h) >= (1ull << 32)) return false;
    if ((st.dst_pitch & 7) || (st.off_x & 1) || !P.dst_aligned16) return false;                 // 8-byte stores
    if (to_rt) {
        // the whole rect inside the window (no per-pixel clipping here)
        if (st.off_x < 0 || st.off_y < 0 || (st.clip_w > 0 && (st.off_x + c.out_w > st.clip_w || st.off_y + c.out_h > st.clip_h))) return false;
        if (st.mode == ST_FINAL)
            return c.out_fmt == SF_RGB10A2 && st.mid_fmt == SF_RGB10A2 && st.dst_fmt == SF_BGRA8 && st.quant == 255 &&
                   FinalPassMultiplier(255, 1023) != 0;
        return st.dst_fmt == c.out_fmt;                                   // straight copy of the internal-format texels
    }
    return st.dst_fmt == c.out_fmt && st.mode == ST_SURFACE;
}

hipError_t LaunchConvertBlocks(const FusedParams &P, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s,
                               size_t batch_stride, const FusedFrame *frames_host)
{
    uint8_t *batch_dst = batch_stride ? (uint8_t *)P.store.dst : nullptr;
    // up to 32 frames travel in the kernel arguments (no table upload in front of the launch)
    FrameTable32 tab;
    tab.n = 0;
    if (frames_host && n_frames <= 32) {
        tab.n = n_frames;
        for (int i = 0; i < n_frames; i++) tab.f[i] = frames_host[i];
        for (int i = n_frames; i < 32; i++) tab.f[i] = FusedFrame{nullptr, nullptr};
    } else if (!frames_dev && n_frames != 1) return hipErrorInvalidValue;
    FusedArgs a;
    FillFusedArgs(P, a);
    const ConvertParams &c = P.conv;
    const bool fin = P.store.mode == ST_FINAL;
    const int tailk = TailKind(P), srck = SourceKind(P);
    // four blocks per lane (8- / 16-byte loads, 16-byte stores) where every row allows it
    static const int no_wide = EnvInt("MPCVR_NO_WIDE_CONVERT", 0);
    const int lb = srck == SRC_P01X ? 16 : 8;                   // bytes of a lane's luma / chroma load
    const bool wide = !no_wide && srck != SRC_GENERIC && (c.out_w & 7) == 0 && (c.rect_l & 7) == 0 && (c.pitch[0] % lb) == 0 &&
                      (c.pitch[1] % lb) == 0 && (P.plane_off[1] % lb) == 0 && P.dst_aligned16 && (P.store.off_x & 3) == 0 &&
                      (P.store.dst_pitch & 15) == 0 && P.src_aligned16;
    const int strip_w = wide ? 512 : 128;
    const int strips = (c.out_w + strip_w - 1) / strip_w, npairs = c.out_h / 2 + 1;
    // row pairs per wave: enough waves to fill the chip a few times over, few enough to amortise the table staging
    int pairs = 16;
    while (pairs > 2 && (long)strips * ((npairs + pairs - 1) / pairs) * n_frames < (wide ? 4096 : 8192)) pairs >>= 1;
    const dim3 grid(strips, (npairs + 4 * pairs - 1) / (4 * pairs), n_frames), block(256, 1, 1);
    const size_t lds = (fin ? 4096 : 0) + (tailk == TAILK_PQ_LUT ? LDS_T : 0);
#define MPCVR_CB3(TK, SK, FN) hipLaunchKernelGGL((k_convert_blocks<TK, SK, FN>), grid, block, lds, s, a, frames_dev, single, pairs, batch_dst, batch_stride, tab)
#define MPCVR_CBW(TK, SK, FN) hipLaunchKernelGGL((k_convert_blocks_wide<TK, SK, FN>), grid, block, lds, s, a, frames_dev, single, pairs, batch_dst, batch_stride, tab)
#define MPCVR_CB2(TK, SK) do { if (fin) MPCVR_CB3(TK, SK, true); else MPCVR_CB3(TK, SK, false); } while (0)
#define MPCVR_CBW2(TK, SK) do { if (fin) MPCVR_CBW(TK, SK, true); else MPCVR_CBW(TK, SK, false); } while (0)
#define MPCVR_CB(TK) do { if (wide && srck == SRC_P01X) MPCVR_CBW2(TK, SRC_P01X); else if (wide) MPCVR_CBW2(TK, SRC_NV12); \
                          else if (srck == SRC_P01X) MPCVR_CB2(TK, SRC_P01X); else if (srck == SRC_NV12) MPCVR_CB2(TK, SRC_NV12); else MPCVR_CB2(TK, SRC_GENERIC); } while (0)
    if (tailk == TAILK_NONE) MPCVR_CB(TAILK_NONE);
    else if (tailk == TAILK_PQ_LUT) MPCVR_CB(TAILK_PQ_LUT);
    else if (tailk == TAILK_HLG) MPCVR_CB(TAILK_HLG);
    else MPCVR_CB(TAILK_ALU);
#undef MPCVR_CB
#undef MPCVR_CBW2
#undef MPCVR_CB2
#undef MPCVR_CBW
#undef MPCVR_CB3
    return hipGetLastError();
}

hipError_t LaunchFusedUp2x(const FusedParams &P, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s)
{
    static const int seg_env = EnvInt("MPCVR_FUSED_SEG", 0);
    if (!frames_dev && n_frames != 1) return hipErrorInvalidValue;
    const ConvertParams &c = P.conv;
    FusedArgs a;
    FillFusedArgs(P, a);
    const int nt = P.wx.ntaps;
    for (int t = 0; t < 6; t++) { a.we[t] = P.wx.w_even[t]; a.wo[t] = P.wx.w_odd[t]; }
    int knt = nt;
    if (nt == 6 && P.wx.q1_quirk) {   // taps 0 and 1 read the same texel: fold tap 1's weight into tap 0 => 5 taps
        a.we[0] += a.we[1]; a.wo[0] += a.wo[1];
        for (int t = 1; t < 5; t++) { a.we[t] = a.we[t + 1]; a.wo[t] = a.wo[t + 1]; }
        a.we[5] = a.wo[5] = 0.0f;
        knt = 5;
    }

    const int strips = (c.out_w + S - 1) / S;
    // segment height: long segments recompute less (6 rows each), short ones balance the last round of waves;
    // aim for >= 4 rounds of the ~3072 resident waves
    int seg = seg_env;
    if (seg <= 0) {
        seg = 72;
        for (int cand : {180, 144, 120, 108, 90, 72, 60, 48, 36, 24})
            if ((long)strips * ((c.out_h + cand - 1) / cand) * n_frames >= 12288 || cand == 24) { seg = cand; break; }
    }
    seg = (seg + 1) & ~1;
    if (seg > c.out_h) seg = c.out_h;
    a.seg_rows = seg;

    const dim3 grid((strips + WAVES - 1) / WAVES, (c.out_h + seg - 1) / seg, n_frames);
    const dim3 block(256, 1, 1);
    const int tailk = TailKind(P);
    static const int lds_pad = EnvInt("MPCVR_FUSED_LDS_PAD", 0);   // experiments: lower the occupancy by claiming more LDS
    const size_t lds = LDS_A + LDS_D + LDS_DB + (tailk == TAILK_PQ_LUT ? LDS_T : 0) + (size_t)lds_pad;
    const int srck = SourceKind(P);
    // the specialised epilogues use 16-byte stores / dither reads: off_x % 4 == 0 and 16-byte aligned rows; the integer
    // final pass additionally needs k*M + (j << 14) < 2^32 and M < 2^24 (true for 10-bit internal -> 8-bit target)
    const bool aligned = P.dst_aligned16 && (a.off_x & 3) == 0 && (a.dst_pitch & 15) == 0;
    const int epik = !aligned || a.out10 ? EPI_GENERIC
                   : (a.final_pass && a.epi_mul != 0) ? EPI_DITHER8
                   : (!a.final_pass && P.store.dst_fmt == SF_BGRA8 && P.store.quant == 255) ? EPI_DIRECT8 : EPI_GENERIC;
    // instantiated (source, epilogue) pairs: each source with the epilogue it normally meets + the generic one
#define MPCVR_LAUNCH3(NT, TK, SK, EK) hipLaunchKernelGGL((k_fused_up2x<NT, TK, SK, EK>), grid, block, lds, s, a, frames_dev, single)
#define MPCVR_LAUNCH(NT, TK) do { \
        if (srck == SRC_P01X && epik == EPI_DITHER8) MPCVR_LAUNCH3(NT, TK, SRC_P01X, EPI_DITHER8); \
        else if (srck == SRC_P01X) MPCVR_LAUNCH3(NT, TK, SRC_P01X, EPI_GENERIC); \
        else if (srck == SRC_NV12 && epik == EPI_DIRECT8) MPCVR_LAUNCH3(NT, TK, SRC_NV12, EPI_DIRECT8); \
        else if (srck == SRC_NV12) MPCVR_LAUNCH3(NT, TK, SRC_NV12, EPI_GENERIC); \
        else if (epik == EPI_DITHER8) MPCVR_LAUNCH3(NT, TK, SRC_GENERIC, EPI_DITHER8); \
        else MPCVR_LAUNCH3(NT, TK, SRC_GENERIC, EPI_GENERIC); } while (0)
#define MPCVR_LAUNCH_NT(NT) \
    do { if (tailk == TAILK_NONE) MPCVR_LAUNCH(NT, TAILK_NONE); else if (tailk == TAILK_PQ_LUT) MPCVR_LAUNCH(NT, TAILK_PQ_LUT); \
         else if (tailk == TAILK_HLG) MPCVR_LAUNCH(NT, TAILK_HLG); else MPCVR_LAUNCH(NT, TAILK_ALU); } while (0)
    if (knt == 4) MPCVR_LAUNCH_NT(4);
    else if (knt == 5) MPCVR_LAUNCH_NT(5);
    else MPCVR_LAUNCH_NT(6);
#undef MPCVR_LAUNCH_NT
#undef MPCVR_LAUNCH
#undef MPCVR_LAUNCH3
    return hipGetLastError();
}

}  // namespace mpcvr
