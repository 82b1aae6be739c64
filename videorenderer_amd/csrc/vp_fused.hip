// vp_fused.hip — host side of the fused exact-2x path (kernel: vp_fused_up2x.h) and the block-convert kernels.
// The fused exact-2x path: convert -> X pass -> Y pass -> final pass in ONE kernel.
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
#include "vp_fused_dev.h"

#include <map>
#include <mutex>
#include <utility>

namespace mpcvr {

namespace {

// ------------------------------------------------------------------------------------------------
// The convert stage on its own (pass-per-kernel path and same-size frames): the 2x2-block convert of the fused kernel —
// shared chroma fetch, scalar siting, packed matrix, table tone map — without the resize behind it.  A wave owns a strip of
// 128 rect columns and walks `pairs` row pairs (a, a+1), a odd (so both rows take their chroma from the same two chroma
// rows); the first and the last pair of a frame clamp to one useful row.
// FINAL = false: the block is stored as texels of the internal UNORM format (m_TexConvertOutput, or the render target when
// nothing follows); FINAL = true: 10-bit internal -> ps_final_pass in integers -> B8G8R8A8 (see the fused epilogue).
// ------------------------------------------------------------------------------------------------
// DV != DV_NONE: the Dolby Vision variant (TAIL is then TAILK_ALU: the tone-map table is not used) — LDS holds the PQ EOTF table
// and a copy of the frame's DoviParams instead.
// CHR = 1: CHROMA_CatmullRom instead of CHROMA_Bilinear (4 x 5 chroma texels per block, convert_block_cr).
template <int TAIL, int SRC, bool FINAL, int DV, int CHR, int XC>
__device__ __forceinline__ void convert_blocks_body(const FusedArgs &P, const FusedFrame *__restrict__ frames, const FusedFrame &single, int pairs,
                                                    uint8_t *batch_dst, size_t batch_stride, const FrameTable32 &tab)
{
    // waves per workgroup: the tables are per workgroup, so the variant with two of them (67 KiB) shares them among 8 waves —
    // two workgroups per CU are then 4 waves per SIMD instead of 2
    constexpr int NTH = DV == DV_SDR_L2 ? 512 : 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *Di = (uint32_t *)smem;                                   // dither as j << 14 (FINAL)
    f2 *T = (f2 *)(smem + (FINAL ? 4096 : 0));
    float *TE = (float *)T;                                            // DV: the EOTF table takes the tone-map table's place ...
    DoviParams *DL = (DoviParams *)(smem + (FINAL ? 4096 : 0) + LDS_E);
    if (DV == DV_SDR_L2) T = (f2 *)(smem + (FINAL ? 4096 : 0) + LDS_E + LDS_V);     // ... and the tone-map table follows the curves
    if (FINAL)
        for (int i = threadIdx.x; i < 1024; i += NTH)
            Di[i] = (uint32_t)(__half2float(__ushort_as_half(P.dither[i])) * 1024.0f + 0.5f) << 14;
    if (DV != DV_NONE) {
        if (DV == DV_SDR || DV == DV_SDR_L2)
            for (int i = threadIdx.x; i < EOTF_N + 2; i += NTH) TE[i] = P.eotf_lut[min(i, EOTF_N)];       // (one pad entry: t = N reads N, N + 1)
        if (DV == DV_SDR_L2) {
            for (int i = threadIdx.x; i < LUT_N; i += NTH) {
                const float v = P.lut[i], n = P.lut[min(i + 1, LUT_N - 1)];
                T[i] = f2{v, n - v};
            }
            // the PQ encode table ({value, slope}) behind the tone-map table: convert_block finds it at T + LUT_N
            for (int i = threadIdx.x; i < kPqEncSize; i += NTH) {
                const float v = P.eotf_lut[kPqEncOffset + i], n = P.eotf_lut[kPqEncOffset + i + 1];
                T[LUT_N + i] = f2{v, n - v};
            }
        }
        const DoviParams *dvp = P.dovi + (P.dovi_per_frame ? blockIdx.z : 0u);          // (one RPU per frame of the batch, or one for the launch)
        for (int i = threadIdx.x; i < (int)(sizeof(DoviParams) / 4); i += NTH) ((uint32_t *)DL)[i] = ((const uint32_t *)dvp)[i];
    } else if (tail_has_table(TAIL))
        for (int i = threadIdx.x; i < LUT_N; i += NTH) {
            const float v = P.lut[i], n = P.lut[min(i + 1, LUT_N - 1)];
            T[i] = f2{v, n - v};
        }
    if (FINAL || tail_has_table(TAIL) || DV != DV_NONE) __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int W = P.W, H = P.H;
    const int X = blockIdx.x * 128 + 2 * lane;                         // rect columns X, X+1
    const int pair0 = (blockIdx.y * (NTH / 64) + wave) * pairs;                 // pair p covers rect rows 2p-1, 2p
    if (X >= W || 2 * pair0 - 1 >= H) return;
    const FusedFrame frame = tab.n ? tab.f[blockIdx.z] : frames ? frames[blockIdx.z] : single;     // tab: the table in the kernel arguments
    auto uniform_ptr = [](const void *q) {
        const uint64_t v = (uint64_t)q;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const gcptr py = (gcptr)uniform_ptr(frame.src);
    // batch_dst: frame z of the launch goes into an intermediate (batch_dst + z * batch_stride) instead of the table's target
    const gptr pdst = (gptr)uniform_ptr(batch_dst ? (void *)(batch_dst + (size_t)blockIdx.z * batch_stride) : frame.dst);

    // colour matrix: the launch's, or (Dolby Vision, one RPU per frame) frame z's own — a wave-uniform read
    float m9[9] = {P.m[0], P.m[1], P.m[2], P.m[3], P.m[4], P.m[5], P.m[6], P.m[7], P.m[8]}, c3[3] = {P.c[0], P.c[1], P.c[2]};
    if (DV != DV_NONE && P.dovi_per_frame) {
        const dv_cptr<float> t = (dv_cptr<float>)(uintptr_t)(P.dovi_cm + 12u * blockIdx.z);       // (read-only for the launch: scalar loads)
        for (int i = 0; i < 9; i++) m9[i] = t[i];
        for (int i = 0; i < 3; i++) c3[i] = t[9 + i];
    }
    const f2 MM[5] = {f2{m9[0], m9[1]}, f2{m9[2], m9[3]}, f2{m9[4], m9[5]}, f2{m9[6], m9[7]}, f2{m9[8], 0.0f}};
    const f2 GG[5] = {f2{P.gamut[0], P.gamut[1]}, f2{P.gamut[2], P.gamut[3]}, f2{P.gamut[4], P.gamut[5]}, f2{P.gamut[6], P.gamut[7]}, f2{P.gamut[8], 0.0f}};
    const f2 cmax2 = splat(P.maxv);
    f2 big2 = splat(8388608.0f);
    asm volatile("" : "+v"(big2));
    f2 CC[3] = {splat(c3[0]), splat(c3[1]), splat(c3[2])};
    asm volatile("" : "+v"(CC[0]), "+v"(CC[1]), "+v"(CC[2]));

    DoviRegs DRG;
    if (DV != DV_NONE) load_dovi_regs(P.dovi + (P.dovi_per_frame ? blockIdx.z : 0u), DRG);
    RawAddr ra;
    RawAddrCR rac;
    if (CHR) make_raw_addr_cr<SRC>(P, X, rac); else make_raw_addr<SRC>(P, X, ra);
    const uint32_t lane_off = (uint32_t)(P.off_x + X) * 4u;
    Raw raw;
    RawCR rawc;
    {
        const int a = 2 * pair0 - 1;
        if (CHR) load_raw_cr<SRC>(P, py, rac, clampi(a, 0, H - 1), clampi(a + 1, 0, H - 1), rawc);
        else load_raw<SRC>(P, py, ra, clampi(a, 0, H - 1), clampi(a + 1, 0, H - 1), raw);
    }
    for (int p = 0; p < pairs; p++) {
        const int a = 2 * (pair0 + p) - 1;                             // rows a, a+1
        if (a >= H) break;
        f2 rc[2][3];
        constexpr int OUTK = XC == XC_ALWAYS ? OUT_CODE_I : OUT_NORM;
        if (CHR) convert_block_cr<TAIL, SRC, DV, XC, OUTK>(P, MM, GG, CC, rawc, P.rect_t + clampi(a, 0, H - 1), P.rect_t + clampi(a + 1, 0, H - 1), T, rc, DL, TE, &DRG);
        else convert_block<TAIL, SRC, DV, XC, OUTK>(P, MM, GG, CC, raw, P.rect_t + clampi(a, 0, H - 1), P.rect_t + clampi(a + 1, 0, H - 1), T, rc, DL, TE, &DRG);
        if (p + 1 < pairs && a + 2 < H) {
            if (CHR) load_raw_cr<SRC>(P, py, rac, clampi(a + 2, 0, H - 1), clampi(a + 3, 0, H - 1), rawc);
            else load_raw<SRC>(P, py, ra, clampi(a + 2, 0, H - 1), clampi(a + 3, 0, H - 1), raw);
        }
        // UNORM store of m_TexConvertOutput: floor(sat(x)*maxv + 0.5); x*maxv + 2^23 leaves the code in the low mantissa bits
        uint32_t code[2][3][2];                                        // [column][channel][row]
#pragma unroll
        for (int col = 0; col < 2; col++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const f2 q = XC == XC_ALWAYS ? rc[col][c] : pk_fma(rc[col][c], cmax2, big2);              // (the exact form hands over the codes as integers)
                code[col][c][0] = __float_as_uint(q.x); code[col][c][1] = __float_as_uint(q.y);      // 0x4B000000 | k: every use below drops the high byte for free
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
                } else if (P.out10) {       // the shifts push 0x4B out of the word; 0x4B + 0x75 = 0xC0 = the two alpha bits
                    px[col] = (cr + (XC == XC_ALWAYS ? 0xC0000000u : 0x75000000u)) | (cg << 10) | (cb << 20);
                } else {
                    const uint32_t bg = __builtin_amdgcn_perm(cg, cb, 0x0c0c0400u);     // [B, G, 0, 0]
                    px[col] = __builtin_amdgcn_perm(cr, bg, 0x0d040100u);               // [B, G, R, 0xff]
                }
            }
            const gptr rowp = pdst + (uint32_t)wy * (uint32_t)P.dst_pitch;
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            *(__attribute__((address_space(1))) u32x2 *)(rowp + opaque(lane_off)) = u32x2{px[0], px[1]};
        }
    }
}

template <int TAIL, int SRC, bool FINAL, int DV = DV_NONE, int CHR = 0, int XC = XC_NEVER>
__global__ __launch_bounds__(DV == DV_SDR_L2 ? 512 : 256) void k_convert_blocks(FusedArgs P, const FusedFrame *__restrict__ frames, FusedFrame single, int pairs,
                                                       uint8_t *batch_dst, size_t batch_stride, FrameTable32 tab)
{
    convert_blocks_body<TAIL, SRC, FINAL, DV, CHR, XC>(P, frames, single, pairs, batch_dst, batch_stride, tab);
}
// the kernel of an instantiation: its exact-form twin where one exists and the launch asks for it (exact_capable, vp_fused_dev.h;
// FINAL = the 10 -> 8 final pass behind it: a 10-bit internal format, the fast form)
template <int TAIL, int SRC, bool FINAL, int CHR>
inline auto convert_blocks_kernel(bool exact) -> decltype(&k_convert_blocks<TAIL, SRC, FINAL, DV_NONE, CHR, XC_NEVER>)
{
    if constexpr (exact_capable<TAIL, SRC, FINAL>() == XC_RUNTIME) { if (exact) return k_convert_blocks<TAIL, SRC, FINAL, DV_NONE, CHR, XC_ALWAYS>; }
    return k_convert_blocks<TAIL, SRC, FINAL, DV_NONE, CHR, XC_NEVER>;
}

// ------------------------------------------------------------------------------------------------
// Same-size frames as a STREAM (round 4): BASELINE configs[0] (1080p NV12 -> BGRA8) is 11.4 MB per frame — a 32-frame launch
// of round 3's 8-columns-per-lane variant of k_convert_blocks lasted ~100 us, its waves lived for two row pairs (prologue, one
// exposed load round trip, two half-line stores per lane and row) and the launch ramped up and drained for a tenth of that
// (same box, profiles/r04/ab_call1_same_box.jsonl: c1 291 k -> 322 k frames/s, 4K P010 PQ -> SDR 54 k -> 67 k).  Here the launch
// holds exactly the waves the chip keeps resident and each of them walks a long run of row pairs:
//   * one strip of 256 rect columns per wave (4 px per lane: ONE 16-byte store per lane and row, 1 KiB contiguous per
//     wavefront) — the strip never changes, so every per-lane byte offset is loop-invariant;
//   * the launch's row pairs — all frames of the batch laid end to end, G = frames x (H/2 + 1) per strip — are dealt to the
//     strip's waves as contiguous runs [g0, g1) that differ by at most one pair: no tail, no quantisation, whatever the
//     frame size; a run crosses frame boundaries (the frame table entry is re-read by scalar loads when it does);
//   * raw codes travel TWO row pairs ahead of the arithmetic (two buffers, used alternately), across frame boundaries too:
//     a wave has 6-12 loads in flight whenever it computes.
// Arithmetic: convert_block (vp_fused_dev.h) on the lane's two 2x2 blocks — the same code, the same results as
// k_convert_blocks.  SRC is SRC_NV12 or SRC_P01X; FINAL as in k_convert_blocks.
// ------------------------------------------------------------------------------------------------
struct StreamArgs {
    int n_strips;              // strips of 256 columns per frame row
    int n_slots;               // waves per strip
    int npairs;                // row pairs per frame: H / 2 + 1 (pair p = rows 2p - 1, 2p)
    int total;                 // n_frames * npairs
    int n_frames;
    uint8_t *batch_dst; size_t batch_stride;
};

// at least six waves per SIMD (<= 80 VGPRs): the PQ-table variants would take 105 for four — measured on one box with experiment builds
// (tools/build_variant.sh, profiles/r04/ab_call4_same_box.jsonl): 4K P010 PQ -> SDR 65.9 k frames/s unconstrained, 66.2 k at five, 67.2 k at six
#ifndef MPCVR_STREAM_WAVES_PER_EU
#define MPCVR_STREAM_WAVES_PER_EU 6
#endif
#define MPCVR_STREAM_OCC __attribute__((amdgpu_waves_per_eu(MPCVR_STREAM_WAVES_PER_EU)))
template <int TAIL, int SRC, bool FINAL>
__global__ __launch_bounds__(512) MPCVR_STREAM_OCC void k_convert_stream(FusedArgs P, const FusedFrame *__restrict__ frames, FusedFrame single, StreamArgs Q, FrameTable128 tab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *Di = (uint32_t *)smem;
    f2 *T = (f2 *)(smem + (FINAL ? 4096 : 0));
    if (FINAL)
        for (int i = threadIdx.x; i < 1024; i += blockDim.x)
            Di[i] = (uint32_t)(__half2float(__ushort_as_half(P.dither[i])) * 1024.0f + 0.5f) << 14;
    if (tail_has_table(TAIL))
        for (int i = threadIdx.x; i < LUT_N; i += blockDim.x) {
            const float v = P.lut[i], n = P.lut[min(i + 1, LUT_N - 1)];
            T[i] = f2{v, n - v};
        }
    if (FINAL || tail_has_table(TAIL)) __syncthreads();

    constexpr bool WIDE16 = SRC == SRC_P01X;
    constexpr int LW = WIDE16 ? 2 : 1;                                 // dwords of a lane's 4 luma samples / 2 chroma texels
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    const int slot = w / Q.n_strips, strip = w - slot * Q.n_strips;
    if (slot >= Q.n_slots) return;
    // the strip's run of global pairs: [g0, g1), sizes differ by at most one
    const int g0 = (int)(((long)Q.total * slot) / Q.n_slots), g1 = (int)(((long)Q.total * (slot + 1)) / Q.n_slots);
    if (g0 >= g1) return;
    const int W = P.W, H = P.H;
    const int X = strip * 256 + 4 * lane;                              // rect columns X .. X+3
    const bool active = X < W;                                         // (W is a multiple of 4: launcher)
    const int Xc = active ? X : W - 4;                                 // lanes beyond the right edge load what the last lane loads and store nothing

    const f2 MM[5] = {f2{P.m[0], P.m[1]}, f2{P.m[2], P.m[3]}, f2{P.m[4], P.m[5]}, f2{P.m[6], P.m[7]}, f2{P.m[8], 0.0f}};
    const f2 GG[5] = {f2{P.gamut[0], P.gamut[1]}, f2{P.gamut[2], P.gamut[3]}, f2{P.gamut[4], P.gamut[5]}, f2{P.gamut[6], P.gamut[7]}, f2{P.gamut[8], 0.0f}};
    const f2 cmax2 = splat(P.maxv);
    f2 big2 = splat(8388608.0f);
    asm volatile("" : "+v"(big2));
    f2 CC[3] = {splat(P.c[0]), splat(P.c[1]), splat(P.c[2])};
    asm volatile("" : "+v"(CC[0]), "+v"(CC[1]), "+v"(CC[2]));

    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int sx0 = P.rect_l + Xc, c0 = sx0 >> 1;
    const uint32_t yoff = (uint32_t)(WIDE16 ? 2 * sx0 : sx0);           // luma bytes of the lane's 4 px; its 2 chroma texels start at the same byte offset of a UV row
    const uint32_t xoff = (uint32_t)((WIDE16 ? 4 : 2) * clampi(c0 + 2, 0, P.cw - 1));     // the chroma texel right of the lane's two
    const uint32_t lane_off = (uint32_t)(P.off_x + Xc) * 4u;
    const uint32_t dix = (uint32_t)((P.off_x + Xc) & 31);              // dither texels of the 4 px: one aligned 16-byte LDS read (off_x % 4 == 0: launcher)

    auto uniform_ptr = [](const void *q) {
        const uint64_t v = (uint64_t)q;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    auto frame_src = [&](int z) -> gcptr {
        const FusedFrame f = tab.n ? tab.f[z] : frames ? frames[z] : single;
        return (gcptr)uniform_ptr(f.src);
    };
    auto frame_dst = [&](int z) -> gptr {
        const FusedFrame f = tab.n ? tab.f[z] : frames ? frames[z] : single;
        return (gptr)uniform_ptr(Q.batch_dst ? (void *)(Q.batch_dst + (size_t)z * Q.batch_stride) : f.dst);
    };

    // raw codes of one row pair: luma rows a, a+1 and chroma rows n, n+1 (the lane's two texels + the one to their right)
    struct RawPair { uint32_t l[2][LW], c[2][LW], x[2]; };
    auto load_pair = [&](gcptr py, int p, RawPair &r) __attribute__((always_inline)) {
        const int a = 2 * p - 1;
        const int sy0 = P.rect_t + clampi(a, 0, H - 1), sy1 = P.rect_t + clampi(a + 1, 0, H - 1);
        const int n = chroma_v4(P, sy0) >> 2;
        const gcptr pu = py + P.off_u;
        const gcptr ry[2] = {py + (uint32_t)sy0 * (uint32_t)P.pitch_y, py + (uint32_t)sy1 * (uint32_t)P.pitch_y};
        const gcptr rc[2] = {pu + (uint32_t)clampi(n, 0, P.ch - 1) * (uint32_t)P.pitch_c, pu + (uint32_t)clampi(n + 1, 0, P.ch - 1) * (uint32_t)P.pitch_c};
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if constexpr (WIDE16) {
                const u32x2 l = *(const __attribute__((address_space(1))) u32x2 *)(ry[k] + opaque(yoff));
                const u32x2 c = *(const __attribute__((address_space(1))) u32x2 *)(rc[k] + opaque(yoff));
                r.l[k][0] = l.x; r.l[k][1] = l.y; r.c[k][0] = c.x; r.c[k][1] = c.y;
                r.x[k] = ld_u32(rc[k] + opaque(xoff));
            } else {
                r.l[k][0] = ld_u32(ry[k] + opaque(yoff));
                r.c[k][0] = ld_u32(rc[k] + opaque(yoff));
                r.x[k] = ld_u16(rc[k] + opaque(xoff));
            }
        }
    };

    // consumer cursor (z, p) and producer cursor (zp, pp), two pairs ahead
    int z = g0 / Q.npairs, p = g0 - z * Q.npairs;
    int zp = z, pp = p;
    gptr pdst = frame_dst(z);
    gcptr py_p = frame_src(z);
    auto advance_producer = [&]() __attribute__((always_inline)) {
        if (++pp == Q.npairs) { pp = 0; zp = min(zp + 1, Q.n_frames - 1); py_p = frame_src(zp); }
    };
    RawPair buf[2];
    load_pair(py_p, pp, buf[0]);
    advance_producer();
    load_pair(py_p, pp, buf[1]);           // (at most two pairs past the run: valid pairs of the batch, read and dropped)
    advance_producer();

    // one row pair: convert the lane's two blocks out of `r`, send the loads of the pair after next into `r`, store both rows
    auto step = [&](RawPair &r) __attribute__((always_inline)) {
        const int a = 2 * p - 1;
        uint32_t code[4][3][2];                                        // [column][channel][row]: 0x4B000000 | UNORM code, see k_convert_blocks
#pragma unroll
        for (int b = 0; b < 2; b++) {
            Raw raw;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                raw.c[k][0] = 0;
                if constexpr (WIDE16) {
                    raw.y[k] = r.l[k][b];
                    raw.c[k][1] = r.c[k][b];
                    raw.c[k][2] = b == 0 ? r.c[k][1] : r.x[k];
                } else {
                    raw.y[k] = b == 0 ? (r.l[k][0] & 0xffffu) : (r.l[k][0] >> 16);
                    const uint32_t e0 = b == 0 ? (r.c[k][0] & 0xffffu) : (r.c[k][0] >> 16), e1 = b == 0 ? (r.c[k][0] >> 16) : r.x[k];
                    raw.c[k][1] = (e0 & 0xffu) | ((e0 >> 8) << 16);    // U | V << 16
                    raw.c[k][2] = (e1 & 0xffu) | ((e1 >> 8) << 16);
                }
            }
            f2 rc[2][3];
            convert_block<TAIL, SRC>(P, MM, GG, CC, raw, P.rect_t + clampi(a, 0, H - 1), P.rect_t + clampi(a + 1, 0, H - 1), T, rc);
#pragma unroll
            for (int col = 0; col < 2; col++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const f2 q = pk_fma(rc[col][c], cmax2, big2);
                    code[2 * b + col][c][0] = __float_as_uint(q.x); code[2 * b + col][c][1] = __float_as_uint(q.y);
                }
        }
        load_pair(py_p, pp, r);
        advance_producer();
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int y = a + k;
            if (y < 0 || y >= H) continue;                             // wave-uniform: the half pairs at a frame's top and bottom
            const int wy = P.off_y + y;
            uint32_t dj[4] = {0, 0, 0, 0};
            if (FINAL) {
                const u32x4 dd = *(const u32x4 *)(Di + (wy & 31) * 32 + dix);
                dj[0] = dd.x; dj[1] = dd.y; dj[2] = dd.z; dj[3] = dd.w;
            }
            uint32_t px[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t cr = code[i][0][k], cg = code[i][1][k], cb = code[i][2][k];
                if (FINAL) {
                    const uint32_t ib = __umul24(cb, P.epi_mul) + dj[i], ig = __umul24(cg, P.epi_mul) + dj[i], ir = __umul24(cr, P.epi_mul) + dj[i];
                    const uint32_t bg = __builtin_amdgcn_perm(ig, ib, 0x0c0c0703u);
                    px[i] = __builtin_amdgcn_perm(ir, bg, 0x0d070100u);
                } else if (P.out10) {
                    px[i] = (cr + 0x75000000u) | (cg << 10) | (cb << 20);
                } else {
                    const uint32_t bg = __builtin_amdgcn_perm(cg, cb, 0x0c0c0400u);     // [B, G, 0, 0]
                    px[i] = __builtin_amdgcn_perm(cr, bg, 0x0d040100u);                 // [B, G, R, 0xff]
                }
            }
            const gptr rowp = pdst + (uint32_t)wy * (uint32_t)P.dst_pitch;
            if (active) *(__attribute__((address_space(1))) u32x4 *)(rowp + opaque(lane_off)) = u32x4{px[0], px[1], px[2], px[3]};
        }
        if (++p == Q.npairs) { p = 0; z = min(z + 1, Q.n_frames - 1); pdst = frame_dst(z); }
    };
    for (int g = g0; g < g1; g += 2) {
        step(buf[0]);
        if (g + 1 < g1) step(buf[1]);
    }
}

}  // namespace

// Source layouts and chroma filters convert_block has a loader and a rule for (vp_fused_dev.h):
//   planar / bi-planar 4:2:0 — nearest, bilinear (and Catmull-Rom where `catmull_420`: the block convert kernel only);
//   planar / bi-planar 4:2:2 — nearest, bilinear;  packed 4:2:2 — its one linear filter (Nearest == Bilinear, Shaders.cpp:195-229);
//   everything with a chroma sample per pixel, where no chroma setting applies — planar 4:4:4 (YUV and G,B,R planes), packed 4:4:4,
//   gray.  Interleaved RGB has no convert stage of this kind.
bool BlockConvertLayout(const FusedParams &P, bool catmull_420)
{
    const ConvertParams &c = P.conv;
    switch (c.fmt.layout) {
    case LAY_PLANAR:
        if (c.fmt.subsampling == 444) return c.fmt.planes == 3;
        if (c.fmt.subsampling == 422) return c.chroma_scaling != 2;
        return c.fmt.subsampling == 420 && (c.chroma_scaling != 2 || catmull_420);
    case LAY_PACKED422: return c.chroma_scaling != 2 && !(c.tex_w & 1) && (c.pitch[0] & 3) == 0;    // whole-texel loads: a dword (8-bit) or two
    case LAY_PACKED444: return (c.pitch[0] & 3) == 0;
    case LAY_GRAY: return true;
    default: return false;
    }
}

bool FusedUp2xSupported(const FusedParams &P)
{
    const ConvertParams &c = P.conv;
    if (P.out_w != 2 * c.out_w || P.out_h != 2 * c.out_h) return false;
    if (!P.jinc_tab) {      // (the 2-D Jinc2m filter brings its own table)
        if (P.wx.ntaps != 4 && P.wx.ntaps != 6) return false;
        if (P.wy.ntaps != P.wx.ntaps || P.wy.q1_quirk != P.wx.q1_quirk) return false;
        if (std::memcmp(P.wx.w_even, P.wy.w_even, sizeof(P.wx.w_even)) || std::memcmp(P.wx.w_odd, P.wy.w_odd, sizeof(P.wx.w_odd))) return false;
    }
    if (c.out_fmt != SF_BGRA8 && c.out_fmt != SF_RGB10A2) return false;
    if (!BlockConvertLayout(P, false) || c.blend_deint) return false;
    if (c.out_w < 8 || c.out_h < 8 || (c.out_w & 1) || (c.out_h & 1)) return false;
    if (!P.fast_convert) return false;            // dword loads need aligned rows / rect (host-checked)
    // 32-bit row offsets inside the kernel
    if ((uint64_t)c.pitch[0] * (uint64_t)(c.rect_t + c.out_h + 2) >= (1ull << 32)) return false;
    if ((uint64_t)P.plane_off[1] >= (1ull << 31) || (uint64_t)P.plane_off[2] >= (1ull << 31)) return false;
    if ((uint64_t)P.store.dst_pitch * (uint64_t)(P.store.off_y + P.out_h) >= (1ull << 32)) return false;
    return true;
}

// CHROMA_CatmullRom for 4:2:0 (Shaders.cpp:66-72,242-251): the phase t of even / odd luma columns and rows for a chroma siting, and
// catmull_weights(t) with the shader's own expressions in fp32
void ChromaCatmullWeights(int chroma_loc, float wx[2][4], float wy[2][4])
{
    for (int par = 0; par < 2; par++) {
        float tx = par ? 0.75f : 0.25f, ty = par ? 0.75f : 0.25f;
        if (chroma_loc == CLOC_COSITED) { tx += -0.25f; ty += -0.25f; }
        else if (chroma_loc == CLOC_MPEG1) { tx += -0.5f; ty += -0.5f; }
        else { tx += -0.25f; ty += -0.5f; }
        for (int axis = 0; axis < 2; axis++) {
            const float t = axis ? ty : tx;
            float *w = axis ? wy[par] : wx[par];
            const float t2 = t * t, t3 = t * t2;
            w[0] = t2 - (t3 + t) / 2;
            w[1] = t3 * 1.5f + 1 - t2 * 2.5f;
            w[2] = t2 * 2 + t / 2 - t3 * 1.5f;
            w[3] = (t3 - t2) / 2;
        }
    }
}

// the convert-side and store-side constants both kernels of this file take
void FillFusedArgs(const FusedParams &P, FusedArgs &a, int resize_follows)
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
    // 4:2:2 has one siting (Shaders.cpp:319-325: u' = sx/2 + 0.25, v' = sy): the 4:2:0 switches stay off
    a.sub422 = c.fmt.subsampling == 422;
    a.sub444 = c.fmt.subsampling == 444 || c.fmt.layout == LAY_GRAY;
    a.packed422 = c.fmt.layout == LAY_PACKED422;
    a.packed444 = c.fmt.layout != LAY_PACKED444 ? 0 : c.fmt.bits10 ? 2 : c.fmt.bytes == 1 ? 1 : 3;
    a.gray = c.fmt.layout == LAY_GRAY;
    if (a.packed444) a.bytes = a.packed444 == 1 ? 1 : 2;     // (width of the fields the loader packs the luma pair into)
    for (int i = 0; i < 4; i++) a.ci[i] = c.fmt.ci[i];
    a.nearest = c.chroma_scaling == 0 && c.fmt.layout == LAY_PLANAR && (c.fmt.subsampling == 420 || c.fmt.subsampling == 422);
    a.cw_own = a.sub444 ? 0.0f : a.nearest ? 1.0f : 0.5f;
    a.cw_next = a.sub444 ? 1.0f : a.nearest ? 0.0f : 0.5f;
    a.center_h = c.fmt.subsampling == 420 && c.chroma_loc == CLOC_MPEG1 && !a.nearest;      // (no siting without a filter)
    a.v_off4 = (c.fmt.subsampling == 420 && c.chroma_loc == CLOC_COSITED && !a.nearest) ? 1 : 0;
    // chroma_v4 (vp_fused_dev.h): 4 sy (4:2:2 / 4:4:4), 4 (sy >> 1) (nearest), 2 sy - 1 + v_off4 (4:2:0 filtered)
    a.vk1 = (a.sub422 | a.sub444) ? 4 : a.nearest ? 0 : 2;
    a.vk2 = (a.sub422 | a.sub444) ? 0 : a.nearest ? 2 : 0;
    a.vk3 = (a.sub422 | a.sub444 | a.nearest) ? 0 : a.v_off4 - 1;
    // UNORM scale: v/255, or (v << shift)/65535 for planar data; interleaved UV planes carry no shift
    const float sy = c.fmt.bits10 ? 1.0f / 1023.0f : c.fmt.bytes == 1 ? 1.0f / 255.0f : (float)(1 << c.fmt.shift) / 65535.0f;
    const float sc = c.fmt.bits10 ? 1.0f / 1023.0f : c.fmt.bytes == 1 ? 1.0f / 255.0f : (float)(1 << (c.fmt.planes == 2 ? 0 : c.fmt.shift)) / 65535.0f;
    const bool dv = c.dovi != nullptr;       // Dolby Vision: the reshaping curves sit between the texel and the matrix, so the scale stays outside
    for (int i = 0; i < 3; i++) {
        a.m[3 * i + 0] = c.cm[3 * i + 0] * (dv ? 1.0f : sy);
        a.m[3 * i + 1] = c.cm[3 * i + 1] * (dv ? 1.0f : sc);
        a.m[3 * i + 2] = c.cm[3 * i + 2] * (dv ? 1.0f : sc);
        a.c[i] = c.cm[9 + i];
    }
    // the exact form of the convert stage (convert_block_exact): an 8-bit internal format in front of a resize — there a texel one code off
    // the reference's comes out of a negative-lobe filter up to two codes off.  10-bit internal formats keep the fast form (the same effect
    // is a quarter of an 8-bit code), and so do the tails (their own transcendentals decide the last code) and Dolby Vision.
    static const int exact8 = EnvInt("MPCVR_EXACT8", 1);          // 0: the fast form everywhere (A/B)
    // ... and, round 5's last finding (the fuzz tool with the tier flags, seed 208, case 600): 10-bit internal formats too where an HDR10 tone-mapping
    // operator follows — one code of the intermediate came out of operator 5 as five ten-bit codes
    const bool exact_fmt = c.out_fmt == SF_BGRA8 || (P.exact_wide && c.out_fmt == SF_RGB10A2);
    a.exact_cv = (exact8 && ((resize_follows >= 0 ? resize_follows : P.exact_convert) || P.exact_wide) && exact_fmt && c.tail == TAIL_NONE && !dv) ? 1 : 0;
    for (int i = 0; i < 9; i++) a.xm[i] = c.cm[i];
    for (int i = 0; i < 3; i++) a.xc[i] = c.cm[9 + i];
    {
        const float maxc = c.fmt.bits10 ? 1023.0f : c.fmt.bytes == 1 ? 255.0f : 65535.0f;
        const float py = c.fmt.bytes == 2 && !c.fmt.bits10 ? (float)(1 << c.fmt.shift) : 1.0f;
        const float pc = c.fmt.bytes == 2 && !c.fmt.bits10 && c.fmt.planes != 2 ? (float)(1 << c.fmt.shift) : 1.0f;
        a.xdy = maxc / py; a.xry = (1.0f / maxc) * py;            // exact scalings by a power of two
        a.xdc = maxc / pc; a.xrc = (1.0f / maxc) * pc;
    }
    a.dovi = c.dovi; a.eotf_lut = P.eotf_lut; a.sy = sy; a.sc = sc;
    a.dovi_cm = dv ? P.dovi_cm : nullptr; a.dovi_per_frame = a.dovi_cm ? 1 : 0;
    ChromaCatmullWeights(c.chroma_loc, a.crx, a.cry);
    a.tail = c.tail; a.gamma = c.gamma; a.lum_scale = c.lum_scale;
    std::memcpy(a.gamut, c.gamut, sizeof(a.gamut));
    a.lut = c.tail == TAIL_HLG_TO_SDR ? P.hlg_lut : P.pq_lut;
    a.maxv = c.out_fmt == SF_RGB10A2 ? 1023.0f : 255.0f;
    a.inv_maxv = 1.0f / a.maxv;
    a.q_over_maxv = (float)P.store.quant / a.maxv;
    a.epi_mul = FinalPassMultiplier(P.store.quant, (int)a.maxv);
    a.dst_pitch = P.store.dst_pitch; a.off_x = P.store.off_x; a.off_y = P.store.off_y;
    a.final_pass = P.store.mode == ST_FINAL; a.out10 = P.store.dst_fmt == SF_RGB10A2;
    a.quant = (float)P.store.quant;
    a.dither = P.store.dither;
}

int FusedTailKind(const FusedParams &P)
{
    const ConvertParams &c = P.conv;
    // P.pq_lut is null when MPCVR_FLAG_NO_LUT asks for the literal ALU chains (A/B testing)
    return c.tail == TAIL_NONE ? TAILK_NONE : (c.tail == TAIL_PQ_TO_SDR && P.pq_lut) ? TAILK_PQ_LUT
         : (c.tail == TAIL_HLG_TO_SDR && !P.literal_tail && P.hlg_lut) ? TAILK_HLG : TAILK_ALU;
}
int FusedDoviKind(const FusedParams &P)
{
    if (!P.conv.dovi) return DV_NONE;
    if (P.conv.tail != TAIL_PQ_TO_SDR || P.literal_tail) return DV_GENERAL;
    return !P.dovi_l2 ? DV_SDR : P.pq_lut ? DV_SDR_L2 : DV_GENERAL;
}
int FusedSourceKind(const FusedParams &P)
{
    // source specialisations: bi-planar 16-bit (P010/P016) and bi-planar 8-bit (NV12) with MPEG-2 / co-sited chroma;
    // everything else (planar, MPEG-1 siting) runs through the variant that reads these properties at run time
    const ConvertParams &c = P.conv;
    const bool centred = c.fmt.subsampling == 420 && c.chroma_loc == CLOC_MPEG1 && c.chroma_scaling != 0;
    // 8-bit samples behind a PQ / HLG / BT.2020 tail (streams that hardly exist) read through the run-time variant as well: the
    // (8-bit loader, tail) products of every fused family are not built
    if (c.fmt.bytes == 1 && c.tail != TAIL_NONE) return SRC_GENERIC;
    // 16-bit samples behind a forced 8-bit internal format (TEXFMT_8INT on a P010 / YUV420P10 stream: hardly ever): the exact form of the convert
    // stage lives in the 8-bit loaders and in the run-time variant only (exact_capable, vp_fused_dev.h) — the 16-bit loaders keep their registers
    if (c.fmt.bytes == 2 && (c.out_fmt == SF_BGRA8 || (P.exact_wide && c.out_fmt == SF_RGB10A2)) && c.tail == TAIL_NONE && !c.dovi) return SRC_GENERIC;
    const bool biplanar_fast = c.fmt.planes == 2 && !centred, planar_fast = c.fmt.planes == 3 && !centred;
    return (biplanar_fast && c.fmt.bytes == 2) ? SRC_P01X : (biplanar_fast && c.fmt.bytes == 1) ? SRC_NV12
         : (planar_fast && c.fmt.bytes == 2) ? SRC_PLANAR16 : (planar_fast && c.fmt.bytes == 1) ? SRC_PLANAR8 : SRC_GENERIC;
}

bool ConvertBlocksSupported(const FusedParams &P, bool to_rt)
{
    const ConvertParams &c = P.conv;
    if (c.out_fmt != SF_BGRA8 && c.out_fmt != SF_RGB10A2) return false;
    if (!BlockConvertLayout(P, true) || c.blend_deint) return false;
    if (c.dovi && !P.eotf_lut) return false;       // the Dolby Vision variant decodes PQ from a table (MPCVR_FLAG_NO_LUT: per-pixel kernel)
    if (c.chroma_scaling == 2 && c.dovi) return false;     // Catmull-Rom chroma: no Dolby Vision variant instantiated
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

// compute units of the current device, and the workgroups of a kernel one of them keeps resident (queried once per kernel, block
// size, LDS size and device): k_convert_stream launches exactly that many
int DeviceCuCount()
{
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return cus[dev] = n;
}
static int ResidentWorkgroups(const void *kern, int threads, size_t lds)
{
    static std::mutex mu;
    static std::map<std::pair<std::pair<const void *, int>, std::pair<int, size_t>>, int> memo;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_pair(std::make_pair(kern, dev), std::make_pair(threads, lds));
    std::lock_guard<std::mutex> lock(mu);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, lds) != hipSuccess || n <= 0) n = 2;
    return memo[key] = n;
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
    } else if (!frames_dev && n_frames != 1 && !(frames_host && n_frames <= kHostTableMax)) return hipErrorInvalidValue;
    FusedArgs a;
    FillFusedArgs(P, a);
    const ConvertParams &c = P.conv;
    const bool fin = P.store.mode == ST_FINAL;
    const int tailk = FusedTailKind(P), srck = FusedSourceKind(P);
    const int dvk = FusedDoviKind(P);
    const bool catmull = c.chroma_scaling == 2 && c.fmt.subsampling == 420;
    // the streaming kernel (k_convert_stream): bi-planar 4:2:0 / 4:2:2 samples whose rows take the lane's 4- / 8-byte loads and 16-byte stores
    static const int no_stream = EnvInt("MPCVR_NO_STREAM_CONVERT", 0);
    const int lbs = srck == SRC_P01X ? 8 : 4;
    // (the exact form of the convert stage — an 8-bit texture in front of an UNFUSED resize — is k_convert_blocks' alone: the streaming kernel is
    // C1's, and the second form behind a branch cost it 8 registers and 15 % with the branch never taken)
    const bool stream = !no_stream && !catmull && dvk == DV_NONE && !a.exact_cv && (srck == SRC_P01X || srck == SRC_NV12) && c.out_w >= 8 && (c.out_w & 3) == 0 && (c.rect_l & 3) == 0 &&
                        (c.pitch[0] % lbs) == 0 && (c.pitch[1] % lbs) == 0 && (P.plane_off[1] % lbs) == 0 && P.dst_aligned16 && (P.store.off_x & 3) == 0 &&
                        (P.store.dst_pitch & 15) == 0 && P.src_aligned16 && (frames_dev || frames_host || n_frames == 1);
    if (!stream && !frames_dev && n_frames != 1 && !tab.n) return hipErrorInvalidValue;     // (a host table of more than 32 frames serves the streaming kernel only)
    if (stream) {
        FrameTable128 tab128;
        tab128.n = 0;
        if (frames_host && n_frames <= kHostTableMax) {
            tab128.n = n_frames;
            for (int i = 0; i < n_frames; i++) tab128.f[i] = frames_host[i];
            for (int i = n_frames; i < kHostTableMax; i++) tab128.f[i] = FusedFrame{nullptr, nullptr};
        }
        const bool tables = fin || tail_has_table(tailk);
        const size_t lds = (fin ? 4096 : 0) + (tail_has_table(tailk) ? LDS_T : 0);
        StreamArgs q{};
        q.n_strips = (c.out_w + 255) / 256;
        q.npairs = c.out_h / 2 + 1;
        q.n_frames = n_frames;
        q.total = n_frames * q.npairs;
        q.batch_dst = batch_dst; q.batch_stride = batch_stride;
        // workgroups of 8 waves where tables are staged (once per workgroup), of 4 otherwise; exactly the waves the chip keeps resident
        static const int wg_env = EnvInt("MPCVR_STREAM_WG_WAVES", 0), occ_env = EnvInt("MPCVR_STREAM_WG_PER_CU", 0), min_pairs_env = EnvInt("MPCVR_STREAM_MIN_PAIRS", 0);
        const int wgw = wg_env >= 1 && wg_env <= 8 ? wg_env : tables ? 8 : 4;
        const dim3 block(64 * wgw, 1, 1);
        hipError_t err = hipSuccess;
        auto launch = [&](auto kern) {
            const int per_cu = occ_env > 0 ? occ_env : ResidentWorkgroups((const void *)kern, 64 * wgw, lds);
            const long n_waves = (long)std::max(per_cu, 1) * DeviceCuCount() * wgw;
            const int min_pairs = min_pairs_env > 0 ? min_pairs_env : 4;            // a wave's run: at least this many row pairs (a short launch spreads thinner, not shorter)
            q.n_slots = (int)std::max<long>(1, std::min<long>(n_waves / q.n_strips, std::max(1, q.total / min_pairs)));
            const dim3 grid((unsigned)(((long)q.n_slots * q.n_strips + wgw - 1) / wgw), 1, 1);
            hipLaunchKernelGGL(kern, grid, block, lds, s, a, frames_dev, single, q, tab128);
            err = hipGetLastError();
        };
#define MPCVR_ST3(TK, SK, FN) launch(k_convert_stream<TK, SK, FN>)
#define MPCVR_ST2(TK, SK) do { if (fin) MPCVR_ST3(TK, SK, true); else MPCVR_ST3(TK, SK, false); } while (0)
        // (NV12 reaches this point without a tail only — FusedSourceKind — so the tails exist for the 16-bit loader)
        if (srck == SRC_NV12) MPCVR_ST2(TAILK_NONE, SRC_NV12);
        else if (tailk == TAILK_NONE) MPCVR_ST2(TAILK_NONE, SRC_P01X);
        else if (tailk == TAILK_PQ_LUT) MPCVR_ST2(TAILK_PQ_LUT, SRC_P01X);
        else if (tailk == TAILK_HLG) MPCVR_ST2(TAILK_HLG, SRC_P01X);
        else MPCVR_ST2(TAILK_ALU, SRC_P01X);
#undef MPCVR_ST2
#undef MPCVR_ST3
        return err;
    }
    const int strips = (c.out_w + 127) / 128, npairs = c.out_h / 2 + 1;
    // row pairs per wave: enough waves to fill the chip a few times over, few enough to amortise the table staging
    int pairs = 16;
    static const int pairs_waves = EnvInt("MPCVR_CB_WAVES", 0);
    // kernels that stage tables in LDS (dither, tone map) amortise that over long waves; without tables short waves win: a wave
    // keeps one row pair of loads in flight, so the number of resident waves is the memory-level parallelism (C1: 244 k frames/s
    // with 4 k waves per launch, 284 k with 64 k)
    const bool tables = fin || tail_has_table(tailk) || dvk != DV_NONE;
    const long want_waves = pairs_waves > 0 ? pairs_waves : tables ? 8192 : 65536;
    while (pairs > 2 && (long)strips * ((npairs + pairs - 1) / pairs) * n_frames < want_waves) pairs >>= 1;
    const int wg_waves = dvk == DV_SDR_L2 ? 8 : 4;         // (k_convert_blocks: NTH)
    const dim3 grid(strips, (npairs + wg_waves * pairs - 1) / (wg_waves * pairs), n_frames), block(64 * wg_waves, 1, 1);
    const size_t lds = (fin ? 4096 : 0) + (dvk != DV_NONE ? LDS_E + LDS_V + (dvk == DV_SDR_L2 ? LDS_T + LDS_PE : 0) : tail_has_table(tailk) ? LDS_T : 0);
    if (dvk != DV_NONE) {       // Dolby Vision: 16-bit bi-planar (P010 / P016) or whatever the generic source variant reads
#define MPCVR_CBD(SK, FN, DK) hipLaunchKernelGGL((k_convert_blocks<TAILK_ALU, SK, FN, DK>), grid, block, lds, s, a, frames_dev, single, pairs, batch_dst, batch_stride, tab)
#define MPCVR_CBD2(SK, FN) do { if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_convert_blocks<TAILK_ALU, SK, FN, DV_SDR_L2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                                if (dvk == DV_SDR) MPCVR_CBD(SK, FN, DV_SDR); else if (dvk == DV_SDR_L2) MPCVR_CBD(SK, FN, DV_SDR_L2); else MPCVR_CBD(SK, FN, DV_GENERAL); } while (0)
        if (srck == SRC_P01X) { if (fin) MPCVR_CBD2(SRC_P01X, true); else MPCVR_CBD2(SRC_P01X, false); }
        else { if (fin) MPCVR_CBD2(SRC_GENERIC, true); else MPCVR_CBD2(SRC_GENERIC, false); }
#undef MPCVR_CBD2
#undef MPCVR_CBD
        return hipGetLastError();
    }
#define MPCVR_CB3(TK, SK, FN) do { if (catmull) hipLaunchKernelGGL((convert_blocks_kernel<TK, SK, FN, 1>(a.exact_cv != 0)), grid, block, lds, s, a, frames_dev, single, pairs, batch_dst, batch_stride, tab); \
                                   else hipLaunchKernelGGL((convert_blocks_kernel<TK, SK, FN, 0>(a.exact_cv != 0)), grid, block, lds, s, a, frames_dev, single, pairs, batch_dst, batch_stride, tab); } while (0)
#define MPCVR_CB2(TK, SK) do { if (fin) MPCVR_CB3(TK, SK, true); else MPCVR_CB3(TK, SK, false); } while (0)
#define MPCVR_CB(TK) do { if (srck == SRC_P01X) MPCVR_CB2(TK, SRC_P01X); else if (srck == SRC_PLANAR16) MPCVR_CB2(TK, SRC_PLANAR16); else MPCVR_CB2(TK, SRC_GENERIC); } while (0)
    if (tailk == TAILK_NONE) {          // (the 8-bit loaders exist without a tail only: FusedSourceKind)
        if (srck == SRC_NV12) MPCVR_CB2(TAILK_NONE, SRC_NV12); else if (srck == SRC_PLANAR8) MPCVR_CB2(TAILK_NONE, SRC_PLANAR8); else MPCVR_CB(TAILK_NONE);
    }
    else if (tailk == TAILK_PQ_LUT) MPCVR_CB(TAILK_PQ_LUT);
    else if (tailk == TAILK_HLG) MPCVR_CB(TAILK_HLG);
    else MPCVR_CB(TAILK_ALU);
#undef MPCVR_CB
#undef MPCVR_CB2
#undef MPCVR_CB3
    return hipGetLastError();
}

#ifndef MPCVR_UP2X_WAVES_HOST
#define MPCVR_UP2X_WAVES_HOST 3     // = MPCVR_UP2X_WAVES of vp_fused_up2x.h: waves per SIMD the fused 2x kernel is allocated for
#endif
hipError_t LaunchFusedUp2x(const FusedParams &P, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s)
{
    static const int seg_env = EnvInt("MPCVR_FUSED_SEG", 0);
    if (!frames_dev && n_frames != 1) return hipErrorInvalidValue;
    const ConvertParams &c = P.conv;
    FusedArgs a;
    FillFusedArgs(P, a, 1);
    if (P.jinc_tab) return LaunchFusedJinc2x(P, a, P.jinc_tab, frames_dev, single, n_frames, s);
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
        // frames that run side by side (a batch, or single frames on the context's lanes) share the chip: a batch aims at >= 4 rounds of
        // the ~3072 resident waves, overlapping single frames at a round and a quarter between them — the fewest, longest segments that still fill it
        const long side = (long)n_frames * (P.inflight > 1 ? P.inflight : 1);
        // (single frames on four lanes, one box: 72-row segments 19.9 k frames/s, 60: 19.7 k, 90: 19.2 k, 120: 17.8 k, 180: 14.2 k — 3,840 waves)
        const long want = n_frames > 1 ? 12288 : 3840;
        for (int cand : {180, 144, 120, 108, 90, 72, 60, 48, 36, 24})
            if ((long)strips * ((c.out_h + cand - 1) / cand) * side >= want || cand == 24) { seg = cand; break; }
        if (n_frames > 1) {
            // A batch starts together and runs in rounds of the resident waves (3 per SIMD): a half-empty last round costs a whole one, and every
            // segment walks 6 rows of run-in.  Cost of a candidate = rounds x (rows + 6); the longest segment among the cheapest.  32 frames of
            // 1080p: 180 rows = 3,072 items = ONE round, 96.2 k frames/s (C2) where the rule above took 36 rows (5 rounds, 17 % run-in): 87.9 k.
            // (Single frames on the context's lanes do not start together: the measured table above stands for them.)
            const long resident = (long)DeviceCuCount() * 4 * MPCVR_UP2X_WAVES_HOST;
            long best = -1;
            for (int cand : {180, 144, 120, 108, 90, 72, 60, 48, 36, 24}) {
                const long items = (long)strips * ((c.out_h + cand - 1) / cand) * side;
                const long cost = ((items + resident - 1) / resident) * (long)(std::min(cand, c.out_h) + 6);
                if (best < 0 || cost < best) { best = cost; seg = cand; }
            }
        }
    }
    seg = (seg + 1) & ~1;
    if (seg > c.out_h) seg = c.out_h;
    a.seg_rows = seg;

    // (the matrix-core variant of the taps — k_fused_up2x_mx, MPCVR_FLAG_FUSED_MFMA — left the build in round 6: parity-green, never faster
    // than the packed-fp32 chains on this part (profiles/r03/mfma_overlap_ubench.txt: the matrix pipe does not overlap v_pk_fma_f32), 16
    // instantiations and half a CPU-minute of build; its source is kept under profiles/r06/experiments/matrix_core_taps/.  The flag is accepted
    // and ignored.)
    if (knt == 4) return LaunchFusedUp2xNT<4>(P, a, strips, seg, frames_dev, single, n_frames, s);
    if (knt == 5) return LaunchFusedUp2xNT<5>(P, a, strips, seg, frames_dev, single, n_frames, s);
    return LaunchFusedUp2xNT<6>(P, a, strips, seg, frames_dev, single, n_frames, s);
}

}  // namespace mpcvr
