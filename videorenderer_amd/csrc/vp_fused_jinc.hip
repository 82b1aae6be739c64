// vp_fused_jinc.hip — Jinc2m (Shaders/examples/ps_resize_onepass_jinc2.hlsl:44-101) at exactly 2x IN ONE KERNEL with the convert draw
// in front of it and the final pass behind it: what the reference does as ConvertColorPass -> m_TexConvertOutput -> one 2-D resize draw ->
// m_TexsPostScale -> FinalPass (DX11VideoProcessor.cpp:3285-3424, :2921: m_pShaderUpscaleY = m_pShaderUpscaleX) without an intermediate in
// HBM.  Until round 5 this ratio ran as k_convert_* + k_jinc2_quad (vp_jinc.hip): the convert output written and read back, 141 VALU
// instructions per output pixel, a third of them address arithmetic of a 128 x 8 tile.
//
// Design — the fused 2x kernel's (vp_fused_up2x.h) with a ring of converted rows in LDS where that kernel has a register window of X-pass
// results (the 2-D filter is not separable: there is no X pass to park):
//   one wavefront owns a strip of S = 120 source columns (4-column halo each side; the filter needs 2) and marches down a segment two
//   source rows per iteration, no workgroup barrier in the loop:
//     stage C  (= the fused 2x kernel's) lane j converts the 2x2 block {cols 2j, 2j+1} x {rows a, a+1} (a odd: the two rows lie between the
//              same two chroma rows), rounds it to the internal UNORM format and writes it as 0..1 floats into slot t mod 3 of the wave's
//              ring: R[slot][channel][row][128 columns];
//     stage J  lane l owns source columns 2l, 2l+1 and, this iteration, source rows k = a-2, a-1: four 2x2 output quads.  The two quads of
//              a source row sit one column apart and see the SAME four phases, so a packed FMA serves both: its operand is the pair of
//              horizontally adjacent texels (v[m], v[m+1]), m = 0..4 — the even pairs are aligned 8-byte reads of the ring row, the odd
//              ones the same row read again 4 bytes on (LDS has the room, the VALU does not: no register shuffles).  Per source row of
//              the 5 x 6 neighbourhood: 15 pairs, 96 v_pk_fma_f32 with the weight broadcast from one half of an SGPR pair;
//              anti-ringing min / max of the inner 2x2 as v_min3 / v_max3; normalise, pull 80 % towards the clamp (packed, the last FMA
//              saturates), UNORM round, dither in integers, one 16-byte store per lane and output row.
//   Weights: 4 phases x 16, re-ordered on the host into the order stage J meets them (source row of the neighbourhood x output row parity
//   x column parity x 4 taps) and read with scalar loads inside the loop: 16 SGPRs at a time instead of 64 for the whole table.
//   LDS: 9 KiB of ring per wave (three row pairs: stage C of the next pair runs BEHIND stage J), 12 waves per workgroup, + the dither tables
//   and the tone-map table: 146 KiB -> 3 waves per SIMD.  The slot of a row pair is a run-time offset (t mod 3 has no power-of-two unroll).
// Arithmetic identical to k_jinc2_quad's (same products, same order: row by row, left to right, FMAs) on the same converted texels.
#include "vp_fused_dev.h"

namespace mpcvr {

namespace {

#ifndef MPCVR_JINC_WAVES
#define MPCVR_JINC_WAVES 12
#endif
#ifndef MPCVR_JINC_SLOTS
#define MPCVR_JINC_SLOTS 3
#endif
constexpr int JWAVES = MPCVR_JINC_WAVES;           // strips per workgroup
constexpr int JSLOTS = MPCVR_JINC_SLOTS;           // row pairs in the ring: the three stage J reads (+ one being written meanwhile: 4 slots, 8 waves —
                                                   // the first version, 2 waves per SIMD; 3 slots and stage C behind stage J: 12 waves, 3 per SIMD)
constexpr int JSLOT_FLOATS = 3 * 2 * AW;           // [channel][row][column]
constexpr int JRING_FLOATS = JSLOTS * JSLOT_FLOATS;
// waves per workgroup by tail kind.  Without a tail there is no tone-map table in LDS and (with the anti-ringing state kept as texel pairs) the
// kernel fits 128 VGPRs: 16 waves = 4 per SIMD are possible — built and measured on one box (1080p NV12 -> 4K, bench.py --workload
// jinc1080_nv12): 47.0 k frames/s against 47.9 k with 12 waves.  The kernel issues VALU instructions three quarters of the time with three
// waves per SIMD; a fourth adds 16-wave workgroup granularity and nothing else.  The knob stays (-DMPCVR_JINC_WAVES_NOTAIL=16).
#ifndef MPCVR_JINC_WAVES_NOTAIL
#define MPCVR_JINC_WAVES_NOTAIL 12
#endif
__host__ __device__ constexpr int jinc_waves(int tail) { return tail == TAILK_NONE ? MPCVR_JINC_WAVES_NOTAIL : JWAVES; }
__host__ __device__ constexpr int jinc_ring_bytes(int tail) { return jinc_waves(tail) * JRING_FLOATS * 4; }
// the host's weight table (FusedJincTable): [source row sr of the 5-row neighbourhood][output row parity rp][column parity cp][tap i] =
// w[rp][cp][(sr - rp) * 4 + i], zero where sr - rp is no tap row; then 1 / wsum per phase [rp][cp]
constexpr int JTAB_W = 5 * 16, JTAB_FLOATS = JTAB_W + 4;

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));      // a pair at any float of a ring row
typedef const __attribute__((address_space(4))) float *jcptr;           // the constant address space: scalar loads
// min / max of four texel values: v_min3 + v_min spelt out — through fminf the compiler canonicalises every operand first (v_max x, x: the
// values come out of LDS and could be signalling NaNs for all it knows), 4.5 instructions per output pixel for nothing
__device__ __forceinline__ float jmin4(float a, float b, float c, float d)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3\n\tv_min_f32 %0, %0, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
    return r;
}
__device__ __forceinline__ float jmax4(float a, float b, float c, float d)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
    return r;
}
__device__ __forceinline__ void jinc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int TAIL, int SRC, int EPI, int XC>
__device__ __forceinline__ void fused_jinc2x_body(const FusedArgs &P, const float *__restrict__ jtab_g, const FusedFrame *__restrict__ frames, const FusedFrame &single)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Rall = (float *)smem;
    constexpr int JW = jinc_waves(TAIL), LDS_JRING = jinc_ring_bytes(TAIL);
    unsigned short *D = (unsigned short *)(smem + LDS_JRING);
    uint32_t *Di = (uint32_t *)(smem + LDS_JRING + LDS_D);
    f2 *T = (f2 *)(smem + LDS_JRING + LDS_D + LDS_DB);

    for (int i = threadIdx.x; i < 1024; i += 64 * JW) {
        const unsigned short d = P.dither[i];
        D[i] = d;
        Di[i] = (uint32_t)(__half2float(__ushort_as_half(d)) * 1024.0f + 0.5f) << 14;     // d = j/1024 exactly (dither32x32float16.bin)
    }
    if (tail_has_table(TAIL))
        for (int i = threadIdx.x; i < LUT_N; i += 64 * JW) {
            const float v = P.lut[i], n = P.lut[min(i + 1, LUT_N - 1)];
            T[i] = f2{v, n - v};
        }
    __syncthreads();                                   // the only workgroup barrier: tables visible

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int W = P.W, H = P.H;
    // work items = (strip, segment) pairs, strips fastest, dealt to the waves of the workgroups in order: a frame's 16 strips need not be a
    // multiple of the workgroup's waves (12 waves by whole rows of strips: the second workgroup of a row two thirds empty — 28.7 k instead of 37 k frames/s)
    const int n_strips = (W + S - 1) / S;
    const int item = blockIdx.x * JW + wave;
    const int x0 = (item % n_strips) * S;
    const int s0 = (item / n_strips) * P.seg_rows;
    if (s0 >= H) return;
    const int s1 = min(s0 + P.seg_rows, H);
    float *R = Rall + wave * JRING_FLOATS;

    const FusedFrame frame = frames ? frames[blockIdx.z] : single;
    auto uniform_ptr = [](const void *q) {
        const uint64_t v = (uint64_t)q;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const uint64_t src_u = uniform_ptr(frame.src), dst_u = uniform_ptr(frame.dst);
    const gcptr py = (gcptr)src_u;
    const gptr pdst = (gptr)dst_u;

    // stage C role: ring columns 2*lane, 2*lane+1 = rect columns X, X+1; the block is fetched at Xg (inside the rect)
    const int X = x0 - 4 + 2 * lane;
    const int Xg = clampi(X, 0, W - 2);
    const bool edge_wave = x0 == 0 || x0 + 2 * 63 - 4 > W - 2;      // wave-uniform: some lane's block hangs over the rect
    // stage J role: source columns x0 + 2l, x0 + 2l + 1 -> output columns ox .. ox+3 (rect-relative); lanes 60..63 idle there
    const bool j_active = lane < 60;
    const int ox = 2 * x0 + 4 * lane;
    const bool store_ok = j_active && ox < 2 * W;
    const int wx0 = P.off_x + ox;
    const bool d_aligned = (wx0 & 3) == 0;
    const bool st_aligned = d_aligned && (((uintptr_t)dst_u | (uintptr_t)P.dst_pitch) & 15) == 0;
    const uint32_t lane_off = (uint32_t)wx0 * 4u;

    const f2 MM[5] = {f2{P.m[0], P.m[1]}, f2{P.m[2], P.m[3]}, f2{P.m[4], P.m[5]}, f2{P.m[6], P.m[7]}, f2{P.m[8], 0.0f}};
    const f2 GG[5] = {f2{P.gamut[0], P.gamut[1]}, f2{P.gamut[2], P.gamut[3]}, f2{P.gamut[4], P.gamut[5]}, f2{P.gamut[6], P.gamut[7]}, f2{P.gamut[8], 0.0f}};
    constexpr bool FASTEPI = EPI == EPI_DITHER8;
    const f2 maxv2 = splat((FASTEPI || P.final_pass) ? P.maxv : P.quant);
    const f2 cmax2 = splat(P.maxv), cinv2 = splat(P.inv_maxv);
    f2 big2 = splat(8388608.0f);
    asm volatile("" : "+v"(big2));
    f2 CC[3] = {splat(P.c[0]), splat(P.c[1]), splat(P.c[2])};
    asm volatile("" : "+v"(CC[0]), "+v"(CC[1]), "+v"(CC[2]));
    const f2 k08 = f2{0.8f, 0.0f};                    // the anti-ringing strength (ps_resize_onepass_jinc2.hlsl: AR_STRENGTH)

    // iteration t converts virtual rows a, a+1 with a = s0 - 3 + 2t into slot t mod 3 — an ODD row first: rows 2m-1, 2m lie between the same two
    // chroma rows (load_raw fetches the pair's chroma once), rows 2m, 2m+1 do not — and from t = 2 on emits the output rows of k = a-2, a-1
    // (the first of them belongs to the segment above in the first emitting iteration, the second to the one below in the last)
    const int n_iter = (s1 - s0 + 1) / 2 + 3;
    Raw raw2[2];
    RawAddr ra;
    make_raw_addr<SRC>(P, Xg, ra);
    load_raw<SRC>(P, py, ra, clampi(s0 - 3, 0, H - 1), clampi(s0 - 2, 0, H - 1), raw2[0]);
    load_raw<SRC>(P, py, ra, clampi(s0 - 1, 0, H - 1), clampi(s0, 0, H - 1), raw2[1]);

    // stage C for virtual rows ar, ar+1 (raw codes in buffer b) into ring slot `slot`; prefetches rows ar+4, ar+5
    auto stage_c = [&](int ar, int b, int slot_off) {
        f2 rc[2][3];
        convert_block<TAIL, SRC, DV_NONE, XC, XC == XC_ALWAYS ? OUT_CODE_F : OUT_NORM>(P, MM, GG, CC, raw2[b], P.rect_t + clampi(ar, 0, H - 1), P.rect_t + clampi(ar + 1, 0, H - 1), T, rc);
        load_raw<SRC>(P, py, ra, clampi(ar + 4, 0, H - 1), clampi(ar + 5, 0, H - 1), raw2[b]);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // store to m_TexConvertOutput (UNORM: floor(sat(x)*maxv + 0.5)) and read back (q/maxv to 1 ulp); the exact form hands over the codes
            const f2 qe = (XC == XC_ALWAYS ? rc[0][c] : unorm_round2(rc[0][c], cmax2, big2)) * cinv2;             // even column, rows (a, a+1)
            const f2 qo = (XC == XC_ALWAYS ? rc[1][c] : unorm_round2(rc[1][c], cmax2, big2)) * cinv2;             // odd column
            float *r0 = R + slot_off + (c * 2 + 0) * AW + 2 * lane, *r1 = r0 + AW;
            *(f2 *)r0 = f2{qe.x, qo.x};
            *(f2 *)r1 = f2{qe.y, qo.y};
            if (edge_wave) {       // clamp-to-edge of the convert texture: patch the column that hangs over (rare wave)
                if (X < 0) { r0[1] = qe.x; r1[1] = qe.y; }
                else if (X > W - 2) { r0[0] = qo.x; r1[0] = qo.y; }
            }
        }
    };
    stage_c(s0 - 3, 0, 0);
    int slot_t = 0;                                   // t mod JSLOTS (wave-uniform)
    auto slot_back = [&](int d) { const int v = slot_t - d; return (v < 0 ? v + JSLOTS : v) * JSLOT_FLOATS; };      // float offset of the pair of iteration t - d

    jcptr jtab = (jcptr)(uintptr_t)jtab_g;
    for (int tb = 0; tb < n_iter; tb += 2) {
        // (keeps the weight loads inside the loop: hoisted, the 84 values would sit in SGPRs — and spill into VGPR lanes — for its whole length)
        asm volatile("" : "+s"(jtab));
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int t = tb + u;
            if (t >= n_iter) break;
            const int a = s0 - 3 + 2 * t;
            jinc_wave_sync();
            const int next_off = (slot_t + 1 == JSLOTS ? 0 : slot_t + 1) * JSLOT_FLOATS;
            // ---------------- stage C of the NEXT iteration: in front of stage J where its slot is not among the three stage J reads ----------------
            if (JSLOTS >= 4 && t + 1 < n_iter) stage_c(a + 2, (u + 1) & 1, next_off);
            const float *rb[3] = {R + slot_back(2) + 2 * lane + 2, R + slot_back(1) + 2 * lane + 2, R + slot_back(0) + 2 * lane + 2};      // ring column of source column 2l - 2
            if (t >= 2 && store_ok) {
            // ---------------- stage J + final pass ----------------
#pragma unroll
            for (int qr = 0; qr < 2; qr++) {
                const int k = a - 2 + qr;                       // source row -> output rows 2k, 2k+1
                if (k < s0 || k >= s1) continue;                // (wave-uniform: the neighbouring segments' rows)
                f2 acc[2][2][3];                                // [row parity][column parity][channel] = (left quad, right quad)
                f2 in1[3][3], in2[3][3];                        // the inner pairs (m = 1 .. 3) of the previous source row and of the one before: what the
                                                                // anti-ringing clamp of the row being finished reads (36 registers; its min / max are formed
                                                                // where they are used — per pixel and held from tap row 2 on they were 48 + a full previous row)
#pragma unroll
                for (int sr = 0; sr < 5; sr++) {
                    const int ri = qr + sr;                     // row of the six this iteration's quads read: rows a-4 .. a+1
                    // the 16 weights of this neighbourhood row: [rp][cp][i], two to an SGPR pair
                    f2 wq[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) wq[i] = f2{jtab[sr * 16 + 2 * i], jtab[sr * 16 + 2 * i + 1]};
                    f2 pr[3][5];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float *row = rb[ri >> 1] + (c * 2 + (ri & 1)) * AW;
                        pr[c][0] = *(const f2 *)(row); pr[c][2] = *(const f2 *)(row + 2); pr[c][4] = *(const f2 *)(row + 4);
                        pr[c][1] = *(const f2u *)(row + 1); pr[c][3] = *(const f2u *)(row + 3);
                    }
#pragma unroll
                    for (int rp = 0; rp < 2; rp++) {
                        const int j = sr - rp;                  // this source row is tap row j of the output row with parity rp
                        if (j < 0 || j > 3) continue;
#pragma unroll
                        for (int i = 0; i < 4; i++)
#pragma unroll
                            for (int cp = 0; cp < 2; cp++)
#pragma unroll
                                for (int c = 0; c < 3; c++) {
                                    const f2 w = wq[rp * 4 + cp * 2 + (i >> 1)];
                                    const f2 x = pr[c][cp + i];
                                    if (j == 0 && i == 0) acc[rp][cp][c] = (i & 1) ? pk_mul_w<1>(w, x) : pk_mul_w<0>(w, x);
                                    else acc[rp][cp][c] = (i & 1) ? pk_fma_w<1, false>(w, x, acc[rp][cp][c]) : pk_fma_w<0, false>(w, x, acc[rp][cp][c]);
                                }
                        if (j != 3) continue;
                        // ---- the output row 2k + rp is complete: normalise, anti-ringing, final pass, store ----
                        const f2 iw = f2{jtab[JTAB_W + 2 * rp], jtab[JTAB_W + 2 * rp + 1]};          // 1 / wsum of (rp, cp = 0), (rp, cp = 1)
                        f2 fin[3][2];                           // [channel][column parity] = (left quad's pixel, right quad's): px = 2 * half + cp
#pragma unroll
                        for (int c = 0; c < 3; c++)
#pragma unroll
                            for (int cp = 0; cp < 2; cp++) {
                                const f2 v = cp ? pk_mul_w<1>(iw, acc[rp][cp][c]) : pk_mul_w<0>(iw, acc[rp][cp][c]);
                                // clamp(v, mn, mx) with mn <= mx is the median of the three
                                // inner 2x2 = tap rows 1, 2 (the two source rows in front of this one) x tap columns 1, 2 (pairs cp + 1, cp + 2)
                                const f2 p1 = in2[c][cp], p2 = in2[c][cp + 1], q1 = in1[c][cp], q2 = in1[c][cp + 1];
                                const f2 lo = f2{jmin4(p1.x, p2.x, q1.x, q2.x), jmin4(p1.y, p2.y, q1.y, q2.y)}, hi = f2{jmax4(p1.x, p2.x, q1.x, q2.x), jmax4(p1.y, p2.y, q1.y, q2.y)};
                                const f2 cl = f2{__builtin_amdgcn_fmed3f(v.x, lo.x, hi.x), __builtin_amdgcn_fmed3f(v.y, lo.y, hi.y)};
                                fin[c][cp] = pk_fma_w<0, true>(k08, cl - v, v);             // lerp(v, cl, 0.8), saturated (every store below clamps to 0..1)
                            }
                        const int wy = P.off_y + 2 * k + rp;
                        uint32_t pk[4];
                        if (FASTEPI) {
                            // m_TexsPostScale store / load + ps_final_pass.hlsl:29 in integers: see vp_fused_up2x.h
                            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                            const u32x4 dd = *(const u32x4 *)(Di + (wy & 31) * 32 + (wx0 & 31));
                            const uint32_t dj[4] = {dd.x, dd.y, dd.z, dd.w};
                            f2 uq[3][2];
#pragma unroll
                            for (int c = 0; c < 3; c++)
#pragma unroll
                                for (int cp = 0; cp < 2; cp++) uq[c][cp] = pk_fma(fin[c][cp], maxv2, big2);
#pragma unroll
                            for (int px = 0; px < 4; px++) {
                                const uint32_t ib = __umul24(__float_as_uint(uq[2][px & 1][px >> 1]), P.epi_mul) + dj[px];
                                const uint32_t ig = __umul24(__float_as_uint(uq[1][px & 1][px >> 1]), P.epi_mul) + dj[px];
                                const uint32_t ir = __umul24(__float_as_uint(uq[0][px & 1][px >> 1]), P.epi_mul) + dj[px];
                                const uint32_t bg = __builtin_amdgcn_perm(ig, ib, 0x0c0c0703u);    // [B, G, 0, 0]
                                pk[px] = __builtin_amdgcn_perm(ir, bg, 0x0d070100u);               // [B, G, R, 0xff]
                            }
                        } else if (EPI == EPI_DIRECT8) {
                            f2 uq[3][2];
#pragma unroll
                            for (int c = 0; c < 3; c++)
#pragma unroll
                                for (int cp = 0; cp < 2; cp++) uq[c][cp] = pk_fma(fin[c][cp], maxv2, big2);
#pragma unroll
                            for (int px = 0; px < 4; px++) {
                                const uint32_t cr = __float_as_uint(uq[0][px & 1][px >> 1]), cg = __float_as_uint(uq[1][px & 1][px >> 1]), cb = __float_as_uint(uq[2][px & 1][px >> 1]);
                                if (P.out10) {
                                    pk[px] = (cb << 20) | ((cg << 10) | (cr + 0x75000000u));       // 0x4B000000 | k: see vp_fused_up2x.h
                                } else {
                                    const uint32_t bg = __builtin_amdgcn_perm(cg, cb, 0x0c0c0400u);
                                    pk[px] = __builtin_amdgcn_perm(cr, bg, 0x0d040100u);
                                }
                            }
                        } else {
                            // generic epilogue: no final pass (straight UNORM store into the RT) and / or R10G10B10A2 target, any alignment
#pragma unroll
                            for (int px = 0; px < 4; px++) {
                                float c3[3];
#pragma unroll
                                for (int c = 0; c < 3; c++) {
                                    const float q = floorf(fmaf(fin[c][px & 1][px >> 1], P.final_pass ? P.maxv : P.quant, 0.5f));
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
                        const gptr rowp = pdst + (uint32_t)wy * (uint32_t)P.dst_pitch;
                        if (EPI != EPI_GENERIC || st_aligned) {
                            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                            u32x4 v4 = {pk[0], pk[1], pk[2], pk[3]};
                            *(__attribute__((address_space(1))) u32x4 *)(rowp + opaque(lane_off)) = v4;
                        } else {
                            __attribute__((address_space(1))) uint32_t *dst = (__attribute__((address_space(1))) uint32_t *)(rowp + lane_off);
                            dst[0] = pk[0]; dst[1] = pk[1]; dst[2] = pk[2]; dst[3] = pk[3];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 3; c++)
#pragma unroll
                        for (int m = 0; m < 3; m++) { in2[c][m] = in1[c][m]; in1[c][m] = pr[c][m + 1]; }
                }
            }
            }
            if (JSLOTS < 4 && t + 1 < n_iter) {       // the oldest pair's slot is free now (the wave's LDS reads and writes execute in order)
                jinc_wave_sync();
                stage_c(a + 2, (u + 1) & 1, next_off);
            }
            slot_t = slot_t + 1 == JSLOTS ? 0 : slot_t + 1;
        }
    }
}

template <int TAIL, int SRC, int EPI, int XC = XC_NEVER>
__global__ __launch_bounds__(64 * jinc_waves(TAIL)) void k_fused_jinc2x(FusedArgs P, const float *__restrict__ jtab, const FusedFrame *__restrict__ frames, FusedFrame single)
{
    fused_jinc2x_body<TAIL, SRC, EPI, XC>(P, jtab, frames, single);
}
// the kernel of an instantiation: its exact-form twin where one exists and the launch asks for it (exact_capable, vp_fused_dev.h)
template <int TAIL, int SRC, int EPI>
inline auto fused_jinc2x_kernel(bool exact) -> decltype(&k_fused_jinc2x<TAIL, SRC, EPI, XC_NEVER>)
{
    if constexpr (exact_capable<TAIL, SRC, EPI == EPI_DITHER8>() == XC_RUNTIME) { if (exact) return k_fused_jinc2x<TAIL, SRC, EPI, XC_ALWAYS>; }
    return k_fused_jinc2x<TAIL, SRC, EPI, XC_NEVER>;
}

}  // namespace

size_t FusedJincTableBytes() { return JTAB_FLOATS * sizeof(float); }
// dynamic LDS one workgroup of the plan's instantiation claims (the ring of converted rows of its waves + the dither tables + the tone-map
// table of a tail that has one): 114 - 146 KiB.  The planner compares it with DeviceLdsLimit() and keeps the convert + k_jinc2 draws where a
// device (or partition) grants less — the launch would otherwise fail on every frame of such a plan (advisor, round 5).
size_t FusedJincLdsBytes(const FusedParams &P)
{
    const int tailk = FusedTailKind(P);
    return (size_t)jinc_ring_bytes(tailk) + LDS_D + LDS_DB + (tail_has_table(tailk) ? LDS_T : 0);
}
// The phase table of a 2x draw (BuildJincPhases: [phase y][phase x][j * 4 + i]) in the order stage J reads it.
void BuildFusedJincTable(const void *phases, float *out)
{
    const JincPhases &t = *(const JincPhases *)phases;
    for (int sr = 0; sr < 5; sr++)
        for (int rp = 0; rp < 2; rp++)
            for (int cp = 0; cp < 2; cp++)
                for (int i = 0; i < 4; i++) {
                    const int j = sr - rp;
                    out[sr * 16 + rp * 8 + cp * 4 + i] = (j >= 0 && j <= 3) ? t.w[rp][cp][j * 4 + i] : 0.0f;
                }
    for (int rp = 0; rp < 2; rp++)
        for (int cp = 0; cp < 2; cp++) out[JTAB_W + rp * 2 + cp] = 1.0f / t.wsum[rp][cp];
}

// `a` is complete but for seg_rows; jtab_dev: FusedJincTableBytes() on the device
hipError_t LaunchFusedJinc2x(const FusedParams &P, const FusedArgs &a_in, const float *jtab_dev, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s)
{
    static const int seg_env = EnvInt("MPCVR_JINC_SEG", 0);
    const ConvertParams &c = P.conv;
    FusedArgs a = a_in;
    const int strips = (c.out_w + S - 1) / S;
    const int tailk = FusedTailKind(P), srck = FusedSourceKind(P), jw = jinc_waves(tailk);
    int seg = seg_env;
    if (seg <= 0) {
        // One workgroup per CU (LDS), so a launch runs in rounds of CUs x 12 waves and a last round that is half empty costs a whole one
        // (32 frames of 1080p: 90-row segments = 6,144 items = 2 rounds: 45.7 k frames/s; 120 rows = 1.5 rounds: 35.7 k; 72 rows = 2.5: 39.0 k).
        // Cost of a candidate = rounds x rows an item walks (its segment + 7 rows of run-in); the longest segment among the cheapest.
        const long side = (long)n_frames * (P.inflight > 1 ? P.inflight : 1);
        const long resident = (long)DeviceCuCount() * jw;
        long best = -1;
        for (int cand : {180, 144, 120, 108, 90, 72, 60, 48, 36, 24}) {
            const long items = (long)strips * ((c.out_h + cand - 1) / cand) * side;
            const long cost = ((items + resident - 1) / resident) * (long)(std::min(cand, c.out_h) + 7);
            if (best < 0 || cost < best) { best = cost; seg = cand; }
        }
    }
    seg = (seg + 1) & ~1;
    if (seg > c.out_h) seg = c.out_h;
    a.seg_rows = seg;
    const dim3 grid((strips * ((c.out_h + seg - 1) / seg) + jw - 1) / jw, 1, n_frames), block(64 * jw, 1, 1);
    const size_t lds = FusedJincLdsBytes(P);
    if (lds > DeviceLdsLimit()) return hipErrorInvalidValue;            // (UpdatePlan does not pick this route then)
    const bool aligned = P.dst_aligned16 && (a.off_x & 3) == 0 && (a.dst_pitch & 15) == 0;
    const int epik = !aligned ? EPI_GENERIC
                   : (!a.out10 && a.final_pass && a.epi_mul != 0) ? EPI_DITHER8
                   : (!a.final_pass && P.store.dst_fmt == SF_BGRA8 && P.store.quant == 255) ? EPI_DIRECT8
                   : (!a.final_pass && P.store.dst_fmt == SF_RGB10A2 && P.store.quant == 1023) ? EPI_DIRECT8 : EPI_GENERIC;
    // instantiated: the bi-planar 16-bit loader and the run-time one with the integer final pass or the generic epilogue, per tail; NV12's own
    // loader (no tail) with the straight store or the generic epilogue.  Everything else runs one of these (the straight 10-bit store of an
    // HDR passthrough: the generic epilogue)
#define MPCVR_JL3(TK, SK, EK) do { \
        auto kern = fused_jinc2x_kernel<TK, SK, EK>(a.exact_cv != 0); \
        if (lds > 48 * 1024) { const hipError_t ea = AllowLargeLds((const void *)kern, lds); if (ea != hipSuccess) return ea; } \
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, jtab_dev, frames_dev, single); } while (0)
#define MPCVR_JL(TK) do { \
        if (srck == SRC_P01X && epik == EPI_DITHER8) MPCVR_JL3(TK, SRC_P01X, EPI_DITHER8); \
        else if (srck == SRC_P01X) MPCVR_JL3(TK, SRC_P01X, EPI_GENERIC); \
        else if (epik == EPI_DITHER8) MPCVR_JL3(TK, SRC_GENERIC, EPI_DITHER8); \
        else MPCVR_JL3(TK, SRC_GENERIC, EPI_GENERIC); } while (0)
    if (tailk == TAILK_NONE) {
        if (srck == SRC_NV12 && epik == EPI_DIRECT8) MPCVR_JL3(TAILK_NONE, SRC_NV12, EPI_DIRECT8);
        else if (srck == SRC_NV12) MPCVR_JL3(TAILK_NONE, SRC_NV12, EPI_GENERIC);
        else MPCVR_JL(TAILK_NONE);
    } else if (tailk == TAILK_PQ_LUT) MPCVR_JL(TAILK_PQ_LUT);
    else if (tailk == TAILK_HLG) MPCVR_JL(TAILK_HLG);
    else MPCVR_JL(TAILK_ALU);
#undef MPCVR_JL
#undef MPCVR_JL3
    return hipGetLastError();
}

}  // namespace mpcvr
