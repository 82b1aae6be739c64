// vp_fused_period.h — the fused path for the everyday RATIONAL ratios (4:3, 3:2, 2:3, 1:2, 3:1 on the vertical axis; any ratio along
// the rows): convert -> X draw -> Y draw -> final pass in ONE kernel, the vertical window in REGISTERS.
//
// The reference runs every geometry through the same draws (ResizeShaderPass, DX11VideoProcessor.cpp:3103-3187); the exact-2x
// kernel (vp_fused_up2x.h) owes its speed to two things the arbitrary-ratio kernel (vp_fused_strip.hip) gives up: the vertical
// window lives in registers with compile-time slots, and the taps are packed fp32 FMAs on pixel pairs.  Both need nothing but a
// PERIODIC vertical phase pattern: with out : in = P : Q the source row of output row y = PB*m + r is base(r) + 6m (PB = 6P/Q
// outputs per 6 source rows), so over a body of three source row pairs every window slot and every emission point is static.
// 1080p -> 1440p and 720p -> 960p (4:3), 720p -> 1080p and 1440p -> 4K (3:2), 4K -> 1440p and 1080p -> 720p (2:3, the
// interpolation shader: DX11VideoProcessor.cpp:3108 takes the downscale shader only beyond 2x with bInterpolateAt50pct) and
// 4K -> 1080p (1:2) are the ratios of that kind.  Every intermediate rounding of the reference is kept (convert output -> UNORM,
// X draw -> fp16, Y draw -> UNORM, final pass in integers), and so are the reference's own per-column / per-row WEIGHTS: they come
// from BuildAxisTaps' tables (FillVertices' fp32 texture coordinates and all), only the tap INDICES are compile-time.
//
// One wavefront = one strip of up to 128 output columns (2 adjacent pixels per lane: one 8-byte store per lane and output row) x one
// segment of output rows, marching down the source rows two at a time, no workgroup barrier in the loop:
//   stage C  as in k_fused_strip: lane j converts the 2x2 blocks of the strip's source window, 64 per pass, raw codes of pass 0
//            prefetched one row pair ahead; the UNORM-rounded values go to this wave's LDS slice A as (row 0, row 1) fp32 pairs,
//            A[column][channel] (24 bytes per column: one immediate offset per channel)
//   stage X  per tap and channel one ds_read_b64 at the lane's own (tap-table) offset + one v_pk_fma_f32 on the (row 0, row 1)
//            pair with the lane's weight out of a VGPR pair (op_sel picks the half): 3 packed FMAs per tap and pixel for both rows;
//            the fp16 rounding (m_TexResize) doubles as the (row pair) -> (pixel pair) transposition
//   stage Y  six window rows win[slot][channel] = (pixel 0, pixel 1); row R sits in slot (R + 1) mod 6, inserted one at a time;
//            after row R every output row whose last tap is R is emitted: v_pk_fma_f32 on the pixel pair with the row's weights
//            from the scalar cache (s_load through the constant address space), the last tap saturating; then vp_fused_up2x.h's
//            integer final pass (v_mad_u32_u24 + two v_perm_b32 per pixel) or the straight UNORM store
// Instantiated per (P, Q) by vp_fused_period_{4_3,3_2,2_3,1_2}.hip.
#pragma once
#include <type_traits>
#include <utility>

#include "vp_fused_dev.h"

namespace mpcvr {

struct PeriodArgs {
    const int32_t *xi_t; const float *xw_t;     // X taps, tap-major [NT][out_w]; Lanczos3's shared texel (quirk Q1) folded: 5 taps
    const float *yw;                            // Y weights, [out_h][8] (NT used, zero padding), same folding; 3:1: slot 7 of a body's first row = its centre-row bits
    const int32_t *xstrip;                      // [n_strips][2] {smallest, largest} source column any tap of the strip reads
    int out_w, out_h, n_strips, seg_rows, acols;
    int strip_w;                                // output columns per wavefront (even, <= 128): 2 per lane
    int own;                                    // which two columns a lane owns (PeriodLaneColumn, vp_plan.h): 1 = (lane, 64 + lane), 2 = (2 (lane & 31) +
                                                // (lane >> 5), 64 + the same) — the planner picks the one whose X-stage reads meet no LDS bank twice; each
                                                // row leaves as two dword stores of 256 contiguous bytes per wavefront.  0 = the adjacent pair (2 lane,
                                                // 2 lane + 1) of rounds 2-3, kept for A/B runs (MPCVR_PERIOD_OWN): every ds_read_b64 of a 4:3 frame a 2-way conflict
    // SRC_SURFACE: no convert stage — the X draw samples m_TexConvertOutput as another convert kernel wrote it (Dolby Vision, Catmull-Rom
    // chroma ...: B8G8R8A8 or R10G10B10A2 texels); frame z of a batch reads surf + z * surf_stride (null: FusedFrame::src)
    const uint8_t *surf; int surf_fmt, surf_pitch, surf_w; size_t surf_stride;
    const int32_t *other;                       // SRC_SURFACE: the X draw's row map (row of m_TexResize -> surface row: a source rect, rotation 180); null = identity
};

namespace {

constexpr int kPeriodStripMax = 128;            // output columns per wavefront at most: 2 per lane
#ifndef MPCVR_PERIOD_THREADS
#define MPCVR_PERIOD_THREADS 1024
#endif
constexpr int kPeriodMaxThreads = MPCVR_PERIOD_THREADS;      // waves per workgroup x 64: 16 waves = 4 per SIMD = 128 VGPRs

// floor((2r + 1) Q - P) / (2P)): base source row (relative to 6m) of output row PB*m + r — pos = (r + .5) Q/P - .5
__host__ __device__ constexpr int period_base(int P, int Q, int r)
{
    const int num = (2 * r + 1) * Q - P, den = 2 * P;
    return num >= 0 ? num / den : -((-num + den - 1) / den);
}
__host__ __device__ constexpr int mod6(int v) { return ((v % 6) + 6) % 6; }
template <int NT> __host__ __device__ constexpr int tap_hi() { return NT == 4 ? 2 : 3; }
// output phase r is emitted right after the source row in slot period_rho (its last tap) went in; it belongs to period j + delta.
// sh = 1: the same phase one source row lower (period_centre rows whose fp32 texcoord fell below the texel centre)
template <int NT> __host__ __device__ constexpr int period_rho(int P, int Q, int r, int sh = 0) { return mod6(1 + period_base(P, Q, r) - sh + tap_hi<NT>()); }
template <int NT> __host__ __device__ constexpr int period_delta(int P, int Q, int r, int sh = 0)
{
    return (period_rho<NT>(P, Q, r, sh) - 1 - (period_base(P, Q, r) - sh) - tap_hi<NT>()) / 6;      // exact: 0, -1 (or -2)
}
// P and Q both odd (3:1): phase r sits EXACTLY on a texel centre n — pos = (r + .5) Q / P - .5 is an integer — and the reference's fp32
// texcoord decides row by row whether the shader reads base n at t = 0 or base n - 1 at t = 1 - eps.  Both emissions are compiled in;
// the planner's bit per row (PeriodArgs::yw, slot 7 of each body's first row) picks one
__host__ __device__ constexpr bool period_centre(int P, int Q, int r) { return ((2 * r + 1) * Q - P) % (2 * P) == 0; }
__host__ __device__ constexpr bool period_has_centres(int P, int Q) { return (P & 1) && (Q & 1); }
// lowest base (relative to 6m) any phase of a body reads from, the lowered centre rows included
__host__ __device__ constexpr int period_low_base(int P, int Q)
{
    int lo = period_base(P, Q, 0);
    for (int r = 0; r < 6 * P / Q; r++) {
        const int b = period_base(P, Q, r) - (period_centre(P, Q, r) ? 1 : 0);
        lo = b < lo ? b : lo;
    }
    return lo;
}

template <typename T> using pcptr = const __attribute__((address_space(4))) T *;
template <typename T> __device__ __forceinline__ pcptr<T> p_const(const T *p) { return (pcptr<T>)(uintptr_t)p; }

__device__ __forceinline__ void period_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// f(integral_constant<0>) ... f(integral_constant<N - 1>): loop indices that stay compile-time constants (window slots)
template <int I, int N, typename F>
__device__ __forceinline__ void period_static_for(F &f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); period_static_for<I + 1, N>(f); }
}
template <int HALF>
__device__ __forceinline__ f2 pk_mul_wv(f2 w, f2 b)
{
    f2 r;
    if (HALF == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(w), "v"(b));
    else           asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(w), "v"(b));
    return r;
}

// PP : QQ = output rows : source rows; NT = taps per output (4: base-1..base+2; 5: Lanczos3 as Direct3D 11 draws it; 6: base-2..base+3)
template <int PP, int QQ, int NT, int TAIL, int SRC, int EPI, int XC>
__device__ __forceinline__ void fused_period_body(const FusedArgs &P, const PeriodArgs &Q, const FusedFrame *__restrict__ frames, const FusedFrame &single)
{
    static_assert(6 % QQ == 0 && (6 * PP) % QQ == 0, "a body of six source rows must hold whole periods");
    static_assert(6 * PP / QQ <= 32, "one word of centre-row bits per body");
    static_assert(EPI == EPI_DITHER8 || EPI == EPI_DIRECT8, "the generic epilogue stays with k_fused_strip");
    constexpr int PB = 6 * PP / QQ;                 // output rows per body
    constexpr int NP = (NT + 1) / 2;                // weight pairs
    constexpr bool FASTEPI = EPI == EPI_DITHER8;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *Di = (uint32_t *)smem;                                   // dither as j << 14 (FASTEPI)
    f2 *T = (f2 *)(smem + (FASTEPI ? LDS_DB : 0));
    unsigned char *wbase = smem + (FASTEPI ? LDS_DB : 0) + (tail_has_table(TAIL) ? LDS_T : 0);
    if (FASTEPI)
        for (int i = threadIdx.x; i < 1024; i += blockDim.x)
            Di[i] = (uint32_t)(__half2float(__ushort_as_half(P.dither[i])) * 1024.0f + 0.5f) << 14;
    if (tail_has_table(TAIL))
        for (int i = threadIdx.x; i < LUT_N; i += blockDim.x) {
            const float v = P.lut[i], n = P.lut[min(i + 1, LUT_N - 1)];
            T[i] = f2{v, n - v};
        }
    if (FASTEPI || tail_has_table(TAIL)) __syncthreads();             // the only workgroup barrier: tables visible

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int bx, bz;
    xcd_contiguous_block(bx, bz);                                     // (workgroup -> (item group, frame): an XCD works on neighbours, vp_fused_dev.h)
    const int item = bx * (int)(blockDim.x >> 6) + wave;
    const int seg_i = item / Q.n_strips, strip = item - seg_i * Q.n_strips;
    const int y0 = seg_i * Q.seg_rows;                                  // a multiple of PB (launcher)
    if (y0 >= Q.out_h) return;
    const int y1 = min(y0 + Q.seg_rows, Q.out_h);
    const int W = P.W, H = P.H;
    unsigned char *const Aw = wbase + wave * (Q.acols * 24);

    const FusedFrame frame = frames ? frames[bz] : single;
    auto uniform_ptr = [](const void *q) {
        const uint64_t v = (uint64_t)q;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const gcptr py = (gcptr)uniform_ptr(SRC == SRC_SURFACE && Q.surf ? (const void *)(Q.surf + (size_t)bz * Q.surf_stride) : (const void *)frame.src);
    const uint64_t dst_u = uniform_ptr(frame.dst);
    const gptr pdst = (gptr)dst_u;

    // the strip's source window: columns c0 .. hi as 2x2 blocks, 64 per pass
    const int c0 = p_const(Q.xstrip)[2 * strip] & ~1;
    const int nb = ((p_const(Q.xstrip)[2 * strip + 1] - c0) >> 1) + 1, npass = (nb + 63) >> 6;

    // stage X / Y role: output columns xq[0], xq[1] (PeriodLaneColumn: adjacent, or 64 apart so that a 32-lane group's reads stay inside
    // 32 source columns = the 64 banks a ds_read_b64 group has)
    const int xs = strip * Q.strip_w;
    // 3:1 reads at most 22 source columns per 64 outputs: the adjacent pair is conflict-free there and keeps its one 8-byte store per row
    // (same box, round 4: 720p -> 2160p 70.7 k frames/s against 69.6 k with two dword stores) — compile-time, so neither path pays for the other
    constexpr bool PAIR = PP == 3 && QQ == 1;
    const int own = PAIR ? 0 : Q.own;
    const int lcol = own == 0 ? 2 * lane : own == 1 ? lane : 2 * (lane & 31) + (lane >> 5);
    const int xq[2] = {xs + lcol, xs + lcol + (own == 0 ? 1 : 64)};
    const int x_first = xq[0];
    const int x_end = min(xs + Q.strip_w, Q.out_w);                   // the strip's columns inside the frame
    const bool act[2] = {xq[0] < x_end, xq[1] < x_end};
    typedef __attribute__((address_space(3))) const f2 *lds_f2;
    const uint32_t aw_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)Aw;
    uint32_t xo[2][NT]; f2 xwp[2][NP];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int xc = min(xq[q], x_end - 1);
        float wq[2 * NP];
#pragma unroll
        for (int k = 0; k < 2 * NP; k++) wq[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < NT; k++) {
            xo[q][k] = aw_lds + (uint32_t)(Q.xi_t[xc + (size_t)k * Q.out_w] - c0) * 24u;    // absolute LDS address: one VGPR per tap, the channel as an immediate
            wq[k] = Q.xw_t[xc + (size_t)k * Q.out_w] * P.inv_maxv;       // A holds UNORM CODES: the 1 / maxv of the texel read is folded into the weight
        }
#pragma unroll
        for (int k = 0; k < NP; k++) xwp[q][k] = f2{wq[2 * k], wq[2 * k + 1]};
    }

    const f2 MM[5] = {f2{P.m[0], P.m[1]}, f2{P.m[2], P.m[3]}, f2{P.m[4], P.m[5]}, f2{P.m[6], P.m[7]}, f2{P.m[8], 0.0f}};
    const f2 GG[5] = {f2{P.gamut[0], P.gamut[1]}, f2{P.gamut[2], P.gamut[3]}, f2{P.gamut[4], P.gamut[5]}, f2{P.gamut[6], P.gamut[7]}, f2{P.gamut[8], 0.0f}};
    const f2 cmax2 = splat(P.maxv);
    const f2 qmax2 = splat(FASTEPI ? P.maxv : P.quant);          // Y result -> m_TexsPostScale codes (final pass) or the target's own codes
    f2 big2 = splat(8388608.0f);                     // 2^23, pinned in VGPRs (see unorm_round2)
    asm volatile("" : "+v"(big2));
    const f2 CC[3] = {splat(P.c[0]), splat(P.c[1]), splat(P.c[2])};        // (not pinned in VGPRs: the window needs the registers more than the convert stage six moves less)

    // raw codes of pass 0 are prefetched one row pair ahead
    constexpr int YSRC = SRC == SRC_SURFACE ? SRC_GENERIC : SRC;      // (the YUV helpers are not instantiated for a surface)
    RawAddr ra0;
    if (SRC != SRC_SURFACE) make_raw_addr<YSRC>(P, min(c0 + 2 * lane, W - 2), ra0);
    Raw rawn;
    auto fetch = [&](int pp, const RawAddr &ra, Raw &r) __attribute__((always_inline)) {
        if (SRC != SRC_SURFACE) load_raw<YSRC>(P, py, ra, clampi(2 * pp - 1, 0, H - 1), clampi(2 * pp, 0, H - 1), r);
    };
    // SRC_SURFACE: the 2x2 texels of block b of pair pp, [row][column], one dword each (clamp addressing of the draw)
    uint32_t sraw[2][2];
    auto fetch_s = [&](int pp, int b, uint32_t (&t)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int row = clampi(2 * pp - 1 + r, 0, H - 1);
            const gcptr rowp = py + (uint32_t)(Q.other ? p_const(Q.other)[row] : row) * (uint32_t)Q.surf_pitch;
#pragma unroll
            for (int col = 0; col < 2; col++) t[r][col] = ld_u32(rowp + (uint32_t)min(c0 + 2 * b + col, Q.surf_w - 1) * 4u);
        }
    };
    // texel -> (r, g, b) codes as floats: the integers the UNORM texel holds
    auto unpack_s = [&](uint32_t t, float (&c3)[3]) __attribute__((always_inline)) {
        if (Q.surf_fmt == SF_RGB10A2) { c3[0] = (float)(t & 0x3ffu); c3[1] = (float)((t >> 10) & 0x3ffu); c3[2] = (float)((t >> 20) & 0x3ffu); }
        else { c3[0] = (float)((t >> 16) & 0xffu); c3[1] = (float)((t >> 8) & 0xffu); c3[2] = (float)(t & 0xffu); }      // B8G8R8A8: r = byte 2
    };

    // pair pp = source rows 2pp-1, 2pp (rect-relative, clamped to the rect: clamp-to-edge addressing of the draws = replicated rows).
    // stage C: convert the pair's blocks of the strip's source window into A (raw codes of pass 0 were prefetched; pair pp+1's go out now)
    auto stage_c = [&](int pp) __attribute__((always_inline)) {
        const int r0 = 2 * pp - 1;
        const int sy0 = P.rect_t + clampi(r0, 0, H - 1), sy1 = P.rect_t + clampi(r0 + 1, 0, H - 1);
        for (int pass = 0; pass < npass; pass++) {
            const int b = pass * 64 + lane;
            if constexpr (SRC == SRC_SURFACE) {     // no convert stage: the texels are m_TexConvertOutput's own codes
                uint32_t t[2][2];
                if (pass == 0) {
#pragma unroll
                    for (int r = 0; r < 2; r++) { t[r][0] = sraw[r][0]; t[r][1] = sraw[r][1]; }
                    fetch_s(pp + 1, lane, sraw);
                } else fetch_s(pp, b, t);
                float k[2][2][3];                   // [row][column][channel]
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int col = 0; col < 2; col++) unpack_s(t[r][col], k[r][col]);
                if (b < nb) {
                    f4 *dst = (f4 *)(Aw + 48 * b);
                    dst[0] = f4{k[0][0][0], k[1][0][0], k[0][0][1], k[1][0][1]};
                    dst[1] = f4{k[0][0][2], k[1][0][2], k[0][1][0], k[1][1][0]};
                    dst[2] = f4{k[0][1][1], k[1][1][1], k[0][1][2], k[1][1][2]};
                }
                continue;
            }
            f2 rc[2][3];
            if (pass == 0) {
                convert_block<TAIL, YSRC, DV_NONE, XC, XC == XC_ALWAYS ? OUT_CODE_F : OUT_NORM>(P, MM, GG, CC, rawn, sy0, sy1, T, rc);
                fetch(pp + 1, ra0, rawn);
            } else {
                RawAddr ra; Raw rw;
                make_raw_addr<YSRC>(P, min(c0 + 2 * b, W - 2), ra);
                fetch(pp, ra, rw);
                convert_block<TAIL, YSRC, DV_NONE, XC, XC == XC_ALWAYS ? OUT_CODE_F : OUT_NORM>(P, MM, GG, CC, rw, sy0, sy1, T, rc);
            }
            // store to m_TexConvertOutput (UNORM: floor(sat(x)*maxv + 0.5)): the integer codes as floats, (row 0, row 1) pairs
            f2 q[2][3];
#pragma unroll
            for (int col = 0; col < 2; col++)
#pragma unroll
                for (int c = 0; c < 3; c++) q[col][c] = XC == XC_ALWAYS ? rc[col][c] : unorm_round2(rc[col][c], cmax2, big2);      // (the exact form hands over the codes)
            if (b < nb) {           // A[column][channel]: the block's two columns are 48 contiguous bytes
                f4 *dst = (f4 *)(Aw + 48 * b);
                dst[0] = f4{q[0][0].x, q[0][0].y, q[0][1].x, q[0][1].y};
                dst[1] = f4{q[0][2].x, q[0][2].y, q[1][0].x, q[1][0].y};
                dst[2] = f4{q[1][1].x, q[1][1].y, q[1][2].x, q[1][2].y};
            }
        }
    };
    // stage X: the X draw's result of the pair in A for the lane's two pixels, rounded through fp16: rowA = row 2pp-1, rowB = row 2pp,
    // [channel] = (px 0, px 1)
    auto stage_x = [&](f2 (&rowA)[3], f2 (&rowB)[3]) __attribute__((always_inline)) {
        f2 acc[2][3];                                        // [pixel][channel] = (row 0, row 1)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            f2 t[NT][3];
#pragma unroll
            for (int k = 0; k < NT; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) t[k][c] = ((lds_f2)(uintptr_t)xo[q][k])[c];
#pragma unroll
            for (int k = 0; k < NT; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (k == 0) acc[q][c] = pk_mul_wv<0>(xwp[q][0], t[0][c]);
                    else acc[q][c] = (k & 1) ? pk_fma_wv<1>(xwp[q][k >> 1], t[k][c], acc[q][c]) : pk_fma_wv<0>(xwp[q][k >> 1], t[k][c], acc[q][c]);
                }
        }
        // m_TexResize is R16G16B16A16_FLOAT (:3155): round to fp16 (RNE), keep the rounded value as fp32
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const f2 h0 = half_round2(acc[0][c]), h1 = half_round2(acc[1][c]);
            rowA[c] = f2{h0.x, h1.x};
            rowB[c] = f2{h0.y, h1.y};
        }
    };

    // ---------------- the march ----------------
    f2 win[6][3];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int c = 0; c < 3; c++) win[i][c] = splat(0.0f);

    const uint32_t lane_off = (uint32_t)(P.off_x + x_first) * 4u;
    const uint32_t px1_off = own == 0 ? 4u : 256u;                        // wave-uniform: byte distance of the lane's second pixel
    // dither texels of the lane's two columns (32-periodic table: columns 64 apart meet the same one)
    const uint32_t dix[2] = {(uint32_t)((P.off_x + xq[0]) & 31), (uint32_t)((P.off_x + xq[1]) & 31)};
    const bool wave_full = Q.strip_w == kPeriodStripMax && xs + kPeriodStripMax <= Q.out_w;    // (lanes outside the frame filter clamped columns and store nothing)
    const pcptr<f2> ywp = (pcptr<f2>)(uintptr_t)Q.yw;

    // output row y = PB*m + r (r static): taps from the window, epilogue, one 8-byte store
    auto emit_row = [&](auto RC, auto SHC, int m, uint32_t below) __attribute__((always_inline)) {
        constexpr int r = decltype(RC)::value, SH = decltype(SHC)::value;
        const int y = PB * m + r;
        if (y < y0 || y >= y1) return;                                 // wave-uniform: rows of the neighbouring segments
        if constexpr (period_centre(PP, QQ, r))                        // (wave-uniform) the other emission of this phase draws the row
            if (((below >> r) & 1u) != (uint32_t)SH) return;
        const pcptr<f2> wr = ywp + (size_t)y * 4;
        const f2 WP[3] = {wr[0], wr[1], wr[2]};
        const int wy = P.off_y + y;
        uint32_t dj[2] = {0, 0};
        if (FASTEPI) {          // dither texels first: the LDS round trip hides behind the taps; off_x + xs is even (launcher)
            const uint32_t *drow = Di + (wy & 31) * 32;
            if constexpr (PAIR) {
                const u32x2 dd = *(const u32x2 *)(drow + dix[0]);
                dj[0] = dd.x; dj[1] = dd.y;
            } else { dj[0] = drow[dix[0]]; dj[1] = drow[dix[1]]; }
        }
        f2 res[3];
        constexpr int base = period_base(PP, QQ, r) - SH;
        tapsN<NT, true, 3>([&](int) -> const f2 (&)[3] { return WP; },
                           [&](int c, int tt) { return win[mod6(base + tap_off<NT>(tt) + 1)][c]; }, res);
        f2 uq[3];
#pragma unroll
        for (int c = 0; c < 3; c++) uq[c] = pk_fma(res[c], qmax2, big2);       // the UNORM code in the low mantissa bits
        uint32_t pk[2];
#pragma unroll
        for (int px = 0; px < 2; px++) {
            const uint32_t cr = __float_as_uint(uq[0][px]), cg = __float_as_uint(uq[1][px]), cb = __float_as_uint(uq[2][px]);
            if (FASTEPI) {      // m_TexsPostScale store/load + ps_final_pass.hlsl:29 in integers, see vp_fused_up2x.h
                const uint32_t ib = __umul24(cb, P.epi_mul) + dj[px], ig = __umul24(cg, P.epi_mul) + dj[px], ir = __umul24(cr, P.epi_mul) + dj[px];
                const uint32_t bg = __builtin_amdgcn_perm(ig, ib, 0x0c0c0703u);    // [B, G, 0, 0]
                pk[px] = __builtin_amdgcn_perm(ir, bg, 0x0d070100u);               // [B, G, R, 0xff]
            } else if (P.out10) {   // 0x4B000000 | k: shifted left by 10 or 20 only k remains; + 0x75000000 turns the red code into k | 3 << 30
                pk[px] = (cb << 20) | ((cg << 10) | (cr + 0x75000000u));
            } else {
                const uint32_t bg = __builtin_amdgcn_perm(cg, cb, 0x0c0c0400u);
                pk[px] = __builtin_amdgcn_perm(cr, bg, 0x0d040100u);
            }
        }
        const gptr rowp = pdst + (uint32_t)wy * (uint32_t)P.dst_pitch;
        // two dword stores per row: the lane's columns are 64 apart, so each store is 256 contiguous bytes per wavefront
        if constexpr (PAIR) {           // adjacent pair: one 8-byte store
            if (wave_full) *(__attribute__((address_space(1))) u32x2 *)(rowp + opaque(lane_off)) = u32x2{pk[0], pk[1]};
            else if (act[0] && act[1]) *(__attribute__((address_space(1))) u32x2 *)(rowp + lane_off) = u32x2{pk[0], pk[1]};
            else if (act[0]) *(__attribute__((address_space(1))) uint32_t *)(rowp + lane_off) = pk[0];
        } else {
            const gptr rowq = rowp + px1_off;
            if (wave_full) {                // wave-uniform: every lane of the strip owns two pixels inside the frame
                *(__attribute__((address_space(1))) uint32_t *)(rowp + opaque(lane_off)) = pk[0];
                *(__attribute__((address_space(1))) uint32_t *)(rowq + opaque(lane_off)) = pk[1];
            } else {                        // the frame's last strip / a narrow strip: lanes beyond its right edge store nothing
                if (act[0]) *(__attribute__((address_space(1))) uint32_t *)(rowp + lane_off) = pk[0];
                if (act[1]) *(__attribute__((address_space(1))) uint32_t *)(rowq + lane_off) = pk[1];
            }
        }
    };
    // after source row 6j - 1 + RHO went into slot RHO: every output phase whose last tap it is
    // below_cur / below_prev: the centre-row bits of bodies j and j - 1 (period_has_centres ratios; 0 otherwise)
    uint32_t below_cur = 0, below_prev = 0;
    auto emit_after = [&](auto RHOC, int j) __attribute__((always_inline)) {
        constexpr int RHO = decltype(RHOC)::value;
        auto one = [&](auto RC) __attribute__((always_inline)) {
            constexpr int r = decltype(RC)::value;
            if constexpr (period_rho<NT>(PP, QQ, r) == RHO) {
                constexpr int d = period_delta<NT>(PP, QQ, r);
                static_assert(!period_centre(PP, QQ, r) || d == 0 || d == -1, "centre rows are drawn within two bodies");
                emit_row(RC, std::integral_constant<int, 0>{}, j + d, d == 0 ? below_cur : below_prev);
            }
            if constexpr (period_centre(PP, QQ, r) && period_rho<NT>(PP, QQ, r, 1) == RHO) {
                constexpr int d = period_delta<NT>(PP, QQ, r, 1);
                static_assert(d == 0 || d == -1, "centre rows are drawn within two bodies");
                emit_row(RC, std::integral_constant<int, 1>{}, j + d, d == 0 ? below_cur : below_prev);
            }
        };
        period_static_for<0, PB>(one);
    };
    const int n_bodies = (Q.out_h + PB - 1) / PB;
    auto body_bits = [&](int m) __attribute__((always_inline)) -> uint32_t {
        return m >= 0 && m < n_bodies ? p_const((const uint32_t *)Q.yw)[(size_t)m * (PB * 8) + 7] : 0u;
    };

    // rows the segment's outputs read: [need_lo, need_hi] (virtual: outside 0..H-1 they replicate the edge rows)
    const int m_first = y0 / PB, m_last = (y1 - 1) / PB;
    const int need_lo = 6 * m_first + period_low_base(PP, QQ) + tap_off<NT>(0);
    const int need_hi = 6 * m_last + period_base(PP, QQ, PB - 1) + tap_hi<NT>();
    // pairs pp_lo .. pp_hi cover them: pair pp = rows 2pp - 1, 2pp
    const int pp_lo = (need_lo + 1) >> 1, pp_hi = (need_hi + 1) >> 1;      // floor((row + 1) / 2), rows may be negative
    const int j_lo = pp_lo >= 0 ? pp_lo / 3 : -((-pp_lo + 2) / 3), j_hi = pp_hi >= 0 ? pp_hi / 3 : -((-pp_hi + 2) / 3);
    // Software pipeline per pair: X(pp) -> C(pp + 1) -> the rows pair pp completes, so A's LDS write -> read round trip (and the global
    // prefetch behind it) hides behind the Y work.  A is exchanged between the lanes of this wave only: LDS operations of one wave
    // execute in order; the fences keep the compiler from reordering the reads and writes (unrelated, lane by lane).
    if (SRC == SRC_SURFACE) fetch_s(pp_lo, lane, sraw); else fetch(pp_lo, ra0, rawn);
    stage_c(pp_lo);
    if constexpr (period_has_centres(PP, QQ)) below_cur = body_bits(j_lo - 1);
    for (int j = j_lo; j <= j_hi; j++) {
        if constexpr (period_has_centres(PP, QQ)) { below_prev = below_cur; below_cur = body_bits(j); }
        auto pair_step = [&](auto IC) __attribute__((always_inline)) {
            constexpr int i = decltype(IC)::value;
            const int pp = 3 * j + i;
            if (pp < pp_lo || pp > pp_hi) return;                       // wave-uniform: outside the segment's rows
            f2 rowA[3], rowB[3];
            period_wave_sync();
            stage_x(rowA, rowB);
            period_wave_sync();
            if (pp < pp_hi) stage_c(pp + 1);
#pragma unroll
            for (int c = 0; c < 3; c++) win[2 * i][c] = rowA[c];
            emit_after(std::integral_constant<int, 2 * i>{}, j);
#pragma unroll
            for (int c = 0; c < 3; c++) win[2 * i + 1][c] = rowB[c];
            emit_after(std::integral_constant<int, 2 * i + 1>{}, j);
        };
        pair_step(std::integral_constant<int, 0>{});
        pair_step(std::integral_constant<int, 1>{});
        pair_step(std::integral_constant<int, 2>{});
    }
}

template <int PP, int QQ, int NT, int TAIL, int SRC, int EPI, int XC = XC_NEVER>
__global__ __launch_bounds__(kPeriodMaxThreads) void k_fused_period(FusedArgs P, PeriodArgs Q, const FusedFrame *__restrict__ frames, FusedFrame single)
{
    fused_period_body<PP, QQ, NT, TAIL, SRC, EPI, XC>(P, Q, frames, single);
}
// the kernel of an instantiation: its exact-form twin where one exists and the launch asks for it (exact_capable, vp_fused_dev.h)
template <int PP, int QQ, int NT, int TAIL, int SRC, int EPI>
inline auto fused_period_kernel(bool exact) -> decltype(&k_fused_period<PP, QQ, NT, TAIL, SRC, EPI, XC_NEVER>)
{
    if constexpr (exact_capable<TAIL, SRC, EPI == EPI_DITHER8>() == XC_RUNTIME) { if (exact) return k_fused_period<PP, QQ, NT, TAIL, SRC, EPI, XC_ALWAYS>; }
    return k_fused_period<PP, QQ, NT, TAIL, SRC, EPI, XC_NEVER>;
}

}  // namespace

// per-(P, Q) launcher, instantiated by vp_fused_period_*.hip: every (taps, tail, source, epilogue) combination the planner can pick
template <int PP, int QQ>
hipError_t LaunchFusedPeriodPQ(const FusedArgs &a, const PeriodArgs &q, int nt, int tailk, int srck, int epik, dim3 grid, dim3 block, size_t lds,
                               const FusedFrame *frames_dev, FusedFrame single, hipStream_t s)
{
#define MPCVR_PD5(NT, TK, SK, EK) do { \
        auto kern = fused_period_kernel<PP, QQ, NT, TK, SK, EK>(a.exact_cv != 0); \
        if (lds > 48 * 1024) { \
            const hipError_t ea = AllowLargeLds((const void *)kern, lds); \
            if (ea != hipSuccess) return ea; \
        } \
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, q, frames_dev, single); } while (0)
    // (source, epilogue) pairs as for the exact-2x kernel: each specialised source with the epilogue it normally meets
#define MPCVR_PD3(NT, TK) do { \
        if (srck == SRC_P01X && epik == EPI_DITHER8) MPCVR_PD5(NT, TK, SRC_P01X, EPI_DITHER8); \
        else if (srck == SRC_P01X) MPCVR_PD5(NT, TK, SRC_P01X, EPI_DIRECT8); \
        else if (epik == EPI_DITHER8) MPCVR_PD5(NT, TK, SRC_GENERIC, EPI_DITHER8); \
        else MPCVR_PD5(NT, TK, SRC_GENERIC, EPI_DIRECT8); } while (0)
    // (the NV12 loader exists without a tail only: FusedSourceKind)
#define MPCVR_PD2(NT) do { if (tailk == TAILK_NONE && srck == SRC_NV12 && epik == EPI_DIRECT8) MPCVR_PD5(NT, TAILK_NONE, SRC_NV12, EPI_DIRECT8); \
                           else if (tailk == TAILK_NONE) MPCVR_PD3(NT, TAILK_NONE); else if (tailk == TAILK_PQ_LUT) MPCVR_PD3(NT, TAILK_PQ_LUT); \
                           else if (tailk == TAILK_HLG) MPCVR_PD3(NT, TAILK_HLG); else return hipErrorNotSupported; } while (0)
    // four taps without a tail (SDR content through Mitchell / Catmull-Rom / Lanczos2): k_fused_strip is as fast or faster there (FusedPeriodTakes),
    // so only the table tails are built with four taps
#define MPCVR_PD2_TAILS(NT) do { if (tailk == TAILK_PQ_LUT) MPCVR_PD3(NT, TAILK_PQ_LUT); else if (tailk == TAILK_HLG) MPCVR_PD3(NT, TAILK_HLG); \
                                 else return hipErrorNotSupported; } while (0)
    if (srck == SRC_SURFACE) {      // the convert output of another kernel: no tail, both epilogues
#define MPCVR_PDS(NT) do { if (epik == EPI_DITHER8) MPCVR_PD5(NT, TAILK_NONE, SRC_SURFACE, EPI_DITHER8); else MPCVR_PD5(NT, TAILK_NONE, SRC_SURFACE, EPI_DIRECT8); } while (0)
        if (nt == 4) MPCVR_PDS(4); else if (nt == 5) MPCVR_PDS(5); else return hipErrorNotSupported;
#undef MPCVR_PDS
        return hipGetLastError();
    }
#ifdef MPCVR_PERIOD_DEV_ONLY
#ifndef MPCVR_PERIOD_DEV_NT
#define MPCVR_PERIOD_DEV_NT 5
#endif
    MPCVR_PD5(MPCVR_PERIOD_DEV_NT, TAILK_PQ_LUT, SRC_P01X, EPI_DITHER8);
#else
    // (round 5: the 6-tap variants — Spline36, the as-intended Lanczos3 of MPCVR_FLAG_LANCZOS3_FIXED — are no longer built: 75 instantiations, a
    // fifth of the library's build time, for two settings no reference build offers; k_fused_strip draws those frames)
    if (nt == 4) MPCVR_PD2_TAILS(4);
    else if (nt == 5) MPCVR_PD2(5);
    else return hipErrorNotSupported;
#endif
#undef MPCVR_PD2_TAILS
#undef MPCVR_PD2
#undef MPCVR_PD3
#undef MPCVR_PD5
    return hipGetLastError();
}

}  // namespace mpcvr
