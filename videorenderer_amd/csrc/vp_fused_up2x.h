// vp_fused_up2x.h — the fused exact-2x kernel (the headline): convert -> X pass -> Y pass -> final pass in ONE kernel.
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
// This header holds the kernel template and its per-tap-count launcher; vp_fused_up2x_nt{4,5,6}.hip instantiate one tap count
// each (36 kernels per translation unit, compiled side by side), vp_fused.hip keeps the block-convert kernels and the dispatch.
// -DMPCVR_UP2X_HEADLINE_ONLY (tools/isa_headline.sh): only the headline instantiation <5, PQ table, P01x, integer dither>.
#pragma once
#include "vp_fused_dev.h"

namespace mpcvr {

namespace {

#ifndef MPCVR_UP2X_WAVES
#define MPCVR_UP2X_WAVES 3     // waves per SIMD the register allocation aims at (experiment builds: tools/build_variant.sh)
#endif
template <int NT, int TAIL, int SRC, int EPI, int XC>
__device__ __forceinline__ void fused_up2x_body(const FusedArgs &P, const FusedFrame *__restrict__ frames, const FusedFrame &single)
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
    if (tail_has_table(TAIL))
        for (int i = threadIdx.x; i < LUT_N; i += 256) {
            const float v = P.lut[i], n = P.lut[min(i + 1, LUT_N - 1)];
            T[i] = f2{v, n - v};
        }
    __syncthreads();                                   // the only workgroup barrier: tables visible

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;      // (wave-uniform, and the compiler should know)
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
    // raw codes are prefetched TWO iterations ahead, in two buffers used alternately: vmcnt counts loads and stores in issue
    // order, so a load only reports back once the output stores issued before it have been acknowledged — with one iteration
    // of distance every iteration waited for the previous iteration's stores
    Raw raw2[2];
    RawAddr ra;
    make_raw_addr<SRC>(P, Xg, ra);
    load_raw<SRC>(P, py, ra, clampi(s0 - 3, 0, H - 1), clampi(s0 - 2, 0, H - 1), raw2[0]);
    load_raw<SRC>(P, py, ra, clampi(s0 - 1, 0, H - 1), clampi(s0, 0, H - 1), raw2[1]);

    // stage C for virtual rows ar, ar+1 (whose raw codes were prefetched into buffer b): convert, write A, prefetch rows ar+4, ar+5
    auto stage_c = [&](int ar, int b) {
        f2 rc[2][3];
        convert_block<TAIL, SRC, DV_NONE, XC, XC == XC_ALWAYS ? OUT_CODE_F : OUT_NORM>(P, MM, GG, CC, raw2[b], P.rect_t + clampi(ar, 0, H - 1), P.rect_t + clampi(ar + 1, 0, H - 1), T, rc);
        load_raw<SRC>(P, py, ra, clampi(ar + 4, 0, H - 1), clampi(ar + 5, 0, H - 1), raw2[b]);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // store to m_TexConvertOutput (UNORM: floor(sat(x)*maxv + 0.5)) and read back (q/maxv to 1 ulp)
            // (the exact form hands over the codes themselves)
            f2 qe = (XC == XC_ALWAYS ? rc[0][c] : unorm_round2(rc[0][c], cmax2, big2)) * cinv2;             // even column, rows (a, a+1)
            f2 qo = (XC == XC_ALWAYS ? rc[1][c] : unorm_round2(rc[1][c], cmax2, big2)) * cinv2;             // odd column
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
    stage_c(s0 - 3, 0);

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
                    f2 o[4];                                  // 4 output columns x (row a, row a+1), four chains in lockstep
                    // output column i = 2*kk + odd: even output 2k: base = k-1; odd output 2k+1: base = k;
                    // source k = 2l + kk (kk = 0, 1) -> av index of k is kk + 4
                    tapsN<NT, false, 4>([&](int i) -> const f2 (&)[3] { return WT[i & 1]; },
                                        [&](int i, int tt) { return av[4 + (i >> 1) + ((i & 1) ? 0 : -1) + tap_off<NT>(tt)]; }, o);
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
            if (t + 1 < n_iter) stage_c(a + 2, (u + 1) & 1);

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
                        f2 res6[6];                               // [channel * 2 + pixel pair], saturated: six chains in lockstep
                        tapsN<NT, true, 6>([&](int) -> const f2 (&)[3] { return WT[par]; },
                                           [&](int i, int tt) { return win[(2 * u + 2 + kk + 2 + par + tap_off<NT>(tt)) & 7][i >> 1][i & 1]; }, res6);
                        f2 res[3][2];
#pragma unroll
                        for (int c = 0; c < 3; c++) { res[c][0] = res6[2 * c]; res[c][1] = res6[2 * c + 1]; }
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
                                const uint32_t cr = __float_as_uint(uq[0][px >> 1][px & 1]), cg = __float_as_uint(uq[1][px >> 1][px & 1]), cb = __float_as_uint(uq[2][px >> 1][px & 1]);
                                if (P.out10) {      // R10G10B10A2 target (HDR passthrough): the codes are 0x4B000000 | k — shifted left by 10 / 20
                                                    // only k remains, and + 0x75000000 turns the red code into k | 3 << 30
                                    pk[px] = (cb << 20) | ((cg << 10) | (cr + 0x75000000u));
                                } else {
                                    const uint32_t bg = __builtin_amdgcn_perm(cg, cb, 0x0c0c0400u);
                                    pk[px] = __builtin_amdgcn_perm(cr, bg, 0x0d040100u);
                                }
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
                            __attribute__((address_space(1))) u32x4 *sp = (__attribute__((address_space(1))) u32x4 *)(rowp + opaque(lane_off));
                            // (plain stores: non-temporal ones and a raised wave priority around the store were A/B-measured in round 3 — both
                            // inside the run-to-run spread; the template argument that carried them is gone)
                            *sp = v4;
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

template <int NT, int TAIL, int SRC, int EPI, int XC = XC_NEVER>
__global__ __launch_bounds__(256, MPCVR_UP2X_WAVES) void k_fused_up2x(FusedArgs P, const FusedFrame *__restrict__ frames, FusedFrame single)
{
    fused_up2x_body<NT, TAIL, SRC, EPI, XC>(P, frames, single);
}
// the kernel of an instantiation: its exact-form twin where one exists and the launch asks for it (exact_capable, vp_fused_dev.h)
template <int NT, int TAIL, int SRC, int EPI>
inline auto fused_up2x_kernel(bool exact) -> decltype(&k_fused_up2x<NT, TAIL, SRC, EPI, XC_NEVER>)
{
    if constexpr (exact_capable<TAIL, SRC, EPI == EPI_DITHER8>() == XC_RUNTIME) { if (exact) return k_fused_up2x<NT, TAIL, SRC, EPI, XC_ALWAYS>; }
    return k_fused_up2x<NT, TAIL, SRC, EPI, XC_NEVER>;
}

}  // namespace

// grid / LDS / (source, epilogue, tail) dispatch of one tap count; `a` is complete (FillFusedArgs + weights + seg_rows)
template <int NT>
hipError_t LaunchFusedUp2xNT(const FusedParams &P, const FusedArgs &a_in, int strips, int seg, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s)
{
    const ConvertParams &c = P.conv;
    const dim3 grid((strips + WAVES - 1) / WAVES, (c.out_h + seg - 1) / seg, n_frames);
    const dim3 block(256, 1, 1);
    const int tailk = FusedTailKind(P);
    static const int lds_pad = EnvInt("MPCVR_FUSED_LDS_PAD", 0);   // experiments: lower the occupancy by claiming more LDS
    const FusedArgs &a = a_in;
    const size_t lds = LDS_A + LDS_D + LDS_DB + (tail_has_table(tailk) ? LDS_T : 0) + (size_t)lds_pad;
    const int srck = FusedSourceKind(P);
    // the specialised epilogues use 16-byte stores / dither reads: off_x % 4 == 0 and 16-byte aligned rows; the integer
    // final pass additionally needs k*M + (j << 14) < 2^32 and M < 2^24 (true for 10-bit internal -> 8-bit target)
    const bool aligned = P.dst_aligned16 && (a.off_x & 3) == 0 && (a.dst_pitch & 15) == 0;
    const int epik = !aligned ? EPI_GENERIC
                   : (!a.out10 && a.final_pass && a.epi_mul != 0) ? EPI_DITHER8
                   : (!a.final_pass && P.store.dst_fmt == SF_BGRA8 && P.store.quant == 255) ? EPI_DIRECT8
                   : (!a.final_pass && P.store.dst_fmt == SF_RGB10A2 && P.store.quant == 1023) ? EPI_DIRECT8 : EPI_GENERIC;
    // instantiated (source, epilogue) pairs: each source with the epilogue it normally meets + the generic one
#define MPCVR_LAUNCH3(NTK, TK, SK, EK) hipLaunchKernelGGL((fused_up2x_kernel<NTK, TK, SK, EK>(a.exact_cv != 0)), grid, block, lds, s, a, frames_dev, single)
#define MPCVR_LAUNCH(NT, TK) do { \
        if (srck == SRC_P01X && epik == EPI_DITHER8) MPCVR_LAUNCH3(NT, TK, SRC_P01X, EPI_DITHER8); \
        else if (srck == SRC_P01X && epik == EPI_DIRECT8) MPCVR_LAUNCH3(NT, TK, SRC_P01X, EPI_DIRECT8); \
        else if (srck == SRC_P01X) MPCVR_LAUNCH3(NT, TK, SRC_P01X, EPI_GENERIC); \
        else if (srck == SRC_PLANAR16 && epik == EPI_DITHER8) MPCVR_LAUNCH3(NT, TK, SRC_PLANAR16, EPI_DITHER8); \
        else if (epik == EPI_DITHER8) MPCVR_LAUNCH3(NT, TK, SRC_GENERIC, EPI_DITHER8); \
        else MPCVR_LAUNCH3(NT, TK, SRC_GENERIC, EPI_GENERIC); } while (0)
    // the 8-bit loaders exist without a tail only (FusedSourceKind sends 8-bit samples behind a tail through the run-time variant)
#define MPCVR_LAUNCH_8(NT) do { \
        if (srck == SRC_NV12 && epik == EPI_DIRECT8) MPCVR_LAUNCH3(NT, TAILK_NONE, SRC_NV12, EPI_DIRECT8); \
        else if (srck == SRC_NV12) MPCVR_LAUNCH3(NT, TAILK_NONE, SRC_NV12, EPI_GENERIC); \
        else if (srck == SRC_PLANAR8 && epik == EPI_DIRECT8) MPCVR_LAUNCH3(NT, TAILK_NONE, SRC_PLANAR8, EPI_DIRECT8); \
        else MPCVR_LAUNCH(NT, TAILK_NONE); } while (0)
#define MPCVR_LAUNCH_NT(NT) \
    do { if (tailk == TAILK_NONE) MPCVR_LAUNCH_8(NT); else if (tailk == TAILK_PQ_LUT) MPCVR_LAUNCH(NT, TAILK_PQ_LUT); \
         else if (tailk == TAILK_HLG) MPCVR_LAUNCH(NT, TAILK_HLG); else MPCVR_LAUNCH(NT, TAILK_ALU); } while (0)
#ifdef MPCVR_UP2X_HEADLINE_ONLY
    MPCVR_LAUNCH3(NT, TAILK_PQ_LUT, SRC_P01X, EPI_DITHER8);
#else
    MPCVR_LAUNCH_NT(NT);
#endif
#undef MPCVR_LAUNCH_NT
#undef MPCVR_LAUNCH_8
#undef MPCVR_LAUNCH
#undef MPCVR_LAUNCH3
    return hipGetLastError();
}

}  // namespace mpcvr
