// vp_fused_mx.hip — the fused exact-2x kernel with both resize passes on the MATRIX CORES.
//
// Same path, same roundings and the same wave-autonomous strip design as k_fused_up2x (vp_fused.hip: convert -> X pass -> Y pass
// -> final pass in one kernel, every intermediate rounding of the reference kept).  What changes is where the 5- / 6-tap filters
// run: k_fused_up2x spends 180 of its ~500 VALU instructions per iteration on them (v_pk_fma_f32 chains) and is VALU-issue-bound;
// here they are v_mfma_f32_16x16x32_f16 instructions, which execute on the matrix pipe beside the VALU work.
//
// Why fp16 operands are EXACT here: the X pass reads m_TexConvertOutput texels — UNORM8/10 codes k, and k * 2^-10 is an fp16
// number — and the Y pass reads m_TexResize texels, which the reference itself stores as fp16 (DX11VideoProcessor.cpp:3155).
// Only the weights are not fp16; they are split hi + lo (two MFMAs accumulate), which keeps 22 bits of them.  Products are exact
// in fp32 and the accumulation is fp32, so results differ from the fp32 FMA chain only in the last ulp of the sum.
//
// "Own column" operand layout.  D = A x B with A[i][k] (lane = i + 16*(k/8)), B[k][j] (lane = j + 16*(k/8)), D[i][j] in lane
// j + 16*(i/4), register i%4.  A holds the weights and is block-diagonal: row block i/4 == g is non-zero only for k/8 == g.  Then
// D[4g+q][j] = sum_t A[4g+q][8g+t] * B[8g+t][j], i.e. LANE L's FOUR RESULTS DEPEND ONLY ON LANE L's EIGHT B VALUES:
//     X pass: B = the 8 source columns a lane's 4 output columns read (one row, one channel)   -> D = its 4 output columns
//     Y pass: B = the 8-row window of one output column / channel (4 VGPRs of packed fp16)     -> D = its 4 output rows
// No cross-lane traffic, no transposes; the register window of the Y pass halves (packed fp16 instead of fp32).
// Scales (powers of two, so that no fp16 operand is subnormal): B_x = k * 2^-10, A_x = w * 2^12 * 1024/maxv, the X result enters
// the window as fp16(result * 2^12) — the same rounding as fp16(result) — A_y = w * 2^6, D_y = result * 2^18.
#include "vp_fused_dev.h"

namespace mpcvr {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int HW = 136;                                  // fp16 per row of a wave's A slice: index = A column + 1 (1..128 used)
constexpr int MX_A_HALFS = 3 * 2 * HW;                   // [channel][row a | row a+1][index]
constexpr int MX_LDS_A = WAVES * MX_A_HALFS * 2;         // 6528 B
constexpr int MX_LDS_WY = 4 * 2 * 64 * 16;               // Y weights: [window rotation u][hi | lo][lane] x 16 B
constexpr float SX = 4096.0f, SY = 64.0f;

__device__ __forceinline__ f2 pk_mul_clamp(f2 a, f2 b)
{
    f2 r;
#ifdef MPCVR_NO_PK
    return f2{__builtin_amdgcn_fmed3f(a.x * b.x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(a.y * b.y, 0.0f, 1.0f)};
#endif
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t cvt_pk_f16(float lo, float hi)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, h2v));
}
__device__ __forceinline__ void split8(const float (&w)[8], h8 &hi, h8 &lo)
{
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const _Float16 h = (_Float16)w[t];
        hi[t] = h;
        lo[t] = (_Float16)(w[t] - (float)h);
    }
}

template <int NT, int TAIL, int SRC, int EPI>
__global__ __launch_bounds__(256, 3) void k_fused_up2x_mx(FusedArgs P, const FusedFrame *__restrict__ frames, FusedFrame single)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *AHall = (_Float16 *)smem;
    h8 *WY = (h8 *)(smem + MX_LDS_A);
    unsigned short *D = (unsigned short *)(smem + MX_LDS_A + MX_LDS_WY);
    uint32_t *Di = (uint32_t *)(smem + MX_LDS_A + MX_LDS_WY + LDS_D);
    f2 *T = (f2 *)(smem + MX_LDS_A + MX_LDS_WY + LDS_D + LDS_DB);

    for (int i = threadIdx.x; i < 1024; i += 256) {
        const unsigned short d = P.dither[i];
        D[i] = d;
        Di[i] = (uint32_t)(__half2float(__ushort_as_half(d)) * 1024.0f + 0.5f) << 14;
    }
    if (tail_has_table(TAIL))
        for (int i = threadIdx.x; i < LUT_N; i += 256) {
            const float v = P.lut[i], n = P.lut[min(i + 1, LUT_N - 1)];
            T[i] = f2{v, n - v};
        }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // weight fragments.  A lane holds row i = lane & 15 of A for k = 8 * (lane >> 4) .. +7; only the diagonal blocks are non-zero.
    const bool diag = ((lane & 15) >> 2) == (lane >> 4);
    const int q = lane & 3;                               // X: output column within the lane's four; Y: output row within its four
    {   // Y pass: output row q = 2*kk + par reads window rows rho = 2 + kk + par + tap_off(t) (rows a-6 .. a+1 = rho 0..7); the
        // window is a ring of four row PAIRS and iteration u (mod 4) has pair p in slot (u + 1 + p) & 3, so the weights rotate
        // with u: one fragment per u, built by wave u
        const int u = wave, kk = q >> 1, par = q & 1;
        float wsel[NT];                                  // this row's phase weights: selects between kernel arguments (SGPRs)
#pragma unroll
        for (int tt = 0; tt < NT; tt++) { const float a0 = P.we[tt], a1 = P.wo[tt]; wsel[tt] = par ? a1 : a0; }
        float w[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int rho = 2 * (((t >> 1) - u - 1) & 3) + (t & 1);
            float s = 0.0f;
#pragma unroll
            for (int tt = 0; tt < NT; tt++) s += (2 + kk + par + tap_off<NT>(tt) == rho) ? wsel[tt] : 0.0f;
            w[t] = diag ? s * SY : 0.0f;
        }
        h8 hi, lo;
        split8(w, hi, lo);
        WY[(u * 2 + 0) * 64 + lane] = hi;
        WY[(u * 2 + 1) * 64 + lane] = lo;
    }
    __syncthreads();                                      // the only workgroup barrier: tables visible

    const int W = P.W, H = P.H;
    const int x0 = (blockIdx.x * WAVES + wave) * S;
    const int s0 = blockIdx.y * P.seg_rows;
    if (x0 >= W || s0 >= H) return;
    const int s1 = min(s0 + P.seg_rows, H);
    _Float16 *AH = AHall + wave * MX_A_HALFS;

    // X pass weights: output column q = 2*ko + odd of the lane reads sources k0 - 3 + t, t = 3 + ko - (odd ? 0 : 1) + tap_off(tt)
    h8 axh, axl;
    {
        const int ko = q >> 1, odd = q & 1;
        float wsel[NT];
#pragma unroll
        for (int tt = 0; tt < NT; tt++) { const float a0 = P.we[tt], a1 = P.wo[tt]; wsel[tt] = odd ? a1 : a0; }
        float w[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            float s = 0.0f;
#pragma unroll
            for (int tt = 0; tt < NT; tt++) s += (3 + ko - (odd ? 0 : 1) + tap_off<NT>(tt) == t) ? wsel[tt] : 0.0f;
            w[t] = diag ? s * (SX * 1024.0f * P.inv_maxv) : 0.0f;
        }
        split8(w, axh, axl);
    }

    const FusedFrame frame = frames ? frames[blockIdx.z] : single;
    auto uniform_ptr = [](const void *p) {
        const uint64_t v = (uint64_t)p;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const uint64_t src_u = uniform_ptr(frame.src), dst_u = uniform_ptr(frame.dst);
    const gcptr py = (gcptr)src_u;
    const gptr pdst = (gptr)dst_u;

    // stage C role: A columns 2*lane, 2*lane+1 = rect columns X, X+1; the block is fetched at Xg (inside the rect)
    const int X = x0 - 4 + 2 * lane;
    const int Xg = clampi(X, 0, W - 2);
    const bool edge_wave = x0 == 0 || x0 + 2 * 63 - 4 > W - 2;
    // stage X / Y role: output columns ox .. ox+3 (rect-relative); lanes 60..63 compute on a clamped window and store nothing
    const int lx = min(lane, 59);
    const int ox = 2 * x0 + 4 * lane;
    const bool store_ok = lane < 60 && ox < 2 * W;
    const int wx0 = P.off_x + ox;
    const bool st_aligned = (wx0 & 3) == 0 && (((uintptr_t)dst_u | (uintptr_t)P.dst_pitch) & 15) == 0;
    const uint32_t lane_off = (uint32_t)wx0 * 4u;

    const f2 MM[5] = {f2{P.m[0], P.m[1]}, f2{P.m[2], P.m[3]}, f2{P.m[4], P.m[5]}, f2{P.m[6], P.m[7]}, f2{P.m[8], 0.0f}};
    const f2 GG[5] = {f2{P.gamut[0], P.gamut[1]}, f2{P.gamut[2], P.gamut[3]}, f2{P.gamut[4], P.gamut[5]}, f2{P.gamut[6], P.gamut[7]}, f2{P.gamut[8], 0.0f}};
    constexpr bool FASTEPI = EPI == EPI_DITHER8;
    const f2 maxv2 = splat((FASTEPI || P.final_pass) ? P.maxv : P.quant);
    // convert output -> UNORM code k as k * 2^-10: x * maxv/1024 + 2^13 rounds to a multiple of 2^-10 (ulp of 2^13), i.e. to
    // rne(x * maxv) * 2^-10 — the store rounding of m_TexConvertOutput — and 2^13 comes off exactly
    const f2 cms2 = splat(P.maxv * (1.0f / 1024.0f));
    f2 b13 = splat(8192.0f), big2 = splat(8388608.0f), sdn2 = splat(1.0f / (SX * SY));
    asm volatile("" : "+v"(b13), "+v"(big2), "+v"(sdn2));
    f2 CC[3] = {splat(P.c[0]), splat(P.c[1]), splat(P.c[2])};
    asm volatile("" : "+v"(CC[0]), "+v"(CC[1]), "+v"(CC[2]));

    // ring of four row pairs per (output column, channel): packed fp16 (row a | row a+1 << 16) of result * 2^12
    uint32_t win[4][3][4];
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int sl = 0; sl < 4; sl++) win[o][c][sl] = 0u;

    const int n_iter = (s1 - s0 + 1) / 2 + 3;
    Raw raw;
    RawAddr ra;
    make_raw_addr<SRC>(P, Xg, ra);
    load_raw<SRC>(P, py, ra, clampi(s0 - 3, 0, H - 1), clampi(s0 - 2, 0, H - 1), raw);

    auto stage_c = [&](int ar) {
        f2 rc[2][3];
        if (!(P.dbg & 8)) convert_block<TAIL, SRC>(P, MM, GG, CC, raw, P.rect_t + clampi(ar, 0, H - 1), P.rect_t + clampi(ar + 1, 0, H - 1), T, rc);
        else { for (int i = 0; i < 2; i++) for (int c = 0; c < 3; c++) rc[i][c] = f2{__uint_as_float(raw.y[0] & 0x3f000000u), __uint_as_float(raw.c[0][1] & 0x3f000000u)}; }
        load_raw<SRC>(P, py, ra, clampi(ar + 2, 0, H - 1), clampi(ar + 3, 0, H - 1), raw);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            f2 qe = pk_fma(rc[0][c], cms2, b13) - b13;             // even column, rows (a, a+1): k * 2^-10
            f2 qo = pk_fma(rc[1][c], cms2, b13) - b13;             // odd column
            if (edge_wave) {                                        // clamp-to-edge of the convert texture (rare wave)
                if (X < 0) qo = qe;
                else if (X > W - 2) qe = qo;
            }
            const h2v r0 = __builtin_convertvector(f2{qe.x, qo.x}, h2v), r1 = __builtin_convertvector(f2{qe.y, qo.y}, h2v);
            // four 2-byte stores: the pair (index 2l+1, 2l+2) straddles a dword, and a dword store off its natural alignment is
            // replayed by the LDS (measured: SQ_WAIT_INST_LDS 43 % of the wave cycles when the compiler merged them)
            typedef volatile __attribute__((address_space(3))) _Float16 *lds_h;
            lds_h p0 = (lds_h)(AH + (c * 2 + 0) * HW + 2 * lane + 1), p1 = (lds_h)(AH + (c * 2 + 1) * HW + 2 * lane + 1);
            p0[0] = r0.x; p0[1] = r0.y;
            p1[0] = r1.x; p1[1] = r1.y;
        }
    };
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

            // ---------------- stage X: 6 (row, channel) operands, two MFMAs each ----------------
            if (!(P.dbg & 1)) {
                h8 bx[2][3];
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const uint32_t *p32 = (const uint32_t *)(AH + (c * 2 + r) * HW + 2 * lx + 2);     // fp16 index 2l+2 = A column 2l+1
                        bx[r][c] = __builtin_bit_cast(h8, u32x4{p32[0], p32[1], p32[2], p32[3]});
                    }
                f4v dx[2][3];
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int r = 0; r < 2; r++) dx[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axh, bx[r][c], f4v{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int r = 0; r < 2; r++) dx[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axl, bx[r][c], dx[r][c], 0, 0, 0);
                // m_TexResize is R16G16B16A16_FLOAT (:3155): the fp16 rounding IS the window's storage format
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int o = 0; o < 4; o++) win[o][c][u] = cvt_pk_f16(dx[0][c][o], dx[1][c][o]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---------------- stage C of the NEXT iteration ----------------
            if (t + 1 < n_iter) stage_c(a + 2);

            // ---------------- stage Y + final pass: output rows 2(a-3) .. 2(a-3)+3 ----------------
            // The MFMAs run for the WHOLE wave, outside any lane-divergent branch: a lane's results use the weight fragment held by
            // another lane (rows 4g+q of A live in lane 4g+q+16g), so no lane may skip the fragment load or the instruction.
            if (t >= 3) {
                const h8 ayh = WY[(u * 2 + 0) * 64 + lane], ayl = WY[(u * 2 + 1) * 64 + lane];
                const int wy0 = P.off_y + 2 * (a - 3);
                u32x4 djr[4];
                if (FASTEPI) {
#pragma unroll
                    for (int m = 0; m < 4; m++) djr[m] = *(const u32x4 *)(Di + ((wy0 + m) & 31) * 32 + (wx0 & 31));
                }
                uint32_t pk[4][4];                                    // [row][pixel]
#pragma unroll
              for (int hp = 0; hp < 4; hp++) {                        // one output column at a time: 3 accumulators live
                f4v dy[4][3];
                if (P.dbg & 2) { for (int c = 0; c < 3; c++) dy[hp][c] = f4v{(float)win[hp][c][0], (float)win[hp][c][1], (float)win[hp][c][2], (float)win[hp][c][3]}; } else {
#pragma unroll
                for (int o = hp; o < hp + 1; o++)
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        dy[o][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ayh, __builtin_bit_cast(h8, u32x4{win[o][c][0], win[o][c][1], win[o][c][2], win[o][c][3]}),
                                                                         f4v{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
#pragma unroll
                for (int o = hp; o < hp + 1; o++)
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        dy[o][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ayl, __builtin_bit_cast(h8, u32x4{win[o][c][0], win[o][c][1], win[o][c][2], win[o][c][3]}),
                                                                         dy[o][c], 0, 0, 0);
                // the epilogue's first readers are inline asm, for which hipcc pads no MFMA -> VALU wait states: 8 of them here,
                // tied to the accumulators so that the statement stays between the MFMAs and their readers
                asm volatile("s_nop 7" : "+v"(dy[hp][0]), "+v"(dy[hp][1]), "+v"(dy[hp][2]));
                }
                if (store_ok && !(P.dbg & 4)) {
#pragma unroll
                    for (int o = hp; o < hp + 1; o++) {
                        if (P.dbg & 32) { for (int m = 0; m < 4; m++) pk[m][o] = __float_as_uint(dy[o][m & 1][m]); }
                        else if (FASTEPI || EPI == EPI_DIRECT8) {
                            // saturate (the shader's output lands in a UNORM texture), then x*maxv + 2^23 leaves the code in the
                            // low mantissa bits; integer final pass as in k_fused_up2x
                            f2 uq[3][2];
#pragma unroll
                            for (int c = 0; c < 3; c++)
#pragma unroll
                                for (int h = 0; h < 2; h++)
                                    uq[c][h] = pk_fma(pk_mul_clamp(f2{dy[o][c][2 * h], dy[o][c][2 * h + 1]}, sdn2), maxv2, big2);
#pragma unroll
                            for (int m = 0; m < 4; m++) {
                                const uint32_t ur = __float_as_uint(uq[0][m >> 1][m & 1]), ug = __float_as_uint(uq[1][m >> 1][m & 1]),
                                               ub = __float_as_uint(uq[2][m >> 1][m & 1]);
                                if (FASTEPI) {
                                    const uint32_t dj = djr[m][o];
                                    const uint32_t ib = __umul24(ub, P.epi_mul) + dj, ig = __umul24(ug, P.epi_mul) + dj, ir = __umul24(ur, P.epi_mul) + dj;
                                    const uint32_t bg = __builtin_amdgcn_perm(ig, ib, 0x0c0c0703u);    // [B, G, 0, 0]
                                    pk[m][o] = __builtin_amdgcn_perm(ir, bg, 0x0d070100u);             // [B, G, R, 0xff]
                                } else {
                                    const uint32_t bg = __builtin_amdgcn_perm(ug, ub, 0x0c0c0400u);
                                    pk[m][o] = __builtin_amdgcn_perm(ur, bg, 0x0d040100u);
                                }
                            }
                        } else {
                            // generic epilogue: no final pass (straight UNORM store into the RT) and/or R10G10B10A2 target
#pragma unroll
                            for (int m = 0; m < 4; m++) {
                                const int wy = wy0 + m;
                                float c3[3];
#pragma unroll
                                for (int c = 0; c < 3; c++) {
                                    const float x = saturate(dy[o][c][m] * (1.0f / (SX * SY)));
                                    const float qv = floorf(fmaf(x, P.final_pass ? P.maxv : P.quant, 0.5f));
                                    float v = qv;
                                    if (P.final_pass) {
                                        const float d = __half2float(__ushort_as_half(D[(wy & 31) * 32 + ((wx0 + o) & 31)]));
                                        v = fminf(fmaxf(floorf(fmaf(qv, P.q_over_maxv, d)), 0.0f), P.quant);
                                    }
                                    c3[c] = v;
                                }
                                pk[m][o] = P.out10 ? pack_rgb10a2(c3[0], c3[1], c3[2]) : pack_bgra8(c3[0], c3[1], c3[2]);
                            }
                        }
                    }
                }
              }   // hp
                if (P.dbg & 16) { uint32_t acc = 0; for (int m = 0; m < 4; m++) for (int o = 0; o < 4; o++) acc ^= pk[m][o]; if (acc == 0x9e3779b9u && store_ok) *(__attribute__((address_space(1))) uint32_t *)(pdst) = acc; }
                else if (store_ok && !(P.dbg & 4)) {
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        const gptr rowp = pdst + (uint32_t)(wy0 + m) * (uint32_t)P.dst_pitch;
                        if (EPI != EPI_GENERIC || st_aligned) {
                            *(__attribute__((address_space(1))) u32x4 *)(rowp + opaque(lane_off)) = u32x4{pk[m][0], pk[m][1], pk[m][2], pk[m][3]};
                        } else {
                            __attribute__((address_space(1))) uint32_t *dst = (__attribute__((address_space(1))) uint32_t *)(rowp + lane_off);
                            dst[0] = pk[m][0]; dst[1] = pk[m][1]; dst[2] = pk[m][2]; dst[3] = pk[m][3];
                        }
                    }
                }
            }   // t >= 3
        }   // u
    }   // tb
}

}  // namespace

hipError_t LaunchFusedUp2xMx(const FusedParams &P, const FusedArgs &a_in, int knt, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s)
{
    FusedArgs a = a_in;
    static const int dbg = EnvInt("MPCVR_MX_DBG", 0);       // ablation experiments (tools/prof_headline.sh): 1 no X stage, 2 no Y MFMAs,
    a.dbg = dbg;                                             // 4 no epilogue / stores, 8 no convert arithmetic
    const ConvertParams &c = P.conv;
    const int strips = (c.out_w + S - 1) / S;
    const int seg = a.seg_rows;
    const dim3 grid((strips + WAVES - 1) / WAVES, (c.out_h + seg - 1) / seg, n_frames);
    const dim3 block(256, 1, 1);
    const int tailk = FusedTailKind(P), srck = FusedSourceKind(P);
    static const int lds_pad = EnvInt("MPCVR_FUSED_LDS_PAD", 0);
    const size_t lds = MX_LDS_A + MX_LDS_WY + LDS_D + LDS_DB + (tail_has_table(tailk) ? LDS_T : 0) + (size_t)lds_pad;
    const bool aligned = P.dst_aligned16 && (a.off_x & 3) == 0 && (a.dst_pitch & 15) == 0;
    const int epik = !aligned || a.out10 ? EPI_GENERIC
                   : (a.final_pass && a.epi_mul != 0) ? EPI_DITHER8
                   : (!a.final_pass && P.store.dst_fmt == SF_BGRA8 && P.store.quant == 255) ? EPI_DIRECT8 : EPI_GENERIC;
    // Instantiated: what the experiment was measured on and the suite launches — 4 and 5 taps (Mitchell / Catmull-Rom, Lanczos3 as
    // Direct3D 11 draws it); SDR content from any source layout (the generic and the P01x loaders) with both epilogues; the HDR tails
    // (PQ table with both epilogues; HLG and the literal chains with the integer final pass) on P01x.  Everything else — 6 taps, an
    // R10G10B10A2 target, NV12's own loader — answers hipErrorNotSupported and runs on the packed-fp32 kernel (LaunchFusedUp2x).
    // (Round 3 built all 72 combinations of a kernel nobody selects by default; 56 of them were never launched by any test.)
    if (knt != 4 && knt != 5) return hipErrorNotSupported;
    // the convert stage that rounds like the reference's (an 8-bit internal format, vp_fused_dev.h: exact_capable) exists in the packed-fp32
    // kernels only: this one steps aside instead of drawing such a frame in the fast form
    if (a.exact_cv) return hipErrorNotSupported;
    const int sk = srck == SRC_P01X ? SRC_P01X : SRC_GENERIC, ek = epik == EPI_DITHER8 ? EPI_DITHER8 : EPI_GENERIC;
    if (tailk != TAILK_NONE && sk != SRC_P01X) return hipErrorNotSupported;
    if ((tailk == TAILK_HLG || tailk == TAILK_ALU) && ek != EPI_DITHER8) return hipErrorNotSupported;
#define MPCVR_LAUNCH3(NT, TK, SK, EK) hipLaunchKernelGGL((k_fused_up2x_mx<NT, TK, SK, EK>), grid, block, lds, s, a, frames_dev, single)
#define MPCVR_LAUNCH_NT(NT) do { \
        if (tailk == TAILK_NONE) { \
            if (sk == SRC_P01X) { if (ek == EPI_DITHER8) MPCVR_LAUNCH3(NT, TAILK_NONE, SRC_P01X, EPI_DITHER8); else MPCVR_LAUNCH3(NT, TAILK_NONE, SRC_P01X, EPI_GENERIC); } \
            else { if (ek == EPI_DITHER8) MPCVR_LAUNCH3(NT, TAILK_NONE, SRC_GENERIC, EPI_DITHER8); else MPCVR_LAUNCH3(NT, TAILK_NONE, SRC_GENERIC, EPI_GENERIC); } \
        } else if (tailk == TAILK_PQ_LUT) { if (ek == EPI_DITHER8) MPCVR_LAUNCH3(NT, TAILK_PQ_LUT, SRC_P01X, EPI_DITHER8); else MPCVR_LAUNCH3(NT, TAILK_PQ_LUT, SRC_P01X, EPI_GENERIC); } \
        else if (tailk == TAILK_HLG) MPCVR_LAUNCH3(NT, TAILK_HLG, SRC_P01X, EPI_DITHER8); \
        else MPCVR_LAUNCH3(NT, TAILK_ALU, SRC_P01X, EPI_DITHER8); } while (0)
    if (knt == 4) MPCVR_LAUNCH_NT(4);
    else MPCVR_LAUNCH_NT(5);
#undef MPCVR_LAUNCH_NT
#undef MPCVR_LAUNCH3
    return hipGetLastError();
}

}  // namespace mpcvr
