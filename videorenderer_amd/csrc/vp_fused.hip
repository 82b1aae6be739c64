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
// Design: WAVE-AUTONOMOUS STRIPS WITH A REGISTER-RESIDENT VERTICAL WINDOW.
//   Exact 2x => two fixed phases per axis (t = 0.75 for even outputs, base = k-1; t = 0.25 for odd, base = k).
//   One wavefront owns a strip of S = 120 source columns (240 output columns = 60 lanes x 4 px = one
//   16-byte store per lane and output row) and marches down a segment of source rows, two rows per
//   iteration, with no workgroup barrier inside the loop:
//     stage C  2 rows x 128 px (4-px halo each side): 4 px per lane from raw codes prefetched one iteration
//              ahead -> this wave's LDS slice A (fp32, already rounded to the internal UNORM format)
//     stage X  lane l reads A columns 2l..2l+9 and produces the 4 output columns it owns for both new rows;
//              the fp16-rounded results (m_TexResize) go into an 8-row register window — no LDS, no HBM
//     stage Y  from the window: 4 output rows x 4 px per lane, UNORM rounding (m_TexsPostScale), dither, store
//   The four waves of a workgroup share only the read-only tables (dither, PQ->SDR LUT).
//   Recomputed: the horizontal halo (8 of 128 columns) and 6 rows per segment.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "vp_device.h"
#include "vp_launch.h"

namespace mpcvr {

namespace {

constexpr int S = 120;             // source pixels per strip
constexpr int AW = 128;            // LDS A row width: rect columns x0-4 .. x0+123
constexpr int WAVES = 4;           // strips per workgroup
constexpr int A_FLOATS = 2 * 3 * AW;
constexpr int LUT_N = kPqLutSize;  // PQ->SDR per-channel table (vp_params.h)
constexpr int LDS_A = WAVES * A_FLOATS * 4;
constexpr int LDS_D = 32 * 32 * 2;
constexpr int LDS_T = (LUT_N + 4) * 4;   // + a duplicated last entry (and padding)

typedef const __attribute__((address_space(1))) uint8_t *gcptr;
typedef __attribute__((address_space(1))) uint8_t *gptr;

enum { TAILK_NONE = 0, TAILK_PQ_LUT = 1, TAILK_ALU = 2 };

// everything the kernel needs, flattened (kernel argument => SGPRs)
struct FusedArgs {
    size_t off_u, off_v;           // byte offsets of the chroma plane(s) inside a sample (u = interleaved UV when biplanar)
    int pitch_y, pitch_c;
    int tex_w, cw, ch;             // luma width, chroma size
    int rect_l, rect_t, W, H;      // source rect origin and size (== convert-output size)
    int bytes, planes;
    int center_h;                  // MPEG-1 siting: chroma sample centred between luma columns
    float v_off;                   // +0.25 chroma rows for co-sited
    float m[9], c[3];              // colour matrix with the UNORM scale (and CopyPlane10to16 shift) folded in
    int tail; float gamma, lum_scale;
    float gamut[9];
    const float *lut;              // LUT_N floats (device) for TAILK_PQ_LUT
    float maxv, inv_maxv;          // internal UNORM format
    float q_over_maxv;             // ps_final_pass QUANTIZATION / maxv
    float we[6], wo[6];            // phase weights (even/odd outputs); Q1-merged on the host side of the launch
    int dst_pitch, off_x, off_y;
    int final_pass, out10;
    float quant;
    const uint16_t *dither;
    int seg_rows;
};

// tap offsets relative to `base` (ps_interpolation_*.hlsl); with the D3D11 Lanczos3 quirk Q1 the second
// tap re-reads the first tap's texel (ps_interpolation_lanczos3.hlsl:33-34): the launcher folds its weight
// into tap 0 and zeroes it, so the kernel can keep the regular offsets.
// NT = 5 is that case: taps {-2, 0, 1, 2, 3} with the first weight = w0 + w1 (folded by the launcher).
template <int NT>
__host__ __device__ constexpr int tap_off(int t) { return NT == 4 ? (t - 1) : NT == 6 ? (t - 2) : (t == 0 ? -2 : t - 1); }

// exact q/maxv (correctly rounded like the UNORM->float load) from the reciprocal: one Newton step
__device__ __forceinline__ float unorm_to_float(float q, float maxv, float inv)
{
    const float r0 = q * inv;
    const float e = fmaf(-r0, maxv, q);
    return fmaf(e, inv, r0);
}

__device__ __forceinline__ float lut_eval(const float *T, float x)
{
    const float t = saturate(x) * (float)(LUT_N - 1);      // T has LUT_N + 1 entries (last one duplicated)
    const int i = (int)t;
    const float fr = __builtin_amdgcn_fractf(t);
    const float a = T[i], b = T[i + 1];
    return fmaf(b - a, fr, a);
}

// raw codes of one 4-pixel group of one source row, prefetched one iteration ahead
struct Raw {
    uint32_t y0, y1;         // luma: 2 dwords (16-bit) or y0 only (8-bit: 4 bytes)
    uint32_t c[2][4];        // chroma rows r0/r1, columns c0-1..c0+2: packed (U | V<<16) codes
};

__device__ __forceinline__ uint32_t ld_u8(gcptr p) { return *p; }
__device__ __forceinline__ uint32_t ld_u16(gcptr p) { return *(const __attribute__((address_space(1))) uint16_t *)p; }
__device__ __forceinline__ uint32_t ld_u32(gcptr p) { return *(const __attribute__((address_space(1))) uint32_t *)p; }

// chroma texel (col,row) as U | V << 16 (raw codes), clamp addressing on the column
__device__ __forceinline__ uint32_t ld_uv(const FusedArgs &P, gcptr pu, gcptr pv, int col, int row)
{
    col = clampi(col, 0, P.cw - 1);
    const size_t ro = (size_t)row * P.pitch_c;
    if (P.planes == 2) {
        if (P.bytes == 2) return ld_u32(pu + ro + 4 * col);
        const uint32_t d = ld_u16(pu + ro + 2 * col);
        return (d & 0xffu) | ((d >> 8) << 16);
    }
    if (P.bytes == 2) return ld_u16(pu + ro + 2 * col) | (ld_u16(pv + ro + 2 * col) << 16);
    return ld_u8(pu + ro + col) | (ld_u8(pv + ro + col) << 16);
}

// Xg: first rect column of the group actually fetched (== X for interior groups; clamped at the rect edges)
__device__ __forceinline__ void load_raw(const FusedArgs &P, gcptr py, gcptr pu, gcptr pv, int Xg, int y, Raw &r)
{
    const int sx0 = P.rect_l + Xg, sy = P.rect_t + y;
    const gcptr ry = py + (size_t)sy * P.pitch_y;
    if (P.bytes == 2) {
        const int i0 = sx0 >> 1, i1 = min(i0 + 1, (P.tex_w >> 1) - 1);
        r.y0 = ld_u32(ry + 4 * i0);
        r.y1 = ld_u32(ry + 4 * i1);
    } else {
        r.y0 = ld_u32(ry + sx0);          // 4 bytes; sx0 % 4 == 0 and pitch % 4 == 0 (checked on the host)
        r.y1 = 0;
    }
    // vertical chroma position (Shaders.cpp:118-138): v' = (sy+0.5)/2 [+0.25 co-sited] - 0.5
    const float fv = ((float)sy + 0.5f) * 0.5f + P.v_off - 0.5f;
    const int iv = (int)floorf(fv);
    const int r0 = clampi(iv, 0, P.ch - 1), r1 = clampi(iv + 1, 0, P.ch - 1);
    const int c0 = sx0 >> 1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i == 0 && !P.center_h) { r.c[0][0] = r.c[1][0] = 0; continue; }
        r.c[0][i] = ld_uv(P, pu, pv, c0 - 1 + i, r0);
        r.c[1][i] = ld_uv(P, pu, pv, c0 - 1 + i, r1);
    }
}

// 4:2:0 bilinear chroma + matrix for the 4 pixels of a group (ShaderGetPixels' CHROMA_Bilinear branch,
// Shaders.cpp:265-270,319-325): same sample positions and weights, evaluated in code units
// (vertical lerp first), UNORM scale folded into the matrix.
template <int TAIL>
__device__ __forceinline__ void convert4(const FusedArgs &P, const Raw &r, int sy, const float *T, f3 out[4])
{
    float Y[4];
    if (P.bytes == 2) {
        Y[0] = (float)(r.y0 & 0xffffu); Y[1] = (float)(r.y0 >> 16); Y[2] = (float)(r.y1 & 0xffffu); Y[3] = (float)(r.y1 >> 16);
    } else {
        Y[0] = (float)(r.y0 & 0xffu); Y[1] = (float)((r.y0 >> 8) & 0xffu); Y[2] = (float)((r.y0 >> 16) & 0xffu); Y[3] = (float)(r.y0 >> 24);
    }
    const float fv = ((float)sy + 0.5f) * 0.5f + P.v_off - 0.5f;
    const float wy = fv - floorf(fv), wy0 = 1.0f - wy;
    float U[4], V[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        U[i] = fmaf((float)(r.c[1][i] & 0xffffu), wy, (float)(r.c[0][i] & 0xffffu) * wy0);
        V[i] = fmaf((float)(r.c[1][i] >> 16), wy, (float)(r.c[0][i] >> 16) * wy0);
    }
    float Ue[4], Ve[4];
    if (P.center_h) {       // u' = sx/2 - 0.25
        Ue[0] = fmaf(U[1], 0.75f, U[0] * 0.25f); Ve[0] = fmaf(V[1], 0.75f, V[0] * 0.25f);
        Ue[1] = fmaf(U[2], 0.25f, U[1] * 0.75f); Ve[1] = fmaf(V[2], 0.25f, V[1] * 0.75f);
        Ue[2] = fmaf(U[2], 0.75f, U[1] * 0.25f); Ve[2] = fmaf(V[2], 0.75f, V[1] * 0.25f);
        Ue[3] = fmaf(U[3], 0.25f, U[2] * 0.75f); Ve[3] = fmaf(V[3], 0.25f, V[2] * 0.75f);
    } else {                // u' = sx/2
        Ue[0] = U[1];                          Ve[0] = V[1];
        Ue[1] = fmaf(U[2], 0.5f, U[1] * 0.5f); Ve[1] = fmaf(V[2], 0.5f, V[1] * 0.5f);
        Ue[2] = U[2];                          Ve[2] = V[2];
        Ue[3] = fmaf(U[3], 0.5f, U[2] * 0.5f); Ve[3] = fmaf(V[3], 0.5f, V[2] * 0.5f);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        f3 v;
        v.x = fmaf(P.m[0], Y[e], fmaf(P.m[1], Ue[e], fmaf(P.m[2], Ve[e], P.c[0])));
        v.y = fmaf(P.m[3], Y[e], fmaf(P.m[4], Ue[e], fmaf(P.m[5], Ve[e], P.c[1])));
        v.z = fmaf(P.m[6], Y[e], fmaf(P.m[7], Ue[e], fmaf(P.m[8], Ve[e], P.c[2])));
        if (TAIL == TAILK_PQ_LUT) {
            // Shaders.cpp:870-923: per-channel saturate -> ST2084ToLinear*scale -> Hable/hable(4.8) from the LDS
            // table, then the 2020->709 matrix, saturate and pow 1/2.2 in ALU
            const float a = lut_eval(T, v.x), b = lut_eval(T, v.y), c = lut_eval(T, v.z);
            v.x = fmaf(P.gamut[0], a, fmaf(P.gamut[1], b, P.gamut[2] * c));
            v.y = fmaf(P.gamut[3], a, fmaf(P.gamut[4], b, P.gamut[5] * c));
            v.z = fmaf(P.gamut[6], a, fmaf(P.gamut[7], b, P.gamut[8] * c));
            v.x = hlsl_pow(saturate(v.x), 1.0f / 2.2f);
            v.y = hlsl_pow(saturate(v.y), 1.0f / 2.2f);
            v.z = hlsl_pow(saturate(v.z), 1.0f / 2.2f);
        } else if (TAIL == TAILK_ALU) {
            v = hdr_tail(v, P.tail, P.gamma, P.lum_scale, P.gamut);
        }
        out[e] = v;
    }
}

__device__ __forceinline__ float sel4(const float v[4], int i)
{
    return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3];
}

__device__ __forceinline__ uint32_t pack_h2(float a, float b)
{
    return (uint32_t)__half_as_ushort(__float2half_rn(a)) | ((uint32_t)__half_as_ushort(__float2half_rn(b)) << 16);
}
__device__ __forceinline__ float h_lo(uint32_t u) { return __half2float(__ushort_as_half((unsigned short)(u & 0xffffu))); }
__device__ __forceinline__ float h_hi(uint32_t u) { return __half2float(__ushort_as_half((unsigned short)(u >> 16))); }

// acc' = fp32(half(h2.lo|hi)) * w + acc as ONE v_fma_mix_f32 (the fp16 operand is converted exactly inside the
// instruction, so this equals cvt + fma bit for bit); CLAMP saturates the result to [0,1] for free.
// hipcc only forms v_fma_mix_f32 when the converted half has a single use; here every window value feeds an
// even and an odd output row, which otherwise costs a separate v_cvt_f32_f16 per value.
template <bool HI, bool FIRST, bool CLAMP>
__device__ __forceinline__ float mix_fma(uint32_t h2, float w, float acc)
{
    float r;
    if (FIRST) {
        if (HI) { if (CLAMP) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(h2), "s"(w));
                  else       asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "s"(w)); }
        else    { if (CLAMP) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(h2), "s"(w));
                  else       asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "s"(w)); }
    } else {
        if (HI) { if (CLAMP) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(h2), "s"(w), "v"(acc));
                  else       asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "s"(w), "v"(acc)); }
        else    { if (CLAMP) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(h2), "s"(w), "v"(acc));
                  else       asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "s"(w), "v"(acc)); }
    }
    return r;
}

// one vertical tap chain: sum over the NT window rows of weight * window value, saturated
template <int NT, int PX>
__device__ __forceinline__ float ytaps(const uint32_t (&win)[8][3][2], int c, const int (&slot)[6], const float (&w)[6])
{
    float acc = 0.0f;
#pragma unroll
    for (int tt = 0; tt < NT; tt++) {
        const uint32_t d = win[slot[tt]][c][PX >> 1];
        if (tt == 0) acc = mix_fma<(PX & 1) != 0, true, false>(d, w[0], 0.0f);
        else if (tt == NT - 1) acc = mix_fma<(PX & 1) != 0, false, true>(d, w[tt], acc);
        else acc = mix_fma<(PX & 1) != 0, false, false>(d, w[tt], acc);
    }
    return acc;
}

template <int NT, int TAIL>
__global__ __launch_bounds__(256, 3) void k_fused_up2x(FusedArgs P, const FusedFrame *__restrict__ frames, FusedFrame single)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Aall = (float *)smem;
    unsigned short *D = (unsigned short *)(smem + LDS_A);
    float *T = (float *)(smem + LDS_A + LDS_D);

    for (int i = threadIdx.x; i < 1024; i += 256) D[i] = P.dither[i];
    if (TAIL == TAILK_PQ_LUT)
        for (int i = threadIdx.x; i <= LUT_N; i += 256) T[i] = P.lut[min(i, LUT_N - 1)];
    __syncthreads();                                   // the only workgroup barrier: tables visible

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int W = P.W, H = P.H;
    const int x0 = (blockIdx.x * WAVES + wave) * S;
    const int s0 = blockIdx.y * P.seg_rows;
    if (x0 >= W || s0 >= H) return;
    const int s1 = min(s0 + P.seg_rows, H);
    float *A = Aall + wave * A_FLOATS;

    const FusedFrame frame = frames ? frames[blockIdx.z] : single;
    const gcptr py = (gcptr)frame.src;
    const gcptr pu = (gcptr)(frame.src + P.off_u);
    const gcptr pv = (gcptr)(frame.src + P.off_v);
    const gptr pdst = (gptr)frame.dst;

    // stage C role: row (0/1) of the pair and 4-px group; group start X in rect coordinates
    const int cr = lane >> 5, cj = lane & 31;
    const int X = x0 - 4 + 4 * cj;
    const int Xg = clampi(X, 0, (W - 1) & ~3);
    const bool interior = X >= 0 && X + 3 <= W - 1;
    // stage X / Y role: output columns ox .. ox+3 (rect-relative); lanes 60..63 idle there
    const bool xy_active = lane < 60;
    const int ox = 2 * x0 + 4 * lane;
    const bool store_ok = xy_active && ox < 2 * W;
    const int wx0 = P.off_x + ox;
    const bool d_aligned = (wx0 & 3) == 0;

    float we[6], wo[6];
#pragma unroll
    for (int t = 0; t < 6; t++) { we[t] = P.we[t]; wo[t] = P.wo[t]; }

    // 8-row window of X-pass results (fp16 x 4 px packed in 2 dwords) per channel
    uint32_t win[8][3][2];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int c = 0; c < 3; c++) win[i][c][0] = win[i][c][1] = 0;

    // iteration t adds virtual rows a, a+1 with a = s0 - 3 + 2t; from t = 3 on it emits output rows of k = a-3, a-2
    const int n_iter = (s1 - s0 + 1) / 2 + 3;
    Raw raw;
    load_raw(P, py, pu, pv, Xg, clampi(s0 - 3 + cr, 0, H - 1), raw);

    for (int tb = 0; tb < n_iter; tb += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = tb + u;
            if (t >= n_iter) break;
            const int a = s0 - 3 + 2 * t;

            // ---------------- stage C ----------------
            {
                const int y = clampi(a + cr, 0, H - 1);
                f3 v[4];
                convert4<TAIL>(P, raw, P.rect_t + y, T, v);
                // prefetch the next pair of rows while this one is processed
                load_raw(P, py, pu, pv, Xg, clampi(a + 2 + cr, 0, H - 1), raw);
                float q[3][4];
#pragma unroll
                for (int e = 0; e < 4; e++) {     // store to m_TexConvertOutput (UNORM) and read back
                    q[0][e] = unorm_q(v[e].x, P.maxv) * P.inv_maxv;       // q/maxv to within 1 ulp
                    q[1][e] = unorm_q(v[e].y, P.maxv) * P.inv_maxv;
                    q[2][e] = unorm_q(v[e].z, P.maxv) * P.inv_maxv;
                }
                if (!interior) {                  // clamp-to-edge of the convert texture: replicate its border pixel
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        float s[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) s[e] = sel4(q[c], clampi(X + e, 0, W - 1) - Xg);
#pragma unroll
                        for (int e = 0; e < 4; e++) q[c][e] = s[e];
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; c++)
                    *(float4 *)(A + (cr * 3 + c) * AW + 4 * cj) = make_float4(q[c][0], q[c][1], q[c][2], q[c][3]);
            }
            // A is exchanged between lanes of this wave only: LDS operations of one wave execute in order,
            // the fence keeps the compiler from moving the reads above the writes.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---------------- stage X ----------------
            // lane owns output columns 4l..4l+3 = sources k = 2l (e = 0,1), 2l+1 (e = 2,3); A column of source k is k+4.
            // av[i] = A column 2l+i, i = 0..9  =>  source k' = 2l + i - 4.
            if (xy_active) {
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float2 *ap = (const float2 *)(A + (rr * 3 + c) * AW + 2 * lane);
                        float av[10];
#pragma unroll
                        for (int i = 0; i < 5; i++) { const float2 p2 = ap[i]; av[2 * i] = p2.x; av[2 * i + 1] = p2.y; }
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int kk = e >> 1;                 // source k = 2l + kk  -> av index of k is kk + 4
                            const bool odd = e & 1;
                            float acc = 0.0f;
#pragma unroll
                            for (int tt = 0; tt < NT; tt++) {
                                // even output 2k: base = k-1; odd output 2k+1: base = k
                                const int idx = kk + 4 + (odd ? 0 : -1) + tap_off<NT>(tt);
                                const float w = odd ? wo[tt] : we[tt];
                                acc = tt == 0 ? w * av[idx] : fmaf(w, av[idx], acc);
                            }
                            o[e] = acc;
                        }
                        win[(2 * u + rr) & 7][c][0] = pack_h2(o[0], o[1]);     // m_TexResize is R16G16B16A16_FLOAT (:3155)
                        win[(2 * u + rr) & 7][c][1] = pack_h2(o[2], o[3]);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // ---------------- stage Y + final pass ----------------
            // window slot of virtual row r is (r - (s0-3)) & 7; rows a-6 .. a+1 are live: slot(a-6+i) = (2u+2+i) & 7
            if (t >= 3 && store_ok) {
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const int k = a - 3 + kk;                     // source row -> output rows 2k, 2k+1
                    if (k >= s1) break;
                    // even: base = k-1 -> row k-1+off = a-6 + (kk+2+off); odd: base = k -> a-6 + (kk+3+off)
                    int se[6], so[6];
#pragma unroll
                    for (int tt = 0; tt < 6; tt++) {
                        se[tt] = tt < NT ? ((2 * u + 2 + kk + 2 + tap_off<NT>(tt)) & 7) : 0;
                        so[tt] = tt < NT ? ((2 * u + 2 + kk + 3 + tap_off<NT>(tt)) & 7) : 0;
                    }
                    float res[2][4][3];            // saturated Y-pass results
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        res[0][0][c] = ytaps<NT, 0>(win, c, se, we); res[1][0][c] = ytaps<NT, 0>(win, c, so, wo);
                        res[0][1][c] = ytaps<NT, 1>(win, c, se, we); res[1][1][c] = ytaps<NT, 1>(win, c, so, wo);
                        res[0][2][c] = ytaps<NT, 2>(win, c, se, we); res[1][2][c] = ytaps<NT, 2>(win, c, so, wo);
                        res[0][3][c] = ytaps<NT, 3>(win, c, se, we); res[1][3][c] = ytaps<NT, 3>(win, c, so, wo);
                    }
#pragma unroll
                    for (int par = 0; par < 2; par++) {
                        const int wy = P.off_y + 2 * k + par;
                        float d4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (P.final_pass) {           // sampler WRAP+POINT: dither texel (wx mod 32, wy mod 32)
                            const unsigned short *drow = D + (wy & 31) * 32;
                            if (d_aligned) {
                                const uint2 dd = *(const uint2 *)(drow + (wx0 & 31));
                                d4[0] = h_lo(dd.x); d4[1] = h_hi(dd.x); d4[2] = h_lo(dd.y); d4[3] = h_hi(dd.y);
                            } else {
#pragma unroll
                                for (int px = 0; px < 4; px++) d4[px] = __half2float(__ushort_as_half(drow[(wx0 + px) & 31]));
                            }
                        }
                        uint32_t pk[4];
#pragma unroll
                        for (int px = 0; px < 4; px++) {
                            float c3[3];
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                // res is already saturated (clamp on the last tap).  m_TexsPostScale store/load:
                                // q = floor(x*maxv + 0.5), p = q/maxv; ps_final_pass.hlsl:29: floor(p*Q + d).
                                // p*Q is evaluated as q*(Q/maxv) inside one FMA (<= 1 ulp from the two-step form).
                                const float q = floorf(fmaf(res[par][px][c], P.final_pass ? P.maxv : P.quant, 0.5f));
                                c3[c] = P.final_pass ? floorf(fmaf(q, P.q_over_maxv, d4[px])) : q;
                            }
                            if (P.out10) pk[px] = pack_rgb10a2(c3[0], c3[1], c3[2]);
                            else {      // exact small integers: v_cvt_pk_u8_f32 converts and places a byte per instruction
                                uint32_t v = __builtin_amdgcn_cvt_pk_u8_f32(c3[2], 0, 0xff000000u);      // B
                                v = __builtin_amdgcn_cvt_pk_u8_f32(c3[1], 1, v);                         // G
                                pk[px] = __builtin_amdgcn_cvt_pk_u8_f32(c3[0], 2, v);                    // R
                            }
                        }
                        __attribute__((address_space(1))) uint32_t *dst =
                            (__attribute__((address_space(1))) uint32_t *)(pdst + (size_t)wy * P.dst_pitch) + wx0;
                        if ((((uintptr_t)dst) & 15) == 0) {
                            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                            u32x4 v4 = {pk[0], pk[1], pk[2], pk[3]};
                            *(__attribute__((address_space(1))) u32x4 *)dst = v4;
                        } else { dst[0] = pk[0]; dst[1] = pk[1]; dst[2] = pk[2]; dst[3] = pk[3]; }
                    }
                }
            }
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
    if (c.fmt.subsampling != 420 || c.chroma_scaling != 1) return false;
    if (c.out_w < 8 || c.out_h < 8 || (c.out_w & 1) || (c.out_h & 1)) return false;
    if (!P.fast_convert) return false;            // dword loads need aligned rows / rect (host-checked)
    return true;
}

hipError_t LaunchFusedUp2x(const FusedParams &P, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s)
{
    static const int seg_env = EnvInt("MPCVR_FUSED_SEG", 0);
    if (!frames_dev && n_frames != 1) return hipErrorInvalidValue;
    const ConvertParams &c = P.conv;
    FusedArgs a;
    std::memset(&a, 0, sizeof(a));
    const bool swap_uv = c.fmt.planes == 3 && c.fmt.v_first;
    a.off_u = swap_uv ? P.plane_off[2] : P.plane_off[1];
    a.off_v = swap_uv ? P.plane_off[1] : P.plane_off[2];
    a.pitch_y = c.pitch[0]; a.pitch_c = c.pitch[1];
    a.tex_w = c.tex_w; a.cw = c.cw; a.ch = c.ch;
    a.rect_l = c.rect_l; a.rect_t = c.rect_t; a.W = c.out_w; a.H = c.out_h;
    a.bytes = c.fmt.bytes; a.planes = c.fmt.planes;
    a.center_h = c.chroma_loc == CLOC_MPEG1;
    a.v_off = c.chroma_loc == CLOC_COSITED ? 0.25f : 0.0f;
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
    const int nt = P.wx.ntaps;
    for (int t = 0; t < 6; t++) { a.we[t] = P.wx.w_even[t]; a.wo[t] = P.wx.w_odd[t]; }
    int knt = nt;
    if (nt == 6 && P.wx.q1_quirk) {   // taps 0 and 1 read the same texel: fold tap 1's weight into tap 0 => 5 taps
        a.we[0] += a.we[1]; a.wo[0] += a.wo[1];
        for (int t = 1; t < 5; t++) { a.we[t] = a.we[t + 1]; a.wo[t] = a.wo[t + 1]; }
        a.we[5] = a.wo[5] = 0.0f;
        knt = 5;
    }
    a.dst_pitch = P.store.dst_pitch; a.off_x = P.store.off_x; a.off_y = P.store.off_y;
    a.final_pass = P.store.mode == ST_FINAL; a.out10 = P.store.dst_fmt == SF_RGB10A2;
    a.quant = (float)P.store.quant;
    a.dither = P.store.dither;
    int seg = seg_env > 0 ? seg_env : 72;
    seg = (seg + 1) & ~1;
    if (seg > c.out_h) seg = c.out_h;
    a.seg_rows = seg;

    const int strips = (c.out_w + S - 1) / S;
    const dim3 grid((strips + WAVES - 1) / WAVES, (c.out_h + seg - 1) / seg, n_frames);
    const dim3 block(256, 1, 1);
    const int tailk = c.tail == TAIL_NONE ? TAILK_NONE : (c.tail == TAIL_PQ_TO_SDR && P.pq_lut) ? TAILK_PQ_LUT : TAILK_ALU;
    const size_t lds = LDS_A + LDS_D + (tailk == TAILK_PQ_LUT ? LDS_T : 0);
#define MPCVR_LAUNCH(NT, TK) hipLaunchKernelGGL((k_fused_up2x<NT, TK>), grid, block, lds, s, a, frames_dev, single)
#define MPCVR_LAUNCH_NT(NT) \
    do { if (tailk == TAILK_NONE) MPCVR_LAUNCH(NT, TAILK_NONE); else if (tailk == TAILK_PQ_LUT) MPCVR_LAUNCH(NT, TAILK_PQ_LUT); \
         else MPCVR_LAUNCH(NT, TAILK_ALU); } while (0)
    if (knt == 4) MPCVR_LAUNCH_NT(4);
    else if (knt == 5) MPCVR_LAUNCH_NT(5);
    else MPCVR_LAUNCH_NT(6);
#undef MPCVR_LAUNCH_NT
#undef MPCVR_LAUNCH
    return hipGetLastError();
}

}  // namespace mpcvr
