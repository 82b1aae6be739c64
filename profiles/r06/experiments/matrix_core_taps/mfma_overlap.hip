// tools/ubench/mfma_overlap.hip — does the matrix pipe run beside the NON-packed remainder of the fused kernels?
// (round-2 review, item 3d: mfma_owncol.hip measured the overlap only against v_pk_fma_f32 and v_fmac_f32.)
//
// One wave issues NM MFMAs and NV filler instructions per iteration, interleaved one MFMA : NV/NM fillers, three waves per SIMD
// like the headline kernel (768 workgroups of 256 threads on 256 CUs).  Timed: MFMA alone (m), filler alone (v), both (b).
// overlap = (m + v - b) / min(m, v): 1 = the shorter stream is free, 0 = the two pipes take turns.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_overlap mfma_overlap.hip && ./mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum Filler { F_PK_FMA, F_FMAC, F_FMA_SGPR, F_MAD_U24, F_PERM, F_CVT_UBYTE, F_CVT_PK_U8, F_CVT_F16, F_LOG, F_EXP, F_LSHL_OR, F_DS_READ, F_COUNT };
static const char *kFillerName[F_COUNT] = {"v_pk_fma_f32", "v_fmac_f32 (VOP2)", "v_fma_f32 with an SGPR", "v_mad_u32_u24", "v_perm_b32", "v_cvt_f32_ubyte0",
                                           "v_cvt_pk_u8_f32", "v_cvt_f16_f32 + v_cvt_f32_f16", "v_log_f32", "v_exp_f32", "v_lshl_or_b32", "ds_read_b64"};

template <int KIND>
__device__ __forceinline__ void filler(f2 (&p)[16], uint32_t (&u)[16], int j, float s, uint32_t lds_addr)
{
    const int a = j & 15, b = (j + 5) & 15, c = (j + 11) & 15;
    if (KIND == F_PK_FMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[a]) : "v"(p[b]), "v"(p[c]));
    else if (KIND == F_FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(p[a].x) : "v"(p[b].y), "v"(p[c].y));
    else if (KIND == F_FMA_SGPR) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[a].x) : "v"(p[b].y), "s"(s));
    else if (KIND == F_MAD_U24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(u[a]) : "v"(u[b]), "v"(u[c]));
    else if (KIND == F_PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]), "v"(0x07050301u));
    else if (KIND == F_CVT_UBYTE) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(p[a].x) : "v"(u[b]));
    else if (KIND == F_CVT_PK_U8) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[a]) : "v"(p[b].x));
    else if (KIND == F_CVT_F16) { uint32_t t; asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(t) : "v"(p[b].x)); asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(p[a].y) : "v"(t)); }
    else if (KIND == F_LOG) asm volatile("v_log_f32 %0, %1" : "=v"(p[a].x) : "v"(p[b].y));
    else if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %1" : "=v"(p[a].x) : "v"(p[b].y));
    else if (KIND == F_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %1, 10, %2" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]));
    else if (KIND == F_DS_READ) asm volatile("ds_read_b64 %0, %1" : "=v"(p[a]) : "v"(lds_addr));
}

// MK: 0 = v_mfma_f32_16x16x32_f16 (8 passes: the own-column prototype's), 1 = v_mfma_f32_4x4x1_16b_f32 (D rows = 4 output rows of one
// source row's outer product with a weight column: the layout a bit-exact fp32 Y stage on the matrix pipe would use)
template <int MK, int NM, int NV, int KIND>
__global__ __launch_bounds__(256) void k_overlap(float *out, int iters, float s)
{
    __shared__ float lds[1024];
    const int l = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = (float)i;
    __syncthreads();
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(l * 0.001f + i); b[i] = (_Float16)(i * 0.5f - l * 0.002f); }
    float af = l * 0.25f, bf = 1.0f - l * 0.001f;
    f4v acc[8];
    for (int i = 0; i < 8; i++) acc[i] = f4v{0, 0, 0, 0};
    f2 p[16]; uint32_t u[16];
    for (int i = 0; i < 16; i++) { p[i] = f2{1.0f + l * 0.01f, 0.5f + i * 0.01f}; u[i] = (uint32_t)(l * 37 + i * 101) & 0xFFFFFF; }
    const uint32_t lds_addr = (uint32_t)(uintptr_t)lds + (uint32_t)l * 8;
    constexpr int PER = NM ? (NV + NM - 1) / NM : 0;
    for (int it = 0; it < iters; it++) {
        if (NM == 0) {
#pragma unroll
            for (int j = 0; j < NV; j++) filler<KIND>(p, u, j, s, lds_addr);
        } else {
#pragma unroll
            for (int m = 0; m < NM; m++) {
                if (MK == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(af), "v"(bf));
#pragma unroll
                for (int j = 0; j < PER; j++) if (m * PER + j < NV) filler<KIND>(p, u, m * PER + j, s, lds_addr);
            }
        }
        if (KIND == F_DS_READ) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r = 0;
    for (int m = 0; m < 8; m++) for (int i = 0; i < 4; i++) r += acc[m][i];
    for (int i = 0; i < 16; i++) r += p[i].x + p[i].y + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MK, int NM, int NV, int KIND>
static double run(float *d_out, int blocks)
{
    const int iters = 3000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_overlap<MK, NM, NV, KIND>), dim3(blocks), dim3(256), 0, 0, d_out, 10, 0.999f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_overlap<MK, NM, NV, KIND>), dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters / (blocks / 256.0);                  // ns of SIMD time per wave-iteration
}

template <int MK, int NM, int NV, int KIND>
static void row(float *d_out, int blocks)
{
    const double m = run<MK, NM, 0, KIND>(d_out, blocks), v = run<MK, 0, NV, KIND>(d_out, blocks), b = run<MK, NM, NV, KIND>(d_out, blocks);
    printf("  %-32s %3d x  alone %7.1f   beside %2d MFMA (%6.1f alone) %7.1f   sum %7.1f   overlap %5.2f   ns per filler alone %.2f\n",
           kFillerName[KIND], NV, v, NM, m, b, m + v, (m + v - b) / fmin(m, v), v / NV);
}

template <int MK, int NM>
static void table(float *d_out, int blocks)
{
    row<MK, NM, 72, F_PK_FMA>(d_out, blocks);
    row<MK, NM, 144, F_FMAC>(d_out, blocks);
    row<MK, NM, 72, F_FMA_SGPR>(d_out, blocks);
    row<MK, NM, 72, F_MAD_U24>(d_out, blocks);
    row<MK, NM, 72, F_PERM>(d_out, blocks);
    row<MK, NM, 72, F_CVT_UBYTE>(d_out, blocks);
    row<MK, NM, 72, F_CVT_PK_U8>(d_out, blocks);
    row<MK, NM, 36, F_CVT_F16>(d_out, blocks);
    row<MK, NM, 36, F_LOG>(d_out, blocks);
    row<MK, NM, 36, F_EXP>(d_out, blocks);
    row<MK, NM, 72, F_LSHL_OR>(d_out, blocks);
    row<MK, NM, 72, F_DS_READ>(d_out, blocks);
}

int main()
{
    float *d_out;
    CK(hipMalloc(&d_out, 4096 * 256 * 4));
    for (int blocks : {768, 256}) {
        printf("v_mfma_f32_16x16x32_f16 (36 per iteration) beside each instruction class; %d waves per SIMD; ns of SIMD time per wave-iteration\n", blocks / 256);
        table<0, 36>(d_out, blocks);
        printf("v_mfma_f32_4x4x1_16b_f32 (72 per iteration) beside each instruction class; %d waves per SIMD\n", blocks / 256);
        table<1, 72>(d_out, blocks);
    }
    return 0;
}
