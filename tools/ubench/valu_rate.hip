// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the instruction kinds
// the fused kernel leans on, at 1..4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s0, float s1)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4];
#pragma unroll
    for (int i = 0; i < 4; i++) p[i] = f2{a[2 * i], a[2 * i + 1]};
    unsigned h = __float_as_uint(a[0]);
    const f2 sp = f2{s0, s1};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(a[(i + 1) & 7]));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i & 3]) : "v"(p[(i + 1) & 3]));
                if (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(h), "s"(s0));
                if (KIND == 3) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 5) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 6) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(h) : "v"(a[i]));
                if (KIND == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(s1));
                if (KIND == 8) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(h));
                if (KIND == 9) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 10) asm volatile("v_med3_f32 %0, %0, 0, 1.0" : "+v"(a[i]));
                if (KIND == 11) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 12) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 13) asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(p[i & 3]) : "s"(sp));
                if (KIND == 14) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,0] op_sel_hi:[0,1]" : "+v"(p[i & 3]) : "s"(sp));
                if (KIND == 15) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 3]) : "v"(p[(i + 1) & 3]));
                if (KIND == 16) asm volatile("v_pk_fma_f32 %0, %1, %0, %0 clamp" : "+v"(p[i & 3]) : "s"(sp));
                if (KIND == 17) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(h));
                if (KIND == 18) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(h) : "v"(a[i]));
                if (KIND == 19) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(h));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(h);
}

template <int KIND> int run(const char *name, float *d)
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (int wps = 3; wps <= 3; wps++) {           // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
        const int blocks = cus * wps, iters = 2000;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.9999f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)iters * 64 * wps;      // per SIMD: wps waves x iters x 64 instr
        const double ns_per_instr = ms * 1e6 / instr_per_simd;
        printf("%-16s waves/SIMD=%d  %.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", name, wps, ns_per_instr, ns_per_instr * 2.4);
    }
    return 0;
}

int main()
{
    float *d; CHECK(hipMalloc(&d, 256 * 1024 * 16 * sizeof(float)));
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_fma_mix_f32", d); run<7>("v_mul_f32", d);
    run<3>("v_floor_f32", d); run<9>("v_fract_f32", d); run<10>("v_med3_f32", d); run<8>("v_cvt_f32_f16", d);
    run<11>("v_cvt_pk_f16_f32", d); run<6>("v_cvt_pk_u8_f32", d);
    run<4>("v_exp_f32", d); run<5>("v_log_f32", d); run<12>("v_rcp_f32", d);
    run<13>("pk_fma sgpr opsel", d); run<14>("pk_mul sgpr opsel", d); run<15>("v_pk_add_f32", d); run<16>("pk_fma sgpr clamp", d);
    run<17>("cvt_f32_f16 sdwa", d); run<18>("v_cvt_i32_f32", d); run<19>("cvt_f32_u32 sdwa", d);
    return 0;
}
