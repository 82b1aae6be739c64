// op_rate.hip — issue cost of individual gfx950 VALU instructions (wave64), many independent chains, 3 waves/SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o op_rate op_rate.hip && ./op_rate
// Result on MI355X (ns per wave-instruction per SIMD): see DESIGN.md "VALU cost model".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// OP(j): one instruction working on register set j (0..15); a[] = u32 regs, f[] = float regs, d[] = 64-bit pairs
#define KERNEL(NAME, BODY)                                                                          \
    __global__ __launch_bounds__(256) void k_##NAME(float *out, int iters, float s, uint32_t m)   \
    {                                                                                               \
        const int l = threadIdx.x;                                                                  \
        uint32_t a[16]; float f[16]; double d[16];                                                  \
        for (int i = 0; i < 16; i++) { a[i] = l * 7 + i; f[i] = l * 0.001f + i; d[i] = l + i; }     \
        for (int it = 0; it < iters; it++) {                                                        \
            _Pragma("unroll") for (int j = 0; j < 32; j++) { const int q = j & 15, q2 = (j + 5) & 15; (void)q2; BODY; } \
        }                                                                                           \
        float r = 0; for (int i = 0; i < 16; i++) r += f[i] + (float)a[i] + (float)d[i];            \
        out[blockIdx.x * 256 + l] = r;                                                              \
    }

KERNEL(v_fma_f32,        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[q]) : "v"(f[q2]), "v"(s)))
KERNEL(v_fma_f32_sgpr,   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[q]) : "s"(s), "v"(f[q2])))
KERNEL(v_fma_f32_clamp,  asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(f[q]) : "v"(f[q2]), "v"(s)))
KERNEL(v_mul_f32,        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[q]) : "v"(f[q2])))
KERNEL(v_add_f32,        asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[q]) : "v"(f[q2])))
KERNEL(v_sub_f32,        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[q]) : "v"(f[q2])))
KERNEL(v_mac_like_fmac,  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[q]) : "v"(f[q2]), "v"(s)))
KERNEL(v_max_f32,        asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[q]) : "v"(f[q2])))
KERNEL(v_med3_f32,       asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[q]) : "v"(f[q2]), "v"(s)))
KERNEL(v_mov_b32,        asm volatile("v_mov_b32 %0, %1" : "=v"(f[q]) : "v"(f[q2])))
KERNEL(v_mov_b64,        asm volatile("v_mov_b64 %0, %1" : "=v"(d[q]) : "v"(d[q2])))
KERNEL(v_cndmask_b32,    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[q]) : "v"(f[q2])))
KERNEL(v_and_b32,        asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[q]) : "v"(a[q2])))
KERNEL(v_add_u32,        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[q]) : "v"(a[q2])))
KERNEL(v_lshlrev_b32,    asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[q])))
KERNEL(v_lshl_add_u32,   asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[q]) : "v"(a[q2])))
KERNEL(v_mad_u32_u24,    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[q]) : "v"(a[q2]), "v"(m)))
KERNEL(v_mul_u32_u24,    asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[q]) : "v"(a[q2])))
KERNEL(v_perm_b32,       asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[q]) : "v"(a[q2]), "v"(m)))
KERNEL(v_bfe_u32,        asm volatile("v_bfe_u32 %0, %0, 3, 10" : "+v"(a[q])))
KERNEL(v_cvt_f32_u32,    asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[q]) : "v"(a[q2])))
KERNEL(v_cvt_f32_ubyte0, asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[q]) : "v"(a[q2])))
KERNEL(v_cvt_u32_f32,    asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(a[q]) : "v"(f[q2])))
KERNEL(v_cvt_i32_f32,    asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a[q]) : "v"(f[q2])))
KERNEL(v_fract_f32,      asm volatile("v_fract_f32 %0, %1" : "=v"(f[q]) : "v"(f[q2])))
KERNEL(v_floor_f32,      asm volatile("v_floor_f32 %0, %1" : "=v"(f[q]) : "v"(f[q2])))
KERNEL(v_rndne_f32,      asm volatile("v_rndne_f32 %0, %1" : "=v"(f[q]) : "v"(f[q2])))
KERNEL(v_exp_f32,        asm volatile("v_exp_f32 %0, %1" : "=v"(f[q]) : "v"(f[q2])))
KERNEL(v_log_f32,        asm volatile("v_log_f32 %0, %1" : "=v"(f[q]) : "v"(f[q2])))
KERNEL(v_rcp_f32,        asm volatile("v_rcp_f32 %0, %1" : "=v"(f[q]) : "v"(f[q2])))
KERNEL(v_cvt_f32_f16,    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f[q]) : "v"(a[q2])))
KERNEL(v_cvt_f32_f16_hi, asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f[q]) : "v"(a[q2])))
KERNEL(v_cvt_f16_f32,    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(a[q]) : "v"(f[q2])))
KERNEL(v_cvt_pk_f16_f32, asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a[q]) : "v"(f[q2]), "v"(f[q])))
KERNEL(v_cvt_pk_u8_f32,  asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[q]) : "v"(f[q2])))
KERNEL(v_pk_fma_f32,     asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[q]) : "v"(d[q2])))
KERNEL(v_pk_mul_f32,     asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[q]) : "v"(d[q2])))
KERNEL(v_pk_add_f32,     asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[q]) : "v"(d[q2])))
KERNEL(v_fma_mix_f32,    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(f[q]) : "v"(s), "v"(a[q2])))
KERNEL(v_pk_fma_f16,     asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(a[q]) : "v"(a[q2])))
KERNEL(v_dot2_f32_f16,   asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(f[q]) : "v"(a[q2]), "v"(m)))
KERNEL(v_mov_dpp,        asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(f[q]) : "v"(f[q2])))
KERNEL(v_add_f32_dpp,    asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(f[q]) : "v"(f[q2])))
KERNEL(v_readlane,       asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(f[q]) : "s20"))
KERNEL(v_cvt_pknorm_u16, asm volatile("v_cvt_pknorm_u16_f32 %0, %1, %2" : "=v"(a[q]) : "v"(f[q2]), "v"(f[q])))
KERNEL(v_alignbit_b32,   asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(a[q]) : "v"(a[q2])))
KERNEL(v_lshl_or_b32,    asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a[q]) : "v"(a[q2])))
KERNEL(v_and_or_b32,     asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[q]) : "v"(a[q2]), "v"(m)))

template <typename K>
static void run(const char *name, K kern, float *d_out)
{
    const int iters = 2000, blocks = 768;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 10, 0.999f, 0x3ffu);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, 0x3ffu);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / iters / 32.0 / (blocks / 256.0);
    printf("%-20s %6.2f ns / wave-instruction / SIMD\n", name, ns);
}
#define RUN(NAME) run(#NAME, k_##NAME, d_out)

int main()
{
    float *d_out; CK(hipMalloc(&d_out, 768 * 256 * 4));
    RUN(v_fma_f32); RUN(v_fma_f32_sgpr); RUN(v_fma_f32_clamp); RUN(v_mul_f32); RUN(v_add_f32); RUN(v_sub_f32); RUN(v_mac_like_fmac);
    RUN(v_max_f32); RUN(v_med3_f32); RUN(v_mov_b32); RUN(v_mov_b64); RUN(v_cndmask_b32); RUN(v_and_b32); RUN(v_add_u32);
    RUN(v_lshlrev_b32); RUN(v_lshl_add_u32); RUN(v_mad_u32_u24); RUN(v_mul_u32_u24); RUN(v_perm_b32); RUN(v_bfe_u32);
    RUN(v_cvt_f32_u32); RUN(v_cvt_f32_ubyte0); RUN(v_cvt_u32_f32); RUN(v_cvt_i32_f32); RUN(v_fract_f32); RUN(v_floor_f32); RUN(v_rndne_f32);
    RUN(v_exp_f32); RUN(v_log_f32); RUN(v_rcp_f32); RUN(v_cvt_f32_f16); RUN(v_cvt_f32_f16_hi); RUN(v_cvt_f16_f32); RUN(v_cvt_pk_f16_f32);
    RUN(v_cvt_pk_u8_f32); RUN(v_pk_fma_f32); RUN(v_pk_mul_f32); RUN(v_pk_add_f32); RUN(v_fma_mix_f32); RUN(v_pk_fma_f16); RUN(v_dot2_f32_f16);
    RUN(v_mov_dpp); RUN(v_add_f32_dpp); RUN(v_readlane); RUN(v_cvt_pknorm_u16); RUN(v_alignbit_b32); RUN(v_lshl_or_b32); RUN(v_and_or_b32);
    return 0;
}
