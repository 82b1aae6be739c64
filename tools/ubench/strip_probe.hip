// strip_probe.hip — two hardware questions behind the arbitrary-ratio strip kernel (vp_fused_strip.hip):
//   1. can a workgroup claim more than 64 KiB of dynamic LDS on gfx950 (160 KiB per CU)?
//   2. does v_fma_mix_f32 read fp16 SUBNORMAL operands exactly (UNORM codes 0..1023 stored as their integer bits
//      = k * 2^-24 as fp16), or are they flushed?
//   hipcc --offload-arch=gfx950 -O3 -o strip_probe strip_probe.hip && ./strip_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_lds(uint32_t *out, int words)
{
    extern __shared__ uint32_t sm[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) sm[i] = i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < words; i += blockDim.x) acc ^= sm[words - 1 - i];
    atomicXor(out, acc);
}

__global__ void k_mix(const uint32_t *codes, float w, float *out)
{
    const uint32_t h = codes[threadIdx.x];       // lo half = code a, hi half = code b (integer bits)
    float lo, hi;
    asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(h), "v"(w));
    asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(h), "v"(w));
    out[2 * threadIdx.x] = lo; out[2 * threadIdx.x + 1] = hi;
}

int main()
{
    int dev = 0, maxlds = 0, maxlds_optin = 0;
    CK(hipSetDevice(dev));
    CK(hipDeviceGetAttribute(&maxlds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    (void)hipDeviceGetAttribute(&maxlds_optin, hipDeviceAttributeSharedMemPerBlockOptin, dev);
    printf("MaxSharedMemoryPerBlock = %d, optin = %d\n", maxlds, maxlds_optin);
    uint32_t *d;
    CK(hipMalloc(&d, 4));
    for (int kb : {48, 64, 80, 100, 128, 160}) {
        CK(hipMemset(d, 0, 4));
        hipError_t ea = hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        hipLaunchKernelGGL(k_lds, dim3(512), dim3(256), (size_t)kb * 1024, 0, d, kb * 256);
        hipError_t el = hipGetLastError();
        hipError_t es = hipDeviceSynchronize();
        printf("LDS %3d KiB: setattr=%s launch=%s sync=%s\n", kb, hipGetErrorName(ea), hipGetErrorName(el), hipGetErrorName(es));
        if (es != hipSuccess) break;
    }
    // fp16 subnormal operands
    uint32_t hc[64]; float ho[128];
    for (int i = 0; i < 64; i++) hc[i] = (uint32_t)(i * 16 + 3) | ((uint32_t)(1023 - i * 16) << 16);
    uint32_t *dc; float *dout;
    CK(hipMalloc(&dc, sizeof(hc))); CK(hipMalloc(&dout, sizeof(ho)));
    CK(hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice));
    const float w = 0.37f * 16777216.0f / 1023.0f;
    hipLaunchKernelGGL(k_mix, dim3(1), dim3(64), 0, 0, dc, w, dout);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        const float ea = (float)(hc[i] & 0xffff) * (1.0f / 16777216.0f) * w, eb = (float)(hc[i] >> 16) * (1.0f / 16777216.0f) * w;
        if (ho[2 * i] != ea || ho[2 * i + 1] != eb) { if (bad < 4) printf("mix mismatch %d: got %g %g want %g %g\n", i, ho[2 * i], ho[2 * i + 1], ea, eb); bad++; }
    }
    printf("v_fma_mix_f32 with fp16 subnormal operands: %s (%d of 64 lanes differ)\n", bad ? "FLUSHED / inexact" : "exact", bad);
    return 0;
}
