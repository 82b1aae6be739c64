// What does v_cvt_pk_u8_f32 do with fractions (rounding mode) and out-of-range values on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, int n)
{
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0);
}
int main()
{
    const float h[] = {0.0f, 0.25f, 0.5f, 0.75f, 0.999f, 1.0f, 1.5f, 2.5f, 3.5f, 127.49f, 127.5f, 127.99f, 254.5f, 254.999f, 255.0f, 255.5f, 256.0f, 300.0f, -0.5f, -3.0f};
    const int n = sizeof(h) / sizeof(h[0]);
    float *d; unsigned *o; unsigned ho[64];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%9.3f -> %u\n", h[i], ho[i]);
    return 0;
}
