// fill_probe.hip — what one MI355X sustains for the headline kernel's TRAFFIC SHAPE without any arithmetic: per frame 24.9 MB read
// (a 4K P010 sample) and 132.7 MB written (an 8K B8G8R8A8 target), over a ring of frames larger than the Infinity Cache.
//   fill : 16-byte stores only                       (pure write roof)
//   shape: the sample read with 16-byte loads, folded into the stored value, 10.7 bytes written per byte read
//   copy : 1 byte written per byte read              (the usual "HBM bandwidth" figure)
//   hipcc --offload-arch=gfx950 -O3 -o fill_probe fill_probe.hip && ./fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void k_fill(u4 *dst, size_t n16, uint32_t v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = u4{v, v + 1, v + 2, v + 3};
}
__global__ void k_fill_nt(u4 *dst, size_t n16, uint32_t v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(u4{v, v + 1, v + 2, v + 3}, dst + i);
}
// each workgroup owns a contiguous 64 KiB chunk at a time (the fused kernels' pattern: a wave writes whole row segments)
__global__ void k_fill_chunk(u4 *dst, size_t n16, uint32_t v)
{
    const size_t chunk = 4096;       // 16-byte words
    for (size_t c = blockIdx.x; c * chunk < n16; c += gridDim.x)
        for (size_t i = threadIdx.x; i < chunk && c * chunk + i < n16; i += blockDim.x) dst[c * chunk + i] = u4{v, v + 1, v + 2, v + 3};
}
// every 16-byte source word fans out to `fan` destination words (row-major neighbours: what a 2x upscale of packed samples does)
__global__ void k_shape(const u4 *src, u4 *dst, size_t n16_src, int fan)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16_src; i += (size_t)gridDim.x * blockDim.x) {
        const u4 s = src[i];
        for (int k = 0; k < fan; k++) dst[(size_t)k * n16_src + i] = u4{s.x + k, s.y, s.z, s.w};
    }
}

int main()
{
    CK(hipSetDevice(0));
    const size_t src_b = 3840ull * 2160 * 3, dst_b = 7680ull * 4320 * 4;       // P010: 3 B per pixel
    const int ring = 24;                                                       // 24 * (24.9 + 132.7) MB = 3.8 GB
    uint8_t *src, *dst;
    CK(hipMalloc(&src, src_b * ring)); CK(hipMalloc(&dst, dst_b * ring));
    CK(hipMemset(src, 1, src_b * ring));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 256 * 16, threads = 256;
    auto timed = [&](const char *name, double bytes_per_frame, auto launch) {
        for (int w = 0; w < ring; w++) launch(w);
        CK(hipDeviceSynchronize());
        const int reps = 4 * ring;
        CK(hipEventRecord(a));
        for (int r = 0; r < reps; r++) launch(r % ring);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-34s %8.3f ms / frame  %7.2f TB/s\n", name, ms / reps, bytes_per_frame * reps / (ms * 1e-3) / 1e12);
    };
    timed("fill  (132.7 MB written)", (double)dst_b, [&](int f) {
        hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(threads), 0, 0, (u4 *)(dst + dst_b * f), dst_b / 16, (uint32_t)f); });
    timed("fill, non-temporal stores", (double)dst_b, [&](int f) {
        hipLaunchKernelGGL(k_fill_nt, dim3(blocks), dim3(threads), 0, 0, (u4 *)(dst + dst_b * f), dst_b / 16, (uint32_t)f); });
    timed("fill, 64 KiB chunks per workgroup", (double)dst_b, [&](int f) {
        hipLaunchKernelGGL(k_fill_chunk, dim3(blocks), dim3(threads), 0, 0, (u4 *)(dst + dst_b * f), dst_b / 16, (uint32_t)f); });
    for (int bl : {256 * 4, 256 * 64})
        timed(bl == 1024 ? "fill, 1024 workgroups" : "fill, 16384 workgroups", (double)dst_b, [&](int f) {
            hipLaunchKernelGGL(k_fill, dim3(bl), dim3(threads), 0, 0, (u4 *)(dst + dst_b * f), dst_b / 16, (uint32_t)f); });
    // the headline's shape: 24.9 MB P010 sample -> fan 5 = 124 MB (closest integer fan to 132.7 / 24.9 = 5.33)
    timed("shape (24.9 MB read, 124.4 written)", (double)src_b * 6, [&](int f) {
        hipLaunchKernelGGL(k_shape, dim3(blocks), dim3(threads), 0, 0, (const u4 *)(src + src_b * f), (u4 *)(dst + dst_b * f), src_b / 16, 5); });
    timed("copy  (124.4 MB read, 124.4 written)", (double)src_b * 10, [&](int f) {
        hipLaunchKernelGGL(k_shape, dim3(blocks), dim3(threads), 0, 0, (const u4 *)(dst + dst_b * ((f + 7) % ring)), (u4 *)(dst + dst_b * f), src_b * 5 / 16, 1); });
    return 0;
}
