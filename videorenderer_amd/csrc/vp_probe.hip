// vp_probe.hip — a measurement aid, not part of the video path: what this GPU sustains for the headline kernel's TRAFFIC SHAPE with no
// arithmetic at all — every 16-byte word of a source buffer read once and fanned out to `fan` destination words (24.9 MB of a 4K P010
// sample in, 132.7 MB of an 8K B8G8R8A8 target out is fan = 5.33; the probe takes an integer).  bench.py times it over the same ring of
// frames as the workload and prints the rate beside the 8 TB/s figure and torch's copy rate (roofline.empirical_shape_peak_GBps): a
// device-to-device copy reads as many bytes as it writes, the fused kernels write five times what they read.
// (tools/ubench/fill_probe.hip is the stand-alone original; round 2's numbers with it: 5.0 - 5.8 TB/s.)
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/mpcvr.h"
#include "vp_crmath.h"

namespace mpcvr {
namespace {
typedef uint32_t pr_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_probe_shape(const pr_u4 *__restrict__ src, pr_u4 *__restrict__ dst, size_t n16_src, int fan)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16_src; i += (size_t)gridDim.x * blockDim.x) {
        const pr_u4 s = src[i];
        for (int k = 0; k < fan; k++) dst[(size_t)k * n16_src + i] = pr_u4{s.x + (uint32_t)k, s.y, s.z, s.w};
    }
}
// the plain tier's transcendentals over an array (tests: device == the CPU evaluation of the same definition, bit for bit)
__global__ __launch_bounds__(256) void k_eval_transcendental(int fn, const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        out[i] = fn == 0 ? crm_log2f(v) : fn == 1 ? crm_exp2f(v) : fn == 2 ? crm_expf(v) : fn == 3 ? crm_powf(v, y[i]) : fn == 4 ? crm_sinf(v) : crm_cosf(v);
    }
}
}  // namespace
}  // namespace mpcvr

extern "C" int32_t mpcvr_eval_transcendental(int32_t fn, const float *x_dev, const float *y_dev, float *out_dev, size_t n, void *stream)
{
    if (!x_dev || !out_dev || (fn == 3 && !y_dev)) return MPCVR_E_POINTER;
    if (fn < 0 || fn > 5) return MPCVR_E_INVALIDARG;
    if (n == 0) return MPCVR_S_OK;
    hipLaunchKernelGGL(mpcvr::k_eval_transcendental, dim3(1024), dim3(256), 0, (hipStream_t)stream, (int)fn, x_dev, y_dev, out_dev, n);
    return hipGetLastError() == hipSuccess ? MPCVR_S_OK : MPCVR_E_FAIL;
}

extern "C" int32_t mpcvr_bandwidth_probe(const void *src_dev, void *dst_dev, size_t src_bytes, int32_t fan, void *stream)
{
    if (!src_dev || !dst_dev) return MPCVR_E_POINTER;
    if (fan < 1 || fan > 64 || src_bytes < 16 || (((uintptr_t)src_dev | (uintptr_t)dst_dev) & 15)) return MPCVR_E_INVALIDARG;
    hipLaunchKernelGGL(mpcvr::k_probe_shape, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const mpcvr::pr_u4 *)src_dev, (mpcvr::pr_u4 *)dst_dev, src_bytes / 16, (int)fan);
    return hipGetLastError() == hipSuccess ? MPCVR_S_OK : MPCVR_E_FAIL;
}

extern "C" int32_t mpcvr_eval_transcendental_host(int32_t fn, const float *x, const float *y, float *out, size_t n)
{
    if (!x || !out || (fn == 3 && !y)) return MPCVR_E_POINTER;
    if (fn < 0 || fn > 5) return MPCVR_E_INVALIDARG;
    for (size_t i = 0; i < n; i++)
        out[i] = fn == 0 ? mpcvr::crm_log2f(x[i]) : fn == 1 ? mpcvr::crm_exp2f(x[i]) : fn == 2 ? mpcvr::crm_expf(x[i]) : fn == 3 ? mpcvr::crm_powf(x[i], y[i]) : fn == 4 ? mpcvr::crm_sinf(x[i]) : mpcvr::crm_cosf(x[i]);
    return MPCVR_S_OK;
}
