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
// The headline kernel's own traffic shape over a whole batch in ONE launch (round 6; the round-5 probe above is one 30 us launch per frame whose
// lanes feed five write streams 24.9 MB apart — not what k_fused_up2x does): a wavefront owns a strip of 120 source columns x a segment of
// source rows of one frame and marches down it two rows at a time; per step lanes 0..59 read one dword of each luma row and one of the
// chroma row (4:2:0 bi-planar 16-bit: 240 contiguous bytes per row and wavefront) and write FOUR output rows, one 16-byte piece per lane and
// row (960 contiguous bytes per row and wavefront, rows dst_pitch apart) — no arithmetic beyond keeping the loads alive.
// mode 0: read + write; 1: write only (a fill in the kernel's store pattern); 2: read only (one store per wavefront at the end).
// strip_cols: 120 = the kernel's strips (960-byte row pieces: every other strip starts in the middle of a 128-byte line); 128 = what a kernel
// whose wavefronts wrote whole 1 KiB-aligned row pieces would do — the A/B that says what the store pattern itself costs.
struct ProbeBatch { const uint8_t *src[64]; uint8_t *dst[64]; };
__global__ __launch_bounds__(256) void k_probe_up2x(ProbeBatch B, int mode, int W, int H, int seg_rows, int n_strips, int n_segs, int strip_cols)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave;
    if (item >= n_strips * n_segs) return;
    const int seg = item / n_strips, strip = item - seg * n_strips;
    const uint8_t *src = B.src[blockIdx.z];
    uint8_t *dst = B.dst[blockIdx.z];
    const size_t spitch = (size_t)W * 2, dpitch = (size_t)W * 2 * 4;           // P010 rows; 2W output pixels of 4 bytes
    const int col = min(strip * strip_cols + 2 * lane, W - 2);               // two source pixels per lane
    const bool active = 2 * lane < strip_cols && strip * strip_cols + 2 * lane < W;
    const uint8_t *luma = src + (size_t)col * 2, *chroma = src + spitch * H + (size_t)col * 2;
    uint8_t *out = dst + (size_t)col * 2 * 4;                                // output column 2 * col
    const int r0 = seg * seg_rows, r1 = min(r0 + seg_rows, H);
    uint32_t acc = 0;
    for (int r = r0; r < r1; r += 2) {
        uint32_t y0 = 0, y1 = 0, c = 0;
        if (mode != 1 && active) {
            y0 = *(const uint32_t *)(luma + spitch * r);
            y1 = *(const uint32_t *)(luma + spitch * (r + 1));
            c = *(const uint32_t *)(chroma + spitch * (r >> 1));
        }
        if (mode == 2) { acc += y0 ^ y1 ^ c; continue; }
        if (active) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                *(pr_u4 *)(out + dpitch * (size_t)(2 * r + k)) = pr_u4{y0 + (uint32_t)k, y1, c, (uint32_t)r};
        }
    }
    if (mode == 2 && acc == 0x12345u) *(uint32_t *)out = acc;                 // (keeps the loads; practically never taken)
}

// the plain tier's transcendentals over an array (tests: device == the CPU evaluation of the same definition, bit for bit)
__global__ __launch_bounds__(256) void k_eval_transcendental(int fn, const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        out[i] = fn == 0 ? crm_log2f(v) : fn == 1 ? crm_exp2f(v) : fn == 2 ? crm_expf(v) : fn == 3 ? crm_powf(v, y[i]) : fn == 4 ? crm_sinf(v) : fn == 5 ? crm_cosf(v) : fn == 6 ? v / y[i] : __builtin_sqrtf(v);
    }
}
}  // namespace
}  // namespace mpcvr

extern "C" int32_t mpcvr_eval_transcendental(int32_t fn, const float *x_dev, const float *y_dev, float *out_dev, size_t n, void *stream)
{
    if (!x_dev || !out_dev || ((fn == 3 || fn == 6) && !y_dev)) return MPCVR_E_POINTER;
    if (fn < 0 || fn > 7) return MPCVR_E_INVALIDARG;
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

extern "C" int32_t mpcvr_bandwidth_probe_up2x(int32_t mode, int32_t n, const void *const *srcs_dev, void *const *dsts_dev, int32_t src_w, int32_t src_h, int32_t seg_rows,
                                              int32_t strip_cols, void *stream)
{
    if (!srcs_dev || !dsts_dev) return MPCVR_E_POINTER;
    if (mode < 0 || mode > 2 || n < 1 || n > 64 || src_w < 2 || (src_w & 1) || src_h < 2 || (src_h & 1) || seg_rows < 2 || (seg_rows & 1) ||
        strip_cols < 2 || strip_cols > 128 || (strip_cols & 1)) return MPCVR_E_INVALIDARG;
    mpcvr::ProbeBatch b{};
    for (int i = 0; i < n; i++) {
        if (!srcs_dev[i] || !dsts_dev[i] || (((uintptr_t)srcs_dev[i] | (uintptr_t)dsts_dev[i]) & 15)) return MPCVR_E_INVALIDARG;
        b.src[i] = (const uint8_t *)srcs_dev[i]; b.dst[i] = (uint8_t *)dsts_dev[i];
    }
    const int n_strips = (src_w + strip_cols - 1) / strip_cols, n_segs = (src_h + seg_rows - 1) / seg_rows;
    hipLaunchKernelGGL(mpcvr::k_probe_up2x, dim3((n_strips * n_segs + 3) / 4, 1, n), dim3(256), 0, (hipStream_t)stream, b, (int)mode, (int)src_w, (int)src_h, (int)seg_rows, n_strips, n_segs, (int)strip_cols);
    return hipGetLastError() == hipSuccess ? MPCVR_S_OK : MPCVR_E_FAIL;
}
