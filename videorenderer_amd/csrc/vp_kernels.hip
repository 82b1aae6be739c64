// vp_kernels.hip — pass-per-kernel path: one HIP kernel per reference draw call
// (ConvertColorPass, TextureResizeShader x/y, FinalPass).  Handles every format / scaler / rect the
// build accepts; intermediates live in HBM exactly where the reference keeps textures.
// Compiled with -ffp-contract=off: every a*b+c is two roundings, so results are bit-identical to the
// CPU oracle except through the transcendental instructions.  The fused fast path is vp_fused.hip.
#include <hip/hip_runtime.h>

#include "vp_convert.h"
#include "vp_device.h"
#include "vp_launch.h"

namespace mpcvr {

// ConvertColorPass — DX11VideoProcessor.cpp:3048-3101 : one thread per pixel of m_TexConvertOutput
__global__ __launch_bounds__(256) void k_convert(ConvertParams P, Surface out)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= P.out_w || j >= P.out_h) return;
    store_surface(out.ptr, out.pitch, out.fmt, i, j, convert_pixel(P, i, j));
}

// TextureResizeShader — DX11VideoProcessor.cpp:332-377 with ps_interpolation_* / ps_convolution.
// AXIS = filtered axis; `other` maps the unfiltered output coordinate to a source texel (point sample).
// AXIS = SCREEN axis the tap table is indexed by; SWAP (rotation 90/270, FillVertices :130-179): the taps address texture
// rows when they run along screen x (and columns along screen y).
template <int AXIS, bool SWAP>
__global__ __launch_bounds__(256) void k_resize(Surface in, AxisTaps taps, const int32_t *__restrict__ other,
                                               int out_w, int out_h, StoreParams st)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= out_w || y >= out_h) return;
    const int f = AXIS == 0 ? x : y;
    const int o = other[AXIS == 0 ? y : x];
    const int32_t *idx = taps.idx + (size_t)f * taps.ntaps;
    const float *w = taps.w + (size_t)f * taps.ntaps;
    constexpr bool TAPS_ON_TEX_X = (AXIS == 0) != SWAP;
    f3 acc;
    {
        const f3 q = TAPS_ON_TEX_X ? load_surface(in, idx[0], o) : load_surface(in, o, idx[0]);
        acc.x = w[0] * q.x; acc.y = w[0] * q.y; acc.z = w[0] * q.z;
    }
    for (int k = 1; k < taps.ntaps; k++) {
        const f3 q = TAPS_ON_TEX_X ? load_surface(in, idx[k], o) : load_surface(in, o, idx[k]);
        acc.x = acc.x + w[k] * q.x; acc.y = acc.y + w[k] * q.y; acc.z = acc.z + w[k] * q.z;
    }
    if (taps.normalise) {
        const float ww = taps.wsum[f];
        acc.x = acc.x / ww; acc.y = acc.y / ww; acc.z = acc.z / ww;
    }
    store_epilogue(st, x, y, acc);
}

// ps_resize_onepass_jinc2.hlsl:44-101 ("Jinc2m"): one 2-D draw — 4x4 texels around the sample position weighted by the
// windowed jinc of their distance, normalised, then anti-ringing towards the min/max of the inner 2x2 (strength 0.8)
__global__ __launch_bounds__(256) void k_jinc2(Surface in, DrawCoords dc, int out_w, int out_h, StoreParams st)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= out_w || y >= out_h) return;
    const float pi = 3.14159274101257324f;                 // acos(-1) folded to fp32
    const float wa = 0.416f * pi, wb = 0.985f * pi;
    const float cx = dc.rev_x ? (float)(dc.org_x + dc.len_x) - ((float)x + 0.5f) * dc.step_x : (float)dc.org_x + ((float)x + 0.5f) * dc.step_x;
    const float cy = dc.rev_y ? (float)(dc.org_y + dc.len_y) - ((float)y + 0.5f) * dc.step_y : (float)dc.org_y + ((float)y + 0.5f) * dc.step_y;
    const float pcx = dc.swap ? cy : cx, pcy = dc.swap ? cx : cy;      // pc = Tex * wh
    const float tcx = floorf(pcx - 0.5f) + 0.5f, tcy = floorf(pcy - 0.5f) + 0.5f;
    const int bx = (int)floorf(tcx), by = (int)floorf(tcy);
    float wsum = 0.0f;
    f3 color = {0.0f, 0.0f, 0.0f}, mn = {0, 0, 0}, mx = {0, 0, 0};
    for (int j = 0; j < 4; j++) {
        float w[4]; f3 c[4];
        float rowsum = 0.0f;
        for (int i = 0; i < 4; i++) {
            const float vx = (tcx + (float)(i - 1)) - pcx, vy = (tcy + (float)(j - 1)) - pcy;
            const float dd = sqrtf(vx * vx + vy * vy);
            w[i] = (dd == 0.0f) ? wa * wb : sinf(dd * wa) * sinf(dd * wb) / (dd * dd);
            rowsum = i == 0 ? w[i] : rowsum + w[i];
            c[i] = load_surface(in, clampi(bx + i - 1, 0, in.w - 1), clampi(by + j - 1, 0, in.h - 1));
        }
        wsum = j == 0 ? rowsum : wsum + rowsum;
        f3 r;
        r.x = w[0] * c[0].x; r.x = r.x + w[1] * c[1].x; r.x = r.x + w[2] * c[2].x; r.x = r.x + w[3] * c[3].x;
        r.y = w[0] * c[0].y; r.y = r.y + w[1] * c[1].y; r.y = r.y + w[2] * c[2].y; r.y = r.y + w[3] * c[3].y;
        r.z = w[0] * c[0].z; r.z = r.z + w[1] * c[1].z; r.z = r.z + w[2] * c[2].z; r.z = r.z + w[3] * c[3].z;
        if (j == 0) color = r; else { color.x = color.x + r.x; color.y = color.y + r.y; color.z = color.z + r.z; }
        if (j == 1) {
            mn.x = fminf(c[1].x, c[2].x); mn.y = fminf(c[1].y, c[2].y); mn.z = fminf(c[1].z, c[2].z);
            mx.x = fmaxf(c[1].x, c[2].x); mx.y = fmaxf(c[1].y, c[2].y); mx.z = fmaxf(c[1].z, c[2].z);
        } else if (j == 2) {
            mn.x = fminf(fminf(mn.x, c[1].x), c[2].x); mn.y = fminf(fminf(mn.y, c[1].y), c[2].y); mn.z = fminf(fminf(mn.z, c[1].z), c[2].z);
            mx.x = fmaxf(fmaxf(mx.x, c[1].x), c[2].x); mx.y = fmaxf(fmaxf(mx.y, c[1].y), c[2].y); mx.z = fmaxf(fmaxf(mx.z, c[1].z), c[2].z);
        }
    }
    color.x = color.x / wsum; color.y = color.y / wsum; color.z = color.z / wsum;
    f3 cl;
    cl.x = fminf(fmaxf(color.x, mn.x), mx.x); cl.y = fminf(fmaxf(color.y, mn.y), mx.y); cl.z = fminf(fmaxf(color.z, mn.z), mx.z);
    color.x = color.x + 0.8f * (cl.x - color.x); color.y = color.y + 0.8f * (cl.y - color.y); color.z = color.z + 0.8f * (cl.z - color.z);
    store_epilogue(st, x, y, color);
}

// TextureCopyRect(..., m_pPSHDR10ToneMapping, ...) — the HDR10 local tone-mapping post-scale step (:3359-3367)
__global__ __launch_bounds__(256) void k_hdr10_tonemap(Surface in, HdrToneMapParams tm, int out_w, int out_h, StoreParams st)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= out_w || y >= out_h) return;
    store_epilogue(st, x, y, hdr10_tonemap(load_surface(in, x, y), tm));
}

// TextureCopyRect(ps_simple) / FinalPass straight from the convert output (no size change)
__global__ __launch_bounds__(256) void k_copy(Surface in, int out_w, int out_h, StoreParams st)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= out_w || y >= out_h) return;
    store_epilogue(st, x, y, load_surface(in, x, y));
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline dim3 grid2d(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4, 1); }

// CopyFrameV210 (Helper.cpp:709-748) on the device: v210 dwords -> Y210 words (10 bits in the MSBs), one thread per
// pair of dwords (= 6 words), plus the reference's one-dword remainder at the end of a row
__global__ __launch_bounds__(256) void k_repack_v210(const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch,
                                                      int lines, int line_blocks, int remainder)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (y >= lines || i > line_blocks) return;
    const uint32_t *src32 = (const uint32_t *)(src + (size_t)y * src_pitch) + 2 * i;
    uint16_t *dst16 = (uint16_t *)(dst + (size_t)y * dst_pitch) + 6 * i;
    if (i < line_blocks) {
        const uint32_t s0 = src32[0], s1 = src32[1];
        dst16[0] = (uint16_t)((s0 >> 4) & 0xffc0);
        dst16[1] = (uint16_t)((s0 << 6) & 0xffc0);
        dst16[2] = (uint16_t)((s1 << 6) & 0xffc0);
        dst16[3] = (uint16_t)((s0 >> 14) & 0xffc0);
        dst16[4] = (uint16_t)((s1 >> 14) & 0xffc0);
        dst16[5] = (uint16_t)((s1 >> 4) & 0xffc0);
    } else if (remainder) {
        const uint32_t v = src32[0];
        dst16[0] = (uint16_t)((v >> 4) & 0xffc0);
        dst16[1] = (uint16_t)((v << 6) & 0xffc0);
    }
}

// The CopyFrame* functions of the interleaved RGB formats (Helper.cpp:444-482 RGB24, 548-566 RGB48, 600-645 BGR48,
// 647-663 BGRA64, 665-683 b64a, 770-787 r210, 414-428 as-is) as one texel per thread.  `n_px` = pixels the reference loop
// writes per row (RGB48: whole groups of four only, as written); bottom_up: negative source pitch (:1243-1248).
__global__ __launch_bounds__(256) void k_repack_rgb(int kind, const uint8_t *src, int src_pitch_abs, int bottom_up,
                                                     uint8_t *dst, int dst_pitch, int n_px, int lines)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (y >= lines || i >= n_px) return;
    const uint8_t *srow = src + (size_t)(bottom_up ? lines - 1 - y : y) * src_pitch_abs;
    uint8_t *drow = dst + (size_t)y * dst_pitch;
    if (kind == RPK_NONE) { ((uint32_t *)drow)[i] = ((const uint32_t *)srow)[i]; return; }
    if (kind == RPK_RGB24) {
        const uint8_t *p = srow + 3 * i;
        ((uint32_t *)drow)[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xff000000u;
        return;
    }
    if (kind == RPK_R210) {
        const uint32_t t = ((const uint32_t *)srow)[i];
        const uint32_t r = ((t & 0x0000003fu) << 4) | ((t & 0x0000f000u) >> 12);
        const uint32_t g = ((t & 0x00fc0000u) >> 8) | ((t & 0x00000f00u) << 8);
        const uint32_t b = ((t & 0xff000000u) >> 4) | ((t & 0x00030000u) << 12);
        ((uint32_t *)drow)[i] = r | g | b;
        return;
    }
    uint16_t c0, c1, c2;
    if (kind == RPK_RGB48 || kind == RPK_BGR48) {
        const uint16_t *p = (const uint16_t *)srow + 3 * i;
        c0 = p[0]; c1 = p[1]; c2 = p[2];
        if (kind == RPK_BGR48) { const uint16_t t = c0; c0 = c2; c2 = t; }
    } else if (kind == RPK_BGRA64) {
        const uint16_t *p = (const uint16_t *)srow + 4 * i;
        c0 = p[2]; c1 = p[1]; c2 = p[0];
    } else {    // b64a: big-endian words A,R,G,B
        const uint8_t *p = srow + 8 * i;
        c0 = (uint16_t)((p[2] << 8) | p[3]); c1 = (uint16_t)((p[4] << 8) | p[5]); c2 = (uint16_t)((p[6] << 8) | p[7]);
    }
    uint16_t *d = (uint16_t *)drow + 4 * i;
    d[0] = c0; d[1] = c1; d[2] = c2; d[3] = 0xffff;
}

hipError_t LaunchRepackRgb(int kind, const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch, int width, int lines, hipStream_t s)
{
    const int ap = src_pitch < 0 ? -src_pitch : src_pitch;
    const int bpp = kind == RPK_RGB24 ? 3 : (kind == RPK_RGB48 || kind == RPK_BGR48) ? 6 : (kind == RPK_BGRA64 || kind == RPK_B64A) ? 8 : 4;
    int n_px = ap / bpp;                                   // line_pixels of the reference loops
    if (n_px > width) n_px = width;                        // the texture row holds `width` texels
    if (kind == RPK_RGB48) n_px &= ~3;                     // CopyFrameRGB48 has no remainder branch (:552-563)
    if (n_px <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_repack_rgb, dim3((n_px + 255) / 256, lines, 1), dim3(256, 1, 1), 0, s,
                       kind, src, ap, src_pitch < 0 ? 1 : 0, dst, dst_pitch, n_px, lines);
    return hipGetLastError();
}

hipError_t LaunchRepackV210(const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch, int lines, hipStream_t s)
{
    const int dq = dst_pitch / 12, dr = dst_pitch % 12, sq = src_pitch / 8, sr = src_pitch % 8;
    int line_blocks, remainder;
    if (dq <= sq) { line_blocks = dq; remainder = dr != 0; } else { line_blocks = sq; remainder = sr != 0; }
    hipLaunchKernelGGL(k_repack_v210, dim3((line_blocks + 1 + 255) / 256, lines, 1), dim3(256, 1, 1), 0, s,
                       src, src_pitch, dst, dst_pitch, lines, line_blocks, remainder);
    return hipGetLastError();
}

hipError_t LaunchConvert(const ConvertParams &P, const Surface &out, hipStream_t s)
{
    hipLaunchKernelGGL(k_convert, grid2d(P.out_w, P.out_h), dim3(64, 4, 1), 0, s, P, out);
    return hipGetLastError();
}

hipError_t LaunchResize(int axis, bool swap, const Surface &in, const AxisTaps &taps, const int32_t *other,
                        int out_w, int out_h, const StoreParams &st, hipStream_t s)
{
    const dim3 g = grid2d(out_w, out_h), b(64, 4, 1);
    if (axis == 0 && !swap) hipLaunchKernelGGL((k_resize<0, false>), g, b, 0, s, in, taps, other, out_w, out_h, st);
    else if (axis == 0)     hipLaunchKernelGGL((k_resize<0, true>), g, b, 0, s, in, taps, other, out_w, out_h, st);
    else if (!swap)         hipLaunchKernelGGL((k_resize<1, false>), g, b, 0, s, in, taps, other, out_w, out_h, st);
    else                    hipLaunchKernelGGL((k_resize<1, true>), g, b, 0, s, in, taps, other, out_w, out_h, st);
    return hipGetLastError();
}

hipError_t LaunchHdr10ToneMap(const Surface &in, const HdrToneMapParams &tm, int out_w, int out_h, const StoreParams &st, hipStream_t s)
{
    hipLaunchKernelGGL(k_hdr10_tonemap, grid2d(out_w, out_h), dim3(64, 4, 1), 0, s, in, tm, out_w, out_h, st);
    return hipGetLastError();
}

hipError_t LaunchJinc2(const Surface &in, const DrawCoords &dc, int out_w, int out_h, const StoreParams &st, hipStream_t s)
{
    hipLaunchKernelGGL(k_jinc2, grid2d(out_w, out_h), dim3(64, 4, 1), 0, s, in, dc, out_w, out_h, st);
    return hipGetLastError();
}

hipError_t LaunchCopy(const Surface &in, int out_w, int out_h, const StoreParams &st, hipStream_t s)
{
    hipLaunchKernelGGL(k_copy, grid2d(out_w, out_h), dim3(64, 4, 1), 0, s, in, out_w, out_h, st);
    return hipGetLastError();
}

}  // namespace mpcvr
