// vp_jinc.hip — Jinc2m (Shaders/examples/ps_resize_onepass_jinc2.hlsl:44-101) at exactly 2x on both axes: a 2x2 output QUAD per lane.
//
// The one-draw 2-D upscaler weighs the 4x4 texels around the sample position with a windowed jinc of their distance, normalises,
// and pulls the result 80 % of the way into the min / max of the inner 2x2 (anti-ringing).  At 2x the four pixels of an output
// quad see the four phases of the weight table (BuildJincPhases) and their 4x4 neighbourhoods overlap in all but one row and one
// column: 5 x 5 texels serve the whole quad.  k_jinc2_phases (vp_kernels.hip) gives every output pixel its own lane, its own 16
// texel reads and its own 16 weight reads from LDS — 32 LDS reads per pixel; here a lane reads the 25 texels once, row by row, and
// the 64 weights as 16 wave-uniform 16-byte reads: 11 LDS reads per pixel, and the taps are real FMAs (this file is compiled with
// fused multiply-adds, the per-pixel kernels round twice): <= 1 LSB against them, like the other kernels of the default tier.
#include <hip/hip_runtime.h>

#include "vp_device.h"
#include "vp_launch.h"
#include "vp_plan.h"

namespace mpcvr {

namespace {

// floor(tc) of output o — the shader's expression (tc = floor(pc - 0.5) + 0.5, pc = Tex * wh)
__device__ __forceinline__ int jinc_base(int org, int o, float step)
{
    const float pc = (float)org + ((float)o + 0.5f) * step;
    return (int)floorf(floorf(pc - 0.5f) + 0.5f);
}

// EPI: 0 = store_epilogue (any target, window clipping); 1 = 10-bit m_TexsPostScale + ps_final_pass in integers into B8G8R8A8
// (the fused kernels' epilogue: (k * M + (j << 14)) >> 24, vp_fused.hip); 2 = straight UNORM store into B8G8R8A8 / R10G10B10A2.
// 1 and 2 need the quad inside the window and 8-byte aligned rows (one 8-byte store per lane and output row).
struct JincEpi { float maxv; uint32_t epi_mul; int out10; };
template <int INFMT, int EPI>
__global__ __launch_bounds__(256) void k_jinc2_quad(Surface in, DrawCoords dc, const JincPhases *__restrict__ tab, int out_w, int out_h, StoreParams st, JincEpi E, ResizeBatch bt)
{
    // a batch: frame z reads in.ptr + z * in_stride and writes frames[z].dst (vp_launch.h)
    in.ptr = (uint8_t *)in.ptr + (size_t)blockIdx.z * bt.in_stride;
    st.dst = bt.frames ? bt.frames[blockIdx.z].dst : (void *)((uint8_t *)st.dst + (size_t)blockIdx.z * bt.dst_stride);
    // the 128 x 8 outputs of the workgroup read (64 + 4) x (4 + 4) source texels: decoded once into LDS, clamp addressing applied there
    constexpr int TW = 72, TH = 8;
    __shared__ float4 W4[4][4];          // [phase = py * 2 + px][j] = weights of row j (i = 0..3)
    __shared__ float INVW[4];
    __shared__ float4 tile[TH * TW];
    __shared__ uint32_t Di[EPI == 1 ? 1024 : 1];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if (EPI == 1)
        for (int i = tid; i < 1024; i += 256) Di[i] = (uint32_t)(__half2float(__ushort_as_half(st.dither[i])) * 1024.0f + 0.5f) << 14;
    if (tid < 16) {
        const int ph = tid >> 2, j = tid & 3;
        const float *w = tab->w[ph >> 1][ph & 1] + 4 * j;
        W4[ph][j] = make_float4(w[0], w[1], w[2], w[3]);
        if (j == 0) INVW[ph] = 1.0f / tab->wsum[ph >> 1][ph & 1];
    }
    const int qx0 = blockIdx.x * 64, qy0 = blockIdx.y * 4;
    const int bx_lo = jinc_base(dc.org_x, 2 * qx0, dc.step_x) - 1, by_lo = jinc_base(dc.org_y, 2 * qy0, dc.step_y) - 1;
    const int bx_hi = jinc_base(dc.org_x, min(2 * qx0 + 127, out_w - 1), dc.step_x) + 2;
    const int by_hi = jinc_base(dc.org_y, min(2 * qy0 + 7, out_h - 1), dc.step_y) + 2;
    const int ncols = min(bx_hi - bx_lo + 1, TW), nrows = min(by_hi - by_lo + 1, TH);
    // (no division per texel: wavefront w takes rows w and w + 4, its lanes 64 columns; the last eight columns of all eight rows go to wavefront 0)
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int r = threadIdx.y + 4 * rr, c = threadIdx.x;
        if (r < nrows && c < ncols) {
            const f3 q = decode_texel<INFMT>(load_texel_raw<INFMT>(in, clampi(bx_lo + c, 0, in.w - 1), clampi(by_lo + r, 0, in.h - 1)));
            tile[r * TW + c] = make_float4(q.x, q.y, q.z, 0.0f);
        }
    }
    if (threadIdx.y == 0) {
        const int r = threadIdx.x >> 3, c = 64 + (threadIdx.x & 7);
        if (r < nrows && c < ncols) {
            const f3 q = decode_texel<INFMT>(load_texel_raw<INFMT>(in, clampi(bx_lo + c, 0, in.w - 1), clampi(by_lo + r, 0, in.h - 1)));
            tile[r * TW + c] = make_float4(q.x, q.y, q.z, 0.0f);
        }
    }
    __syncthreads();
    const int x = 2 * (qx0 + threadIdx.x), y = 2 * (qy0 + threadIdx.y);
    if (x >= out_w || y >= out_h) return;
    // even outputs read columns / rows base-1 .. base+2, odd outputs one further on (base(2q+1) = base(2q) + 1 at exactly 2x)
    const int cb = jinc_base(dc.org_x, x, dc.step_x) - 1 - bx_lo, rb = jinc_base(dc.org_y, y, dc.step_y) - 1 - by_lo;
    f3 col[4], mn[4], mx[4];             // pixel p = (row parity) * 2 + (column parity)
#pragma unroll
    for (int sr = 0; sr < 5; sr++) {
        float4 t[5];
#pragma unroll
        for (int i = 0; i < 5; i++) t[i] = tile[(rb + sr) * TW + cb + i];
#pragma unroll
        for (int rp = 0; rp < 2; rp++) {
            const int j = sr - rp;                      // this source row is tap row j of the output row with parity rp
            if (j < 0 || j > 3) continue;
#pragma unroll
            for (int cp = 0; cp < 2; cp++) {
                const int p = rp * 2 + cp;
                const float4 w = W4[p][j];
                const float4 c0 = t[cp], c1 = t[cp + 1], c2 = t[cp + 2], c3 = t[cp + 3];
                f3 r;
                // (vp_device.h switches contraction off for the rest of the translation unit: the FMAs are spelt out)
                if (j == 0) r = f3{w.x * c0.x, w.x * c0.y, w.x * c0.z};
                else r = f3{__builtin_fmaf(w.x, c0.x, col[p].x), __builtin_fmaf(w.x, c0.y, col[p].y), __builtin_fmaf(w.x, c0.z, col[p].z)};
                r.x = __builtin_fmaf(w.y, c1.x, r.x); r.y = __builtin_fmaf(w.y, c1.y, r.y); r.z = __builtin_fmaf(w.y, c1.z, r.z);
                r.x = __builtin_fmaf(w.z, c2.x, r.x); r.y = __builtin_fmaf(w.z, c2.y, r.y); r.z = __builtin_fmaf(w.z, c2.z, r.z);
                r.x = __builtin_fmaf(w.w, c3.x, r.x); r.y = __builtin_fmaf(w.w, c3.y, r.y); r.z = __builtin_fmaf(w.w, c3.z, r.z);
                col[p] = r;
                if (j == 1) {
                    mn[p] = f3{fminf(c1.x, c2.x), fminf(c1.y, c2.y), fminf(c1.z, c2.z)};
                    mx[p] = f3{fmaxf(c1.x, c2.x), fmaxf(c1.y, c2.y), fmaxf(c1.z, c2.z)};
                } else if (j == 2) {
                    mn[p] = f3{fminf(fminf(mn[p].x, c1.x), c2.x), fminf(fminf(mn[p].y, c1.y), c2.y), fminf(fminf(mn[p].z, c1.z), c2.z)};
                    mx[p] = f3{fmaxf(fmaxf(mx[p].x, c1.x), c2.x), fmaxf(fmaxf(mx[p].y, c1.y), c2.y), fmaxf(fmaxf(mx[p].z, c1.z), c2.z)};
                }
            }
        }
    }
    uint32_t pk[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int ox = x + (p & 1), oy = y + (p >> 1);
        const float iw = INVW[p];
        f3 c = {col[p].x * iw, col[p].y * iw, col[p].z * iw};
        // clamp(c, mn, mx) with mn <= mx is the median of the three: one v_med3_f32 per channel (min(max()) was two)
        const f3 cl = {__builtin_amdgcn_fmed3f(c.x, mn[p].x, mx[p].x), __builtin_amdgcn_fmed3f(c.y, mn[p].y, mx[p].y), __builtin_amdgcn_fmed3f(c.z, mn[p].z, mx[p].z)};
        c.x = __builtin_fmaf(0.8f, cl.x - c.x, c.x); c.y = __builtin_fmaf(0.8f, cl.y - c.y, c.y); c.z = __builtin_fmaf(0.8f, cl.z - c.z, c.z);
        if (EPI == 0) {
            if (ox < out_w && oy < out_h) store_epilogue(st, ox, oy, c);
            continue;
        }
        // x*maxv + 2^23 leaves the UNORM code in the low mantissa bits
        const uint32_t kr = __float_as_uint(__builtin_fmaf(__builtin_amdgcn_fmed3f(c.x, 0.0f, 1.0f), E.maxv, 8388608.0f));
        const uint32_t kg = __float_as_uint(__builtin_fmaf(__builtin_amdgcn_fmed3f(c.y, 0.0f, 1.0f), E.maxv, 8388608.0f));
        const uint32_t kb = __float_as_uint(__builtin_fmaf(__builtin_amdgcn_fmed3f(c.z, 0.0f, 1.0f), E.maxv, 8388608.0f));
        if (EPI == 1) {
            const uint32_t dj = Di[((oy + st.off_y) & 31) * 32 + ((ox + st.off_x) & 31)];
            const uint32_t ib = __umul24(kb, E.epi_mul) + dj, ig = __umul24(kg, E.epi_mul) + dj, ir = __umul24(kr, E.epi_mul) + dj;
            pk[p] = __builtin_amdgcn_perm(ir, __builtin_amdgcn_perm(ig, ib, 0x0c0c0703u), 0x0d070100u);       // [B, G, R, 0xff]
        } else if (E.out10) {
            pk[p] = (kb << 20) | ((kg << 10) | (kr + 0x75000000u));              // 0x4B000000 | k: see vp_fused_strip.hip
        } else {
            pk[p] = __builtin_amdgcn_perm(kr, __builtin_amdgcn_perm(kg, kb, 0x0c0c0400u), 0x0d040100u);
        }
    }
    if (EPI != 0) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int rp = 0; rp < 2; rp++) {
            if (y + rp >= out_h) continue;
            uint8_t *row = (uint8_t *)st.dst + (size_t)(y + rp + st.off_y) * st.dst_pitch + (size_t)(x + st.off_x) * 4;
            *(u32x2 *)row = u32x2{pk[2 * rp], pk[2 * rp + 1]};                   // out_w is even: the pair is inside
        }
    }
}

}  // namespace

bool Jinc2QuadSupported(const Surface &in, const DrawCoords &dc, int out_w, int out_h, const StoreParams &)
{
    if (dc.swap || dc.rev_x || dc.rev_y || dc.step_x != 0.5f || dc.step_y != 0.5f) return false;
    if (out_w < 2 || out_h < 2 || (out_w & 1) || (out_h & 1)) return false;
    return in.fmt == SF_BGRA8 || in.fmt == SF_RGB10A2 || in.fmt == SF_RGBA16F;
}

hipError_t LaunchJinc2Quad(const Surface &in, const DrawCoords &dc, int out_w, int out_h, const StoreParams &st, hipStream_t s, const void *phases_dev,
                           const ResizeBatch *batch)
{
    const ResizeBatch one{}, &bt = batch ? *batch : one;
    const dim3 grid((out_w / 2 + 63) / 64, (out_h / 2 + 3) / 4, bt.n), block(64, 4, 1);
    const JincPhases *tab = (const JincPhases *)phases_dev;
    // fast epilogues: the whole image inside the window, rows and origin 8-byte aligned
    const bool inside = st.off_x >= 0 && st.off_y >= 0 && (st.clip_w <= 0 || (st.off_x + out_w <= st.clip_w && st.off_y + out_h <= st.clip_h));
    const bool aligned = (st.off_x & 1) == 0 && (st.dst_pitch & 7) == 0 && ((uintptr_t)st.dst & 7) == 0 && (bt.dst_stride & 7) == 0 && bt.dst_aligned8;
    JincEpi e{255.0f, 0u, 0};
    int epi = 0;
    if (inside && aligned && st.mode == ST_FINAL && st.mid_fmt == SF_RGB10A2 && st.dst_fmt == SF_BGRA8 && st.quant == 255 && FinalPassMultiplier(255, 1023) != 0) {
        epi = 1; e.maxv = 1023.0f; e.epi_mul = FinalPassMultiplier(255, 1023);
    } else if (inside && aligned && st.mode == ST_SURFACE && (st.dst_fmt == SF_BGRA8 || st.dst_fmt == SF_RGB10A2)) {
        epi = 2; e.out10 = st.dst_fmt == SF_RGB10A2; e.maxv = e.out10 ? 1023.0f : 255.0f;
    }
#define MPCVR_JQ(F) do { if (epi == 1) hipLaunchKernelGGL((k_jinc2_quad<F, 1>), grid, block, 0, s, in, dc, tab, out_w, out_h, st, e, bt); \
                         else if (epi == 2) hipLaunchKernelGGL((k_jinc2_quad<F, 2>), grid, block, 0, s, in, dc, tab, out_w, out_h, st, e, bt); \
                         else hipLaunchKernelGGL((k_jinc2_quad<F, 0>), grid, block, 0, s, in, dc, tab, out_w, out_h, st, e, bt); } while (0)
    if (in.fmt == SF_BGRA8) MPCVR_JQ(SF_BGRA8);
    else if (in.fmt == SF_RGB10A2) MPCVR_JQ(SF_RGB10A2);
    else {      // an fp16 texture is the fp16 internal format's convert output (no source arrives as fp16): its final pass is never the 10 -> 8 one
        if (epi == 1) return hipErrorNotSupported;
        if (epi == 2) hipLaunchKernelGGL((k_jinc2_quad<SF_RGBA16F, 2>), grid, block, 0, s, in, dc, tab, out_w, out_h, st, e, bt);
        else hipLaunchKernelGGL((k_jinc2_quad<SF_RGBA16F, 0>), grid, block, 0, s, in, dc, tab, out_w, out_h, st, e, bt);
    }
#undef MPCVR_JQ
    return hipGetLastError();
}

}  // namespace mpcvr
