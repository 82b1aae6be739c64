// vp_kernels.hip — pass-per-kernel path: one HIP kernel per reference draw call
// (ConvertColorPass, TextureResizeShader x/y, FinalPass).  Handles every format / scaler / rect the
// build accepts; intermediates live in HBM exactly where the reference keeps textures.
// Compiled with -ffp-contract=off: every a*b+c is two roundings, so results are bit-identical to the
// CPU oracle except through the transcendental instructions.  The fused fast path is vp_fused.hip.
#include <hip/hip_runtime.h>

#include "vp_convert.h"
#include "vp_device.h"
#include <cstdlib>
#include <cstring>

#include "vp_launch.h"

namespace mpcvr {

// ConvertColorPass — DX11VideoProcessor.cpp:3048-3101 : one thread per pixel of m_TexConvertOutput
__global__ __launch_bounds__(256) void k_convert(ConvertParams P, Surface out)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= P.out_w || j >= P.out_h) return;
    P.pq_lut = nullptr;                    // the plain kernels evaluate every tail literally
    store_surface(out.ptr, out.pitch, out.fmt, i, j, convert_pixel(P, i, j));
}

// ConvertColorPass + the copy / FinalPass that follows it when nothing is resized (Process :3321-3367 with no resize draw):
// the value takes the rounding of m_TexConvertOutput in registers and goes straight into the last draw's epilogue
__global__ __launch_bounds__(256) void k_convert_direct(ConvertParams P, StoreParams st)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= P.out_w || j >= P.out_h) return;
    P.pq_lut = nullptr;
    store_epilogue(st, i, j, round_to_fmt(convert_pixel(P, i, j), P.out_fmt));
}

// The same two kernels with the source description folded at compile time for the everyday sources — planar / bi-planar
// 4:2:0 with bilinear chroma, progressive, no Dolby Vision: the D3D11 path compiles one shader per such combination
// (GetShaderConvertColor); here the combination becomes template arguments, the per-pixel arithmetic is the generic code's,
// expression for expression, and only the never-taken branches disappear.
// OFMT = format of m_TexConvertOutput; DMODE 0: store into it, 1: ST_SURFACE epilogue into the render target (format DFMT),
// 2: ST_FINAL epilogue (final pass) into the render target.
// DV: Dolby Vision variant (P.dovi stays: reshaping, LMS step, trims) — 16-bit containers with the PQ->SDR tail or none.
template <int PLANES, int BYTES, int TAIL, int OFMT, int DMODE, int DFMT, bool DV = false>
__global__ __launch_bounds__(256) void k_convert_420(ConvertParams P, Surface out, StoreParams st)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= P.out_w || j >= P.out_h) return;
    P.fmt.layout = LAY_PLANAR; P.fmt.planes = PLANES; P.fmt.bytes = BYTES;
    P.fmt.subsampling = 420; P.fmt.div_w = 2; P.fmt.div_h = 2;
    P.chroma_scaling = 1; P.blend_deint = 0; P.tail = TAIL;
    if (!DV) P.dovi = nullptr;
    if (BYTES == 1 || PLANES == 2) P.fmt.shift = 0;
    if (TAIL != TAIL_PQ_TO_SDR || DV) P.pq_lut = nullptr;
    const f3 v = convert_pixel(P, i, j);
    if (DMODE == 0) { store_surface(out.ptr, out.pitch, OFMT, i, j, v); return; }
    st.mode = DMODE == 2 ? ST_FINAL : ST_SURFACE; st.mid_fmt = OFMT; st.dst_fmt = DFMT;
    st.quant = DFMT == SF_RGB10A2 ? 1023 : 255;
    store_epilogue(st, i, j, round_to_fmt(v, OFMT));
}

// TextureResizeShader — DX11VideoProcessor.cpp:332-377 with ps_interpolation_* / ps_convolution.
// AXIS = filtered axis; `other` maps the unfiltered output coordinate to a source texel (point sample).
// AXIS = SCREEN axis the tap table is indexed by; SWAP (rotation 90/270, FillVertices :130-179): the taps address texture
// rows when they run along screen x (and columns along screen y).
template <int AXIS, bool SWAP>
__global__ __launch_bounds__(256) void k_resize(Surface in, AxisTaps taps, const int32_t *__restrict__ other,
                                               int out_w, int out_h, StoreParams st, ResizeBatch bt)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= out_w || y >= out_h) return;
    in.ptr = (uint8_t *)in.ptr + (size_t)blockIdx.z * bt.in_stride;        // a batch: frame z's texture and target (vp_launch.h)
    st.dst = bt.frames ? bt.frames[blockIdx.z].dst : (void *)((uint8_t *)st.dst + (size_t)blockIdx.z * bt.dst_stride);
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

// ------------------------------------------------------------------------------------------------
// the same draw, folded: tap count, source format and epilogue at compile time; the arithmetic per output pixel is
// k_resize's, expression for expression (first tap a product, then acc + w*q in tap order).
// ------------------------------------------------------------------------------------------------
enum EpiCode : int { EPI_RUNTIME = 0, EPI_TO_FP16 = 1, EPI_FINAL_10_TO_8 = 2, EPI_FINAL_16F_TO_8 = 3, EPI_FINAL_16F_TO_10 = 4,
                     EPI_TO_BGRA8 = 5, EPI_TO_RGB10 = 6 };
template <int EPI>
__device__ __forceinline__ void store_epi(StoreParams st, int x, int y, f3 v, int dither_pre = -1)
{
    if (EPI == EPI_TO_FP16) { st.mode = ST_SURFACE; st.dst_fmt = SF_RGBA16F; }
    if (EPI == EPI_FINAL_10_TO_8) { st.mode = ST_FINAL; st.mid_fmt = SF_RGB10A2; st.dst_fmt = SF_BGRA8; st.quant = 255; }
    if (EPI == EPI_FINAL_16F_TO_8) { st.mode = ST_FINAL; st.mid_fmt = SF_RGBA16F; st.dst_fmt = SF_BGRA8; st.quant = 255; }
    if (EPI == EPI_FINAL_16F_TO_10) { st.mode = ST_FINAL; st.mid_fmt = SF_RGBA16F; st.dst_fmt = SF_RGB10A2; st.quant = 1023; }
    if (EPI == EPI_TO_BGRA8) { st.mode = ST_SURFACE; st.dst_fmt = SF_BGRA8; }
    if (EPI == EPI_TO_RGB10) { st.mode = ST_SURFACE; st.dst_fmt = SF_RGB10A2; }
    store_epilogue(st, x, y, v, dither_pre);
}
static int EpiOf(const StoreParams &st)
{
    if (st.mode == ST_SURFACE) return st.dst_fmt == SF_RGBA16F ? EPI_TO_FP16 : st.dst_fmt == SF_BGRA8 ? EPI_TO_BGRA8 : st.dst_fmt == SF_RGB10A2 ? EPI_TO_RGB10 : EPI_RUNTIME;
    if (st.mid_fmt == SF_RGB10A2 && st.dst_fmt == SF_BGRA8 && st.quant == 255) return EPI_FINAL_10_TO_8;
    if (st.mid_fmt == SF_RGBA16F && st.dst_fmt == SF_BGRA8 && st.quant == 255) return EPI_FINAL_16F_TO_8;
    if (st.mid_fmt == SF_RGBA16F && st.dst_fmt == SF_RGB10A2 && st.quant == 1023) return EPI_FINAL_16F_TO_10;
    return EPI_RUNTIME;
}

// the row-tap kernel's LDS window: worth it when the 8 output rows of a group share their source rows at least 3:1
// (measured: 1.33x Lanczos3, 12 rows for 48 tap reads, -11 %; 1.5x 4-tap downscale, 16 rows for 32 reads, +20 %)
__host__ __device__ inline bool RowWindowPays(const AxisTaps &taps, int nt)
{
    return taps.blk8_lo && taps.blk8_span <= kResizeRowSpanMax && taps.blk8_span * 3 <= nt * 8;
}

// Both kernels give a thread several output pixels: a wave that loads its arguments, its taps, one texel per tap and the
// dither texel in sequence and then retires spends its life waiting (four dependent memory latencies for ~85 ALU
// instructions) and the launch becomes latency x (waves / waves in flight); with 4 pixels per thread every one of those
// latencies is shared by 4x the loads.
//
// taps run down the texture rows, columns map 1:1 (the second draw of every two-pass resize, or a Y-only resize): one
// output row per block row, so the tap indices and weights of the row are wave-uniform (scalar loads) and every tap
// is one fully coalesced row read.  A thread owns 4 consecutive columns.
template <int NT, int INFMT, int EPI, int PX>
__global__ __launch_bounds__(256) void k_resize_rows(Surface in, AxisTaps taps, int out_w, int out_h, int gx, StoreParams st, ResizeBatch bt)
{
    // A workgroup owns RW consecutive output rows of its columns and walks them with the texels of the next row already
    // in flight: one wave = one row would spend its life in four dependent round trips (arguments, tap table, texels,
    // dither) for ~85 ALU instructions per pixel; here those latencies are paid once per RW rows.
    constexpr int RW = 8;
    // workgroups are dealt round-robin to the 8 XCDs, each with its own L2: give XCD k the k-th contiguous band of
    // output rows (its taps then re-read rows its own L2 already holds) instead of every 8th row group.  1-D grid of
    // 8 * ceil(units / 8) workgroups; logical id = (id mod 8) * (grid / 8) + id / 8 is a bijection on it.
    // With a batch the bands run over whole frames: an XCD works on its own frames.
    const int per = gridDim.x >> 3, lid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int groups = (out_h + RW - 1) / RW;
    const int rg = lid / gx, bx = lid - rg * gx;
    const int z = rg / groups, y0 = (rg - z * groups) * RW;
    if (z >= bt.n) return;
    in.ptr = (uint8_t *)in.ptr + (size_t)z * bt.in_stride;
    st.dst = bt.frames ? bt.frames[z].dst : (void *)((uint8_t *)st.dst + (size_t)z * bt.dst_stride);
    // pixel p of a thread sits 256 columns after pixel p-1: every load of a wave stays one contiguous row segment
    const int x0 = bx * (256 * PX) + threadIdx.x;
    if (x0 >= out_w) return;
    const int nt = NT ? NT : taps.ntaps;
    int xs[PX];
#pragma unroll
    for (int p = 0; p < PX; p++) xs[p] = min(x0 + p * 256, out_w - 1);
    const bool final = EPI == EPI_FINAL_10_TO_8 || EPI == EPI_FINAL_16F_TO_8 || EPI == EPI_FINAL_16F_TO_10;

    if (NT == 0) {          // run-time tap count (convolution downscalers): no register prefetch
        in.fmt = INFMT;
        for (int r = 0; r < RW && y0 + r < out_h; r++) {
            const int y = y0 + r;
            const int32_t *idx = taps.idx + (size_t)y * nt;
            const float *w = taps.w + (size_t)y * nt;
            f3 acc[PX];
#pragma unroll
            for (int p = 0; p < PX; p++) {
                const f3 q = load_surface(in, xs[p], idx[0]);
                acc[p].x = w[0] * q.x; acc[p].y = w[0] * q.y; acc[p].z = w[0] * q.z;
            }
            for (int k = 1; k < nt; k++) {
#pragma unroll
                for (int p = 0; p < PX; p++) {
                    const f3 q = load_surface(in, xs[p], idx[k]);
                    acc[p].x = acc[p].x + w[k] * q.x; acc[p].y = acc[p].y + w[k] * q.y; acc[p].z = acc[p].z + w[k] * q.z;
                }
            }
            if (taps.normalise) {
                const float ww = taps.wsum[y];
#pragma unroll
                for (int p = 0; p < PX; p++) { acc[p].x = acc[p].x / ww; acc[p].y = acc[p].y / ww; acc[p].z = acc[p].z / ww; }
            }
#pragma unroll
            for (int p = 0; p < PX; p++)
                if (x0 + p * 256 < out_w) store_epi<EPI>(st, x0 + p * 256, y, acc[p]);
        }
        return;
    }

    constexpr int NTC = NT ? NT : 1;
    // The 8 output rows of the group touch only blk8_span (<= 14) distinct source rows (12 for a 1.33x Lanczos3 instead
    // of 8 x 6 tap reads): each lane fetches its columns of those rows ONCE, parks them in LDS — used here as a register
    // file a wave can index with a (wave-uniform) run-time row number; no lane reads another lane's data, so there is no
    // barrier — and takes its taps from there.  (An ablation of this kernel on MI355X, 1440p Lanczos3: without the tap
    // reads it runs 2.3x faster, without the stores 1.25x, without the dither reads 1.06x, pure ALU 3.7x.)
    if (RowWindowPays(taps, NTC)) {
        extern __shared__ __attribute__((aligned(16))) unsigned char resize_smem[];
        const int span = taps.blk8_span;
        uint2 *const W = (uint2 *)resize_smem + (size_t)(threadIdx.x >> 6) * span * PX * 64 + (threadIdx.x & 63);
        const int lo = taps.blk8_lo[y0 >> 3];
        {
            uint2 t[kResizeRowSpanMax][PX];
#pragma unroll
            for (int sl = 0; sl < kResizeRowSpanMax; sl++)
                if (sl < span) {
                    const int ys = min(lo + sl, in.h - 1);
#pragma unroll
                    for (int p = 0; p < PX; p++) t[sl][p] = load_texel_raw<INFMT>(in, xs[p], ys);
                }
#pragma unroll
            for (int sl = 0; sl < kResizeRowSpanMax; sl++)
                if (sl < span) {
#pragma unroll
                    for (int p = 0; p < PX; p++) W[(sl * PX + p) * 64] = t[sl][p];
                }
        }
#pragma unroll 2
        for (int r = 0; r < RW; r++) {
            const int y = y0 + r;
            if (y >= out_h) break;
            const int32_t *idx = taps.idx + (size_t)y * NTC;
            const float *w = taps.w + (size_t)y * NTC;
            int dth[PX];
#pragma unroll
            for (int p = 0; p < PX; p++)
                dth[p] = final ? (int)st.dither[((y + st.off_y) & 31) * 32 + ((xs[p] + st.off_x) & 31)] : -1;
            f3 acc[PX];
#pragma unroll
            for (int p = 0; p < PX; p++) {
                const f3 q = decode_texel<INFMT>(W[((idx[0] - lo) * PX + p) * 64]);
                acc[p].x = w[0] * q.x; acc[p].y = w[0] * q.y; acc[p].z = w[0] * q.z;
            }
#pragma unroll
            for (int k = 1; k < NTC; k++) {
#pragma unroll
                for (int p = 0; p < PX; p++) {
                    const f3 q = decode_texel<INFMT>(W[((idx[k] - lo) * PX + p) * 64]);
                    acc[p].x = acc[p].x + w[k] * q.x; acc[p].y = acc[p].y + w[k] * q.y; acc[p].z = acc[p].z + w[k] * q.z;
                }
            }
            if (taps.normalise) {
                const float ww = taps.wsum[y];
#pragma unroll
                for (int p = 0; p < PX; p++) { acc[p].x = acc[p].x / ww; acc[p].y = acc[p].y / ww; acc[p].z = acc[p].z / ww; }
            }
#pragma unroll
            for (int p = 0; p < PX; p++)
                if (x0 + p * 256 < out_w) store_epi<EPI>(st, x0 + p * 256, y, acc[p], dth[p]);
        }
        return;
    }
    uint2 raw[2][NTC][PX];
    int dth[2][PX];
    auto fetch = [&](int y, int slot) {
        const int yc = min(y, out_h - 1);
        const int32_t *idx = taps.idx + (size_t)yc * NTC;
#pragma unroll
        for (int k = 0; k < NTC; k++)
#pragma unroll
            for (int p = 0; p < PX; p++) raw[slot][k][p] = load_texel_raw<INFMT>(in, xs[p], idx[k]);
#pragma unroll
        for (int p = 0; p < PX; p++)
            dth[slot][p] = final ? (int)st.dither[((yc + st.off_y) & 31) * 32 + ((xs[p] + st.off_x) & 31)] : -1;
    };
    fetch(y0, 0);
#pragma unroll
    for (int r = 0; r < RW; r++) {
        const int y = y0 + r, cur = r & 1;
        if (y >= out_h) break;
        if (r + 1 < RW) fetch(y + 1, cur ^ 1);
        const float *w = taps.w + (size_t)y * NTC;
        f3 acc[PX];
#pragma unroll
        for (int p = 0; p < PX; p++) {
            const f3 q = decode_texel<INFMT>(raw[cur][0][p]);
            acc[p].x = w[0] * q.x; acc[p].y = w[0] * q.y; acc[p].z = w[0] * q.z;
        }
#pragma unroll
        for (int k = 1; k < NTC; k++) {
#pragma unroll
            for (int p = 0; p < PX; p++) {
                const f3 q = decode_texel<INFMT>(raw[cur][k][p]);
                acc[p].x = acc[p].x + w[k] * q.x; acc[p].y = acc[p].y + w[k] * q.y; acc[p].z = acc[p].z + w[k] * q.z;
            }
        }
        if (taps.normalise) {
            const float ww = taps.wsum[y];
#pragma unroll
            for (int p = 0; p < PX; p++) { acc[p].x = acc[p].x / ww; acc[p].y = acc[p].y / ww; acc[p].z = acc[p].z / ww; }
        }
#pragma unroll
        for (int p = 0; p < PX; p++)
            if (x0 + p * 256 < out_w) store_epi<EPI>(st, x0 + p * 256, y, acc[p], dth[cur][p]);
    }
}

// taps run along the texture columns: a wave owns 64 consecutive outputs of 4 rows, decodes the source texels they touch
// ONCE into LDS (taps.blk_lo / blk_span, built with the table) and then reads its taps from there, instead of decoding
// every texel once per tap that uses it; the tap table of an output column is read once for the 4 rows.
template <int NT, int INFMT, int EPI>
__global__ __launch_bounds__(256) void k_resize_cols(Surface in, AxisTaps taps, const int32_t *__restrict__ other,
                                                    int out_w, int out_h, StoreParams st, ResizeBatch bt)
{
    constexpr int R = 4;
    // [wave][row][span] decoded texels; span = taps.blk_span rounded up (dynamic: upscales need ~70 texels per row, and the
    // smaller the tile the more workgroups a CU holds — this kernel lives on occupancy, not on ALU)
    extern __shared__ __attribute__((aligned(16))) unsigned char resize_smem[];
    const int span = (taps.blk_span + 3) & ~3;
    float4 *const tile_w = (float4 *)resize_smem + (size_t)threadIdx.y * R * span;
    in.ptr = (uint8_t *)in.ptr + (size_t)blockIdx.z * bt.in_stride;
    st.dst = bt.frames ? bt.frames[blockIdx.z].dst : (void *)((uint8_t *)st.dst + (size_t)blockIdx.z * bt.dst_stride);
    const int lane = threadIdx.x, wv = threadIdx.y;
    const int x = blockIdx.x * 64 + lane, yb = blockIdx.y * (4 * R) + wv * R;
    const int nt = NT ? NT : taps.ntaps;
    in.fmt = INFMT;
    const int lo = taps.blk_lo[blockIdx.x];
    const int hi = min(lo + taps.blk_span, in.w);
    {
        // all texel reads of the tile first (up to 3 per row: span <= 192), then decode and park them: issued from inside the
        // per-row loop, each read waited for its own round trip (an ablation put 300 of this kernel's 590 us there)
        constexpr int IT = (kResizeSpanMax + 63) / 64;
        uint2 rawv[R][IT];
        int orow[R];
#pragma unroll
        for (int r = 0; r < R; r++) orow[r] = other[min(yb + r, out_h - 1)];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int p = lo + lane + 64 * it;
                if (p < hi) rawv[r][it] = load_texel_raw<INFMT>(in, p, orow[r]);
            }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int p = lo + lane + 64 * it;
                if (p < hi) {
                    const f3 q = decode_texel<INFMT>(rawv[r][it]);
                    tile_w[r * span + (p - lo)] = make_float4(q.x, q.y, q.z, 0.0f);
                }
            }
    }
    __syncthreads();
    if (x >= out_w) return;
    const int32_t *idx = taps.idx_t + x;
    const float *w = taps.w_t + x;
    const size_t n = (size_t)taps.n_out;
    f3 acc[R];
    {
        const int i0 = idx[0] - lo;
        const float w0 = w[0];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float4 q = tile_w[r * span + i0];
            acc[r].x = w0 * q.x; acc[r].y = w0 * q.y; acc[r].z = w0 * q.z;
        }
    }
#pragma unroll
    for (int k = 1; k < nt; k++) {
        const int ik = idx[k * n] - lo;
        const float wk = w[k * n];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float4 q = tile_w[r * span + ik];
            acc[r].x = acc[r].x + wk * q.x; acc[r].y = acc[r].y + wk * q.y; acc[r].z = acc[r].z + wk * q.z;
        }
    }
    if (taps.normalise) {
        const float ww = taps.wsum[x];
#pragma unroll
        for (int r = 0; r < R; r++) { acc[r].x = acc[r].x / ww; acc[r].y = acc[r].y / ww; acc[r].z = acc[r].z / ww; }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
        if (yb + r < out_h) store_epi<EPI>(st, x, yb + r, acc[r]);
}

// Both draws of a two-pass resize in one kernel, tiled through LDS: a workgroup owns a 64 x 32 tile of the output, runs the X
// draw for the source rows the tile touches (8 rows at a time: decoded m_TexConvertOutput texels in LDS slice A, taps read
// from there) and parks its results — rounded to fp16 exactly like m_TexResize — in LDS slice B, then runs the Y draw from B
// straight into the final pass.  m_TexResize never exists in memory, and the Y draw's taps — the L2 reads the row kernel
// spends most of its time on — become LDS reads.  Same tap order, same roundings as the two kernels above.
template <int NT, int INFMT, int EPI>
__global__ __launch_bounds__(256) void k_resize_2d(Surface in, AxisTaps tx, AxisTaps ty, const int32_t *__restrict__ other, int mid_h,
                                                  int out_w, int out_h, int tiles_x, int tiles_y, StoreParams st, ResizeBatch bt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char resize_smem[];
    constexpr int CH = 8;                                   // source rows per X-draw chunk
    const int spanX = (tx.blk_span + 3) & ~3, spanY = ty.blk32_span;
    float4 *const A = (float4 *)resize_smem;                 // [CH][spanX] decoded texels
    uint2 *const B = (uint2 *)(resize_smem + (size_t)CH * spanX * sizeof(float4));      // [spanY][64] X-draw results as half4
    // XCD-contiguous tile ranges (see k_resize_rows); with a batch an XCD works on its own frames
    const int per = gridDim.x >> 3, lid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int tiles = tiles_x * tiles_y, z = lid / tiles;
    if (z >= bt.n) return;
    const int t = lid - z * tiles, tyi = t / tiles_x, txi = t - tyi * tiles_x;
    in.ptr = (uint8_t *)in.ptr + (size_t)z * bt.in_stride;
    st.dst = bt.frames ? bt.frames[z].dst : (void *)((uint8_t *)st.dst + (size_t)z * bt.dst_stride);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = txi * 64 + lane, xc = min(x, out_w - 1);
    const int ntx = NT ? NT : tx.ntaps, nty = NT ? NT : ty.ntaps;
    const int loX = tx.blk_lo[txi], nX = min(tx.blk_span, in.w - loX);
    // rows of the X draw's result (m_TexResize) = entries of `other`: mid_h, which is the source RECT's height — not in.h when the
    // draw reads the source texture itself (interleaved RGB without a convert draw)
    const int loY = ty.blk32_lo[tyi], nY = min(spanY, mid_h - loY);
    constexpr int NTC = NT ? NT : 1;
    // the X taps of this lane's column, once (tap-major tables)
    int ix[NTC]; float wx[NTC];
    if (NT) {
#pragma unroll
        for (int k = 0; k < NTC; k++) { ix[k] = tx.idx_t[xc + (size_t)k * tx.n_out] - loX; wx[k] = tx.w_t[xc + (size_t)k * tx.n_out]; }
    }
    const float wsx = tx.normalise ? tx.wsum[xc] : 1.0f;

    for (int c0 = 0; c0 < nY; c0 += CH) {
        {   // stage CH source rows: wave w takes rows w and w + 4 of the chunk; every texel read first, then decode + park
            constexpr int IT = (kResizeSpanMax + 63) / 64;
            uint2 rawv[2][IT];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int rr = wave + 4 * h;
                const int o = other[min(loY + c0 + rr, mid_h - 1)];
#pragma unroll
                for (int it = 0; it < IT; it++) {
                    const int p = lane + 64 * it;
                    if (p < nX && c0 + rr < nY) rawv[h][it] = load_texel_raw<INFMT>(in, loX + p, o);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int rr = wave + 4 * h;
#pragma unroll
                for (int it = 0; it < IT; it++) {
                    const int p = lane + 64 * it;
                    if (p < nX && c0 + rr < nY) {
                        const f3 q = decode_texel<INFMT>(rawv[h][it]);
                        A[rr * spanX + p] = make_float4(q.x, q.y, q.z, 0.0f);
                    }
                }
            }
        }
        __syncthreads();
        // X draw of the chunk: wave w filters rows 2w and 2w + 1
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int rr = wave * 2 + j;
            if (c0 + rr >= nY) break;
            f3 acc;
            if (NT) {
                const float4 q0 = A[rr * spanX + ix[0]];
                acc.x = wx[0] * q0.x; acc.y = wx[0] * q0.y; acc.z = wx[0] * q0.z;
#pragma unroll
                for (int k = 1; k < NTC; k++) {
                    const float4 q = A[rr * spanX + ix[k]];
                    acc.x = acc.x + wx[k] * q.x; acc.y = acc.y + wx[k] * q.y; acc.z = acc.z + wx[k] * q.z;
                }
            } else {
                const float4 q0 = A[rr * spanX + tx.idx_t[xc] - loX];
                const float w0 = tx.w_t[xc];
                acc.x = w0 * q0.x; acc.y = w0 * q0.y; acc.z = w0 * q0.z;
                for (int k = 1; k < ntx; k++) {
                    const float4 q = A[rr * spanX + tx.idx_t[xc + (size_t)k * tx.n_out] - loX];
                    const float wk = tx.w_t[xc + (size_t)k * tx.n_out];
                    acc.x = acc.x + wk * q.x; acc.y = acc.y + wk * q.y; acc.z = acc.z + wk * q.z;
                }
            }
            if (tx.normalise) { acc.x = acc.x / wsx; acc.y = acc.y / wsx; acc.z = acc.z / wsx; }
            // the store into m_TexResize (fp16, RNE), kept as its bits
            const __half2 lo = __halves2half2(__float2half_rn(acc.x), __float2half_rn(acc.y));
            const __half2 hi = __halves2half2(__float2half_rn(acc.z), __float2half_rn(1.0f));
            uint2 u;
            u.x = *(const uint32_t *)&lo; u.y = *(const uint32_t *)&hi;
            B[(c0 + rr) * 64 + lane] = u;
        }
        __syncthreads();
    }

    // Y draw + epilogue: wave w produces output rows 8w .. 8w + 7 of the tile
    if (x >= out_w) return;
    const bool final = EPI == EPI_FINAL_10_TO_8 || EPI == EPI_FINAL_16F_TO_8 || EPI == EPI_FINAL_16F_TO_10;
#pragma unroll 2
    for (int j = 0; j < 8; j++) {
        const int y = tyi * 32 + wave * 8 + j;
        if (y >= out_h) break;
        const int32_t *idx = ty.idx + (size_t)y * nty;
        const float *w = ty.w + (size_t)y * nty;
        const int dth = final ? (int)st.dither[((y + st.off_y) & 31) * 32 + ((x + st.off_x) & 31)] : -1;
        f3 acc;
        {
            const f3 q = decode_texel<SF_RGBA16F>(B[(idx[0] - loY) * 64 + lane]);
            acc.x = w[0] * q.x; acc.y = w[0] * q.y; acc.z = w[0] * q.z;
        }
        if (NT) {
#pragma unroll
            for (int k = 1; k < NTC; k++) {
                const f3 q = decode_texel<SF_RGBA16F>(B[(idx[k] - loY) * 64 + lane]);
                acc.x = acc.x + w[k] * q.x; acc.y = acc.y + w[k] * q.y; acc.z = acc.z + w[k] * q.z;
            }
        } else {
            for (int k = 1; k < nty; k++) {
                const f3 q = decode_texel<SF_RGBA16F>(B[(idx[k] - loY) * 64 + lane]);
                acc.x = acc.x + w[k] * q.x; acc.y = acc.y + w[k] * q.y; acc.z = acc.z + w[k] * q.z;
            }
        }
        if (ty.normalise) {
            const float ww = ty.wsum[y];
            acc.x = acc.x / ww; acc.y = acc.y / ww; acc.z = acc.z / ww;
        }
        store_epi<EPI>(st, x, y, acc, dth);
    }
}

// ps_resize_onepass_jinc2.hlsl:44-101 ("Jinc2m"): one 2-D draw — 4x4 texels around the sample position weighted by the
// windowed jinc of their distance, normalised, then anti-ringing towards the min/max of the inner 2x2 (strength 0.8)
// ctr: Tex * wh of every output column, then of every output row (BuildDrawCentres: the host evaluates TexCenter once per column and row
// — its fp64 interpolation and division — instead of every pixel twice on the device)
__global__ __launch_bounds__(256) void k_jinc2(Surface in, DrawCoords dc, const float *__restrict__ ctr, int out_w, int out_h, StoreParams st, ResizeBatch bt)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= out_w || y >= out_h) return;
    in.ptr = (uint8_t *)in.ptr + (size_t)blockIdx.z * bt.in_stride;        // a batch: frame z's texture and target (vp_launch.h)
    st.dst = bt.frames ? bt.frames[blockIdx.z].dst : (void *)((uint8_t *)st.dst + (size_t)blockIdx.z * bt.dst_stride);
    const float pi = 3.14159274101257324f;                 // acos(-1) folded to fp32
    const float wa = 0.416f * pi, wb = 0.985f * pi;
    const float cx = ctr[x], cy = ctr[dc.n_x + y];
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
            w[i] = (dd == 0.0f) ? wa * wb : crm_sinf(dd * wa) * crm_sinf(dd * wb) / (dd * dd);      // sin as the oracle defines it (vp_crmath.h): the weights carry its bits
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
// (a batch: frame blockIdx.z reads in + z * in_stride and writes its own render target)
__global__ __launch_bounds__(256) void k_hdr10_tonemap(Surface in, HdrToneMapParams tm, int out_w, int out_h, StoreParams st, ResizeBatch bt)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= out_w || y >= out_h) return;
    in.ptr = (uint8_t *)in.ptr + (size_t)blockIdx.z * bt.in_stride;
    st.dst = bt.frames ? bt.frames[blockIdx.z].dst : (void *)((uint8_t *)st.dst + (size_t)blockIdx.z * bt.dst_stride);
    store_epilogue(st, x, y, hdr10_tonemap(load_surface(in, x, y), tm));
}

// TextureCopyRect(ps_simple) / FinalPass straight from the convert output (no size change)
__global__ __launch_bounds__(256) void k_copy(Surface in, int out_w, int out_h, StoreParams st)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= out_w || y >= out_h) return;
    store_epilogue(st, x, y, load_surface(in, x, y));
}

// Jinc2m at a dyadic ratio (2x, 4x, or an unscaled axis): the sample position's offset from its 4x4 neighbourhood — all
// the weights depend on — takes 1/step values per axis, exactly (every term of the shader's expression is exact in fp32
// there), so the 16 weights and their sum come from a <= 4 x 4 phase table built on the host with the shader's own
// expressions instead of 16 x (sqrt, 2 sin, divide) per pixel.  The accumulation is k_jinc2's, in the same order.
__global__ __launch_bounds__(256) void k_jinc2_phases(Surface in, DrawCoords dc, const JincPhases *__restrict__ tab, int out_w, int out_h, StoreParams st, ResizeBatch bt)
{
    in.ptr = (uint8_t *)in.ptr + (size_t)blockIdx.z * bt.in_stride;        // a batch: frame z's texture and target (vp_launch.h)
    st.dst = bt.frames ? bt.frames[blockIdx.z].dst : (void *)((uint8_t *)st.dst + (size_t)blockIdx.z * bt.dst_stride);
    // the 64 x 4 outputs of the workgroup read at most (64 + 4) x (4 + 4) source texels (step <= 1): decoded once into LDS
    // (with the clamp addressing applied there), the 16 taps of a pixel are LDS reads
    constexpr int TW = 68, THh = 8;
    __shared__ float W[16 * 16];
    __shared__ float WS[16];
    __shared__ float4 tile[THh * TW];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    W[tid] = (&tab->w[0][0][0])[tid];
    if (tid < 16) WS[tid] = (&tab->wsum[0][0])[tid];
    auto base_of = [](int org, int o, float step) {         // floor(tc) of output o, the shader's expression
        const float pc = (float)org + ((float)o + 0.5f) * step;
        return (int)floorf(floorf(pc - 0.5f) + 0.5f);
    };
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 4;
    const int bx_lo = base_of(dc.org_x, x0, dc.step_x) - 1, bx_hi = base_of(dc.org_x, min(x0 + 63, out_w - 1), dc.step_x) + 2;
    const int by_lo = base_of(dc.org_y, y0, dc.step_y) - 1, by_hi = base_of(dc.org_y, min(y0 + 3, out_h - 1), dc.step_y) + 2;
    const int ncols = bx_hi - bx_lo + 1, nrows = by_hi - by_lo + 1;           // <= TW, <= THh
    for (int t = tid; t < ncols * nrows; t += 256) {
        const int r = t / ncols, c = t - r * ncols;
        const f3 q = load_surface(in, clampi(bx_lo + c, 0, in.w - 1), clampi(by_lo + r, 0, in.h - 1));
        tile[r * TW + c] = make_float4(q.x, q.y, q.z, 0.0f);
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= out_w || y >= out_h) return;
    const int bx = base_of(dc.org_x, x, dc.step_x), by = base_of(dc.org_y, y, dc.step_y);
    const int ph = (y & (tab->py - 1)) * 4 + (x & (tab->px - 1));
    const float *w = W + ph * 16;
    const float wsum = WS[ph];
    f3 color = {0.0f, 0.0f, 0.0f}, mn = {0, 0, 0}, mx = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f3 c[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float4 q = tile[(by + j - 1 - by_lo) * TW + (bx + i - 1 - bx_lo)];
            c[i] = f3{q.x, q.y, q.z};
        }
        const float *wj = w + 4 * j;
        f3 r;
        r.x = wj[0] * c[0].x; r.x = r.x + wj[1] * c[1].x; r.x = r.x + wj[2] * c[2].x; r.x = r.x + wj[3] * c[3].x;
        r.y = wj[0] * c[0].y; r.y = r.y + wj[1] * c[1].y; r.y = r.y + wj[2] * c[2].y; r.y = r.y + wj[3] * c[3].y;
        r.z = wj[0] * c[0].z; r.z = r.z + wj[1] * c[1].z; r.z = r.z + wj[2] * c[2].z; r.z = r.z + wj[3] * c[3].z;
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

// The correction shaders (m_pPSCorrection: ps_fix_bt2020 / ps_fix_ycgco / ps_fixconvert_pq_to_sdr / ps_fixconvert_hlg_to_sdr /
// ps_convert_pq_to_sdr / ps_convert_hlg_to_pq) as a same-size pass over one 32-bit surface.  KIND = MPCVR_CORR_*.
struct CorrParams { float fix[16]; float gamut[9]; float lum_scale; };
template <int KIND>
__global__ __launch_bounds__(256) void k_correction(Surface in, Surface out, CorrParams K)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= out.w || y >= out.h) return;
    f3 c = load_surface(in, x, y);
    const float a = 1.0f;                                   // alpha never reaches r, g, b (4th matrix column is zero)
    if (KIND != 5 && KIND != 6) {                           // mul(fix_*_matrix, color)
        f3 r;
        r.x = K.fix[0] * c.x + K.fix[1] * c.y + K.fix[2] * c.z + K.fix[3] * a;
        r.y = K.fix[4] * c.x + K.fix[5] * c.y + K.fix[6] * c.z + K.fix[7] * a;
        r.z = K.fix[8] * c.x + K.fix[9] * c.y + K.fix[10] * c.z + K.fix[11] * a;
        c = r;
    }
    if (KIND == 6) {                                        // ps_convert_hlg_to_pq.hlsl:15-19
        c.x = saturate(c.x); c.y = saturate(c.y); c.z = saturate(c.z);
        c = hlg_to_linear(c);
        c.x = linear_to_st2084(c.x, 1000.0f); c.y = linear_to_st2084(c.y, 1000.0f); c.z = linear_to_st2084(c.z, 1000.0f);
    } else if (KIND != 2) {
        c.x = saturate(c.x); c.y = saturate(c.y); c.z = saturate(c.z);
        if (KIND == 1) {                                    // ps_fix_bt2020.hlsl:24-25: sRGB to linear
            c.x = hlsl_pow(c.x, 2.2f); c.y = hlsl_pow(c.y, 2.2f); c.z = hlsl_pow(c.z, 2.2f);
        } else {
            if (KIND == 4) {                                // ps_fixconvert_hlg_to_sdr.hlsl:31-34: HLG to PQ
                c = hlg_to_linear(c);
                c.x = saturate(linear_to_st2084(c.x, 1000.0f)); c.y = saturate(linear_to_st2084(c.y, 1000.0f)); c.z = saturate(linear_to_st2084(c.z, 1000.0f));
            }
            c.x = st2084_to_linear(c.x, K.lum_scale); c.y = st2084_to_linear(c.y, K.lum_scale); c.z = st2084_to_linear(c.z, K.lum_scale);
            const float div = hable_div();
            c.x = hable(c.x) / div; c.y = hable(c.y) / div; c.z = hable(c.z) / div;
        }
        c = mat3_mul(make_mat3(K.gamut), c);
        c.x = hlsl_pow(saturate(c.x), 1.0f / 2.2f); c.y = hlsl_pow(saturate(c.y), 1.0f / 2.2f); c.z = hlsl_pow(saturate(c.z), 1.0f / 2.2f);
    }
    store_surface(out.ptr, out.pitch, out.fmt, x, y, c);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline dim3 grid2d(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4, 1); }

// CopyFrameV210 (Helper.cpp:709-748) on the device: v210 dwords -> Y210 words (10 bits in the MSBs), one thread per
// pair of dwords (= 6 words), plus the reference's one-dword remainder at the end of a row
// (srcs.n != 0: a batch — frame z = blockIdx.z reads srcs.f[z].src and writes dst + z * dst_stride)
__global__ __launch_bounds__(256) void k_repack_v210(const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch,
                                                      int lines, int line_blocks, int remainder, SrcTable32 srcs, size_t dst_stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (y >= lines || i > line_blocks) return;
    if (srcs.n) { src = srcs.p[blockIdx.z]; dst += (size_t)blockIdx.z * dst_stride; }
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
                                                     uint8_t *dst, int dst_pitch, int n_px, int lines, SrcTable32 srcs, size_t dst_stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (y >= lines || i >= n_px) return;
    if (srcs.n) { src = srcs.p[blockIdx.z]; dst += (size_t)blockIdx.z * dst_stride; }
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

// batches: up to 32 source pointers travel in the kernel arguments per launch
static SrcTable32 SrcTable(const void *const *srcs, int at, int n)
{
    SrcTable32 t;
    t.n = n;
    for (int i = 0; i < 32; i++) t.p[i] = i < n ? (const uint8_t *)srcs[at + i] : nullptr;
    return t;
}

hipError_t LaunchRepackRgb(int kind, const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch, int width, int lines, hipStream_t s,
                           const void *const *srcs, int n, size_t dst_stride)
{
    const int ap = src_pitch < 0 ? -src_pitch : src_pitch;
    const int bpp = kind == RPK_RGB24 ? 3 : (kind == RPK_RGB48 || kind == RPK_BGR48) ? 6 : (kind == RPK_BGRA64 || kind == RPK_B64A) ? 8 : 4;
    int n_px = ap / bpp;                                   // line_pixels of the reference loops
    if (n_px > width) n_px = width;                        // the texture row holds `width` texels
    if (kind == RPK_RGB48) n_px &= ~3;                     // CopyFrameRGB48 has no remainder branch (:552-563)
    if (n_px <= 0) return hipSuccess;
    if (!srcs) {
        hipLaunchKernelGGL(k_repack_rgb, dim3((n_px + 255) / 256, lines, 1), dim3(256, 1, 1), 0, s,
                           kind, src, ap, src_pitch < 0 ? 1 : 0, dst, dst_pitch, n_px, lines, SrcTable32{}, (size_t)0);
        return hipGetLastError();
    }
    for (int at = 0; at < n; at += 32) {
        const int m = n - at < 32 ? n - at : 32;
        hipLaunchKernelGGL(k_repack_rgb, dim3((n_px + 255) / 256, lines, m), dim3(256, 1, 1), 0, s,
                           kind, nullptr, ap, src_pitch < 0 ? 1 : 0, dst + (size_t)at * dst_stride, dst_pitch, n_px, lines, SrcTable(srcs, at, m), dst_stride);
    }
    return hipGetLastError();
}

hipError_t LaunchRepackV210(const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch, int lines, hipStream_t s,
                            const void *const *srcs, int n, size_t dst_stride)
{
    const int dq = dst_pitch / 12, dr = dst_pitch % 12, sq = src_pitch / 8, sr = src_pitch % 8;
    int line_blocks, remainder;
    if (dq <= sq) { line_blocks = dq; remainder = dr != 0; } else { line_blocks = sq; remainder = sr != 0; }
    if (!srcs) {
        hipLaunchKernelGGL(k_repack_v210, dim3((line_blocks + 1 + 255) / 256, lines, 1), dim3(256, 1, 1), 0, s,
                           src, src_pitch, dst, dst_pitch, lines, line_blocks, remainder, SrcTable32{}, (size_t)0);
        return hipGetLastError();
    }
    for (int at = 0; at < n; at += 32) {
        const int m = n - at < 32 ? n - at : 32;
        hipLaunchKernelGGL(k_repack_v210, dim3((line_blocks + 1 + 255) / 256, lines, m), dim3(256, 1, 1), 0, s,
                           nullptr, src_pitch, dst + (size_t)at * dst_stride, dst_pitch, lines, line_blocks, remainder, SrcTable(srcs, at, m), dst_stride);
    }
    return hipGetLastError();
}

// picks the folded instantiation of k_convert_420 for this source / destination, or returns false
template <int OFMT, int DMODE, int DFMT>
static void LaunchConvert420T(const ConvertParams &P, const Surface &out, const StoreParams &st, hipStream_t s)
{
    const dim3 g = grid2d(P.out_w, P.out_h), b(64, 4, 1);
#define MPCVR_C420(PL, BY, TL) hipLaunchKernelGGL((k_convert_420<PL, BY, TL, OFMT, DMODE, DFMT>), g, b, 0, s, P, out, st)
#define MPCVR_C420_T(PL, BY) \
    switch (P.tail) { case TAIL_NONE: MPCVR_C420(PL, BY, TAIL_NONE); break; case TAIL_PQ_TO_SDR: MPCVR_C420(PL, BY, TAIL_PQ_TO_SDR); break; \
                      case TAIL_HLG_TO_SDR: MPCVR_C420(PL, BY, TAIL_HLG_TO_SDR); break; default: MPCVR_C420(PL, BY, TAIL_GAMMA_GAMUT); break; }
    if (P.dovi) {         // Convert420Eligible admits Dolby Vision for 16-bit containers with TAIL_NONE / TAIL_PQ_TO_SDR only
#define MPCVR_C420_DV(PL, TL) hipLaunchKernelGGL((k_convert_420<PL, 2, TL, OFMT, DMODE, DFMT, true>), g, b, 0, s, P, out, st)
        if (P.fmt.planes == 2) { if (P.tail == TAIL_NONE) MPCVR_C420_DV(2, TAIL_NONE); else MPCVR_C420_DV(2, TAIL_PQ_TO_SDR); }
        else                   { if (P.tail == TAIL_NONE) MPCVR_C420_DV(3, TAIL_NONE); else MPCVR_C420_DV(3, TAIL_PQ_TO_SDR); }
#undef MPCVR_C420_DV
        return;
    }
    // (8-bit samples: without a tail only — Convert420Eligible)
    if (P.fmt.planes == 2) { if (P.fmt.bytes == 1) MPCVR_C420(2, 1, TAIL_NONE); else { MPCVR_C420_T(2, 2) } }
    else                   { if (P.fmt.bytes == 1) MPCVR_C420(3, 1, TAIL_NONE); else { MPCVR_C420_T(3, 2) } }
#undef MPCVR_C420_T
#undef MPCVR_C420
}

static bool Convert420Eligible(const ConvertParams &P)
{
    return P.fmt.layout == LAY_PLANAR && P.fmt.subsampling == 420 && P.fmt.div_w == 2 && P.fmt.div_h == 2 && P.chroma_scaling == 1 &&
           !P.blend_deint && (P.fmt.planes == 2 || P.fmt.planes == 3) && (P.fmt.bytes == 1 || P.fmt.bytes == 2) &&
           P.tail >= TAIL_NONE && P.tail <= TAIL_GAMMA_GAMUT && (P.fmt.bytes == 2 || P.tail == TAIL_NONE) &&    // (8-bit PQ / HLG / BT.2020: the generic kernel)
           (!P.dovi || (P.fmt.bytes == 2 && (P.tail == TAIL_NONE || P.tail == TAIL_PQ_TO_SDR)));
}

hipError_t LaunchConvert(const ConvertParams &P, const Surface &out, hipStream_t s, bool generic)
{
    const StoreParams none{};
    if (!generic && Convert420Eligible(P) && out.fmt == P.out_fmt) {
        if (out.fmt == SF_BGRA8) LaunchConvert420T<SF_BGRA8, 0, 0>(P, out, none, s);
        else if (out.fmt == SF_RGB10A2) LaunchConvert420T<SF_RGB10A2, 0, 0>(P, out, none, s);
        else if (out.fmt == SF_RGBA16F) LaunchConvert420T<SF_RGBA16F, 0, 0>(P, out, none, s);
        else generic = true;
    } else generic = true;
    if (generic) hipLaunchKernelGGL(k_convert, grid2d(P.out_w, P.out_h), dim3(64, 4, 1), 0, s, P, out);
    return hipGetLastError();
}

hipError_t LaunchConvertDirect(const ConvertParams &P, const StoreParams &st, hipStream_t s)
{
    const Surface none{};
    bool done = false;
    if (Convert420Eligible(P) && st.mid_fmt == P.out_fmt && st.quant == (st.dst_fmt == SF_RGB10A2 ? 1023 : 255)) {
        const int o = P.out_fmt, d = st.dst_fmt;
        done = true;
        if (st.mode == ST_FINAL && o == SF_RGB10A2 && d == SF_BGRA8) LaunchConvert420T<SF_RGB10A2, 2, SF_BGRA8>(P, none, st, s);
        else if (st.mode == ST_FINAL && o == SF_RGBA16F && d == SF_BGRA8) LaunchConvert420T<SF_RGBA16F, 2, SF_BGRA8>(P, none, st, s);
        else if (st.mode == ST_FINAL && o == SF_RGBA16F && d == SF_RGB10A2) LaunchConvert420T<SF_RGBA16F, 2, SF_RGB10A2>(P, none, st, s);
        else if (st.mode == ST_SURFACE && o == SF_BGRA8 && d == SF_BGRA8) LaunchConvert420T<SF_BGRA8, 1, SF_BGRA8>(P, none, st, s);
        else if (st.mode == ST_SURFACE && o == SF_RGB10A2 && d == SF_RGB10A2) LaunchConvert420T<SF_RGB10A2, 1, SF_RGB10A2>(P, none, st, s);
        else done = false;
    }
    if (!done) hipLaunchKernelGGL(k_convert_direct, grid2d(P.out_w, P.out_h), dim3(64, 4, 1), 0, s, P, st);
    return hipGetLastError();
}

// folded instantiations: NT in {4, 6, runtime}, INFMT in {UNORM8, UNORM10, fp16}, every EpiCode
template <int NT, int INFMT, int EPI>
static bool LaunchResizeFast(bool rows, const Surface &in, const AxisTaps &taps, const int32_t *other, int out_w, int out_h,
                             const StoreParams &st, hipStream_t s, const ResizeBatch &bt)
{
    // Combinations no plan produces are not instantiated (round 4: every shipped kernel is launched by the suite): the row kernel is the
    // second or only draw and never fills m_TexResize (EPI_TO_FP16); the column kernel reads the convert output or the source texture,
    // which is fp16 only with the fp16 internal format — never in front of a 10-bit post-scale texture (EPI_FINAL_10_TO_8)
    if (rows) {
        if constexpr (EPI != EPI_TO_FP16) {
            constexpr int PX = 2;        // 1 and 4 measured slower on MI355X (1440p Lanczos3: 335 / 279 / 303 us per 16 frames)
            const int gx = (out_w + 256 * PX - 1) / (256 * PX), grid = (gx * ((out_h + 7) / 8) * bt.n + 7) / 8 * 8;     // RW = 8 rows per workgroup
            const size_t lds = (NT && RowWindowPays(taps, NT)) ? (size_t)4 * taps.blk8_span * PX * 64 * sizeof(uint2) : 0;
            hipLaunchKernelGGL((k_resize_rows<NT, INFMT, EPI, PX>), dim3(grid, 1, 1), dim3(256, 1, 1), lds, s, in, taps, out_w, out_h, gx, st, bt);
            return true;
        }
    } else if constexpr (!(INFMT == SF_RGBA16F && EPI == EPI_FINAL_10_TO_8)) {
        hipLaunchKernelGGL((k_resize_cols<NT, INFMT, EPI>), dim3((out_w + 63) / 64, (out_h + 15) / 16, bt.n), dim3(64, 4, 1),
                           (size_t)4 * 4 * ((taps.blk_span + 3) & ~3) * sizeof(float4), s, in, taps, other, out_w, out_h, st, bt);
        return true;
    }
    return false;           // not built: the caller runs the one-kernel-fits-all k_resize
}
template <int NT, int INFMT>
static bool LaunchResizeFastE(int epi, bool rows, const Surface &in, const AxisTaps &taps, const int32_t *other, int out_w, int out_h,
                              const StoreParams &st, hipStream_t s, const ResizeBatch &bt)
{
    switch (epi) {
    case EPI_TO_FP16: return LaunchResizeFast<NT, INFMT, EPI_TO_FP16>(rows, in, taps, other, out_w, out_h, st, s, bt);
    case EPI_FINAL_10_TO_8: return LaunchResizeFast<NT, INFMT, EPI_FINAL_10_TO_8>(rows, in, taps, other, out_w, out_h, st, s, bt);
    case EPI_FINAL_16F_TO_8: return LaunchResizeFast<NT, INFMT, EPI_FINAL_16F_TO_8>(rows, in, taps, other, out_w, out_h, st, s, bt);
    case EPI_FINAL_16F_TO_10: return LaunchResizeFast<NT, INFMT, EPI_FINAL_16F_TO_10>(rows, in, taps, other, out_w, out_h, st, s, bt);
    case EPI_TO_BGRA8: return LaunchResizeFast<NT, INFMT, EPI_TO_BGRA8>(rows, in, taps, other, out_w, out_h, st, s, bt);
    case EPI_TO_RGB10: return LaunchResizeFast<NT, INFMT, EPI_TO_RGB10>(rows, in, taps, other, out_w, out_h, st, s, bt);
    default: return false;
    }
}
template <int INFMT>
static bool LaunchResizeFastN(int epi, bool rows, const Surface &in, const AxisTaps &taps, const int32_t *other, int out_w, int out_h,
                              const StoreParams &st, hipStream_t s, const ResizeBatch &bt)
{
    if (taps.ntaps == 4) return LaunchResizeFastE<4, INFMT>(epi, rows, in, taps, other, out_w, out_h, st, s, bt);
    if (taps.ntaps == 6) return LaunchResizeFastE<6, INFMT>(epi, rows, in, taps, other, out_w, out_h, st, s, bt);
    return LaunchResizeFastE<0, INFMT>(epi, rows, in, taps, other, out_w, out_h, st, s, bt);
}

template <int NT, int INFMT>
static bool LaunchResize2DE(int epi, const Surface &in, const AxisTaps &tx, const AxisTaps &ty, const int32_t *other, int mid_h, int out_w, int out_h,
                            const StoreParams &st, hipStream_t s, const ResizeBatch &bt)
{
    const int tiles_x = (out_w + 63) / 64, tiles_y = (out_h + 31) / 32;
    const dim3 grid((tiles_x * tiles_y * bt.n + 7) / 8 * 8, 1, 1), block(256, 1, 1);
    const size_t lds = (size_t)8 * ((tx.blk_span + 3) & ~3) * sizeof(float4) + (size_t)ty.blk32_span * 64 * sizeof(uint2);
#define MPCVR_R2D(E) hipLaunchKernelGGL((k_resize_2d<NT, INFMT, E>), grid, block, lds, s, in, tx, ty, other, mid_h, out_w, out_h, tiles_x, tiles_y, st, bt)
    switch (epi) {
    case EPI_FINAL_10_TO_8: if constexpr (INFMT != SF_RGBA16F) { MPCVR_R2D(EPI_FINAL_10_TO_8); return true; } else return false;     // (an fp16 convert output has no 10-bit post-scale texture behind it)
    case EPI_FINAL_16F_TO_8: MPCVR_R2D(EPI_FINAL_16F_TO_8); return true;
    case EPI_FINAL_16F_TO_10: MPCVR_R2D(EPI_FINAL_16F_TO_10); return true;
    case EPI_TO_BGRA8: MPCVR_R2D(EPI_TO_BGRA8); return true;
    case EPI_TO_RGB10: MPCVR_R2D(EPI_TO_RGB10); return true;
    default: return false;
    }
#undef MPCVR_R2D
}

// the tiled two-draw kernel: first draw = column taps of an unrotated source, second draw = row taps with a 1:1 column map,
// equal tap counts (4 / 6, or run-time), windows that fit LDS
bool Resize2DSupported(const Surface &in, const AxisTaps &tx, const AxisTaps &ty, const StoreParams &st)
{
    const int epi = EpiOf(st);
    if (epi == EPI_RUNTIME || epi == EPI_TO_FP16) return false;
    if (in.fmt != SF_BGRA8 && in.fmt != SF_RGB10A2 && in.fmt != SF_RGBA16F) return false;
    if (!tx.blk_lo || !tx.idx_t || tx.blk_span <= 0 || tx.blk_span > kResizeSpanMax) return false;
    if (!ty.blk32_lo || !ty.other_identity || ty.blk32_span <= 0 || ty.blk32_span > kResizeTileRowsMax) return false;
    return true;          // equal tap counts of 4 or 6 get the unrolled instantiation, anything else the run-time loops
}

hipError_t LaunchResize2D(const Surface &in, const AxisTaps &tx, const AxisTaps &ty, const int32_t *other, int mid_h, int out_w, int out_h,
                          const StoreParams &st, hipStream_t s, const ResizeBatch *batch)
{
    if (!Resize2DSupported(in, tx, ty, st)) return hipErrorNotSupported;
    const ResizeBatch one{}, &bt = batch ? *batch : one;
    const int epi = EpiOf(st);
    const int nt = (tx.ntaps == ty.ntaps && (tx.ntaps == 4 || tx.ntaps == 6)) ? tx.ntaps : 0;
    bool done = false;
#define MPCVR_R2D_F(F) (nt == 4 ? LaunchResize2DE<4, F>(epi, in, tx, ty, other, mid_h, out_w, out_h, st, s, bt) \
                       : nt == 6 ? LaunchResize2DE<6, F>(epi, in, tx, ty, other, mid_h, out_w, out_h, st, s, bt) \
                                 : LaunchResize2DE<0, F>(epi, in, tx, ty, other, mid_h, out_w, out_h, st, s, bt))
    if (in.fmt == SF_BGRA8) done = MPCVR_R2D_F(SF_BGRA8);
    else if (in.fmt == SF_RGB10A2) done = MPCVR_R2D_F(SF_RGB10A2);
    else done = MPCVR_R2D_F(SF_RGBA16F);
#undef MPCVR_R2D_F
    return done ? hipGetLastError() : hipErrorNotSupported;
}

// taps along screen y with a 1:1 column map -> row kernel; taps along screen x with a block table -> column kernel
static bool FoldedRows(int axis, bool swap, const AxisTaps &taps) { return !swap && axis == 1 && taps.other_identity; }
static bool FoldedCols(int axis, bool swap, const AxisTaps &taps)
{
    return !swap && axis == 0 && taps.blk_lo && taps.idx_t && taps.blk_span > 0 && taps.blk_span <= kResizeSpanMax;
}
bool ResizeHasFoldedKernel(int axis, bool swap, const Surface &in, const AxisTaps &taps, const StoreParams &st)
{
    return (FoldedRows(axis, swap, taps) || FoldedCols(axis, swap, taps)) && EpiOf(st) != EPI_RUNTIME &&
           (in.fmt == SF_BGRA8 || in.fmt == SF_RGB10A2 || in.fmt == SF_RGBA16F);
}

hipError_t LaunchResize(int axis, bool swap, const Surface &in, const AxisTaps &taps, const int32_t *other,
                        int out_w, int out_h, const StoreParams &st, hipStream_t s, bool generic, const ResizeBatch *batch)
{
    if (!generic && ResizeHasFoldedKernel(axis, swap, in, taps, st)) {
        const bool rows = FoldedRows(axis, swap, taps);
        const int epi = EpiOf(st);
        const ResizeBatch one{}, &bt = batch ? *batch : one;
        bool done = false;
        if (in.fmt == SF_BGRA8) done = LaunchResizeFastN<SF_BGRA8>(epi, rows, in, taps, other, out_w, out_h, st, s, bt);
        else if (in.fmt == SF_RGB10A2) done = LaunchResizeFastN<SF_RGB10A2>(epi, rows, in, taps, other, out_w, out_h, st, s, bt);
        else if (in.fmt == SF_RGBA16F) done = LaunchResizeFastN<SF_RGBA16F>(epi, rows, in, taps, other, out_w, out_h, st, s, bt);
        if (done) return hipGetLastError();
    }
    // the one-kernel-fits-all version (quarter turns, tap tables without a block structure): a frame dimension like the folded ones
    const ResizeBatch one{}, &gb = batch ? *batch : one;
    dim3 g = grid2d(out_w, out_h);
    g.z = (unsigned)gb.n;
    const dim3 b(64, 4, 1);
    if (axis == 0 && !swap) hipLaunchKernelGGL((k_resize<0, false>), g, b, 0, s, in, taps, other, out_w, out_h, st, gb);
    else if (axis == 0)     hipLaunchKernelGGL((k_resize<0, true>), g, b, 0, s, in, taps, other, out_w, out_h, st, gb);
    else if (!swap)         hipLaunchKernelGGL((k_resize<1, false>), g, b, 0, s, in, taps, other, out_w, out_h, st, gb);
    else                    hipLaunchKernelGGL((k_resize<1, true>), g, b, 0, s, in, taps, other, out_w, out_h, st, gb);
    return hipGetLastError();
}

hipError_t LaunchHdr10ToneMap(const Surface &in, const HdrToneMapParams &tm, int out_w, int out_h, const StoreParams &st, hipStream_t s, const ResizeBatch *batch)
{
    const ResizeBatch bt = batch ? *batch : ResizeBatch{};
    dim3 g = grid2d(out_w, out_h);
    g.z = (unsigned)bt.n;
    hipLaunchKernelGGL(k_hdr10_tonemap, g, dim3(64, 4, 1), 0, s, in, tm, out_w, out_h, st, bt);
    return hipGetLastError();
}

// the phase table of k_jinc2_phases for this draw, or false when the draw is not dyadic / is rotated or mirrored
bool BuildJincPhases(const DrawCoords &dc, void *out_table)
{
    JincPhases &t = *(JincPhases *)out_table;
    auto period = [](float step) { return step == 1.0f ? 1 : step == 0.5f ? 2 : step == 0.25f ? 4 : 0; };
    t.px = period(dc.step_x); t.py = period(dc.step_y);
    if (!t.px || !t.py || dc.swap || dc.rev_x || dc.rev_y) return false;
    // The phase kernels take an output pixel's tap base from org + (o + 0.5) * step; the shader takes it from Tex * wh — the interpolated
    // texture coordinate (TexCenter), which can sit one ulp beside that.  At 2x / 4x the positions k + 0.25 ... are far from the floor's
    // step and the ulp changes nothing; on a 1:1 axis (the OTHER axis of a two-draw Jinc2m, always) the position is k + 0.5 — exactly ON the
    // step: one ulp low and the shader's 4 x 4 window sits a texel further left (found by the Jinc2m mode of tests/tools/fuzz_strip.py: 4 columns
    // of an 88-wide frame, up to 53 codes).  So: every output index is checked here, and a draw with one such index keeps the per-pixel kernel.
    auto axis_ok = [](int org, int len, int tex, int n, int rev, float step) {
        for (int i = 0; i < n; i++) {
            const float pc = TexCenter(org, len, tex, i, n, rev), nominal = (float)org + ((float)i + 0.5f) * step;
            if (floorf(pc - 0.5f) != floorf(nominal - 0.5f)) return false;
        }
        return true;
    };
    if (!axis_ok(dc.org_x, dc.len_x, dc.tex_x, dc.n_x, dc.rev_x, dc.step_x) || !axis_ok(dc.org_y, dc.len_y, dc.tex_y, dc.n_y, dc.rev_y, dc.step_y)) return false;
    const float pi = 3.14159274101257324f, wa = 0.416f * pi, wb = 0.985f * pi;
    for (int py = 0; py < 4; py++)
        for (int px = 0; px < 4; px++) {
            // the shader's expressions at output (px, py) of a draw starting at the origin (integer origins drop out exactly)
            const float pcx = ((float)(px % t.px) + 0.5f) * dc.step_x, pcy = ((float)(py % t.py) + 0.5f) * dc.step_y;
            const float tcx = floorf(pcx - 0.5f) + 0.5f, tcy = floorf(pcy - 0.5f) + 0.5f;
            float wsum = 0.0f;
            for (int j = 0; j < 4; j++) {
                float rowsum = 0.0f;
                for (int i = 0; i < 4; i++) {
                    const float vx = (tcx + (float)(i - 1)) - pcx, vy = (tcy + (float)(j - 1)) - pcy;
                    const float dd = sqrtf(vx * vx + vy * vy);
                    const float w = (dd == 0.0f) ? wa * wb : crm_sinf(dd * wa) * crm_sinf(dd * wb) / (dd * dd);
                    t.w[py][px][j * 4 + i] = w;
                    rowsum = i == 0 ? w : rowsum + w;
                }
                wsum = j == 0 ? rowsum : wsum + rowsum;
            }
            t.wsum[py][px] = wsum;
        }
    return true;
}
size_t JincPhasesBytes() { return sizeof(JincPhases); }

// Tex * wh of the draw's output columns [0, n_x) and rows [n_x, n_x + n_y) for the plain Jinc2m kernel: TexCenter, once per index
void BuildDrawCentres(const DrawCoords &dc, float *out)
{
    for (int x = 0; x < dc.n_x; x++) out[x] = TexCenter(dc.org_x, dc.len_x, dc.tex_x, x, dc.n_x, dc.rev_x);
    for (int y = 0; y < dc.n_y; y++) out[dc.n_x + y] = TexCenter(dc.org_y, dc.len_y, dc.tex_y, y, dc.n_y, dc.rev_y);
}

hipError_t LaunchJinc2(const Surface &in, const DrawCoords &dc, int out_w, int out_h, const StoreParams &st, hipStream_t s, const void *phases_dev, bool fast,
                       const ResizeBatch *batch, const float *centres_dev)
{
    // exact 2x on both axes, default tier: a 2x2 output quad per lane (vp_jinc.hip)
    if (fast && phases_dev && Jinc2QuadSupported(in, dc, out_w, out_h, st)) return LaunchJinc2Quad(in, dc, out_w, out_h, st, s, phases_dev, batch);
    const ResizeBatch one{}, &gb = batch ? *batch : one;
    dim3 g = grid2d(out_w, out_h);
    g.z = (unsigned)gb.n;
    if (phases_dev) {
        hipLaunchKernelGGL(k_jinc2_phases, g, dim3(64, 4, 1), 0, s, in, dc, (const JincPhases *)phases_dev, out_w, out_h, st, gb);
        return hipGetLastError();
    }
    if (!centres_dev || out_w > dc.n_x || out_h > dc.n_y) return hipErrorInvalidValue;       // the plain kernel reads its texcoords from the table
    hipLaunchKernelGGL(k_jinc2, g, dim3(64, 4, 1), 0, s, in, dc, centres_dev, out_w, out_h, st, gb);
    return hipGetLastError();
}

hipError_t LaunchCorrection(int kind, const Surface &in, const Surface &out, const float fix16[16], const float gamut9[9], float lum_scale, hipStream_t s)
{
    CorrParams K;
    std::memcpy(K.fix, fix16, sizeof(K.fix)); std::memcpy(K.gamut, gamut9, sizeof(K.gamut)); K.lum_scale = lum_scale;
    const dim3 g = grid2d(out.w, out.h), b(64, 4, 1);
    switch (kind) {
    case 1: hipLaunchKernelGGL(k_correction<1>, g, b, 0, s, in, out, K); break;
    case 2: hipLaunchKernelGGL(k_correction<2>, g, b, 0, s, in, out, K); break;
    case 3: hipLaunchKernelGGL(k_correction<3>, g, b, 0, s, in, out, K); break;
    case 4: hipLaunchKernelGGL(k_correction<4>, g, b, 0, s, in, out, K); break;
    case 5: hipLaunchKernelGGL(k_correction<5>, g, b, 0, s, in, out, K); break;
    case 6: hipLaunchKernelGGL(k_correction<6>, g, b, 0, s, in, out, K); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t LaunchCopy(const Surface &in, int out_w, int out_h, const StoreParams &st, hipStream_t s)
{
    hipLaunchKernelGGL(k_copy, grid2d(out_w, out_h), dim3(64, 4, 1), 0, s, in, out_w, out_h, st);
    return hipGetLastError();
}


// ---- verification aid (mpcvr_eval_dovi_tail): the plain tier's Dolby Vision tail, stage by stage, over an array of PQ-coded RGB triples ----
// stage 0: PQ EOTF -> LMS matrix -> PQ OETF (Shaders.cpp:844-859); 1: + saturate + level-2 trims (:870-877); 2: + ST2084ToLinear(., scale);
// 3: + Hable / hable(4.8); 4: + 2020 -> 709; 5: + saturate, pow 1/2.2 (the whole tail).  Compiled here so that it IS the plain kernels' arithmetic.
struct DoviTailEval { float lms[9], k[5], gamut[9], lum_scale; int stage, l2; };
__global__ __launch_bounds__(256) void k_eval_dovi_tail(const float *__restrict__ rgb, float *__restrict__ out, size_t n, DoviTailEval P)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f3 c = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        c.x = st2084_to_linear(fmaxf(c.x, 0.0f), 1.0f); c.y = st2084_to_linear(fmaxf(c.y, 0.0f), 1.0f); c.z = st2084_to_linear(fmaxf(c.z, 0.0f), 1.0f);
        f3 r;
        r.x = P.lms[0] * c.x + P.lms[1] * c.y + P.lms[2] * c.z;
        r.y = P.lms[3] * c.x + P.lms[4] * c.y + P.lms[5] * c.z;
        r.z = P.lms[6] * c.x + P.lms[7] * c.y + P.lms[8] * c.z;
        c.x = linear_to_st2084(fmaxf(r.x, 0.0f), 1.0f); c.y = linear_to_st2084(fmaxf(r.y, 0.0f), 1.0f); c.z = linear_to_st2084(fmaxf(r.z, 0.0f), 1.0f);
        if (P.stage >= 1) {
            c.x = saturate(c.x); c.y = saturate(c.y); c.z = saturate(c.z);
            if (P.l2) c = dovi_trims(c, P.k);
        }
        if (P.stage >= 2) { c.x = st2084_to_linear(c.x, P.lum_scale); c.y = st2084_to_linear(c.y, P.lum_scale); c.z = st2084_to_linear(c.z, P.lum_scale); }
        if (P.stage >= 3) { const float div = hable_div(); c.x = hable(c.x) / div; c.y = hable(c.y) / div; c.z = hable(c.z) / div; }
        if (P.stage >= 4) c = mat3_mul(make_mat3(P.gamut), c);
        if (P.stage >= 5) { c.x = hlsl_pow(saturate(c.x), 1.0f / 2.2f); c.y = hlsl_pow(saturate(c.y), 1.0f / 2.2f); c.z = hlsl_pow(saturate(c.z), 1.0f / 2.2f); }
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}
hipError_t LaunchEvalDoviTail(const float *rgb_dev, float *out_dev, size_t n, const float lms[9], const float k[5], const float gamut[9], int l2, float lum_scale, int stage, hipStream_t s)
{
    DoviTailEval P{};
    std::memcpy(P.lms, lms, sizeof(P.lms)); std::memcpy(P.k, k, sizeof(P.k));
    std::memcpy(P.gamut, gamut, sizeof(P.gamut));
    P.lum_scale = lum_scale; P.stage = stage; P.l2 = l2;
    hipLaunchKernelGGL(k_eval_dovi_tail, dim3(1024), dim3(256), 0, s, rgb_dev, out_dev, n, P);
    return hipGetLastError();
}

}  // namespace mpcvr
