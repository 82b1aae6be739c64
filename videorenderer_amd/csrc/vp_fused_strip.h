// vp_fused_strip.h — device side of the arbitrary-ratio fused kernel; instantiated per tap count by vp_fused_strip_nt{4,6,8,16}.hip,
// launched by vp_fused_strip.hip (preconditions, work decomposition).
//
// The reference runs every geometry through the same draws (ConvertColorPass -> m_TexConvertOutput, TextureResizeShader X ->
// fp16 m_TexResize, TextureResizeShader Y -> m_TexsPostScale, FinalPass; DX11VideoProcessor.cpp:3103-3187,3285-3424).  The
// exact-2x kernel (vp_fused.hip) lives on its two fixed phases; here the per-output tap tables of BuildAxisTaps drive the
// same wave-autonomous strip design for any ratio, up or down, with every intermediate rounding of the reference kept:
//   convert output -> UNORM8/10, X draw -> fp16 (RNE), Y draw -> UNORM8/10 (m_TexsPostScale), final pass floor(p*Q + d).
//
// One wavefront owns a strip of `strip_w` output columns (PXL adjacent pixels per lane) and a segment of output rows, and
// marches down the SOURCE rows two at a time (a 4:2:0 row pair shares its two chroma rows), no workgroup barrier in the loop:
//   stage C  lane j converts the 2x2 blocks {cols c0+2j(+128..), +1} x {rows 2p-1, 2p} of the strip's source window from raw
//            codes (prefetched one pair ahead), rounds them to the internal UNORM format and parks the CODES in this wave's
//            LDS slice A as fp16 bit patterns: an integer k < 1024 IS the fp16 subnormal k * 2^-24, which v_fma_mix_f32
//            reads exactly (tools/ubench/strip_probe.hip) — no int->float conversion anywhere, 8 bytes per texel
//   stage X  lane l filters its PXL output columns of both rows: per tap one ds_read_b64 at the lane's own (tap-table)
//            offset + three v_fma_mix_f32 (fp16 operand x fp32 weight, the 2^24/maxv scale folded into the weight);
//            the fp16-rounded results (m_TexResize) enter an LDS ring window ring[row & mask][lane] — private to the lane,
//            so no barrier; LDS as a register file with a run-time (wave-uniform) row index
//   stage Y  for every output row whose source rows are in the ring: taps and weights are wave-uniform (scalar loads from
//            the row-major table), per tap one LDS read for the lane's PXL pixels + v_fma_mix_f32 with an SGPR weight;
//            then m_TexsPostScale rounding + ps_final_pass in integers (vp_fused.hip's epilogue) or the generic epilogue
//            (store_epilogue: any target format, window clipping, post-scale textures)
// HBM traffic = the source window once (+ halo rows per segment) + the render target once: nothing in between.
#pragma once
#include "vp_fused_dev.h"

namespace mpcvr {

struct StripArgs {
    // PlanFusedStrip's copies of the two tap tables: NT taps per output (zero-weight padding), ps_convolution's
    // normalisation folded into the weights
    const int32_t *xi_t; const float *xw_t;     // X taps, tap-major [NT][out_w]
    const int32_t *yi; const float *yw;         // Y taps, row-major [out_h][NT]
    const int32_t *yrange;       // [out_h][2] {smallest, largest} source row any tap of output row y reads
    const int32_t *xstrip;       // [n_strips][2] {smallest, largest} source column any tap of the strip reads
    int out_w, out_h;
    int n_strips, strip_w;       // output columns per strip (<= 64 * PXL, a multiple of PXL)
    int ring_mask;               // ring rows - 1 (8, 16 or 32 rows)
    int seg_rows;                // output rows per segment
    int acols;                   // columns of an A row (even)
    float a_scale;               // 2^24 / maxv: A holds integer codes as fp16 subnormals (1 for an fp16 surface: real halfs)
    // SRC_SURFACE: the X draw samples a surface (m_TexConvertOutput of any convert kernel, or the source texture of an
    // interleaved RGB sample) instead of converting raw YUV: its format, pitch, width (clamp), the row map of the draw
    // (row of m_TexResize -> surface row; null = identity) and the distance between the frames of a batch
    const uint8_t *surf; const int32_t *other;
    int surf_fmt, surf_pitch, surf_w;
    size_t surf_stride;
};

namespace {

// v_fma_mix_f32: fp16 operand (lo / hi half of a dword) x fp32 weight + fp32 accumulator
template <bool HI, bool SW, bool CLAMP = false>
__device__ __forceinline__ float fmix(uint32_t h, float w, float acc)
{
    float r;
    if (CLAMP) {        // the last tap of the Y draw saturates (UNORM store of m_TexsPostScale): SGPR weight
        if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(h), "s"(w), "v"(acc));
        else    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(h), "s"(w), "v"(acc));
    } else if (SW) {
        if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(w), "v"(acc));
        else    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(w), "v"(acc));
    } else {
        if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(w), "v"(acc));
        else    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(w), "v"(acc));
    }
    return r;
}
template <bool HI, bool SW>
__device__ __forceinline__ float fmix0(uint32_t h, float w)
{
    float r;
    if (SW) {
        if (HI) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(w));
        else    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(w));
    } else {
        if (HI) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(w));
        else    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(w));
    }
    return r;
}
// one texel {r | g << 16, b in the lo (or, BHI, hi) half of bx} (fp16 bit patterns) times weight w into acc[3]
template <bool SW, bool FIRST, bool BHI = false, bool CLAMP = false>
__device__ __forceinline__ void tap3(uint32_t rg, uint32_t bx, float w, float (&acc)[3])
{
    if (FIRST) { acc[0] = fmix0<false, SW>(rg, w); acc[1] = fmix0<true, SW>(rg, w); acc[2] = fmix0<BHI, SW>(bx, w); }
    else { acc[0] = fmix<false, SW, CLAMP>(rg, w, acc[0]); acc[1] = fmix<true, SW, CLAMP>(rg, w, acc[1]); acc[2] = fmix<BHI, SW, CLAMP>(bx, w, acc[2]); }
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// The tap tables are read-only for the whole launch and read at wave-uniform addresses: through the constant address space
// the loads become s_load (SGPR results, scalar cache).  Left as plain global pointers the compiler must assume the kernel's
// own stores could alias them and issues per-lane vector loads — the row's taps and weights then arrive in VGPRs and every
// ring address costs a v_mul_lo_u32.
template <typename T> using cptr = const __attribute__((address_space(4))) T *;
template <typename T> __device__ __forceinline__ cptr<T> as_const(const T *p) { return (cptr<T>)(uintptr_t)p; }

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// NT: taps per output on both axes (4, 6, 8 or 16; the host pads shorter tables with zero weights)
template <int NT, int PXL, int TAIL, int SRC, int EPI, int XC>
__device__ __forceinline__ void fused_strip_body(const FusedArgs &P, const StripArgs &Q, StoreParams st, const FusedFrame *__restrict__ frames, const FusedFrame &single)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool FASTEPI = EPI == EPI_DITHER8;
    uint32_t *Di = (uint32_t *)smem;                                   // dither as j << 14 (FASTEPI)
    f2 *T = (f2 *)(smem + (FASTEPI ? LDS_DB : 0));
    unsigned char *wbase = smem + (FASTEPI ? LDS_DB : 0) + (tail_has_table(TAIL) ? LDS_T : 0);
    if (FASTEPI)
        for (int i = threadIdx.x; i < 1024; i += blockDim.x)
            Di[i] = (uint32_t)(__half2float(__ushort_as_half(P.dither[i])) * 1024.0f + 0.5f) << 14;
    if (tail_has_table(TAIL))
        for (int i = threadIdx.x; i < LUT_N; i += blockDim.x) {
            const float v = P.lut[i], n = P.lut[min(i + 1, LUT_N - 1)];
            T[i] = f2{v, n - v};
        }
    if (FASTEPI || tail_has_table(TAIL)) __syncthreads();             // the only workgroup barrier: tables visible

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // work items (strip, segment) are dealt to the waves of the grid in one sequence, neighbouring waves = neighbouring strips of one segment
    int bx, bz;
    xcd_contiguous_block(bx, bz);                                     // (workgroup -> (item group, frame): an XCD works on neighbours, vp_fused_dev.h)
    const int item = bx * (int)(blockDim.x >> 6) + wave;
    const int seg_i = item / Q.n_strips, strip = item - seg_i * Q.n_strips;
    const int y0 = seg_i * Q.seg_rows;
    if (y0 >= Q.out_h) return;
    const int y1 = min(y0 + Q.seg_rows, Q.out_h);
    const int W = P.W, H = P.H;
    const int ring_rows = Q.ring_mask + 1;
    // ring row: per lane {r|g, b|-} (PXL = 1, 8 B) or {r0|g0, r1|g1, b0|b1} (PXL = 2, 12 B): one address per tap
    constexpr int ring_row = PXL == 2 ? 64 * 12 : 64 * 8;
    const int a_row = Q.acols * 8;
    unsigned char *const Aw = wbase + wave * (2 * a_row + ring_rows * ring_row);
    // the lane's ring address as an integer the compiler cannot split into "register + large constant": ds_read2_b32 has no room
    // for a large immediate, and a second v_add per tap to add the constant part is what it would cost
    typedef __attribute__((address_space(3))) uint32_t *lds_u32;
    const uint32_t ring_addr = opaque((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(Aw + 2 * a_row + lane * (PXL == 2 ? 12 : 8)));

    const FusedFrame frame = frames ? frames[bz] : single;
    auto uniform_ptr = [](const void *q) {
        const uint64_t v = (uint64_t)q;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const gcptr py = (gcptr)uniform_ptr(SRC == SRC_SURFACE && Q.surf ? (const void *)(Q.surf + (size_t)bz * Q.surf_stride) : (const void *)frame.src);
    const uint64_t dst_u = uniform_ptr(frame.dst);
    const gptr pdst = (gptr)dst_u;
    st.dst = (void *)dst_u;

    // the strip's source window: columns c0 .. hi as 2x2 blocks, 64 per pass
    const int c0 = as_const(Q.xstrip)[2 * strip] & ~1;
    const int nb = ((as_const(Q.xstrip)[2 * strip + 1] - c0) >> 1) + 1, npass = (nb + 63) >> 6;

    // stage X / Y role: output columns xs + PXL*lane + q
    const int xs = strip * Q.strip_w;
    const int x_first = xs + PXL * lane;
    const bool xy_active = PXL * lane < Q.strip_w && x_first < Q.out_w;
    uint32_t xo[PXL][NT]; float xw[PXL][NT];
#pragma unroll
    for (int q = 0; q < PXL; q++) {
        const int xc = min(x_first + q, Q.out_w - 1);
#pragma unroll
        for (int k = 0; k < NT; k++) {
            xo[q][k] = (uint32_t)(Q.xi_t[xc + (size_t)k * Q.out_w] - c0) * 16u;
            xw[q][k] = Q.xw_t[xc + (size_t)k * Q.out_w] * Q.a_scale;
        }
    }

    const f2 MM[5] = {f2{P.m[0], P.m[1]}, f2{P.m[2], P.m[3]}, f2{P.m[4], P.m[5]}, f2{P.m[6], P.m[7]}, f2{P.m[8], 0.0f}};
    const f2 GG[5] = {f2{P.gamut[0], P.gamut[1]}, f2{P.gamut[2], P.gamut[3]}, f2{P.gamut[4], P.gamut[5]}, f2{P.gamut[6], P.gamut[7]}, f2{P.gamut[8], 0.0f}};
    const f2 cmax2 = splat(P.maxv);
    f2 big2 = splat(8388608.0f);                     // 2^23, pinned in VGPRs (see unorm_round2)
    asm volatile("" : "+v"(big2));
    f2 CC[3] = {splat(P.c[0]), splat(P.c[1]), splat(P.c[2])};
    asm volatile("" : "+v"(CC[0]), "+v"(CC[1]), "+v"(CC[2]));

    // raw codes of pass 0 are prefetched one row pair ahead
    constexpr int YSRC = SRC == SRC_SURFACE ? SRC_GENERIC : SRC;      // (the YUV helpers are not instantiated for a surface)
    RawAddr ra0;
    if (SRC != SRC_SURFACE) make_raw_addr<YSRC>(P, min(c0 + 2 * lane, W - 2), ra0);
    Raw rawn;
    auto fetch = [&](int pp, const RawAddr &ra, Raw &r) {
        load_raw<YSRC>(P, py, ra, clampi(2 * pp - 1, 0, H - 1), clampi(2 * pp, 0, H - 1), r);
    };
    // SRC_SURFACE: the 2x2 texels of block b of pair pp, [row][column], as stored (one dword; two for fp16)
    u32x2 sraw[2][2];
    auto fetch_s = [&](int pp, int b, u32x2 (&t)[2][2]) {
        const cptr<int32_t> other = as_const(Q.other);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int row = clampi(2 * pp - 1 + r, 0, H - 1);
            const gcptr rowp = py + (uint32_t)(Q.other ? other[row] : row) * (uint32_t)Q.surf_pitch;
#pragma unroll
            for (int col = 0; col < 2; col++) {
                const uint32_t x = (uint32_t)min(c0 + 2 * b + col, Q.surf_w - 1);
                if (Q.surf_fmt == SF_RGBA16F) t[r][col] = *(const __attribute__((address_space(1))) u32x2 *)(rowp + x * 8u);
                else t[r][col] = u32x2{ld_u32(rowp + x * 4u), 0u};
            }
        }
    };
    // texel -> {r | g << 16, b}: UNORM codes (fp16 subnormals for stage X) or the halfs of an fp16 texel
    auto unpack_s = [&](u32x2 t, uint32_t &rg, uint32_t &bb) {
        if (Q.surf_fmt == SF_RGBA16F) { rg = t.x; bb = t.y & 0xffffu; }
        else if (Q.surf_fmt == SF_RGB10A2) { rg = (t.x & 0x3ffu) | ((t.x << 6) & 0x03ff0000u); bb = (t.x >> 20) & 0x3ffu; }
        else { rg = __builtin_amdgcn_perm(0u, t.x, 0x0c010c02u); bb = t.x & 0xffu; }       // B8G8R8A8: r = byte 2, g = byte 1
    };

    // pair pp = source rows 2pp-1, 2pp (rect-relative; the first and the last pair of a frame hold one useful row)
    auto produce = [&](int pp) {
        const int r0 = 2 * pp - 1;
        const int sy0 = P.rect_t + clampi(r0, 0, H - 1), sy1 = P.rect_t + clampi(r0 + 1, 0, H - 1);
        for (int pass = 0; pass < npass; pass++) {
            const int b = pass * 64 + lane;
            if (SRC == SRC_SURFACE) {       // no convert stage: the texels are m_TexConvertOutput's (or the source texture's) own codes
                u32x2 t[2][2];
                if (pass == 0) {
#pragma unroll
                    for (int r = 0; r < 2; r++) { t[r][0] = sraw[r][0]; t[r][1] = sraw[r][1]; }
                    fetch_s(pp + 1, lane, sraw);
                } else fetch_s(pp, b, t);
                uint32_t rg[2][2], bb[2][2];            // [column][row]
#pragma unroll
                for (int col = 0; col < 2; col++)
#pragma unroll
                    for (int r = 0; r < 2; r++) unpack_s(t[r][col], rg[col][r], bb[col][r]);
                if (b < nb) {
                    *(u32x4 *)(Aw + 32 * b) = u32x4{rg[0][0], bb[0][0], rg[0][1], bb[0][1]};
                    *(u32x4 *)(Aw + 32 * b + 16) = u32x4{rg[1][0], bb[1][0], rg[1][1], bb[1][1]};
                }
                continue;
            }
            f2 rc[2][3];
            if (pass == 0) {
                convert_block<TAIL, YSRC, DV_NONE, XC, XC == XC_ALWAYS ? OUT_CODE_I : OUT_NORM>(P, MM, GG, CC, rawn, sy0, sy1, T, rc);
                fetch(pp + 1, ra0, rawn);
            } else {
                RawAddr ra; Raw rw;
                make_raw_addr<YSRC>(P, min(c0 + 2 * b, W - 2), ra);
                fetch(pp, ra, rw);
                convert_block<TAIL, YSRC, DV_NONE, XC, XC == XC_ALWAYS ? OUT_CODE_I : OUT_NORM>(P, MM, GG, CC, rw, sy0, sy1, T, rc);
            }
            // store to m_TexConvertOutput (UNORM: floor(sat(x)*maxv + 0.5)): x*maxv + 2^23 leaves the code in the low
            // mantissa bits — as an fp16 bit pattern that code is the subnormal k * 2^-24
            uint32_t rg[2][2], bb[2][2];                // [column][row]
#pragma unroll
            for (int col = 0; col < 2; col++) {
                // (the exact form hands over the codes as integers: the same low halves)
                const f2 qr = XC == XC_ALWAYS ? rc[col][0] : pk_fma(rc[col][0], cmax2, big2), qg = XC == XC_ALWAYS ? rc[col][1] : pk_fma(rc[col][1], cmax2, big2),
                         qb = XC == XC_ALWAYS ? rc[col][2] : pk_fma(rc[col][2], cmax2, big2);
                rg[col][0] = __builtin_amdgcn_perm(__float_as_uint(qg.x), __float_as_uint(qr.x), 0x05040100u);
                rg[col][1] = __builtin_amdgcn_perm(__float_as_uint(qg.y), __float_as_uint(qr.y), 0x05040100u);
                bb[col][0] = __float_as_uint(qb.x) & 0xffffu; bb[col][1] = __float_as_uint(qb.y) & 0xffffu;
            }
            if (b < nb) {           // A[column] = {row 0 texel, row 1 texel}: a tap of stage X is one 16-byte read for both rows
                *(u32x4 *)(Aw + 32 * b) = u32x4{rg[0][0], bb[0][0], rg[0][1], bb[0][1]};
                *(u32x4 *)(Aw + 32 * b + 16) = u32x4{rg[1][0], bb[1][0], rg[1][1], bb[1][1]};
            }
        }
        wave_sync();
        // ---------------- stage X ----------------
        if (xy_active) {
            float acc[2][PXL][3];                                // [row][pixel][channel]
#pragma unroll
            for (int q = 0; q < PXL; q++) {
                u32x4 t[NT];
#pragma unroll
                for (int k = 0; k < NT; k++) t[k] = *(const u32x4 *)(Aw + xo[q][k]);
                tap3<false, true>(t[0].x, t[0].y, xw[q][0], acc[0][q]);
                tap3<false, true>(t[0].z, t[0].w, xw[q][0], acc[1][q]);
#pragma unroll
                for (int k = 1; k < NT; k++) {
                    tap3<false, false>(t[k].x, t[k].y, xw[q][k], acc[0][q]);
                    tap3<false, false>(t[k].z, t[k].w, xw[q][k], acc[1][q]);
                }
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int row = r0 + r;
                if (row < 0 || row >= H) continue;               // wave-uniform
                // m_TexResize is R16G16B16A16_FLOAT (:3155): round to fp16 (RNE)
                const lds_u32 dstp = (lds_u32)(uintptr_t)(ring_addr + (uint32_t)((row & Q.ring_mask) * ring_row));
                const h2v h0 = __builtin_convertvector(f2{acc[r][0][0], acc[r][0][1]}, h2v);
                dstp[0] = __builtin_bit_cast(uint32_t, h0);
                if (PXL == 2) {
                    const h2v h1 = __builtin_convertvector(f2{acc[r][PXL - 1][0], acc[r][PXL - 1][1]}, h2v), hb = __builtin_convertvector(f2{acc[r][0][2], acc[r][PXL - 1][2]}, h2v);
                    dstp[1] = __builtin_bit_cast(uint32_t, h1); dstp[2] = __builtin_bit_cast(uint32_t, hb);
                } else {
                    const h2v hb = __builtin_convertvector(f2{acc[r][0][2], 0.0f}, h2v);
                    dstp[1] = __builtin_bit_cast(uint32_t, hb);
                }
            }
        }
        wave_sync();
    };

    // ---------------- the march ----------------
    const cptr<int32_t> yrange = as_const(Q.yrange);
    int p = (yrange[2 * y0] + 1) >> 1;
    int have = 2 * p - 2;                                // largest source row in the ring
    if (SRC == SRC_SURFACE) fetch_s(p, lane, sraw); else fetch(p, ra0, rawn);
    const uint32_t lane_off = (uint32_t)(P.off_x + x_first) * 4u;
    const bool st8 = PXL == 2 && ((P.off_x + xs) & 1) == 0 && (((uintptr_t)dst_u | (uintptr_t)P.dst_pitch) & 7) == 0;
    const bool dpair = PXL == 2 && ((P.off_x + xs) & 1) == 0;            // the lane's two dither texels are one aligned 8-byte read
    const float maxv = P.maxv;
    for (int y = y0; y < y1; y++) {
        const int hi = yrange[2 * y + 1];
        while (have < hi) { produce(p); have = 2 * p; p++; }
        if (!xy_active) continue;
        // ---------------- stage Y + final pass ----------------
        const cptr<int32_t> yi = as_const(Q.yi) + (size_t)y * NT;
        const cptr<float> yw = as_const(Q.yw) + (size_t)y * NT;
        float acc[PXL][3];
        // the last tap saturates in the FMA itself (a zero-weight padding tap saturates just the same)
        constexpr bool CLAMPED = true;
        uint32_t dj[PXL];
        if (FASTEPI) {          // dither texels first: the LDS round trip hides behind the taps
            const uint32_t *drow = Di + ((P.off_y + y) & 31) * 32;
            if (PXL == 2 && dpair) {
                const u32x2 dd = *(const u32x2 *)(drow + ((P.off_x + x_first) & 31));
                dj[0] = dd.x; dj[PXL - 1] = dd.y;
            } else {
#pragma unroll
                for (int q = 0; q < PXL; q++) dj[q] = drow[(P.off_x + x_first + q) & 31];
            }
        }
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const int slot_off = (yi[k] & Q.ring_mask) * ring_row;
            const float w = yw[k];
            const lds_u32 tp = (lds_u32)(uintptr_t)(ring_addr + (uint32_t)slot_off);
            const uint32_t t0 = tp[0], t1 = tp[1];
            if (PXL == 2) {
                const uint32_t tb = tp[PXL];
                if (k == 0) { tap3<true, true, false>(t0, tb, w, acc[0]); tap3<true, true, true>(t1, tb, w, acc[PXL - 1]); }
                else if (CLAMPED && k == NT - 1) { tap3<true, false, false, true>(t0, tb, w, acc[0]); tap3<true, false, true, true>(t1, tb, w, acc[PXL - 1]); }
                else { tap3<true, false, false>(t0, tb, w, acc[0]); tap3<true, false, true>(t1, tb, w, acc[PXL - 1]); }
            } else {
                if (k == 0) tap3<true, true>(t0, t1, w, acc[0]);
                else if (CLAMPED && k == NT - 1) tap3<true, false, false, true>(t0, t1, w, acc[0]);
                else tap3<true, false>(t0, t1, w, acc[0]);
            }
        }
        if (EPI == EPI_DIRECT8) {
            // no post-scale step: the Y result is stored straight into the target's UNORM format (B8G8R8A8 or R10G10B10A2),
            // floor(x*q + 0.5).  x*q + 2^23 leaves the code in the low mantissa bits
            const int wy = P.off_y + y;
            const float qd = P.out10 ? 1023.0f : 255.0f;
            const f2 q2 = splat(qd);
            uint32_t pk[PXL];
            uint32_t codes[PXL][3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (PXL == 2) {
                    const f2 u = pk_fma(f2{acc[0][c], acc[PXL - 1][c]}, q2, big2);
                    codes[0][c] = __float_as_uint(u.x); codes[PXL - 1][c] = __float_as_uint(u.y);
                } else {
                    codes[0][c] = __float_as_uint(fmaf(acc[0][c], qd, 8388608.0f));
                }
            }
#pragma unroll
            for (int q = 0; q < PXL; q++) {
                if (P.out10) {      // 0x4B000000 | k: shifted left by 10 or 20 only k remains; + 0x75000000 turns the red code into k | 3 << 30
                    uint32_t t = codes[q][0] + 0x75000000u;
                    t = (codes[q][1] << 10) | t;
                    pk[q] = (codes[q][2] << 20) | t;
                } else {
                    const uint32_t bg = __builtin_amdgcn_perm(codes[q][1], codes[q][2], 0x0c0c0400u);     // [B, G, 0, 0]
                    pk[q] = __builtin_amdgcn_perm(codes[q][0], bg, 0x0d040100u);                         // [B, G, R, 0xff]
                }
            }
            const gptr rowp = pdst + (uint32_t)wy * (uint32_t)P.dst_pitch;
            if (PXL == 2 && st8 && x_first + 1 < Q.out_w) {
                *(__attribute__((address_space(1))) u32x2 *)(rowp + opaque(lane_off)) = u32x2{pk[0], pk[PXL - 1]};
            } else {
#pragma unroll
                for (int q = 0; q < PXL; q++)
                    if (x_first + q < Q.out_w) *(__attribute__((address_space(1))) uint32_t *)(rowp + lane_off + 4 * q) = pk[q];
            }
        } else if (FASTEPI) {
            // m_TexsPostScale store/load + ps_final_pass.hlsl:29 in integers, see vp_fused.hip
            const int wy = P.off_y + y;
            uint32_t pk[PXL];
            uint32_t codes[PXL][3];
            if (PXL == 2) {         // x*maxv + 2^23 for the pixel pair of a channel in one packed FMA
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    f2 v = f2{acc[0][c], acc[PXL - 1][c]};
                    if (!CLAMPED) v = f2{__builtin_amdgcn_fmed3f(v.x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(v.y, 0.0f, 1.0f)};
                    const f2 u = pk_fma(v, cmax2, big2);
                    codes[0][c] = __float_as_uint(u.x); codes[PXL - 1][c] = __float_as_uint(u.y);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float v = CLAMPED ? acc[0][c] : __builtin_amdgcn_fmed3f(acc[0][c], 0.0f, 1.0f);
                    codes[0][c] = __float_as_uint(fmaf(v, maxv, 8388608.0f));
                }
            }
#pragma unroll
            for (int q = 0; q < PXL; q++) {
                const uint32_t *code = codes[q];
                const uint32_t djq = dj[q];
                const uint32_t ib = __umul24(code[2], P.epi_mul) + djq, ig = __umul24(code[1], P.epi_mul) + djq, ir = __umul24(code[0], P.epi_mul) + djq;
                const uint32_t bg = __builtin_amdgcn_perm(ig, ib, 0x0c0c0703u);    // [B, G, 0, 0]
                pk[q] = __builtin_amdgcn_perm(ir, bg, 0x0d070100u);               // [B, G, R, 0xff]
            }
            const gptr rowp = pdst + (uint32_t)wy * (uint32_t)P.dst_pitch;
            if (PXL == 2 && st8 && x_first + 1 < Q.out_w) {
                *(__attribute__((address_space(1))) u32x2 *)(rowp + opaque(lane_off)) = u32x2{pk[0], pk[PXL - 1]};
            } else {
#pragma unroll
                for (int q = 0; q < PXL; q++)
                    if (x_first + q < Q.out_w) *(__attribute__((address_space(1))) uint32_t *)(rowp + lane_off + 4 * q) = pk[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < PXL; q++)
                if (x_first + q < Q.out_w) store_epilogue(st, x_first + q, y, f3{acc[q][0], acc[q][1], acc[q][2]});
        }
    }
}

template <int NT, int PXL, int TAIL, int SRC, int EPI, int XC = XC_NEVER>
__global__ __launch_bounds__(1024) void k_fused_strip(FusedArgs P, StripArgs Q, StoreParams st, const FusedFrame *__restrict__ frames, FusedFrame single)
{
    fused_strip_body<NT, PXL, TAIL, SRC, EPI, XC>(P, Q, st, frames, single);
}
// the kernel of an instantiation: its exact-form twin where one exists and the launch asks for it (exact_capable, vp_fused_dev.h;
// SRC_SURFACE has no convert stage and no twin)
template <int NT, int PXL, int TAIL, int SRC, int EPI>
inline auto fused_strip_kernel(bool exact) -> decltype(&k_fused_strip<NT, PXL, TAIL, SRC, EPI, XC_NEVER>)
{
    if constexpr (exact_capable<TAIL, SRC, EPI == EPI_DITHER8>() == XC_RUNTIME) { if (exact) return k_fused_strip<NT, PXL, TAIL, SRC, EPI, XC_ALWAYS>; }
    return k_fused_strip<NT, PXL, TAIL, SRC, EPI, XC_NEVER>;
}

}  // namespace

// which epilogue a launch runs (vp_fused_strip.hip decides): the integer final pass, the straight UNORM store, or store_epilogue
enum { STRIP_EPI_FAST = 0, STRIP_EPI_DIRECT = 1, STRIP_EPI_GENERIC = 2 };

// per-tap-count launcher, instantiated by vp_fused_strip_nt*.hip: every (pixels per lane, tail, source, epilogue) combination the planner
// can pick for NT taps: two pixels per lane up to 8 taps, one for NT = 16 (9..16-tap downscales: ps_convolution beyond ~2x with bicubic /
// Lanczos).  PlanFusedStrip's cost model never chose a one-pixel strip below 9 taps (its X / Y stages cost the same per lane and the
// convert pass is shared by half as many outputs), so those 189 kernels are not built; a plan that asks for one is refused, not skipped.
template <int NT>
hipError_t LaunchFusedStripNT(const FusedArgs &a, const StripArgs &q, const StoreParams &st, int pxl, int tailk, int srck, int epi, bool surface_mode,
                              dim3 grid, dim3 block, size_t lds, const FusedFrame *frames_dev, FusedFrame single, hipStream_t s)
{
#define MPCVR_ST5(PX, TK, SK, EK) do { \
        auto kern = fused_strip_kernel<NT, PX, TK, SK, EK>(a.exact_cv != 0); \
        if (lds > 48 * 1024) { \
            const hipError_t ea = AllowLargeLds((const void *)kern, lds); \
            if (ea != hipSuccess) return ea; \
        } \
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, q, st, frames_dev, single); } while (0)
#define MPCVR_ST4(PX, TK, SK) do { if (epi == STRIP_EPI_FAST) MPCVR_ST5(PX, TK, SK, EPI_DITHER8); else if (epi == STRIP_EPI_DIRECT) MPCVR_ST5(PX, TK, SK, EPI_DIRECT8); \
                                   else MPCVR_ST5(PX, TK, SK, EPI_GENERIC); } while (0)
#define MPCVR_ST3(PX, TK) do { if (srck == SRC_P01X) MPCVR_ST4(PX, TK, SRC_P01X); else if (srck == SRC_PLANAR16) MPCVR_ST4(PX, TK, SRC_PLANAR16); \
                               else MPCVR_ST4(PX, TK, SRC_GENERIC); } while (0)
    // (the 8-bit loaders exist without a tail only: FusedSourceKind)
#define MPCVR_ST2(PX) do { if (surface_mode) MPCVR_ST4(PX, TAILK_NONE, SRC_SURFACE); \
                           else if (tailk == TAILK_NONE && srck == SRC_NV12) MPCVR_ST4(PX, TAILK_NONE, SRC_NV12); \
                           else if (tailk == TAILK_NONE && srck == SRC_PLANAR8) MPCVR_ST4(PX, TAILK_NONE, SRC_PLANAR8); \
                           else if (tailk == TAILK_NONE) MPCVR_ST3(PX, TAILK_NONE); else if (tailk == TAILK_PQ_LUT) MPCVR_ST3(PX, TAILK_PQ_LUT); \
                           else if (tailk == TAILK_HLG) MPCVR_ST3(PX, TAILK_HLG); else MPCVR_ST3(PX, TAILK_ALU); } while (0)
    if constexpr (NT <= 8) {
        if (pxl != 2) return hipErrorNotSupported;
        MPCVR_ST2(2);
    } else {
        if (pxl != 1) return hipErrorNotSupported;
        MPCVR_ST2(1);
    }
#undef MPCVR_ST2
#undef MPCVR_ST3
#undef MPCVR_ST4
#undef MPCVR_ST5
    return hipGetLastError();
}

}  // namespace mpcvr
