// vp_errdiff.hip — the error-diffusion final pass (EXTENSION, bUseDither = 2): R10G10B10A2 frames as a 10-bit swap chain would receive
// them -> B8G8R8A8 render targets, Floyd-Steinberg in integers.  Definition and schedule: vp_errdiff_core.h (no reference counterpart;
// the serial model oracle/mpcvr_oracle.c orc_error_diffusion is its only check, and the kernel must equal it bit for bit).
//
// One workgroup per frame, 16 wavefronts, each running a band of 64 rows with a two-column skew from lane to lane:
//   * what a row hands to the row below (D, three channels) moves by ONE DPP wave shift per step and channel — no LDS, no barrier;
//   * the bottom row of a band parks its D in an LDS row buffer, read by lane 0 of the band below two slots (= two workgroup barriers)
//     later; the buffer is rewritten in place, 127 columns behind its reader;
//   * a lane reads its row as 8-byte pixel pairs one group of 8 steps ahead and writes 8-byte pairs.
// The pass is a chain of W + 2 H dependent steps per frame with ~30 integer instructions per channel and pixel: it is bound by VALU
// issue and by its own serial depth, not by HBM (DESIGN.md §4.6 has the numbers); frames of a batch are what fills the chip.
#include <hip/hip_runtime.h>

#include "vp_errdiff_core.h"
#include "vp_launch.h"

namespace mpcvr {

hipError_t AllowLargeLds(const void *kern, size_t lds);      // vp_fused_strip.hip
size_t DeviceLdsLimit();

namespace {

typedef uint32_t ed_u2 __attribute__((ext_vector_type(2)));
typedef int32_t ed_i4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) uint8_t *ed_gcptr;
typedef __attribute__((address_space(1))) uint8_t *ed_gptr;

// the value of lane - 1 (lane 0: anything — it reads the row buffer instead)
template <int SHIFT>
__device__ __forceinline__ int32_t ed_from_lane_above(int32_t v, int lane)
{
    if (SHIFT == 0) return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    return __builtin_amdgcn_ds_bpermute(((lane - 1) & 63) << 2, v);
}

// LDS: int32 rowbuf[3][brw], brw = slots_per_band * kEdChunk + 8 (every step of lane 0 has its own entry: no clamping in the loop)
__host__ __device__ inline int ed_rowbuf_stride(const EdSchedule &S) { return S.slots_per_band * kEdChunk + 8; }

template <int SHIFT>
__global__ __launch_bounds__(kEdWaves * 64) void k_error_diffusion(ErrDiffParams P, const FusedFrame *__restrict__ frames, FusedFrame single)
{
    extern __shared__ __attribute__((aligned(16))) int32_t ed_rowbuf[];
    const EdSchedule S = ed_schedule(P.x0, P.x1, P.y1 - P.y0);
    const int brw = ed_rowbuf_stride(S);
    for (int i = threadIdx.x; i < 3 * brw; i += blockDim.x) ed_rowbuf[i] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const FusedFrame fr = frames ? frames[blockIdx.x] : single;
    const int a0 = P.x0 & ~1;
    const int rows = P.y1 - P.y0;

    EdChannel st[3];
    int32_t dprev[3];
    ed_gcptr src_row = nullptr;
    ed_gptr dst_row = nullptr;
    bool row_ok = false;

    for (int slot = 0; slot < S.total_slots; slot++) {
        int band = 0, chunk = 0;
        if (ed_slot_work(S, wave, slot, &band, &chunk)) {        // wave-uniform
            if (chunk == 0) {
#pragma unroll
                for (int c = 0; c < 3; c++) { st[c] = EdChannel{0, 0, 0, 0}; dprev[c] = 0; }
                const int r = band * kEdRows + lane;
                row_ok = r < rows;
                const int y = P.y0 + (row_ok ? r : rows - 1);
                src_row = (ed_gcptr)fr.src + (size_t)y * (size_t)P.src_pitch + (size_t)a0 * 4u;
                dst_row = (ed_gptr)fr.dst + (size_t)y * (size_t)P.dst_pitch + (size_t)a0 * 4u;
            }
            const int tbase = chunk * kEdChunk;
            const bool has_above = band > 0;
            // pixel pairs of a group of 8 steps: xr = t - 2 lane is even on the even step of a pair in every lane
            auto load_group = [&](int t0, ed_u2 (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int xr = t0 + 2 * p - kEdSkew * lane;
                    v[p] = ed_u2{0u, 0u};
                    if (row_ok && xr >= 0 && xr < S.wl) v[p] = *(const __attribute__((address_space(1))) ed_u2 *)(src_row + (size_t)xr * 4u);
                }
            };
            ed_u2 cur[4], nxt[4];
            load_group(tbase, cur);
            for (int g = 0; g < kEdChunk / 8; g++) {
                const int t0 = tbase + 8 * g;
                if (g + 1 < kEdChunk / 8) load_group(t0 + 8, nxt);
                // lane 0: D of the band above for the eight columns of this group (the band above finished them a slot ago)
                int32_t top[3][8];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    ed_i4 a = ed_i4{0, 0, 0, 0}, b = ed_i4{0, 0, 0, 0};
                    if (has_above && lane == 0) {
                        const ed_i4 *q = (const ed_i4 *)(ed_rowbuf + c * brw + t0);      // t0 is a multiple of 8, brw of 8: 16-byte aligned
                        a = q[0]; b = q[1];
                    }
                    top[c][0] = a.x; top[c][1] = a.y; top[c][2] = a.z; top[c][3] = a.w;
                    top[c][4] = b.x; top[c][5] = b.y; top[c][6] = b.z; top[c][7] = b.w;
                }
                uint32_t even_px = 0;
                bool even_live = false;
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const int xr = t0 + s - kEdSkew * lane;
                    const bool live = row_ok && xr >= S.lead && xr < S.wl;
                    const uint32_t code = (s & 1) ? cur[s >> 1].y : cur[s >> 1].x;
                    int q[3];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        int32_t din = ed_from_lane_above<SHIFT>(dprev[c], lane);
                        if (lane == 0) din = top[c][s];
                        q[c] = ed_step(st[c], live, (int)((code >> (10 * c)) & 0x3ffu), din, dprev[c]);
                    }
                    const uint32_t px = 0xff000000u | ((uint32_t)q[0] << 16) | ((uint32_t)q[1] << 8) | (uint32_t)q[2];     // B8G8R8A8: R = byte 2
                    if ((s & 1) == 0) { even_px = px; even_live = live; }
                    else {
                        const ed_gptr at = dst_row + (size_t)(xr - 1) * 4u;
                        if (P.pair_stores && even_live && live) *(__attribute__((address_space(1))) ed_u2 *)at = ed_u2{even_px, px};
                        else {
                            if (even_live) *(__attribute__((address_space(1))) uint32_t *)at = even_px;
                            if (live) *(__attribute__((address_space(1))) uint32_t *)(at + 4) = px;
                        }
                    }
                    // the band's bottom row: D(xr - 1) for the band below
                    if (lane == 63 && xr >= 1) {
#pragma unroll
                        for (int c = 0; c < 3; c++) ed_rowbuf[c * brw + (xr - 1)] = dprev[c];
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; p++) cur[p] = nxt[p];
            }
        }
        __syncthreads();
    }
}

}  // namespace

size_t ErrorDiffusionLdsBytes(const ErrDiffParams &P)
{
    const EdSchedule S = ed_schedule(P.x0, P.x1, P.y1 - P.y0);
    return (size_t)3 * ed_rowbuf_stride(S) * sizeof(int32_t);
}

bool ErrorDiffusionSupported(const ErrDiffParams &P)
{
    return P.x1 > P.x0 && P.y1 > P.y0 && ErrorDiffusionLdsBytes(P) <= DeviceLdsLimit();
}

hipError_t LaunchErrorDiffusion(const ErrDiffParams &P, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s)
{
    if (n_frames <= 0 || (!frames_dev && n_frames != 1)) return hipErrorInvalidValue;
    if (!ErrorDiffusionSupported(P)) return hipErrorInvalidValue;
    const size_t lds = ErrorDiffusionLdsBytes(P);
    auto launch = [&](auto kern) -> hipError_t {
        if (lds > 48 * 1024) {
            const hipError_t e = AllowLargeLds((const void *)kern, lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)n_frames), dim3(kEdWaves * 64), lds, s, P, frames_dev, single);
        return hipGetLastError();
    };
    return P.shift == 1 ? launch(k_error_diffusion<1>) : launch(k_error_diffusion<0>);
}

}  // namespace mpcvr
