// vp_errdiff.hip — the error-diffusion final pass (EXTENSION, bUseDither = 2): R10G10B10A2 frames as a 10-bit swap chain would receive
// them -> B8G8R8A8 render targets, Floyd-Steinberg in integers.  Definition and schedule: vp_errdiff_core.h (no reference counterpart;
// the serial model oracle/mpcvr_oracle.c orc_error_diffusion is its only check, and the kernel must equal it bit for bit).
//
// One wavefront per band of 64 rows, one workgroup per wavefront, (frames x bands) workgroups per launch, all free-running:
//   * lane i = row i of the band, two columns behind lane i - 1; what a row hands to the row below (D, three channels) moves by ONE
//     DPP wave shift per step and channel — no LDS, no barrier anywhere in the kernel;
//   * the bottom row of a band publishes its D as tagged words in a device-memory hand-off row (relaxed agent-scope atomic stores: value
//     and tag in one word, so there is nothing to order), the 24 words of a group of 8 steps in ONE store of lanes 0-23; lane 0 of the band
//     below — another workgroup, usually on another XCD — fetches them one group of 8 columns ahead (lanes 0-23 load a group at once) and
//     spins, politely and with a bound, only when the band above has not written them yet; the launcher zeroes the rows first;
//   * a lane reads its row 32 pixels (one cache line's worth, eight 16-byte loads) a block of 32 steps ahead and writes 8-byte pairs; no
//     memory instruction of the loop is predicated (clamped addresses, a dummy slot for lanes off the region), so the waits the compiler inserts are exact.
// Round 4's first version ran a frame in ONE workgroup (16 waves taking turns behind workgroup barriers, hand-off rows in LDS): 32 of
// 256 CUs busy on a 32-frame batch, 630 frames/s at 4K -> 8K.  The pass is a chain of W + 2 H dependent steps per frame with ~30 integer
// instructions per channel and pixel: bound by VALU issue and by its own serial depth, not by HBM.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "vp_errdiff_core.h"
#include "vp_launch.h"

namespace mpcvr {

namespace {

typedef uint32_t ed_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t ed_u4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) uint8_t *ed_gcptr;
typedef __attribute__((address_space(1))) uint8_t *ed_gptr;

constexpr int kEdOccupancyLds = 0;             // dynamic LDS claimed per workgroup (never touched): 160 KiB / that = workgroups per CU
constexpr int kEdSpinLimit = 1 << 21;       // polls (~1 us each) a band grants the band above before it gives up and flags the launch

// the value of lane - 1; lane 0 — the band's top row, which has no lane above — gets `top` (wave-uniform: the band above's D of this column).
// DPP: `top` rides in as the instruction's old-value operand, which a lane without a source keeps (v_mov + v_mov_dpp; the select
// behind a zero-initialised shift was four instructions per channel and step)
template <int SHIFT>
__device__ __forceinline__ int32_t ed_from_lane_above(int32_t v, int32_t top, int lane)
{
    if (SHIFT == 0) return __builtin_amdgcn_update_dpp(top, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const int32_t d = __builtin_amdgcn_ds_bpermute(((lane - 1) & 63) << 2, v);
    return lane == 0 ? top : d;
}

// PAIR: every pixel pair (even column, odd column) of the region is whole (x0 and x1 even) and every target row starts on an 8-byte
// boundary: one 8-byte store per pair.  Every vector memory instruction of the loop is UNCONDITIONAL — lanes off the region read a
// clamped address and write into a dummy slot — so that the compiler can count them: its s_waitcnt for the pixel pairs fetched a group
// ago then leaves this group's stores and fetches in flight (with predicated accesses it must assume vmcnt(0) at every wait: the first
// cut of this kernel waited for its own prefetch, 670 cycles per step)
template <int SHIFT, bool PAIR>
__global__ __launch_bounds__(64) void k_error_diffusion(ErrDiffParams P, const FusedFrame *__restrict__ frames, FusedFrame single)
{
    const EdSchedule S = ed_schedule(P.x0, P.x1, P.y1 - P.y0);
    const int lane = threadIdx.x;
    // workgroup -> (frame, band); either way a band's producer has a lower index.  order 0: a frame's bands are neighbours (neighbouring
    // workgroups = neighbouring phases of one frame); 1: a band of every frame, then the next band (neighbours = the same phase of all frames)
    const int nfr = (int)gridDim.x / S.bands;
    const int z = P.order ? (int)blockIdx.x % nfr : (int)blockIdx.x / S.bands;
    const int band = P.order ? (int)blockIdx.x / nfr : (int)blockIdx.x - z * S.bands;
    const FusedFrame fr = frames ? frames[z] : single;
    const int a0 = P.x0 & ~1;
    const int rows = P.y1 - P.y0;
    const int r = band * kEdRows + lane;
    const bool row_ok = r < rows;
    const int y = P.y0 + (row_ok ? r : rows - 1);
    const ed_gcptr src_row = (ed_gcptr)fr.src + (size_t)y * (size_t)P.src_pitch + (size_t)a0 * 4u;
    const ed_gptr dst_row = (ed_gptr)fr.dst + (size_t)y * (size_t)P.dst_pitch + (size_t)a0 * 4u;
    uint32_t *const mine = P.handoff + ((size_t)z * S.bands + band) * (size_t)S.stride;
    const bool has_above = band > 0;
    const uint32_t *const above = has_above ? mine - S.stride : mine;                    // (the first band reads its own row: zeros, ignored)
    // where lanes off the region store: slots of this band's own row (all bands writing one shared dummy meant every wavefront of the launch
    // pushing write-through stores at the same 64 bytes: 32 frames took eight times as long as with predicated stores)
    uint32_t *const spare = mine + 3 * kEdGroup * S.groups;                              // the spare group: 24 words nobody waits for
    const ed_gptr dummy = (ed_gptr)(spare + 3 * kEdGroup) + (size_t)lane * 8u;
    const int xr_last = (S.wl - 1) & ~1;                                                 // last even column of the region

    EdChannel st[3];
    int32_t dprev[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { st[c] = EdChannel{0, 0, 0, 0}; dprev[c] = 0; }

    // A block = 32 steps = the next 32 pixels of the lane's row (one cache line's worth), fetched as eight 16-byte loads a block ahead:
    // the eight loads of a lane hit one or two lines back to back.  (8-byte loads a group of 8 steps ahead, the first cut, had every
    // load instruction of a wavefront touch 64 lines that the CU's other wavefronts had evicted since the last visit: 16x the useful bytes
    // from L2, and a 32-frame batch ran at a third of the single-frame rate.)  The loads are dword-aligned (xr is even, not a multiple of
    // four in odd lanes); off the region they read a clamped position: the caller's image has a margin of two pixels in front of every
    // row it does not own and slack behind the last one.
    constexpr int BLK = kEdGroup * kEdBlockGroups;
    auto load_block = [&](int tb, ed_u4 (&v)[BLK / 4]) __attribute__((always_inline)) {
        const int xs = tb - kEdSkew * lane;
#pragma unroll
        for (int p = 0; p < BLK / 4; p++) {
            const int xr = min(max(xs + 4 * p, -2), xr_last);
            v[p] = *(const __attribute__((address_space(1))) ed_u4 *)(src_row + (ptrdiff_t)xr * 4);
        }
    };
    // the 24 hand-off words of a group (columns t0 .. t0 + 7, word 3 column + channel): lane l < 24 fetches word l (the others: word 0)
    const int wlane = lane < 3 * kEdGroup ? lane : 0;
    auto fetch_above = [&](int t0) __attribute__((always_inline)) -> uint32_t {
        return __hip_atomic_load(above + 3 * t0 + wlane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    ed_u4 cur[BLK / 4], nxt[BLK / 4];
    load_block(0, cur);
    uint32_t wnext = fetch_above(0);
    uint32_t outw = 0;              // lanes 0-23: the 24 hand-off words the previous group's steps produced
    const bool full_band = (band + 1) * kEdRows <= rows;
    // one block of 32 steps; ALL: every lane stands on a pixel of the region at every step of the block (wave-uniform, true for all but the
    // first four and the last block or two of a full band) — no live test, no select of e, no dummy addresses
    auto run_block = [&](auto ALLC, int blk) __attribute__((always_inline)) {
        constexpr bool ALL = decltype(ALLC)::value;
#pragma unroll
        for (int gi = 0; gi < kEdBlockGroups; gi++) {
            const int t0 = BLK * blk + kEdGroup * gi;
            // publish the previous group's words (lane 63's D of columns t0 - 135 .. t0 - 128): value and tag in one word, ONE store of lanes 0-23
            {
                const int col0 = t0 - kEdGroup - (kEdSkew * (kEdRows - 1) + 1);
                const int col = col0 + lane / 3;
                const bool pub = lane < 3 * kEdGroup && col >= 0 && col < S.wl;
                uint32_t *at = pub ? mine + 3 * col0 + lane : spare + (lane & 15);
                __hip_atomic_store(at, outw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // D of the band above for the eight columns of this group: fetched a group ago; wait for what is not there yet
            uint32_t w = wnext;
            if (has_above) {
                const bool need = lane < 3 * kEdGroup && t0 + lane / 3 < S.wl;      // (columns beyond the region are never written, nor used)
                int spins = 0;
                while (__ballot(need && !(w & 1u)) != 0) {                           // wave-uniform
                    if (++spins > kEdSpinLimit) { if (lane == 0) *P.status = 1; break; }
                    __builtin_amdgcn_s_sleep(4);
                    w = fetch_above(t0);
                }
            }
            wnext = fetch_above(t0 + kEdGroup);                                      // (the row has a spare group of entries behind the last one)
            if (!has_above) w = 0;                                                   // (the frame's first band: nothing comes down; untag(0) = 0)
            uint32_t even_px = 0;
            bool even_live = false;
#pragma unroll
            for (int s = 0; s < kEdGroup; s++) {
                const int sb = kEdGroup * gi + s;                                    // step inside the block
                const int xr = t0 + s - kEdSkew * lane;
                const bool live = ALL || (row_ok && xr >= S.lead && xr < S.wl);
                const uint32_t code = cur[sb >> 2][sb & 3];
                int q[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int32_t top = ed_untag((uint32_t)__builtin_amdgcn_readlane((int)w, 3 * s + c));      // (scalar)
                    const int32_t din = ed_from_lane_above<SHIFT>(dprev[c], top, lane);
                    q[c] = ed_step(st[c], live, (int)((code >> (10 * c)) & 0x3ffu), din, dprev[c]);
                }
                const uint32_t px = ed_pack_bgra(q[0], q[1], q[2]);                  // (ed_step answers q + 16: the three biases leave in one subtraction)
                if ((s & 1) == 0) { even_px = px; even_live = live; }
                else {
                    const ed_gptr at = dst_row + (ptrdiff_t)(xr - 1) * 4;
                    if (PAIR) *(__attribute__((address_space(1))) ed_u2 *)(ALL || live ? at : dummy) = ed_u2{even_px, px};      // (whole pairs: live == even_live)
                    else {
                        *(__attribute__((address_space(1))) uint32_t *)(ALL || even_live ? at : dummy) = even_px;
                        *(__attribute__((address_space(1))) uint32_t *)(ALL || live ? at + 4 : dummy + 4) = px;
                    }
                }
                // the band's bottom row: D(xr - 1) for the band below — lane 63's value travels through a scalar into lane 3 s + c of outw
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const uint32_t word = ed_tag(__builtin_amdgcn_readlane(dprev[c], kEdRows - 1));
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(outw) : "s"(word), "n"(3 * s + c));
                }
            }
        }
    };
    for (int blk = 0; blk < S.groups / kEdBlockGroups; blk++) {
        load_block(BLK * (blk + 1), nxt);                                            // (clamped: the block behind the last one reads the row's end again)
        const int tb = BLK * blk;
        if (full_band && tb - kEdSkew * (kEdRows - 1) >= S.lead && tb + BLK - 1 < S.wl) run_block(std::true_type{}, blk);
        else run_block(std::false_type{}, blk);
#pragma unroll
        for (int p = 0; p < BLK / 4; p++) cur[p] = nxt[p];
    }
    // the last group's words: columns up to groups * 8 - 128 >= wl - 1
    {
        const int col0 = kEdGroup * S.groups - kEdGroup - (kEdSkew * (kEdRows - 1) + 1);
        const int col = col0 + lane / 3;
        if (lane < 3 * kEdGroup && col >= 0 && col < S.wl) __hip_atomic_store(mine + 3 * col0 + lane, outw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

size_t ErrorDiffusionHandoffBytes(const ErrDiffParams &P, int n_frames)
{
    const EdSchedule S = ed_schedule(P.x0, P.x1, P.y1 - P.y0);
    return (size_t)n_frames * S.bands * S.stride * sizeof(uint32_t);
}

hipError_t LaunchErrorDiffusion(const ErrDiffParams &P, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s)
{
    if (n_frames <= 0 || (!frames_dev && n_frames != 1)) return hipErrorInvalidValue;
    if (P.x1 <= P.x0 || P.y1 <= P.y0 || !P.handoff || !P.status) return hipErrorInvalidValue;
    const EdSchedule S = ed_schedule(P.x0, P.x1, P.y1 - P.y0);
    // zero = "not written yet": the hand-off rows are cleared in front of every launch (a few MB per frame, in stream order)
    const hipError_t e = hipMemsetAsync(P.handoff, 0, ErrorDiffusionHandoffBytes(P, n_frames), s);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)((size_t)n_frames * S.bands)), block(64);
    // Bands in flight per SIMD.  A band is a chain of dependent steps and a frame is a chain of bands: a wavefront that shares its SIMD with
    // three busy ones runs at a quarter of its speed and so does everything behind it, so the pass wants FEW resident wavefronts, each at
    // full speed — the kernel claims LDS it never touches to hold the occupancy down (MPCVR_ERRDIFF_LDS overrides, bytes; A/B in DESIGN.md)
    static const size_t lds = [] { const char *e = std::getenv("MPCVR_ERRDIFF_LDS"); return e ? (size_t)std::atol(e) : (size_t)kEdOccupancyLds; }();
    const bool pair = P.pair_stores && !(P.x0 & 1) && !(P.x1 & 1);
    if (P.shift == 1) {
        if (pair) hipLaunchKernelGGL((k_error_diffusion<1, true>), grid, block, lds, s, P, frames_dev, single);
        else hipLaunchKernelGGL((k_error_diffusion<1, false>), grid, block, lds, s, P, frames_dev, single);
    } else {
        if (pair) hipLaunchKernelGGL((k_error_diffusion<0, true>), grid, block, lds, s, P, frames_dev, single);
        else hipLaunchKernelGGL((k_error_diffusion<0, false>), grid, block, lds, s, P, frames_dev, single);
    }
    return hipGetLastError();
}

}  // namespace mpcvr
