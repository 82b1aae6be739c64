// vp_errdiff.hip — the error-diffusion final pass (EXTENSION, bUseDither = 2): R10G10B10A2 frames as a 10-bit swap chain would receive
// them -> B8G8R8A8 render targets, Floyd-Steinberg in integers.  Definition and schedule: vp_errdiff_core.h (no reference counterpart;
// the serial model oracle/mpcvr_oracle.c orc_error_diffusion is its only check, and the kernel must equal it bit for bit).
//
// One wavefront per band of 21 rows, lane = 21 * channel + row (round 5: one channel per lane — until round 4 a lane carried the three
// independent chains of its row one after the other), one wavefront per workgroup, bands taken by ticket:
//   * the lanes of row i are two columns behind those of row i - 1; what a row hands to the row below (D) moves by ONE DPP wave shift
//     per step (lanes 0, 21, 42 — a row 0 each — take the band above's value instead): no barrier anywhere in the kernel;
//   * pixels travel through LDS: every block of 32 steps a row's next 32 pixels (128 bytes of it) are fetched as 16-byte pieces — each of
//     the row's three lanes fetches a third, a block ahead — and laid into the wavefront's input tile; a step reads its pixel word from the
//     tile (the row's three lanes read the same word) and writes its 8-bit code as ONE BYTE into the output tile, whose rows leave as
//     16-byte pieces at the end of the block: three store instructions per 32 steps instead of sixteen, a row's 128 bytes within a few
//     hundred cycles of each other;
//   * the bottom row publishes its D as tagged words in a device-memory hand-off row (relaxed agent-scope atomic stores: value and tag
//     in one word, nothing to order), the 24 words of a group of 8 steps gathered through LDS into ONE store of lanes 0-23; the band below —
//     another wavefront, usually on another XCD — fetches them one group ahead and spins, politely and with a bound, only when the band
//     above has not written them yet; the launcher zeroes the rows (and the ticket counter) first;
//   * no global memory instruction of the all-live block body is predicated (a lane without a piece of its own repeats its neighbour's),
//     so the waits the compiler inserts are exact counts.
// Round 4's versions for the record: one workgroup per frame (630 frames/s at 4K -> 8K), one wavefront per band of 64 rows with three
// channels per lane (4.0 k frames/s in 32-frame batches, 2.9 ms a frame, a third of the VALU issue slots busy).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>

#include "vp_errdiff_core.h"
#include "vp_launch.h"

namespace mpcvr {

namespace {

typedef uint32_t ed_u4 __attribute__((ext_vector_type(4), aligned(4)));          // (global pieces are dword aligned, no more is promised)
typedef uint32_t ed_l4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) uint8_t *ed_gcptr;
typedef __attribute__((address_space(1))) uint8_t *ed_gptr;

constexpr int kEdTileRow = 144;                      // bytes of a tile row: 32 pixels + a pad that keeps the 21 rows off each other's banks, 16-byte aligned
constexpr int kEdTile = kEdRows * kEdTileRow;        // 3024
constexpr int kEdLdsIn = 0, kEdLdsOut = kEdTile, kEdLdsTop = 2 * kEdTile, kEdLdsHand = kEdLdsTop + 2 * 32 * 4, kEdLdsDummy = kEdLdsHand + 96;
constexpr int kEdLds = kEdLdsDummy + 64 * 4 + 128;   // (a lane's dummy word + the largest immediate offset of a step's hand-off write)

__device__ __forceinline__ void ed_wave_sync()
{
    // LDS instructions of one wavefront execute in order; the fences only keep the COMPILER from moving a lane's read in front of another
    // lane's write (which look unrelated thread by thread)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64) void k_error_diffusion(ErrDiffParams P, const FusedFrame *__restrict__ frames, FusedFrame single, int n_frames)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kEdLds];
    const EdSchedule S = ed_schedule(P.x0, P.x1, P.y1 - P.y0);
    const int lane = threadIdx.x;
    const bool idle = lane == 63;                    // (computes along as a second copy of lane 42's row; its results go to dummy slots)
    const int ch = idle ? 2 : lane / kEdRows, i = idle ? 0 : lane - ch * kEdRows;
    const int rows = P.y1 - P.y0;
    const int total = n_frames * S.bands;
    uint32_t *const ticket = P.handoff + (size_t)total * (size_t)S.stride;
    const uint32_t sh = 10u * (uint32_t)ch;
    // lanes 0, 21, 42 are the band's top row of a channel: the band above's D instead of the shifted one
    const bool row0 = i == 0 && !idle;
    const bool bottom = i == kEdRows - 1;
    const uint32_t gen = (uint32_t)P.gen;

    unsigned char *const in_row = lds + kEdLdsIn + i * kEdTileRow;
    unsigned char *const out_row = lds + kEdLdsOut + i * kEdTileRow;
    unsigned char *const out_byte = idle ? lds + kEdLdsDummy + 4 * lane : out_row + (2 - ch);          // B8G8R8A8: R = byte 2
    uint32_t *const top_word = (uint32_t *)(lds + kEdLdsTop) + ch;                                      // + 3 s: D of column t0 + s, this channel
    uint32_t *const hand_word = bottom ? (uint32_t *)(lds + kEdLdsHand) + ch : (uint32_t *)(lds + kEdLdsDummy) + lane;      // + 3 s
    // the alpha bytes of the output tile are written once: the steps touch bytes 0-2 only
    for (int w = lane; w < kEdTile / 4; w += 64) ((uint32_t *)(lds + kEdLdsOut))[w] = 0xff000000u;

    // the three (or two) 16-byte pieces of its row's 128 bytes a lane fetches and stores: pieces ch, ch + 3, ch + 6 (a lane without a third
    // piece repeats piece 7 — the same bytes at the same address as its neighbour)
    int piece[3];
#pragma unroll
    for (int m = 0; m < 3; m++) piece[m] = min(ch + 3 * m, 7);

    for (;;) {
        int tk = 0;
        if (lane == 0) tk = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk >= total) break;
        int z, band;
        ed_ticket(tk, n_frames, S.bands, P.order, z, band);
        const FusedFrame fr = frames ? frames[z] : single;
        const int r = band * kEdRows + i;
        const bool row_ok = r < rows && !idle;
        const int y = P.y0 + min(r, rows - 1);
        const ed_gcptr src_row = (ed_gcptr)fr.src + (size_t)y * (size_t)P.src_pitch + (size_t)P.x0 * 4u;
        const ed_gptr dst_row = (ed_gptr)fr.dst + (size_t)y * (size_t)P.dst_pitch + (size_t)P.x0 * 4u;
        uint32_t *const mine = P.handoff + ((size_t)z * S.bands + band) * (size_t)S.stride;
        const bool has_above = band > 0;
        const uint32_t *const above = has_above ? mine - S.stride : mine;                    // (the first band reads its own row: zeros, ignored)
        uint32_t *const spare = mine + 3 * kEdGroup * S.groups;                              // the spare group: 24 words nobody waits for
        const bool full_band = (band + 1) * kEdRows <= rows;
        const bool stall = P.test_stall && band == 0 && z == 0;                              // (tests: this band never publishes, the one below gives up)

        EdChannel st{0, 0, 0, 0};
        int32_t dprev = 0;
        uint32_t pend_w = 0; int pend_t0 = 0; bool have_pending = false;      // a group's hand-off words on their way from LDS to the band's row

        // pieces of block tb: columns tb - 2 i + 4 j .. + 3 of the row; a piece that lies outside the region altogether reads a clamped
        // position (the caller's image is readable from two pixels in front of a row of the region to three behind it)
        auto load_block = [&](int tb, ed_u4 (&v)[3]) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < 3; m++) {
                const int xr = min(max(tb - kEdSkew * i + 4 * piece[m], -2), S.wl - 1);
                v[m] = *(const __attribute__((address_space(1))) ed_u4 *)(src_row + (ptrdiff_t)xr * 4);
            }
        };
        auto tile_block = [&](const ed_u4 (&v)[3]) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < 3; m++) *(ed_l4 *)(in_row + 16 * piece[m]) = ed_l4{v[m].x, v[m].y, v[m].z, v[m].w};
        };
        const int wlane = lane < 3 * kEdGroup ? lane : 0;
        // the band above's D, a group of 8 columns = 24 tagged words at a time (word l in lane l), fetched FOUR groups ahead into a ring of
        // four registers: the load crosses to another XCD's memory (a microsecond or two), a group of 8 steps takes 0.7 us.  A group's words
        // are looked at one group before its steps (the steps' LDS reads are issued a group ahead): the band runs as close behind the band
        // above as its words allow, whatever the distance of the prefetch.
        auto fetch_above = [&](int group) __attribute__((always_inline)) -> uint32_t {
            return __hip_atomic_load(above + 3 * kEdGroup * group + wlane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // words of `group` (in w, or fetched again until they carry the launch's tag) -> LDS slot group & 1, untagged.  The first look is
        // straight-line code: a wait in a loop header makes the compiler drain EVERY outstanding memory instruction, the store of the group
        // just published included — a full memory round trip per group.
        auto stage_above = [&](int group, uint32_t w) __attribute__((always_inline)) {
            if (has_above) {
                const bool need = lane < 3 * kEdGroup && kEdGroup * group + lane / 3 < S.wl;      // (columns beyond the region are never written, nor used)
                if (__ballot(need && !ed_tagged(w, gen)) != 0) {                                  // wave-uniform
                    int spins = 0;
                    do {
                        if (++spins > P.spin_limit) { if (lane == 0) *P.status = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                        w = fetch_above(group);
                    } while (__ballot(need && !ed_tagged(w, gen)) != 0);
                }
            } else w = 0;                                                                          // (the frame's first band: nothing comes down; untag(0) = 0)
            if (lane < 32) ((uint32_t *)(lds + kEdLdsTop))[32 * (group & 1) + lane] = (uint32_t)ed_untag(w);      // (lanes 24-31: words nobody reads)
        };

        ed_u4 nxt[3];
        load_block(0, nxt);
        uint32_t wq[4];
#pragma unroll
        for (int j = 0; j < 4; j++) wq[j] = fetch_above(j);
        ed_wave_sync();                              // (the previous band's last reads of the tiles lie in front of these writes)
        tile_block(nxt);
        stage_above(0, wq[0]);
        wq[0] = fetch_above(4);
        ed_wave_sync();
        int32_t tops[kEdGroup];                      // the band above's D for the steps of the group at hand (this lane's channel)
#pragma unroll
        for (int s = 0; s < kEdGroup; s++) tops[s] = (int32_t)top_word[3 * s];

        // one block of 32 steps; ALL: every row stands on a pixel of the region at every step of the block (wave-uniform, true for all but
        // the first two and the last two blocks of a full band) — no live test, no select of e, 16-byte stores
        auto run_block = [&](auto ALLC, int blk) __attribute__((always_inline)) {
            constexpr bool ALL = decltype(ALLC)::value;
            const int tb = kEdBlock * blk;
            load_block(tb + kEdBlock, nxt);                                              // the next block's pieces, a block ahead
            // Every LDS read a step needs — its pixel word and the band above's D for row 0 — is issued a GROUP ahead into registers: a read
            // right in front of its use makes the wavefront sit out an LDS round trip (behind its own two writes of the step before) at every
            // step, which was 40 % of a lone band's time.  pxs / tops: this group's; pxn / topn: the next one's, in flight.
            uint32_t pxs[kEdGroup], pxn[kEdGroup];
            int32_t topn[kEdGroup];
#pragma unroll
            for (int s = 0; s < kEdGroup; s++) pxs[s] = *(const uint32_t *)(in_row + 4 * s);
#pragma unroll
            for (int gi = 0; gi < kEdBlockGroups; gi++) {
                const int t0 = tb + kEdGroup * gi;
                const int G = kEdBlockGroups * blk + gi;
                // the NEXT group's words of the band above: fetched four groups ago -> LDS -> this lane's channel, in flight during this group's steps
                stage_above(G + 1, wq[(gi + 1) & 3]);
                wq[(gi + 1) & 3] = fetch_above(G + 5);                                   // (the row has spare entries behind the last group)
                ed_wave_sync();
#pragma unroll
                for (int s = 0; s < kEdGroup; s++) {
                    topn[s] = (int32_t)top_word[32 * ((gi + 1) & 1) + 3 * s];
                    if (gi + 1 < kEdBlockGroups) pxn[s] = *(const uint32_t *)(in_row + 4 * (kEdGroup * (gi + 1) + s));
                }
#pragma unroll
                for (int s = 0; s < kEdGroup; s++) {
                    const int sb = kEdGroup * gi + s;                                    // step inside the block = position inside the tile row
                    const int xr = t0 + s - kEdSkew * i;
                    const bool live = ALL ? !idle : (row_ok && xr >= 0 && xr < S.wl);
                    const int32_t shifted = __builtin_amdgcn_mov_dpp(dprev, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
                    const int32_t din = row0 ? tops[s] : shifted;
                    const int qb = ed_step(st, ALL || live, (int)((pxs[s] >> sh) & 0x3ffu), din, dprev);
                    out_byte[4 * sb] = (unsigned char)(qb - 16);
                    hand_word[3 * s] = ed_tag(dprev, gen);                               // the bottom row: D(xr - 1) for the band below
                    if (s == 1 && have_pending) {
                        // publish the PREVIOUS group's 24 words (its LDS read was issued at its end, two steps ago): ONE store of lanes 0-23,
                        // value and tag in one word
                        const int col0 = pend_t0 - kEdFlush;
                        const int col = col0 + lane / 3;
                        const bool pub = lane < 3 * kEdGroup && col >= 0 && col < S.wl && !stall;
                        uint32_t *at = pub ? mine + 3 * col0 + lane : spare + (lane & 15);
                        __hip_atomic_store(at, pend_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                ed_wave_sync();
                // this group's 24 words (the bottom row's D of columns t0 - 41 .. t0 - 34) gathered from the lanes that wrote them
                pend_w = ((const uint32_t *)(lds + kEdLdsHand))[wlane];
                pend_t0 = t0; have_pending = true;
                ed_wave_sync();                                                          // (the next group's steps overwrite the words: the read above stays in front of them)
#pragma unroll
                for (int s = 0; s < kEdGroup; s++) { pxs[s] = pxn[s]; tops[s] = topn[s]; }
            }
            // the block's 32 pixels of every row leave the output tile, the next block's enter the input tile
            ed_l4 o[3];
#pragma unroll
            for (int m = 0; m < 3; m++) o[m] = *(const ed_l4 *)(out_row + 16 * piece[m]);
            tile_block(nxt);
            ed_wave_sync();
#pragma unroll
            for (int m = 0; m < 3; m++) {
                const int xr = tb - kEdSkew * i + 4 * piece[m];
                const ed_gptr at = dst_row + (ptrdiff_t)xr * 4;
                if (ALL) *(__attribute__((address_space(1))) ed_u4 *)at = ed_u4{o[m].x, o[m].y, o[m].z, o[m].w};
                else if (m < 2 || ch < 2) {
#pragma unroll
                    for (int p = 0; p < 4; p++)
                        if (row_ok && xr + p >= 0 && xr + p < S.wl) *(__attribute__((address_space(1))) uint32_t *)(at + 4 * p) = o[m][p];
                }
            }
        };
        for (int blk = 0; blk < S.groups / kEdBlockGroups; blk++) {
            const int tb = kEdBlock * blk;
            if (full_band && tb - kEdSkew * (kEdRows - 1) >= 0 && tb + kEdBlock - 1 < S.wl) run_block(std::true_type{}, blk);
            else run_block(std::false_type{}, blk);
        }
        {   // the last group's words
            const int col0 = pend_t0 - kEdFlush;
            const int col = col0 + lane / 3;
            if (have_pending && lane < 3 * kEdGroup && col >= 0 && col < S.wl && !stall) __hip_atomic_store(mine + 3 * col0 + lane, pend_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

size_t ErrorDiffusionHandoffBytes(const ErrDiffParams &P, int n_frames)
{
    const EdSchedule S = ed_schedule(P.x0, P.x1, P.y1 - P.y0);
    return ((size_t)n_frames * S.bands * S.stride + 16) * sizeof(uint32_t);          // the hand-off rows + the ticket counter
}

// wavefronts of the pass the current device keeps resident (one per workgroup): queried once per device
static int ErrDiffResidentWaves()
{
    static std::mutex mu;
    static std::map<int, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256 * 16;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_error_diffusion, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 16;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cache[dev] = per_cu * cus;
}

hipError_t LaunchErrorDiffusion(const ErrDiffParams &P_in, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s)
{
    if (n_frames <= 0 || (!frames_dev && n_frames != 1)) return hipErrorInvalidValue;
    if (P_in.x1 <= P_in.x0 || P_in.y1 <= P_in.y0 || !P_in.handoff || !P_in.status) return hipErrorInvalidValue;
    ErrDiffParams P = P_in;
    if (P.spin_limit <= 0) P.spin_limit = 1 << 21;           // polls (~1 us each) a band grants the band above before it gives up and flags the launch
    const EdSchedule S = ed_schedule(P.x0, P.x1, P.y1 - P.y0);
    // the ticket counter starts at zero; the hand-off rows are cleared only when the caller does not vouch for them (gen = 0: no word of an
    // earlier launch on this buffer may carry the generation this launch tags its words with — then it runs as generation 1 on cleared rows)
    const size_t rows_bytes = ErrorDiffusionHandoffBytes(P, n_frames) - 16 * sizeof(uint32_t);
    hipError_t e = hipSuccess;
    if (P.gen <= 0 || P.gen > (int)kEdGenMask) { P.gen = 1; e = hipMemsetAsync(P.handoff, 0, rows_bytes + 16 * sizeof(uint32_t), s); }
    else e = hipMemsetAsync((uint8_t *)P.handoff + rows_bytes, 0, 16 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    // as many wavefronts as the chip keeps resident (or as there are bands); each takes bands by ticket until none is left.  MPCVR_ERRDIFF_WAVES
    // overrides the number (A/B: fewer resident wavefronts = fewer bands sharing a SIMD)
    static const int waves_env = [] { const char *v = std::getenv("MPCVR_ERRDIFF_WAVES"); return v ? std::atoi(v) : 0; }();
    const long total = (long)n_frames * S.bands;
    const long want = waves_env > 0 ? waves_env : ErrDiffResidentWaves();
    const dim3 grid((unsigned)std::max<long>(1, std::min<long>(total, want))), block(64);
    hipLaunchKernelGGL(k_error_diffusion, grid, block, 0, s, P, frames_dev, single, n_frames);
    return hipGetLastError();
}

}  // namespace mpcvr
