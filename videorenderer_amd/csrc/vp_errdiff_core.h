// vp_errdiff_core.h — the arithmetic and the wavefront schedule of the error-diffusion final pass (EXTENSION: bUseDither = 2,
// MPCVR_DITHER_ErrorDiffusion_EXT).  The reference has no such pass (its final pass is the ordered dither of ps_final_pass.hlsl;
// `grep -ri diffusion` over the reference is empty): BASELINE.json's config 4 names it, nothing pins it.  So the definition below IS
// the specification, in integers so that a GPU schedule and a serial loop cannot differ by a rounding:
//
//   the frame is rendered as for a 10-bit swap chain (R10G10B10A2, no final pass); inside (video rect ∩ window), rows top to bottom,
//   every row left to right, per channel, with U = 16 * 1023 error units per 8-bit code:
//       T  = 4080 k + E(x, y)                        k = the UNORM10 code (k / 1023 * 255 codes = 4080 k units), E = errors received
//       q  = clamp(floor((T + U / 2) / U), 0, 255)   the 8-bit code stored
//       e  = T - q U
//       right (7 e) >> 4, below-left (3 e) >> 4, below (5 e) >> 4 (arithmetic shifts = floor), below-right the remainder
//   (Floyd-Steinberg weights; what falls outside the region is dropped; alpha = 0xFF).
//
// Schedule (round 5: ONE CHANNEL PER LANE).  Pixel (x, y) needs (x-1, y), (x-1, y-1), (x, y-1), (x+1, y-1), so row y may run two columns
// behind row y-1; the three channels are independent chains.  A wavefront owns a BAND of 21 rows: lane = 21 * channel + row (lane 63 idles),
// at step t the lanes of row i work on column t - 2 i; what a row passes to the row below,
//       D(x) = below-right(x-1) + below(x) + below-left(x+1),
// is complete one step before the lane below needs it and travels there by one DPP wave shift per step (the three lanes that are a row 0
// take the band above's value instead).  Until round 4 a lane carried all three channels of its row (64 rows per band): three chains
// serialised in one lane, 76 instructions per step, 2.9 ms per 8K frame and a third of the issue slots busy in a batch.  Now a step is a
// third as long, a frame has three times as many bands running (206 instead of 68 for 4320 rows) and every SIMD holds bands of six phases.
// A band's bottom row publishes D column by column as tagged words (value << 12 | the launch's generation) in device memory — a relaxed
// agent-scope atomic store per group of 8 steps, value and tag in one word, so nothing needs a fence — and the band below reads them a
// group ahead and waits only when the band above has not got there yet (in steady state it runs ~56 columns behind).
// Bands are handed out by TICKET (an atomic counter, band-major over the frames of a launch): a band's producer always holds a lower
// ticket, i.e. it was taken by a wavefront that is running or done — forward progress does not hang on the order in which the hardware
// starts workgroups (round 4's version did: workgroup index = band), and a launch never needs more wavefronts resident than the chip holds.
// This header holds everything a host emulation of that schedule shares with the kernel (tests/tools/errdiff_emulate.cpp: bands taking
// turns in random order against the serial model, no GPU needed).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MPCVR_ED_HD __host__ __device__ __forceinline__
#else
#define MPCVR_ED_HD inline
#endif

// e * 7 as a full-rate 24-bit multiply on the device (the compiler picks the quarter-rate v_mul_lo_u32 for it otherwise)
#if defined(__HIP_DEVICE_COMPILE__)
#define MPCVR_ED_MUL24(a, b) __mul24((a), (b))
#else
#define MPCVR_ED_MUL24(a, b) ((a) * (b))
#endif

namespace mpcvr {

constexpr int kEdUnit = 16 * 1023;        // error units per 8-bit code
constexpr int kEdCode = 16 * 255;         // one UNORM10 code in those units
constexpr int kEdRows = 21;               // rows per band: 3 channels x 21 rows = 63 lanes of a wavefront
constexpr int kEdSkew = 2;                // columns a row runs behind the row above
constexpr int kEdGroup = 8;               // steps between two looks at the band above
constexpr int kEdBlockGroups = 4;         // groups per block: the 32 pixels a row works on in a block are one 128-byte piece of it
constexpr int kEdBlock = kEdGroup * kEdBlockGroups;
constexpr int kEdFlush = kEdSkew * (kEdRows - 1) + 1;     // the bottom row hands D(c) down at step c + kEdFlush

// floor((T + U / 2) / U) clamped to a byte, as a BIASED code q + 16 from the biased sum Tb = T + U/2 + 16 U: Tb is positive for every
// reachable T (|E| stays within a few U) and below 2^23; n = Tb >> 4 is below 2^19, where floor(n / 1023) = mulhi(n, ceil(2^32 / 1023))
// exactly (the excess 1019 n / (1023 * 2^32) stays below 1 / 1023 up to n = 4.2 M) — tests/test_errdiff.py checks the whole range against
// the division.  The masks change no value (n < 2^19, the magic < 2^23) and let the compiler take the full-rate 24-bit multiply-high.
constexpr uint32_t kEdMagic = 4198405u;   // ceil(2^32 / 1023)
constexpr int32_t kEdBias = kEdUnit / 2 + 16 * kEdUnit;
MPCVR_ED_HD int ed_quant_biased(int32_t Tb)
{
    const uint32_t n = ((uint32_t)Tb >> 4) & 0xffffffu;
    const uint32_t qb = (uint32_t)(((uint64_t)n * (uint64_t)(kEdMagic & 0xffffffu)) >> 32);
    const uint32_t lo = qb > 16u ? qb : 16u;             // (max, then min: one v_med3_u32)
    return (int)(lo < 271u ? lo : 271u);
}
MPCVR_ED_HD int ed_quant(int32_t T) { return ed_quant_biased(T + kEdBias) - 16; }

// one channel of one row: what the lane carries from pixel to pixel
struct EdChannel {
    int32_t er;          // right share of the previous pixel
    int32_t b1, br1;     // below / below-right shares of the previous pixel
    int32_t br2;         // below-right share of the pixel before that
};

// One step of one channel.  live: the lane stands on a pixel of the region (k = its code, din = D of that column from the row above);
// otherwise the step only flushes the shares still in flight (e = 0).  dout = D(x - 1) for the row below; returns the BIASED 8-bit code
// q + 16 (meaningful on live pixels only).
MPCVR_ED_HD int ed_step(EdChannel &s, bool live, int k, int32_t din, int32_t &dout)
{
    // (branch-free: off the region Tb is whatever the shares in flight add up to, the code is not used and e is forced to zero)
    const int32_t Tb = (k * kEdCode + s.er) + din + kEdBias;
    const int qb = ed_quant_biased(Tb);
    // T - q U = Tb - (q + 16) U - U/2; q + 16 < 2^9: a 24-bit multiply
    const int32_t e = live ? Tb - (int32_t)(((uint32_t)qb & 0x1ffu) * (uint32_t)kEdUnit) - kEdUnit / 2 : 0;
    const int32_t r = MPCVR_ED_MUL24(e, 7) >> 4, bl = (3 * e) >> 4, b = (5 * e) >> 4, br = e - r - bl - b;      // (|e| stays far below 2^23)
    dout = s.br2 + s.b1 + bl;
    s.er = r; s.br2 = s.br1; s.br1 = br; s.b1 = b;
    return qb;
}

// The B8G8R8A8 texel of three BIASED codes (ed_step's answers): R = byte 2, alpha 0xFF.  (The kernel writes the three bytes from three
// lanes; the host emulation packs them here.)
MPCVR_ED_HD uint32_t ed_pack_bgra(int qr, int qg, int qb)
{
    return (((uint32_t)qr << 16) + ((uint32_t)qg << 8) + (uint32_t)qb) + (0xff000000u - 0x00101010u);
}

// ---- the schedule ----
// Region columns are counted from x0 (xr = column - x0); wl = x1 - x0 of them.
struct EdSchedule {
    int wl;              // columns of the region
    int bands;           // ceil(rows / 21)
    int groups;          // groups of kEdGroup steps per band: the bottom row must reach the flush step at xr = wl; a multiple of kEdBlockGroups
    int stride;          // words of one band's hand-off row: 3 per column (R, G, B), every step of a row 0 has its entry, spare groups
                         // behind them (the band below fetches five groups ahead; the lanes of a group's store that have nothing to publish write there)
};
MPCVR_ED_HD EdSchedule ed_schedule(int x0, int x1, int rows)
{
    EdSchedule s;
    s.wl = x1 - x0;
    s.bands = (rows + kEdRows - 1) / kEdRows;
    s.groups = (s.wl + kEdFlush + kEdGroup - 1) / kEdGroup;          // steps 0 .. wl + kEdFlush - 1
    s.groups = (s.groups + kEdBlockGroups - 1) / kEdBlockGroups * kEdBlockGroups;
    s.stride = 3 * kEdGroup * (s.groups + 6);
    return s;
}
// hand-off words: D of column c, channel ch of a band's bottom row sits at word 3 c + ch, as D << 12 | gen: gen = the launch's generation
// (1 .. 4095, counted by the owner of the rows), so that a word of an EARLIER launch reads as "not written yet" and the rows need no clearing
// between launches of one geometry (they are cleared when the generation wraps or the layout changes; round 4 cleared them in front of every
// launch — 600 MB for a 32-frame batch at 21 rows per band).  |D| stays below 2^18 (three shares of an error of a few U), 20 bits hold it.
constexpr int kEdGenBits = 12;
constexpr uint32_t kEdGenMask = (1u << kEdGenBits) - 1u;
MPCVR_ED_HD uint32_t ed_tag(int32_t d, uint32_t gen) { return ((uint32_t)d << kEdGenBits) | gen; }
MPCVR_ED_HD int32_t ed_untag(uint32_t w) { return (int32_t)w >> kEdGenBits; }
MPCVR_ED_HD bool ed_tagged(uint32_t w, uint32_t gen) { return (w & kEdGenMask) == gen; }
// ticket -> (frame, band): band-major (a band of every frame, then the next band) or frame-major; either way a band's producer has a lower ticket
MPCVR_ED_HD void ed_ticket(int ticket, int n_frames, int bands, int order, int &z, int &band)
{
    if (order) { band = ticket / n_frames; z = ticket - band * n_frames; }
    else { z = ticket / bands; band = ticket - z * bands; }
}

}  // namespace mpcvr
