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
// Schedule: pixel (x, y) needs (x-1, y), (x-1, y-1), (x, y-1), (x+1, y-1), so row y may run two columns behind row y-1.  A wavefront
// owns a BAND of 64 rows, lane i = row i, at step t lane i works on column t - 2 i; what a row passes to the row below,
//       D(x) = below-right(x-1) + below(x) + below-left(x+1),
// is complete one step before the lane below needs it and travels there by one DPP wave shift per step.  Every band is a workgroup
// of ONE wavefront, free-running: its bottom row publishes D column by column as tagged words (value << 1 | 1, zero = not there yet)
// in device memory — a relaxed agent-scope atomic store each, so value and tag arrive together and no fence is needed — and lane 0 of
// the band below reads them eight columns ahead and waits only when the band above has not got there yet (in steady state it runs
// ~130 columns behind).  A band depends on the band above alone = the workgroup with the next lower index: workgroups are started in
// index order per XCD, so the lowest unfinished one always finds its producer finished or running — no deadlock, whatever is resident.
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
constexpr int kEdRows = 64;               // rows per band = lanes of a wavefront
constexpr int kEdSkew = 2;                // columns a row runs behind the row above
constexpr int kEdGroup = 8;               // steps between two looks at the band above (and two loads of pixel pairs)
constexpr int kEdDummyWords = 128;       // 64 lanes x 8 bytes
constexpr int kEdBlockGroups = 4;        // groups per block of pixel loads: a lane fetches 32 pixels of its row (one cache line's worth) at a time

// floor((T + U / 2) / U) clamped to a byte, as a BIASED code q + 16 from the biased sum Tb = T + U/2 + 16 U: Tb is positive for every
// reachable T (|E| stays within a few U) and below 2^23; n = Tb >> 4 is below 2^19, where floor(n / 1023) = mulhi(n, ceil(2^32 / 1023))
// exactly (the excess 1019 n / (1023 * 2^32) stays below 1 / 1023 up to n = 4.2 M) — tests/test_errdiff.py checks the whole range against
// the division.  The masks change no value (n < 2^19, the magic < 2^23) and let the compiler take the full-rate 24-bit multiply-high; the
// bias saves the kernel an addition per channel (it subtracts 16 from the three codes of a pixel at once, in the packed word)
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

// The B8G8R8A8 texel of three BIASED codes (ed_step's answers): R = byte 2, alpha 0xFF.  A sum, not an or: a biased code reaches 271 and
// carries into the field above it until the three biases are taken out in one subtraction (shared with the host emulation: the first
// cut or-ed the fields in the kernel only, and only the GPU tests could see it)
MPCVR_ED_HD uint32_t ed_pack_bgra(int qr, int qg, int qb)
{
    return (((uint32_t)qr << 16) + ((uint32_t)qg << 8) + (uint32_t)qb) + (0xff000000u - 0x00101010u);
}

// ---- the schedule ----
// Region columns are counted from A0 = x0 & ~1 (xr = column - A0), so that xr and the step index have the same parity in every lane:
// a pair of steps (even, odd) covers one 8-byte aligned pixel pair.  wl = x1 - A0 columns, the first x0 - A0 (0 or 1) of them outside.
struct EdSchedule {
    int wl;              // columns counted from A0
    int lead;            // x0 - A0
    int bands;           // ceil(rows / 64)
    int groups;          // groups of kEdGroup steps per band: lane 63 must reach the flush step at xr = wl
    int stride;          // words of one band's hand-off row: 3 per column (R, G, B), every step of lane 0 has its entry, a spare group
                         // behind them (where the lanes of a group's store that have nothing to publish write), then kEdDummyWords of
                         // dummy slots for this band's pixel stores off the region
};
MPCVR_ED_HD EdSchedule ed_schedule(int x0, int x1, int rows)
{
    EdSchedule s;
    const int a0 = x0 & ~1;
    s.wl = x1 - a0; s.lead = x0 - a0;
    s.bands = (rows + kEdRows - 1) / kEdRows;
    s.groups = (s.wl + 1 + kEdSkew * (kEdRows - 1) + kEdGroup - 1) / kEdGroup;
    s.groups = (s.groups + kEdBlockGroups - 1) / kEdBlockGroups * kEdBlockGroups;
    s.stride = 3 * kEdGroup * (s.groups + 1) + kEdDummyWords;
    return s;
}
// hand-off words: D of column c, channel ch of a band's bottom row sits at word 3 c + ch; tagged so that zero means "not written yet"
MPCVR_ED_HD uint32_t ed_tag(int32_t d) { return ((uint32_t)d << 1) | 1u; }
MPCVR_ED_HD int32_t ed_untag(uint32_t w) { return (int32_t)w >> 1; }

}  // namespace mpcvr
