// tests/tools/errdiff_emulate.cpp — TEST TOOL (never linked into libmpcvr.so): the schedule of k_error_diffusion
// (videorenderer_amd/csrc/vp_errdiff.hip) executed on the host from the SAME header the kernel is built from (vp_errdiff_core.h:
// ed_step, ed_quant, ed_schedule, the tagged hand-off words).  Every band of 21 rows (one channel per lane) is a free-running wavefront that advances one
// group of 8 steps at a time and may do so only when the hand-off words it needs from the band above are there; here the bands take
// turns in an order drawn from `seed` (or strictly top-down / bottom-up first), which is how a missing dependency, a word read before
// it is written or a band that can never proceed would show.  tests/test_errdiff.py compares the result with the serial model of the
// oracle: the dependency analysis of the schedule is checked without a GPU.  Returns the number of (band, group) turns that had to
// wait, or -1 when no band can proceed (deadlock).
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../videorenderer_amd/csrc/vp_errdiff_core.h"

using namespace mpcvr;

extern "C" int ed_quant_host(int32_t T) { return ed_quant(T); }
extern "C" void ed_quant_range(int32_t lo, int32_t hi, int32_t *out) { for (int32_t t = lo; t < hi; t++) out[t - lo] = ed_quant(t); }

namespace {
struct Band { EdChannel st[kEdRows][3]; int32_t dprev[kEdRows][3]; int g; };
constexpr uint32_t kGen = 7;      // the launch's generation: words of other generations (here: the cleared rows) read as not written
uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }
}

extern "C" int ed_emulate(const uint32_t *src10, int src_pitch, uint8_t *dst, int dst_pitch, int x0, int y0, int x1, int y1, uint32_t seed)
{
    const EdSchedule S = ed_schedule(x0, x1, y1 - y0);
    std::vector<uint32_t> handoff((size_t)S.bands * S.stride, 0u);           // zero = not written yet (the launcher's memset)
    const int rows = y1 - y0;
    std::vector<Band> bands(S.bands);
    for (Band &b : bands) std::memset(&b, 0, sizeof(b));
    int waits = 0, idle = 0;
    size_t left = (size_t)S.bands * S.groups;
    uint32_t rs = seed;
    // the kernel's wait: every word of a column inside the region must carry its tag
    auto ready = [&](int b) {
        if (b == 0) return true;
        const uint32_t *above = &handoff[(size_t)(b - 1) * S.stride];
        const int t0 = kEdGroup * bands[b].g;
        for (int l = 0; l < 3 * kEdGroup; l++)
            if (t0 + l / 3 < S.wl && !ed_tagged(above[3 * t0 + l], kGen)) return false;
        return true;
    };
    while (left) {
        // whose turn: seed 0 = the lowest unfinished band (the serial order), 1 = the HIGHEST unfinished band that can move (every band runs as
        // close behind the band above as the hand-off allows), else a random unfinished band, which waits when its words are not there
        int b = -1;
        if (seed == 0) { for (int i = 0; i < S.bands && b < 0; i++) if (bands[i].g < S.groups) b = i; }
        else if (seed == 1) {
            for (int i = S.bands - 1; i >= 0 && b < 0; i--)
                if (bands[i].g < S.groups) { if (ready(i)) b = i; else waits++; }
            if (b < 0) return -1;
        } else { do b = (int)(lcg(rs) % (uint32_t)S.bands); while (bands[b].g >= S.groups); }
        if (!ready(b)) {
            waits++;
            if (++idle > 64 * S.bands + 64) return -1;      // nobody moved for a long time: a band that can never proceed
            continue;
        }
        Band &W = bands[b];
        const int t0 = kEdGroup * W.g;
        const uint32_t *above = b ? &handoff[(size_t)(b - 1) * S.stride] : nullptr;
        uint32_t *mine = &handoff[(size_t)b * S.stride];
        idle = 0;
        for (int s = 0; s < kEdGroup; s++) {
            int32_t shifted[kEdRows][3];
            for (int row = 0; row < kEdRows; row++)
                for (int c = 0; c < 3; c++) shifted[row][c] = row ? W.dprev[row - 1][c] : 0;      // the DPP wave shift (lane = 21 c + row)
            for (int row = 0; row < kEdRows; row++) {
                const int r = b * kEdRows + row;
                const bool row_ok = r < rows;
                const int xr = t0 + s - kEdSkew * row;
                const bool live = row_ok && xr >= 0 && xr < S.wl;
                uint32_t code = 0;
                if (live) code = *(const uint32_t *)((const uint8_t *)src10 + (size_t)(y0 + r) * src_pitch + (size_t)(x0 + xr) * 4);
                int q[3];
                for (int c = 0; c < 3; c++) {
                    const int32_t din = row == 0 ? (above ? ed_untag(above[3 * (t0 + s) + c]) : 0) : shifted[row][c];
                    q[c] = ed_step(W.st[row][c], live, (int)((code >> (10 * c)) & 0x3ffu), din, W.dprev[row][c]);
                }
                if (live) {         // (ed_step answers the biased code q + 16; the kernel writes the three bytes q - 16 from three lanes)
                    const uint32_t texel = ed_pack_bgra(q[0], q[1], q[2]);
                    std::memcpy(dst + (size_t)(y0 + r) * dst_pitch + (size_t)(x0 + xr) * 4, &texel, 4);
                }
                if (row == kEdRows - 1 && xr >= 1 && xr <= S.wl)
                    for (int c = 0; c < 3; c++) mine[3 * (xr - 1) + c] = ed_tag(W.dprev[row][c], kGen);
            }
        }
        W.g++; left--;
    }
    return waits;
}
