// tests/tools/errdiff_emulate.cpp — TEST TOOL (never linked into libmpcvr.so): the wavefront schedule of k_error_diffusion
// (videorenderer_amd/csrc/vp_errdiff.hip) executed on the host, lane by lane and slot by slot, from the SAME header the kernel is
// built from (vp_errdiff_core.h: ed_step, ed_quant, ed_schedule, ed_slot_work).  tests/test_errdiff.py compares it with the serial
// model of the oracle: the dependency analysis of the schedule (skew, lag, the in-place row buffer) is checked without a GPU.
// Waves of a slot run here one after the other in an order the caller picks (`order` = 0: ascending, 1: descending, 2: interleaved),
// which is how a race between waves inside a slot would show.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../videorenderer_amd/csrc/vp_errdiff_core.h"

using namespace mpcvr;

extern "C" int ed_quant_host(int32_t T) { return ed_quant(T); }
extern "C" void ed_quant_range(int32_t lo, int32_t hi, int32_t *out) { for (int32_t t = lo; t < hi; t++) out[t - lo] = ed_quant(t); }

extern "C" int ed_emulate(const uint32_t *src10, int src_pitch, uint8_t *dst, int dst_pitch, int x0, int y0, int x1, int y1, int order)
{
    const EdSchedule S = ed_schedule(x0, x1, y1 - y0);
    const int brw = S.slots_per_band * kEdChunk + 8;
    std::vector<int32_t> rowbuf((size_t)3 * brw, 0);
    const int a0 = x0 & ~1, rows = y1 - y0;
    struct Wave { EdChannel st[64][3]; int32_t dprev[64][3]; };
    std::vector<Wave> waves(kEdWaves);
    for (int slot = 0; slot < S.total_slots; slot++) {
        // reads of the row buffer by lane 0 happen at group starts, writes by lane 63 per step: emulate in program order per wave
        for (int wi = 0; wi < kEdWaves; wi++) {
            const int w = order == 0 ? wi : order == 1 ? kEdWaves - 1 - wi : ((wi & 1) ? kEdWaves - 1 - wi / 2 : wi / 2);
            int band, chunk;
            if (!ed_slot_work(S, w, slot, &band, &chunk)) continue;
            Wave &W = waves[w];
            if (chunk == 0) std::memset(&W, 0, sizeof(W));
            const int tbase = chunk * kEdChunk;
            for (int g = 0; g < kEdChunk / 8; g++) {
                const int t0 = tbase + 8 * g;
                int32_t top[3][8];
                for (int c = 0; c < 3; c++)
                    for (int s = 0; s < 8; s++) top[c][s] = band > 0 ? rowbuf[(size_t)c * brw + t0 + s] : 0;
                for (int s = 0; s < 8; s++) {
                    int32_t shifted[64][3];
                    for (int lane = 0; lane < 64; lane++)
                        for (int c = 0; c < 3; c++) shifted[lane][c] = lane ? W.dprev[lane - 1][c] : 0;      // the DPP wave shift
                    for (int lane = 0; lane < 64; lane++) {
                        const int r = band * kEdRows + lane;
                        const bool row_ok = r < rows;
                        const int xr = t0 + s - kEdSkew * lane;
                        const bool live = row_ok && xr >= S.lead && xr < S.wl;
                        uint32_t code = 0;
                        if (row_ok && xr >= 0 && xr < S.wl + 1 && a0 + xr < x1)        // (the kernel reads pairs; only live codes matter)
                            code = *(const uint32_t *)((const uint8_t *)src10 + (size_t)(y0 + r) * src_pitch + (size_t)(a0 + xr) * 4);
                        int q[3];
                        for (int c = 0; c < 3; c++) {
                            const int32_t din = lane == 0 ? top[c][s] : shifted[lane][c];
                            q[c] = ed_step(W.st[lane][c], live, (int)((code >> (10 * c)) & 0x3ffu), din, W.dprev[lane][c]);
                        }
                        if (live) {
                            uint8_t *px = dst + (size_t)(y0 + r) * dst_pitch + (size_t)(a0 + xr) * 4;
                            px[0] = (uint8_t)q[2]; px[1] = (uint8_t)q[1]; px[2] = (uint8_t)q[0]; px[3] = 0xff;
                        }
                        if (lane == 63 && xr >= 1)
                            for (int c = 0; c < 3; c++) rowbuf[(size_t)c * brw + (xr - 1)] = W.dprev[63][c];
                    }
                }
            }
        }
    }
    return S.total_slots;
}
