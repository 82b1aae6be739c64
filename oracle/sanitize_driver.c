/* oracle/sanitize_driver.c — TEST INFRASTRUCTURE: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer.
 * `make -C oracle sanitize` compiles mpcvr_oracle.c together with this driver (-fsanitize=address,undefined, no OpenMP) and runs
 * whole frames of awkward shapes through orc_process: every ColorFormat_t value, odd sizes, source rects, up / down / mixed
 * ratios, windows that clip the video rect, rotations, HDR tails; and the error-diffusion extension's serial model over regions of
 * awkward shapes.  A memory error or UB report makes it exit non-zero. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpcvr_oracle.h"

static uint64_t rng = 0x4D50435652ull;
static uint32_t next32(void) { rng += 0x9E3779B97F4A7C15ull; uint64_t z = rng; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return (uint32_t)((z ^ (z >> 31)) >> 16); }
static int rnd(int lo, int hi) { return lo + (int)(next32() % (uint32_t)(hi - lo + 1)); }

int main(void)
{
    int runs = 0, refused = 0;
    uint16_t dither[1024];                                   /* thresholds j/1024 as fp16 bits: any permutation will do here */
    for (int i = 0; i < 1024; i++) {
        const float d = (float)((i * 421) & 1023) / 1024.0f;
        uint32_t u; memcpy(&u, &d, 4);
        dither[i] = d == 0.0f ? 0 : (uint16_t)((((u >> 23) - 112) << 10) | ((u >> 13) & 0x3ff));
    }
    for (int cf = 1; cf <= 39; cf++)
        for (int rep = 0; rep < 3; rep++) {
            orc_params p;
            orc_params_default(&p);
            p.cformat = cf;
            p.width = 2 * rnd(4, 40); p.height = 2 * rnd(4, 30);
            if (cf == 10) p.width = (p.width / 6 + 1) * 6;                 /* v210 rows come in groups of 6 pixels */
            int pitch = 0;
            const size_t bytes = orc_frame_bytes(cf, p.width, p.height, &pitch);
            if (!bytes) { refused++; continue; }
            uint8_t *src = (uint8_t *)malloc(bytes);
            for (size_t i = 0; i < bytes; i++) src[i] = (uint8_t)next32();
            static const uint32_t ex[] = {0u, 0x7B4B0502u /* HDR10-ish */, 0x8B4B0502u /* HLG-ish */};
            p.exfmt = ex[rep];
            p.iUpscaling = rnd(0, 5); p.iDownscaling = rnd(0, 5); p.iChromaScaling = rnd(0, 2);
            p.bInterpolateAt50pct = rnd(0, 1);
            const int dw = 4 + rnd(0, 2 * p.width), dh = 4 + rnd(0, 2 * p.height);
            p.window_w = dw + rnd(0, 9); p.window_h = dh + rnd(0, 9);
            const int ox = rnd(-6, 8), oy = rnd(-6, 8);
            p.video_rect[0] = ox; p.video_rect[1] = oy; p.video_rect[2] = ox + dw; p.video_rect[3] = oy + dh;
            if (rep == 1 && cf < 29) { p.src_rect[0] = 2; p.src_rect[1] = 2; p.src_rect[2] = p.width - 2; p.src_rect[3] = p.height - 2; }
            if (rep == 2) { p.rotation = 90 * rnd(0, 3); p.flip = rnd(0, 1); }
            p.output_format = rnd(0, 1);
            uint8_t *dst = (uint8_t *)calloc((size_t)p.window_w * p.window_h, 4);
            const int rc = orc_process(&p, src, pitch, dither, dst, p.window_w * 4);
            if (rc != 0) refused++;
            runs++;
            free(dst); free(src);
        }
    /* the error-diffusion extension's serial model: regions of awkward shapes inside larger images (one column, one row, odd edges) */
    int ed_runs = 0;
    for (int rep = 0; rep < 40; rep++) {
        const int w = rnd(1, 90), h = rnd(1, 150);
        const int x0 = rnd(0, w - 1), y0 = rnd(0, h - 1), x1 = rnd(x0 + 1, w), y1 = rnd(y0 + 1, h);
        uint32_t *img = (uint32_t *)malloc((size_t)w * h * 4);
        for (int i = 0; i < w * h; i++) img[i] = next32() & 0x3fffffffu;
        uint8_t *out = (uint8_t *)calloc((size_t)w * h, 4);
        if (orc_error_diffusion(img, w * 4, out, w * 4, x0, y0, x1, y1) != 0) refused++;
        ed_runs++;
        free(out); free(img);
    }
    printf("sanitize_driver: %d frames processed, %d error-diffusion regions, %d refused combinations, no sanitizer report\n", runs, ed_runs, refused);
    return 0;
}
