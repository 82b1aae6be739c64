/* c_multi_gpu.c — the multi-GPU use of the path from plain C, one process: a context per HIP device (0 .. N-1, however many
 * the node has — it degrades to 1), rank 0's parameter blob handed to the others (mpcvr_get_param_blob / mpcvr_set_param_blob:
 * what videorenderer_amd/dist.py broadcasts over RCCL when the contexts live in separate processes), frames dealt by index
 * (frame i -> device i % N), no data exchanged between devices.  Every frame is rendered through mpcvr_process into device
 * memory owned by the library (mpcvr_render / mpcvr_get_current_image) and checksummed; identical frames must give identical
 * checksums on every device.
 *
 *   gcc -O2 -std=c99 -Iinclude examples/c_multi_gpu.c -o c_multi_gpu -Lvideorenderer_amd -lmpcvr -Wl,-rpath,$PWD/videorenderer_amd
 *   ./c_multi_gpu [max_devices] [frames]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpcvr.h"

#define MAXDEV 16

static uint32_t fnv1a(const uint8_t *p, size_t n)
{
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 16777619u; }
    return h;
}

int main(int argc, char **argv)
{
    const int want = argc > 1 ? atoi(argv[1]) : MAXDEV, frames = argc > 2 ? atoi(argv[2]) : 8;
    const int w = 256, h = 144;
    mpcvr_ctx *ctx[MAXDEV] = {0};
    int n = 0;
    for (int d = 0; d < want && d < MAXDEV; d++) {                 /* a context per device until the ordinal runs out */
        mpcvr_settings s;
        mpcvr_settings_default(&s);
        s.iUpscaling = MPCVR_UPSCALE_Lanczos3;
        if (d > 0) s.iSDRDisplayNits = 200;                         /* deliberately different: the blob below overrides it */
        mpcvr_ctx *c = NULL;
        const int32_t hr = mpcvr_create(&s, d, &c);
        if (hr < 0) break;
        ctx[n++] = c;
    }
    if (n == 0) { fprintf(stderr, "no HIP device\n"); return 1; }
    /* P010, BT.2020 / PQ (HDR10), limited range, MPEG-2 chroma siting: the headline configuration at a small size */
    const uint32_t extfmt = (5u << 8) | (2u << 12) | (4u << 15) | (9u << 22) | (15u << 27);
    const mpcvr_rect out = {0, 0, 2 * w, 2 * h};
    for (int d = 0; d < n; d++) {
        if (mpcvr_set_input(ctx[d], MPCVR_CF_P010, w, h, 0, NULL, extfmt) < 0 || mpcvr_set_window_rect(ctx[d], &out) < 0 ||
            mpcvr_set_video_rect(ctx[d], &out) < 0) { fprintf(stderr, "device %d: %s\n", d, mpcvr_last_error(ctx[d])); return 1; }
    }
    size_t blob_size = 0;
    mpcvr_get_param_blob(ctx[0], NULL, &blob_size);
    void *blob = malloc(blob_size);
    if (mpcvr_get_param_blob(ctx[0], blob, &blob_size) < 0) { fprintf(stderr, "blob: %s\n", mpcvr_last_error(ctx[0])); return 1; }
    for (int d = 1; d < n; d++)
        if (mpcvr_set_param_blob(ctx[d], blob, blob_size) < 0) { fprintf(stderr, "device %d: %s\n", d, mpcvr_last_error(ctx[d])); return 1; }

    size_t bytes = 0; int32_t pitch = 0;
    mpcvr_get_frame_bytes(ctx[0], &bytes, &pitch);
    uint16_t *frame = (uint16_t *)malloc(bytes);
    size_t isz = 0;
    uint32_t first = 0;
    int bad = 0;
    for (int i = 0; i < frames; i++) {
        const int d = i % n;                                        /* frames shard by index */
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) frame[(size_t)y * (pitch / 2) + x] = (uint16_t)((64 + ((x * 7 + y * 3) % 876)) << 6);
        for (int y = 0; y < h / 2; y++)
            for (int x = 0; x < w; x++) frame[(size_t)(h + y) * (pitch / 2) + x] = (uint16_t)((64 + ((x * 5 + y * 11) % 896)) << 6);
        if (mpcvr_copy_sample(ctx[d], frame, pitch, MPCVR_MEM_HOST) < 0 || mpcvr_render(ctx[d], 1) < 0 || mpcvr_synchronize(ctx[d]) < 0) {
            fprintf(stderr, "frame %d on device %d: %s\n", i, d, mpcvr_last_error(ctx[d])); return 1;
        }
        /* read the back buffer through the snapshot call at source size as a cheap device->host path */
        mpcvr_get_current_image(ctx[d], NULL, &isz);
        uint8_t *img = (uint8_t *)malloc(isz);
        if (mpcvr_get_current_image(ctx[d], img, &isz) < 0) { fprintf(stderr, "snapshot: %s\n", mpcvr_last_error(ctx[d])); return 1; }
        const uint32_t sum = fnv1a(img, isz);
        free(img);
        if (i == 0) first = sum;
        if (sum != first) bad++;
        printf("frame %d device %d fnv1a=%08x\n", i, d, sum);
    }
    char info[128] = "";
    mpcvr_get_path_info(ctx[0], info, sizeof info);
    printf("devices=%d frames=%d path=%s identical=%s\n", n, frames, info, bad ? "NO" : "yes");
    for (int d = 0; d < n; d++) mpcvr_destroy(ctx[d]);
    free(frame); free(blob);
    return bad ? 2 : 0;
}
