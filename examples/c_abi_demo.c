/* c_abi_demo.c — libmpcvr.so driven from plain C through include/mpcvr.h only (no HIP, no Python): what an adapter inside the
 * renderer does per frame — InitMediaType, SetVideoRect / SetWindowRect, CopySample from host memory, Process through
 * GetCurentImage — on a synthetic NV12 frame.  Prints the path description and an FNV-1a checksum of the BGRX image.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o c_abi_demo -Lvideorenderer_amd -lmpcvr -Wl,-rpath,$PWD/videorenderer_amd
 *   ./c_abi_demo [width height]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mpcvr.h"

#define CHECK(call) do { int32_t hr_ = (call); if (hr_ < 0) { fprintf(stderr, "%s -> 0x%08X: %s\n", #call, (unsigned)hr_, ctx ? mpcvr_last_error(ctx) : ""); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int w = argc > 2 ? atoi(argv[1]) : 128, h = argc > 2 ? atoi(argv[2]) : 72;
    mpcvr_ctx *ctx = NULL;
    mpcvr_settings s;
    mpcvr_settings_default(&s);
    CHECK(mpcvr_create(&s, 0, &ctx));

    /* NV12, BT.709 limited range (DXVA2_ExtendedFormat: chroma MPEG-2, range 16-235, matrix BT.709) */
    const uint32_t extfmt = (5u << 8) | (2u << 12) | (1u << 15);
    CHECK(mpcvr_set_input(ctx, MPCVR_CF_NV12, w, h, 0, NULL, extfmt));
    const mpcvr_rect r = {0, 0, w, h};
    CHECK(mpcvr_set_window_rect(ctx, &r));
    CHECK(mpcvr_set_video_rect(ctx, &r));

    size_t bytes = 0; int32_t pitch = 0;
    CHECK(mpcvr_get_frame_bytes(ctx, &bytes, &pitch));
    uint8_t *frame = (uint8_t *)malloc(bytes);
    for (int y = 0; y < h; y++)                                   /* luma ramp with a checker */
        for (int x = 0; x < w; x++) frame[(size_t)y * pitch + x] = (uint8_t)(16 + (x * 219) / (w - 1) - (((x >> 3) ^ (y >> 3)) & 1) * 8);
    for (int y = 0; y < h / 2; y++)                               /* chroma gradients */
        for (int x = 0; x < w / 2; x++) {
            frame[(size_t)(h + y) * pitch + 2 * x] = (uint8_t)(64 + (x * 128) / (w / 2));
            frame[(size_t)(h + y) * pitch + 2 * x + 1] = (uint8_t)(192 - (y * 128) / (h / 2));
        }
    CHECK(mpcvr_copy_sample(ctx, frame, pitch, MPCVR_MEM_HOST));

    size_t size = 0;
    CHECK(mpcvr_get_current_image(ctx, NULL, &size));
    uint8_t *image = (uint8_t *)malloc(size);
    CHECK(mpcvr_get_current_image(ctx, image, &size));

    char info[256];
    CHECK(mpcvr_get_path_info(ctx, info, sizeof(info)));
    uint32_t fnv = 2166136261u;
    for (size_t i = 0; i < size; i++) { fnv ^= image[i]; fnv *= 16777619u; }
    printf("%s %dx%d bytes=%zu fnv1a=%08x first=%02x%02x%02x%02x\n", info, w, h, size, fnv, image[0], image[1], image[2], image[3]);
    free(image); free(frame);
    mpcvr_destroy(ctx);
    return 0;
}
