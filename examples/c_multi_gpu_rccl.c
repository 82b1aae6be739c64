/* c_multi_gpu_rccl.c — SURVEY.md 8e from a host without Python: ONE process drives every visible HIP device (it degrades to one),
 * an RCCL communicator per device (ncclCommInitAll), ONE collective — rank 0's parameter blob (colour matrix, tone-map table,
 * resize phase weights, dither table) broadcast to every device over xGMI by mpcvr_broadcast_param_blob_begin / _end — and then
 * frames dealt by index (frame i -> device i % N), nothing exchanged on the data path.  Rank 0 is configured differently from the
 * others on purpose (200-nit SDR target): identical checksums on every device prove the broadcast blob is what the frames ran on.
 *
 *   gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_multi_gpu_rccl.c -o c_multi_gpu_rccl \
 *       -Lvideorenderer_amd -lmpcvr -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,$PWD/videorenderer_amd -Wl,-rpath,/opt/rocm/lib
 *   ./c_multi_gpu_rccl [max_devices] [frames]
 *
 * (The process-per-GPU form of the same exchange is videorenderer_amd/dist.py over torch.distributed, or
 *  mpcvr_broadcast_param_blob(ctx, comm, 0, rank) behind ncclCommInitRank in a C++ host.)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>

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
    int devs[MAXDEV];
    int n = 0;
    for (int d = 0; d < want && d < MAXDEV; d++) {                 /* a context per device until the ordinal runs out */
        mpcvr_settings s;
        mpcvr_settings_default(&s);
        s.iUpscaling = MPCVR_UPSCALE_Lanczos3;
        s.iSDRDisplayNits = d == 0 ? 200 : 125;                     /* deliberately different: the broadcast overrides it */
        mpcvr_ctx *c = NULL;
        if (mpcvr_create(&s, d, &c) < 0) break;
        devs[n] = d;
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

    /* the one collective: ncclBroadcast of rank 0's blob, all ranks of this process inside one group */
    ncclComm_t comm[MAXDEV];
    ncclResult_t rc = ncclCommInitAll(comm, n, devs);
    if (rc != ncclSuccess) { fprintf(stderr, "ncclCommInitAll: %s\n", ncclGetErrorString(rc)); return 1; }
    int version = 0;
    ncclGetVersion(&version);
    if ((rc = ncclGroupStart()) != ncclSuccess) { fprintf(stderr, "ncclGroupStart: %s\n", ncclGetErrorString(rc)); return 1; }
    for (int d = 0; d < n; d++)
        if (mpcvr_broadcast_param_blob_begin(ctx[d], (void *)comm[d], 0, d) < 0) { fprintf(stderr, "broadcast begin, device %d: %s\n", d, mpcvr_last_error(ctx[d])); return 1; }
    if ((rc = ncclGroupEnd()) != ncclSuccess) { fprintf(stderr, "ncclGroupEnd: %s\n", ncclGetErrorString(rc)); return 1; }
    for (int d = 0; d < n; d++)
        if (mpcvr_broadcast_param_blob_end(ctx[d]) < 0) { fprintf(stderr, "broadcast end, device %d: %s\n", d, mpcvr_last_error(ctx[d])); return 1; }
    printf("rccl=%d ranks=%d broadcast=ok\n", version, n);

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
        mpcvr_get_current_image(ctx[d], NULL, &isz);
        uint8_t *img = (uint8_t *)malloc(isz);
        if (mpcvr_get_current_image(ctx[d], img, &isz) < 0) { fprintf(stderr, "snapshot: %s\n", mpcvr_last_error(ctx[d])); return 1; }
        const uint32_t sum = fnv1a(img, isz);
        free(img);
        if (i == 0) first = sum;
        if (sum != first) bad++;
        printf("frame %d device %d fnv1a=%08x\n", i, d, sum);
    }
    /* the blob every rank ended up with is rank 0's, bit for bit */
    size_t bs = 0;
    mpcvr_get_param_blob(ctx[0], NULL, &bs);
    uint8_t *b0 = (uint8_t *)malloc(bs), *bd = (uint8_t *)malloc(bs);
    mpcvr_get_param_blob(ctx[0], b0, &bs);
    for (int d = 1; d < n; d++) {
        size_t s2 = bs;
        mpcvr_get_param_blob(ctx[d], bd, &s2);
        if (s2 != bs || memcmp(b0, bd, bs) != 0) { fprintf(stderr, "device %d holds a different blob\n", d); bad++; }
    }
    char info[128] = "";
    mpcvr_get_path_info(ctx[0], info, sizeof info);
    printf("devices=%d frames=%d path=%s identical=%s\n", n, frames, info, bad ? "NO" : "yes");
    for (int d = 0; d < n; d++) { mpcvr_destroy(ctx[d]); ncclCommDestroy(comm[d]); }
    free(frame); free(b0); free(bd);
    return bad ? 2 : 0;
}
