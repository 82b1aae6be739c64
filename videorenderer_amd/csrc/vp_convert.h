// vp_convert.h — the generated convert shader (Shaders.cpp:593-930) as a device function, shared by
// the pass-per-kernel path (vp_kernels.hip, fp-contract off) and the fused path (vp_fused.hip).
#pragma once
#include "vp_device.h"

namespace mpcvr {

// ------------------------------------------------------------------------------------------------
// chroma fetch — ShaderGetPixels, DX11 branch (Shaders.cpp:82-329)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sample_chroma_linear(const ConvertParams &P, int c, float u, float v)
{
    const float fu = u - 0.5f, fv = v - 0.5f;
    const float iu = floorf(fu), iv = floorf(fv);
    const float wx = fu - iu, wy = fv - iv;
    const int x0 = (int)iu, y0 = (int)iv;
    const float c00 = load_chroma(P, c, x0, y0), c10 = load_chroma(P, c, x0 + 1, y0);
    const float c01 = load_chroma(P, c, x0, y0 + 1), c11 = load_chroma(P, c, x0 + 1, y0 + 1);
    const float top = c00 * (1.0f - wx) + c10 * wx;
    const float bot = c01 * (1.0f - wx) + c11 * wx;
    return top * (1.0f - wy) + bot * wy;
}

__device__ __forceinline__ void catmull_weights(float t, float w[4])    // Shaders.cpp:66-72
{
    const float t2 = t * t, t3 = t * t2;
    w[0] = t2 - (t3 + t) / 2;
    w[1] = t3 * 1.5f + 1 - t2 * 2.5f;
    w[2] = t2 * 2 + t / 2 - t3 * 1.5f;
    w[3] = (t3 - t2) / 2;
}

__device__ __forceinline__ void fetch_chroma(const ConvertParams &P, int sx, int sy, float uv[2])
{
    const int sub = P.fmt.subsampling;
    if (P.chroma_scaling == 0 /*Nearest*/ || sub == 444) {               // :239-241,282-287
        const int cx = sx / P.fmt.div_w, cy = sy / P.fmt.div_h;
        uv[0] = load_chroma(P, 0, cx, cy); uv[1] = load_chroma(P, 1, cx, cy);
        return;
    }
    if (P.chroma_scaling == 2 /*CatmullRom*/ && sub == 420) {            // :242-251,288-299
        float tx = (sx & 1) ? 0.75f : 0.25f, ty = (sy & 1) ? 0.75f : 0.25f;
        if (P.chroma_loc == CLOC_COSITED) { tx += -0.25f; ty += -0.25f; }
        else if (P.chroma_loc == CLOC_MPEG1) { tx += -0.5f; ty += -0.5f; }
        else { tx += -0.25f; ty += -0.5f; }
        float wx[4], wy[4];
        catmull_weights(tx, wx); catmull_weights(ty, wy);
        const int bx = sx >> 1, by = sy >> 1;
        for (int c = 0; c < 2; c++) {
            float Q[4];
            for (int y = 0; y < 4; y++) {
                const float c0 = load_chroma(P, c, bx - 1, by + y - 1), c1 = load_chroma(P, c, bx, by + y - 1);
                const float c2 = load_chroma(P, c, bx + 1, by + y - 1), c3 = load_chroma(P, c, bx + 2, by + y - 1);
                Q[y] = c0 * wx[0] + c1 * wx[1] + c2 * wx[2] + c3 * wx[3];
            }
            uv[c] = Q[0] * wy[0] + Q[1] * wy[1] + Q[2] * wy[2] + Q[3] * wy[3];
        }
        return;
    }
    if (P.chroma_scaling == 2 && sub == 422) {                           // :252-264,300-318
        if ((sx & 1) == 0) {
            uv[0] = load_chroma(P, 0, sx >> 1, sy); uv[1] = load_chroma(P, 1, sx >> 1, sy);
        } else {
            const int k = (sx - 1) >> 1;
            for (int c = 0; c < 2; c++) {
                const float c0 = load_chroma(P, c, k - 1, sy), c1 = load_chroma(P, c, k, sy);
                const float c2 = load_chroma(P, c, k + 1, sy), c3 = load_chroma(P, c, k + 2, sy);
                uv[c] = (9 * (c1 + c2) - (c0 + c3)) * 0.0625f;
            }
        }
        return;
    }
    // CHROMA_Bilinear :265-270,319-325 — texUV.Sample(sampL, Tex + strChromaPos), coordinates in chroma texels
    float u = (sx + 0.5f) / (float)P.fmt.div_w, v = (sy + 0.5f) / (float)P.fmt.div_h;
    if (sub == 420) {
        if (P.chroma_loc == CLOC_COSITED) { u += 0.25f; v += 0.25f; }
        else if (P.chroma_loc == CLOC_MPEG2) { u += 0.25f; }
    } else {
        u += 0.25f;
    }
    uv[0] = sample_chroma_linear(P, 0, u, v);
    uv[1] = sample_chroma_linear(P, 1, u, v);
}

// (Y,U,V) — or (G,B,R) / (Y,0,0) — of source pixel (sx,sy): ShaderGetPixels' switch on the plane count / format
__device__ __forceinline__ void fetch_pixel(const ConvertParams &P, int sx, int sy, float &y, float uv[2])
{
    if (P.fmt.layout == LAY_PLANAR) {
        y = load_luma(P, sx, sy);                                         // :231,274
        if (P.blend_deint) {                                              // blendDeint420 :232-237,275-280
            const float y1 = load_luma(P, sx, sy - 1), y2 = load_luma(P, sx, sy + 1);
            y = (y * 2 + y1 + y2) / 4;
        }
        fetch_chroma(P, sx, sy, uv);
        return;
    }
    if (P.fmt.layout == LAY_GRAY) {        // float4 color = tex.Sample(samp, Tex) of an R8/R16 texture (:184)
        y = load_luma(P, sx, sy); uv[0] = 0; uv[1] = 0;
        return;
    }
    if (P.fmt.layout == LAY_RGB) {         // float4 color = tex.Sample(samp, Tex) (:184): (R,G,B) of the texture
        y = load_packed(P, sx, sy, P.fmt.ci[0]); uv[0] = load_packed(P, sx, sy, P.fmt.ci[1]); uv[1] = load_packed(P, sx, sy, P.fmt.ci[2]);
        return;
    }
    if (P.fmt.layout == LAY_PACKED444) {   // .zyxw (AYUV) / .yxzw (Y410, Y416) (:186-193)
        y = load_packed(P, sx, sy, P.fmt.ci[0]); uv[0] = load_packed(P, sx, sy, P.fmt.ci[1]); uv[1] = load_packed(P, sx, sy, P.fmt.ci[2]);
        return;
    }
    // packed 4:2:2 (:195-229): the even pixel takes the texel's own chroma, the odd pixel the mean with the next texel
    // (or CATMULLROM_05 over texels tx-1..tx+2); CHROMA_Nearest is not distinguished from Bilinear
    const int tx = sx >> 1;
    const int cu = P.fmt.ci[1], cv = P.fmt.ci[3];
    if ((sx & 1) == 0) {                   // fmod(Tex.x*w, 2) < 1.0
        y = load_packed(P, tx, sy, P.fmt.ci[0]); uv[0] = load_packed(P, tx, sy, cu); uv[1] = load_packed(P, tx, sy, cv);
        return;
    }
    y = load_packed(P, tx, sy, P.fmt.ci[2]);
    for (int c = 0; c < 2; c++) {
        const int k = c ? cv : cu;
        if (P.chroma_scaling == 2 /*CatmullRom*/) {
            const float c0 = load_packed(P, tx - 1, sy, k), c1 = load_packed(P, tx, sy, k);
            const float c2 = load_packed(P, tx + 1, sy, k), c3 = load_packed(P, tx + 2, sy, k);
            uv[c] = (9 * (c1 + c2) - (c0 + c3)) * 0.0625f;               // CATMULLROM_05 :145
        } else {
            uv[c] = (load_packed(P, tx, sy, k) + load_packed(P, tx + 1, sy, k)) * 0.5f;
        }
    }
}

// one output pixel of the generated convert shader (Shaders.cpp:593-930), before the RT store
__device__ __forceinline__ f3 convert_pixel(const ConvertParams &P, int i, int j)
{
    const int sx = P.rect_l + i, sy = P.rect_t + j;
    float y, uv[2];
    fetch_pixel(P, sx, sy, y, uv);
    if (P.dovi) {                                   // Shaders.cpp:786-792
        const f3 r = dovi_reshape(*P.dovi, f3{y, uv[0], uv[1]});
        y = r.x; uv[0] = r.y; uv[1] = r.z;
    }
    f3 c;
    c.x = (P.cm[0] * y + P.cm[1] * uv[0] + P.cm[2] * uv[1]) + P.cm[9];
    c.y = (P.cm[3] * y + P.cm[4] * uv[0] + P.cm[5] * uv[1]) + P.cm[10];
    c.z = (P.cm[6] * y + P.cm[7] * uv[0] + P.cm[8] * uv[1]) + P.cm[11];
    if (P.dovi) {
        c = dovi_lms_step(*P.dovi, c);
        return hdr_tail(c, P.tail, P.gamma, P.lum_scale, make_mat3(P.gamut), P.dovi->l2_enabled ? P.dovi->l2k : nullptr);
    }
    return hdr_tail(c, P.tail, P.gamma, P.lum_scale, make_mat3(P.gamut), nullptr, P.tail == TAIL_PQ_TO_SDR ? P.pq_lut : nullptr);
}

}  // namespace mpcvr
