// vp_device.h — device functions shared by all kernels: texel loads, the HLSL include library
// restated for CDNA4 (Shaders/convert/*.hlsl), store-format rounding, ps_final_pass.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "vp_params.h"
#ifdef MPCVR_EXACT_FP
#include "vp_crmath.h"
#endif

namespace mpcvr {

struct f3 { float x, y, z; };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }   // NaN -> 0

// UNORM load: code / maxv, correctly rounded, for an INTEGER code in [0, maxv] and maxv in {255, 1023, 65535}.
// q = code * (1/maxv) followed by one Newton correction with two FMAs equals the IEEE division for every such code
// (exhaustively checked, tests/test_host_logic.py::test_unorm_division_shortcut_is_exact) at 3 instructions instead of
// the ~10 of the division expansion; results stay bit-identical to the oracle's `code / maxv`.
template <int MAXV>
__device__ __forceinline__ float unorm_div(float code)
{
    constexpr float d = (float)MAXV, r = 1.0f / (float)MAXV;
    const float q = code * r;
    return __builtin_fmaf(__builtin_fmaf(-q, d, code), r, q);
}

// The transfer / tone-map chain below is ill-conditioned where the 2020->709 matrix cancels to ~0 and
// pow(1/2.2) then magnifies the residue, so its expression shapes are kept identical in every
// translation unit (no FMA contraction): the fused and the pass-per-kernel path then agree bit for bit.
#pragma clang fp contract(off)

// HLSL pow(x, y) = exp2(y * log2(x)).  The pass-per-kernel tier (vp_kernels.hip, MPCVR_EXACT_FP) takes every step as the correctly
// rounded fp32 function (vp_crmath.h: the definition the CPU restatement uses, so the two agree bit for bit behind a PQ / HLG / gamma
// tail too); everywhere else raw v_log_f32 / v_exp_f32 (≈1 ulp each; behind pow(x, 6.28) that is ~35 ulp), log2(0) = -inf -> 0.
__device__ __forceinline__ float hlsl_pow(float x, float y)
{
#ifdef MPCVR_EXACT_FP
    return crm_powf(x, y);
#else
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
#endif
}
__device__ __forceinline__ float hlsl_exp(float x)
{
#ifdef MPCVR_EXACT_FP
    return crm_expf(x);
#else
    return __expf(x);
#endif
}

// ---- Shaders/convert/st2084.hlsl:1-25 ----
#define MPCVR_ST2084_m1 (2610.0f / (4096.0f * 4.0f))
#define MPCVR_ST2084_m2 ((2523.0f / 4096.0f) * 128.0f)
#define MPCVR_ST2084_c1 (3424.0f / 4096.0f)
#define MPCVR_ST2084_c2 ((2413.0f / 4096.0f) * 32.0f)
#define MPCVR_ST2084_c3 ((2392.0f / 4096.0f) * 32.0f)

__device__ __forceinline__ float st2084_to_linear(float x, float factor)
{
    x = hlsl_pow(x, 1.0f / MPCVR_ST2084_m2);
    x = fmaxf(x - MPCVR_ST2084_c1, 0.0f) / (MPCVR_ST2084_c2 - MPCVR_ST2084_c3 * x);
    x = hlsl_pow(x, 1.0f / MPCVR_ST2084_m1);
    return x * factor;
}
__device__ __forceinline__ float linear_to_st2084(float x, float divider)
{
    x = x / divider;
    x = hlsl_pow(x, MPCVR_ST2084_m1);
    x = (MPCVR_ST2084_c1 + MPCVR_ST2084_c2 * x) / (1.0f + MPCVR_ST2084_c3 * x);
    return hlsl_pow(x, MPCVR_ST2084_m2);
}

// ---- Shaders/convert/hlg.hlsl:1-20 ----
__device__ __forceinline__ float inverse_hlg(float v)
{
    const float a = 0.17883277f, b = 0.28466892f, c = 0.55991073f;
    return (v <= 0.5f) ? v * v * 4.0f : hlsl_exp((v - c) / a) + b;
}
__device__ __forceinline__ f3 hlg_to_linear(f3 v)
{
    v.x = inverse_hlg(v.x); v.y = inverse_hlg(v.y); v.z = inverse_hlg(v.z);
    const float ys = 2000.0f * (0.2627f * v.x + 0.6780f * v.y + 0.0593f * v.z);
    const float g = hlsl_pow(ys, 0.2f);
    v.x *= g; v.y *= g; v.z *= g;
    return v;
}

// ---- Shaders/convert/hdr_tone_mapping.hlsl:1-13 ----
__device__ __forceinline__ float hable(float x)
{
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + (C * B)) + (D * E)) / (x * (A * x + B) + (D * F))) - E / F;
}
// hable(4.8) folded in fp32 exactly like `static const float3 HABLE_DIV = hable(4.8)`
__device__ __forceinline__ float hable_div()
{
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f, x = 4.8f;
    return ((x * (A * x + (C * B)) + (D * E)) / (x * (A * x + B) + (D * F))) - E / F;
}

struct mat3 { float m[9]; };          // by value: keeps kernel-argument matrices in SGPRs (no address taken)
__device__ __forceinline__ mat3 make_mat3(const float (&a)[9])
{
    mat3 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.m[i] = a[i];
    return r;
}
__device__ __forceinline__ f3 mat3_mul(const mat3 &g, f3 v)
{
    f3 r;
    r.x = g.m[0] * v.x + g.m[1] * v.y + g.m[2] * v.z;
    r.y = g.m[3] * v.x + g.m[4] * v.y + g.m[5] * v.z;
    r.z = g.m[6] * v.x + g.m[7] * v.y + g.m[8] * v.z;
    return r;
}

// ---- Dolby Vision ----
// reshape_mmr — Shaders.cpp:734-762 (dot products summed left to right like mul())
__device__ __forceinline__ float dovi_reshape_mmr(const DoviCurve &cv, const float co[4], f3 sig)
{
    const uint32_t at = cv.mmr_single ? 0u : (uint32_t)co[1];
    const float (*w)[4] = cv.mmr + at;
    float s = co[0];
    float cx[4] = {sig.x * sig.y, sig.x * sig.z, sig.y * sig.z, 0.0f};
    cx[3] = cx[0] * sig.z;
    s += w[0][0] * sig.x + w[0][1] * sig.y + w[0][2] * sig.z;
    s += w[1][0] * cx[0] + w[1][1] * cx[1] + w[1][2] * cx[2] + w[1][3] * cx[3];
    if (cv.max_order >= 2) {
        const uint32_t order = (uint32_t)co[3];
        if (cv.min_order < 2 && order < 2) return s;
        const f3 s2 = {sig.x * sig.x, sig.y * sig.y, sig.z * sig.z};
        float cx2[4];
#pragma unroll
        for (int i = 0; i < 4; i++) cx2[i] = cx[i] * cx[i];
        s += w[2][0] * s2.x + w[2][1] * s2.y + w[2][2] * s2.z;
        s += w[3][0] * cx2[0] + w[3][1] * cx2[1] + w[3][2] * cx2[2] + w[3][3] * cx2[3];
        if (cv.max_order == 3) {
            if (cv.min_order < 3 && order < 3) return s;
            s += w[4][0] * (s2.x * sig.x) + w[4][1] * (s2.y * sig.y) + w[4][2] * (s2.z * sig.z);
            s += w[5][0] * (cx2[0] * cx[0]) + w[5][1] * (cx2[1] * cx[1]) + w[5][2] * (cx2[2] * cx[2]) + w[5][3] * (cx2[3] * cx[3]);
        }
    }
    return s;
}
// ShaderDoviReshape / ShaderDoviReshapePoly — Shaders.cpp:531-589: (Y,U,V) through the per-component piecewise curves
__device__ __forceinline__ f3 dovi_reshape(const DoviParams &D, f3 color)
{
    const f3 sig = {saturate(color.x), saturate(color.y), saturate(color.z)};
    float in[3] = {sig.x, sig.y, sig.z}, out[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const DoviCurve &cv = D.curves[c];
        float s = in[c];
        const float *pv = cv.pivots;
        const int piece = (s < pv[3]) ? ((s < pv[1]) ? ((s < pv[0]) ? 0 : 1) : ((s < pv[2]) ? 2 : 3))
                                      : ((s < pv[5]) ? ((s < pv[4]) ? 4 : 5) : ((s < pv[6]) ? 6 : 7));
        const float4 q = *reinterpret_cast<const float4 *>(cv.coeffs[piece]);
        const float co[4] = {q.x, q.y, q.z, q.w};
        const bool poly = !D.has_mmr || cv.methods == DOVI_RESHAPE_POLY ||
                          (cv.methods == DOVI_RESHAPE_POLY + DOVI_RESHAPE_MMR && co[3] == 0.0f);
        if (poly) s = (co[2] * s + co[1]) * s + co[0];
        else s = dovi_reshape_mmr(cv, co, sig);
        out[c] = saturate(s);
    }
    return f3{out[0], out[1], out[2]};
}
// PQ EOTF -> LMS matrix -> PQ OETF — Shaders.cpp:844-859
__device__ __forceinline__ f3 dovi_lms_step(const DoviParams &D, f3 c)
{
    c.x = st2084_to_linear(fmaxf(c.x, 0.0f), 1.0f); c.y = st2084_to_linear(fmaxf(c.y, 0.0f), 1.0f); c.z = st2084_to_linear(fmaxf(c.z, 0.0f), 1.0f);
    f3 r;
    r.x = D.lms[0] * c.x + D.lms[1] * c.y + D.lms[2] * c.z;
    r.y = D.lms[3] * c.x + D.lms[4] * c.y + D.lms[5] * c.z;
    r.z = D.lms[6] * c.x + D.lms[7] * c.y + D.lms[8] * c.z;
    r.x = linear_to_st2084(fmaxf(r.x, 0.0f), 1.0f); r.y = linear_to_st2084(fmaxf(r.y, 0.0f), 1.0f); r.z = linear_to_st2084(fmaxf(r.z, 0.0f), 1.0f);
    return r;
}
// DolbyVisionTrims on PQ-coded colour — Shaders.cpp:766-773; k = {ChromaWeight, SaturationGain, TrimSlope, TrimOffset, TrimPower}
__device__ __forceinline__ f3 dovi_trims(f3 c, const float *k)
{
    c.x = hlsl_pow((c.x * k[2]) + k[3], k[4]); c.y = hlsl_pow((c.y * k[2]) + k[3], k[4]); c.z = hlsl_pow((c.z * k[2]) + k[3], k[4]);
    const float Y = 0.2627f * c.x + 0.6780f * c.y + 0.0593f * c.z;
    c.x = c.x * hlsl_pow((1.0f + k[0]) * c.x / Y, k[1]);
    c.y = c.y * hlsl_pow((1.0f + k[0]) * c.y / Y, k[1]);
    c.z = c.z * hlsl_pow((1.0f + k[0]) * c.z / Y, k[1]);
    return c;
}

// The tail GetShaderConvertColor appends after "//convert color" — Shaders.cpp:861-923; l2k != null: Dolby Vision L2 trims (:873-877)
// hable(ST2084ToLinear(x, scale)) / hable(4.8) from the 4096-entry table the fused kernel uses, evaluated the same way
// (linear interpolation, one FMA): replaces 4 transcendentals and 3 divisions per channel
__device__ __forceinline__ float pq_sdr_lut(const float *__restrict__ lut, float x)
{
    const float t = x * (float)(kPqLutSize - 1);
    const int i = (int)t;
    const float v = lut[i], n = lut[min(i + 1, kPqLutSize - 1)];
    return __builtin_fmaf(n - v, t - (float)i, v);
}

__device__ __forceinline__ f3 hdr_tail(f3 c, int tail, float gamma, float lum_scale, const mat3 &gamut, const float *l2k = nullptr,
                                       const float *pq_lut = nullptr)
{
    if (tail == TAIL_NONE) return c;
    if (tail == TAIL_HLG_TO_PQ) {          // bConvertHLGtoPQ (:885-891)
        c.x = saturate(c.x); c.y = saturate(c.y); c.z = saturate(c.z);
        c = hlg_to_linear(c);
        c.x = linear_to_st2084(c.x, 1000.0f); c.y = linear_to_st2084(c.y, 1000.0f); c.z = linear_to_st2084(c.z, 1000.0f);
        return c;
    }
    if (tail == TAIL_PQ_TO_SDR || tail == TAIL_HLG_TO_SDR) {
        if (tail == TAIL_HLG_TO_SDR) {
            c.x = saturate(c.x); c.y = saturate(c.y); c.z = saturate(c.z);
            c = hlg_to_linear(c);
            c.x = linear_to_st2084(c.x, 1000.0f); c.y = linear_to_st2084(c.y, 1000.0f); c.z = linear_to_st2084(c.z, 1000.0f);
        }
        c.x = saturate(c.x); c.y = saturate(c.y); c.z = saturate(c.z);
        if (l2k) c = dovi_trims(c, l2k);
        if (pq_lut && !l2k) {
            c.x = pq_sdr_lut(pq_lut, c.x); c.y = pq_sdr_lut(pq_lut, c.y); c.z = pq_sdr_lut(pq_lut, c.z);
        } else {
            c.x = st2084_to_linear(c.x, lum_scale);
            c.y = st2084_to_linear(c.y, lum_scale);
            c.z = st2084_to_linear(c.z, lum_scale);
            const float div = hable_div();
            c.x = hable(c.x) / div; c.y = hable(c.y) / div; c.z = hable(c.z) / div;
        }
        c = mat3_mul(gamut, c);
    } else {   // TAIL_GAMMA_GAMUT
        c.x = saturate(c.x); c.y = saturate(c.y); c.z = saturate(c.z);
        if (gamma != 1.0f) { c.x = hlsl_pow(c.x, gamma); c.y = hlsl_pow(c.y, gamma); c.z = hlsl_pow(c.z, gamma); }
        c = mat3_mul(gamut, c);
    }
    c.x = hlsl_pow(saturate(c.x), 1.0f / 2.2f);
    c.y = hlsl_pow(saturate(c.y), 1.0f / 2.2f);
    c.z = hlsl_pow(saturate(c.z), 1.0f / 2.2f);
    return c;
}

// HDR10 -> HDR10 local tone mapping — Shaders/d3d11/ps_hdr10_tonemap.hlsl:257-336
__device__ __forceinline__ float lerp_f(float a, float b, float t) { return a + t * (b - a); }
__device__ __forceinline__ float pl_smoothstep(float e0, float e1, float x)
{
    float t = (x - e0) / (e1 - e0);
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    return t * t * (3.0f - 2.0f * t);
}
__device__ __forceinline__ f3 hdr10_tonemap(f3 c, const HdrToneMapParams &k)
{
    c.x = st2084_to_linear(saturate(c.x), 10000.0f); c.y = st2084_to_linear(saturate(c.y), 10000.0f); c.z = st2084_to_linear(saturate(c.z), 10000.0f);
    if (k.l2_enabled) {                                                                     // DolbyVisionTrims :257-270
        c.x = linear_to_st2084(c.x, 10000.0f); c.y = linear_to_st2084(c.y, 10000.0f); c.z = linear_to_st2084(c.z, 10000.0f);
        c = dovi_trims(c, k.l2k);
        c.x = st2084_to_linear(c.x, 10000.0f); c.y = st2084_to_linear(c.y, 10000.0f); c.z = st2084_to_linear(c.z, 10000.0f);
    }
    if (k.selection == 5) {                                                                 // BT2390Tonemap :68-124
        float safe = k.max_cll;
        if (safe <= 10.0f) safe = k.max_mastering;
        if (safe <= 10.0f) safe = 1000.0f;
        if (!(k.display_max >= safe)) {
            const float avg = 0.2627f * c.x + 0.6780f * c.y + 0.0593f * c.z;
            if (!(avg <= 0.000001f)) {
                const float max_pq = linear_to_st2084(safe, 10000.0f), tgt_pq = linear_to_st2084(k.display_max, 10000.0f);
                const float e1 = linear_to_st2084(avg, 10000.0f);
                float ks = 1.5f * tgt_pq - 0.5f * max_pq;
                ks = fmaxf(0.0f, ks);
                float e2 = e1;
                if (e1 > ks) {
                    const float t = (e1 - ks) / fmaxf(1e-6f, max_pq - ks), t2 = t * t, t3 = t2 * t;
                    e2 = (2.0f * t3 - 3.0f * t2 + 1.0f) * ks + (t3 - 2.0f * t2 + t) * (max_pq - ks) + (-2.0f * t3 + 3.0f * t2) * tgt_pq;
                }
                const float lin = st2084_to_linear(e2, 10000.0f);
                const float g = lin / avg;
                c.x = c.x * g; c.y = c.y * g; c.z = c.z * g;
            }
        }
        c.x = linear_to_st2084(c.x, 10000.0f); c.y = linear_to_st2084(c.y, 10000.0f); c.z = linear_to_st2084(c.z, 10000.0f);
        return c;
    }
    if (k.selection == 6) {                                                                 // ST209410Tonemap :133-205
        if (!(k.display_max >= k.max_cll)) {
            const float src_min = linear_to_st2084(k.min_mastering, 10000.0f), src_max = linear_to_st2084(k.max_cll, 10000.0f);
            const float src_avg = linear_to_st2084(k.max_fall, 10000.0f);
            const float dst_min = linear_to_st2084(0.0f, 10000.0f), dst_max = linear_to_st2084(k.display_max, 10000.0f);
            const float min_knee = 0.1f, max_knee = 0.8f, def_knee = 0.4f, knee_adaptation = 0.4f;
            const float src_knee_min = lerp_f(src_min, src_max, min_knee), src_knee_max = lerp_f(src_min, src_max, max_knee);
            const float dst_knee_min = lerp_f(dst_min, dst_max, min_knee), dst_knee_max = lerp_f(dst_min, dst_max, max_knee);
            float src_knee = (k.max_fall > 0.0f) ? src_avg : lerp_f(src_min, src_max, def_knee);
            src_knee = fminf(fmaxf(src_knee, src_knee_min), src_knee_max);
            const float target = (src_knee - src_min) / (src_max - src_min);
            const float adapted = lerp_f(dst_min, dst_max, target);
            const float tuning = 1.0f - pl_smoothstep(max_knee, def_knee, target) * pl_smoothstep(min_knee, def_knee, target);
            const float adaptation = lerp_f(knee_adaptation, 1.0f, tuning);
            float dst_knee = lerp_f(src_knee, adapted, adaptation);
            dst_knee = fminf(fmaxf(dst_knee, dst_knee_min), dst_knee_max);
            const float x2 = st2084_to_linear(src_knee, 10000.0f), y2 = st2084_to_linear(dst_knee, 10000.0f);
            const float x1 = k.min_mastering, x3 = k.max_cll, y1 = 0.0f, y3 = k.display_max;
            const float m00 = x2 * x3 * (y2 - y3), m01 = x1 * x3 * (y3 - y1), m02 = x1 * x2 * (y1 - y2);
            const float m10 = x3 * y3 - x2 * y2, m11 = x1 * y1 - x3 * y3, m12 = x2 * y2 - x1 * y1;
            const float m20 = x3 - x2, m21 = x1 - x3, m22 = x2 - x1;
            const float coef0 = m00 * y1 + m01 * y2 + m02 * y3, coef1 = m10 * y1 + m11 * y2 + m12 * y3, coef2 = m20 * y1 + m21 * y2 + m22 * y3;
            const float kk = 1.0f / (x3 * y3 * (x1 - x2) + x2 * y2 * (x3 - x1) + x1 * y1 * (x2 - x3));
            const float c1 = kk * coef0, c2 = kk * coef1, c3 = kk * coef2;
            const float xn = 0.2627f * c.x + 0.6780f * c.y + 0.0593f * c.z;
            const float yn = (c1 + c2 * xn) / (1.0f + c3 * xn);
            const float g = (xn > 0.0f) ? (yn / xn) : 1.0f;
            c.x = c.x * g; c.y = c.y * g; c.z = c.z * g;
        }
        c.x = linear_to_st2084(c.x, 10000.0f); c.y = linear_to_st2084(c.y, 10000.0f); c.z = linear_to_st2084(c.z, 10000.0f);
        return c;
    }
    const float base = fmaxf(k.display_max, k.max_mastering);                               // :299-306
    const float eff = fminf(base, k.max_cll);
    const float fall = fminf(base / k.max_fall, 1.0f);
    float v[3] = {c.x, c.y, c.z};
    for (int i = 0; i < 3; i++) {
        float t = v[i] * (1.0f / eff);
        t = saturate(t);
        t = t * fall;
        if (k.selection == 2) t = t / (1.0f + t);
        else if (k.selection == 3) {
            const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
            t = ((t * (A * t + C * B) + D * E) / (t * (A * t + B) + D * F)) - E / F;
        } else if (k.selection == 4) t = t / (1.0f + t / (k.display_max + 1e-6f));
        else t = (t * (2.51f * t + 0.03f)) / (t * (2.43f * t + 0.59f) + 0.14f);
        t = t * k.display_max;
        v[i] = linear_to_st2084(t, 10000.0f);
    }
    c.x = v[0]; c.y = v[1]; c.z = v[2];
    return c;
}

#ifndef MPCVR_EXACT_FP              // vp_kernels.hip is built -ffp-contract=off -DMPCVR_EXACT_FP and stays off
#pragma clang fp contract(fast)     // back to the compiler default for HIP device code
#endif

// ---- source texel loads: UNORM8/16 -> float, clamp addressing, CopyPlane10to16 shift on the fly ----
__device__ __forceinline__ float load_sample(const uint8_t *plane, int pitch, int bytes, int shift, int x, int y)
{
    const uint8_t *row = plane + (size_t)y * pitch;
    if (bytes == 1) return unorm_div<255>((float)row[x]);
    const unsigned v = (((const uint16_t *)row)[x] << shift) & 0xffffu;
    return unorm_div<65535>((float)v);
}
__device__ __forceinline__ float load_luma(const ConvertParams &P, int x, int y)
{
    x = clampi(x, 0, P.tex_w - 1); y = clampi(y, 0, P.tex_h - 1);
    return load_sample(P.plane[0], P.pitch[0], P.fmt.bytes, P.fmt.shift, x, y);
}
// c: 0 = U, 1 = V   (constant plane indices only: dynamic indexing would push the params into scratch)
__device__ __forceinline__ float load_chroma(const ConvertParams &P, int c, int x, int y)
{
    x = clampi(x, 0, P.cw - 1); y = clampi(y, 0, P.ch - 1);
    if (P.fmt.planes == 2)      // interleaved UV: no shift (P010/P016/P21x carry MSB-aligned data)
        return load_sample(P.plane[1], P.pitch[1], P.fmt.bytes, 0, 2 * x + c, y);
    const bool second = (c == 0) == (P.fmt.v_first != 0);                       // Shaders.cpp:159-165
    const uint8_t *pl = second ? P.plane[2] : P.plane[1];
    return load_sample(pl, P.pitch[1], P.fmt.bytes, P.fmt.shift, x, y);
}

// one-plane formats: component k of texel (tx,y) of the RGBA8 / RGBA16 / R10G10B10A2 texture, clamp addressing
__device__ __forceinline__ float load_packed(const ConvertParams &P, int tx, int y, int k)
{
    const int tw = (P.fmt.layout == LAY_PACKED422) ? P.tex_w / 2 : P.tex_w;
    tx = clampi(tx, 0, tw - 1); y = clampi(y, 0, P.tex_h - 1);
    const uint8_t *row = P.plane[0] + (size_t)y * P.pitch[0];
    if (P.fmt.bits10) {
        const uint32_t d = ((const uint32_t *)row)[tx];
        return unorm_div<1023>((float)((d >> (10 * k)) & 0x3ffu));
    }
    if (P.fmt.bytes == 1) return unorm_div<255>((float)row[4 * tx + k]);
    return unorm_div<65535>((float)((const uint16_t *)row)[4 * tx + k]);
}

// ---- store-format rounding ----
__device__ __forceinline__ float unorm_q(float x, float maxv) { return floorf(saturate(x) * maxv + 0.5f); }
__device__ __forceinline__ float half_round(float x) { return __half2float(__float2half_rn(x)); }

// value a texture of format fmt returns after `v` was written to it
__device__ __forceinline__ f3 round_to_fmt(f3 v, int fmt)
{
    if (fmt == SF_BGRA8) { v.x = unorm_div<255>(unorm_q(v.x, 255.0f)); v.y = unorm_div<255>(unorm_q(v.y, 255.0f)); v.z = unorm_div<255>(unorm_q(v.z, 255.0f)); }
    else if (fmt == SF_RGB10A2) { v.x = unorm_div<1023>(unorm_q(v.x, 1023.0f)); v.y = unorm_div<1023>(unorm_q(v.y, 1023.0f)); v.z = unorm_div<1023>(unorm_q(v.z, 1023.0f)); }
    else if (fmt == SF_RGBA16) { v.x = unorm_div<65535>(unorm_q(v.x, 65535.0f)); v.y = unorm_div<65535>(unorm_q(v.y, 65535.0f)); v.z = unorm_div<65535>(unorm_q(v.z, 65535.0f)); }
    else { v.x = half_round(v.x); v.y = half_round(v.y); v.z = half_round(v.z); }
    return v;
}

__device__ __forceinline__ uint32_t pack_bgra8(float r, float g, float b)   // already integral 0..255
{
    return (uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16) | 0xff000000u;
}
__device__ __forceinline__ uint32_t pack_rgb10a2(float r, float g, float b)
{
    return (uint32_t)r | ((uint32_t)g << 10) | ((uint32_t)b << 20) | 0xc0000000u;
}

// write v (float RGB, alpha = 1) into a surface of format fmt
__device__ __forceinline__ void store_surface(void *base, int pitch, int fmt, int x, int y, f3 v)
{
    uint8_t *row = (uint8_t *)base + (size_t)y * pitch;
    if (fmt == SF_BGRA8) ((uint32_t *)row)[x] = pack_bgra8(unorm_q(v.x, 255.0f), unorm_q(v.y, 255.0f), unorm_q(v.z, 255.0f));
    else if (fmt == SF_RGB10A2) ((uint32_t *)row)[x] = pack_rgb10a2(unorm_q(v.x, 1023.0f), unorm_q(v.y, 1023.0f), unorm_q(v.z, 1023.0f));
    else {
        const __half2 lo = __halves2half2(__float2half_rn(v.x), __float2half_rn(v.y));
        const __half2 hi = __halves2half2(__float2half_rn(v.z), __float2half_rn(1.0f));
        uint2 u;
        u.x = *(const uint32_t *)&lo; u.y = *(const uint32_t *)&hi;
        ((uint2 *)row)[x] = u;
    }
}

__device__ __forceinline__ f3 load_surface(const Surface &s, int x, int y)
{
    const uint8_t *row = (const uint8_t *)s.ptr + (size_t)y * s.pitch;
    f3 v;
    if (s.fmt == SF_BGRA8) {
        const uint32_t u = ((const uint32_t *)row)[x];
        v.x = unorm_div<255>((float)((u >> 16) & 255u)); v.y = unorm_div<255>((float)((u >> 8) & 255u)); v.z = unorm_div<255>((float)(u & 255u));
    } else if (s.fmt == SF_RGB10A2) {
        const uint32_t u = ((const uint32_t *)row)[x];
        v.x = unorm_div<1023>((float)(u & 1023u)); v.y = unorm_div<1023>((float)((u >> 10) & 1023u)); v.z = unorm_div<1023>((float)((u >> 20) & 1023u));
    } else if (s.fmt == SF_RGBA16) {
        const uint2 u = ((const uint2 *)row)[x];
        v.x = unorm_div<65535>((float)(u.x & 0xffffu)); v.y = unorm_div<65535>((float)(u.x >> 16)); v.z = unorm_div<65535>((float)(u.y & 0xffffu));
    } else {
        const uint2 u = ((const uint2 *)row)[x];
        const __half2 lo = *(const __half2 *)&u.x, hi = *(const __half2 *)&u.y;
        v.x = __low2float(lo); v.y = __high2float(lo); v.z = __low2float(hi);
    }
    return v;
}

// load_surface split in two, so that a kernel can have the texels of its next row in flight while it filters the current one
template <int FMT> __device__ __forceinline__ uint2 load_texel_raw(const Surface &s, int x, int y)
{
    const uint8_t *row = (const uint8_t *)s.ptr + (size_t)y * s.pitch;
    if (FMT == SF_BGRA8 || FMT == SF_RGB10A2) return uint2{((const uint32_t *)row)[x], 0u};
    return ((const uint2 *)row)[x];
}
template <int FMT> __device__ __forceinline__ f3 decode_texel(uint2 u)
{
    f3 v;
    if (FMT == SF_BGRA8) {
        v.x = unorm_div<255>((float)((u.x >> 16) & 255u)); v.y = unorm_div<255>((float)((u.x >> 8) & 255u)); v.z = unorm_div<255>((float)(u.x & 255u));
    } else if (FMT == SF_RGB10A2) {
        v.x = unorm_div<1023>((float)(u.x & 1023u)); v.y = unorm_div<1023>((float)((u.x >> 10) & 1023u)); v.z = unorm_div<1023>((float)((u.x >> 20) & 1023u));
    } else if (FMT == SF_RGBA16) {
        v.x = unorm_div<65535>((float)(u.x & 0xffffu)); v.y = unorm_div<65535>((float)(u.x >> 16)); v.z = unorm_div<65535>((float)(u.y & 0xffffu));
    } else {
        const __half2 lo = *(const __half2 *)&u.x, hi = *(const __half2 *)&u.y;
        v.x = __low2float(lo); v.y = __high2float(lo); v.z = __low2float(hi);
    }
    return v;
}

// epilogue of the last draw: plain surface store, or m_TexsPostScale rounding + ps_final_pass.hlsl:23-31
// (dither_pre: the dither texel of this pixel, fetched earlier by the caller; < 0 = fetch here)
__device__ __forceinline__ void store_epilogue(const StoreParams &S, int x, int y, f3 v, int dither_pre = -1)
{
    const int wx = x + S.off_x, wy = y + S.off_y;
    if (S.clip_w > 0 && (wx < 0 || wy < 0 || wx >= S.clip_w || wy >= S.clip_h)) return;
    if (S.mode == ST_SURFACE) { store_surface(S.dst, S.dst_pitch, S.dst_fmt, wx, wy, v); return; }
    v = round_to_fmt(v, S.mid_fmt);
    // sampler WRAP + POINT, ditherCoordScale = texSize/32  =>  texel (wx mod 32, wy mod 32)
    const float d = __half2float(__ushort_as_half(dither_pre >= 0 ? (unsigned short)dither_pre : S.dither[(wy & 31) * 32 + (wx & 31)]));
    const float q = (float)S.quant;
    const float r = fminf(fmaxf(floorf(v.x * q + d), 0.0f), q);
    const float g = fminf(fmaxf(floorf(v.y * q + d), 0.0f), q);
    const float b = fminf(fmaxf(floorf(v.z * q + d), 0.0f), q);
    uint32_t *row = (uint32_t *)((uint8_t *)S.dst + (size_t)wy * S.dst_pitch);
    row[wx] = (S.dst_fmt == SF_RGB10A2) ? pack_rgb10a2(r, g, b) : pack_bgra8(r, g, b);
}

}  // namespace mpcvr
