// vp_crmath.h — the shader transcendentals as DEFINED functions, for the pass-per-kernel ("plain") tier.
//
// d3dcompiler lowers pow(x, y) to exp2(y * log2(x)) (Shaders/convert/st2084.hlsl:9-25, Shaders.cpp:895-922) and Direct3D leaves log2 /
// exp2 / exp a few ulp of freedom, so the reference does not define the last bits behind a PQ / HLG / gamma tail: one ulp of log2 is ~35
// ulp of pow(x, 1/m1 = 6.28).  The tier whose job is to agree with the CPU restatement bit for bit therefore takes every step as the
// CORRECTLY ROUNDED fp32 function — log2f = RN(log2 x), y * l in fp32, exp2f = RN(2^t), expf = RN(e^x) — which is the one definition no
// evaluator can disagree about.  It is evaluated in IEEE fp64 spelt out operation by operation (+, -, *, /, floor, bit moves; contraction
// off): accurate to ~4e-16, so the rounding to fp32 is the correct one on all but ~1e-8 of arguments, and any IEEE machine running the
// same sequence produces the same bits — the tests run this file's functions against a CPU evaluation of the same sequence and require
// equality on every argument (mpcvr_eval_transcendental).
//
//   log2: x = 2^k m, m in [sqrt(1/2), sqrt 2);  s = (m - 1) / (m + 1);  log2 m = (2 / ln 2) s (1 + z/3 + ... + z^10/21), z = s^2
//   exp2: t = n + f, n = floor(t + 1/2);  2^f = sum_{k <= 13} (f ln 2)^k / k!;  times 2^n
//   sin / cos (windowed sinc / jinc weights of the resize shaders): x = n pi/2 + r, three-piece pi/2, Taylor to r^17 / r^18
// The fused tiers do NOT use this (v_log_f32 / v_exp_f32 and LDS tables: speed); they are held to the +-1 LSB bar instead.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mpcvr {

#pragma clang fp contract(off)

// log2 of a positive finite float (subnormals included: every float is a normal double)
__host__ __device__ __forceinline__ double crm_log2_pos(float xf)
{
    const double x = (double)xf;
    const uint64_t b = (uint64_t)__builtin_bit_cast(int64_t, x);
    int k = (int)((b >> 52) & 0x7ff) - 1023;
    double m = __builtin_bit_cast(double, (int64_t)((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));      // [1, 2)
    if (m > 0x1.6a09e667f3bcdp+0) { m = m * 0.5; k = k + 1; }                                                  // [sqrt(1/2), sqrt 2): exact
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 0x1.8618618618618p-5;            // 1/21
    p = p * z + 0x1.af286bca1af28p-5;           // 1/19
    p = p * z + 0x1.e1e1e1e1e1e1ep-5;           // 1/17
    p = p * z + 0x1.1111111111111p-4;           // 1/15
    p = p * z + 0x1.3b13b13b13b14p-4;           // 1/13
    p = p * z + 0x1.745d1745d1746p-4;           // 1/11
    p = p * z + 0x1.c71c71c71c71cp-4;           // 1/9
    p = p * z + 0x1.2492492492492p-3;           // 1/7
    p = p * z + 0x1.999999999999ap-3;           // 1/5
    p = p * z + 0x1.5555555555555p-2;           // 1/3
    const double r = s + s * (z * p);           // atanh s
    return (double)k + r * 0x1.71547652b82fep+1;        // k + (2 / ln 2) atanh s
}

// 2^t as a double; t clamped to where an fp32 result is 0 or inf for certain
__host__ __device__ __forceinline__ double crm_exp2_d(double t)
{
    if (t > 200.0) t = 200.0;
    if (t < -200.0) t = -200.0;
    const double n = __builtin_floor(t + 0.5);
    const double u = (t - n) * 0x1.62e42fefa39efp-1;    // f ln 2, |f| <= 1/2 (t - n is exact)
    double p = 0x1.6124613a86d09p-33;           // 1/13!
    p = p * u + 0x1.1eed8eff8d898p-29;          // 1/12!
    p = p * u + 0x1.ae64567f544e4p-26;          // 1/11!
    p = p * u + 0x1.27e4fb7789f5cp-22;          // 1/10!
    p = p * u + 0x1.71de3a556c734p-19;          // 1/9!
    p = p * u + 0x1.a01a01a01a01ap-16;          // 1/8!
    p = p * u + 0x1.a01a01a01a01ap-13;          // 1/7!
    p = p * u + 0x1.6c16c16c16c17p-10;          // 1/6!
    p = p * u + 0x1.1111111111111p-7;           // 1/5!
    p = p * u + 0x1.5555555555555p-5;           // 1/4!
    p = p * u + 0x1.5555555555555p-3;           // 1/3!
    p = p * u + 0.5;
    p = p * u + 1.0;
    p = p * u + 1.0;
    return p * __builtin_bit_cast(double, (int64_t)(((int64_t)n + 1023) << 52));        // times 2^n: exact
}

__host__ __device__ __forceinline__ float crm_log2f(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    return (float)crm_log2_pos(x);
}
__host__ __device__ __forceinline__ float crm_exp2f(float t)
{
    if (t != t) return t;
    return (float)crm_exp2_d((double)t);        // round to nearest even, gradual underflow, overflow to inf
}
__host__ __device__ __forceinline__ float crm_expf(float x)
{
    if (x != x) return x;
    return (float)crm_exp2_d((double)x * 0x1.71547652b82fep+0);
}
/* sin / cos of a float argument as a double: x = n (pi/2) + r with pi/2 in three pieces (33 + 33 + 53 bits: n P1 and n P2 are exact for
 * |n| < 2^20, i.e. |x| < 1.6e6 — the path's arguments are below 10), |r| <= pi/4, Taylor polynomials to r^17 / r^18 (first dropped
 * terms: 8e-20, 3e-21).  quadrant: 0 sin, 1 cos of the same argument. */
__host__ __device__ __forceinline__ double crm_sincos_d(double x, int quadrant)
{
    const double n = __builtin_floor(x * 0x1.45f306dc9c883p-1 + 0.5);
    double r = x - n * 0x1.921fb54400000p+0;
    r = r - n * 0x1.0b4611a600000p-34;
    r = r - n * 0x1.3198a2e037073p-69;
    const int q = (int)(((long long)n + quadrant) & 3);
    const double z = r * r;
    double v;
    if (q & 1) {                                 // cos r
        double p = -0x1.6827863b97d97p-53;       // -1/18!
        p = p * z + 0x1.ae7f3e733b81fp-45;       // 1/16!
        p = p * z + -0x1.93974a8c07c9dp-37;      // -1/14!
        p = p * z + 0x1.1eed8eff8d898p-29;       // 1/12!
        p = p * z + -0x1.27e4fb7789f5cp-22;      // -1/10!
        p = p * z + 0x1.a01a01a01a01ap-16;       // 1/8!
        p = p * z + -0x1.6c16c16c16c17p-10;      // -1/6!
        p = p * z + 0x1.5555555555555p-5;        // 1/4!
        p = p * z + -0.5;
        v = 1.0 + z * p;
    } else {                                     // sin r
        double p = 0x1.952c77030ad4ap-49;        // 1/17!
        p = p * z + -0x1.ae7f3e733b81fp-41;      // -1/15!
        p = p * z + 0x1.6124613a86d09p-33;       // 1/13!
        p = p * z + -0x1.ae64567f544e4p-26;      // -1/11!
        p = p * z + 0x1.71de3a556c734p-19;       // 1/9!
        p = p * z + -0x1.a01a01a01a01ap-13;      // -1/7!
        p = p * z + 0x1.1111111111111p-7;        // 1/5!
        p = p * z + -0x1.5555555555555p-3;       // -1/3!
        v = r + r * (z * p);
    }
    return (q & 2) ? -v : v;
}
// sinf / cosf: RN(sin x), RN(cos x) for |x| < 1.6e6 (beyond that the same sequence, no longer accurate: not an argument of this path)
__host__ __device__ __forceinline__ float crm_sinf(float x)
{
    if (x != x || x == __builtin_inff() || x == -__builtin_inff()) return __builtin_nanf("");
    if (x == 0.0f) return x;
    return (float)crm_sincos_d((double)x, 0);
}
__host__ __device__ __forceinline__ float crm_cosf(float x)
{
    if (x != x || x == __builtin_inff() || x == -__builtin_inff()) return __builtin_nanf("");
    return (float)crm_sincos_d((double)x, 1);
}
// HLSL pow as d3dcompiler emits it: exp2(y * log2(x)), every step rounded to fp32
__host__ __device__ __forceinline__ float crm_powf(float x, float y)
{
    return crm_exp2f(y * crm_log2f(x));
}

}  // namespace mpcvr
