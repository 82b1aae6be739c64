// ref_runtime.cpp — shared state of the compiled reference shaders (TEST INFRASTRUCTURE ONLY; ours).
#include <omp.h>
#include "ref_draw.h"

namespace hlsl {

thread_local CbStream g_cb[4];

static float half_round(float f)            // fp32 -> fp16 (round to nearest even, overflow to inf) -> fp32
{
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = x & 0x80000000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return f;                                   // inf / nan
    if (x >= 0x477ff000u) { uint32_t r = sign | 0x7f800000u; float o; std::memcpy(&o, &r, 4); return o; }   // >= 65520 -> inf
    if (x < 0x33000001u) { float o; std::memcpy(&o, &sign, 4); return o; }                                  // < 2^-25 -> 0
    int e = (int)(x >> 23) - 127;
    int drop = e < -14 ? 13 + (-14 - e) : 13;                          // mantissa bits that do not fit
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    uint32_t keep = m >> drop, rem = m & ((1u << drop) - 1), half = 1u << (drop - 1);
    if (rem > half || (rem == half && (keep & 1))) keep++;
    double v = (double)keep * std::ldexp(1.0, e - 23 + drop);
    float o = (float)v;
    return sign ? -o : o;
}

float rt_round(float x, int fmt, int c)
{
    switch (fmt) {
    case 8:  return ::floorf(saturate(x) * 255.0f + 0.5f) / 255.0f;
    case 10: { const float mv = c == 3 ? 3.0f : 1023.0f; return ::floorf(saturate(x) * mv + 0.5f) / mv; }
    case 16: return half_round(x);
    default: return x;
    }
}

}  // namespace hlsl

extern "C" float ref_half_round(float x) { return hlsl::half_round(x); }
