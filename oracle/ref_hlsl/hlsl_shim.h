// hlsl_shim.h — a small HLSL-on-C++ execution model, so that the REFERENCE's own pixel-shader text
// (/root/reference/Shaders/**/*.hlsl and the text Source/Shaders.cpp generates at run time) compiles with g++ and
// runs on the CPU.  TEST INFRASTRUCTURE ONLY (see ../mpcvr_oracle.h): it pins the oracle's restatement of the
// per-pixel arithmetic to the reference text.  This file is ours; no reference code is in it.
//
// What is modelled (Direct3D 11 functional behaviour, not reference code):
//   * float2/3/4 with the swizzles, component-wise operators and scalar promotion HLSL has;
//   * float3x3 / float4x4 / float4x3 (row-major), mul(), transpose();
//   * the intrinsics the reference shaders call; pow(x,y) = exp2(y*log2(x)) as fxc lowers it (pow(0,y) = 0 for y > 0);
//   * Texture2D::Sample with point / linear filtering, clamp / wrap addressing, texel offsets; surfaces are RGBA fp32
//     arrays that already hold what the texture format can represent (UNORM decode / fp16 is the harness's business);
//   * cbuffers: consecutive 32-bit words in declaration order, popped by the member initialisers hlsl2cpp.py writes.
// No fused multiply-add is formed (build with -ffp-contract=off), unsuffixed literals are made float by hlsl2cpp.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../crmath.h"      // the transcendentals as defined functions (correctly rounded fp32 steps): the same definition as the oracle's

// ---- the draw call (C ABI; mirrored by ref_hlsl.py) ----------------------------------------------------------------------
extern "C" {
struct RefSurface { float* data; int32_t w, h; };
struct RefDraw {
    RefSurface tex[4];            // register(t0..t3)
    int32_t samp_filter[4];       // register(s0..s3): 0 point, 1 linear
    int32_t samp_address[4];      // 0 clamp, 1 wrap
    const uint32_t* cb[4];        // register(b0..b3): 32-bit words in declaration order
    int32_t cb_words[4];
    RefSurface rt;                // render target, RGBA fp32 holding what its format can represent
    int32_t rt_fmt;               // 0 fp32 (no rounding), 8 UNORM8 x4, 10 R10G10B10A2, 16 fp16 x4
    int32_t vp_x, vp_y, vp_w, vp_h;   // viewport = the pixels drawn
    float uv[3][2];               // texcoord at the viewport's top-left, top-right, bottom-left corner
    int32_t threads;              // 0 = OpenMP default
};
typedef void (*ref_draw_fn)(const RefDraw*);
}

namespace hlsl {

typedef unsigned int uint;

struct float2; struct float3; struct float4;

// ---- scalar intrinsics ---------------------------------------------------------------------------------------------
inline float h_pow(float x, float y) { return crm_powf(x, y); }      // exp2(y * log2(x)), each step rounded to fp32 (d3dcompiler's lowering)
inline float pow(float x, float y) { return h_pow(x, y); }
inline float exp(float x) { return crm_expf(x); }
inline float sin(float x) { return crm_sinf(x); }
inline float cos(float x) { return crm_cosf(x); }
inline float acos(float x) { return ::acosf(x); }
inline float sqrt(float x) { return ::sqrtf(x); }
inline float floor(float x) { return ::floorf(x); }
inline float ceil(float x) { return ::ceilf(x); }
inline float frac(float x) { return x - ::floorf(x); }
inline float fmod(float x, float y) { float q = x / y; q = q < 0 ? ::ceilf(q) : ::floorf(q); return x - y * q; }
inline float abs(float x) { return ::fabsf(x); }
inline float min(float a, float b) { return a < b ? a : b; }         // D3D min/max: NaN loses
inline float max(float a, float b) { return a >= b ? a : (b != b ? a : b); }
inline float saturate(float x) { return (x != x) ? 0.0f : (x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x)); }
inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }

// ---- swizzle proxies: views of the parent's storage (members of the parent's anonymous union) -------------------------
template <int P, int A, int B> struct sw2 {
    float d[P];
    operator float2() const;
    sw2& operator=(const float2& v);
    sw2& operator=(const sw2& o);
    template <int Q, int C, int D> sw2& operator=(const sw2<Q, C, D>& o);
    sw2& operator*=(const float2& v); sw2& operator*=(float s);
    sw2& operator+=(const float2& v); sw2& operator-=(const float2& v); sw2& operator/=(const float2& v);
};
template <int P, int A, int B, int C> struct sw3 {
    float d[P];
    operator float3() const;
    sw3& operator=(const float3& v);
    sw3& operator=(const sw3& o);
    template <int Q, int E, int F, int G> sw3& operator=(const sw3<Q, E, F, G>& o);
    sw3& operator*=(const float3& v); sw3& operator*=(float s);
    sw3& operator+=(const float3& v); sw3& operator-=(const float3& v); sw3& operator/=(const float3& v);
};
template <int P, int A, int B, int C, int D> struct sw4 {
    float d[P];
    operator float4() const;
    sw4& operator=(const float4& v);
    sw4& operator=(const sw4& o);
};

// member lists: every 2-, 3- and 4-letter combination over the parent's components, xyzw and rgba spellings
#define HLSL_L2a(F, ...) F(x, r, 0, __VA_ARGS__) F(y, g, 1, __VA_ARGS__)
#define HLSL_L2b(F, ...) F(x, r, 0, __VA_ARGS__) F(y, g, 1, __VA_ARGS__)
#define HLSL_L2c(F, ...) F(x, r, 0, __VA_ARGS__) F(y, g, 1, __VA_ARGS__)
#define HLSL_L2d(F, ...) F(x, r, 0, __VA_ARGS__) F(y, g, 1, __VA_ARGS__)
#define HLSL_L3a(F, ...) HLSL_L2a(F, __VA_ARGS__) F(z, b, 2, __VA_ARGS__)
#define HLSL_L3b(F, ...) HLSL_L2b(F, __VA_ARGS__) F(z, b, 2, __VA_ARGS__)
#define HLSL_L3c(F, ...) HLSL_L2c(F, __VA_ARGS__) F(z, b, 2, __VA_ARGS__)
#define HLSL_L3d(F, ...) HLSL_L2d(F, __VA_ARGS__) F(z, b, 2, __VA_ARGS__)
#define HLSL_L4a(F, ...) HLSL_L3a(F, __VA_ARGS__) F(w, a, 3, __VA_ARGS__)
#define HLSL_L4b(F, ...) HLSL_L3b(F, __VA_ARGS__) F(w, a, 3, __VA_ARGS__)
#define HLSL_L4c(F, ...) HLSL_L3c(F, __VA_ARGS__) F(w, a, 3, __VA_ARGS__)
#define HLSL_L4d(F, ...) HLSL_L3d(F, __VA_ARGS__) F(w, a, 3, __VA_ARGS__)

#define HLSL_M2(n2, m2, i2, P, n1, m1, i1) sw2<P, i1, i2> n1##n2, m1##m2;
#define HLSL_M3(n3, m3, i3, P, n1, m1, i1, n2, m2, i2) sw3<P, i1, i2, i3> n1##n2##n3, m1##m2##m3;
#define HLSL_M4(n4, m4, i4, P, n1, m1, i1, n2, m2, i2, n3, m3, i3) sw4<P, i1, i2, i3, i4> n1##n2##n3##n4, m1##m2##m3##m4;

// expand per parent size (macros cannot recurse, hence the a/b/c/d copies of the letter lists)
#define HLSL_E2(n1, m1, i1, P, LB) LB(HLSL_M2, P, n1, m1, i1)
#define HLSL_E3b(n2, m2, i2, P, LC, n1, m1, i1) LC(HLSL_M3, P, n1, m1, i1, n2, m2, i2)
#define HLSL_E4c(n3, m3, i3, P, LD, n1, m1, i1, n2, m2, i2) LD(HLSL_M4, P, n1, m1, i1, n2, m2, i2, n3, m3, i3)

#define HLSL_E3_2(n1, m1, i1, P) HLSL_L2b(HLSL_E3b, P, HLSL_L2c, n1, m1, i1)
#define HLSL_E3_3(n1, m1, i1, P) HLSL_L3b(HLSL_E3b, P, HLSL_L3c, n1, m1, i1)
#define HLSL_E3_4(n1, m1, i1, P) HLSL_L4b(HLSL_E3b, P, HLSL_L4c, n1, m1, i1)
#define HLSL_E4b_2(n2, m2, i2, P, n1, m1, i1) HLSL_L2c(HLSL_E4c, P, HLSL_L2d, n1, m1, i1, n2, m2, i2)
#define HLSL_E4b_3(n2, m2, i2, P, n1, m1, i1) HLSL_L3c(HLSL_E4c, P, HLSL_L3d, n1, m1, i1, n2, m2, i2)
#define HLSL_E4b_4(n2, m2, i2, P, n1, m1, i1) HLSL_L4c(HLSL_E4c, P, HLSL_L4d, n1, m1, i1, n2, m2, i2)
#define HLSL_E4_2(n1, m1, i1, P) HLSL_L2b(HLSL_E4b_2, P, n1, m1, i1)
#define HLSL_E4_3(n1, m1, i1, P) HLSL_L3b(HLSL_E4b_3, P, n1, m1, i1)
#define HLSL_E4_4(n1, m1, i1, P) HLSL_L4b(HLSL_E4b_4, P, n1, m1, i1)

#define HLSL_SWIZZLES_2 HLSL_L2a(HLSL_E2, 2, HLSL_L2b) HLSL_L2a(HLSL_E3_2, 2) HLSL_L2a(HLSL_E4_2, 2)
#define HLSL_SWIZZLES_3 HLSL_L3a(HLSL_E2, 3, HLSL_L3b) HLSL_L3a(HLSL_E3_3, 3) HLSL_L3a(HLSL_E4_3, 3)
#define HLSL_SWIZZLES_4 HLSL_L4a(HLSL_E2, 4, HLSL_L4b) HLSL_L4a(HLSL_E3_4, 4) HLSL_L4a(HLSL_E4_4, 4)

struct float2 {
    union {
        float v[2];
        struct { float x, y; };
        struct { float r, g; };
        HLSL_SWIZZLES_2
    };
    float2() : v{0, 0} {}
    float2(float s) : v{s, s} {}
    float2(float a, float b) : v{a, b} {}
    float2(const float2& o) : v{o.v[0], o.v[1]} {}
    float2& operator=(const float2& o) { v[0] = o.v[0]; v[1] = o.v[1]; return *this; }
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
struct float3 {
    union {
        float v[3];
        struct { float x, y, z; };
        struct { float r, g, b; };
        HLSL_SWIZZLES_3
    };
    float3() : v{0, 0, 0} {}
    float3(float s) : v{s, s, s} {}
    float3(float a, float b, float c) : v{a, b, c} {}
    float3(const float2& a, float c) : v{a.v[0], a.v[1], c} {}
    float3(float a, const float2& b) : v{a, b.v[0], b.v[1]} {}
    float3(const float3& o) : v{o.v[0], o.v[1], o.v[2]} {}
    float3& operator=(const float3& o) { v[0] = o.v[0]; v[1] = o.v[1]; v[2] = o.v[2]; return *this; }
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
struct float4 {
    union {
        float v[4];
        struct { float x, y, z, w; };
        struct { float r, g, b, a; };
        HLSL_SWIZZLES_4
    };
    float4() : v{0, 0, 0, 0} {}
    float4(float s) : v{s, s, s, s} {}
    float4(float a, float b, float c, float d) : v{a, b, c, d} {}
    float4(const float3& a, float d) : v{a.v[0], a.v[1], a.v[2], d} {}
    float4(float a, const float3& b) : v{a, b.v[0], b.v[1], b.v[2]} {}
    float4(float a, const float2& b, float d) : v{a, b.v[0], b.v[1], d} {}
    float4(const float2& a, const float2& b) : v{a.v[0], a.v[1], b.v[0], b.v[1]} {}
    float4(const float2& a, float c, float d) : v{a.v[0], a.v[1], c, d} {}
    float4(const float4& o) : v{o.v[0], o.v[1], o.v[2], o.v[3]} {}
    float4& operator=(const float4& o) { for (int i = 0; i < 4; i++) v[i] = o.v[i]; return *this; }
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
struct int2 { int x, y; int2(int a, int b) : x(a), y(b) {} };

// proxy bodies
template <int P, int A, int B> sw2<P, A, B>::operator float2() const { return float2(d[A], d[B]); }
template <int P, int A, int B> sw2<P, A, B>& sw2<P, A, B>::operator=(const float2& v) { d[A] = v.v[0]; d[B] = v.v[1]; return *this; }
template <int P, int A, int B> sw2<P, A, B>& sw2<P, A, B>::operator=(const sw2& o) { float2 t = o; return *this = t; }
template <int P, int A, int B> template <int Q, int C, int D> sw2<P, A, B>& sw2<P, A, B>::operator=(const sw2<Q, C, D>& o) { float2 t = o; return *this = t; }
template <int P, int A, int B, int C> sw3<P, A, B, C>::operator float3() const { return float3(d[A], d[B], d[C]); }
template <int P, int A, int B, int C> sw3<P, A, B, C>& sw3<P, A, B, C>::operator=(const float3& v) { d[A] = v.v[0]; d[B] = v.v[1]; d[C] = v.v[2]; return *this; }
template <int P, int A, int B, int C> sw3<P, A, B, C>& sw3<P, A, B, C>::operator=(const sw3& o) { float3 t = o; return *this = t; }
template <int P, int A, int B, int C> template <int Q, int E, int F, int G> sw3<P, A, B, C>& sw3<P, A, B, C>::operator=(const sw3<Q, E, F, G>& o) { float3 t = o; return *this = t; }
template <int P, int A, int B, int C, int D> sw4<P, A, B, C, D>::operator float4() const { return float4(d[A], d[B], d[C], d[D]); }
template <int P, int A, int B, int C, int D> sw4<P, A, B, C, D>& sw4<P, A, B, C, D>::operator=(const float4& v) { d[A] = v.v[0]; d[B] = v.v[1]; d[C] = v.v[2]; d[D] = v.v[3]; return *this; }
template <int P, int A, int B, int C, int D> sw4<P, A, B, C, D>& sw4<P, A, B, C, D>::operator=(const sw4& o) { float4 t = o; return *this = t; }

// ---- component-wise operators ------------------------------------------------------------------------------------------
#define HLSL_BINOP(T, N, op) \
    inline T operator op(const T& a, const T& b) { T r; for (int i = 0; i < N; i++) r.v[i] = a.v[i] op b.v[i]; return r; } \
    inline T operator op(const T& a, float b) { T r; for (int i = 0; i < N; i++) r.v[i] = a.v[i] op b; return r; } \
    inline T operator op(float a, const T& b) { T r; for (int i = 0; i < N; i++) r.v[i] = a op b.v[i]; return r; } \
    inline T& operator op##=(T& a, const T& b) { for (int i = 0; i < N; i++) a.v[i] = a.v[i] op b.v[i]; return a; } \
    inline T& operator op##=(T& a, float b) { for (int i = 0; i < N; i++) a.v[i] = a.v[i] op b; return a; }
#define HLSL_VEC(T, N) \
    HLSL_BINOP(T, N, +) HLSL_BINOP(T, N, -) HLSL_BINOP(T, N, *) HLSL_BINOP(T, N, /) \
    inline T operator-(const T& a) { T r; for (int i = 0; i < N; i++) r.v[i] = -a.v[i]; return r; }
HLSL_VEC(float2, 2) HLSL_VEC(float3, 3) HLSL_VEC(float4, 4)

template <int P, int A, int B> sw2<P, A, B>& sw2<P, A, B>::operator*=(const float2& v) { float2 t = *this; return *this = t * v; }
template <int P, int A, int B> sw2<P, A, B>& sw2<P, A, B>::operator*=(float s) { float2 t = *this; return *this = t * s; }
template <int P, int A, int B> sw2<P, A, B>& sw2<P, A, B>::operator+=(const float2& v) { float2 t = *this; return *this = t + v; }
template <int P, int A, int B> sw2<P, A, B>& sw2<P, A, B>::operator-=(const float2& v) { float2 t = *this; return *this = t - v; }
template <int P, int A, int B> sw2<P, A, B>& sw2<P, A, B>::operator/=(const float2& v) { float2 t = *this; return *this = t / v; }
template <int P, int A, int B, int C> sw3<P, A, B, C>& sw3<P, A, B, C>::operator*=(const float3& v) { float3 t = *this; return *this = t * v; }
template <int P, int A, int B, int C> sw3<P, A, B, C>& sw3<P, A, B, C>::operator*=(float s) { float3 t = *this; return *this = t * s; }
template <int P, int A, int B, int C> sw3<P, A, B, C>& sw3<P, A, B, C>::operator+=(const float3& v) { float3 t = *this; return *this = t + v; }
template <int P, int A, int B, int C> sw3<P, A, B, C>& sw3<P, A, B, C>::operator-=(const float3& v) { float3 t = *this; return *this = t - v; }
template <int P, int A, int B, int C> sw3<P, A, B, C>& sw3<P, A, B, C>::operator/=(const float3& v) { float3 t = *this; return *this = t / v; }

// comparisons give bool vectors; select() is the component-wise ?: (hlsl2cpp.py rewrites the two vector ternaries)
struct bool3 { bool v[3]; };
struct bool4 { bool v[4]; };
inline bool3 operator<=(const float3& a, float b) { bool3 r; for (int i = 0; i < 3; i++) r.v[i] = a.v[i] <= b; return r; }
inline bool4 operator==(const float4& a, const float4& b) { bool4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] == b.v[i]; return r; }
inline float3 select(const bool3& c, const float3& a, const float3& b) { float3 r; for (int i = 0; i < 3; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }
inline float4 select(const bool4& c, const float4& a, const float4& b) { float4 r; for (int i = 0; i < 4; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }

// ---- vector intrinsics -------------------------------------------------------------------------------------------------
#define HLSL_MAP1(T, N, f) inline T f(const T& a) { T r; for (int i = 0; i < N; i++) r.v[i] = f(a.v[i]); return r; }
#define HLSL_MAP2(T, N, f) \
    inline T f(const T& a, const T& b) { T r; for (int i = 0; i < N; i++) r.v[i] = f(a.v[i], b.v[i]); return r; } \
    inline T f(const T& a, float b) { T r; for (int i = 0; i < N; i++) r.v[i] = f(a.v[i], b); return r; } \
    inline T f(float a, const T& b) { T r; for (int i = 0; i < N; i++) r.v[i] = f(a, b.v[i]); return r; }
#define HLSL_INTR(T, N) \
    HLSL_MAP1(T, N, sin) HLSL_MAP1(T, N, cos) HLSL_MAP1(T, N, exp) HLSL_MAP1(T, N, floor) HLSL_MAP1(T, N, frac) \
    HLSL_MAP1(T, N, saturate) HLSL_MAP1(T, N, sqrt) HLSL_MAP1(T, N, abs) \
    HLSL_MAP2(T, N, pow) HLSL_MAP2(T, N, min) HLSL_MAP2(T, N, max) \
    inline T clamp(const T& x, const T& lo, const T& hi) { return min(max(x, lo), hi); } \
    inline T clamp(const T& x, float lo, float hi) { return min(max(x, lo), hi); } \
    inline T lerp(const T& a, const T& b, float t) { return a + t * (b - a); } \
    inline T lerp(const T& a, const T& b, const T& t) { return a + t * (b - a); }
HLSL_INTR(float2, 2) HLSL_INTR(float3, 3) HLSL_INTR(float4, 4)

inline float dot(const float2& a, const float2& b) { return a.x * b.x + a.y * b.y; }
inline float dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dot(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline float dot(float a, const float3& b) { return dot(float3(a), b); }
inline float dot(const float4& a, float b) { return dot(a, float4(b)); }

// ---- matrices (row-major, m[r][c]) -------------------------------------------------------------------------------------
struct float3x3 {
    float3 r[3];
    float3x3() {}
    float3x3(float a, float b, float c, float d, float e, float f, float g, float h, float i) { r[0] = float3(a, b, c); r[1] = float3(d, e, f); r[2] = float3(g, h, i); }
    float3x3(const float3& a, const float3& b, const float3& c) { r[0] = a; r[1] = b; r[2] = c; }
    float3& operator[](int i) { return r[i]; }
    const float3& operator[](int i) const { return r[i]; }
};
struct float4x4 {
    float4 r[4];
    float4x4() {}
    float4x4(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3,
             float c0, float c1, float c2, float c3, float d0, float d1, float d2, float d3)
    { r[0] = float4(a0, a1, a2, a3); r[1] = float4(b0, b1, b2, b3); r[2] = float4(c0, c1, c2, c3); r[3] = float4(d0, d1, d2, d3); }
    float4x4(const float4& a, const float4& b, const float4& c, const float4& d) { r[0] = a; r[1] = b; r[2] = c; r[3] = d; }
    float4& operator[](int i) { return r[i]; }
    const float4& operator[](int i) const { return r[i]; }
};
struct float4x3 {           // 4 rows of float3
    float3 r[4];
    float4x3(const float3& a, const float3& b, const float3& c, const float3& d) { r[0] = a; r[1] = b; r[2] = c; r[3] = d; }
};
inline float mul(const float3& a, const float3& b) { return dot(a, b); }                       // vector x vector = dot
inline float3 mul(const float3x3& m, const float3& v) { return float3(dot(m.r[0], v), dot(m.r[1], v), dot(m.r[2], v)); }
inline float4 mul(const float4x4& m, const float4& v) { return float4(dot(m.r[0], v), dot(m.r[1], v), dot(m.r[2], v), dot(m.r[3], v)); }
inline float3 mul(const float4& v, const float4x3& m)                                          // row vector x matrix
{
    float3 o;
    for (int c = 0; c < 3; c++) o.v[c] = v.x * m.r[0].v[c] + v.y * m.r[1].v[c] + v.z * m.r[2].v[c] + v.w * m.r[3].v[c];
    return o;
}
inline float3x3 mul(const float3x3& a, const float3x3& b)
{
    float3x3 o;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.r[i].v[j] = a.r[i].v[0] * b.r[0].v[j] + a.r[i].v[1] * b.r[1].v[j] + a.r[i].v[2] * b.r[2].v[j];
    return o;
}
inline float4x4 mul(const float4x4& a, const float4x4& b)
{
    float4x4 o;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
        o.r[i].v[j] = a.r[i].v[0] * b.r[0].v[j] + a.r[i].v[1] * b.r[1].v[j] + a.r[i].v[2] * b.r[2].v[j] + a.r[i].v[3] * b.r[3].v[j];
    return o;
}
inline float3x3 transpose(const float3x3& m)
{
    float3x3 o;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.r[i].v[j] = m.r[j].v[i];
    return o;
}

// ---- resources ---------------------------------------------------------------------------------------------------------
typedef RefSurface Surface;                                // RGBA fp32, 4 floats per texel, rows back to back

struct SamplerState { int filter = 0; int address = 0; };  // filter: 0 point, 1 linear; address: 0 clamp, 1 wrap

// what Sample() returns: a float4 that may also initialise a float (HLSL truncates the vector; Shaders.cpp:233-234)
struct float4s : float4 {
    float4s(const float4& o) : float4(o) {}
    operator float() const { return v[0]; }
};

struct Texture2D {
    const Surface* s = nullptr;
    float4 texel(int x, int y, int address) const
    {
        if (address == 1) { x %= s->w; if (x < 0) x += s->w; y %= s->h; if (y < 0) y += s->h; }
        else { x = x < 0 ? 0 : (x >= s->w ? s->w - 1 : x); y = y < 0 ? 0 : (y >= s->h ? s->h - 1 : y); }
        const float* p = s->data + ((size_t)y * s->w + x) * 4;
        return float4(p[0], p[1], p[2], p[3]);
    }
    float4s Sample(const SamplerState& sm, const float2& uv) const { return Sample(sm, uv, int2(0, 0)); }
    float4s Sample(const SamplerState& sm, const float2& uv, const int2& off) const
    {
        const float u = uv.x * (float)s->w, v = uv.y * (float)s->h;      // scaled coordinates
        if (sm.filter == 0)
            return float4s(texel((int)::floorf(u) + off.x, (int)::floorf(v) + off.y, sm.address));
        // linear: taps floor(u - .5), +1; the fixed-function filter keeps 8 fractional bits of the weights (D3D11 spec 7.18.8)
        const float fu = u - 0.5f, fv = v - 0.5f;
        const float iu = ::floorf(fu), iv = ::floorf(fv);
        const float wx = ::floorf((fu - iu) * 256.0f + 0.5f) / 256.0f, wy = ::floorf((fv - iv) * 256.0f + 0.5f) / 256.0f;
        const int x0 = (int)iu + off.x, y0 = (int)iv + off.y;
        const float4 c00 = texel(x0, y0, sm.address), c10 = texel(x0 + 1, y0, sm.address);
        const float4 c01 = texel(x0, y0 + 1, sm.address), c11 = texel(x0 + 1, y0 + 1, sm.address);
        const float4 top = c00 * (1.0f - wx) + c10 * wx, bot = c01 * (1.0f - wx) + c11 * wx;
        return float4s(top * (1.0f - wy) + bot * wy);
    }
};

// fixed-size array member of a cbuffer
template <class T, int N> struct arr {
    T e[N];
    T& operator[](int i) { return e[i]; }
    const T& operator[](int i) const { return e[i]; }
};

// cbuffer word streams, one per register(bN); the draw loop points them at the caller's words before it constructs a shader
struct CbStream { const uint32_t* p = nullptr; int n = 0, pos = 0; };
extern thread_local CbStream g_cb[4];
template <class T> inline T cb_pop(int slot)
{
    T t;
    CbStream& s = g_cb[slot];
    const int words = (int)(sizeof(T) / 4);
    if (s.p && s.pos + words <= s.n) std::memcpy((void*)&t, s.p + s.pos, sizeof(T));
    else std::memset((void*)&t, 0, sizeof(T));
    s.pos += words;
    return t;
}

}  // namespace hlsl

