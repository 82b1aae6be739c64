/*
 * mpcvr_oracle.c — plain-C CPU restatement of the MPC Video Renderer shader video processor
 * (convert -> separable resize -> final pass/dither).  See mpcvr_oracle.h for the status header:
 * TEST INFRASTRUCTURE ONLY; parameter maths pinned against the real csputils.cpp (oracle/_ref),
 * HLSL arithmetic "parity unpinned" (no reference tests / no D3D here).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off [-fopenmp] (see oracle/Makefile).  fp-contract is off
 * so every a*b+c below is two IEEE roundings — the restatement is then independent of the host ISA.
 *
 * D3D11 semantics modelled explicitly (functional spec):
 *   UNORM load  v/(2^n-1);  point sample floor(u*W) clamped;  linear sample with exact 1/4-step
 *   weights;  float->UNORM store floor(sat(x)*(2^n-1)+0.5);  float->fp16 round-to-nearest-even;
 *   saturate;  frac;  pow(x,y)=exp2(y*log2(x)).
 */
#include "mpcvr_oracle.h"
#include "crmath.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

static inline int   clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float saturatef(float x) { return (x != x) ? 0.0f : (x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x)); }
/* HLSL pow: exp2(y*log2(x)) (d3dcompiler lowers pow to log/mul/exp). */
/* Sensitivity probe (tests only, orc_set_pow_ulp_bias): every pow() of the chain answers `bias` units in the last place off.
 * Direct3D's pow is exp2(y * log2(x)) with approximate log2 / exp2 — with 1-ulp building blocks the result is off by up to
 * ~0.35 |y log2 x| + 1.5 ulp (4 ulp at x = 1e-4, y = 1/2.2), so an output that moves by several codes under +-4 ulp is one the
 * reference itself does not define: the tests use this to tell ill-conditioned channels from wrong ones. */
static int g_pow_ulp_bias = 0;
static uint32_t g_pow_ulp_seed = 0;
void orc_set_pow_ulp_bias(int bias) { g_pow_ulp_bias = bias; g_pow_ulp_seed = 0; }
/* seed != 0: every call errs by its own amount in [-amplitude, +amplitude], a hash of (x, y, seed) — independent errors per channel
 * and per pow of the chain, as a real approximate pow has them (a uniform bias cancels in the gamut matrix, whose rows sum to 1) */
void orc_set_pow_ulp_noise(int amplitude, uint32_t seed) { g_pow_ulp_bias = amplitude; g_pow_ulp_seed = seed; }
/* Sensitivity probe (tests only): the texture the HDR10 tone-mapping step reads arrives `bias` codes of its UNORM format off — on channel
 * `channel` (0..2; -1: all three), or (seed != 0) every channel of every texel by its own hash-drawn amount in [-|bias|, +|bias|].  The fused
 * tiers of the product are held to one code at every stored intermediate; a local operator maps that code with its own slope (operator 6
 * near black: seven ten-bit codes per code), so a channel beyond the bar behind an operator is shown case by case to lie inside what the
 * oracle answers for inputs one code either side (tests/test_parity_gpu.py compare_behind_tail, operator_input). */
static int g_tm_in_bias = 0, g_tm_in_channel = -1;
static uint32_t g_tm_in_seed = 0;
void orc_set_tonemap_input_bias(int bias, int channel, uint32_t seed) { g_tm_in_bias = bias; g_tm_in_channel = channel; g_tm_in_seed = seed; }
static inline float tm_input_probe(float v, float maxv, int x, int y, int ch)
{
    if (!g_tm_in_bias || maxv <= 0.0f || (g_tm_in_channel >= 0 && g_tm_in_channel != ch && !g_tm_in_seed)) return v;
    int bias = g_tm_in_bias;
    if (g_tm_in_seed) {
        uint32_t h = ((uint32_t)x * 0x9E3779B9u) ^ ((uint32_t)y * 0x85EBCA6Bu) ^ ((uint32_t)ch * 0xC2B2AE35u) ^ g_tm_in_seed;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        const int a = bias < 0 ? -bias : bias;
        bias = (int)(h % (uint32_t)(2 * a + 1)) - a;
    }
    float code = floorf(v * maxv + 0.5f) + (float)bias;
    if (code < 0.0f) code = 0.0f;
    if (code > maxv) code = maxv;
    return code / maxv;
}
/* ... and the same one step earlier, where a GPU's pow() actually errs: log2(x) answers up to `amplitude` ulps off (a hash of the operands
 * and `seed`; seed 0 = every call by +amplitude, or by -amplitude when negative), THEN the product with y and exp2.  Direct3D 11 grants
 * log2 / exp2 a relative error of 2^-21 (four fp32 ulps), v_log_f32 is specified to one; behind pow(x, 78.8) or pow(x, 6.28) — the PQ
 * chains — one ulp of log2 is 35-80 ulps of the result, which the +-4 ulp-of-the-result probe above does not span (soak case 1428). */
static int g_log2_ulp_amp = 0;
static uint32_t g_log2_ulp_seed = 0;
void orc_set_pow_log2_noise(int amplitude, uint32_t seed) { g_log2_ulp_amp = amplitude; g_log2_ulp_seed = seed; }
static inline float pow_with_log2_noise(float x, float y)
{
    float l = crm_log2f(x);
    if (l == l && l != 0.0f && l > -3.0e38f && l < 3.0e38f) {
        int bias = g_log2_ulp_amp;
        if (g_log2_ulp_seed) {
            uint32_t a, b2; memcpy(&a, &x, 4); memcpy(&b2, &y, 4);
            uint32_t h = (a ^ (b2 * 0x9E3779B9u) ^ g_log2_ulp_seed) * 0x85EBCA6Bu;
            h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
            const int amp = bias < 0 ? -bias : bias;
            bias = (int)(h % (uint32_t)(2 * amp + 1)) - amp;
        }
        uint32_t u; memcpy(&u, &l, 4);
        u = (uint32_t)((int32_t)u + ((l < 0.0f) ? -bias : bias));      /* (ulps towards +inf for a positive bias, whatever the sign of l) */
        memcpy(&l, &u, 4);
    }
    return crm_exp2f(y * l);
}
/* Sensitivity probe (tests only): every texel of m_TexConvertOutput is stored `bias` codes of its UNORM format off — on `channel` (0..2; -1: all
 * three), or (seed != 0) every channel of every texel by its own hash-drawn amount in [-|bias|, +|bias|].  The fused tiers' convert stage is held
 * to one code of that texture; what one code becomes behind the draws is the reference's own business — its Bicubic / Lanczos DOWNSCALE shaders
 * divide by a weight sum that is small at some phases (soak case 5624: one code of one luma sample moves ONE output pixel of the oracle by 17
 * ten-bit codes), so a channel beyond the bar behind such a draw is shown to lie inside what the oracle answers for that texture one code off. */
static int g_cv_out_bias = 0, g_cv_out_channel = -1;
static uint32_t g_cv_out_seed = 0;
void orc_set_convert_output_bias(int bias, int channel, uint32_t seed) { g_cv_out_bias = bias; g_cv_out_channel = channel; g_cv_out_seed = seed; }
static inline float hlsl_pow(float x, float y)
{
    float r = g_log2_ulp_amp ? pow_with_log2_noise(x, y) : crm_powf(x, y);          /* exp2(y * log2 x), each step the correctly rounded fp32 function (crmath.h) */
    if (g_pow_ulp_bias && r > 0.0f && r < 3.0e38f) {
        uint32_t u; memcpy(&u, &r, 4);
        int bias = g_pow_ulp_bias;
        if (g_pow_ulp_seed) {
            uint32_t a, b2; memcpy(&a, &x, 4); memcpy(&b2, &y, 4);
            uint32_t h = (a ^ (b2 * 0x9E3779B9u) ^ g_pow_ulp_seed) * 0x85EBCA6Bu;
            h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
            bias = (int)(h % (uint32_t)(2 * g_pow_ulp_bias + 1)) - g_pow_ulp_bias;
        }
        u = (uint32_t)((int32_t)u + bias);
        memcpy(&r, &u, 4);
    }
    return r;
}

/* the defined transcendentals over an array (tests: the product's device evaluation must equal this bit for bit) */
void orc_eval_transcendental(int fn, const float *x, const float *y, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++)
        out[i] = fn == 0 ? crm_log2f(x[i]) : fn == 1 ? crm_exp2f(x[i]) : fn == 2 ? crm_expf(x[i]) : fn == 3 ? crm_powf(x[i], y[i]) : fn == 4 ? crm_sinf(x[i]) : crm_cosf(x[i]);
}

static int g_threads = 0;
int orc_num_threads(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) { g_threads = n; }
#ifdef _OPENMP
#define ORC_PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(orc_num_threads())")
#else
#define ORC_PAR_FOR
#endif

/* fp32 -> fp16 round-to-nearest-even -> fp32 (D3D11 float32->float16 store conversion). */
static uint16_t float_to_half_bits(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007fffffu;
    int32_t  exp  = (int32_t)((x >> 23) & 0xff);
    if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0));   /* inf / nan */
    int32_t e = exp - 127 + 15;
    if (e >= 0x1f) return (uint16_t)(sign | 0x7c00u);                            /* overflow -> inf */
    if (e <= 0) {                                                                /* subnormal / zero */
        if (e < -10) return (uint16_t)sign;
        mant |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t h = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;                      /* may carry into exp: ok */
    return (uint16_t)(sign | h);
}
float orc_half_bits_to_float(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, mant = h & 0x3ffu, x;
    if (exp == 0) {
        if (!mant) x = sign;
        else { int e = -1; do { mant <<= 1; e++; } while (!(mant & 0x400u));
               x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3ffu) << 13); }
    } else if (exp == 0x1f) x = sign | 0x7f800000u | (mant << 13);
    else x = sign | ((exp - 15 + 127) << 23) | (mant << 13);
    float f; memcpy(&f, &x, 4); return f;
}
float orc_half_round(float x) { return orc_half_bits_to_float(float_to_half_bits(x)); }

/* ------------------------------------------------------------------------------------------ */
/* formats — Helper.cpp:295-359 (s_FmtConvMapping, DX11PlaneConfig_t)                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int cformat;
    int planes;        /* 2 = Y + interleaved UV, 3 = Y,U,V */
    int bytes;         /* bytes per sample: 1 (R8) or 2 (R16) */
    int div_w, div_h;  /* chroma divisors (DX11PlaneConfig_t) */
    int packsize;      /* Packsize */
    int pitch_coeff;   /* PitchCoeff: lines = H*coeff/2 */
    int subsampling;   /* 420/422/444 */
    int cdepth;        /* CDepth */
    int shift;         /* CopyPlane10to16 (<<6) on upload — Helper.cpp:386-391,789-803 */
    int v_first;       /* YV12/YV16/YV24: second plane is V — Shaders.cpp:159-165 */
    int layout;        /* LAY_*: how the single texture of a one-plane format is organised */
    int cstype;        /* CST_*: ColorSystem_t (Helper.h:129-133) */
    int ci[4];         /* packed 4:2:2: texel components holding Y0,U,Y1,V (Shaders.cpp:195-229);
                          packed 4:4:4: components holding Y,U,V after the .zyxw/.yxzw swizzle (:186-193) */
    int bits10;        /* texel is one R10G10B10A2 dword (Y410, r210) */
    int repack;        /* RPK_*: GetCopyPlaneFunction (Helper.cpp:377-412) for the interleaved RGB formats */
} fmt_info;
enum { LAY_PLANAR = 0, LAY_PACKED422 = 1, LAY_PACKED444 = 2, LAY_GRAY = 3, LAY_RGB = 4 };
enum { RPK_NONE = 0, RPK_RGB24, RPK_R210, RPK_RGB48, RPK_BGR48, RPK_BGRA64, RPK_B64A };
enum { CST_YUV = 0, CST_RGB = 1, CST_GRAY = 2 };

static const fmt_info s_fmts[] = {
    {ORC_CF_NV12,      2, 1, 2, 2, 1, 3, 420,  8, 0, 0},
    {ORC_CF_P010,      2, 2, 2, 2, 2, 3, 420, 16, 0, 0},
    {ORC_CF_P016,      2, 2, 2, 2, 2, 3, 420, 16, 0, 0},
    {ORC_CF_P210,      2, 2, 2, 1, 2, 4, 422, 16, 0, 0},
    {ORC_CF_P216,      2, 2, 2, 1, 2, 4, 422, 16, 0, 0},
    {ORC_CF_YV12,      3, 1, 2, 2, 1, 3, 420,  8, 0, 1},
    {ORC_CF_YV16,      3, 1, 2, 1, 1, 4, 422,  8, 0, 1},
    {ORC_CF_YV24,      3, 1, 1, 1, 1, 6, 444,  8, 0, 1},
    {ORC_CF_YUV420P8,  3, 1, 2, 2, 1, 3, 420,  8, 0, 0},
    {ORC_CF_YUV422P8,  3, 1, 2, 1, 1, 4, 422,  8, 0, 0},
    {ORC_CF_YUV444P8,  3, 1, 1, 1, 1, 6, 444,  8, 0, 0},
    {ORC_CF_YUV420P10, 3, 2, 2, 2, 2, 3, 420, 10, 6, 0},
    {ORC_CF_YUV420P16, 3, 2, 2, 2, 2, 3, 420, 16, 0, 0},
    {ORC_CF_YUV422P10, 3, 2, 2, 1, 2, 4, 422, 10, 6, 0},
    {ORC_CF_YUV422P16, 3, 2, 2, 1, 2, 4, 422, 16, 0, 0},
    {ORC_CF_YUV444P10, 3, 2, 1, 1, 2, 6, 444, 10, 6, 0},
    {ORC_CF_YUV444P16, 3, 2, 1, 1, 2, 6, 444, 16, 0, 0},
    /* one RGBA8 / RGBA16 texel = two pixels (DX11Plane_RGBA8 / DX11Plane_RGBA16, Helper.cpp:305-307) */
    {ORC_CF_YUY2,      1, 1, 2, 1, 2, 2, 422,  8, 0, 0, LAY_PACKED422, CST_YUV, {0, 1, 2, 3}},
    {ORC_CF_UYVY,      1, 1, 2, 1, 2, 2, 422,  8, 0, 0, LAY_PACKED422, CST_YUV, {1, 0, 3, 2}},
    {ORC_CF_Y210,      1, 2, 2, 1, 4, 2, 422, 10, 0, 0, LAY_PACKED422, CST_YUV, {0, 1, 2, 3}},
    {ORC_CF_Y216,      1, 2, 2, 1, 4, 2, 422, 16, 0, 0, LAY_PACKED422, CST_YUV, {0, 1, 2, 3}},
    {ORC_CF_V210,      1, 2, 2, 1, 0, 2, 422, 10, 0, 0, LAY_PACKED422, CST_YUV, {0, 1, 2, 3}},   /* Y210 after CopyFrameV210 */
    /* one texel = one pixel; memory order AYUV: V,U,Y,A  Y410: U:10,Y:10,V:10,A:2  Y416: U,Y,V,A */
    {ORC_CF_AYUV,      1, 1, 1, 1, 4, 2, 444,  8, 0, 0, LAY_PACKED444, CST_YUV, {2, 1, 0, 3}},
    {ORC_CF_Y410,      1, 4, 1, 1, 4, 2, 444, 10, 0, 0, LAY_PACKED444, CST_YUV, {1, 0, 2, 3}, 1},
    {ORC_CF_Y416,      1, 2, 1, 1, 8, 2, 444, 16, 0, 0, LAY_PACKED444, CST_YUV, {1, 0, 2, 3}},
    /* planar RGB: planes G,B,R sampled as texY,texU,texV; matrix columns rotated (DX11VideoProcessor.cpp:863-867) */
    {ORC_CF_GBRP8,     3, 1, 1, 1, 1, 6, 444,  8, 0, 0, LAY_PLANAR, CST_RGB},
    {ORC_CF_GBRP10,    3, 2, 1, 1, 2, 6, 444, 10, 6, 0, LAY_PLANAR, CST_RGB},
    {ORC_CF_GBRP16,    3, 2, 1, 1, 2, 6, 444, 16, 0, 0, LAY_PLANAR, CST_RGB},
    /* gray: R8 / R16 texture, Sample() returns (Y,0,0,1) */
    {ORC_CF_Y8,        1, 1, 1, 1, 1, 2, 400,  8, 0, 0, LAY_GRAY, CST_GRAY},
    {ORC_CF_Y10,       1, 2, 1, 1, 2, 2, 400, 10, 6, 0, LAY_GRAY, CST_GRAY},
    {ORC_CF_Y16,       1, 2, 1, 1, 2, 2, 400, 16, 0, 0, LAY_GRAY, CST_GRAY},
    /* interleaved RGB (Helper.cpp:345-354): texture B8G8R8X8 / R10G10B10A2 / R16G16B16A16; ci = texel components of R,G,B */
    {ORC_CF_RGB24,     1, 1, 1, 1, 3, 2, 444,  8, 0, 0, LAY_RGB, CST_RGB, {2, 1, 0, 3}, 0, RPK_RGB24},
    {ORC_CF_XRGB32,    1, 1, 1, 1, 4, 2, 444,  8, 0, 0, LAY_RGB, CST_RGB, {2, 1, 0, 3}, 0, RPK_NONE},
    {ORC_CF_ARGB32,    1, 1, 1, 1, 4, 2, 444,  8, 0, 0, LAY_RGB, CST_RGB, {2, 1, 0, 3}, 0, RPK_NONE},
    {ORC_CF_r210,      1, 4, 1, 1, 4, 2, 444, 10, 0, 0, LAY_RGB, CST_RGB, {0, 1, 2, 3}, 1, RPK_R210},
    {ORC_CF_RGB48,     1, 2, 1, 1, 6, 2, 444, 16, 0, 0, LAY_RGB, CST_RGB, {0, 1, 2, 3}, 0, RPK_RGB48},
    {ORC_CF_BGR48,     1, 2, 1, 1, 6, 2, 444, 16, 0, 0, LAY_RGB, CST_RGB, {0, 1, 2, 3}, 0, RPK_BGR48},
    {ORC_CF_BGRA64,    1, 2, 1, 1, 8, 2, 444, 16, 0, 0, LAY_RGB, CST_RGB, {0, 1, 2, 3}, 0, RPK_BGRA64},
    {ORC_CF_B64A,      1, 2, 1, 1, 8, 2, 444, 16, 0, 0, LAY_RGB, CST_RGB, {0, 1, 2, 3}, 0, RPK_B64A},
};
static const fmt_info *find_fmt(int cf)
{
    for (size_t i = 0; i < sizeof(s_fmts) / sizeof(s_fmts[0]); i++)
        if (s_fmts[i].cformat == cf) return &s_fmts[i];
    return NULL;
}

/* pitch / lines rules — DX11VideoProcessor.cpp:1789-1803 */
size_t orc_frame_bytes(int cformat, int width, int height, int *pitch_out)
{
    const fmt_info *f = find_fmt(cformat);
    if (!f) return 0;
    int pitch = width * f->packsize;
    if (cformat == ORC_CF_NV12 || cformat == ORC_CF_Y8 || cformat == ORC_CF_RGB24 || cformat == ORC_CF_BGR48)
        pitch = (pitch + 3) & ~3;                                                     /* ALIGN(m_srcPitch, 4) :1792-1796 */
    if (cformat == ORC_CF_V210) pitch = (((width + 5) / 6 * 16) + 127) & ~127;        /* :1798-1799 */
    if (pitch_out) *pitch_out = pitch;
    return (size_t)pitch * (size_t)(height * f->pitch_coeff / 2);   /* m_srcLines */
}

/* ------------------------------------------------------------------------------------------ */
/* DXVA2_ExtendedFormat — dxva2api.h bit layout (LSB first): SampleFormat:8, ChromaSubsampling:4, */
/* NominalRange:3, TransferMatrix:3, Lighting:4, Primaries:5, TransferFunction:5                  */
/* ------------------------------------------------------------------------------------------ */
#define EXF_CHROMA(v)   (((v) >> 8)  & 0xf)
#define EXF_RANGE(v)    (((v) >> 12) & 0x7)
#define EXF_MATRIX(v)   (((v) >> 15) & 0x7)
#define EXF_LIGHT(v)    (((v) >> 18) & 0xf)
#define EXF_PRIM(v)     (((v) >> 22) & 0x1f)
#define EXF_TRC(v)      (((v) >> 27) & 0x1f)
static inline uint32_t exf_set(uint32_t v, int shift, uint32_t mask, uint32_t x)
{ return (v & ~(mask << shift)) | ((x & mask) << shift); }

enum { CHROMA_MPEG1 = 1, CHROMA_MPEG2 = 5, CHROMA_COSITED = 7 };
enum { RANGE_0_255 = 1, RANGE_16_235 = 2 };
enum { MATRIX_709 = 1, MATRIX_601 = 2, MATRIX_240M = 3, MATRIX_2020 = 4, MATRIX_YCGCO = 7 };
enum { PRIM_709 = 2, PRIM_470M = 3, PRIM_470BG = 4, PRIM_170M = 5, PRIM_240M = 6, PRIM_2020 = 9, PRIM_DCIP3 = 11 };
enum { TRC_10 = 1, TRC_18 = 2, TRC_20 = 3, TRC_22 = 4, TRC_709 = 5, TRC_240M = 6, TRC_SRGB = 7, TRC_28 = 8,
       TRC_26 = 14, TRC_2084 = 15, TRC_HLG = 16 };

/* SpecifyExtendedFormat — Helper.cpp:1169-1211: CS_RGB -> 0, CS_YUV -> defaults, CS_GRAY untouched */
uint32_t orc_specify_extfmt(uint32_t v, int cformat, int w, int h)
{
    const fmt_info *f = find_fmt(cformat);
    if (!f) return v;
    if (f->cstype == CST_RGB) return 0;
    if (f->cstype == CST_GRAY) return v;
    if (f->subsampling != 420)            v = exf_set(v, 8, 0xf, 0);
    else if (EXF_CHROMA(v) == 0)          v = exf_set(v, 8, 0xf, CHROMA_MPEG2);
    if (EXF_RANGE(v) == 0)                v = exf_set(v, 12, 0x7, RANGE_16_235);
    if (EXF_MATRIX(v) == 0)               v = exf_set(v, 15, 0x7, (w <= 1024 && h <= 576) ? MATRIX_601 : MATRIX_709);
    if (EXF_LIGHT(v) == 0)                v = exf_set(v, 18, 0xf, 3 /* dim */);
    if (EXF_PRIM(v) == 0)                 v = exf_set(v, 22, 0x1f, PRIM_709);
    if (EXF_TRC(v) == 0)                  v = exf_set(v, 27, 0x1f, TRC_709);
    return v;
}

/* ------------------------------------------------------------------------------------------ */
/* csputils restatement — csputils.cpp:341-509 (mp_get_csp_mul, luma_coeffs, mp_get_csp_matrix) */
/* mp_csp: 1=BT_601 2=BT_709 3=SMPTE_240M 4=BT_2020_NC 8=YCGCO ; levels: 1=TV 2=PC            */
/* ------------------------------------------------------------------------------------------ */
static void luma_coeffs(float m[3][3], float lr, float lg, float lb)
{   /* csputils.cpp:380-389 — all in float, exactly as the initialiser list evaluates */
    m[0][0] = 1; m[0][1] = 0;                        m[0][2] = 2 * (1 - lr);
    m[1][0] = 1; m[1][1] = -2 * (1 - lb) * lb / lg;  m[1][2] = -2 * (1 - lr) * lr / lg;
    m[2][0] = 1; m[2][1] = 2 * (1 - lb);             m[2][2] = 0;
}

void orc_csp_matrix(int space, int levels_in, int bits, float brightness, float contrast,
                    float hue, float saturation, int gray, float mo[9], float co[3])
{
    float m[3][3];
    if (space <= 0 || space >= 9) space = 1;               /* AUTO -> BT_601 (csputils.cpp:395-396) */
    if (levels_in <= 0 || levels_in >= 3) levels_in = 1;   /* AUTO -> TV */
    if (space == 6) levels_in = -1;                        /* MP_CSP_RGB: identity, "anyfull" levels (:416-420) */
    switch (space) {
    case 6: { const float y[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}; memcpy(m, y, sizeof(y)); break; }
    case 1: luma_coeffs(m, 0.299f,  0.587f,  0.114f);  break;
    case 2: luma_coeffs(m, 0.2126f, 0.7152f, 0.0722f); break;
    case 3: luma_coeffs(m, 0.2122f, 0.7013f, 0.0865f); break;
    case 4: luma_coeffs(m, 0.2627f, 0.6780f, 0.0593f); break;
    case 8: { const float y[3][3] = {{1, -1, 1}, {1, 1, 0}, {1, -1, -1}}; memcpy(m, y, sizeof(y)); break; }
    default: luma_coeffs(m, 0.299f, 0.587f, 0.114f); break; /* other spaces never reach this path */
    }
    if (space >= 1 && space <= 4) {                         /* csputils.cpp:447-459 */
        float huecos = gray ? 0 : saturation * cosf(hue);
        float huesin = gray ? 0 : saturation * sinf(hue);
        for (int i = 0; i < 3; i++) {
            float u = m[i][1], v = m[i][2];
            m[i][1] = huecos * u - huesin * v;
            m[i][2] = huesin * u + huecos * v;
        }
    }
    /* mp_get_csp_mul(colorspace, input_bits, texture_bits) with input_bits == texture_bits == CDepth
       (DX11VideoProcessor.cpp:845): (1<<bits) / ((1<<bits) - 1.) * 255 / 256   — csputils.cpp:357 */
    double mul = (double)(1LL << bits) / ((double)(1LL << bits) - 1.) * 255 / 256;
    if (space == 6) mul = ((double)(1LL << bits) - 1.) / ((double)(1LL << bits) - 1.);   /* RGB: full range (:351-353) */
    double s = mul / 255;
    double ymin, ymax, cmax, cmid;
    if (levels_in == 1)       { ymin = 16 * s; ymax = 235 * s; cmax = 240 * s; cmid = 128 * s; }
    else if (levels_in == -1) { ymin = 0 * s;  ymax = 255 * s; cmax = 255 * s / 2; cmid = 0; }   /* anyfull (:474) */
    else                      { ymin = 0 * s;  ymax = 255 * s; cmax = 255 * s; cmid = 128 * s; }
    const double rgbmin = 0, rgbmax = 1;                     /* levels_out = PC */
    double ymul = (rgbmax - rgbmin) / (ymax - ymin);
    double cmul = (rgbmax - rgbmin) / (cmax - cmid) / 2;
    ymul *= contrast;
    cmul *= contrast;
    for (int i = 0; i < 3; i++) {
        m[i][0] = (float)(m[i][0] * ymul);
        m[i][1] = (float)(m[i][1] * cmul);
        m[i][2] = (float)(m[i][2] * cmul);
        float uv = m[i][1] + m[i][2];                        /* float + float, as written */
        co[i] = (float)(rgbmin - m[i][0] * ymin - uv * cmid + brightness);
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) mo[i * 3 + j] = m[i][j];
}

/* set_colorspace — Helper.cpp:949-1004 (only the fields the matrix uses) */
static void extfmt_to_csp(uint32_t v, int *space, int *levels)
{
    if (v == 0) { *space = 6; *levels = 2; return; }      /* MP_CSP_RGB, PC (:953-957) */
    switch (EXF_RANGE(v)) { case RANGE_0_255: *levels = 2; break; case RANGE_16_235: *levels = 1; break; default: *levels = 0; }
    switch (EXF_MATRIX(v)) {
    case MATRIX_709: *space = 2; break;  case MATRIX_601: *space = 1; break;
    case MATRIX_240M: *space = 3; break; case MATRIX_2020: *space = 4; break;
    case MATRIX_YCGCO: *space = 8; break; default: *space = 0;
    }
}

static void resolve_rect(const orc_params *p, int r[4])
{
    memcpy(r, p->src_rect, sizeof(int) * 4);
    if (!r[0] && !r[1] && !r[2] && !r[3]) { r[2] = p->width; r[3] = p->height; }
}

/* SetShaderConvertColorParams — DX11VideoProcessor.cpp:813-887 */
int orc_color_matrix(const orc_params *p, float out[12])
{
    const fmt_info *f = find_fmt(p->cformat);
    if (!f) return -1;
    int r[4]; resolve_rect(p, r);
    uint32_t ex = orc_specify_extfmt(p->exfmt, p->cformat, r[2] - r[0], r[3] - r[1]);
    int space, levels; extfmt_to_csp(ex, &space, &levels);
    float brightness = p->brightness / 255;                              /* :839 */
    float contrast = p->contrast;                                        /* :840 */
    float hue = (float)(p->hue / 180 * acos(-1));                        /* :841 */
    float m[9], c[3];
    if (p->dovi) {                                                       /* :817-834 — hue / saturation are not applied */
        for (int i = 0; i < 9; i++) m[i] = (float)p->dovi->ycc_to_rgb_matrix[i] * contrast;
        for (int i = 0; i < 3; i++) {
            /* `cmatrix.c[i] -= cmatrix.m[i][j] * offset[j]` (:828-830): float -= float * double, i.e. every step is rounded to float.  The
               accumulator is volatile on purpose: gcc 11 -O3 vectorises rows 0 and 1 of the plain loop and drops the two intermediate
               roundings (three subpd, one cvtpd2ps) — one ulp off the reference's arithmetic, found by the round-6 fuzz (case 6375: the plain
               tier, whose host code clang compiles as written, disagreed with this oracle on two channels of a 54 k-pixel frame). */
            volatile float acc = brightness;
            for (int j = 0; j < 3; j++) acc = (float)((double)acc - (double)m[3 * i + j] * p->dovi->ycc_to_rgb_offset[j]);
            c[i] = acc;
        }
    } else
    orc_csp_matrix(space, levels, f->cdepth, brightness, contrast, hue, p->saturation, f->cstype == CST_GRAY, m, c);
    if (f->cstype == CST_RGB && f->layout == LAY_PLANAR && f->planes == 3) {     /* GBRP: (x,y,z) -> (y,z,x) per row, :863-867 */
        for (int i = 0; i < 3; i++) { float x = m[3 * i], y = m[3 * i + 1], z = m[3 * i + 2]; m[3 * i] = y; m[3 * i + 1] = z; m[3 * i + 2] = x; }
    } else if (f->cstype == CST_GRAY) {                        /* :868-873 */
        m[3] = m[4]; m[4] = 0;
        m[6] = m[8]; m[8] = 0;
    }
    memcpy(out, m, sizeof(m)); memcpy(out + 9, c, sizeof(c));
    return 0;
}

float orc_luminance_scale(int nits) { return 10000.0f / nits; }          /* :891 */

/* ------------------------------------------------------------------------------------------ */
/* gamut matrix — csputils.cpp:10-49 (invert/mul), :51-200 (primaries), :228-259 (rgb2xyz), :549-557 */
/* mp_csp_prim: 3=BT_709 4=BT_2020                                                             */
/* ------------------------------------------------------------------------------------------ */
static void invert3x3(float m[3][3])
{
    float m00 = m[0][0], m01 = m[0][1], m02 = m[0][2],
          m10 = m[1][0], m11 = m[1][1], m12 = m[1][2],
          m20 = m[2][0], m21 = m[2][1], m22 = m[2][2];
    m[0][0] =  (m11 * m22 - m21 * m12);
    m[0][1] = -(m01 * m22 - m21 * m02);
    m[0][2] =  (m01 * m12 - m11 * m02);
    m[1][0] = -(m10 * m22 - m20 * m12);
    m[1][1] =  (m00 * m22 - m20 * m02);
    m[1][2] = -(m00 * m12 - m10 * m02);
    m[2][0] =  (m10 * m21 - m20 * m11);
    m[2][1] = -(m00 * m21 - m20 * m01);
    m[2][2] =  (m00 * m11 - m10 * m01);
    float det = m00 * m[0][0] + m10 * m[0][1] + m20 * m[0][2];
    det = 1.0f / det;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] *= det;
}
static void mul3x3(float a[3][3], float b[3][3])
{
    float a00 = a[0][0], a01 = a[0][1], a02 = a[0][2],
          a10 = a[1][0], a11 = a[1][1], a12 = a[1][2],
          a20 = a[2][0], a21 = a[2][1], a22 = a[2][2];
    for (int i = 0; i < 3; i++) {
        a[0][i] = a00 * b[0][i] + a01 * b[1][i] + a02 * b[2][i];
        a[1][i] = a10 * b[0][i] + a11 * b[1][i] + a12 * b[2][i];
        a[2][i] = a20 * b[0][i] + a21 * b[1][i] + a22 * b[2][i];
    }
}
typedef struct { float rx, ry, gx, gy, bx, by, wx, wy; } prim_t;
static prim_t primaries(int prim)
{
    const float d65x = 0.31271f, d65y = 0.32902f;                         /* csputils.cpp:73 */
    switch (prim) {
    case 4:  { prim_t p = {0.708f, 0.292f, 0.170f, 0.797f, 0.131f, 0.046f, d65x, d65y}; return p; }
    case 2:  { prim_t p = {0.640f, 0.330f, 0.290f, 0.600f, 0.150f, 0.060f, d65x, d65y}; return p; } /* 601-625 */
    case 1:  { prim_t p = {0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, d65x, d65y}; return p; } /* 601-525 */
    default: { prim_t p = {0.640f, 0.330f, 0.300f, 0.600f, 0.150f, 0.060f, d65x, d65y}; return p; } /* 709 */
    }
}
static void rgb2xyz(prim_t s, float m[3][3])
{
    float S[3], X[4], Z[4];
    X[0] = s.rx / s.ry; X[1] = s.gx / s.gy; X[2] = s.bx / s.by; X[3] = s.wx / s.wy;
    Z[0] = (1 - s.rx - s.ry) / s.ry; Z[1] = (1 - s.gx - s.gy) / s.gy;
    Z[2] = (1 - s.bx - s.by) / s.by; Z[3] = (1 - s.wx - s.wy) / s.wy;
    for (int i = 0; i < 3; i++) { m[0][i] = X[i]; m[1][i] = 1; m[2][i] = Z[i]; }
    invert3x3(m);
    for (int i = 0; i < 3; i++) S[i] = m[i][0] * X[3] + m[i][1] * 1 + m[i][2] * Z[3];
    for (int i = 0; i < 3; i++) { m[0][i] = S[i] * X[i]; m[1][i] = S[i] * 1; m[2][i] = S[i] * Z[i]; }
}
void orc_gamut_matrix(int prim_in, int prim_out, float out[9])
{
    float in[3][3], m[3][3];
    rgb2xyz(primaries(prim_in), in);
    rgb2xyz(primaries(prim_out), m);
    invert3x3(m);
    mul3x3(m, in);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out[i * 3 + j] = m[i][j];
}
void orc_gamut_2020_to_709(float out[9]) { orc_gamut_matrix(4, 3, out); }

/* ------------------------------------------------------------------------------------------ */
/* shader device functions                                                                    */
/* ------------------------------------------------------------------------------------------ */
/* Shaders/convert/st2084.hlsl:1-5 */
#define ST2084_m1 (2610.0f / (4096.0f * 4.0f))
#define ST2084_m2 ((2523.0f / 4096.0f) * 128.0f)
#define ST2084_c1 (3424.0f / 4096.0f)
#define ST2084_c2 ((2413.0f / 4096.0f) * 32.0f)
#define ST2084_c3 ((2392.0f / 4096.0f) * 32.0f)

float orc_st2084_to_linear(float x, float factor)       /* st2084.hlsl:9-16 */
{
    x = hlsl_pow(x, 1.0f / ST2084_m2);
    x = fmaxf(x - ST2084_c1, 0.0f) / (ST2084_c2 - ST2084_c3 * x);
    x = hlsl_pow(x, 1.0f / ST2084_m1);
    x *= factor;
    return x;
}
float orc_linear_to_st2084(float x, float divider)      /* st2084.hlsl:18-25 */
{
    x /= divider;
    x = hlsl_pow(x, ST2084_m1);
    x = (ST2084_c1 + ST2084_c2 * x) / (1.0f + ST2084_c3 * x);
    x = hlsl_pow(x, ST2084_m2);
    return x;
}
void orc_hlg_to_linear(float rgb[3])                    /* hlg.hlsl:1-20 */
{
    const float B67_a = 0.17883277f, B67_b = 0.28466892f, B67_c = 0.55991073f, B67_inv_r2 = 4.0f;
    for (int i = 0; i < 3; i++)
        rgb[i] = (rgb[i] <= 0.5f) ? rgb[i] * rgb[i] * B67_inv_r2 : crm_expf((rgb[i] - B67_c) / B67_a) + B67_b;
    float ootf_ys = 2000.0f * (0.2627f * rgb[0] + 0.6780f * rgb[1] + 0.0593f * rgb[2]);
    float g = hlsl_pow(ootf_ys, 0.2f);
    for (int i = 0; i < 3; i++) rgb[i] *= g;
}
float orc_hable(float x)                                /* hdr_tone_mapping.hlsl:1-6 */
{
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + (C * B)) + (D * E)) / (x * (A * x + B) + (D * F))) - E / F;
}
void orc_tonemap_hable(float rgb[3])                    /* hdr_tone_mapping.hlsl:8-13 */
{
    const float div = orc_hable(4.8f);
    for (int i = 0; i < 3; i++) rgb[i] = orc_hable(rgb[i]) / div;
}

static void mat3_apply(const float m[9], float rgb[3])
{   /* HLSL mul(float3x3, float3): row dot products, left-to-right */
    float r = m[0] * rgb[0] + m[1] * rgb[1] + m[2] * rgb[2];
    float g = m[3] * rgb[0] + m[4] * rgb[1] + m[5] * rgb[2];
    float b = m[6] * rgb[0] + m[7] * rgb[1] + m[8] * rgb[2];
    rgb[0] = r; rgb[1] = g; rgb[2] = b;
}

/* The tail GetShaderConvertColor appends after "//convert color" — Shaders.cpp:861-923.
 * (alpha follows the same scalar ops in HLSL but is forced to 1 by every store format; not modelled) */
void orc_hdr_tail(float rgb[3], int trc, int prim, int convert_to_sdr, float lum_scale)
{
    orc_hdr_tail_ex(rgb, trc, prim, convert_to_sdr, lum_scale, 0);
}

/* hdr_output = m_bHdrPassthroughSupport && (m_bHdrPassthrough || m_bHdrLocalToneMapping): convertType
 * (DX11VideoProcessor.cpp:2948-2950) is then never SHADER_CONVERT_TO_SDR, and SHADER_CONVERT_TO_PQ for HLG */
typedef struct { int active, l2_enabled; float lms[9], k[5]; } dovi_tail_t;
static void dovi_trims_convert(float c[3], const float k[5]);
static void hdr_tail_full(float rgb[3], int trc, int prim, int convert_to_sdr, float lum_scale, int hdr_output, const dovi_tail_t *dv);

void orc_hdr_tail_ex(float rgb[3], int trc, int prim, int convert_to_sdr, float lum_scale, int hdr_output)
{
    hdr_tail_full(rgb, trc, prim, convert_to_sdr, lum_scale, hdr_output, NULL);
}

static void hdr_tail_full(float rgb[3], int trc, int prim, int convert_to_sdr, float lum_scale, int hdr_output, const dovi_tail_t *dv)
{
    float gm[9];
    const int dovi = dv && dv->active;
    const int bt2020 = (prim == PRIM_2020);
    const int hdr2sdr = convert_to_sdr && !hdr_output && (trc == TRC_2084 || trc == TRC_HLG || dovi);   /* :614, :2948 */
    const int apply_hlg = (trc == TRC_HLG) && !dovi;                                   /* bApplyHLG :615 */
    int is_linear = 0;
    if (dovi) {                                                                        /* :826-859 */
        for (int i = 0; i < 3; i++) rgb[i] = orc_st2084_to_linear(fmaxf(rgb[i], 0.0f), 1.0f);
        mat3_apply(dv->lms, rgb);
        for (int i = 0; i < 3; i++) rgb[i] = orc_linear_to_st2084(fmaxf(rgb[i], 0.0f), 1.0f);
    }
    if (!hdr2sdr && hdr_output && apply_hlg) {                                    /* bConvertHLGtoPQ :616,885-891 */
        for (int i = 0; i < 3; i++) rgb[i] = saturatef(rgb[i]);
        orc_hlg_to_linear(rgb);
        for (int i = 0; i < 3; i++) rgb[i] = orc_linear_to_st2084(rgb[i], 1000.0f);
        return;
    }
    if (hdr2sdr) {
        if (apply_hlg) {                                                               /* :862-868 */
            for (int i = 0; i < 3; i++) rgb[i] = saturatef(rgb[i]);
            orc_hlg_to_linear(rgb);
            for (int i = 0; i < 3; i++) rgb[i] = orc_linear_to_st2084(rgb[i], 1000.0f);
        }
        for (int i = 0; i < 3; i++) rgb[i] = saturatef(rgb[i]);                        /* :870-872 */
        if (dovi && dv->l2_enabled) dovi_trims_convert(rgb, dv->k);                    /* :873-877 */
        for (int i = 0; i < 3; i++) rgb[i] = orc_st2084_to_linear(rgb[i], lum_scale);  /* :879 */
        orc_tonemap_hable(rgb);                                                        /* :880 */
        orc_gamut_2020_to_709(gm); mat3_apply(gm, rgb);                                /* :881 */
        is_linear = 1;
    } else if (bt2020) {                                                               /* :892-915 */
        float g = 0;
        switch (trc) {
        case TRC_10: g = 1.0f; break;  /* emitted text is not valid HLSL; treated as "nothing" */
        case TRC_18: g = 1.8f; break;
        case TRC_20: g = 2.0f; break;
        case TRC_HLG: case TRC_22: case TRC_709: case TRC_240M: case TRC_SRGB: g = 2.2f; break;
        case TRC_28: g = 2.8f; break;
        case TRC_26: g = 2.6f; break;
        default: g = 0; break;
        }
        if (g != 0) {
            for (int i = 0; i < 3; i++) rgb[i] = saturatef(rgb[i]);
            if (trc != TRC_10) for (int i = 0; i < 3; i++) rgb[i] = hlsl_pow(rgb[i], g);
            orc_gamut_2020_to_709(gm); mat3_apply(gm, rgb);
            is_linear = 1;
        }
    }
    if (is_linear)                                                                     /* :917-923 */
        for (int i = 0; i < 3; i++) rgb[i] = hlsl_pow(saturatef(rgb[i]), 1.0f / 2.2f);
}

/* ------------------------------------------------------------------------------------------ */
/* Dolby Vision: reshaping curves, IPT-PQ -> LMS -> RGB, level-2 trims                          */
/* ------------------------------------------------------------------------------------------ */
#define DOVI_RESHAPE_POLY 1u
#define DOVI_RESHAPE_MMR  2u

/* SetShaderDoviCurves — DX11VideoProcessor.cpp:1055-1141.  SetShaderDoviCurvesPoly (:990-1053), used when no piece of any
 * curve is MMR, fills pivots and polynomial coefficients with the same expressions. */
void orc_dovi_pack_curves(const orc_dovi *d, orc_dovi_cb cb[3], int *has_mmr_any)
{
    memset(cb, 0, sizeof(orc_dovi_cb) * 3);
    *has_mmr_any = 0;
    for (int c = 0; c < 3; c++) {
        const orc_dovi_curve *curve = &d->curves[c];
        orc_dovi_cb *out = &cb[c];
        int has_poly = 0, has_mmr = 0, mmr_single = 1;
        uint32_t mmr_idx = 0, min_order = 3, max_order = 1;
        const float scale_coef = 1.0f / (1 << d->coef_log2_denom);                       /* :1068 */
        const int num_coef = curve->num_pivots - 1;
        for (int i = 0; i < num_coef; i++) {
            switch (curve->mapping_idc[i]) {
            case 0:                                                                       /* :1072-1078 */
                has_poly = 1;
                out->coeffs[i][0] = scale_coef * curve->poly_coef[i][0];
                out->coeffs[i][1] = (curve->poly_order[i] >= 1) ? scale_coef * curve->poly_coef[i][1] : 0.0f;
                out->coeffs[i][2] = (curve->poly_order[i] >= 2) ? scale_coef * curve->poly_coef[i][2] : 0.0f;
                out->coeffs[i][3] = 0.0f;
                break;
            case 1:                                                                       /* :1079-1100 */
                min_order = (uint32_t)((int)min_order < (int)curve->mmr_order[i] ? (int)min_order : (int)curve->mmr_order[i]);
                max_order = (uint32_t)((int)max_order > (int)curve->mmr_order[i] ? (int)max_order : (int)curve->mmr_order[i]);
                mmr_single = !has_mmr;
                has_mmr = 1;
                out->coeffs[i][0] = scale_coef * curve->mmr_constant[i];
                out->coeffs[i][1] = (float)mmr_idx;
                out->coeffs[i][3] = (float)curve->mmr_order[i];
                for (int j = 0; j < curve->mmr_order[i]; j++) {
                    out->mmr[mmr_idx][0] = scale_coef * curve->mmr_coef[i][j][0];
                    out->mmr[mmr_idx][1] = scale_coef * curve->mmr_coef[i][j][1];
                    out->mmr[mmr_idx][2] = scale_coef * curve->mmr_coef[i][j][2];
                    out->mmr[mmr_idx][3] = 0.0f;
                    mmr_idx++;
                    out->mmr[mmr_idx][0] = scale_coef * curve->mmr_coef[i][j][3];
                    out->mmr[mmr_idx][1] = scale_coef * curve->mmr_coef[i][j][4];
                    out->mmr[mmr_idx][2] = scale_coef * curve->mmr_coef[i][j][5];
                    out->mmr[mmr_idx][3] = scale_coef * curve->mmr_coef[i][j][6];
                    mmr_idx++;
                }
                break;
            }
        }
        const float scale = 1.0f / ((1 << d->bl_bit_depth) - 1);                         /* :1104 */
        const int n = curve->num_pivots - 2;
        for (int i = 0; i < n; i++) out->pivots[i] = scale * curve->pivots[i + 1];
        for (int i = n; i < 7; i++) out->pivots[i] = 1e9f;
        if (has_poly) out->methods = DOVI_RESHAPE_POLY;                                  /* :1113-1121 */
        if (has_mmr) {
            out->methods |= DOVI_RESHAPE_MMR;
            out->mmr_single = (uint32_t)mmr_single;
            out->min_order = min_order;
            out->max_order = max_order;
            *has_mmr_any = 1;                                                             /* :2305-2311 */
        }
    }
}

/* reshape_mmr — Shaders.cpp:734-762.  dot() is modelled like mul(): products summed left to right, unfused. */
static float dovi_reshape_mmr(const orc_dovi_cb *cv, const float coeffs[4], const float sig[3])
{
    const uint32_t mmr_idx = cv->mmr_single ? 0u : (uint32_t)coeffs[1];
    const float (*m)[4] = cv->mmr;
    float s = coeffs[0];
    float sigX[4] = {sig[0] * sig[1], sig[0] * sig[2], sig[1] * sig[2], 0.0f};
    sigX[3] = sigX[0] * sig[2];
    s += m[mmr_idx][0] * sig[0] + m[mmr_idx][1] * sig[1] + m[mmr_idx][2] * sig[2];
    s += m[mmr_idx + 1][0] * sigX[0] + m[mmr_idx + 1][1] * sigX[1] + m[mmr_idx + 1][2] * sigX[2] + m[mmr_idx + 1][3] * sigX[3];
    if (cv->max_order >= 2) {
        const uint32_t order = (uint32_t)coeffs[3];
        if (cv->min_order < 2 && order < 2) return s;
        float sig2[3], sigX2[4];
        for (int i = 0; i < 3; i++) sig2[i] = sig[i] * sig[i];
        for (int i = 0; i < 4; i++) sigX2[i] = sigX[i] * sigX[i];
        s += m[mmr_idx + 2][0] * sig2[0] + m[mmr_idx + 2][1] * sig2[1] + m[mmr_idx + 2][2] * sig2[2];
        s += m[mmr_idx + 3][0] * sigX2[0] + m[mmr_idx + 3][1] * sigX2[1] + m[mmr_idx + 3][2] * sigX2[2] + m[mmr_idx + 3][3] * sigX2[3];
        if (cv->max_order == 3) {
            if (cv->min_order < 3 && order < 3) return s;
            float sig3[3], sigX3[4];
            for (int i = 0; i < 3; i++) sig3[i] = sig2[i] * sig[i];
            for (int i = 0; i < 4; i++) sigX3[i] = sigX2[i] * sigX[i];
            s += m[mmr_idx + 4][0] * sig3[0] + m[mmr_idx + 4][1] * sig3[1] + m[mmr_idx + 4][2] * sig3[2];
            s += m[mmr_idx + 5][0] * sigX3[0] + m[mmr_idx + 5][1] * sigX3[1] + m[mmr_idx + 5][2] * sigX3[2] + m[mmr_idx + 5][3] * sigX3[3];
        }
    }
    return s;
}

/* ShaderDoviReshape (has_mmr) / ShaderDoviReshapePoly — Shaders.cpp:531-589 */
void orc_dovi_reshape(const orc_dovi_cb cb[3], int has_mmr, float color[3])
{
    const float sig[3] = {saturatef(color[0]), saturatef(color[1]), saturatef(color[2])};
    for (int c = 0; c < 3; c++) {
        const orc_dovi_cb *cv = &cb[c];
        float s = sig[c];
#define DV_TEST(i) (s < cv->pivots[i])
        const int k = DV_TEST(3) ? (DV_TEST(1) ? (DV_TEST(0) ? 0 : 1) : (DV_TEST(2) ? 2 : 3))
                                 : (DV_TEST(5) ? (DV_TEST(4) ? 4 : 5) : (DV_TEST(6) ? 6 : 7));
#undef DV_TEST
        const float *co = cv->coeffs[k];
        if (!has_mmr) {
            s = (co[2] * s + co[1]) * s + co[0];
        } else if (cv->methods == DOVI_RESHAPE_POLY + DOVI_RESHAPE_MMR) {
            if (co[3] == 0.0f) s = (co[2] * s + co[1]) * s + co[0];
            else s = dovi_reshape_mmr(cv, co, sig);
        } else if (cv->methods == DOVI_RESHAPE_POLY) {
            s = (co[2] * s + co[1]) * s + co[0];
        } else {
            s = dovi_reshape_mmr(cv, co, sig);
        }
        color[c] = saturatef(s);
    }
}

/* Shaders.cpp:826-842: mat = dovi_lms2rgb x (float)rgb_to_lms_matrix, mul_matrix3x3 of csputils.cpp:531-538 */
void orc_dovi_lms_matrix(const orc_dovi *d, float out[9])
{
    static const float lms2rgb[3][3] = {
        { 3.06441879f, -2.16597676f,  0.10155818f},
        {-0.65612108f,  1.78554118f, -0.12943749f},
        { 0.01736321f, -0.04725154f,  1.03004253f},
    };
    float b[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) b[i][j] = (float)d->rgb_to_lms_matrix[i * 3 + j];
    for (int i = 0; i < 3; i++)
        for (int r = 0; r < 3; r++)
            out[r * 3 + i] = lms2rgb[r][0] * b[0][i] + lms2rgb[r][1] * b[1][i] + lms2rgb[r][2] * b[2][i];
}

/* host-side PQ helpers of CopySample — DX11VideoProcessor.cpp:2324-2345 (libm powf, not the shader pow) */
static float dovi_pq_to_nits(float x)
{
    x = powf(x, 1.0f / ST2084_m2);
    x = fmaxf(x - ST2084_c1, 0.0f) / (ST2084_c2 - ST2084_c3 * x);
    x = powf(x, 1.0f / ST2084_m1);
    return x * 10000.0f;
}
static float dovi_nits_to_pq(float y)
{
    y /= 10000.0f;
    y = fmaxf(y, 0.0f);
    y = powf(y, ST2084_m1);
    y = (ST2084_c1 + ST2084_c2 * y) / (1.0f + ST2084_c3 * y);
    return powf(y, ST2084_m2);
}

int orc_dovi_l1_nits(const orc_dovi *d, uint32_t out[3])                          /* :2347-2372 */
{
    out[0] = out[1] = out[2] = 0;
    if (!d->l1_present) return 0;
    uint32_t mn = d->l1_min_pq, mx = d->l1_max_pq, av = d->l1_avg_pq;      /* UINT fields */
    if (d->l3_present) {
        mn = mn + d->l3_min_pq_offset - 2048;
        mx = mx + d->l3_max_pq_offset - 2048;
        av = av + d->l3_avg_pq_offset - 2048;
    }
    out[0] = (uint32_t)dovi_pq_to_nits(mn / 4095.f);
    out[1] = (uint32_t)dovi_pq_to_nits(mx / 4095.f);
    out[2] = (uint32_t)dovi_pq_to_nits(av / 4095.f);
    return 1;
}

static float lerp_std(float a, float b, float t)
{   /* std::lerp for finite a, b and t in [0,1] (the only range the caller produces): exact at both ends, monotonic */
    if ((a <= 0 && b >= 0) || (a >= 0 && b <= 0)) return t * b + (1 - t) * a;
    if (t == 1) return b;
    const float x = a + t * (b - a);
    return (t > 1) == (b > a) ? (b < x ? x : b) : (b > x ? x : b);
}

int orc_dovi_l2_constants(const orc_dovi *d, int display_nits, float k[5])       /* :2383-2469, :954-960 */
{
    const float display_pq = dovi_nits_to_pq((float)display_nits);
    int lower = -1, upper = -1, present = 0;
    float dl = 1.0f, du = 1.0f;
    const int n = d->n_l2 > 32 ? 32 : (int)d->n_l2;
    for (int i = 0; i < n; i++) {
        present = 1;
        const float target_pq = d->l2[i].target_max_pq / 4095.0f;
        if (target_pq <= display_pq) { const float dist = display_pq - target_pq; if (dist < dl) { dl = dist; lower = i; } }
        else { const float dist = target_pq - display_pq; if (dist < du) { du = dist; upper = i; } }
    }
    float l2[5] = {0, 0, 0, 0, 0};      /* m_DoviExtensionMetadata.L2 zero-initialised: chroma, sat, slope, offset, power */
    if (present) {
        float t_slope = 1.0f, t_offset = 0.0f, t_power = 1.0f, t_chroma = 0.0f, t_sat = 0.0f;
        if (lower != -1 && upper != -1) {
            const orc_dovi_l2 *a = &d->l2[lower], *b = &d->l2[upper];
            const float lower_pq = a->target_max_pq / 4095.0f, upper_pq = b->target_max_pq / 4095.0f;
            float w = (upper_pq != lower_pq) ? (display_pq - lower_pq) / (upper_pq - lower_pq) : 0.0f;
            w = w < 0.0f ? 0.0f : (w > 1.0f ? 1.0f : w);
            t_slope = lerp_std((float)a->trim_slope, (float)b->trim_slope, w);
            t_offset = lerp_std((float)a->trim_offset, (float)b->trim_offset, w);
            t_power = lerp_std((float)a->trim_power, (float)b->trim_power, w);
            t_chroma = lerp_std((float)a->trim_chroma_weight, (float)b->trim_chroma_weight, w);
            t_sat = lerp_std((float)a->trim_saturation_gain, (float)b->trim_saturation_gain, w);
        } else if (lower != -1) {
            const orc_dovi_l2 *a = &d->l2[lower];
            const float master_pq = d->source_max_pq / 4095.0f, lower_pq = a->target_max_pq / 4095.0f;
            float w = (master_pq > lower_pq) ? (display_pq - lower_pq) / (master_pq - lower_pq) : 0.0f;
            w = w < 0.0f ? 0.0f : (w > 1.0f ? 1.0f : w);
            t_slope = lerp_std((float)a->trim_slope, 2048.0f, w);
            t_offset = lerp_std((float)a->trim_offset, 2048.0f, w);
            t_power = lerp_std((float)a->trim_power, 2048.0f, w);
            t_chroma = lerp_std((float)a->trim_chroma_weight, 2048.0f, w);
            t_sat = lerp_std((float)a->trim_saturation_gain, 2048.0f, w);
        } else if (upper != -1) {
            const orc_dovi_l2 *b = &d->l2[upper];
            t_slope = b->trim_slope; t_offset = b->trim_offset; t_power = b->trim_power;
            t_chroma = b->trim_chroma_weight; t_sat = b->trim_saturation_gain;
        }
        l2[0] = t_chroma / 4096.0f; l2[1] = t_sat / 4096.0f;
        l2[2] = t_slope / 4096.0f; l2[3] = t_offset / 4096.0f; l2[4] = t_power / 4096.0f;
    }
    k[0] = l2[0] - 0.5f; k[1] = l2[1] - 0.5f; k[2] = l2[2] + 0.5f; k[3] = l2[3] - 0.5f; k[4] = l2[4] + 0.5f;
    return present;
}

/* convert-shader DolbyVisionTrims — Shaders.cpp:766-773 (PQ-coded colour) */
/* the Dolby Vision tail stage by stage over an array (tests: the product's plain tier against this, bit for bit; stages as mpcvr_eval_dovi_tail) */
void orc_eval_dovi_tail(int stage, const float *rgb, float *out, size_t n, const float lms[9], const float k[5], int l2, float lum_scale)
{
    float gm[9];
    orc_gamut_2020_to_709(gm);
    for (size_t i = 0; i < n; i++) {
        float c[3] = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        for (int j = 0; j < 3; j++) c[j] = orc_st2084_to_linear(fmaxf(c[j], 0.0f), 1.0f);
        mat3_apply(lms, c);
        for (int j = 0; j < 3; j++) c[j] = orc_linear_to_st2084(fmaxf(c[j], 0.0f), 1.0f);
        if (stage >= 1) { for (int j = 0; j < 3; j++) c[j] = saturatef(c[j]); if (l2) dovi_trims_convert(c, k); }
        if (stage >= 2) for (int j = 0; j < 3; j++) c[j] = orc_st2084_to_linear(c[j], lum_scale);
        if (stage >= 3) orc_tonemap_hable(c);
        if (stage >= 4) mat3_apply(gm, c);
        if (stage >= 5) for (int j = 0; j < 3; j++) c[j] = hlsl_pow(saturatef(c[j]), 1.0f / 2.2f);
        out[3 * i] = c[0]; out[3 * i + 1] = c[1]; out[3 * i + 2] = c[2];
    }
}

static void dovi_trims_convert(float c[3], const float k[5])
{
    for (int i = 0; i < 3; i++) c[i] = hlsl_pow((c[i] * k[2]) + k[3], k[4]);
    const float Y = 0.2627f * c[0] + 0.6780f * c[1] + 0.0593f * c[2];
    for (int i = 0; i < 3; i++) c[i] = c[i] * hlsl_pow((1.0f + k[0]) * c[i] / Y, k[1]);
}

/* Checker for the product's UNORM-load shortcut (vp_device.h unorm_div): q = code*(1/maxv), q' = fma(fma(-q,maxv,code),1/maxv,q)
 * must equal the IEEE quotient code/maxv this oracle uses, for every integer code in [0, maxv].  Returns the mismatches. */
int orc_check_unorm_div(int maxv)
{
    const float d = (float)maxv, r = 1.0f / d;
    int bad = 0;
    for (int x = 0; x <= maxv; x++) {
        const float xf = (float)x, ref = xf / d, q = xf * r;
        const float q2 = fmaf(fmaf(-q, d, xf), r, q);
        if (q2 != ref) bad++;
    }
    return bad;
}

/* ------------------------------------------------------------------------------------------ */
/* HDR10 -> HDR10 local tone mapping — Shaders/d3d11/ps_hdr10_tonemap.hlsl:272-336 (post-scale step of Process,   */
/* DX11VideoProcessor.cpp:3359-3367); constants as SetHDR10ShaderParams sanitises them (:907-917).  The Dolby    */
/* Vision L2 trims (L2Enabled, cbuffer b1 bound at :3362-3364) follow the PQ->linear step.                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float min_m, max_m, max_cll, max_fall, display_max; int selection; } hdr_tm_t;

/* SetHDR10ShaderParams (DX11VideoProcessor.cpp:911-916): defaults and clamps in front of the cbuffer */
static void hdr_tm_sanitise(hdr_tm_t *t)
{
    if (t->min_m <= 0.f) t->min_m = 0.f;
    if (t->max_m <= 10.f) t->max_m = 1000.f;
    if (t->max_cll <= 10.f) t->max_cll = t->max_m;
    if (t->max_fall <= 1.f) t->max_fall = t->max_cll;
    if (t->display_max < 100.f || t->display_max > 10000.f) t->display_max = 1000.f;
    if (t->selection < 1 || t->selection > 6) t->selection = 1;
}
static hdr_tm_t hdr_tm_params(const orc_params *p)
{
    hdr_tm_t t = {p->hdr_min_mastering, p->hdr_max_mastering, p->hdr_max_cll, p->hdr_max_fall, p->hdr_display_max_nits, p->hdr_tonemap_type};
    uint32_t l1[3];
    if (p->dovi && orc_dovi_l1_nits(p->dovi, l1)) {          /* DX11VideoProcessor.cpp:2716-2720: L1 min, max, max, avg; type 5 -> 6 */
        t.min_m = (float)l1[0]; t.max_m = (float)l1[1]; t.max_cll = (float)l1[1]; t.max_fall = (float)l1[2];
        if (t.selection == 5) t.selection = 6;
    }
    hdr_tm_sanitise(&t);
    return t;
}
void orc_hdr10_params(float min_m, float max_m, float max_cll, float max_fall, float display_max, int selection, uint32_t out6[6])
{
    hdr_tm_t t = {min_m, max_m, max_cll, max_fall, display_max, selection};
    hdr_tm_sanitise(&t);
    memcpy(out6, &t, 20);
    out6[5] = (uint32_t)t.selection;
}

static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }
static float pl_smoothstep(float e0, float e1, float x)
{
    float t = (x - e0) / (e1 - e0);
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    return t * t * (3.0f - 2.0f * t);
}

void orc_hdr10_tonemap(float c[3], const orc_params *p)
{
    const hdr_tm_t k = hdr_tm_params(p);
    for (int i = 0; i < 3; i++) c[i] = orc_st2084_to_linear(saturatef(c[i]), 10000.0f);     /* :275-277 */
    float l2k[5];
    if (p->dovi && orc_dovi_l2_constants(p->dovi, (int)p->hdr_display_max_nits, l2k)) {     /* L2Enabled: DolbyVisionTrims :257-270 */
        for (int i = 0; i < 3; i++) c[i] = orc_linear_to_st2084(c[i], 10000.0f);
        dovi_trims_convert(c, l2k);
        for (int i = 0; i < 3; i++) c[i] = orc_st2084_to_linear(c[i], 10000.0f);
    }
    if (k.selection == 5) {                                                                 /* BT2390Tonemap :68-124 */
        float safe = k.max_cll;
        if (safe <= 10.0f) safe = k.max_m;
        if (safe <= 10.0f) safe = 1000.0f;
        if (!(k.display_max >= safe)) {
            const float avg = 0.2627f * c[0] + 0.6780f * c[1] + 0.0593f * c[2];
            if (!(avg <= 0.000001f)) {
                const float max_pq = orc_linear_to_st2084(safe, 10000.0f), tgt_pq = orc_linear_to_st2084(k.display_max, 10000.0f);
                const float e1 = orc_linear_to_st2084(avg, 10000.0f);
                float ks = 1.5f * tgt_pq - 0.5f * max_pq;
                ks = fmaxf(0.0f, ks);
                float e2 = e1;
                if (e1 > ks) {
                    const float t = (e1 - ks) / fmaxf(1e-6f, max_pq - ks), t2 = t * t, t3 = t2 * t;
                    e2 = (2.0f * t3 - 3.0f * t2 + 1.0f) * ks + (t3 - 2.0f * t2 + t) * (max_pq - ks) + (-2.0f * t3 + 3.0f * t2) * tgt_pq;
                }
                const float lin = orc_st2084_to_linear(e2, 10000.0f);
                const float g = lin / avg;
                for (int i = 0; i < 3; i++) c[i] = c[i] * g;
            }
        }
        for (int i = 0; i < 3; i++) c[i] = orc_linear_to_st2084(c[i], 10000.0f);
        return;
    }
    if (k.selection == 6) {                                                                 /* ST209410Tonemap :133-205 */
        if (!(k.display_max >= k.max_cll)) {
            const float src_min = orc_linear_to_st2084(k.min_m, 10000.0f), src_max = orc_linear_to_st2084(k.max_cll, 10000.0f);
            const float src_avg = orc_linear_to_st2084(k.max_fall, 10000.0f);
            const float dst_min = orc_linear_to_st2084(0.0f, 10000.0f), dst_max = orc_linear_to_st2084(k.display_max, 10000.0f);
            const float min_knee = 0.1f, max_knee = 0.8f, def_knee = 0.4f, knee_adaptation = 0.4f;
            const float src_knee_min = lerpf(src_min, src_max, min_knee), src_knee_max = lerpf(src_min, src_max, max_knee);
            const float dst_knee_min = lerpf(dst_min, dst_max, min_knee), dst_knee_max = lerpf(dst_min, dst_max, max_knee);
            float src_knee = (k.max_fall > 0.0f) ? src_avg : lerpf(src_min, src_max, def_knee);
            src_knee = fminf(fmaxf(src_knee, src_knee_min), src_knee_max);
            const float target = (src_knee - src_min) / (src_max - src_min);
            const float adapted = lerpf(dst_min, dst_max, target);
            const float tuning = 1.0f - pl_smoothstep(max_knee, def_knee, target) * pl_smoothstep(min_knee, def_knee, target);
            const float adaptation = lerpf(knee_adaptation, 1.0f, tuning);
            float dst_knee = lerpf(src_knee, adapted, adaptation);
            dst_knee = fminf(fmaxf(dst_knee, dst_knee_min), dst_knee_max);
            const float x2 = orc_st2084_to_linear(src_knee, 10000.0f), y2 = orc_st2084_to_linear(dst_knee, 10000.0f);
            const float x1 = k.min_m, x3 = k.max_cll, y1 = 0.0f, y3 = k.display_max;
            const float m00 = x2 * x3 * (y2 - y3), m01 = x1 * x3 * (y3 - y1), m02 = x1 * x2 * (y1 - y2);
            const float m10 = x3 * y3 - x2 * y2, m11 = x1 * y1 - x3 * y3, m12 = x2 * y2 - x1 * y1;
            const float m20 = x3 - x2, m21 = x1 - x3, m22 = x2 - x1;
            const float coef0 = m00 * y1 + m01 * y2 + m02 * y3, coef1 = m10 * y1 + m11 * y2 + m12 * y3, coef2 = m20 * y1 + m21 * y2 + m22 * y3;
            const float kk = 1.0f / (x3 * y3 * (x1 - x2) + x2 * y2 * (x3 - x1) + x1 * y1 * (x2 - x3));
            const float c1 = kk * coef0, c2 = kk * coef1, c3 = kk * coef2;
            const float xn = 0.2627f * c[0] + 0.6780f * c[1] + 0.0593f * c[2];
            const float yn = (c1 + c2 * xn) / (1.0f + c3 * xn);
            const float g = (xn > 0.0f) ? (yn / xn) : 1.0f;
            for (int i = 0; i < 3; i++) c[i] = c[i] * g;
        }
        for (int i = 0; i < 3; i++) c[i] = orc_linear_to_st2084(c[i], 10000.0f);
        return;
    }
    const float base = fmaxf(k.display_max, k.max_m);                                       /* :299-306 */
    const float eff = fminf(base, k.max_cll);
    const float fall = fminf(base / k.max_fall, 1.0f);
    for (int i = 0; i < 3; i++) {
        float v = c[i] * (1.0f / eff);
        v = saturatef(v);
        v = v * fall;
        if (k.selection == 2) v = v / (1.0f + v);                                                            /* Reinhard :49-52 */
        else if (k.selection == 3) {                                                                         /* Habel :54-58 */
            const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
            v = ((v * (A * v + C * B) + D * E) / (v * (A * v + B) + D * F)) - E / F;
        } else if (k.selection == 4) v = v / (1.0f + v / (k.display_max + 1e-6f));                          /* Moebius :60-65 */
        else v = (v * (2.51f * v + 0.03f)) / (v * (2.43f * v + 0.59f) + 0.14f);                              /* ACES :34-47 */
        v = v * k.display_max;
        c[i] = orc_linear_to_st2084(v, 10000.0f);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* resize weights                                                                             */
/* ------------------------------------------------------------------------------------------ */
#define HLSL_PI 3.14159265358979323846f   /* acos(-1.) folded to fp32 */

int orc_upscale_weights(int method, float t, float w[6])
{
    switch (method) {
    case ORC_UP_MITCHELL: {                         /* ps_interpolation_spline4.hlsl:46-51 */
        float t2 = t * t, t3 = t * t2;
        const float a[4] = {1.f / 18.f, 16.f / 18.f, 1.f / 18.f, 0.f / 18.f};
        const float b[4] = {-.5f, 0.f, .5f, 0.f};
        const float c[4] = {5.f / 6.f, -12.f / 6.f, 9.f / 6.f, -2.f / 6.f};
        const float d[4] = {-7.f / 18.f, 21.f / 18.f, -21.f / 18.f, 7.f / 18.f};
        for (int i = 0; i < 4; i++) w[i] = a[i] + b[i] * t + c[i] * t2 + d[i] * t3;
        return 4;
    }
    case ORC_UP_CATMULLROM: {                       /* ps_interpolation_spline4.hlsl:52-54 */
        float t2 = t * t, t3 = t * t2;
        const float b[4] = {-.5f, 0.f, .5f, 0.f};
        const float c[4] = {1.f, -2.5f, 2.f, -.5f};
        const float d[4] = {-.5f, 1.5f, -1.5f, .5f};
        for (int i = 0; i < 4; i++) w[i] = b[i] * t + c[i] * t2 + d[i] * t3;
        w[1] += 1.f;
        return 4;
    }
    case ORC_UP_LANCZOS2: {                         /* ps_interpolation_lanczos2.hlsl:31-56 */
        if (t == 0.0f) { w[0] = 0; w[1] = 1; w[2] = 0; w[3] = 0; return 4; }   /* "return Q1" */
        float ws[4] = {1.f + t, 0.f + t, 1.f - t, 2.f - t};
        float s = 0;
        for (int i = 0; i < 4; i++) {
            float a = ws[i] * HLSL_PI;
            w[i] = crm_sinf(a) * crm_sinf(a * .5f) / (ws[i] * ws[i] * HLSL_PI * HLSL_PI * .5f);
        }
        s = w[0] + w[1] + w[2] + w[3];              /* dot(1., w) */
        float wc = 1.f - s;
        w[1] += wc * (1.f - t);
        w[2] += wc * t;
        return 4;
    }
    case ORC_UP_LANCZOS3: {                         /* ps_interpolation_lanczos3.hlsl:31-64 */
        if (t == 0.0f) { w[0] = w[1] = 0; w[2] = 1; w[3] = w[4] = w[5] = 0; return 6; } /* "return Q2" */
        const float k0[3] = {2.f, 1.f, 0.f}, k1[3] = {1.f, 2.f, 3.f};
        float w0[3], w1[3];
        for (int i = 0; i < 3; i++) {
            float a0 = k0[i] * HLSL_PI + t * HLSL_PI, a1 = k1[i] * HLSL_PI - t * HLSL_PI;
            float a0s = a0 * .5f, a1s = a1 * .5f;
            w0[i] = crm_sinf(a0) * crm_sinf(a0s) / (a0 * a0s);
            w1[i] = crm_sinf(a1) * crm_sinf(a1s) / (a1 * a1s);
        }
        float s = (w0[0] + w1[0]) + (w0[1] + w1[1]) + (w0[2] + w1[2]);   /* dot(1., w0 + w1) */
        float wc = 1.f - s;
        w0[2] += wc * (1.f - t);
        w1[0] += wc * t;
        w[0] = w0[0]; w[1] = w0[1]; w[2] = w0[2]; w[3] = w1[0]; w[4] = w1[1]; w[5] = w1[2];
        return 6;
    }
    case ORC_UP_SPLINE36_EXT: {
        /* EXTENSION, no reference counterpart (the reference's "spline" is Mitchell): the classic three-piece cubic Spline36 kernel
         * over taps base-2 .. base+3 (distances 2+t, 1+t, t, 1-t, 2-t, 3-t), weights divided by their sum; t == 0 returns the
         * centre tap like the Lanczos shaders do */
        if (t == 0.0f) { w[0] = w[1] = 0; w[2] = 1; w[3] = w[4] = w[5] = 0; return 6; }
        const float d[6] = {2.f + t, 1.f + t, t, 1.f - t, 2.f - t, 3.f - t};
        float s = 0;
        for (int i = 0; i < 6; i++) {
            float x = d[i];
            if (x < 1.f) w[i] = ((13.f / 11.f * x - 453.f / 209.f) * x - 3.f / 209.f) * x + 1.f;
            else if (x < 2.f) { x -= 1.f; w[i] = ((-6.f / 11.f * x + 270.f / 209.f) * x - 156.f / 209.f) * x; }
            else { x -= 2.f; w[i] = ((1.f / 11.f * x - 45.f / 209.f) * x + 26.f / 209.f) * x; }
            s += w[i];
        }
        for (int i = 0; i < 6; i++) w[i] /= s;
        return 6;
    }
    default: return 0;
    }
}

/* Shaders/resize/convolution_filters.hlsl:7-86 ; FILTER/A per compile_shaders.cmd:92-103 */
float orc_downscale_filter(int method, float x, float *support)
{
    switch (method) {
    case ORC_DOWN_BOX:
        if (support) *support = 0.5f;
        return (x >= -0.5f && x < 0.5f) ? 1.0f : 0.0f;
    case ORC_DOWN_BILINEAR:
        if (support) *support = 1.0f;
        if (x < 0.0f) x = -x;
        return (x < 1.0f) ? 1.0f - x : 0.0f;
    case ORC_DOWN_HAMMING:
        if (support) *support = 1.0f;
        if (x < 0.0f) x = -x;
        if (x == 0.0f) return 1.0f;
        if (x >= 1.0f) return 0.0f;
        x *= HLSL_PI;
        return crm_sinf(x) / x * (0.54f + 0.46f * crm_cosf(x));
    case ORC_DOWN_BICUBIC:
    case ORC_DOWN_BICUBIC_SHARP: {
        const float A = (method == ORC_DOWN_BICUBIC) ? -0.5f : -1.5f;
        if (support) *support = 2.0f;
        if (x < 0.0f) x = -x;
        if (x < 1.0f) return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1;
        if (x < 2.0f) return (((x - 5) * x + 8) * x - 4) * A;
        return 0.0f;
    }
    case ORC_DOWN_LANCZOS: {
        if (support) *support = 3.0f;
        if (-3.0f <= x && x < 3.0f) {
            float a = x, b = x / 3;
            float sa = (a == 0.0f) ? 1.0f : crm_sinf(a * HLSL_PI) / (a * HLSL_PI);
            float sb = (b == 0.0f) ? 1.0f : crm_sinf(b * HLSL_PI) / (b * HLSL_PI);
            return sa * sb;
        }
        return 0.0f;
    }
    default:
        if (support) *support = 0;
        return 0;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* images and store formats                                                                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int w, h; float *p; } img_t;   /* RGBA fp32, values as a later texture read returns them */

/* The intermediates of one frame (m_TexConvertOutput, m_TexResize, m_TexsPostScale as fp32 RGBA: 1 GB at 4K -> 8K) come out of a small
   pool instead of malloc / free per frame: glibc maps and unmaps anything above 32 MiB, and the page faults of a fresh gigabyte per
   frame cost a 128-thread host a quarter of its frame time (bench.py's cpu_baseline: 2.0 -> 2.8 frames/s; the arithmetic is unchanged).
   Every image is fully written by the pass that produces it before anything reads it, so a reused buffer needs no clearing. */
#define ORC_POOL_N 8
static struct { float *p; size_t bytes; int busy; } g_pool[ORC_POOL_N];
static int img_alloc(img_t *im, int w, int h)
{
    const size_t need = (size_t)w * h * 4 * sizeof(float);
    int best = -1, idle = -1;
    im->w = w; im->h = h; im->p = NULL;
#pragma omp critical(orc_pool)
    {
        for (int i = 0; i < ORC_POOL_N; i++) {
            if (g_pool[i].busy) continue;
            if (g_pool[i].p && g_pool[i].bytes >= need && (best < 0 || g_pool[i].bytes < g_pool[best].bytes)) best = i;
            if (idle < 0 || !g_pool[i].p || (g_pool[idle].p && g_pool[i].bytes < g_pool[idle].bytes)) idle = i;
        }
        if (best < 0 && idle >= 0) {            /* nothing fits: replace the smallest idle buffer (or fill an empty slot) */
            free(g_pool[idle].p);
            g_pool[idle].p = (float *)malloc(need);
            g_pool[idle].bytes = g_pool[idle].p ? need : 0;
            if (g_pool[idle].p) best = idle;
        }
        if (best >= 0) { g_pool[best].busy = 1; im->p = g_pool[best].p; }
    }
    if (!im->p) im->p = (float *)malloc(need);  /* every slot busy (concurrent callers): a plain allocation */
    return im->p ? 0 : -1;
}
static void img_free(img_t *im)
{
    int pooled = 0;
    if (!im->p) return;
#pragma omp critical(orc_pool)
    for (int i = 0; i < ORC_POOL_N; i++)
        if (g_pool[i].p == im->p) { g_pool[i].busy = 0; pooled = 1; }
    if (!pooled) free(im->p);
    im->p = NULL;
}

enum { FMT_BGRA8 = 8, FMT_RGB10A2 = 10, FMT_RGBA16F = 16 };

/* float -> UNORM n store then UNORM load: floor(sat(x)*(2^n-1)+0.5) / (2^n-1) */
static inline float unorm_round(float x, float maxv)
{
    float q = floorf(saturatef(x) * maxv + 0.5f);
    return q / maxv;
}
/* value a render-target write leaves in a texture of format fmt */
static inline void store_fmt(int fmt, const float in[4], float out[4])
{
    switch (fmt) {
    case FMT_BGRA8:   for (int i = 0; i < 4; i++) out[i] = unorm_round(in[i], 255.0f); break;
    case FMT_RGB10A2: for (int i = 0; i < 3; i++) out[i] = unorm_round(in[i], 1023.0f);
                      out[3] = unorm_round(in[3], 3.0f); break;
    default:          for (int i = 0; i < 4; i++) out[i] = orc_half_round(in[i]); break;
    }
}

/* UpdateTexParams — DX11VideoProcessor.cpp:1143-1155 */
static int internal_format(int iTexFormat, int cdepth)
{
    switch (iTexFormat) {
    case ORC_TEXFMT_8INT: return FMT_BGRA8;
    case ORC_TEXFMT_10INT: return FMT_RGB10A2;
    case ORC_TEXFMT_16FLOAT: return FMT_RGBA16F;
    default: return cdepth > 8 ? FMT_RGB10A2 : FMT_BGRA8;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* convert pass — Shaders.cpp:82-329 (ShaderGetPixels, DX11 branch), :593-930                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const fmt_info *f;
    const uint8_t *plane[3];
    int pitch[3];
    int w, h;          /* luma texture size */
    int cw, ch;        /* chroma texture size */
} src_tex;

/* texel fetch with clamp addressing + UNORM load; CopyPlane10to16 shift applied (Helper.cpp:789-803) */
static inline float load_luma(const src_tex *s, int x, int y)
{
    x = clampi(x, 0, s->w - 1); y = clampi(y, 0, s->h - 1);
    if (s->f->bytes == 1) return (float)s->plane[0][(size_t)y * s->pitch[0] + x] / 255.0f;
    uint16_t v = ((const uint16_t *)(s->plane[0] + (size_t)y * s->pitch[0]))[x];
    v = (uint16_t)(v << s->f->shift);
    return (float)v / 65535.0f;
}
/* c: 0 = U, 1 = V */
static inline float load_chroma(const src_tex *s, int c, int x, int y)
{
    x = clampi(x, 0, s->cw - 1); y = clampi(y, 0, s->ch - 1);
    if (s->f->planes == 2) {
        if (s->f->bytes == 1) return (float)s->plane[1][(size_t)y * s->pitch[1] + 2 * x + c] / 255.0f;
        uint16_t v = ((const uint16_t *)(s->plane[1] + (size_t)y * s->pitch[1]))[2 * x + c];
        return (float)v / 65535.0f;
    }
    int pl = s->f->v_first ? (c == 0 ? 2 : 1) : (c == 0 ? 1 : 2);     /* Shaders.cpp:159-165 */
    if (s->f->bytes == 1) return (float)s->plane[pl][(size_t)y * s->pitch[pl] + x] / 255.0f;
    uint16_t v = ((const uint16_t *)(s->plane[pl] + (size_t)y * s->pitch[pl]))[x];
    v = (uint16_t)(v << s->f->shift);
    return (float)v / 65535.0f;
}

/* one-plane formats: component k of texel (tx,y) of the RGBA8 / RGBA16 / R10G10B10A2 texture, clamp addressing */
static inline float load_packed(const src_tex *s, int tx, int y, int k)
{
    const int tw = (s->f->layout == LAY_PACKED422) ? s->w / 2 : s->w;
    tx = clampi(tx, 0, tw - 1); y = clampi(y, 0, s->h - 1);
    const uint8_t *row = s->plane[0] + (size_t)y * s->pitch[0];
    if (s->f->bits10) {
        uint32_t d = ((const uint32_t *)row)[tx];
        uint32_t v = k == 3 ? (d >> 30) : ((d >> (10 * k)) & 0x3ffu);
        return k == 3 ? (float)v / 3.0f : (float)v / 1023.0f;
    }
    if (s->f->bytes == 1) return (float)row[4 * tx + k] / 255.0f;
    return (float)((const uint16_t *)row)[4 * tx + k] / 65535.0f;
}

/* D3D11 linear sample at unnormalised texel coordinate (u,v) = texcoord*size: taps floor(u-.5),+1 with
 * weights frac(u-.5); all positions on this path are multiples of 1/4 so the 8-bit weight precision
 * of the fixed-function filter is exact. */
static inline float sample_chroma_linear(const src_tex *s, int c, float u, float v)
{
    float fu = u - 0.5f, fv = v - 0.5f;
    float iu = floorf(fu), iv = floorf(fv);
    float wx = fu - iu, wy = fv - iv;
    int x0 = (int)iu, y0 = (int)iv;
    float c00 = load_chroma(s, c, x0, y0),     c10 = load_chroma(s, c, x0 + 1, y0);
    float c01 = load_chroma(s, c, x0, y0 + 1), c11 = load_chroma(s, c, x0 + 1, y0 + 1);
    float top = c00 * (1.0f - wx) + c10 * wx;
    float bot = c01 * (1.0f - wx) + c11 * wx;
    return top * (1.0f - wy) + bot * wy;
}

static void catmull_weights(float t, float w[4])      /* Shaders.cpp:66-72 */
{
    float t2 = t * t, t3 = t * t2;
    w[0] = t2 - (t3 + t) / 2;
    w[1] = t3 * 1.5f + 1 - t2 * 2.5f;
    w[2] = t2 * 2 + t / 2 - t3 * 1.5f;
    w[3] = (t3 - t2) / 2;
}

/* chroma for luma pixel (sx,sy) of the source texture */
static void fetch_chroma(const src_tex *s, int chroma_loc, int chroma_scaling, int sx, int sy, float uv[2])
{
    const int sub = s->f->subsampling;
    if (chroma_scaling == ORC_CHROMA_NEAREST || sub == 444) {            /* Shaders.cpp:239-241,282-287 */
        int cx = sx / s->f->div_w, cy = sy / s->f->div_h;                 /* floor((sx+.5)/div) */
        uv[0] = load_chroma(s, 0, cx, cy); uv[1] = load_chroma(s, 1, cx, cy);
        return;
    }
    if (chroma_scaling == ORC_CHROMA_CATMULLROM && sub == 420) {         /* :242-251,288-299 */
        /* t = frac(Tex*(wh*0.5)) + off ; Tex*(wh/2) = ((sx+.5)/2, (sy+.5)/2) */
        float tx = (sx & 1) ? 0.75f : 0.25f, ty = (sy & 1) ? 0.75f : 0.25f;
        switch (chroma_loc) {                                            /* strChromaPos2 :121-137 */
        case CHROMA_COSITED: tx += -0.25f; ty += -0.25f; break;
        case CHROMA_MPEG1:   tx += -0.5f;  ty += -0.5f;  break;
        default:             tx += -0.25f; ty += -0.5f;  break;
        }
        float wx[4], wy[4]; catmull_weights(tx, wx); catmull_weights(ty, wy);
        int bx = sx >> 1, by = sy >> 1;
        for (int c = 0; c < 2; c++) {
            float Q[4];
            for (int y = 0; y < 4; y++) {
                float c0 = load_chroma(s, c, bx - 1, by + y - 1), c1 = load_chroma(s, c, bx, by + y - 1);
                float c2 = load_chroma(s, c, bx + 1, by + y - 1), c3 = load_chroma(s, c, bx + 2, by + y - 1);
                Q[y] = c0 * wx[0] + c1 * wx[1] + c2 * wx[2] + c3 * wx[3];   /* code_Bicubic_UV :74-79 */
            }
            uv[c] = Q[0] * wy[0] + Q[1] * wy[1] + Q[2] * wy[2] + Q[3] * wy[3];
        }
        return;
    }
    if (chroma_scaling == ORC_CHROMA_CATMULLROM && sub == 422) {         /* :252-264,300-318 */
        if ((sx & 1) == 0) {                                             /* fmod(Tex.x*w,2) < 1 */
            uv[0] = load_chroma(s, 0, sx >> 1, sy); uv[1] = load_chroma(s, 1, sx >> 1, sy);
        } else {
            int k = (sx - 1) >> 1;
            for (int c = 0; c < 2; c++) {
                float c0 = load_chroma(s, c, k - 1, sy), c1 = load_chroma(s, c, k, sy);
                float c2 = load_chroma(s, c, k + 1, sy), c3 = load_chroma(s, c, k + 2, sy);
                uv[c] = (9 * (c1 + c2) - (c0 + c3)) * 0.0625f;          /* CATMULLROM_05 :145 */
            }
        }
        return;
    }
    /* CHROMA_Bilinear :265-270,319-325 : texUV.Sample(sampL, Tex + strChromaPos) */
    float u = (sx + 0.5f) / (float)s->f->div_w, v = (sy + 0.5f) / (float)s->f->div_h;   /* Tex * chroma size */
    if (sub == 420) {
        switch (chroma_loc) {                                            /* strChromaPos :121-137 */
        case CHROMA_COSITED: u += 0.25f; v += 0.25f; break;              /* +(dx/2, dy/2) in chroma texels */
        case CHROMA_MPEG1:   break;
        default:             u += 0.25f; break;                          /* +(dx/2, 0) */
        }
    } else {                                                             /* 422 :139-142 */
        u += 0.25f;
    }
    uv[0] = sample_chroma_linear(s, 0, u, v);
    uv[1] = sample_chroma_linear(s, 1, u, v);
}

/* (Y,U,V) — or (G,B,R) / (Y,0,0) — of source pixel (sx,sy): ShaderGetPixels, DX11 branch */
static void fetch_pixel(const src_tex *s, int chroma_loc, int chroma_scaling, int blend_deint, int sx, int sy, float yuv[3])
{
    const fmt_info *f = s->f;
    if (f->layout == LAY_PLANAR) {
        yuv[0] = load_luma(s, sx, sy);                                    /* :231,274 */
        if (blend_deint && f->subsampling == 420) {                       /* blendDeint420 :115,232-237,275-280 */
            float y1 = load_luma(s, sx, sy - 1), y2 = load_luma(s, sx, sy + 1);
            yuv[0] = (yuv[0] * 2 + y1 + y2) / 4;
        }
        fetch_chroma(s, chroma_loc, chroma_scaling, sx, sy, yuv + 1);
        return;
    }
    if (f->layout == LAY_GRAY) {           /* float4 color = tex.Sample(samp, Tex) of an R8/R16 texture (:184) */
        yuv[0] = load_luma(s, sx, sy); yuv[1] = 0; yuv[2] = 0;
        return;
    }
    if (f->layout == LAY_RGB) {            /* float4 color = tex.Sample(samp, Tex) (:184): (R,G,B) of the texture */
        yuv[0] = load_packed(s, sx, sy, f->ci[0]); yuv[1] = load_packed(s, sx, sy, f->ci[1]); yuv[2] = load_packed(s, sx, sy, f->ci[2]);
        return;
    }
    if (f->layout == LAY_PACKED444) {      /* .zyxw (AYUV) / .yxzw (Y410, Y416) (:186-193) */
        yuv[0] = load_packed(s, sx, sy, f->ci[0]); yuv[1] = load_packed(s, sx, sy, f->ci[1]); yuv[2] = load_packed(s, sx, sy, f->ci[2]);
        return;
    }
    /* packed 4:2:2 (:195-229): texel tx = two pixels; even pixel takes the texel's own chroma, the odd pixel the
       mean with the next texel (or CATMULLROM_05 over texels tx-1..tx+2); Nearest is not distinguished */
    const int tx = sx >> 1;
    const int cu = f->ci[1], cv = f->ci[3];
    if ((sx & 1) == 0) {                   /* fmod(Tex.x*w, 2) < 1.0 */
        yuv[0] = load_packed(s, tx, sy, f->ci[0]); yuv[1] = load_packed(s, tx, sy, cu); yuv[2] = load_packed(s, tx, sy, cv);
        return;
    }
    yuv[0] = load_packed(s, tx, sy, f->ci[2]);
    for (int c = 0; c < 2; c++) {
        const int k = c ? cv : cu;
        if (chroma_scaling == ORC_CHROMA_CATMULLROM) {
            float c0 = load_packed(s, tx - 1, sy, k), c1 = load_packed(s, tx, sy, k);
            float c2 = load_packed(s, tx + 1, sy, k), c3 = load_packed(s, tx + 2, sy, k);
            yuv[1 + c] = (9 * (c1 + c2) - (c0 + c3)) * 0.0625f;          /* CATMULLROM_05 :145 */
        } else {
            yuv[1 + c] = (load_packed(s, tx, sy, k) + load_packed(s, tx + 1, sy, k)) * 0.5f;
        }
    }
}

/* CopyFrameV210 — Helper.cpp:709-748: v210 dwords -> Y210 words (10 bits in the MSBs), two dwords at a time */
/* The copy loops below restate the reference's (Helper.cpp), which read and write rows through uint32_t / uint64_t pointers
 * whatever the row's alignment (fine on x86, undefined in ISO C).  Same loops, through alignment-1 may-alias types: UBSan-clean
 * (`make sanitize`). */
typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;
typedef uint32_t __attribute__((aligned(1), may_alias)) u32u;
typedef uint16_t __attribute__((aligned(1), may_alias)) u16u;
void orc_repack_v210(int lines, uint8_t *dst, int dst_pitch, const uint8_t *src, int src_pitch)
{
    const int dq = dst_pitch / 12, dr = dst_pitch % 12, sq = src_pitch / 8, sr = src_pitch % 8;
    int line_blocks, remainder;
    if (dq <= sq) { line_blocks = dq; remainder = dr != 0; } else { line_blocks = sq; remainder = sr != 0; }
    for (int y = 0; y < lines; y++) {
        const uint32_t *src32 = (const uint32_t *)(src + (size_t)y * src_pitch);
        uint16_t *dst16 = (uint16_t *)(dst + (size_t)y * dst_pitch);
        for (int i = 0; i < line_blocks; i++) {
            uint32_t s0 = *src32++, s1 = *src32++;
            *dst16++ = (uint16_t)((s0 >> 4) & 0xffc0);
            *dst16++ = (uint16_t)((s0 << 6) & 0xffc0);
            *dst16++ = (uint16_t)((s1 << 6) & 0xffc0);
            *dst16++ = (uint16_t)((s0 >> 14) & 0xffc0);
            *dst16++ = (uint16_t)((s1 >> 14) & 0xffc0);
            *dst16++ = (uint16_t)((s1 >> 4) & 0xffc0);
        }
        if (remainder) {
            uint32_t v = *src32++;
            *dst16++ = (uint16_t)((v >> 4) & 0xffc0);
            *dst16++ = (uint16_t)((v << 6) & 0xffc0);
        }
    }
}
/* CopyPlaneAsIs / CopyFrameRGB24 / CopyFrameR210 / CopyFrameRGB48 / CopyFrameBGR48 / CopyFrameBGRA64 / CopyFrameB64A —
 * Helper.cpp:414-428,444-482,548-566,600-707,770-787 restated line by line (src_pitch may be negative: bottom-up DIB) */
void orc_repack_rgb(int kind, int lines, uint8_t *dst, int dst_pitch, const uint8_t *src, int src_pitch)
{
    const int ap = src_pitch < 0 ? -src_pitch : src_pitch;
    for (int y = 0; y < lines; y++, src += src_pitch, dst += dst_pitch) {
        if (kind == RPK_NONE) {
            memcpy(dst, src, (size_t)(ap < dst_pitch ? ap : dst_pitch));
        } else if (kind == RPK_RGB24) {
            const unsigned line_pixels = (unsigned)ap / 3, line_pixels4 = line_pixels & ~3u;
            const u32u *src32 = (const u32u *)src; u32u *dst32 = (u32u *)dst;
            unsigned i = 0;
            for (; i < line_pixels4; i += 4) {
                uint32_t sa = *src32++, sb = *src32++, sc = *src32++;
                *dst32++ = sa; *dst32++ = (sa >> 24) | (sb << 8); *dst32++ = (sb >> 16) | (sc << 16); *dst32++ = sc >> 8;
            }
            if (i < line_pixels) {
                if (line_pixels & 1) { *dst32 = *src32; }
                else { uint32_t sa = *src32++, sb = *src32; *dst32++ = sa; *dst32 = (sa >> 24) | (sb << 8); }
            }
        } else if (kind == RPK_R210) {
            const unsigned line_pixels = (unsigned)ap / 4;
            const u32u *src32 = (const u32u *)src; u32u *dst32 = (u32u *)dst;
            for (unsigned i = 0; i < line_pixels; i++) {
                const uint32_t t = src32[i];
                uint32_t r = ((t & 0x0000003f) << 4) | ((t & 0x0000f000) >> 12);
                uint32_t g = ((t & 0x00fc0000) >> 8) | ((t & 0x00000f00) << 8);
                uint32_t b = ((t & 0xff000000) >> 4) | ((t & 0x00030000) << 12);
                dst32[i] = r | g | b;
            }
        } else if (kind == RPK_RGB48) {
            const unsigned line_pixels = (unsigned)ap / 6, line_pixels4 = line_pixels & ~3u;
            const u64u *src64 = (const u64u *)src; u64u *dst64 = (u64u *)dst;
            for (unsigned i = 0; i < line_pixels4; i += 4) {      /* no remainder handling, as written (:552-563) */
                uint64_t sa = src64[0], sb = src64[1], sc = src64[2];
                dst64[i + 0] = sa; dst64[i + 1] = (sa >> 48) | (sb << 16); dst64[i + 2] = (sb >> 32) | (sc << 32); dst64[i + 3] = sc >> 16;
                src64 += 3;
            }
        } else if (kind == RPK_BGR48) {
            const unsigned line_pixels = (unsigned)ap / 6, line_pixels4 = line_pixels & ~3u;
            const u64u *src64 = (const u64u *)src; u64u *dst64 = (u64u *)dst;
            unsigned i = 0;
            for (; i < line_pixels4; i += 4) {
                uint64_t sa = *src64++, sb = *src64++, sc = *src64++;
                *dst64++ = ((sa & 0xffff) << 32) | (sa & 0xffff0000) | ((sa & 0xffff00000000) >> 32);
                *dst64++ = ((sa & 0xffff000000000000) >> 16) | ((sb & 0xffff) << 16) | ((sb & 0xffff0000) >> 16);
                *dst64++ = (sb & 0xffff00000000) | ((sb & 0xffff000000000000) >> 32) | (sc & 0xffff);
                *dst64++ = ((sc & 0xffff0000) << 16) | ((sc & 0xffff00000000) >> 16) | ((sc & 0xffff000000000000) >> 48);
            }
            const unsigned remainder = line_pixels - i;
            if (remainder) {
                uint64_t sa = *src64++;
                *dst64++ = ((sa & 0xffff) << 32) | (sa & 0xffff0000) | ((sa & 0xffff00000000) >> 32);
                if (remainder == 2) {
                    uint64_t sb = *(const u32u *)src64;
                    *dst64 = ((sa & 0xffff000000000000) >> 16) | ((sb & 0xffff) << 16) | ((sb & 0xffff0000) >> 16);
                } else if (remainder == 3) {
                    uint64_t sb = *src64++;
                    uint64_t sc = *(const u32u *)src64;
                    *dst64++ = ((sa & 0xffff000000000000) >> 16) | ((sb & 0xffff) << 16) | ((sb & 0xffff0000) >> 16);
                    *dst64 = (sb & 0xffff00000000) | ((sb & 0xffff000000000000) >> 32) | (sc & 0xffff);
                }
            }
        } else if (kind == RPK_BGRA64) {
            const unsigned line_pixels = (unsigned)ap / 8;
            const u64u *src64 = (const u64u *)src; u64u *dst64 = (u64u *)dst;
            for (unsigned i = 0; i < line_pixels; i++)
                dst64[i] = ((src64[i] & 0x000000000000ffffULL) << 32) | ((src64[i] & 0x0000ffff00000000ULL) >> 32) | (src64[i] & 0xffff0000ffff0000ULL);
        } else if (kind == RPK_B64A) {
            const unsigned line_pixels = (unsigned)ap / 8;
            const u64u *src64 = (const u64u *)src; u64u *dst64 = (u64u *)dst;
            for (unsigned i = 0; i < line_pixels; i++)
                dst64[i] = ((src64[i] & 0xFF00FF00FF000000ULL) >> 24) + ((src64[i] & 0x00FF00FF00FF0000ULL) >> 8) +
                           ((src64[i] & 0x000000000000FF00ULL) << 40) + ((src64[i] & 0x00000000000000FFULL) << 56);
        }
    }
}

/* pitch of the Y210 texture the v210 sample is unpacked into.  The reference uses the driver's mapped pitch of a
   (W/2) x H R16G16B16A16 texture (>= 4W bytes, typically 256-aligned); the stand-in here is 4W rounded up to whole
   12-byte groups, so every pixel of the row is converted exactly as with any larger driver pitch. */
int orc_v210_tex_pitch(int width) { return (4 * width + 11) / 12 * 12; }

typedef struct {
    src_tex tex;
    void *owned;       /* unpacked v210 / the RGB texture the sample was copied into */
    int enable;        /* m_PSConvColorData.bEnable (DX11VideoProcessor.cpp:849-853) */
    int rect[4];
    uint32_t exfmt;
    float cm[12];
    float lum_scale;
    int internal_fmt;
    /* m_Dovi (valid when p->dovi) */
    orc_dovi_cb dovi_cb[3];
    int dovi_has_mmr;
    dovi_tail_t dovi_tail;
} convert_ctx;

static int setup_convert(const orc_params *p, const uint8_t *src, int src_pitch, convert_ctx *c)
{
    const fmt_info *f = find_fmt(p->cformat);
    if (!f || src_pitch == 0 || (src_pitch < 0 && f->layout != LAY_RGB)) return -1;
    if ((f->div_w == 2 && (p->width & 1)) || (f->div_h == 2 && (p->height & 1))) return -2;
    memset(c, 0, sizeof(*c));
    resolve_rect(p, c->rect);
    if (c->rect[0] < 0 || c->rect[1] < 0 || c->rect[2] > p->width || c->rect[3] > p->height ||
        c->rect[2] <= c->rect[0] || c->rect[3] <= c->rect[1]) return -3;
    if (p->cformat == ORC_CF_V210) {                       /* GetCopyPlaneFunction -> CopyFrameV210 (Helper.cpp:379-380) */
        const int tp = orc_v210_tex_pitch(p->width);
        uint8_t *t = (uint8_t *)calloc((size_t)tp * p->height + 16, 1);
        if (!t) return -4;
        orc_repack_v210(p->height, t, tp, src, src_pitch);
        c->owned = t; src = t; src_pitch = tp;
    }
    if (f->layout == LAY_RGB) {
        /* MemCopyToTexSrcVideo :1243-1248: a bottom-up DIB (negative pitch) is walked from its last row */
        /* texture row: `width` texels; widened when the sample's pitch makes the reference loops copy more pixels per
           row than that (they land in the padding of the mapped row) */
        const int tbpp = f->bits10 ? 4 : 4 * f->bytes;
        const int apitch = src_pitch < 0 ? -src_pitch : src_pitch;
        const int row_px = apitch / f->packsize + 1 > p->width ? apitch / f->packsize + 1 : p->width;
        const int tp = row_px * tbpp;
        uint8_t *t = (uint8_t *)calloc((size_t)tp * p->height + 16, 1);
        if (!t) return -4;
        const uint8_t *s0 = (src_pitch < 0) ? src + (ptrdiff_t)src_pitch * (1 - p->height) : src;
        orc_repack_rgb(f->repack, p->height, t, tp, s0, src_pitch);
        c->owned = t; src = t; src_pitch = tp;
    }
    c->tex.f = f; c->tex.w = p->width; c->tex.h = p->height;
    c->tex.cw = p->width / f->div_w; c->tex.ch = p->height / f->div_h;
    /* MemCopyToTexSrcVideo plane walk — DX11VideoProcessor.cpp:1213-1252 */
    c->tex.plane[0] = src; c->tex.pitch[0] = src_pitch;
    int cpitch = (f->planes == 3) ? src_pitch / f->div_w : src_pitch;
    c->tex.plane[1] = src + (size_t)src_pitch * p->height; c->tex.pitch[1] = cpitch;
    c->tex.plane[2] = c->tex.plane[1] + (size_t)cpitch * c->tex.ch; c->tex.pitch[2] = cpitch;
    c->exfmt = orc_specify_extfmt(p->exfmt, p->cformat, c->rect[2] - c->rect[0], c->rect[3] - c->rect[1]);
    if (orc_color_matrix(p, c->cm)) { free(c->owned); c->owned = NULL; return -1; }
    c->lum_scale = orc_luminance_scale(p->iSDRDisplayNits);
    c->internal_fmt = internal_format(p->iTexFormat, f->cdepth);
    /* :849-853 — interleaved RGB skips the convert draw unless brightness / contrast are set */
    c->enable = f->cstype == CST_YUV || (f->cstype == CST_RGB && f->planes == 3) || f->cstype == CST_GRAY ||
                fabsf(p->brightness / 255) > 1e-4f || fabsf(p->contrast - 1.0f) > 1e-4f;
    if (p->dovi) {
        c->enable = 1;                                                   /* :834 */
        orc_dovi_pack_curves(p->dovi, c->dovi_cb, &c->dovi_has_mmr);
        c->dovi_tail.active = 1;
        orc_dovi_lms_matrix(p->dovi, c->dovi_tail.lms);
        c->dovi_tail.l2_enabled = orc_dovi_l2_constants(p->dovi, (int)p->hdr_display_max_nits, c->dovi_tail.k);
    }
    return 0;
}

static void convert_pass(const orc_params *p, const convert_ctx *c, img_t *out)
{
    const int rw = c->rect[2] - c->rect[0], rh = c->rect[3] - c->rect[1];
    const int trc = EXF_TRC(c->exfmt), prim = EXF_PRIM(c->exfmt), cloc = EXF_CHROMA(c->exfmt);
    const float *cm = c->cm;
    ORC_PAR_FOR
    for (int j = 0; j < rh; j++) {
        for (int i = 0; i < rw; i++) {
            int sx = c->rect[0] + i, sy = c->rect[1] + j;
            float yuv[3];
            fetch_pixel(&c->tex, cloc, p->iChromaScaling, p->blend_deint, sx, sy, yuv);
            if (p->dovi) orc_dovi_reshape(c->dovi_cb, c->dovi_has_mmr, yuv);       /* Shaders.cpp:786-792 */
            const float y = yuv[0], *uv = yuv + 1;
            /* color.rgb = float3(mul(cm_r,color), mul(cm_g,color), mul(cm_b,color)) + cm_c  (:820) */
            float rgb[3];
            rgb[0] = (cm[0] * y + cm[1] * uv[0] + cm[2] * uv[1]) + cm[9];
            rgb[1] = (cm[3] * y + cm[4] * uv[0] + cm[5] * uv[1]) + cm[10];
            rgb[2] = (cm[6] * y + cm[7] * uv[0] + cm[8] * uv[1]) + cm[11];
            hdr_tail_full(rgb, trc, prim, p->bConvertToSdr, c->lum_scale, p->hdr_output, p->dovi ? &c->dovi_tail : NULL);
            float px[4] = {rgb[0], rgb[1], rgb[2], 1.0f};
            store_fmt(c->internal_fmt, px, out->p + ((size_t)j * rw + i) * 4);
        }
    }
}

int orc_convert_only(const orc_params *p, const uint8_t *src, int src_pitch, float *rgba_out)
{
    convert_ctx c;
    int rc = setup_convert(p, src, src_pitch, &c);
    if (rc) return rc;
    img_t out = {c.rect[2] - c.rect[0], c.rect[3] - c.rect[1], rgba_out};
    convert_pass(p, &c, &out);
    free(c.owned);
    return c.internal_fmt;
}

/* ------------------------------------------------------------------------------------------ */
/* resize passes — DX11VideoProcessor.cpp:3103-3187, :332-377 + Shaders/d3d11/ps_interpolation_*, */
/* ps_convolution.hlsl                                                                        */
/* ------------------------------------------------------------------------------------------ */
enum { RS_NONE = 0, RS_UP, RS_DOWN };
typedef struct { int kind; int method; } resizer_t;

/* one output index along the filtered axis: tap indices (already clamped) + weights */
#define ORC_MAX_TAPS 128
typedef struct { int n; int idx[ORC_MAX_TAPS]; float w[ORC_MAX_TAPS]; float wsum; int normalise; } taps_t;

/* Tex[AXIS]*wh[AXIS] for output i of n_out, as the shaders see it.  FillVertices (DX11VideoProcessor.cpp:133-138) puts
 * fp32 texture coordinates on the quad's corners: src_dx = 1.0f / texLen, src_l = src_dx * rect.left, src_r = src_dx *
 * rect.right — three fp32 roundings that are the reference's own.  The rasteriser interpolates TEXCOORD linearly to the
 * pixel centre (i + .5) / n_out of the viewport (modelled as exact, rounded ONCE to fp32: the Direct3D functional spec
 * leaves the interpolator's precision to the hardware) and the shader multiplies by its constant wh[AXIS] = (float)texLen
 * in fp32 (ps_interpolation_*.hlsl:25, ps_convolution.hlsl:30).  `rev`: the coordinate runs from the far edge of the
 * source range to the near one (rotation / flip permute the corners, :140-169).
 * Until round 3 this was `src_l + (i + .5f) * srcLen / dstLen` in fp32 — the same number in real arithmetic, without the
 * roundings above; at 3840 -> 7680 the two differ by up to 2^-13 texels, which moved 0.05 % of the 8-bit channels by one
 * code against the reference shader text. */
/* Sensitivity probe (tests only, orc_set_tex_ulp_bias): the interpolated TEXCOORD arrives `bias` units in the last place off what the
 * model above gives.  A real rasteriser interpolates in fixed-point barycentrics and will differ from the model by about that much, so
 * "bit-identical to the reference's shader text" always means UNDER THIS MODELLED INTERPOLATOR; tests/test_oracle_pins.py uses the
 * probe to show which channels hang on the last ulp (texel centres hit exactly at 3:1, the box filter's x < 0.5 edge) and that every
 * other channel moves by at most one code, rarely. */
static int g_tex_ulp_bias = 0;
void orc_set_tex_ulp_bias(int bias) { g_tex_ulp_bias = bias; }

static inline float axis_center(int src_l, int src_len, int tex_len, int i, int n_out, int rev)
{
    const float src_d = 1.0f / (float)tex_len;
    const float c_lo = src_d * (float)src_l, c_hi = src_d * (float)(src_l + src_len);
    const double ua = rev ? c_hi : c_lo, ub = rev ? c_lo : c_hi;
    const double a = ((double)i + 0.5) / (double)n_out;
    float tex = (float)(ua + (ub - ua) * a);
    for (int k = 0; k < g_tex_ulp_bias; k++) tex = nextafterf(tex, 3.0e38f);
    for (int k = 0; k > g_tex_ulp_bias; k--) tex = nextafterf(tex, -3.0e38f);
    return tex * (float)tex_len;
}

/* taps of one output texel whose interpolated coordinate on the filtered axis is `center` (= Tex[AXIS]*wh[AXIS]);
 * `scale` = the shader constant scale[AXIS] (only ps_convolution reads it) */
static int build_taps_at(resizer_t rs, float center, float scale, int tex_len, uint32_t flags, taps_t *t)
{
    t->normalise = 0; t->wsum = 1;
    if (rs.kind == RS_UP) {
        float pos = center - 0.5f;                       /* ps_interpolation_*.hlsl: pos = Tex*wh - 0.5 */
        float fr = pos - floorf(pos);                    /* frac(pos) */
        pos = pos - fr;
        int base = (int)pos;
        float w[6];
        int n = orc_upscale_weights(rs.method, fr, w);
        if (n == 4) {                                    /* Q0..Q3 at pos-0.5 .. pos+2.5 => texels base-1..base+2 */
            t->n = 4;
            for (int k = 0; k < 4; k++) { t->idx[k] = clampi(base - 1 + k, 0, tex_len - 1); t->w[k] = w[k]; }
        } else if (n == 6) {
            /* D3D11 ps_interpolation_lanczos3.hlsl:33-34,42-43 samples Q1 at pos-1.5 (same texel as Q0);
             * the D3D9 twin uses pos-0.5 (Shaders/d3d9/interpolation_lanczos3.hlsl:28-29). */
            static const int off11[6] = {-2, -2, 0, 1, 2, 3}, off9[6] = {-2, -1, 0, 1, 2, 3};
            const int *off = ((flags & ORC_FLAG_LANCZOS3_FIXED) || rs.method != ORC_UP_LANCZOS3) ? off9 : off11;   /* the quirk is the Lanczos3 shader's */
            t->n = 6;
            for (int k = 0; k < 6; k++) { t->idx[k] = clampi(base + off[k], 0, tex_len - 1); t->w[k] = w[k]; }
        } else return -1;
        return 0;
    }
    if (rs.kind == RS_DOWN) {                            /* ps_convolution.hlsl:23-50 */
        float support0; (void)orc_downscale_filter(rs.method, 0.0f, &support0);
        float support = support0 * scale;
        float ss = 1.0f / scale;
        float pos = center + 0.5f;
        int low = (int)floorf(pos - support);
        int high = (int)ceilf(pos + support);
        if (high - low > ORC_MAX_TAPS) return -1;
        float ww = 0.0f; int n = 0;
        for (int k = low; k < high; k++) {
            float w = orc_downscale_filter(rs.method, ((float)k - pos + 0.5f) * ss, NULL);
            ww += w;
            t->idx[n] = clampi(k, 0, tex_len - 1); t->w[n] = w; n++;
        }
        t->n = n; t->wsum = ww; t->normalise = 1;
        return 0;
    }
    /* no shader on this axis: the pass point-samples Tex => nearest */
    t->n = 1; t->idx[0] = clampi((int)floorf(center), 0, tex_len - 1); t->w[0] = 1.0f;
    return 0;
}
static int build_taps(resizer_t rs, int src_l, int src_len, int n_out, int i, float scale, int tex_len, uint32_t flags, taps_t *t)
{
    return build_taps_at(rs, axis_center(src_l, src_len, tex_len, i, n_out, 0), scale, tex_len, flags, t);
}

int orc_axis_taps(int kind, int method, int src_l, int src_len, int n_out, int tex_len, uint32_t flags,
                  int i, int32_t *idx, float *w, float *wsum)
{
    taps_t t;
    resizer_t rs = {kind, method};
    float scale = (float)src_len / (float)n_out;
    if (build_taps(rs, src_l, src_len, n_out, i, scale, tex_len, flags, &t)) return -1;
    for (int k = 0; k < t.n; k++) { idx[k] = t.idx[k]; w[k] = t.w[k]; }
    if (wsum) *wsum = t.wsum;
    return t.n;
}

/* One TextureResizeShader / TextureCopyRect draw of the WHOLE texture `in` onto `out` with the vertex set-up of
 * FillVertices (DX11VideoProcessor.cpp:130-179): rotation 0/90/180/270 (clockwise) and horizontal flip permute which
 * texture coordinate runs along which screen axis and in which direction:
 *     rot   0: U = l + a(r-l)   V = t + b(bm-t)        a = (x+.5)/W', b = (y+.5)/H'
 *     rot  90: U = l + b(r-l)   V = bm - a(bm-t)
 *     rot 180: U = r - a(r-l)   V = bm - b(bm-t)
 *     rot 270: U = r - b(r-l)   V = t + a(bm-t)         flip swaps l and r (:167-169)
 * tex_axis = texture axis the pixel shader filters (0 = X shaders, 1 = Y shaders, -1 = ps_simple); the other
 * coordinate is point-sampled.  The shader constant scale[AXIS] is srcRect/dstRect of the SAME-NAMED screen dimension
 * (:351-354) whatever the rotation, so a rotated ps_convolution draw runs with the other dimension's ratio — as written. */
static int resize_draw(const img_t *in, const int rect[4], img_t *out, int tex_axis, resizer_t rs, int rot, int flip, uint32_t flags, int store)
{
    const int whole[4] = {0, 0, in->w, in->h};
    if (!rect) rect = whole;
    const int rw = rect[2] - rect[0], rh = rect[3] - rect[1];
    const int swap = (rot == 90 || rot == 270);
    /* texture axis and direction along screen x and screen y */
    const int tax = swap ? 1 : 0, tay = swap ? 0 : 1;
    int rev_u = (rot == 180 || rot == 270);               /* U runs against its screen axis */
    const int rev_v = (rot == 90 || rot == 180);
    if (flip) rev_u = !rev_u;
    const int rev_x = tax == 0 ? rev_u : rev_v, rev_y = tay == 0 ? rev_u : rev_v;
    const int len_x = tax == 0 ? rw : rh, len_y = tay == 0 ? rw : rh;               /* srcRect extent run through by x / y */
    const int org_x = tax == 0 ? rect[0] : rect[1], org_y = tay == 0 ? rect[0] : rect[1];
    const int tex_x = tax == 0 ? in->w : in->h, tex_y = tay == 0 ? in->w : in->h;   /* clamp range = whole texture */
    const float cscale = tex_axis == 0 ? (float)rw / (float)out->w : (float)rh / (float)out->h;   /* scale[AXIS] */
    const resizer_t none = {RS_NONE, 0};

    if (rs.kind == RS_UP && rs.method == ORC_UP_JINC2) {
        /* ps_resize_onepass_jinc2.hlsl:44-101 ("Jinc2m", IDF_PS_11_INTERP_JINC2): ONE 2-D draw — 4x4 texels around the
           sample position weighted by the windowed jinc of their distance, normalised, then anti-ringing (clamp towards
           the min/max of the inner 2x2, strength 0.8).  Used for both axes (m_pShaderUpscaleY = m_pShaderUpscaleX, :2921). */
        const float pi = acosf(-1.0f), wa = 0.416f * pi, wb = 0.985f * pi;
        ORC_PAR_FOR
        for (int y = 0; y < out->h; y++) {
            for (int x = 0; x < out->w; x++) {
                const float cx = axis_center(org_x, len_x, tex_x, x, out->w, rev_x);
                const float cy = axis_center(org_y, len_y, tex_y, y, out->h, rev_y);
                const float pcx = tax == 0 ? cx : cy, pcy = tax == 0 ? cy : cx;        /* pc = Tex * wh */
                const float tcx = floorf(pcx - 0.5f) + 0.5f, tcy = floorf(pcy - 0.5f) + 0.5f;
                float w[4][4], wsum = 0.0f;
                const float *c[4][4];
                for (int j = 0; j < 4; j++) {
                    float rowsum = 0.0f;
                    for (int i = 0; i < 4; i++) {
                        const float vx = (tcx + (float)(i - 1)) - pcx, vy = (tcy + (float)(j - 1)) - pcy;
                        const float dd = sqrtf(vx * vx + vy * vy);
                        w[j][i] = (dd == 0.0f) ? wa * wb : crm_sinf(dd * wa) * crm_sinf(dd * wb) / (dd * dd);
                        rowsum = i == 0 ? w[j][i] : rowsum + w[j][i];
                        const int sx = clampi((int)floorf(tcx) + i - 1, 0, in->w - 1), sy = clampi((int)floorf(tcy) + j - 1, 0, in->h - 1);
                        c[j][i] = in->p + ((size_t)sy * in->w + sx) * 4;
                    }
                    wsum = j == 0 ? rowsum : wsum + rowsum;
                }
                float px[4] = {0, 0, 0, 1.0f};
                for (int ch = 0; ch < 3; ch++) {
                    float color = 0.0f;
                    for (int j = 0; j < 4; j++) {
                        float r = w[j][0] * c[j][0][ch];
                        r = r + w[j][1] * c[j][1][ch]; r = r + w[j][2] * c[j][2][ch]; r = r + w[j][3] * c[j][3][ch];
                        color = j == 0 ? r : color + r;
                    }
                    color = color / wsum;
                    const float mn = fminf(fminf(fminf(c[1][1][ch], c[1][2][ch]), c[2][1][ch]), c[2][2][ch]);
                    const float mx = fmaxf(fmaxf(fmaxf(c[1][1][ch], c[1][2][ch]), c[2][1][ch]), c[2][2][ch]);
                    const float cl = fminf(fmaxf(color, mn), mx);
                    px[ch] = color + 0.8f * (cl - color);                               /* lerp(color, clamp(..), 0.8) */
                }
                store_fmt(store, px, out->p + ((size_t)y * out->w + x) * 4);
            }
        }
        return 0;
    }

    taps_t *tx = (taps_t *)malloc(sizeof(taps_t) * (size_t)out->w);
    taps_t *ty = (taps_t *)malloc(sizeof(taps_t) * (size_t)out->h);
    if (!tx || !ty) { free(tx); free(ty); return -1; }
    for (int i = 0; i < out->w; i++) {
        const float c = axis_center(org_x, len_x, tex_x, i, out->w, rev_x);
        if (build_taps_at(tax == tex_axis ? rs : none, c, cscale, tex_x, flags, &tx[i])) { free(tx); free(ty); return -2; }
    }
    for (int i = 0; i < out->h; i++) {
        const float c = axis_center(org_y, len_y, tex_y, i, out->h, rev_y);
        if (build_taps_at(tay == tex_axis ? rs : none, c, cscale, tex_y, flags, &ty[i])) { free(tx); free(ty); return -2; }
    }
    const int filt_x = (tax == tex_axis);      /* taps run along screen x; otherwise along screen y (or nowhere) */
    ORC_PAR_FOR
    for (int y = 0; y < out->h; y++) {
        for (int x = 0; x < out->w; x++) {
            const taps_t *t = filt_x ? &tx[x] : &ty[y];
            const int o = filt_x ? ty[y].idx[0] : tx[x].idx[0];      /* point-sampled coordinate */
            float acc[4] = {0, 0, 0, 0};
            for (int k = 0; k < t->n; k++) {
                /* filtered coordinate is on texture axis (filt_x ? tax : tay) */
                const int fa = filt_x ? tax : tay;
                const int sx = fa == 0 ? t->idx[k] : o, sy = fa == 0 ? o : t->idx[k];
                const float *q = in->p + ((size_t)sy * in->w + sx) * 4;
                if (k == 0 && !t->normalise) { for (int c = 0; c < 4; c++) acc[c] = t->w[0] * q[c]; }
                else for (int c = 0; c < 4; c++) acc[c] = acc[c] + t->w[k] * q[c];
            }
            if (t->normalise) for (int c = 0; c < 4; c++) acc[c] = acc[c] / t->wsum;
            store_fmt(store, acc, out->p + ((size_t)y * out->w + x) * 4);
        }
    }
    free(tx); free(ty);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Process — DX11VideoProcessor.cpp:3285-3424 (shader path)                                    */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t pack_out(int out_fmt, const float v[4])
{
    if (out_fmt == ORC_OUT_RGB10A2) {
        uint32_t r = (uint32_t)floorf(saturatef(v[0]) * 1023.0f + 0.5f);
        uint32_t g = (uint32_t)floorf(saturatef(v[1]) * 1023.0f + 0.5f);
        uint32_t b = (uint32_t)floorf(saturatef(v[2]) * 1023.0f + 0.5f);
        uint32_t a = (uint32_t)floorf(saturatef(v[3]) * 3.0f + 0.5f);
        return r | (g << 10) | (b << 20) | (a << 30);        /* DXGI_FORMAT_R10G10B10A2_UNORM */
    }
    uint32_t r = (uint32_t)floorf(saturatef(v[0]) * 255.0f + 0.5f);
    uint32_t g = (uint32_t)floorf(saturatef(v[1]) * 255.0f + 0.5f);
    uint32_t b = (uint32_t)floorf(saturatef(v[2]) * 255.0f + 0.5f);
    uint32_t a = (uint32_t)floorf(saturatef(v[3]) * 255.0f + 0.5f);
    return b | (g << 8) | (r << 16) | (a << 24);             /* DXGI_FORMAT_B8G8R8A8_UNORM */
}

/* ------------------------------------------------------------------------------------------ */
/* correction passes (m_pPSCorrection)                                                          */
/* ------------------------------------------------------------------------------------------ */
/* convert/colorspace_gamut_conversion.hlsl:1-93 (zimg), evaluated in fp32 in the order written */
static float cg_det2(float a00, float a01, float a10, float a11) { return a00 * a11 - a01 * a10; }
static void cg_inverse(const float m[3][3], float r[3][3])
{
    float det = 0;
    det += m[0][0] * cg_det2(m[1][1], m[1][2], m[2][1], m[2][2]);
    det -= m[0][1] * cg_det2(m[1][0], m[1][2], m[2][0], m[2][2]);
    det += m[0][2] * cg_det2(m[1][0], m[1][1], m[2][0], m[2][1]);
    r[0][0] = cg_det2(m[1][1], m[1][2], m[2][1], m[2][2]) / det;
    r[0][1] = cg_det2(m[0][2], m[0][1], m[2][2], m[2][1]) / det;
    r[0][2] = cg_det2(m[0][1], m[0][2], m[1][1], m[1][2]) / det;
    r[1][0] = cg_det2(m[1][2], m[1][0], m[2][2], m[2][0]) / det;
    r[1][1] = cg_det2(m[0][0], m[0][2], m[2][0], m[2][2]) / det;
    r[1][2] = cg_det2(m[0][2], m[0][0], m[1][2], m[1][0]) / det;
    r[2][0] = cg_det2(m[1][0], m[1][1], m[2][0], m[2][1]) / det;
    r[2][1] = cg_det2(m[0][1], m[0][0], m[2][1], m[2][0]) / det;
    r[2][2] = cg_det2(m[0][0], m[0][1], m[1][0], m[1][1]) / det;
}
static void cg_xy_to_xyz(float x, float y, float out[3]) { out[0] = x / y; out[1] = 1.0f; out[2] = (1.0f - x - y) / y; }
static void cg_rgb_to_xyz(const float prim[3][2], float m[3][3])
{
    float rows[3][3], xyz[3][3], inv[3][3], white[3], sv[3];
    for (int i = 0; i < 3; i++) cg_xy_to_xyz(prim[i][0], prim[i][1], rows[i]);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) xyz[i][j] = rows[j][i];          /* transpose: columns R G B */
    cg_xy_to_xyz(0.3127f, 0.3290f, white);
    cg_inverse(xyz, inv);
    for (int i = 0; i < 3; i++) sv[i] = inv[i][0] * white[0] + inv[i][1] * white[1] + inv[i][2] * white[2];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = xyz[i][j] * sv[j];     /* row * s, component-wise */
}
static void mat4_mul(const float a[16], const float b[16], float c[16])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            c[i * 4 + j] = a[i * 4 + 0] * b[0 * 4 + j] + a[i * 4 + 1] * b[1 * 4 + j] + a[i * 4 + 2] * b[2 * 4 + j] + a[i * 4 + 3] * b[3 * 4 + j];
}
void orc_correction_matrices(float fix2020[16], float fixycgco[16], float gamut[9])
{
    /* convert/conv_matrix.hlsl */
    static const float rgb_ycbcr709[16] = {0.2126f, 0.7152f, 0.0722f, 0.0f, -0.114572f, -0.385428f, 0.5f, 0.0f,
                                           0.5f, -0.454153f, -0.045847f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    static const float ycbcr2020nc_rgb[16] = {1.0f, 0.0f, 1.4746f, 0.0f, 1.0f, -0.164553f, -0.571353f, 0.0f,
                                              1.0f, 1.8814f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    static const float ycgco_rgb[16] = {1.0f, -1.0f, 1.0f, 0.0f, 1.0f, 1.0f, 0.0f, 0.0f, 1.0f, -1.0f, -1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    static const float p709[3][2] = {{0.640f, 0.330f}, {0.300f, 0.600f}, {0.150f, 0.060f}};
    static const float p2020[3][2] = {{0.708f, 0.292f}, {0.170f, 0.797f}, {0.131f, 0.046f}};
    mat4_mul(ycbcr2020nc_rgb, rgb_ycbcr709, fix2020);        /* ps_fix_bt2020.hlsl:7 */
    mat4_mul(ycgco_rgb, rgb_ycbcr709, fixycgco);             /* ps_fix_ycgco.hlsl:6 */
    float m2020[3][3], m709[3][3], inv709[3][3];
    cg_rgb_to_xyz(p2020, m2020); cg_rgb_to_xyz(p709, m709);
    cg_inverse(m709, inv709);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            gamut[i * 3 + j] = inv709[i][0] * m2020[0][j] + inv709[i][1] * m2020[1][j] + inv709[i][2] * m2020[2][j];
}

static void corr_load(const uint8_t *row, int x, int fmt, float c[4])
{
    const uint32_t u = ((const uint32_t *)row)[x];
    if (fmt == 10) { c[0] = (float)(u & 1023u) / 1023.0f; c[1] = (float)((u >> 10) & 1023u) / 1023.0f; c[2] = (float)((u >> 20) & 1023u) / 1023.0f; c[3] = (float)(u >> 30) / 3.0f; }
    else { c[0] = (float)((u >> 16) & 255u) / 255.0f; c[1] = (float)((u >> 8) & 255u) / 255.0f; c[2] = (float)(u & 255u) / 255.0f; c[3] = (float)(u >> 24) / 255.0f; }
}
static void mat4_apply(const float m[16], float c[4])
{
    float r[4];
    for (int i = 0; i < 4; i++) r[i] = m[i * 4] * c[0] + m[i * 4 + 1] * c[1] + m[i * 4 + 2] * c[2] + m[i * 4 + 3] * c[3];
    memcpy(c, r, sizeof(r));
}

int orc_correction_pass(int kind, const uint8_t *src, int src_pitch, int src_fmt, uint8_t *dst, int dst_pitch, int dst_fmt,
                        int w, int h, int sdr_nits)
{
    if (kind < 1 || kind > 6 || (src_fmt != 8 && src_fmt != 10) || (dst_fmt != 8 && dst_fmt != 10) || w <= 0 || h <= 0) return -1;
    float fix2020[16], fixycgco[16], gamut[9];
    orc_correction_matrices(fix2020, fixycgco, gamut);
    const float lum = orc_luminance_scale(sdr_nits);                       /* SetShaderLuminanceParams :889-905 */
    ORC_PAR_FOR
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float c[4];
            corr_load(src + (size_t)y * src_pitch, x, src_fmt, c);
            if (kind == ORC_CORR_FIX_YCGCO) {
                mat4_apply(fixycgco, c);
            } else if (kind == ORC_CORR_CONVERT_HLG_TO_PQ) {
                for (int i = 0; i < 3; i++) c[i] = saturatef(c[i]);
                orc_hlg_to_linear(c);
                for (int i = 0; i < 3; i++) c[i] = orc_linear_to_st2084(c[i], 1000.0f);
            } else {
                if (kind != ORC_CORR_CONVERT_PQ_TO_SDR) mat4_apply(fix2020, c);
                for (int i = 0; i < 3; i++) c[i] = saturatef(c[i]);
                if (kind == ORC_CORR_FIX_BT2020) {
                    for (int i = 0; i < 3; i++) c[i] = hlsl_pow(c[i], 2.2f);
                } else {
                    if (kind == ORC_CORR_FIXCONVERT_HLG_TO_SDR) {
                        orc_hlg_to_linear(c);
                        for (int i = 0; i < 3; i++) c[i] = saturatef(orc_linear_to_st2084(c[i], 1000.0f));
                    }
                    for (int i = 0; i < 3; i++) c[i] = orc_st2084_to_linear(c[i], lum);
                    orc_tonemap_hable(c);
                }
                mat3_apply(gamut, c);
                for (int i = 0; i < 3; i++) c[i] = hlsl_pow(saturatef(c[i]), 1.0f / 2.2f);
            }
            /* the render target store: UNORM rounding, X8 / A2 left opaque like the swap chain */
            uint32_t *o = (uint32_t *)(dst + (size_t)y * dst_pitch) + x;
            if (dst_fmt == 10) {
                const uint32_t r = (uint32_t)floorf(saturatef(c[0]) * 1023.0f + 0.5f), g = (uint32_t)floorf(saturatef(c[1]) * 1023.0f + 0.5f),
                               b = (uint32_t)floorf(saturatef(c[2]) * 1023.0f + 0.5f);
                *o = r | (g << 10) | (b << 20) | 0xc0000000u;
            } else {
                const uint32_t r = (uint32_t)floorf(saturatef(c[0]) * 255.0f + 0.5f), g = (uint32_t)floorf(saturatef(c[1]) * 255.0f + 0.5f),
                               b = (uint32_t)floorf(saturatef(c[2]) * 255.0f + 0.5f);
                *o = b | (g << 8) | (r << 16) | 0xff000000u;
            }
        }
    }
    return 0;
}


/* ------------------------------------------------------------------------------------------ */
/* upload repack: the reference's only per-frame CPU work (MemCopyToTexSrcVideo :1213-1252)    */
/* ------------------------------------------------------------------------------------------ */
/* CopyPlaneAsIs — Helper.cpp:414-428 */
void orc_copy_plane_as_is(unsigned lines, uint8_t *dst, unsigned dst_pitch, const uint8_t *src, int src_pitch)
{
    if ((int)dst_pitch == src_pitch) { memcpy(dst, src, (size_t)dst_pitch * lines); return; }
    const unsigned a = (unsigned)(src_pitch < 0 ? -src_pitch : src_pitch), linesize = a < dst_pitch ? a : dst_pitch;
    for (unsigned y = 0; y < lines; ++y) { memcpy(dst, src, linesize); src += src_pitch; dst += dst_pitch; }
}
/* CopyPlane10to16 — Helper.cpp:789-803 */
void orc_copy_plane_10to16(unsigned lines, uint8_t *dst, unsigned dst_pitch, const uint8_t *src, int src_pitch)
{
    const unsigned line_pixels = (unsigned)src_pitch / 2;
    for (unsigned y = 0; y < lines; ++y) {
        const uint16_t *s16 = (const uint16_t *)src; uint16_t *d16 = (uint16_t *)dst;
        for (unsigned i = 0; i < line_pixels; i++) d16[i] = (uint16_t)(s16[i] << 6);
        src += src_pitch; dst += dst_pitch;
    }
}

void orc_params_default(orc_params *p)
{   /* Settings_t::SetDefault — IVideoRenderer.h:140-185 */
    memset(p, 0, sizeof(*p));
    p->iTexFormat = ORC_TEXFMT_AUTOINT;
    p->iChromaScaling = ORC_CHROMA_BILINEAR;
    p->iUpscaling = ORC_UP_CATMULLROM;
    p->iDownscaling = ORC_DOWN_HAMMING;
    p->bInterpolateAt50pct = 1;
    p->bUseDither = 1;
    p->bConvertToSdr = 1;
    p->iSDRDisplayNits = 125;
    p->output_format = ORC_OUT_BGRA8;
    p->contrast = 1.0f; p->saturation = 1.0f;
}

int orc_process(const orc_params *p, const uint8_t *src, int src_pitch,
                const uint16_t *dither_f16, uint8_t *dst, int dst_pitch)
{
    convert_ctx c;
    int rc = setup_convert(p, src, src_pitch, &c);
    if (rc) return rc;
    const int internal = c.internal_fmt;
    const int w1 = c.rect[2] - c.rect[0], h1 = c.rect[3] - c.rect[1];
    const int dl = p->video_rect[0], dt = p->video_rect[1];
    const int w2 = p->video_rect[2] - dl, h2 = p->video_rect[3] - dt;
    if (w2 <= 0 || h2 <= 0 || p->window_w <= 0 || p->window_h <= 0) { free(c.owned); return -4; }

    /* UpdatePostScaleTexures :2894-2912 */
    const int swap_fmt = (p->output_format == ORC_OUT_RGB10A2) ? FMT_RGB10A2 : FMT_BGRA8;
    const int need_dither = (swap_fmt == FMT_BGRA8 && internal != FMT_BGRA8) ||
                            (swap_fmt == FMT_RGB10A2 && internal == FMT_RGBA16F);
    const int final_pass = p->bUseDither && need_dither && dither_f16 != NULL;
    /* m_pPSHDR10ToneMapping: a post-scale step between the resize and the final pass (:3359-3367), created once the
       renderer has HDR10 metadata for an HDR source shown in HDR (:2716-2727) */
    const int trc_src = EXF_TRC(c.exfmt);
    const int tonemap = p->hdr_output && p->hdr_tonemap_type > 0 && (trc_src == TRC_2084 || trc_src == TRC_HLG || p->dovi);   /* SourceIsHDR() */
    const int has_steps = final_pass || tonemap;
    const float quant = (swap_fmt == FMT_RGB10A2) ? 1023.0f : 255.0f;     /* ps_final_pass QUANTIZATION */

    /* ConvertColorPass -> m_TexConvertOutput (w1 x h1, internal format); rSrc = whole texture :3316-3319 */
    /* with the convert draw disabled (interleaved RGB, default brightness/contrast) the source texture itself feeds
       the resize with rSrc = srcRect (:3321-3323) */
    img_t conv = {0}, mid = {0}, post = {0};
    const int *srect = c.enable ? NULL : c.rect;
    if (c.enable) {
        if (img_alloc(&conv, w1, h1)) { free(c.owned); return -5; }
        convert_pass(p, &c, &conv);
        if (g_cv_out_bias && (internal == FMT_RGB10A2 || internal == FMT_BGRA8)) {
            const float cmax = internal == FMT_RGB10A2 ? 1023.0f : 255.0f;
            const int amp = g_cv_out_bias < 0 ? -g_cv_out_bias : g_cv_out_bias;
            for (int y = 0; y < h1; y++)
                for (int x = 0; x < w1; x++)
                    for (int ch = 0; ch < 3; ch++) {
                        if (!g_cv_out_seed && g_cv_out_channel >= 0 && g_cv_out_channel != ch) continue;
                        int bias = g_cv_out_bias;
                        if (g_cv_out_seed) {
                            uint32_t h = ((uint32_t)x * 0x9E3779B9u) ^ ((uint32_t)y * 0x85EBCA6Bu) ^ ((uint32_t)ch * 0xC2B2AE35u) ^ g_cv_out_seed;
                            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
                            bias = (int)(h % (uint32_t)(2 * amp + 1)) - amp;
                        }
                        float *q = conv.p + ((size_t)y * w1 + x) * 4 + ch;
                        float code = floorf(*q * cmax + 0.5f) + (float)bias;
                        code = code < 0.0f ? 0.0f : (code > cmax ? cmax : code);
                        *q = code / cmax;
                    }
        }
    } else {
        if (img_alloc(&conv, p->width, p->height)) { free(c.owned); return -5; }
        for (int y = 0; y < p->height; y++)
            for (int x = 0; x < p->width; x++) {
                float *q = conv.p + ((size_t)y * p->width + x) * 4, v[3];
                fetch_pixel(&c.tex, 0, 0, 0, x, y, v);
                q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = 1.0f;
            }
    }

    /* ResizeShaderPass :3103-3187 — pick per-axis shader */
    const int k = p->bInterpolateAt50pct ? 2 : 1;
    const int rot = p->rotation, flip = p->flip != 0;
    if (rot != 0 && rot != 90 && rot != 180 && rot != 270) { img_free(&conv); free(c.owned); return -7; }
    const int rotated = (rot == 90 || rot == 270);
    const int sw = rotated ? h1 : w1, sh = rotated ? w1 : h1;            /* w1,h1 of ResizeShaderPass :3112-3123 */
    resizer_t up = {p->iUpscaling == ORC_UP_NEAREST ? RS_NONE : RS_UP, p->iUpscaling};
    resizer_t down = {RS_DOWN, p->iDownscaling};
    resizer_t none = {RS_NONE, 0};
    resizer_t rx = (sw == w2) ? none : (sw > k * w2) ? down : up;        /* filters the screen-x direction */
    resizer_t ry = (sh == h2) ? none : (sh > k * h2) ? down : up;
    /* texture axis each of them runs on in the rotation-carrying draw: X shaders filter texture X, Y shaders texture Y;
       rotated: resizerX is a Y shader, and resizerY is a Y shader of the second draw when resizerX exists, else an X
       shader of the single rotated draw (:3112-3121) */
    const int ax_first = rotated ? 1 : 0;
    /* destination format of the last resize draw: post-scale texture (internal) if a final pass
       follows, else the render target itself (:3334-3352, :3417-3419) */
    const int last_store = has_steps ? internal : swap_fmt;
    /* Process :3348-3352: with post-scale steps the resize is skipped when rSrc == dstRect and rotation == 0 (flip alone
       is then ignored); without them ResizeShaderPass always runs */
    const int same_rect = (w1 == w2 && h1 == h2 && dl == 0 && dt == 0);

    const img_t *result = &conv;
    int result_fmt = internal;
    /* resizerX == resizerY (:3131): rotated frames (two Y shaders) or Jinc2, whose one 2-D shader serves both axes (:2921) */
    const int same_shader = rx.kind != RS_NONE && rx.kind == ry.kind && (rotated || (rx.kind == RS_UP && rx.method == ORC_UP_JINC2));
    if (rx.kind != RS_NONE && ry.kind != RS_NONE && !same_shader) {
        /* two passes through fp16 m_TexResize (w2 x sh) :3143-3167; the second one is unrotated */
        if (img_alloc(&mid, w2, sh) || img_alloc(&post, w2, h2)) { rc = -5; goto done; }
        if ((rc = resize_draw(&conv, srect, &mid, ax_first, rx, rot, flip, p->flags, FMT_RGBA16F))) goto done;
        if ((rc = resize_draw(&mid, NULL, &post, 1, ry, 0, 0, p->flags, last_store))) goto done;
        result = &post; result_fmt = last_store;
    } else if (rx.kind != RS_NONE || ry.kind != RS_NONE || sw != w2 || sh != h2 || rot != 0 || (flip && !(has_steps && same_rect))) {
        /* one draw: one filtered axis (resizerX == resizerY for a rotated frame scaled the same way on both axes:
           :3131-3137 draws once, so only texture Y is filtered), or ps_simple (:3169-3181) */
        if (img_alloc(&post, w2, h2)) { rc = -5; goto done; }
        if (rx.kind != RS_NONE)      rc = resize_draw(&conv, srect, &post, ax_first, rx, rot, flip, p->flags, last_store);
        else if (ry.kind != RS_NONE) rc = resize_draw(&conv, srect, &post, rotated ? 0 : 1, ry, rot, flip, p->flags, last_store);
        else                         rc = resize_draw(&conv, srect, &post, -1, none, rot, flip, p->flags, last_store);
        if (rc) goto done;
        result = &post; result_fmt = last_store;
    } else if (!has_steps) {
        /* TextureCopyRect with ps_simple into the render target :3178-3181 */
        if (img_alloc(&post, w2, h2)) { rc = -5; goto done; }
        for (int y = 0; y < h2; y++)
            for (int x = 0; x < w2; x++)
                store_fmt(swap_fmt, conv.p + ((size_t)(y + (srect ? srect[1] : 0)) * conv.w + x + (srect ? srect[0] : 0)) * 4,
                          post.p + ((size_t)y * w2 + x) * 4);
        result = &post; result_fmt = swap_fmt;
    }
    img_t tm = {0};
    if (tonemap) {      /* TextureCopyRect(..., m_pPSHDR10ToneMapping, ...) into the next post-scale texture or the RT */
        const int ox = (result == &conv && srect) ? srect[0] : 0, oy = (result == &conv && srect) ? srect[1] : 0;
        if (img_alloc(&tm, w2, h2)) { rc = -5; goto done; }
        for (int y = 0; y < h2; y++)
            for (int x = 0; x < w2; x++) {
                const float *q = result->p + ((size_t)(y + oy) * result->w + (x + ox)) * 4;
                float v[4] = {q[0], q[1], q[2], 1.0f};
                if (g_tm_in_bias) {
                    const float in_max = result_fmt == FMT_RGB10A2 ? 1023.0f : result_fmt == FMT_BGRA8 ? 255.0f : 0.0f;
                    for (int ch = 0; ch < 3; ch++) v[ch] = tm_input_probe(v[ch], in_max, x, y, ch);
                }
                orc_hdr10_tonemap(v, p);
                store_fmt(final_pass ? internal : swap_fmt, v, tm.p + ((size_t)y * w2 + x) * 4);
            }
        result = &tm; srect = NULL;
    }

    /* FinalPass :3189-3233 + ps_final_pass.hlsl:23-31, or plain store into the render target */
    ORC_PAR_FOR
    for (int y = 0; y < h2; y++) {
        int wy = dt + y;
        if (wy < 0 || wy >= p->window_h) continue;
        for (int x = 0; x < w2; x++) {
            int wx = dl + x;
            if (wx < 0 || wx >= p->window_w) continue;
            /* the final pass reads the source texture itself when nothing was drawn before it (pTex = pInputTexture :3352) */
            const int ox = (result == &conv && srect) ? srect[0] : 0, oy = (result == &conv && srect) ? srect[1] : 0;
            const float *q = result->p + ((size_t)(y + oy) * result->w + (x + ox)) * 4;
            float v[4] = {q[0], q[1], q[2], q[3]};
            if (final_pass) {
                /* sampler WRAP+POINT, ditherCoordScale = texSize/32 => texel (wx mod 32, wy mod 32) */
                float d = orc_half_bits_to_float(dither_f16[(wy & 31) * 32 + (wx & 31)]);
                for (int ch = 0; ch < 4; ch++) v[ch] = floorf(v[ch] * quant + d) / quant;
            }
            uint32_t px = pack_out(p->output_format, v);
            memcpy(dst + (size_t)wy * dst_pitch + (size_t)wx * 4, &px, 4);
        }
    }
    rc = 0;
done:
    img_free(&conv); img_free(&mid); img_free(&post); img_free(&tm);
    free(c.owned);
    return rc;
}

/* ---- EXTENSION: error-diffusion final pass (bUseDither = 2) ---------------------------------------------------------------------
 * NOT a restatement of reference code: the reference's final pass is the ordered dither above and nothing else (`grep -ri
 * diffusion` over /root/reference is empty).  BASELINE.json's config 4 names "error-diffusion dither"; this serial loop is the
 * definition the product's kernel (videorenderer_amd/csrc/vp_errdiff.hip) is held to, bit for bit — PARITY UNPINNED by construction.
 *
 * The frame is rendered as for a 10-bit swap chain (R10G10B10A2, no final pass: orc_process with output_format = ORC_OUT_RGB10A2);
 * inside [x0, x1) x [y0, y1) (video rect ∩ window), rows top to bottom, each row left to right, per channel, in integers with
 * U = 16 * 1023 error units per 8-bit code:
 *     T = 4080 k + E(x, y);  q = clamp(floor((T + U/2) / U), 0, 255);  e = T - q U
 *     E(x+1, y) += floor(7 e / 16);  E(x-1, y+1) += floor(3 e / 16);  E(x, y+1) += floor(5 e / 16);  E(x+1, y+1) += the remainder
 * Shares that leave the region are dropped.  dst: B8G8R8A8 (alpha 0xFF), touched only inside the region. */
static int32_t ed_floor_div(int64_t a, int32_t b) { return (int32_t)(a >= 0 ? a / b : -((-a + b - 1) / b)); }

int orc_error_diffusion(const uint32_t *src10, int src_pitch, uint8_t *dst, int dst_pitch, int x0, int y0, int x1, int y1)
{
    const int U = 16 * 1023;
    const int w = x1 - x0;
    if (w <= 0 || y1 <= y0) return -1;
    /* errors received by the current and the next row, one guard column on each side */
    int32_t *buf = (int32_t *)calloc((size_t)2 * 3 * (w + 2), sizeof(int32_t));
    if (!buf) return -5;
    int32_t *cur = buf, *nxt = buf + 3 * (w + 2);
    for (int y = y0; y < y1; y++) {
        const uint32_t *srow = (const uint32_t *)((const uint8_t *)src10 + (size_t)y * src_pitch);
        uint8_t *drow = dst + (size_t)y * dst_pitch;
        memset(nxt, 0, sizeof(int32_t) * 3 * (w + 2));
        for (int x = 0; x < w; x++) {
            const uint32_t t = srow[x0 + x];
            int q3[3];
            for (int c = 0; c < 3; c++) {              /* c = 0 R (bits 0-9), 1 G, 2 B */
                int32_t *ec = cur + c * (w + 2) + 1, *en = nxt + c * (w + 2) + 1;
                const int32_t k = (int32_t)((t >> (10 * c)) & 0x3ffu);
                const int32_t T = 4080 * k + ec[x];
                int32_t q = ed_floor_div((int64_t)T + U / 2, U);
                q = q < 0 ? 0 : q > 255 ? 255 : q;
                const int32_t e = T - q * U;
                const int32_t r = ed_floor_div(7 * (int64_t)e, 16), bl = ed_floor_div(3 * (int64_t)e, 16), b = ed_floor_div(5 * (int64_t)e, 16);
                const int32_t br = e - r - bl - b;
                if (x + 1 < w) { ec[x + 1] += r; en[x + 1] += br; }
                if (x > 0) en[x - 1] += bl;
                en[x] += b;
                q3[c] = q;
            }
            uint8_t *px = drow + (size_t)(x0 + x) * 4;
            px[0] = (uint8_t)q3[2]; px[1] = (uint8_t)q3[1]; px[2] = (uint8_t)q3[0]; px[3] = 0xff;
        }
        int32_t *sw = cur; cur = nxt; nxt = sw;
    }
    free(buf);
    return 0;
}
