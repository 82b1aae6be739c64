// vp_plan.cpp — host-side parameter maths (see vp_plan.h).  Compiled with -ffp-contract=off so the
// fp32/fp64 expression shapes below round exactly like the reference's MSVC build.
#include "vp_plan.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "../../include/mpcvr.h"
#include "vp_crmath.h"      // the shaders' sin / cos as defined functions (the weights the resize shaders compute per pixel are computed here, once)

namespace mpcvr {

// ------------------------------------------------------------------------------------------------
// formats — Helper.cpp:295-359
// ------------------------------------------------------------------------------------------------
static const FmtConvParams kFormats[] = {
    //  cformat             str          pl by dw dh pack coeff sub  depth shift vfirst
    {MPCVR_CF_NV12,      "NV12",       2, 1, 2, 2, 1, 3, 420,  8, 0, 0},
    {MPCVR_CF_P010,      "P010",       2, 2, 2, 2, 2, 3, 420, 16, 0, 0},
    {MPCVR_CF_P016,      "P016",       2, 2, 2, 2, 2, 3, 420, 16, 0, 0},
    {MPCVR_CF_P210,      "P210",       2, 2, 2, 1, 2, 4, 422, 16, 0, 0},
    {MPCVR_CF_P216,      "P216",       2, 2, 2, 1, 2, 4, 422, 16, 0, 0},
    {MPCVR_CF_YV12,      "YV12",       3, 1, 2, 2, 1, 3, 420,  8, 0, 1},
    {MPCVR_CF_YV16,      "YV16",       3, 1, 2, 1, 1, 4, 422,  8, 0, 1},
    {MPCVR_CF_YV24,      "YV24",       3, 1, 1, 1, 1, 6, 444,  8, 0, 1},
    {MPCVR_CF_YUV420P8,  "YUV420P8",   3, 1, 2, 2, 1, 3, 420,  8, 0, 0},
    {MPCVR_CF_YUV422P8,  "YUV422P8",   3, 1, 2, 1, 1, 4, 422,  8, 0, 0},
    {MPCVR_CF_YUV444P8,  "YUV444P8",   3, 1, 1, 1, 1, 6, 444,  8, 0, 0},
    {MPCVR_CF_YUV420P10, "YUV420P10",  3, 2, 2, 2, 2, 3, 420, 10, 6, 0},
    {MPCVR_CF_YUV420P16, "YUV420P16",  3, 2, 2, 2, 2, 3, 420, 16, 0, 0},
    {MPCVR_CF_YUV422P10, "YUV422P10",  3, 2, 2, 1, 2, 4, 422, 10, 6, 0},
    {MPCVR_CF_YUV422P16, "YUV422P16",  3, 2, 2, 1, 2, 4, 422, 16, 0, 0},
    {MPCVR_CF_YUV444P10, "YUV444P10",  3, 2, 1, 1, 2, 6, 444, 10, 6, 0},
    {MPCVR_CF_YUV444P16, "YUV444P16",  3, 2, 1, 1, 2, 6, 444, 16, 0, 0},
    // one RGBA8 / RGBA16 texel = two pixels (DX11Plane_RGBA8 / DX11Plane_RGBA16); ci = components of Y0,U,Y1,V
    {MPCVR_CF_YUY2,      "YUY2",       1, 1, 2, 1, 2, 2, 422,  8, 0, 0, LAY_PACKED422, CST_YUV, {0, 1, 2, 3}, 0},
    {MPCVR_CF_UYVY,      "UYVY",       1, 1, 2, 1, 2, 2, 422,  8, 0, 0, LAY_PACKED422, CST_YUV, {1, 0, 3, 2}, 0},
    {MPCVR_CF_Y210,      "Y210",       1, 2, 2, 1, 4, 2, 422, 10, 0, 0, LAY_PACKED422, CST_YUV, {0, 1, 2, 3}, 0},
    {MPCVR_CF_Y216,      "Y216",       1, 2, 2, 1, 4, 2, 422, 16, 0, 0, LAY_PACKED422, CST_YUV, {0, 1, 2, 3}, 0},
    {MPCVR_CF_V210,      "v210",       1, 2, 2, 1, 0, 2, 422, 10, 0, 0, LAY_PACKED422, CST_YUV, {0, 1, 2, 3}, 0},
    // one texel = one pixel; memory order AYUV: V,U,Y,A  Y410: U:10,Y:10,V:10,A:2  Y416: U,Y,V,A (Shaders.cpp:186-193)
    {MPCVR_CF_AYUV,      "AYUV",       1, 1, 1, 1, 4, 2, 444,  8, 0, 0, LAY_PACKED444, CST_YUV, {2, 1, 0, 3}, 0},
    {MPCVR_CF_Y410,      "Y410",       1, 4, 1, 1, 4, 2, 444, 10, 0, 0, LAY_PACKED444, CST_YUV, {1, 0, 2, 3}, 1},
    {MPCVR_CF_Y416,      "Y416",       1, 2, 1, 1, 8, 2, 444, 16, 0, 0, LAY_PACKED444, CST_YUV, {1, 0, 2, 3}, 0},
    // planar RGB: planes G,B,R sampled as texY,texU,texV; matrix columns rotated (DX11VideoProcessor.cpp:863-867)
    {MPCVR_CF_GBRP8,     "GBRP8",      3, 1, 1, 1, 1, 6, 444,  8, 0, 0, LAY_PLANAR, CST_RGB, {0, 0, 0, 0}, 0},
    {MPCVR_CF_GBRP10,    "GBRP10",     3, 2, 1, 1, 2, 6, 444, 10, 6, 0, LAY_PLANAR, CST_RGB, {0, 0, 0, 0}, 0},
    {MPCVR_CF_GBRP16,    "GBRP16",     3, 2, 1, 1, 2, 6, 444, 16, 0, 0, LAY_PLANAR, CST_RGB, {0, 0, 0, 0}, 0},
    // gray: R8 / R16 texture
    {MPCVR_CF_Y8,        "Y8",         1, 1, 1, 1, 1, 2, 400,  8, 0, 0, LAY_GRAY, CST_GRAY, {0, 0, 0, 0}, 0},
    {MPCVR_CF_Y10,       "Y10",        1, 2, 1, 1, 2, 2, 400, 10, 6, 0, LAY_GRAY, CST_GRAY, {0, 0, 0, 0}, 0},
    {MPCVR_CF_Y16,       "Y16",        1, 2, 1, 1, 2, 2, 400, 16, 0, 0, LAY_GRAY, CST_GRAY, {0, 0, 0, 0}, 0},
    // interleaved RGB (Helper.cpp:345-354): texture B8G8R8X8 / R10G10B10A2 / R16G16B16A16; ci = texel components of R,G,B
    {MPCVR_CF_RGB24,     "RGB24",      1, 1, 1, 1, 3, 2, 444,  8, 0, 0, LAY_RGB, CST_RGB, {2, 1, 0, 3}, 0, RPK_RGB24},
    {MPCVR_CF_XRGB32,    "RGB32",      1, 1, 1, 1, 4, 2, 444,  8, 0, 0, LAY_RGB, CST_RGB, {2, 1, 0, 3}, 0, RPK_NONE},
    {MPCVR_CF_ARGB32,    "ARGB32",     1, 1, 1, 1, 4, 2, 444,  8, 0, 0, LAY_RGB, CST_RGB, {2, 1, 0, 3}, 0, RPK_NONE},
    {MPCVR_CF_r210,      "r210",       1, 4, 1, 1, 4, 2, 444, 10, 0, 0, LAY_RGB, CST_RGB, {0, 1, 2, 3}, 1, RPK_R210},
    {MPCVR_CF_RGB48,     "RGB48",      1, 2, 1, 1, 6, 2, 444, 16, 0, 0, LAY_RGB, CST_RGB, {0, 1, 2, 3}, 0, RPK_RGB48},
    {MPCVR_CF_BGR48,     "BGR48",      1, 2, 1, 1, 6, 2, 444, 16, 0, 0, LAY_RGB, CST_RGB, {0, 1, 2, 3}, 0, RPK_BGR48},
    {MPCVR_CF_BGRA64,    "BGRA64",     1, 2, 1, 1, 8, 2, 444, 16, 0, 0, LAY_RGB, CST_RGB, {0, 1, 2, 3}, 0, RPK_BGRA64},
    {MPCVR_CF_B64A,      "b64a",       1, 2, 1, 1, 8, 2, 444, 16, 0, 0, LAY_RGB, CST_RGB, {0, 1, 2, 3}, 0, RPK_B64A},
};

const FmtConvParams *GetFmtConvParams(int cformat)
{
    for (const auto &f : kFormats)
        if (f.cformat == cformat) return &f;
    return nullptr;
}

int DefaultPitch(const FmtConvParams &f, int width)
{
    int pitch = width * f.Packsize;
    if (f.cformat == MPCVR_CF_NV12 || f.cformat == MPCVR_CF_Y8 || f.cformat == MPCVR_CF_RGB24 || f.cformat == MPCVR_CF_BGR48)
        pitch = (pitch + 3) & ~3;                                                             // :1792-1796
    if (f.cformat == MPCVR_CF_V210) pitch = (((width + 5) / 6 * 16) + 127) & ~127;            // :1798-1799
    return pitch;
}

int V210TexPitch(int width) { return (4 * width + 11) / 12 * 12; }

int SourceLines(const FmtConvParams &f, int height) { return height * f.PitchCoeff / 2; }

// ------------------------------------------------------------------------------------------------
// extended format defaults — Helper.cpp:1169-1211 (every accepted format is CS_YUV)
// ------------------------------------------------------------------------------------------------
namespace dxva {
enum { Chroma_MPEG1 = 1, Chroma_MPEG2 = 5, Chroma_Cosited = 7 };
enum { Range_0_255 = 1, Range_16_235 = 2 };
enum { Matrix_BT709 = 1, Matrix_BT601 = 2, Matrix_SMPTE240M = 3, Matrix_BT2020_10 = 4, Matrix_YCgCo = 7 };
enum { Prim_BT709 = 2, Prim_BT2020 = 9 };
enum { TF_10 = 1, TF_18 = 2, TF_20 = 3, TF_22 = 4, TF_709 = 5, TF_240M = 6, TF_sRGB = 7, TF_28 = 8,
       TF_26 = 14, TF_2084 = 15, TF_HLG = 16 };
enum { Lighting_dim = 3 };
}  // namespace dxva

ExtFmt SpecifyExtendedFormat(ExtFmt ex, const FmtConvParams &f, int w, int h)
{
    if (f.CSType == CST_RGB) { ex.value = 0; return ex; }       // :1171-1173
    if (f.CSType == CST_GRAY) return ex;                        // neither branch: left as the decoder gave it
    if (f.Subsampling != 420) ex.set(8, 0xf, 0);
    else if (ex.VideoChromaSubsampling() == 0) ex.set(8, 0xf, dxva::Chroma_MPEG2);
    if (ex.NominalRange() == 0) ex.set(12, 0x7, dxva::Range_16_235);
    if (ex.VideoTransferMatrix() == 0)
        ex.set(15, 0x7, (w <= 1024 && h <= 576) ? dxva::Matrix_BT601 : dxva::Matrix_BT709);
    if (ex.VideoLighting() == 0) ex.set(18, 0xf, dxva::Lighting_dim);
    if (ex.VideoPrimaries() == 0) ex.set(22, 0x1f, dxva::Prim_BT709);
    if (ex.VideoTransferFunction() == 0) ex.set(27, 0x1f, dxva::TF_709);
    return ex;
}

// ------------------------------------------------------------------------------------------------
// YUV->RGB matrix — mp_get_csp_matrix (csputils.cpp:392-509) specialised to what
// SetShaderConvertColorParams (DX11VideoProcessor.cpp:813-887) feeds it:
// levels_out = PC, gamma untouched, input_bits = texture_bits = CDepth.
// ------------------------------------------------------------------------------------------------
namespace {

enum CspSpace { SP_AUTO = 0, SP_601, SP_709, SP_240M, SP_2020NC, SP_YCGCO, SP_RGB };

struct Mat3 { float v[3][3]; };

Mat3 FromLumaWeights(float lr, float lg, float lb)      // csputils.cpp:380-389
{
    Mat3 m;
    m.v[0][0] = 1; m.v[0][1] = 0;                       m.v[0][2] = 2 * (1 - lr);
    m.v[1][0] = 1; m.v[1][1] = -2 * (1 - lb) * lb / lg; m.v[1][2] = -2 * (1 - lr) * lr / lg;
    m.v[2][0] = 1; m.v[2][1] = 2 * (1 - lb);            m.v[2][2] = 0;
    return m;
}

}  // namespace

void ComputeColorMatrix(const ExtFmt &ex, const FmtConvParams &f, const ProcAmp &pa, float out[12])
{
    // set_colorspace — Helper.cpp:949-1004
    int levels_tv;   // 1 = TV, 0 = PC
    switch (ex.NominalRange()) {
    case dxva::Range_0_255: levels_tv = 0; break;
    default: levels_tv = 1; break;                     // 16-235, and AUTO -> TV (csputils.cpp:397-399)
    }
    CspSpace sp;
    switch (ex.VideoTransferMatrix()) {
    case dxva::Matrix_BT709: sp = SP_709; break;
    case dxva::Matrix_BT601: sp = SP_601; break;
    case dxva::Matrix_SMPTE240M: sp = SP_240M; break;
    case dxva::Matrix_BT2020_10: sp = SP_2020NC; break;
    case dxva::Matrix_YCgCo: sp = SP_YCGCO; break;
    default: sp = SP_601; break;                        // AUTO -> BT.601 (csputils.cpp:395-396)
    }
    if (ex.value == 0) { sp = SP_RGB; levels_tv = 0; }   // set_colorspace: value == 0 => MP_CSP_RGB, PC (Helper.cpp:953-957)
    const bool gray = f.CSType == CST_GRAY;             // csp_params.gray (:843)

    const float brightness = pa.brightness / 255;                              // :839
    const float contrast = pa.contrast;                                        // :840
    const float hue = (float)(pa.hue / 180 * std::acos(-1.0));                 // :841
    const float saturation = pa.saturation;                                    // :842

    Mat3 m;
    switch (sp) {
    case SP_709: m = FromLumaWeights(0.2126f, 0.7152f, 0.0722f); break;
    case SP_240M: m = FromLumaWeights(0.2122f, 0.7013f, 0.0865f); break;
    case SP_2020NC: m = FromLumaWeights(0.2627f, 0.6780f, 0.0593f); break;
    case SP_YCGCO: {
        const float y[3][3] = {{1, -1, 1}, {1, 1, 0}, {1, -1, -1}};
        std::memcpy(m.v, y, sizeof(y));
        break;
    }
    case SP_RGB: {                                      // csputils.cpp:416-420: identity, levels_in = -1 ("anyfull")
        const float y[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        std::memcpy(m.v, y, sizeof(y));
        break;
    }
    default: m = FromLumaWeights(0.299f, 0.587f, 0.114f); break;
    }

    if (sp != SP_YCGCO && sp != SP_RGB) {               // hue rotation + saturation, csputils.cpp:447-459
        const float hc = gray ? 0 : saturation * std::cos(hue);    // float overloads, as in the C++ reference
        const float hs = gray ? 0 : saturation * std::sin(hue);
        for (auto &row : m.v) {
            const float u = row[1], v = row[2];
            row[1] = hc * u - hs * v;
            row[2] = hs * u + hc * v;
        }
    }

    // mp_get_csp_mul (csputils.cpp:341-358) with input_bits == texture_bits == CDepth (:845)
    const int bits = f.CDepth;
    const double full = (double)(1LL << bits);
    const double mul = (sp == SP_RGB) ? (full - 1.) / (full - 1.)          // RGB always uses the full range (:351-353)
                                      : full / (full - 1.) * 255 / 256;
    const double s = mul / 255;
    double ymin = (levels_tv ? 16 : 0) * s, ymax = (levels_tv ? 235 : 255) * s;
    double cmax = (levels_tv ? 240 : 255) * s, cmid = 128 * s;
    if (sp == SP_RGB) { ymin = 0 * s; ymax = 255 * s; cmax = 255 * s / 2; cmid = 0; }   // anyfull (:474)
    double ymul = (1.0 - 0.0) / (ymax - ymin);
    double cmul = (1.0 - 0.0) / (cmax - cmid) / 2;
    ymul *= contrast;
    cmul *= contrast;
    for (int i = 0; i < 3; i++) {
        m.v[i][0] = (float)(m.v[i][0] * ymul);
        m.v[i][1] = (float)(m.v[i][1] * cmul);
        m.v[i][2] = (float)(m.v[i][2] * cmul);
        const float uv = m.v[i][1] + m.v[i][2];
        out[9 + i] = (float)(0.0 - m.v[i][0] * ymin - uv * cmid + brightness);
    }
    // cbuffer fix-ups of SetShaderConvertColorParams (DX11VideoProcessor.cpp:863-873)
    if (f.CSType == CST_RGB && f.layout == LAY_PLANAR && f.planes == 3) {    // GBRP: color = (G,B,R) => rows (x,y,z) -> (y,z,x)
        for (auto &row : m.v) { const float x = row[0], y = row[1], z = row[2]; row[0] = y; row[1] = z; row[2] = x; }
    } else if (gray) {
        m.v[1][0] = m.v[1][1]; m.v[1][1] = 0;
        m.v[2][0] = m.v[2][2]; m.v[2][2] = 0;
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[i * 3 + j] = m.v[i][j];
}

// ------------------------------------------------------------------------------------------------
// BT.2020 -> BT.709 gamut matrix — csputils.cpp:10-49,228-259,549-557
// ------------------------------------------------------------------------------------------------
namespace {

void Invert(float m[3][3])
{
    const float a = m[0][0], b = m[0][1], c = m[0][2], d = m[1][0], e = m[1][1], f = m[1][2],
                g = m[2][0], h = m[2][1], k = m[2][2];
    m[0][0] = (e * k - h * f);  m[0][1] = -(b * k - h * c); m[0][2] = (b * f - e * c);
    m[1][0] = -(d * k - g * f); m[1][1] = (a * k - g * c);  m[1][2] = -(a * f - d * c);
    m[2][0] = (d * h - g * e);  m[2][1] = -(a * h - g * b); m[2][2] = (a * e - d * b);
    float det = a * m[0][0] + d * m[0][1] + g * m[0][2];
    det = 1.0f / det;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) m[i][j] *= det;
}

void MulInPlace(float a[3][3], const float b[3][3])
{
    float t[3][3];
    std::memcpy(t, a, sizeof(t));
    for (int i = 0; i < 3; i++)
        for (int r = 0; r < 3; r++) a[r][i] = t[r][0] * b[0][i] + t[r][1] * b[1][i] + t[r][2] * b[2][i];
}

struct XY { float x, y; };
struct Prim { XY r, g, b, w; };

void RgbToXyz(const Prim &p, float m[3][3])
{
    float X[4] = {p.r.x / p.r.y, p.g.x / p.g.y, p.b.x / p.b.y, p.w.x / p.w.y};
    float Z[4] = {(1 - p.r.x - p.r.y) / p.r.y, (1 - p.g.x - p.g.y) / p.g.y,
                  (1 - p.b.x - p.b.y) / p.b.y, (1 - p.w.x - p.w.y) / p.w.y};
    for (int i = 0; i < 3; i++) { m[0][i] = X[i]; m[1][i] = 1; m[2][i] = Z[i]; }
    Invert(m);
    float S[3];
    for (int i = 0; i < 3; i++) S[i] = m[i][0] * X[3] + m[i][1] * 1 + m[i][2] * Z[3];
    for (int i = 0; i < 3; i++) { m[0][i] = S[i] * X[i]; m[1][i] = S[i] * 1; m[2][i] = S[i] * Z[i]; }
}

}  // namespace

void ComputeGamut2020to709(float out[9])
{
    const XY d65 = {0.31271f, 0.32902f};
    const Prim bt2020 = {{0.708f, 0.292f}, {0.170f, 0.797f}, {0.131f, 0.046f}, d65};
    const Prim bt709 = {{0.640f, 0.330f}, {0.300f, 0.600f}, {0.150f, 0.060f}, d65};
    float in[3][3], m[3][3];
    RgbToXyz(bt2020, in);
    RgbToXyz(bt709, m);
    Invert(m);
    MulInPlace(m, in);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[i * 3 + j] = m[i][j];
}

// Shaders.cpp:613-616, 861-915
void SelectTail(const ExtFmt &ex, bool convert_to_sdr, int *tail, float *gamma, bool hdr_output, bool dovi)
{
    const unsigned tf = ex.VideoTransferFunction();
    *tail = TAIL_NONE;
    *gamma = 1.0f;
    if (hdr_output) {
        convert_to_sdr = false;                                            // convertType is never TO_SDR (:2948)
        if (tf == dxva::TF_HLG && !dovi) { *tail = TAIL_HLG_TO_PQ; return; }   // SHADER_CONVERT_TO_PQ (:2949), bApplyHLG (:615)
    }
    if (convert_to_sdr && (tf == dxva::TF_2084 || dovi)) { *tail = TAIL_PQ_TO_SDR; return; }
    if (convert_to_sdr && tf == dxva::TF_HLG) { *tail = TAIL_HLG_TO_SDR; return; }
    if (ex.VideoPrimaries() == dxva::Prim_BT2020) {
        float g = 0;
        switch (tf) {
        case dxva::TF_10: g = 1.0f; break;
        case dxva::TF_18: g = 1.8f; break;
        case dxva::TF_20: g = 2.0f; break;
        case dxva::TF_HLG: case dxva::TF_22: case dxva::TF_709: case dxva::TF_240M: case dxva::TF_sRGB: g = 2.2f; break;
        case dxva::TF_28: g = 2.8f; break;
        case dxva::TF_26: g = 2.6f; break;
        default: break;
        }
        if (g != 0) { *tail = TAIL_GAMMA_GAMUT; *gamma = g; }
    }
}

// ------------------------------------------------------------------------------------------------
// PQ -> SDR per-channel table (Shaders/convert/st2084.hlsl:1-16, hdr_tone_mapping.hlsl:1-13)
// ------------------------------------------------------------------------------------------------
void SanitiseHdr10Params(HdrToneMapParams *k)
{
    if (k->min_mastering <= 0.f) k->min_mastering = 0.f;
    if (k->max_mastering <= 10.f) k->max_mastering = 1000.f;
    if (k->max_cll <= 10.f) k->max_cll = k->max_mastering;
    if (k->max_fall <= 1.f) k->max_fall = k->max_cll;
    if (k->display_max < 100.f || k->display_max > 10000.f) k->display_max = 1000.f;
    if (k->selection < 1 || k->selection > 6) k->selection = 1;
}

uint32_t FinalPassMultiplier(int quant, int maxv)
{
    if (quant <= 0 || maxv <= 0) return 0;
    const uint64_t m = (((uint64_t)quant << 24) + (uint64_t)maxv - 1) / (uint64_t)maxv;
    if (m >= (1u << 24) || (uint64_t)maxv * m + (1023ull << 14) >= (1ull << 32)) return 0;
    return (uint32_t)m;
}

void BuildPqSdrLut(float lum_scale, float out[kPqLutSize])
{
    const float m1 = 2610.0f / (4096.0f * 4.0f), m2 = (2523.0f / 4096.0f) * 128.0f;
    const float c1 = 3424.0f / 4096.0f, c2 = (2413.0f / 4096.0f) * 32.0f, c3 = (2392.0f / 4096.0f) * 32.0f;
    auto hable = [](float x) {
        const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
        return ((x * (A * x + (C * B)) + (D * E)) / (x * (A * x + B) + (D * F))) - E / F;
    };
    const float div = hable(4.8f);
    for (int i = 0; i < kPqLutSize; i++) {
        float x = (float)i / (float)(kPqLutSize - 1);
        x = std::exp2(std::log2(x) * (1.0f / m2));
        x = std::fmax(x - c1, 0.0f) / (c2 - c3 * x);
        x = std::exp2(std::log2(x) * (1.0f / m1));
        x *= lum_scale;
        out[i] = hable(x) / div;
    }
}

// inverse_HLG (Shaders/convert/hlg.hlsl:1-9) per channel at x = i / (N - 1): the HLG -> SDR tail of the fused kernels reads it from LDS
void BuildHlgInverseLut(float out[kPqLutSize])
{
    const float a = 0.17883277f, b = 0.28466892f, c = 0.55991073f;
    for (int i = 0; i < kPqLutSize; i++) {
        const float x = (float)i / (float)(kPqLutSize - 1);
        out[i] = x <= 0.5f ? x * x * 4.0f : std::exp((x - c) / a) + b;
    }
}

void BuildPqEotfLut(float out[kEotfLutSize + 1])
{
    // evaluated in double and rounded once: the table is as exact as an fp32 log2 value can be (1e-6 relative on the decoded value);
    // until round 4 it went through the shader's own fp32 exp2(y log2 x) chain, whose 1e-4 error in log2 rode on every entry
    const double m1 = 2610.0 / (4096.0 * 4.0), m2 = (2523.0 / 4096.0) * 128.0;
    const double c1 = 3424.0 / 4096.0, c2 = (2413.0 / 4096.0) * 32.0, c3 = (2392.0 / 4096.0) * 32.0;
    for (int i = 0; i <= kEotfLutSize; i++) {
        const double t = (double)i / (double)kEotfLutSize;
        double x = t * t;                                  // sampled uniformly in sqrt(x): see convert_block's Dolby Vision stage
        x = std::pow(x, 1.0 / m2);
        x = std::fmax(x - c1, 0.0) / (c2 - c3 * x);
        const double l = x > 0.0 ? std::log2(x) / m1 : -1e9;       // log2 of the EOTF; codes below 7.3e-7 decode to exactly 0
        out[i] = l > -150.0 ? (float)l : -150.0f;          // exp2(-150) == 0 in fp32; a finite floor keeps the interpolation NaN-free
    }
}

void BuildPqEncodeLut(float out[kPqEncSize + 1])
{
    // LinearToST2084(2^t, 1) (st2084.hlsl:18-25) at t = -R + R i / N, in double, rounded once
    const double m1 = 2610.0 / (4096.0 * 4.0), m2 = (2523.0 / 4096.0) * 128.0;
    const double c1 = 3424.0 / 4096.0, c2 = (2413.0 / 4096.0) * 32.0, c3 = (2392.0 / 4096.0) * 32.0;
    for (int i = 0; i <= kPqEncSize; i++) {
        const double t = -(double)kPqEncLog2Range + (double)kPqEncLog2Range * (double)i / (double)kPqEncSize;
        const double z = std::pow(std::exp2(t), m1);
        out[i] = (float)std::pow((c1 + c2 * z) / (1.0 + c3 * z), m2);
    }
}

// ------------------------------------------------------------------------------------------------
// resize weights
// ------------------------------------------------------------------------------------------------
static const float kPi = 3.14159265358979323846f;   // acos(-1.) as an fp32 constant

int UpscaleWeights(int method, float t, float w[6])
{
    const float t2 = t * t, t3 = t * t2;
    switch (method) {
    case MPCVR_UPSCALE_Mitchell: {       // ps_interpolation_spline4.hlsl:50-51
        const float k0[4] = {1.f / 18.f, 16.f / 18.f, 1.f / 18.f, 0.f};
        const float k1[4] = {-.5f, 0.f, .5f, 0.f};
        const float k2[4] = {5.f / 6.f, -12.f / 6.f, 9.f / 6.f, -2.f / 6.f};
        const float k3[4] = {-7.f / 18.f, 21.f / 18.f, -21.f / 18.f, 7.f / 18.f};
        for (int i = 0; i < 4; i++) w[i] = k0[i] + k1[i] * t + k2[i] * t2 + k3[i] * t3;
        return 4;
    }
    case MPCVR_UPSCALE_CatmullRom: {     // ps_interpolation_spline4.hlsl:52-54
        const float k1[4] = {-.5f, 0.f, .5f, 0.f};
        const float k2[4] = {1.f, -2.5f, 2.f, -.5f};
        const float k3[4] = {-.5f, 1.5f, -1.5f, .5f};
        for (int i = 0; i < 4; i++) w[i] = k1[i] * t + k2[i] * t2 + k3[i] * t3;
        w[1] += 1.f;
        return 4;
    }
    case MPCVR_UPSCALE_Lanczos2: {       // ps_interpolation_lanczos2.hlsl:31-56
        if (t == 0.0f) { w[0] = 0; w[1] = 1; w[2] = 0; w[3] = 0; return 4; }
        const float d[4] = {1.f + t, 0.f + t, 1.f - t, 2.f - t};
        for (int i = 0; i < 4; i++) {
            const float a = d[i] * kPi;
            w[i] = crm_sinf(a) * crm_sinf(a * .5f) / (d[i] * d[i] * kPi * kPi * .5f);
        }
        const float wc = 1.f - (w[0] + w[1] + w[2] + w[3]);
        w[1] += wc * (1.f - t);
        w[2] += wc * t;
        return 4;
    }
    case MPCVR_UPSCALE_Lanczos3: {       // ps_interpolation_lanczos3.hlsl:31-64
        if (t == 0.0f) { w[0] = w[1] = 0; w[2] = 1; w[3] = w[4] = w[5] = 0; return 6; }
        float lo[3], hi[3];
        for (int i = 0; i < 3; i++) {
            const float a = (float)(2 - i) * kPi + t * kPi, as = a * .5f;
            const float b = (float)(1 + i) * kPi - t * kPi, bs = b * .5f;
            lo[i] = crm_sinf(a) * crm_sinf(as) / (a * as);
            hi[i] = crm_sinf(b) * crm_sinf(bs) / (b * bs);
        }
        const float wc = 1.f - ((lo[0] + hi[0]) + (lo[1] + hi[1]) + (lo[2] + hi[2]));
        lo[2] += wc * (1.f - t);
        hi[0] += wc * t;
        w[0] = lo[0]; w[1] = lo[1]; w[2] = lo[2]; w[3] = hi[0]; w[4] = hi[1]; w[5] = hi[2];
        return 6;
    }
    case MPCVR_UPSCALE_Spline36_EXT: {   // extension, no reference shader: three-piece cubic over taps base-2 .. base+3, normalised
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

float DownscaleFilter(int method, float x, float *support)
{
    auto sup = [&](float s) { if (support) *support = s; };
    switch (method) {
    case MPCVR_DOWNSCALE_Box:
        sup(0.5f);
        return (x >= -0.5f && x < 0.5f) ? 1.0f : 0.0f;
    case MPCVR_DOWNSCALE_Bilinear:
        sup(1.0f);
        x = x < 0.0f ? -x : x;
        return x < 1.0f ? 1.0f - x : 0.0f;
    case MPCVR_DOWNSCALE_Hamming:
        sup(1.0f);
        x = x < 0.0f ? -x : x;
        if (x == 0.0f) return 1.0f;
        if (x >= 1.0f) return 0.0f;
        x *= kPi;
        return crm_sinf(x) / x * (0.54f + 0.46f * crm_cosf(x));
    case MPCVR_DOWNSCALE_Bicubic:
    case MPCVR_DOWNSCALE_BicubicSharp: {
        const float A = method == MPCVR_DOWNSCALE_Bicubic ? -0.5f : -1.5f;   // compile_shaders.cmd:98-101
        sup(2.0f);
        x = x < 0.0f ? -x : x;
        if (x < 1.0f) return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1;
        if (x < 2.0f) return (((x - 5) * x + 8) * x - 4) * A;
        return 0.0f;
    }
    case MPCVR_DOWNSCALE_Lanczos: {
        sup(3.0f);
        if (!(-3.0f <= x && x < 3.0f)) return 0.0f;
        auto sinc = [](float v) { if (v == 0.0f) return 1.0f; v *= kPi; return crm_sinf(v) / v; };
        return sinc(x) * sinc(x / 3);
    }
    default:
        sup(0.0f);
        return 0.0f;
    }
}

static inline int ClampI(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

bool BuildAxisTaps(Resizer rs, int src_l, int src_len, int n_out, int tex_len, uint32_t flags, HostAxisTaps *out,
                   bool reversed, float shader_scale)
{
    const float step = (float)src_len / (float)n_out;
    const float scale = shader_scale > 0.0f ? shader_scale : step;     // scale[AXIS] of ps_convolution
    auto AxisCenter = [&](int l, int i, float) { return TexCenter(l, src_len, tex_len, i, n_out, reversed ? 1 : 0); };
    out->idx.clear(); out->w.clear(); out->wsum.clear();
    out->normalise = 0;
    if (rs.kind == RS_NONE) {
        out->ntaps = 1;
        out->idx.resize(n_out); out->w.assign(n_out, 1.0f);
        for (int i = 0; i < n_out; i++)
            out->idx[i] = ClampI((int)std::floor(AxisCenter(src_l, i, scale)), 0, tex_len - 1);
        return true;
    }
    if (rs.kind == RS_UP) {
        float probe[6];
        const int n = UpscaleWeights(rs.method, 0.5f, probe);
        if (n != 4 && n != 6) return false;
        out->ntaps = n;
        out->idx.resize((size_t)n_out * n); out->w.resize((size_t)n_out * n);
        // D3D11 Lanczos3 reads Q1 from Q0's texel (ps_interpolation_lanczos3.hlsl:33-34,42-43)
        static const int off4[4] = {-1, 0, 1, 2};
        static const int off6_d3d11[6] = {-2, -2, 0, 1, 2, 3};
        static const int off6_fixed[6] = {-2, -1, 0, 1, 2, 3};
        const int *off = n == 4 ? off4 : ((flags & MPCVR_FLAG_LANCZOS3_FIXED) || rs.method != MPCVR_UPSCALE_Lanczos3) ? off6_fixed : off6_d3d11;
        for (int i = 0; i < n_out; i++) {
            float pos = AxisCenter(src_l, i, scale) - 0.5f;
            const float t = pos - std::floor(pos);
            pos -= t;
            const int base = (int)pos;
            float w[6];
            UpscaleWeights(rs.method, t, w);
            for (int k = 0; k < n; k++) {
                out->idx[(size_t)i * n + k] = ClampI(base + off[k], 0, tex_len - 1);
                out->w[(size_t)i * n + k] = w[k];
            }
        }
        return true;
    }
    // RS_DOWN — ps_convolution.hlsl:23-50
    float support0 = 0;
    DownscaleFilter(rs.method, 0.0f, &support0);
    const float support = support0 * scale;
    const float ss = 1.0f / scale;
    int maxn = 0;
    std::vector<int> lows(n_out), highs(n_out);
    for (int i = 0; i < n_out; i++) {
        const float pos = AxisCenter(src_l, i, scale) + 0.5f;
        lows[i] = (int)std::floor(pos - support);
        highs[i] = (int)std::ceil(pos + support);
        if (highs[i] - lows[i] > maxn) maxn = highs[i] - lows[i];
    }
    if (maxn <= 0 || maxn > 128) return false;
    out->ntaps = maxn;
    out->normalise = 1;
    out->idx.assign((size_t)n_out * maxn, 0); out->w.assign((size_t)n_out * maxn, 0.0f); out->wsum.resize(n_out);
    for (int i = 0; i < n_out; i++) {
        const float pos = AxisCenter(src_l, i, scale) + 0.5f;
        float ww = 0.0f;
        int k = 0;
        for (int n = lows[i]; n < highs[i]; n++, k++) {
            const float w = DownscaleFilter(rs.method, ((float)n - pos + 0.5f) * ss, nullptr);
            ww += w;
            out->idx[(size_t)i * maxn + k] = ClampI(n, 0, tex_len - 1);
            out->w[(size_t)i * maxn + k] = w;
        }
        for (; k < maxn; k++) out->idx[(size_t)i * maxn + k] = ClampI(lows[i], 0, tex_len - 1);   // w = 0 padding
        out->wsum[i] = ww;
    }
    return true;
}

void BuildPointIndex(int src_l, int src_len, int n_out, int tex_len, std::vector<int32_t> *out, bool reversed)
{
    out->resize(n_out);
    for (int i = 0; i < n_out; i++)
        (*out)[i] = ClampI((int)std::floor(TexCenter(src_l, src_len, tex_len, i, n_out, reversed ? 1 : 0)), 0, tex_len - 1);
}

// Strip geometry of the arbitrary-ratio fused kernel.  The kernel converts a strip's source window in passes of 64 2x2 blocks
// per source row pair and filters 64 * pxl output columns from it, so the width of a strip trades idle convert lanes against
// idle filter lanes; the choice below minimises an instruction-count model of one output pixel (convert ~250 VALU per block
// pass, 3 per tap and channel set in the X / Y stages), which is what the kernel is bound by.
bool PlanFusedStrip(const HostAxisTaps &hx, const HostAxisTaps &hy, int n_out_x, int n_out_y, int src_w, int src_h, StripPlan *sp)
{
    if (hx.ntaps < 1 || hx.ntaps > 16 || hy.ntaps < 1 || hy.ntaps > 16 || n_out_x < 1 || n_out_y < 1) return false;
    if (hx.idx.size() != (size_t)n_out_x * hx.ntaps || hy.idx.size() != (size_t)n_out_y * hy.ntaps) return false;
    sp->yrange.resize(2 * (size_t)n_out_y);
    int span = 0, plo = 0, phi = 0;
    for (int y = 0; y < n_out_y; y++) {
        const auto mm = std::minmax_element(hy.idx.begin() + (size_t)y * hy.ntaps, hy.idx.begin() + (size_t)(y + 1) * hy.ntaps);
        const int lo = *mm.first, hi = *mm.second;
        if (lo < 0 || hi >= src_h || (y > 0 && (lo < plo || hi < phi))) return false;      // the march needs monotonic windows
        sp->yrange[2 * y] = lo; sp->yrange[2 * y + 1] = hi;
        span = std::max(span, hi - lo + 1);
        plo = lo; phi = hi;
    }
    sp->ring = span + 1 <= 8 ? 8 : span + 1 <= 16 ? 16 : span + 1 <= 32 ? 32 : 0;
    if (!sp->ring) return false;
    std::vector<int> clo(n_out_x), chi(n_out_x);
    for (int x = 0; x < n_out_x; x++) {
        const auto mm = std::minmax_element(hx.idx.begin() + (size_t)x * hx.ntaps, hx.idx.begin() + (size_t)(x + 1) * hx.ntaps);
        clo[x] = *mm.first; chi[x] = *mm.second;
        if (clo[x] < 0 || chi[x] >= src_w) return false;
    }
    const int mt = std::max(hx.ntaps, hy.ntaps);
    sp->nt = mt <= 4 ? 4 : mt <= 6 ? 6 : mt <= 8 ? 8 : 16;       // 9..16 taps (ps_convolution beyond ~2x with bicubic / Lanczos): one pixel per lane only
    if (sp->ring == 32 && sp->nt != 16) return false;           // (a 32-row ring exists for the 16-tap kernels)
    const double pairs_per_row = 0.5 * (double)src_h / (double)n_out_y;
    double best = 0;
    int best_w = 0, best_cols = 0, best_pxl = 1;
    for (int pxl : {1, 2})
        for (int lanes : {64, 56, 48, 40, 32, 24, 16}) {
            if ((pxl == 2) == (sp->nt == 16)) continue;     // built as: two pixels per lane up to 8 taps, one for 16 (a one-pixel strip never won the model below 9 taps)
            const int sw = lanes * pxl;
            int max_nb = 0;
            for (int x0 = 0; x0 < n_out_x; x0 += sw) {
                const int x1 = std::min(n_out_x, x0 + sw);
                const int lo = *std::min_element(clo.begin() + x0, clo.begin() + x1), hi = *std::max_element(chi.begin() + x0, chi.begin() + x1);
                max_nb = std::max(max_nb, ((hi - (lo & ~1)) >> 1) + 1);
            }
            const int passes = (max_nb + 63) / 64;
            // VALU instructions of a lane (counted in the ISA): convert pass 175, X stage pxl * (6 nt + 3) + 6 per row pair,
            // Y stage pxl * (3 nt + 7) + nt + 6 per output row
            const double cost = (pairs_per_row * (passes * 175.0 + pxl * (6.0 * sp->nt + 3.0) + 6.0) + pxl * (3.0 * sp->nt + 7.0) + sp->nt + 6.0) / (double)sw;
            if (!best_w || cost < best) { best = cost; best_w = sw; best_cols = 2 * max_nb; best_pxl = pxl; }
        }
    sp->pxl = best_pxl;
    sp->strip_w = best_w; sp->acols = best_cols;
    const int n_strips = (n_out_x + best_w - 1) / best_w;
    sp->xstrip.resize(2 * (size_t)n_strips);
    for (int s = 0; s < n_strips; s++) {
        const int x0 = s * best_w, x1 = std::min(n_out_x, x0 + best_w);
        sp->xstrip[2 * s] = *std::min_element(clo.begin() + x0, clo.begin() + x1);
        sp->xstrip[2 * s + 1] = *std::max_element(chi.begin() + x0, chi.begin() + x1);
    }
    // the tables as the kernel reads them
    const int nt = sp->nt;
    sp->xi_t.assign((size_t)nt * n_out_x, 0); sp->xw_t.assign((size_t)nt * n_out_x, 0.0f);
    for (int x = 0; x < n_out_x; x++) {
        const float ww = hx.normalise ? hx.wsum[x] : 1.0f;
        for (int k = 0; k < nt; k++) {
            const bool on = k < hx.ntaps;
            sp->xi_t[(size_t)k * n_out_x + x] = hx.idx[(size_t)x * hx.ntaps + (on ? k : 0)];
            sp->xw_t[(size_t)k * n_out_x + x] = on ? (hx.normalise ? hx.w[(size_t)x * hx.ntaps + k] / ww : hx.w[(size_t)x * hx.ntaps + k]) : 0.0f;
        }
    }
    sp->yi.assign((size_t)nt * n_out_y, 0); sp->yw.assign((size_t)nt * n_out_y, 0.0f);
    for (int y = 0; y < n_out_y; y++) {
        const float ww = hy.normalise ? hy.wsum[y] : 1.0f;
        for (int k = 0; k < nt; k++) {
            const bool on = k < hy.ntaps;
            sp->yi[(size_t)y * nt + k] = hy.idx[(size_t)y * hy.ntaps + (on ? k : 0)];
            sp->yw[(size_t)y * nt + k] = on ? (hy.normalise ? hy.w[(size_t)y * hy.ntaps + k] / ww : hy.w[(size_t)y * hy.ntaps + k]) : 0.0f;
        }
    }
    return true;
}

// Vertical phase pattern of the periodic-phase kernel: output row y = PB*m + r reads source rows 6m + base(r) + off(t), PB = 6P/Q
// (the same integer arithmetic as vp_fused_period.h's period_base; checked against the real table below, so a disagreement can
// only cost the fast path, never a wrong pixel)
static int PeriodBase(int P, int Q, int r)
{
    const int num = (2 * r + 1) * Q - P, den = 2 * P;
    return num >= 0 ? num / den : -((-num + den - 1) / den);
}

bool PlanFusedPeriod(const HostAxisTaps &hx, const HostAxisTaps &hy, int n_out_x, int n_out_y, int src_w, int src_h, bool fold_q1, PeriodPlan *pp, bool heavy_convert)
{
    pp->P = pp->Q = 0;
    if (hx.normalise || hy.normalise) return false;                       // interpolation shaders only (ps_convolution normalises)
    if ((hy.ntaps != 4 && hy.ntaps != 6) || hx.ntaps != hy.ntaps) return false;
    if (n_out_x < 2 || n_out_y < 1 || hx.idx.size() != (size_t)n_out_x * hx.ntaps || hy.idx.size() != (size_t)n_out_y * hy.ntaps) return false;
    static const int ratios[][2] = {{4, 3}, {3, 2}, {2, 3}, {1, 2}, {3, 1}};
    int P = 0, Q = 0;
    for (const auto &r : ratios)
        if ((long)n_out_y * r[1] == (long)src_h * r[0]) { P = r[0]; Q = r[1]; }
    if (!P) return false;
    const int nth = hy.ntaps;
    if (fold_q1 && nth != 6) return false;
    // tap offsets of the table's columns relative to base: {-1..2}, {-2..3}, or Direct3D 11 Lanczos3's {-2, -2, 0, 1, 2, 3}
    static const int off4[4] = {-1, 0, 1, 2}, off6[6] = {-2, -1, 0, 1, 2, 3}, off6q[6] = {-2, -2, 0, 1, 2, 3};
    const int *off = nth == 4 ? off4 : fold_q1 ? off6q : off6;
    const int PB = 6 * P / Q;
    // P and Q both odd (3:1): output rows with ((2r + 1) Q - P) % 2P == 0 sit exactly on a texel centre n, where the reference's fp32
    // texcoord lands on either side row by row: the table holds base n with t = 0, or base n - 1 with t = 1 - eps.  The kernel has both
    // emissions compiled in for those rows and picks by a bit per row: `below`, one word per body, rides in the spare slot 7 of the
    // body's first yw row
    std::vector<uint32_t> below((n_out_y + PB - 1) / PB, 0u);
    for (int y = 0; y < n_out_y; y++) {
        const int m = y / PB, r = y % PB, base = 6 * m + PeriodBase(P, Q, r);
        const bool centre = ((2 * r + 1) * Q - P) % (2 * P) == 0;
        bool fits = false;
        for (int sh = 0; sh <= (centre ? 1 : 0) && !fits; sh++) {
            fits = true;
            for (int k = 0; k < nth && fits; k++) fits = hy.idx[(size_t)y * nth + k] == ClampI(base - sh + off[k], 0, src_h - 1);
            if (fits && sh) below[m] |= 1u << r;
        }
        if (!fits) return false;
    }
    if (fold_q1)
        for (int x = 0; x < n_out_x; x++)
            if (hx.idx[(size_t)x * 6] != hx.idx[(size_t)x * 6 + 1]) return false;
    const int nt = fold_q1 ? 5 : nth;
    pp->nt = nt;
    // the tables as the kernel reads them
    pp->xi_t.assign((size_t)nt * n_out_x, 0); pp->xw_t.assign((size_t)nt * n_out_x, 0.0f);
    for (int x = 0; x < n_out_x; x++)
        for (int k = 0; k < nt; k++) {
            const int ks = fold_q1 && k > 0 ? k + 1 : k;
            pp->xi_t[(size_t)k * n_out_x + x] = hx.idx[(size_t)x * nth + ks];
            pp->xw_t[(size_t)k * n_out_x + x] = (fold_q1 && k == 0) ? hx.w[(size_t)x * nth] + hx.w[(size_t)x * nth + 1] : hx.w[(size_t)x * nth + ks];
        }
    pp->yw.assign((size_t)n_out_y * 8, 0.0f);
    for (int y = 0; y < n_out_y; y++)
        for (int k = 0; k < nt; k++) {
            const int ks = fold_q1 && k > 0 ? k + 1 : k;
            pp->yw[(size_t)y * 8 + k] = (fold_q1 && k == 0) ? hy.w[(size_t)y * nth] + hy.w[(size_t)y * nth + 1] : hy.w[(size_t)y * nth + ks];
        }
    for (size_t m = 0; m < below.size(); m++) std::memcpy(&pp->yw[m * PB * 8 + 7], &below[m], 4);
    // strip width: a convert pass costs the same whether 33 or 64 of its lanes hold a block, so the width is chosen to fill the passes —
    // an instruction-count model of one body (three source row pairs): passes x convert + X stage per pair, Y stage + epilogue per row
    // (counted in the ISA: convert ~230 with a table tail, ~70 without; X ~50 per pair; Y + final pass ~30 per row), per output pixel
    std::vector<int> clo(n_out_x), chi(n_out_x);
    for (int x = 0; x < n_out_x; x++) {
        const auto mm = std::minmax_element(hx.idx.begin() + (size_t)x * nth, hx.idx.begin() + (size_t)(x + 1) * nth);
        clo[x] = *mm.first; chi[x] = *mm.second;
        if (clo[x] < 0 || chi[x] >= src_w) return false;
    }
    auto blocks_of = [&](int sw) {
        int max_nb = 0;
        for (int x0 = 0; x0 < n_out_x; x0 += sw) {
            const int x1 = std::min(n_out_x, x0 + sw);
            const int lo = *std::min_element(clo.begin() + x0, clo.begin() + x1), hi = *std::max_element(chi.begin() + x0, chi.begin() + x1);
            max_nb = std::max(max_nb, ((hi - (lo & ~1)) >> 1) + 1);
        }
        return max_nb;
    };
    int sw = 128;
    {
        double best = 0;
        const double conv = heavy_convert ? 230.0 : 70.0;
        for (int lanes = 64; lanes >= 24; lanes--) {
            const int w = 2 * lanes;
            const int passes = (blocks_of(w) + 63) / 64;
            const double cost = (3.0 * (passes * conv + 50.0) + PB * 30.0) / ((double)PB * w);
            if (best == 0 || cost < best * 0.985) { best = cost; sw = w; }       // a narrower strip must pay for its idle filter lanes by >= 1.5 %
        }
    }
    pp->strip_w = sw;
    const int n_strips = (n_out_x + sw - 1) / sw;
    pp->xstrip.resize(2 * (size_t)n_strips);
    int max_cols = 0;
    for (int s = 0; s < n_strips; s++) {
        const int x0 = s * sw, x1 = std::min(n_out_x, x0 + sw);
        const int lo = *std::min_element(clo.begin() + x0, clo.begin() + x1), hi = *std::max_element(chi.begin() + x0, chi.begin() + x1);
        pp->xstrip[2 * s] = lo; pp->xstrip[2 * s + 1] = hi;
        max_cols = std::max(max_cols, (((hi - (lo & ~1)) >> 1) + 1) * 2);
    }
    pp->acols = max_cols;
    // lane -> output column ownership: the one whose X-stage reads (one ds_read_b64 per tap, pixel and channel at A + 24 * column) cost the
    // fewest LDS cycles, counted exactly on a strip in the middle of the frame: per 32-lane group one cycle + one per extra distinct
    // column on the busiest 8-byte slot (slot = 3 * column mod 32).  The adjacent pair (0) is not a candidate: its two dword stores
    // would interleave inside every cache line.
    {
        const int s = n_strips / 2, x0 = s * sw, lo = pp->xstrip[2 * s] & ~1;
        long best = -1;
        for (int own = 1; own < 3; own++) {
            long cycles = 0;
            for (int q = 0; q < 2; q++)
                for (int k = 0; k < nt; k++)
                    for (int g = 0; g < 2; g++) {
                        int cols[32], n = 0;
                        for (int i = 0; i < 32; i++) {
                            const int x = std::min(x0 + PeriodLaneColumn(own, 32 * g + i, q), n_out_x - 1);
                            const int c = pp->xi_t[(size_t)k * n_out_x + x] - lo;
                            bool dup = false;
                            for (int j = 0; j < n && !dup; j++) dup = cols[j] == c;
                            if (!dup) cols[n++] = c;
                        }
                        int slot[32] = {0}, worst = 1;
                        for (int j = 0; j < n; j++) worst = std::max(worst, ++slot[(3 * cols[j]) & 31]);
                        cycles += worst;
                    }
            if (best < 0 || cycles < best) { best = cycles; pp->own = own; }
        }
    }
    pp->P = P; pp->Q = Q;
    return true;
}

bool DecidePlan(int iTexFormat, int iChromaScaling, int iUpscaling, int iDownscaling, int bInterpolateAt50pct,
                int bUseDither, int output_format, uint32_t flags, const FmtConvParams &f,
                const PlanGeometry &g, PassPlan *plan, std::string *why)
{
    PassPlan p;
    switch (iTexFormat) {                                   // UpdateTexParams :1143-1155
    case MPCVR_TEXFMT_8INT: p.internal_fmt = SF_BGRA8; break;
    case MPCVR_TEXFMT_10INT: p.internal_fmt = SF_RGB10A2; break;
    case MPCVR_TEXFMT_16FLOAT: p.internal_fmt = SF_RGBA16F; break;
    default: p.internal_fmt = f.CDepth > 8 ? SF_RGB10A2 : SF_BGRA8; break;
    }
    // EXTENSION, bUseDither = 2 (no reference counterpart): where the reference's final pass would dither into an 8-bit target, render
    // as for a 10-bit swap chain and let the error-diffusion pass quantise (include/mpcvr.h, vp_errdiff_core.h)
    p.errdiff = bUseDither == MPCVR_DITHER_ErrorDiffusion_EXT && output_format != MPCVR_OUT_RGB10A2 && p.internal_fmt != SF_BGRA8;
    if (p.errdiff) output_format = MPCVR_OUT_RGB10A2;
    p.swap_fmt = output_format == MPCVR_OUT_RGB10A2 ? SF_RGB10A2 : SF_BGRA8;
    const bool needDither = (p.swap_fmt == SF_BGRA8 && p.internal_fmt != SF_BGRA8) ||
                            (p.swap_fmt == SF_RGB10A2 && p.internal_fmt == SF_RGBA16F);   // :2896-2900
    p.final_pass = bUseDither && needDither;
    p.quant = p.swap_fmt == SF_RGB10A2 ? 1023 : 255;

    const int w2 = g.vr - g.vl, h2 = g.vb - g.vt;
    if (g.rotation != 0 && g.rotation != 90 && g.rotation != 180 && g.rotation != 270) { if (why) *why = "rotation must be 0, 90, 180 or 270"; return false; }
    p.rotation = g.rotation; p.flip = g.flip != 0;
    p.convert = g.convert_enabled != 0;
    p.hdr_tonemap = g.hdr_tonemap != 0;
    const bool rotated = g.rotation == 90 || g.rotation == 270;
    const int w1 = rotated ? g.h1 : g.w1, h1 = rotated ? g.w1 : g.h1;                     // :3112-3123
    const int k = bInterpolateAt50pct ? 2 : 1;                                            // :3108
    const Resizer up{iUpscaling == MPCVR_UPSCALE_Nearest ? RS_NONE : RS_UP, iUpscaling};
    const Resizer down{RS_DOWN, iDownscaling};
    const Resizer none{RS_NONE, 0};
    p.rx = (w1 == w2) ? none : (w1 > k * w2) ? down : up;                                 // :3125-3126 (screen x)
    p.ry = (h1 == h2) ? none : (h1 > k * h2) ? down : up;
    // rotated: resizerX and (when it exists) resizerY are both Y shaders; equal shader objects => ONE draw (:3131-3137)
    // ... or Jinc2, whose single 2-D shader serves both axes (m_pShaderUpscaleY = m_pShaderUpscaleX, :2921)
    const bool same_shader = p.rx.kind != RS_NONE && p.rx.kind == p.ry.kind &&
                             (rotated || (p.rx.kind == RS_UP && iUpscaling == MPCVR_UPSCALE_Jinc2));
    p.two_pass = p.rx.kind != RS_NONE && p.ry.kind != RS_NONE && !same_shader;
    // Process :3348-3352: with a final pass the resize step is skipped only when rSrc == dstRect and rotation == 0 (a
    // flip alone is then ignored); without post-scale steps ResizeShaderPass always runs (:3417-3419)
    const bool same_rect = g.w1 == w2 && g.h1 == h2 && g.vl == 0 && g.vt == 0;
    const bool has_steps = p.final_pass || p.hdr_tonemap;                 // GetPostScaleSteps() > 0
    const bool need_draw = w1 != w2 || h1 != h2 || g.rotation != 0 || (p.flip && !(has_steps && same_rect));
    p.one_pass = !p.two_pass && need_draw;
    p.mid_h = h1;
    if (p.two_pass) { p.first_tex_axis = rotated ? 1 : 0; p.first_rs = p.rx; }
    else if (p.one_pass) {
        if (p.rx.kind != RS_NONE) { p.first_tex_axis = rotated ? 1 : 0; p.first_rs = p.rx; }
        else if (p.ry.kind != RS_NONE) { p.first_tex_axis = rotated ? 0 : 1; p.first_rs = p.ry; }
        else { p.first_tex_axis = -1; p.first_rs = none; }
        p.one_pass_axis = (p.rx.kind != RS_NONE || p.ry.kind == RS_NONE) ? 0 : 1;       // screen axis carrying the taps
    }
    p.copy_only = !p.two_pass && !p.one_pass;
    // layouts / chroma filters the block convert inside the fused kernels serves (BlockConvertLayout, vp_fused.hip)
    const bool fused_layout = f.layout == LAY_PLANAR ? (f.Subsampling == 444 ? f.planes == 3 : (f.Subsampling == 420 || f.Subsampling == 422) && iChromaScaling != MPCVR_CHROMA_CatmullRom)
                            : f.layout == LAY_PACKED422 ? iChromaScaling != MPCVR_CHROMA_CatmullRom
                            : (f.layout == LAY_PACKED444 || f.layout == LAY_GRAY);
    // fused 2x candidate: exact 2x on both axes with an interpolation shader, bilinear 4:2:0 chroma,
    // UNORM internal format, destination fully inside the window
    // ... or the one 2-D draw of Jinc2m (same_shader above), which the fused kernel of vp_fused_jinc.hip replaces together with the convert draw
    const bool jinc_draw = p.one_pass && !rotated && p.rx.kind == RS_UP && p.ry.kind == RS_UP && iUpscaling == MPCVR_UPSCALE_Jinc2;
    p.fused_up2x = !(flags & MPCVR_FLAG_NO_FUSED) && !g.dovi && g.rotation == 0 && !p.flip && !p.hdr_tonemap && (p.two_pass || jinc_draw) && w2 == 2 * w1 && h2 == 2 * h1 &&
                   p.rx.kind == RS_UP && p.ry.kind == RS_UP && fused_layout && p.internal_fmt != SF_RGBA16F &&
                   g.vl >= 0 && g.vt >= 0 && g.vr <= g.ww && g.vb <= g.wh && w1 >= 8 && h1 >= 8 && !(w1 & 1);
    p.fused_jinc = p.fused_up2x && jinc_draw;
    p.direct_convert = !(flags & MPCVR_FLAG_NO_FUSED) && p.copy_only && p.convert && !p.hdr_tonemap;
    *plan = p;
    return true;
}

std::string PassPlan::describe() const
{
    if (fused_up2x) return std::string(fused_jinc ? "fused_jinc2x" : "fused_up2x") + (errdiff ? ",errdiff" : "");
    if (direct_convert) return std::string(final_pass ? "direct:convert+final" : "direct:convert+copy") + (errdiff ? ",errdiff" : "");
    std::string s = convert ? "passes:convert" : "passes:source";
    if (two_pass) s += final_pass ? ",resizeX,resizeY+final" : ",resizeX,resizeY";
    else if (one_pass) s += std::string(one_pass_axis == 0 ? ",resizeX" : ",resizeY") + (final_pass ? "+final" : "");
    else s += final_pass ? ",final" : ",copy";
    if (hdr_tonemap) s += ",hdr10tonemap";
    if (errdiff) s += ",errdiff";
    if (rotation) s += ";rot" + std::to_string(rotation);
    if (flip) s += ";flip";
    return s;
}

}  // namespace mpcvr
