// vp_dovi.cpp — host side of the Dolby Vision part of the frame path: what CopySample / SetShaderDoviCurves /
// SetShaderConvertColorParams / SetDolbyVisionDynamicParams compute on the CPU from MediaSideDataDOVIMetadata before a
// frame is drawn (DX11VideoProcessor.cpp:813-834, 954-960, 990-1141, 2270-2475; Shaders.cpp:826-842).
#include "vp_plan.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace mpcvr {

// RPU fields arrive from untrusted media: everything that indexes or shifts below is range-checked here (the reference checks
// num_pivots / mapping_idc only, DX11VideoProcessor.cpp:2279-2322; its shader loops are bounded by the cbuffer layout instead)
bool CheckDoviCurves(const mpcvr_dovi_metadata &md)
{
    if (md.coef_log2_denom > 31 || md.bl_bit_depth < 8 || md.bl_bit_depth > 16) return false;      // 1 << n below
    for (const auto &curve : md.curves) {
        if (curve.num_pivots < 2 || curve.num_pivots > 9) return false;
        for (int i = 0; i < int(curve.num_pivots - 1); i++) {
            if (curve.mapping_idc[i] > 1) return false;
            if (curve.mapping_idc[i] == 0 && curve.poly_order[i] > 2) return false;                  // poly_coef[piece][3]
            if (curve.mapping_idc[i] == 1 && (curve.mmr_order[i] < 1 || curve.mmr_order[i] > 3)) return false;   // mmr_coef[piece][3][7]
        }
    }
    return md.n_l2 <= 32;
}

void PackDoviCurves(const mpcvr_dovi_metadata &md, DoviParams *dst)
{
    std::memset(dst->curves, 0, sizeof(dst->curves));
    dst->has_mmr = 0;
    const float coefScale = 1.0f / (float)(1u << (md.coef_log2_denom & 31));
    const float codeScale = 1.0f / (float)((1u << (md.bl_bit_depth & 31)) - 1);
    for (int c = 0; c < 3; c++) {
        const mpcvr_dovi_curve &in = md.curves[c];
        DoviCurve &cv = dst->curves[c];
        bool anyPoly = false, anyMmr = false, single = true;
        uint32_t slot = 0;
        int loOrder = 3, hiOrder = 1;
        for (int piece = 0; piece + 1 < in.num_pivots; piece++) {
            float *co = cv.coeffs[piece];
            if (in.mapping_idc[piece] == 0) {
                anyPoly = true;
                const int order = in.poly_order[piece];
                co[0] = coefScale * in.poly_coef[piece][0];
                co[1] = order >= 1 ? coefScale * in.poly_coef[piece][1] : 0.0f;
                co[2] = order >= 2 ? coefScale * in.poly_coef[piece][2] : 0.0f;
                co[3] = 0.0f;                                   // order 0 marks a polynomial piece
            } else if (in.mapping_idc[piece] == 1) {
                const int order = in.mmr_order[piece];
                loOrder = std::min(loOrder, order);
                hiOrder = std::max(hiOrder, order);
                single = !anyMmr;                               // true only while exactly one MMR piece was seen
                anyMmr = true;
                co[0] = coefScale * in.mmr_constant[piece];
                co[1] = (float)slot;                            // first float4 of this piece's weights
                co[3] = (float)order;
                for (int o = 0; o < order && o < 3 && slot + 2 <= 48; o++) {    // two float4 per order: (3 + pad) and 4 weights; 48 = mmr[]
                    const int64_t *w = in.mmr_coef[piece][o];
                    float *a = cv.mmr[slot++], *b = cv.mmr[slot++];
                    a[0] = coefScale * w[0]; a[1] = coefScale * w[1]; a[2] = coefScale * w[2]; a[3] = 0.0f;
                    b[0] = coefScale * w[3]; b[1] = coefScale * w[4]; b[2] = coefScale * w[5]; b[3] = coefScale * w[6];
                }
            }
        }
        const int inner = in.num_pivots - 2;                    // the outermost pivots are not tested
        for (int i = 0; i < 7; i++) cv.pivots[i] = i < inner ? codeScale * in.pivots[i + 1] : 1e9f;
        if (anyPoly) cv.methods = DOVI_RESHAPE_POLY;
        if (anyMmr) {
            cv.methods |= DOVI_RESHAPE_MMR;
            cv.mmr_single = single ? 1u : 0u;
            cv.min_order = (uint32_t)loOrder;
            cv.max_order = (uint32_t)hiOrder;
            dst->has_mmr = 1;
        }
    }
}

void DoviLmsMatrix(const mpcvr_dovi_metadata &md, float out[9])
{
    static const float kLmsToRgb[3][3] = {
        { 3.06441879f, -2.16597676f,  0.10155818f},
        {-0.65612108f,  1.78554118f, -0.12943749f},
        { 0.01736321f, -0.04725154f,  1.03004253f},
    };
    float lin[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) lin[r][c] = (float)md.rgb_to_lms_matrix[r * 3 + c];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            out[r * 3 + c] = kLmsToRgb[r][0] * lin[0][c] + kLmsToRgb[r][1] * lin[1][c] + kLmsToRgb[r][2] * lin[2][c];
}

namespace {
// the PQ helpers CopySample defines as lambdas (:2324-2345)
const float kM1 = 2610.f / (4096.f * 4.f), kM2 = 2523.f / 4096.f * 128.f;
const float kC1 = 3424.f / 4096.f, kC2 = 2413.f / 4096.f * 32.f, kC3 = 2392.f / 4096.f * 32.f;

float PqToLinearNits(float x)
{
    x = powf(x, 1.0f / kM2);
    x = fmaxf(x - kC1, 0.0f) / (kC2 - kC3 * x);
    x = powf(x, 1.0f / kM1);
    return x * 10000.0f;
}
float LinearNitsToPq(float y)
{
    y /= 10000.0f;
    y = fmaxf(y, 0.0f);
    y = powf(y, kM1);
    y = (kC1 + kC2 * y) / (1.0f + kC3 * y);
    return powf(y, kM2);
}
}  // namespace

bool DoviL1Nits(const mpcvr_dovi_metadata &md, uint32_t out[3])
{
    out[0] = out[1] = out[2] = 0;
    if (!md.l1_present) return false;
    uint32_t v[3] = {md.l1_min_pq, md.l1_max_pq, md.l1_avg_pq};
    if (md.l3_present) {
        const uint32_t off[3] = {md.l3_min_pq_offset, md.l3_max_pq_offset, md.l3_avg_pq_offset};
        for (int i = 0; i < 3; i++) v[i] = v[i] + off[i] - 2048;
    }
    for (int i = 0; i < 3; i++) out[i] = static_cast<uint32_t>(PqToLinearNits(v[i] / 4095.f));
    return true;
}

bool DoviL2Constants(const mpcvr_dovi_metadata &md, int display_nits, float k[5])
{
    const float displayPq = LinearNitsToPq((float)display_nits);
    const int count = (int)std::min<uint32_t>(md.n_l2, 32u);
    int below = -1, above = -1;
    float gapBelow = 1.0f, gapAbove = 1.0f;
    for (int i = 0; i < count; i++) {
        const float targetPq = md.l2[i].target_max_pq / 4095.0f;
        if (targetPq <= displayPq) {
            if (displayPq - targetPq < gapBelow) { gapBelow = displayPq - targetPq; below = i; }
        } else {
            if (targetPq - displayPq < gapAbove) { gapAbove = targetPq - displayPq; above = i; }
        }
    }
    // L2 {chroma_weight, saturation_gain, slope, offset, power}; zero until a level-2 block was seen
    float l2[5] = {0, 0, 0, 0, 0};
    if (count > 0) {
        auto raw = [](const mpcvr_dovi_l2 &e, float o[5]) {
            o[0] = e.trim_chroma_weight; o[1] = e.trim_saturation_gain; o[2] = e.trim_slope; o[3] = e.trim_offset; o[4] = e.trim_power;
        };
        float t[5] = {0.0f, 0.0f, 1.0f, 0.0f, 1.0f};
        if (below >= 0 && above >= 0) {             // the display sits between two targets
            float a[5], b[5];
            raw(md.l2[below], a); raw(md.l2[above], b);
            const float lowPq = md.l2[below].target_max_pq / 4095.0f, highPq = md.l2[above].target_max_pq / 4095.0f;
            const float w = std::clamp(highPq != lowPq ? (displayPq - lowPq) / (highPq - lowPq) : 0.0f, 0.0f, 1.0f);
            for (int i = 0; i < 5; i++) t[i] = std::lerp(a[i], b[i], w);
        } else if (below >= 0) {                    // brighter than every target: blend towards neutral (2048)
            float a[5];
            raw(md.l2[below], a);
            const float masterPq = md.source_max_pq / 4095.0f, lowPq = md.l2[below].target_max_pq / 4095.0f;
            const float w = std::clamp(masterPq > lowPq ? (displayPq - lowPq) / (masterPq - lowPq) : 0.0f, 0.0f, 1.0f);
            for (int i = 0; i < 5; i++) t[i] = std::lerp(a[i], 2048.0f, w);
        } else if (above >= 0) {                    // dimmer than every target: the lowest one as is
            raw(md.l2[above], t);
        }
        for (int i = 0; i < 5; i++) l2[i] = t[i] / 4096.0f;
    }
    k[0] = l2[0] - 0.5f; k[1] = l2[1] - 0.5f; k[2] = l2[2] + 0.5f; k[3] = l2[3] - 0.5f; k[4] = l2[4] + 0.5f;
    return count > 0;
}

void DoviColorMatrix(const mpcvr_dovi_metadata &md, const FmtConvParams &f, const ProcAmp &pa, float out[12])
{
    const float brightness = pa.brightness / 255;
    const float contrast = pa.contrast;
    float m[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) m[r][c] = (float)md.ycc_to_rgb_matrix[r * 3 + c] * contrast;
    for (int r = 0; r < 3; r++) {
        float acc = brightness;
        for (int c = 0; c < 3; c++) acc = (float)(acc - m[r][c] * md.ycc_to_rgb_offset[c]);     // float -= float * double
        out[9 + r] = acc;
    }
    if (f.CSType == CST_RGB && f.layout == LAY_PLANAR && f.planes == 3) {
        for (auto &row : m) { const float x = row[0], y = row[1], z = row[2]; row[0] = y; row[1] = z; row[2] = x; }
    } else if (f.CSType == CST_GRAY) {
        m[1][0] = m[1][1]; m[1][1] = 0;
        m[2][0] = m[2][2]; m[2][2] = 0;
    }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) out[r * 3 + c] = m[r][c];
}

// ------------------------------------------------------------------------------------------------
// constants of the correction shaders: mul(ycbcr2020nc_rgb, rgb_ycbcr709), mul(ycgco_rgb, rgb_ycbcr709) (conv_matrix.hlsl) and
// the BT.2020 -> BT.709 primaries matrix the HLSL derives from the chromaticities (colorspace_gamut_conversion.hlsl, after zimg)
// ------------------------------------------------------------------------------------------------
namespace {
struct M3 { float v[3][3]; };
float Det2(float a, float b, float c, float d) { return a * d - b * c; }
M3 Inverse3(const M3 &m)
{
    float det = 0;
    det += m.v[0][0] * Det2(m.v[1][1], m.v[1][2], m.v[2][1], m.v[2][2]);
    det -= m.v[0][1] * Det2(m.v[1][0], m.v[1][2], m.v[2][0], m.v[2][2]);
    det += m.v[0][2] * Det2(m.v[1][0], m.v[1][1], m.v[2][0], m.v[2][1]);
    M3 r;
    r.v[0][0] = Det2(m.v[1][1], m.v[1][2], m.v[2][1], m.v[2][2]) / det;
    r.v[0][1] = Det2(m.v[0][2], m.v[0][1], m.v[2][2], m.v[2][1]) / det;
    r.v[0][2] = Det2(m.v[0][1], m.v[0][2], m.v[1][1], m.v[1][2]) / det;
    r.v[1][0] = Det2(m.v[1][2], m.v[1][0], m.v[2][2], m.v[2][0]) / det;
    r.v[1][1] = Det2(m.v[0][0], m.v[0][2], m.v[2][0], m.v[2][2]) / det;
    r.v[1][2] = Det2(m.v[0][2], m.v[0][0], m.v[1][2], m.v[1][0]) / det;
    r.v[2][0] = Det2(m.v[1][0], m.v[1][1], m.v[2][0], m.v[2][1]) / det;
    r.v[2][1] = Det2(m.v[0][1], m.v[0][0], m.v[2][1], m.v[2][0]) / det;
    r.v[2][2] = Det2(m.v[0][0], m.v[0][1], m.v[1][0], m.v[1][1]) / det;
    return r;
}
void XyToXyz(float x, float y, float o[3]) { o[0] = x / y; o[1] = 1.0f; o[2] = (1.0f - x - y) / y; }
M3 RgbToXyz(const float prim[3][2])
{
    float col[3][3], white[3];
    for (int i = 0; i < 3; i++) XyToXyz(prim[i][0], prim[i][1], col[i]);
    M3 xyz;                                     // columns R, G, B
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) xyz.v[r][c] = col[c][r];
    XyToXyz(0.3127f, 0.3290f, white);
    const M3 inv = Inverse3(xyz);
    float s[3];
    for (int i = 0; i < 3; i++) s[i] = inv.v[i][0] * white[0] + inv.v[i][1] * white[1] + inv.v[i][2] * white[2];
    M3 m;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) m.v[r][c] = xyz.v[r][c] * s[c];
    return m;
}
void Mul4(const float a[16], const float b[16], float c[16])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            c[i * 4 + j] = a[i * 4] * b[j] + a[i * 4 + 1] * b[4 + j] + a[i * 4 + 2] * b[8 + j] + a[i * 4 + 3] * b[12 + j];
}
}  // namespace

void CorrectionMatrices(float fix2020[16], float fixycgco[16], float gamut[9])
{
    static const float kRgbToYcbcr709[16] = {0.2126f, 0.7152f, 0.0722f, 0.0f, -0.114572f, -0.385428f, 0.5f, 0.0f,
                                             0.5f, -0.454153f, -0.045847f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    static const float kYcbcr2020ToRgb[16] = {1.0f, 0.0f, 1.4746f, 0.0f, 1.0f, -0.164553f, -0.571353f, 0.0f,
                                              1.0f, 1.8814f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    static const float kYcgcoToRgb[16] = {1.0f, -1.0f, 1.0f, 0.0f, 1.0f, 1.0f, 0.0f, 0.0f, 1.0f, -1.0f, -1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    static const float k709[3][2] = {{0.640f, 0.330f}, {0.300f, 0.600f}, {0.150f, 0.060f}};
    static const float k2020[3][2] = {{0.708f, 0.292f}, {0.170f, 0.797f}, {0.131f, 0.046f}};
    Mul4(kYcbcr2020ToRgb, kRgbToYcbcr709, fix2020);
    Mul4(kYcgcoToRgb, kRgbToYcbcr709, fixycgco);
    const M3 wide = RgbToXyz(k2020), inv = Inverse3(RgbToXyz(k709));
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            gamut[r * 3 + c] = inv.v[r][0] * wide.v[0][c] + inv.v[r][1] * wide.v[1][c] + inv.v[r][2] * wide.v[2][c];
}

}  // namespace mpcvr
