// vp_plan.h — host-side parameter maths of the shader video processor: format table, colourspace
// defaults, YUV->RGB matrix, gamut matrix, pass selection and resize tap tables.
// Mirrors (does not copy) Source/Helper.cpp, Source/csputils.cpp and the constant set-up in
// Source/DX11VideoProcessor.cpp; each function cites what it replaces.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "vp_params.h"
#include "../../include/mpcvr.h"

namespace mpcvr {

// ---- format table (Helper.cpp:295-359 s_FmtConvMapping) ----
struct FmtConvParams {
    int cformat;
    const char *str;
    int planes, bytes, div_w, div_h;
    int Packsize, PitchCoeff;
    int Subsampling, CDepth;
    int shift;      // 10-bit planar formats are <<6 by CopyPlane10to16 (Helper.cpp:386-391)
    int v_first;    // YV12/YV16/YV24
    int layout;     // SrcLayout
    int CSType;     // ColorSystem
    int ci[4];      // component order of the packed formats (see SrcFormat)
    int bits10;     // Y410, r210
    int repack;     // Repack
};
const FmtConvParams *GetFmtConvParams(int cformat);            // Helper.cpp:361-369
int DefaultPitch(const FmtConvParams &f, int width);           // DX11VideoProcessor.cpp:1789-1803
int SourceLines(const FmtConvParams &f, int height);           // m_srcLines
// pitch of the Y210 texture a v210 sample is unpacked into by CopyFrameV210 (Helper.cpp:709-748).  The reference uses
// the driver's mapped pitch (>= 4W, typically 256-aligned); 4W rounded up to whole 12-byte groups converts every pixel of
// the row exactly as any larger driver pitch would.
int V210TexPitch(int width);

// ---- DXVA2_ExtendedFormat (dxva2api.h layout) ----
struct ExtFmt {
    uint32_t value;
    unsigned SampleFormat() const { return value & 0xff; }
    unsigned VideoChromaSubsampling() const { return (value >> 8) & 0xf; }
    unsigned NominalRange() const { return (value >> 12) & 0x7; }
    unsigned VideoTransferMatrix() const { return (value >> 15) & 0x7; }
    unsigned VideoLighting() const { return (value >> 18) & 0xf; }
    unsigned VideoPrimaries() const { return (value >> 22) & 0x1f; }
    unsigned VideoTransferFunction() const { return (value >> 27) & 0x1f; }
    void set(int shift, unsigned mask, unsigned v) { value = (value & ~(mask << shift)) | ((v & mask) << shift); }
};
ExtFmt SpecifyExtendedFormat(ExtFmt ex, const FmtConvParams &f, int w, int h);   // Helper.cpp:1169-1211

// ---- colour maths ----
struct ProcAmp { float brightness = 0, contrast = 1, hue = 0, saturation = 1; };  // DXVA2 units
// SetShaderConvertColorParams (DX11VideoProcessor.cpp:813-887) -> 12 floats cm_r,cm_g,cm_b,cm_c
void ComputeColorMatrix(const ExtFmt &ex, const FmtConvParams &f, const ProcAmp &pa, float out12[12]);
// GetColorspaceGamutConversionMatrix(BT2020 -> BT709) (csputils.cpp:549-557)
void ComputeGamut2020to709(float out9[9]);
// the matrices ps_fix_bt2020 / ps_fix_ycgco / ps_(fix)convert_* fold at compile time (conv_matrix.hlsl,
// colorspace_gamut_conversion.hlsl), in fp32
void CorrectionMatrices(float fix2020[16], float fixycgco[16], float gamut[9]);

// ---- Dolby Vision (vp_dovi.cpp) ----
// the curve part of CheckDoviMetadata (VideoProcessor.cpp:283-292) with maxReshapeMethon = 1
bool CheckDoviCurves(const mpcvr_dovi_metadata &md);
// SetShaderDoviCurves / SetShaderDoviCurvesPoly (DX11VideoProcessor.cpp:990-1141) + has_mmr (:2305-2318)
void PackDoviCurves(const mpcvr_dovi_metadata &md, DoviParams *out);
// dovi_lms2rgb x rgb_to_lms_matrix (Shaders.cpp:826-842)
void DoviLmsMatrix(const mpcvr_dovi_metadata &md, float out9[9]);
// level-2 selection (:2383-2469) + SetDolbyVisionDynamicParams (:954-960); returns L2.present
bool DoviL2Constants(const mpcvr_dovi_metadata &md, int display_nits, float k[5]);
// level 1 (+3) in nits (:2347-2372); returns L1.present
bool DoviL1Nits(const mpcvr_dovi_metadata &md, uint32_t out[3]);
// the Dolby Vision branch of SetShaderConvertColorParams (:817-834) + its cbuffer fix-ups (:863-873)
void DoviColorMatrix(const mpcvr_dovi_metadata &md, const FmtConvParams &f, const ProcAmp &pa, float out12[12]);

// which HDR tail GetShaderConvertColor emits (Shaders.cpp:613-616, 861-923)
// hdr_output = m_bHdrPassthroughSupport && (m_bHdrPassthrough || m_bHdrLocalToneMapping): never TO_SDR, HLG -> PQ (:2948-2950)
// dovi = pDoviMetadata != nullptr: a PQ source whatever the transfer function says, bApplyHLG off (:614-615)
void SelectTail(const ExtFmt &ex, bool convert_to_sdr, int *tail, float *gamma, bool hdr_output = false, bool dovi = false);

// per-channel PQ->SDR chain saturate -> ST2084ToLinear*LuminanceScale -> Hable/hable(4.8) sampled at
// i/(kPqLutSize-1) (st2084.hlsl:9-16, hdr_tone_mapping.hlsl:1-13): the optional tone-map LUT of the fused path
void BuildPqSdrLut(float lum_scale, float out[kPqLutSize]);
void BuildHlgInverseLut(float out[kPqLutSize]);        // per-channel inverse_HLG for the fused kernels' HLG -> SDR tail
// log2 ST2084ToLinear((i / (kPqLutSize - 1))^2, 1) (st2084.hlsl:9-16): the PQ EOTF table of the Dolby Vision block convert, sampled uniformly in sqrt(x)
void BuildPqEotfLut(float out[kEotfLutSize + 1]);
void BuildPqEncodeLut(float out[kPqEncSize + 1]);      // vp_params.h kPqEncSize: the Dolby Vision level-2 variant's PQ encode over log2 x
// ps_final_pass.hlsl:29 in integers (fused kernel): floor(k*quant/maxv + j/1024) == (k*M + (j << 14)) >> 24 for all
// k in [0,maxv], j in [0,1023] with M = ceil(quant * 2^24 / maxv).  Returns M, or 0 when the 32-bit evaluation could
// overflow / M does not fit 24 bits (then the kernel keeps the float epilogue).
uint32_t FinalPassMultiplier(int quant, int maxv);
// SetHDR10ShaderParams (DX11VideoProcessor.cpp:911-916): the defaults and clamps in front of HDRParamsConstantBuffer_t
void SanitiseHdr10Params(HdrToneMapParams *k);

// ---- resize ----
enum ResizerKind { RS_NONE = 0, RS_UP = 1, RS_DOWN = 2 };
struct Resizer { int kind; int method; };

struct HostAxisTaps {
    int ntaps = 0;
    int normalise = 0;
    std::vector<int32_t> idx;
    std::vector<float> w;
    std::vector<float> wsum;
};
// weights for one fractional phase (ps_interpolation_*.hlsl); returns tap count (4/6) or 0
int UpscaleWeights(int iUpscaling, float t, float w[6]);
// convolution kernels (Shaders/resize/convolution_filters.hlsl); *support optional
float DownscaleFilter(int iDownscaling, float x, float *support);
// tap table for n_out outputs of one TextureResizeShader draw along one axis
// (DX11VideoProcessor.cpp:332-377 + the shader bodies).  Returns false if unsupported.
// reversed: the texture coordinate runs from the far edge of the source range backwards (rotation / flip,
// FillVertices :130-179); shader_scale: the constant scale[AXIS] ps_convolution sees (0 => src_len / n_out; a rotated
// draw is handed the ratio of the other screen dimension, :351-354).
bool BuildAxisTaps(Resizer rs, int src_l, int src_len, int n_out, int tex_len, uint32_t flags,
                   HostAxisTaps *out, bool reversed = false, float shader_scale = 0.0f);
// nearest / identity index map of the unfiltered axis of a draw
void BuildPointIndex(int src_l, int src_len, int n_out, int tex_len, std::vector<int32_t> *out, bool reversed = false);

// ---- arbitrary-ratio fused kernel (vp_fused_strip.hip): strip geometry from the two tap tables ----
struct StripPlan {
    int nt = 4;                  // taps per output the kernel runs on both axes (4, 6 or 8; shorter tables are zero-padded)
    int pxl = 1;                 // output pixels per lane
    int strip_w = 64;            // output columns per wavefront
    int ring = 8;                // rows of the vertical LDS window (8 or 16)
    int acols = 0;               // columns of a converted source row the widest strip needs (even)
    std::vector<int32_t> yrange; // [2 * n_out_y] {lo, hi} source row per output row
    std::vector<int32_t> xstrip; // [2 * n_strips] {lo, hi} source column per strip
    // the tap tables as the kernel reads them: nt taps per output (padding: weight 0 on the first tap's texel), weights
    // divided by ps_convolution's weight sum where the draw normalises
    std::vector<int32_t> xi_t, yi;   // [nt][n_out_x] tap-major, [n_out_y][nt] row-major
    std::vector<float> xw_t, yw;
};
// false: the tables do not fit the kernel (more than 16 taps, a vertical span above 31 rows, non-monotonic tables)
bool PlanFusedStrip(const HostAxisTaps &hx, const HostAxisTaps &hy, int n_out_x, int n_out_y, int src_w, int src_h, StripPlan *sp);

// ---- periodic-phase fused kernel (vp_fused_period.h): vertical ratio out : in = P : Q with compile-time tap rows ----
struct PeriodPlan {
    int P = 0, Q = 0;            // output rows : source rows (4:3, 3:2, 2:3, 1:2, 3:1); 0 = the tables do not fit the kernel
    int nt = 0;                  // taps per output on both axes as the kernel runs them: 4, 5 (Lanczos3's shared texel folded) or 6
    int acols = 0;               // columns of a converted source row the widest strip needs (even)
    int strip_w = 128;           // output columns per wavefront (even, <= 128): the width whose convert passes are best filled
    int own = 0;                 // which two output columns a lane owns (PeriodLaneColumn): chosen so that the X stage's ds_read_b64 meet no bank twice
    std::vector<int32_t> xstrip; // [2 * n_strips] {lo, hi} source column per strip
    std::vector<int32_t> xi_t;   // [nt][n_out_x] tap-major
    std::vector<float> xw_t;
    std::vector<float> yw;       // [n_out_y][8]: nt weights, zero padding; slot 7 of row PB * m: body m's centre-row bits (3:1, see PlanFusedPeriod)
};
// The kernel's tap ROWS are compile-time (base(r) + 6m, vp_fused_period.h): the plan succeeds only when hy's index table is exactly that
// pattern (clamped to the texture) for one of the supported ratios; hx may be any 4- or 6-tap table.  fold_q1: both tables are
// Direct3D 11 Lanczos3 tables whose taps 0 and 1 read the same texel (ps_interpolation_lanczos3.hlsl:33-34) — folded to 5 taps.
// heavy_convert: the convert stage carries a table tail (PQ / HLG -> SDR): its passes weigh ~4x what they do on SDR content
// Output column (relative to the strip) of lane `lane`'s pixel q under ownership `own`:
//   0: (2 lane, 2 lane + 1) — adjacent pair, one 8-byte store per row;  1: (lane, 64 + lane);  2: (2 (lane & 31) + (lane >> 5), 64 + the same)
// A ds_read_b64 serves lanes 0-31 and 32-63 as two groups over 64 four-byte banks (MI355X_MICROARCH.md, LDS): two bytes-apart columns of
// A's 24-byte column stride collide when they are 32 columns apart.  With `own` = 0 one group's 32 reads cover 64 output columns — 48 source
// columns at 4:3, every read a 2-way conflict (r03: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50); 1 makes them 32 consecutive output
// columns (<= 32 source columns when upscaling), 2 every other output column (source stride 3 at 2:3: a permutation of the banks).
inline int PeriodLaneColumn(int own, int lane, int q)
{
    return own == 0 ? 2 * lane + q : own == 1 ? lane + 64 * q : 2 * (lane & 31) + (lane >> 5) + 64 * q;
}
bool PlanFusedPeriod(const HostAxisTaps &hx, const HostAxisTaps &hy, int n_out_x, int n_out_y, int src_w, int src_h, bool fold_q1, PeriodPlan *pp, bool heavy_convert = true);

// ---- the pass plan of one Process() (DX11VideoProcessor.cpp:3285-3424, shader path) ----
struct PassPlan {
    int internal_fmt = SF_BGRA8;     // UpdateTexParams :1143-1155
    int swap_fmt = SF_BGRA8;         // render-target format
    bool final_pass = false;         // UpdatePostScaleTexures :2894-2912
    int quant = 255;
    Resizer rx{RS_NONE, 0}, ry{RS_NONE, 0};
    bool two_pass = false;           // X into fp16 m_TexResize, then Y
    bool one_pass = false;           // a single draw (one filtered axis, other point-sampled)
    int one_pass_axis = 0;
    bool copy_only = false;          // no size change: straight copy / final pass from the convert output
    bool fused_up2x = false;         // eligible for the fused 2x kernel
    bool fused_jinc = false;         // ... with the one-draw 2-D Jinc2m filter in the place of the two separable draws (vp_fused_jinc.hip)
    bool direct_convert = false;     // no resize draw and no tone-map step: the convert kernel rounds like m_TexConvertOutput
                                     // and runs the copy / final pass in its own epilogue (one kernel, no intermediate)
    // rotation-carrying (first) draw — FillVertices :130-179, ResizeShaderPass :3112-3137
    int rotation = 0;                // 0/90/180/270 clockwise
    bool flip = false;               // horizontal flip of the source
    int first_tex_axis = -1;         // texture axis the first draw filters: 0 = X shaders, 1 = Y shaders, -1 = ps_simple
    Resizer first_rs{RS_NONE, 0};
    int mid_h = 0;                   // height of m_TexResize in the two-pass case (srcRect extent along screen y)
    bool hdr_tonemap = false;        // ps_hdr10_tonemap step between the resize and the final pass / render target
    bool convert = true;             // ConvertColorPass runs; false: the source texture feeds the resize directly (:3321-3323)
    bool errdiff = false;            // EXTENSION (bUseDither = 2): the plan above is the 10-bit swap chain's (swap_fmt = SF_RGB10A2 into a window-sized
                                     // intermediate), and the error-diffusion pass (vp_errdiff.hip) takes it to the B8G8R8A8 render target
    std::string describe() const;
};

struct PlanGeometry { int w1, h1;            // source rect size (== convert output)
                      int vl, vt, vr, vb;    // video rect
                      int ww, wh;            // window size
                      int rotation = 0; int flip = 0;
                      int convert_enabled = 1;      // m_PSConvColorData.bEnable (:849-853)
                      int hdr_tonemap = 0;          // m_pPSHDR10ToneMapping exists: one more post-scale step (:785-787)
                      int dovi = 0; };              // m_Dovi.bValid: the convert shader reshapes (the fused 2x kernel has no such stage)
// Pure decision logic of UpdateTexParams / UpdatePostScaleTexures / ResizeShaderPass (no device work).
// cfg fields use the Settings_t names; returns false + *why when the combination is not implemented.
struct mpcvr_settings_fwd;
bool DecidePlan(int iTexFormat, int iChromaScaling, int iUpscaling, int iDownscaling, int bInterpolateAt50pct,
                int bUseDither, int output_format, uint32_t flags, const FmtConvParams &f,
                const PlanGeometry &g, PassPlan *plan, std::string *why);

}  // namespace mpcvr
