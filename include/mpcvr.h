/*
 * mpcvr.h — C-ABI of the MI355X-native shader video processor (libmpcvr.so).
 *
 * Drop-in boundary for ONE path of MPC Video Renderer: CDX11VideoProcessor::Process on the shader
 * video processor (convert -> resize -> final pass/dither).  The reference has no FFI; its narrowest
 * waist is the C++ virtual class CVideoProcessor (Source/VideoProcessor.h:40-296) as implemented by
 * CDX11VideoProcessor (Source/DX11VideoProcessor.h:256-384).  Each export below names the method it
 * replaces (file:line relative to the reference tree).  Plain pointers and sizes only.
 *
 * Conventions (mirroring the reference):
 *   - return type int32_t == HRESULT sign convention: 0 = S_OK, 1 = S_FALSE ("ok, nothing done"),
 *     negative = failure (MPCVR_E_*).
 *   - the processor owns all device resources; input sample memory is borrowed for the duration of
 *     mpcvr_copy_sample; output goes to caller-owned memory.
 *   - NOT re-entrant: one caller at a time per context (the reference serialises every entry under
 *     m_RendererLock, VideoRenderer.cpp:443-444,579-586).  Work is asynchronous on the context's HIP
 *     stream unless stated otherwise.
 */
#ifndef MPCVR_H
#define MPCVR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCVR_S_OK            0
#define MPCVR_S_FALSE         1
#define MPCVR_E_FAIL          ((int32_t)0x80004005)
#define MPCVR_E_POINTER       ((int32_t)0x80004003)
#define MPCVR_E_INVALIDARG    ((int32_t)0x80070057)
#define MPCVR_E_UNEXPECTED    ((int32_t)0x8000FFFF)
#define MPCVR_E_NOTIMPL       ((int32_t)0x80004001)
#define MPCVR_E_OUTOFMEMORY   ((int32_t)0x8007000E)
#define MPCVR_E_NOT_VALID_STATE ((int32_t)0x8007139F)

/* ColorFormat_t — Source/Helper.h:86-127 (same numeric values; the formats this build accepts). */
enum mpcvr_cformat {
    MPCVR_CF_NONE = 0,
    MPCVR_CF_NV12 = 1, MPCVR_CF_P010 = 2, MPCVR_CF_P016 = 3,
    MPCVR_CF_YUY2 = 4, MPCVR_CF_UYVY = 5,
    MPCVR_CF_P210 = 6, MPCVR_CF_P216 = 7,
    MPCVR_CF_Y210 = 8, MPCVR_CF_Y216 = 9, MPCVR_CF_V210 = 10,
    MPCVR_CF_AYUV = 11, MPCVR_CF_Y410 = 12, MPCVR_CF_Y416 = 13,
    MPCVR_CF_YV12 = 14, MPCVR_CF_YV16 = 15, MPCVR_CF_YV24 = 16,
    MPCVR_CF_YUV420P8 = 17, MPCVR_CF_YUV422P8 = 18, MPCVR_CF_YUV444P8 = 19,
    MPCVR_CF_YUV420P10 = 20, MPCVR_CF_YUV420P16 = 21,
    MPCVR_CF_YUV422P10 = 22, MPCVR_CF_YUV422P16 = 23,
    MPCVR_CF_YUV444P10 = 24, MPCVR_CF_YUV444P16 = 25,
    MPCVR_CF_GBRP8 = 26, MPCVR_CF_GBRP10 = 27, MPCVR_CF_GBRP16 = 28,
    MPCVR_CF_RGB24 = 29, MPCVR_CF_XRGB32 = 30, MPCVR_CF_ARGB32 = 31, MPCVR_CF_r210 = 32,
    MPCVR_CF_RGB48 = 33, MPCVR_CF_BGR48 = 34, MPCVR_CF_BGRA64 = 35, MPCVR_CF_B64A = 36,
    MPCVR_CF_Y8 = 37, MPCVR_CF_Y10 = 38, MPCVR_CF_Y16 = 39
};

/* Settings enums — Source/IVideoRenderer.h:25-72 (identical values). */
enum { MPCVR_TEXFMT_AUTOINT = 0, MPCVR_TEXFMT_8INT = 8, MPCVR_TEXFMT_10INT = 10, MPCVR_TEXFMT_16FLOAT = 16 };
enum { MPCVR_CHROMA_Nearest = 0, MPCVR_CHROMA_Bilinear = 1, MPCVR_CHROMA_CatmullRom = 2 };
enum { MPCVR_UPSCALE_Nearest = 0, MPCVR_UPSCALE_Mitchell = 1, MPCVR_UPSCALE_CatmullRom = 2,
       MPCVR_UPSCALE_Lanczos2 = 3, MPCVR_UPSCALE_Lanczos3 = 4, MPCVR_UPSCALE_Jinc2 = 5,
       /* EXTENSION — not a reference setting (IVideoRenderer.h:54-62 ends at Jinc2): the 6-tap Spline36 kernel, BASELINE.json
        * config 4's optional run.  Same draws, tap positions and intermediate formats as the other interpolation shaders. */
       MPCVR_UPSCALE_Spline36_EXT = 6 };
enum { MPCVR_DOWNSCALE_Box = 0, MPCVR_DOWNSCALE_Bilinear = 1, MPCVR_DOWNSCALE_Hamming = 2,
       MPCVR_DOWNSCALE_Bicubic = 3, MPCVR_DOWNSCALE_BicubicSharp = 4, MPCVR_DOWNSCALE_Lanczos = 5 };

/* mpcvr_settings.bUseDither: 0 / 1 as the reference's bool (Settings_t::bUseDither, IVideoRenderer.h:117: the ordered dither of
 * ps_final_pass.hlsl).  EXTENSION — 2 is not a reference setting (the reference has no error diffusion at all): BASELINE.json config 4's
 * "error-diffusion dither".  Where the reference would run its final pass into an 8-bit target (internal format above 8 bits), the
 * frame is rendered as for a 10-bit swap chain (R10G10B10A2: no final pass behind the 10-bit internal format; behind the fp16 internal
 * format that chain's own final pass — the ordered dither from fp16 to 10 bits, as the reference runs it there — stays in front of the
 * pass) and Floyd-Steinberg error diffusion in integers takes it to B8G8R8A8 inside video rect ∩ window (definition: csrc/vp_errdiff_core.h; the serial model in oracle/ is its only check).  On a
 * 10-bit target or with the 8-bit internal format it changes nothing, like bUseDither = 1 there.
 * The pass is a chain of bands that wait for each other through device memory; its forward progress does not depend on the order in which
 * the hardware starts workgroups (bands are taken by ticket).  Should a band ever give up waiting (a bounded number of polls), the frame is
 * wrong and the pass flags it: mpcvr_synchronize, mpcvr_get_current_image and the next error-diffusion pass answer MPCVR_E_FAIL — a caller
 * that synchronises its OWN stream instead must call mpcvr_synchronize before it trusts such a frame. */
enum { MPCVR_DITHER_None = 0, MPCVR_DITHER_Ordered = 1, MPCVR_DITHER_ErrorDiffusion_EXT = 2 };

/* Render-target format: stands in for the display-driven m_SwapChainFmt decision
 * (DX11VideoProcessor.cpp:1476-1478, Preferred10BitOutput DX11VideoProcessor.h:290-292). */
enum { MPCVR_OUT_BGRA8 = 0, MPCVR_OUT_RGB10A2 = 1 };

/* mpcvr_settings.flags */
#define MPCVR_FLAG_LANCZOS3_FIXED   0x1u  /* use the D3D9 twin's tap layout instead of the D3D11 shader's (quirk Q1) */
#define MPCVR_FLAG_NO_FUSED         0x2u  /* plain pass-per-kernel path only: no fused 2x kernel, no convert+final in one kernel,
                                             no compile-time-folded kernels, every tail literal (debug / A-B; bit-exact reference) */
#define MPCVR_FLAG_NO_LUT           0x4u  /* evaluate the PQ->SDR chain in ALU instead of the 4096-entry table (fused and folded kernels) */
#define MPCVR_FLAG_NO_FAST_CONVERT  0x8u  /* fused 2x kernel off when its vectorised convert would be needed: the folded pass-per-kernel
                                             path runs instead (debug / A-B) */

#define MPCVR_FLAG_FUSED_VALU       0x10u /* fused 2x kernel with its resize taps as packed-fp32 VALU chains (k_fused_up2x) */
#define MPCVR_FLAG_FUSED_MFMA       0x20u /* accepted and ignored since round 6 (it selected an experiment kernel with the resize taps on the matrix
                                             cores: parity-green, never faster; kept under profiles/r06/experiments/matrix_core_taps/) */
#define MPCVR_FLAG_NO_STRIP         0x40u /* arbitrary-ratio resizes of 4:2:0 sources stay on the block convert + tiled two-draw
                                             kernels instead of the one-kernel strip path (k_fused_strip; debug / A-B) */
#define MPCVR_FLAG_FORCE_PERIOD     0x100u /* take k_fused_period wherever it is built (debug / A-B).  Since round 6 that is where the planner takes it
                                              anyway: the one class it used to add — SDR content with a 4-tap filter, where k_fused_strip is as fast —
                                              is no longer built (35 instantiations, a tenth of the build) */
#define MPCVR_FLAG_NO_FRAME_LANES   0x200u /* mpcvr_process strictly one frame after the other.  Default: a context that owns its stream (no
                                              mpcvr_set_stream) deals consecutive single frames to four internal streams so that one frame's drain
                                              overlaps the next one's ramp-up — frames are independent, as the reference's draws into different
                                              render targets are for the D3D11 driver (Render -> Process, DX11VideoProcessor.cpp:2730).  Results are
                                              complete after mpcvr_synchronize (or any call that reads them back); frames into the SAME target
                                              stay in order.  (debug / A-B) */
#define MPCVR_FLAG_NO_PERIOD        0x80u /* rational vertical ratios (4:3, 3:2, 2:3, 1:2, 3:1) through k_fused_strip's run-time tap tables instead
                                             of the periodic-phase kernel with its register window (k_fused_period; debug / A-B) */

/* Subset of Settings_t (IVideoRenderer.h:104-135) that reaches the shader path; same field names. */
typedef struct mpcvr_settings {
    int32_t  iTexFormat;          /* MPCVR_TEXFMT_*            default AUTOINT   */
    int32_t  iChromaScaling;      /* MPCVR_CHROMA_*            default Bilinear  */
    int32_t  iUpscaling;          /* MPCVR_UPSCALE_*           default CatmullRom*/
    int32_t  iDownscaling;        /* MPCVR_DOWNSCALE_*         default Hamming   */
    int32_t  bInterpolateAt50pct; /*                           default 1         */
    int32_t  bUseDither;          /* MPCVR_DITHER_*            default 1 (Ordered) */
    int32_t  bDeintBlend;         /* blend-deinterlace 4:2:0 samples flagged interlaced (mpcvr_set_sample_format) */
    int32_t  bConvertToSdr;       /*                           default 1         */
    int32_t  iSDRDisplayNits;     /* 25..400                   default 125       */
    int32_t  output_format;       /* MPCVR_OUT_*               default BGRA8     */
    uint32_t flags;
} mpcvr_settings;

typedef struct mpcvr_rect { int32_t left, top, right, bottom; } mpcvr_rect;

/* mem_kind for mpcvr_copy_sample */
enum { MPCVR_MEM_HOST = 0, MPCVR_MEM_DEVICE = 1, MPCVR_MEM_HOST_PINNED = 2 };

/* ProcAmp flags — DXVA2_ProcAmp_* bit values */
#define MPCVR_PROCAMP_BRIGHTNESS 0x1
#define MPCVR_PROCAMP_CONTRAST   0x2
#define MPCVR_PROCAMP_HUE        0x4
#define MPCVR_PROCAMP_SATURATION 0x8

typedef struct mpcvr_ctx mpcvr_ctx;

/* Settings_t::SetDefault — IVideoRenderer.h:140-185 */
int32_t mpcvr_settings_default(mpcvr_settings *s);

/* ctor + Init — DX11VideoProcessor.cpp:381,547.  `device` = HIP device ordinal. */
int32_t mpcvr_create(const mpcvr_settings *settings, int32_t device, mpcvr_ctx **out);
int32_t mpcvr_destroy(mpcvr_ctx *ctx);

/* Use an externally owned hipStream_t (e.g. torch's current stream) for all work.  NULL = the context's own stream, which is a
 * BLOCKING stream (hipStreamDefault): it synchronises implicitly with the legacy null stream, so work a caller queued on stream 0
 * (NULL is also that stream's handle) stays ordered against mpcvr_copy_sample / mpcvr_process.  Other streams are the caller's to
 * order (events), as usual. */
int32_t mpcvr_set_stream(mpcvr_ctx *ctx, void *hip_stream);
int32_t mpcvr_synchronize(mpcvr_ctx *ctx);

/* VerifyMediaType + InitMediaType — DX11VideoProcessor.cpp:1569,1742.
 * cformat: ColorFormat_t value; width/height: biWidth/|biHeight|; pitch: bytes per luma row of the
 * samples that will be handed to mpcvr_copy_sample (0 => the reference's rule, :1789-1803; negative for an RGB format
 * = bottom-up DIB, the way m_srcPitch goes negative for BI_RGB with biHeight > 0, :1801-1803);
 * src_rect: rcSource (NULL or empty => whole frame, :1821-1823);
 * extfmt: DXVA2_ExtendedFormat.value from the media type (0 fields are defaulted per
 * SpecifyExtendedFormat, Helper.cpp:1169-1211).  Returns S_OK, or E_INVALIDARG/E_NOTIMPL. */
int32_t mpcvr_set_input(mpcvr_ctx *ctx, int32_t cformat, int32_t width, int32_t height, int32_t pitch,
                        const mpcvr_rect *src_rect, uint32_t extfmt);

/* SetVideoRect / SetWindowRect — DX11VideoProcessor.cpp:3426-3451.  Window rect = render-target size. */
int32_t mpcvr_set_video_rect(mpcvr_ctx *ctx, const mpcvr_rect *video_rect);
int32_t mpcvr_set_window_rect(mpcvr_ctx *ctx, const mpcvr_rect *window_rect);
/* SetRotation (DX11VideoProcessor.cpp:4052) / SetFlip (VideoProcessor.h:210): 0/90/180/270 degrees clockwise and a
 * horizontal flip of the source, applied in the first resize draw the way FillVertices (:130-179) and ResizeShaderPass
 * (:3112-3137) set it up; the caller sizes the video rect for the rotated picture.  E_INVALIDARG for other angles. */
int32_t mpcvr_set_rotation(mpcvr_ctx *ctx, int32_t degrees);
/* (extension, error-diffusion final pass only) how many polls — about a microsecond each — a band grants the band above before the pass is
 * flagged failed (the next mpcvr_process / mpcvr_synchronize answers MPCVR_E_FAIL instead of hanging); polls <= 0: the default, 2^21.  A host
 * that shares the GPU or runs under a debugger raises it; the environment variable MPCVR_ERRDIFF_SPIN (tests) overrides it per call. */
int32_t mpcvr_set_error_diffusion_patience(mpcvr_ctx *ctx, int32_t polls);
int32_t mpcvr_set_flip(mpcvr_ctx *ctx, int32_t flip);
/* m_SampleFormat as CopySample derives it from AM_SAMPLE2_PROPERTIES::dwTypeSpecificFlags (DX11VideoProcessor.cpp:2209-2219):
 * 0 progressive, 1 interlaced top field first, 2 interlaced bottom field first.  With bDeintBlend an interlaced 4:2:0
 * sample goes through the blend variant of the convert shader (:3075, Shaders.cpp:232-237). */
int32_t mpcvr_set_sample_format(mpcvr_ctx *ctx, int32_t frame_format);
/* HDR output: `enable` stands in for m_bHdrPassthroughSupport && (m_bHdrPassthrough || m_bHdrLocalToneMapping) — the
 * display is in HDR10 mode — so PQ sources pass through unconverted and HLG is converted to PQ (convertType,
 * DX11VideoProcessor.cpp:2948-2950); pick output_format RGB10A2 as the reference's swap chain does.  tone_map_type =
 * m_iHdrLocalToneMappingType (0 off, 1 ACES, 2 Reinhard, 3 Habel, 4 Moebius, 5 BT.2390, 6 ST 2094-10), display_max_nits =
 * m_iHdrDisplayMaxNits: ps_hdr10_tonemap.hlsl then runs as a post-scale step once mpcvr_set_hdr_metadata was called. */
int32_t mpcvr_set_hdr_output(mpcvr_ctx *ctx, int32_t enable, int32_t tone_map_type, float display_max_nits);
/* the values Render() passes to SetHDR10ShaderParams (:907-917, :2716-2727), sanitised the same way */
int32_t mpcvr_set_hdr_metadata(mpcvr_ctx *ctx, float min_mastering_nits, float max_mastering_nits, float max_cll, float max_fall);

/* Dolby Vision RPU of the next sample(s): the fields of MediaSideDataDOVIMetadata (Include/IMediaSideData.h:154-330) the
 * frame path reads when CopySample finds IID_MediaSideDataDOVIMetadataV2 on the sample (DX11VideoProcessor.cpp:2270-2520).
 * The reference struct is #pragma pack(1); this one has natural C alignment, the adapter copies field by field.
 * Extension blocks: the first level-1 block (+ the first level-3 block, if any) and every level-2 block, in
 * Extensions[] order. */
typedef struct mpcvr_dovi_curve {
    uint8_t  num_pivots;            /* [2, 9] */
    uint8_t  mapping_idc[8];        /* 0 polynomial, 1 mmr */
    uint8_t  poly_order[8];         /* [1, 2] */
    uint8_t  mmr_order[8];          /* [1, 3] */
    uint16_t pivots[9];             /* sorted ascending, bl_bit_depth codes */
    int64_t  poly_coef[8][3];       /* x^0, x^1, x^2 ; fixed point, coef_log2_denom fractional bits */
    int64_t  mmr_constant[8];
    int64_t  mmr_coef[8][3][7];
} mpcvr_dovi_curve;
typedef struct mpcvr_dovi_l2 {
    uint16_t target_max_pq, trim_slope, trim_offset, trim_power, trim_chroma_weight, trim_saturation_gain;
} mpcvr_dovi_l2;
typedef struct mpcvr_dovi_metadata {
    uint8_t  bl_bit_depth, coef_log2_denom;                 /* Header */
    uint16_t source_max_pq;                                 /* ColorMetadata */
    uint8_t  l1_present, l3_present;
    uint16_t l1_min_pq, l1_max_pq, l1_avg_pq, l3_min_pq_offset, l3_max_pq_offset, l3_avg_pq_offset;
    uint32_t n_l2;                                          /* <= 32 (LAV_DOVI_MAX_EXTENSIONS) */
    mpcvr_dovi_l2 l2[32];
    double   ycc_to_rgb_matrix[9], ycc_to_rgb_offset[3], rgb_to_lms_matrix[9];   /* ColorMetadata */
    mpcvr_dovi_curve curves[3];                             /* Mapping.curves, per component */
} mpcvr_dovi_metadata;
/* m_Dovi.msd = *md, m_Dovi.bValid = true: the convert pass reshapes (Y,U,V) through the curves (ShaderDoviReshape[Poly],
 * Shaders.cpp:531-589), uses ycc_to_rgb_matrix/offset as the colour matrix (SetShaderConvertColorParams :817-834), goes
 * PQ -> linear -> dovi_lms2rgb x rgb_to_lms_matrix -> PQ (Shaders.cpp:826-859) and continues as a PQ source; level-2 trims
 * are selected for the display peak of mpcvr_set_hdr_output (:2383-2469) and applied in the convert tail (SDR output) and
 * the tone-mapping step (HDR output); level 1 (+3) replaces the HDR10 metadata of that step (:2716-2720).
 * md == NULL ends Dolby Vision mode (m_Dovi = {}).  E_INVALIDARG for what CheckDoviMetadata rejects in the curves
 * (VideoProcessor.cpp:283-292); the profile checks on the RPU header (:275-281) stay with the adapter.
 * Kernels: the reshaping, the LMS step and the tail run in the 2x2-block convert (k_convert_blocks<..., DV_*>: same-size frames in
 * one kernel, straight into the render target; in front of k_fused_strip:surface / k_fused_period:surface when the frame is
 * resized); the exact-2x and the raw-sample strip kernels have no reshaping stage.  The metadata is per context: every frame of an
 * mpcvr_process_batch call runs on the RPU last set (mpcvr_process_batch_dovi takes one RPU per frame). */
int32_t mpcvr_set_dovi_metadata(mpcvr_ctx *ctx, const mpcvr_dovi_metadata *md);

/* The correction shaders (m_pPSCorrection, DX11VideoProcessor.cpp:1893-1930): same-size RGB -> RGB passes the reference runs
 * over the OUTPUT of the fixed-function D3D11 video processor (Process :3354-3357) — BT.2020 / YCgCo matrix fix-ups and the
 * PQ / HLG tone mapping that path cannot do itself.  The shader video processor of this library never needs them (its
 * convert pass does all of it before the resize); they are offered as standalone passes over one surface for a host that
 * decodes / scales elsewhere.  kind = MPCVR_CORR_*; src / dst: DEVICE surfaces of 32-bit texels, fmt = MPCVR_OUT_BGRA8 or
 * MPCVR_OUT_RGB10A2, w x h texels; sdr_nits = iSDRDisplayNits (LuminanceScale = 10000 / nits, :889-905); stream = a
 * hipStream_t or NULL.  In place (src == dst) is allowed. */
#define MPCVR_CORR_FIX_BT2020            1   /* Shaders/d3d11/ps_fix_bt2020.hlsl */
#define MPCVR_CORR_FIX_YCGCO             2   /* ps_fix_ycgco.hlsl */
#define MPCVR_CORR_FIXCONVERT_PQ_TO_SDR  3   /* ps_fixconvert_pq_to_sdr.hlsl */
#define MPCVR_CORR_FIXCONVERT_HLG_TO_SDR 4   /* ps_fixconvert_hlg_to_sdr.hlsl */
#define MPCVR_CORR_CONVERT_PQ_TO_SDR     5   /* ps_convert_pq_to_sdr.hlsl */
#define MPCVR_CORR_CONVERT_HLG_TO_PQ     6   /* ps_convert_hlg_to_pq.hlsl */
int32_t mpcvr_correction_pass(int32_t kind, const void *src, int32_t src_pitch, int32_t src_fmt,
                              void *dst, int32_t dst_pitch, int32_t dst_fmt, int32_t w, int32_t h, int32_t sdr_nits, void *stream);
/* the three matrices those shaders fold at compile time, evaluated in fp32: fix_bt2020_matrix and fix_ycgco_matrix (4x4,
 * row-major) and convert_matrix_2020_to_709 of convert/colorspace_gamut_conversion.hlsl (3x3) */
int32_t mpcvr_plan_correction_matrices(float fix_bt2020_16[16], float fix_ycgco_16[16], float gamut9[9]);

/* Configure — DX11VideoProcessor.cpp:3800-4050: diff each field, rebuild only what changed. */
int32_t mpcvr_configure(mpcvr_ctx *ctx, const mpcvr_settings *settings);

/* SetProcAmpValues — DX11VideoProcessor.cpp:4506-4537; ranges Helper.cpp:182-187
 * (brightness -100..100, contrast 0..2, hue -180..180, saturation 0..2). */
int32_t mpcvr_set_procamp(mpcvr_ctx *ctx, uint32_t flags, float brightness, float contrast,
                          float hue, float saturation);

/* CopySample / MemCopyToTexSrcVideo — DX11VideoProcessor.cpp:2202,1213-1252.
 * data: one media sample (planes back to back); pitch: the row pitch given to mpcvr_set_input (negative for a bottom-up
 * RGB DIB, whose `data` still points at the lowest address, DX11VideoProcessor.cpp:1243-1248).
 * MPCVR_MEM_HOST: copied into one of three pinned staging buffers and uploaded on a copy stream, so the upload of
 * sample n+1 overlaps the processing of sample n (CopyPlane10to16's <<6 / CopyFrameV210 / CopyFrameRGB* run on the
 * device); the caller's buffer is free again when the call returns.  MPCVR_MEM_HOST_PINNED: the buffer is page-locked
 * (hipHostMalloc / hipHostRegister) and is DMA'd from directly; it must stay untouched until mpcvr_synchronize or the
 * third following copy_sample.  MPCVR_MEM_DEVICE: zero-copy — the pointer is used in place and must stay valid until
 * the following process/render call has completed (mirrors the IMediaSampleD3D11 branch :2528-2569). */
int32_t mpcvr_copy_sample(mpcvr_ctx *ctx, const void *data, int32_t pitch, int32_t mem_kind);

/* Process — DX11VideoProcessor.cpp:3285-3424.  dst: DEVICE pointer to a window_w x window_h render
 * target, 4 bytes per pixel (B8G8R8A8 or R10G10B10A2), dst_pitch bytes per row.  src_rect must be
 * NULL or the input's source rect (the reference overrides it with the convert texture, :3316-3319);
 * dst_rect NULL => the context's video rect.  Pixels outside dst_rect are not written. */
int32_t mpcvr_process(mpcvr_ctx *ctx, void *dst_dev, int32_t dst_pitch, const mpcvr_rect *src_rect,
                      const mpcvr_rect *dst_rect, int32_t second_field);

/* n frames the reference's way — mpcvr_copy_sample(samples[i], pitch, mem_kind) then mpcvr_process(dsts_dev[i], dst_pitch, NULL, NULL, 0), frame
 * after frame (ProcessSample -> CopySample -> Render -> Process, DX11VideoProcessor.cpp:2143-2200, :2730) — behind one call, so that a caller
 * in a scripting language measures the path and not its own foreign-function calls.  Not a batch: every frame is its own launch (or lands
 * on the context's frame lanes); stops at the first failure and returns it. */
int32_t mpcvr_process_frames(mpcvr_ctx *ctx, int32_t n, const void *const *samples, int32_t pitch, int32_t mem_kind, void *const *dsts_dev, int32_t dst_pitch);

/* Render minus Present — DX11VideoProcessor.cpp:2599-2813: Process into the context-owned back buffer. */
int32_t mpcvr_render(mpcvr_ctx *ctx, int32_t field);
/* Device pointer / pitch of the context-owned back buffer written by mpcvr_render. */
int32_t mpcvr_get_backbuffer(mpcvr_ctx *ctx, void **dev_ptr, int32_t *pitch, int32_t *width, int32_t *height);

/* GetCurentImage — DX11VideoProcessor.cpp:3493-3608: source-rect-sized BGRX snapshot into host memory.
 * Two-call size protocol (VideoRenderer.cpp:979-988): host_bgra == NULL => *size receives the bytes
 * needed (rect_w*rect_h*4); otherwise *size must be >= that.  Synchronous. */
int32_t mpcvr_get_current_image(mpcvr_ctx *ctx, void *host_bgra, size_t *size);
/* GetDisplayedImage — DX11VideoProcessor.cpp:3610-3683: the back buffer of the last mpcvr_render as it is (nothing is drawn again), as
 * the pixels of a top-down DIB in host memory: a B8G8R8A8 buffer as it is; an R10G10B10A2 one as BGR32 (the top eight bits of each channel,
 * ConvertR10G10B10A2toBGR32, Helper.cpp:805) or, with deep_color != 0 (m_bAllowDeepColorBitmaps), as BGR48 (ConvertR10G10B10A2toBGR48, :836).
 * Rows are ((width * bits + 31) & ~31) / 8 bytes apart (CalcDibRowPitch).  The BITMAPINFOHEADER and the LocalAlloc block around the pixels
 * are the caller's: host_pixels == NULL => *size, *width, *height and *bits_per_pixel are reported (any of the last three may be NULL);
 * otherwise *size must be >= the size reported.  MPCVR_E_NOT_VALID_STATE before the first mpcvr_render.  Synchronous. */
int32_t mpcvr_get_displayed_image(mpcvr_ctx *ctx, void *host_pixels, size_t *size, int32_t deep_color, int32_t *width, int32_t *height, int32_t *bits_per_pixel);

/* Flush (DX11VideoProcessor.cpp:4074) / Reset (:3453). */
int32_t mpcvr_flush(mpcvr_ctx *ctx);
int32_t mpcvr_reset(mpcvr_ctx *ctx);

/* Extension (not in the reference): n frames in one launch sequence to escape the launch-bound
 * regime.  srcs[i]: DEVICE sample pointers (layout/pitch as declared by mpcvr_set_input);
 * dsts[i]: DEVICE render targets (dst_pitch each).
 * The frames of a batch are independent of each other: on every path that can, the whole batch runs as one launch per
 * draw (fused 2x / strip / periodic kernel: one launch; same-size frames: one k_convert_stream launch; pass-per-kernel path:
 * block convert / X draw / Y draw with a frame dimension and batched intermediates, <= 4 GiB; Dolby Vision: the block convert's
 * frame dimension, one RPU per call here, one per frame in mpcvr_process_batch_dovi; the HDR10 tone-mapping step: one launch behind
 * batched post-scale textures; quarter turns, flips outside the strip kernels' reach and Jinc2m in its one- and two-draw forms: every
 * draw kernel has a frame dimension; bUseDither = 2: ONE error-diffusion launch behind the batch's 10-bit frames), otherwise frame by
 * frame (samples that do not start on a dword or need a repack of their own outside the v210 / interleaved-RGB batch textures).
 * mpcvr_get_last_batch_info reports the kernel launches a batch took.  The targets must therefore be distinct buffers; completion is
 * in stream order for the batch as a whole.
 * Round 6: on a context that OWNS its stream (no mpcvr_set_stream) consecutive batches whose plan is one launch with no intermediate
 * surface (exact 2x, the strip / periodic kernel reading the samples, the fused Jinc2m kernel, the same-size block convert; no Dolby Vision,
 * no repack) take turns on two internal lanes, so that two launches are in flight and fill each other's ramp-up and tail (4K -> 8K: +4 %,
 * 1080p -> 1440p: +16 %; MPCVR_NO_BATCH_LANES=1 in the environment turns it off).
 * Batches and single frames that write the same render target (same pointer) stay in the order they were queued; everything that can observe
 * a result (mpcvr_synchronize, the snapshot, a plan change, mpcvr_set_stream) waits for the lanes.  MPCVR_FLAG_NO_FRAME_LANES (or a caller's
 * stream) keeps every batch in stream order; mpcvr_get_last_batch_info names the lane ("lane=0|1", -1 = the context stream). */
int32_t mpcvr_process_batch(mpcvr_ctx *ctx, int32_t n, const void *const *srcs, void *const *dsts,
                            int32_t dst_pitch);
/* mpcvr_process_batch for a Dolby Vision stream: rpus[i] is the RPU of frame i — the reference reads it from every sample in
 * CopySample (IID_MediaSideDataDOVIMetadataV2, DX11VideoProcessor.cpp:2270-2520).  The result is what n rounds of
 * mpcvr_set_dovi_metadata(&rpus[i]) + mpcvr_process(srcs[i] -> dsts[i]) produce, level-1 / level-2 blocks staying as last seen
 * like there, and the context ends up holding the last frame's RPU.  Frames are cut into runs that share a plan and a kernel variant
 * (level-2 trims for this display present or not); a run is ONE launch per stage where the block convert's Dolby Vision variants
 * run (same-size frames; convert + resize kernels): they read frame z's curves, LMS matrix, trims and ycc_to_rgb matrix from
 * device tables indexed by the frame.  A run behind the HDR10 tone-mapping step (its level-1 constants travel by value), and
 * whatever the per-pixel convert serves, goes frame by frame with each RPU's constants uploaded in stream order.
 * E_INVALIDARG, and nothing drawn, when any RPU of the batch fails the checks of mpcvr_set_dovi_metadata. */
int32_t mpcvr_process_batch_dovi(mpcvr_ctx *ctx, int32_t n, const void *const *srcs, void *const *dsts,
                                 int32_t dst_pitch, const mpcvr_dovi_metadata *rpus);

/* Multi-GPU: the parameter blob (colour matrix, luminance scale, gamut matrix, resize phase weights,
 * dither table) a rank-0 context computes and every other rank adopts after an RCCL broadcast.
 * Two-call size protocol. */
int32_t mpcvr_get_param_blob(mpcvr_ctx *ctx, void *buf, size_t *size);
int32_t mpcvr_set_param_blob(mpcvr_ctx *ctx, const void *buf, size_t size);
/* The same exchange done by the library over RCCL ("one ncclBroadcast (RCCL, root 0) at context creation") for hosts
 * without Python: `nccl_comm` is an ncclComm_t the host created (ncclCommInitRank in a process-per-GPU host, ncclCommInitAll when
 * one process drives several devices), `rank` this context's rank in it.  The broadcast runs on the context's stream; the
 * library resolves librccl at the first call (it is not a link-time dependency).
 *   one process per GPU:       mpcvr_broadcast_param_blob(ctx, comm, 0, rank);
 *   one process, N devices:    ncclGroupStart(); for d: mpcvr_broadcast_param_blob_begin(ctx[d], comm[d], 0, d); ncclGroupEnd();
 *                              for d: mpcvr_broadcast_param_blob_end(ctx[d]);            (examples/c_multi_gpu_rccl.c)
 * There is no reference line to cite: the reference is single-GPU (Source/DX11Helper.cpp:81-112). */
int32_t mpcvr_broadcast_param_blob_begin(mpcvr_ctx *ctx, void *nccl_comm, int32_t root, int32_t rank);
int32_t mpcvr_broadcast_param_blob_end(mpcvr_ctx *ctx);
int32_t mpcvr_broadcast_param_blob(mpcvr_ctx *ctx, void *nccl_comm, int32_t root, int32_t rank);

/* Introspection used by tests / stats (GetVPInfo analogue, DX11VideoProcessor.cpp:4100+). */
int32_t mpcvr_get_color_matrix(mpcvr_ctx *ctx, float out12[12]);     /* cm_r, cm_g, cm_b, cm_c */
int32_t mpcvr_get_extfmt(mpcvr_ctx *ctx, uint32_t *extfmt);           /* after SpecifyExtendedFormat */
int32_t mpcvr_get_frame_bytes(mpcvr_ctx *ctx, size_t *bytes, int32_t *pitch);
int32_t mpcvr_get_path_info(mpcvr_ctx *ctx, char *buf, size_t buf_size); /* e.g. "fused_up2x" / "fused_jinc2x" / "passes:convert,resizeX,resizeY+final" */
/* How the last mpcvr_process_batch / mpcvr_process_batch_dovi call ran: "frames=<n>;launches=<kernel launches>[;dovi_runs=<frames>:<tables|frames>,...]".
 * A batch on a whole-batch route launches a handful of kernels whatever n is (one per stage and <= 4 GiB chunk of intermediates); a
 * frame-by-frame one at least n.  dovi_runs: the runs mpcvr_process_batch_dovi cut the frames into and whether a run read its RPUs from the
 * per-frame tables or went frame by frame. */
int32_t mpcvr_get_last_batch_info(mpcvr_ctx *ctx, char *buf, size_t buf_size);
const char *mpcvr_last_error(mpcvr_ctx *ctx);
const char *mpcvr_version(void);

/* Timing of the last process/render on the context stream (hipEvent pair), milliseconds.
 * Mirrors m_RenderStats.paintticks (DX11VideoProcessor.cpp:2790).  Synchronises the stream.  Single frames that overlap on the
 * context's frame lanes are timed one in eight (the first always): two timestamped events around every 45 us kernel cost the per-frame
 * path 1.5 % at 4K -> 8K and 10-40 % on 1080p same-size frames; the figure is the most recent TIMED frame's (MPCVR_LANE_TIMING_EVERY=1
 * times them all; batches and frames off the lanes always are). */
int32_t mpcvr_get_last_process_ms(mpcvr_ctx *ctx, float *ms);
/* The renderer's other per-frame timers (FrameStats.h:145-173): copy_host_ms = wall time the last mpcvr_copy_sample spent on the host
 * (copyticks, DX11VideoProcessor.cpp:2594), upload_ms = its host-to-device transfer on the copy stream (hipEvent pair; absent for
 * device samples), process_ms = mpcvr_get_last_process_ms, readback_ms = the device-to-host copy of the last mpcvr_get_current_image.
 * -1 where nothing of the kind has been timed yet; any pointer may be NULL.  Synchronises the events it reads. */
int32_t mpcvr_get_last_timings(mpcvr_ctx *ctx, float *copy_host_ms, float *upload_ms, float *process_ms, float *readback_ms);

/* ---- host-side parameter maths, usable without a context or a GPU --------------------------------
 * The CPU work the reference does before touching the device.  Each mirrors a reference host function. */
/* pitch / byte size of one media sample — DX11VideoProcessor.cpp:1789-1803 (m_srcPitch, m_srcLines) */
int32_t mpcvr_plan_frame_layout(int32_t cformat, int32_t width, int32_t height, int32_t *pitch, size_t *bytes);
/* SpecifyExtendedFormat (Helper.cpp:1169-1211) + SetShaderConvertColorParams (DX11VideoProcessor.cpp:813-887)
 * -> mp_get_csp_matrix (csputils.cpp:392-509).  ProcAmp in DXVA2 units. */
int32_t mpcvr_plan_color_matrix(int32_t cformat, int32_t rect_w, int32_t rect_h, uint32_t extfmt,
                                float brightness, float contrast, float hue, float saturation,
                                float out12[12], uint32_t *extfmt_out);
/* GetColorspaceGamutConversionMatrix(BT.2020 -> BT.709) — csputils.cpp:549-557 */
int32_t mpcvr_plan_gamut_2020_to_709(float out9[9]);
/* the fused path's tone-map LUT: i/4095 -> Hable(ST2084ToLinear(x, lum_scale)) / hable(4.8) */
int32_t mpcvr_plan_pq_lut(float lum_scale, float out4096[4096]);
/* the fused path's integer form of ps_final_pass.hlsl:29: floor(k*quant/maxv + j/1024) == (k*M + (j << 14)) >> 24;
 * writes M (0 = not representable, the float epilogue is used) */
int32_t mpcvr_plan_final_pass_multiplier(int32_t quant, int32_t maxv, uint32_t *multiplier);
/* Dolby Vision host maths.  cb: the PS_DOVI_CURVE cbuffer of SetShaderDoviCurves (:1055-1141) as 3 x 235 floats — per
 * component pivots[7], coeffs[8][4], mmr[48][4], then {methods, mmr_single, min_order, max_order} as float values;
 * lms9: dovi_lms2rgb x rgb_to_lms_matrix (Shaders.cpp:826-842); l2k: {ChromaWeight, SaturationGain, TrimSlope, TrimOffset,
 * TrimPower} of SetDolbyVisionDynamicParams (:954-960) for a display of display_nits, *l2_enabled = L2Enabled;
 * l1_nits: {min, max, avg} as CopySample stores them (:2347-2372), *l1_present.  Any out pointer may be NULL. */
int32_t mpcvr_plan_dovi(const mpcvr_dovi_metadata *md, int32_t display_nits, float *cb705, int32_t *has_mmr,
                        float lms9[9], float l2k[5], int32_t *l2_enabled, uint32_t l1_nits[3], int32_t *l1_present);
/* ps_interpolation_{spline4,lanczos2,lanczos3}.hlsl weights for phase t; returns the tap count (4/6) or 0 */
int32_t mpcvr_plan_upscale_weights(int32_t iUpscaling, float t, float w6[6]);
/* tap table of one TextureResizeShader draw (DX11VideoProcessor.cpp:332-377): kind 0 = point sample,
 * 1 = upscale shader `method` (MPCVR_UPSCALE_*), 2 = ps_convolution with MPCVR_DOWNSCALE_* `method`.
 * idx/w: [n_out*cap_taps >= n_out*ntaps], dense with stride *ntaps; S_FALSE when cap_taps is too small. */
int32_t mpcvr_plan_axis_taps(int32_t kind, int32_t method, int32_t src_l, int32_t src_len, int32_t n_out,
                             int32_t tex_len, uint32_t flags, int32_t cap_taps, int32_t *idx, float *w,
                             float *wsum, int32_t *ntaps, int32_t *normalise);
/* Geometry of the arbitrary-ratio fused kernel (k_fused_strip) for an unrotated two-pass resize src_w x src_h -> out_w x out_h
 * with the given per-axis draws (kind / method as in mpcvr_plan_axis_taps): out8 = {taps per output the kernel runs (4/6/8),
 * pixels per lane, output columns per strip, rows of the LDS window, columns of a converted source row, strips, LDS bytes per
 * wavefront, 0}.  yrange: [out_h][2] {smallest, largest} source row per output row; xstrip: [strips][2] likewise per strip;
 * xi_t / xw_t: [taps][out_w] and yi / yw: [out_h][taps], the zero-padded, pre-normalised tables the kernel reads.  Any table
 * pointer may be NULL.  MPCVR_E_NOTIMPL: the tables do not fit the kernel (more than 16 taps, window above 31 rows). */
int32_t mpcvr_plan_strip(int32_t kind_x, int32_t method_x, int32_t kind_y, int32_t method_y, int32_t src_w, int32_t src_h,
                         int32_t out_w, int32_t out_h, uint32_t flags, int32_t out8[8], int32_t *yrange, int32_t *xstrip,
                         int32_t *xi_t, float *xw_t, int32_t *yi, float *yw);
/* Geometry of the periodic-phase fused kernel (k_fused_period) for an unrotated two-pass UPSCALE-shader resize (`method` =
 * MPCVR_UPSCALE_*; also what a downscale of at most 2x takes with bInterpolateAt50pct, DX11VideoProcessor.cpp:3108):
 * out6 = {P, Q (output : source rows), taps per output as the kernel runs them (4 / 5 = Lanczos3 with its shared texel folded /
 * 6), strips of *strip_w output columns (the width that fills the convert passes best, <= 128), columns of a converted source row, output rows per body of six source rows}.
 * xi_t / xw_t: [taps][out_w]; yw: [out_h][8]; xstrip: [strips][2]; any may be NULL.  MPCVR_E_NOTIMPL: the vertical ratio is not
 * 4:3 / 3:2 / 2:3 / 1:2 / 3:1 or the table's tap rows are not the periodic pattern the kernel hard-codes. */
int32_t mpcvr_plan_period(int32_t method, int32_t src_w, int32_t src_h, int32_t out_w, int32_t out_h, uint32_t flags,
                          int32_t out6[6], int32_t *xi_t, float *xw_t, float *yw, int32_t *xstrip, int32_t *strip_w);
/* HDRParamsConstantBuffer_t as SetHDR10ShaderParams fills it (DX11VideoProcessor.cpp:907-923: defaults and clamps of the HDR10
 * metadata, the display's peak and the tone-mapping operator): five floats + the selection as words */
int32_t mpcvr_plan_hdr10_params(float min_mastering, float max_mastering, float max_cll, float max_fall, float display_max,
                                int32_t selection, uint32_t out6[6]);
/* log2 of ST2084ToLinear(x, 1) at x = (i/N)^2, i = 0 .. N: the PQ EOTF table of the Dolby Vision block convert (N = 8192 in this
 * build).  Two calls: out = NULL stores the number of floats (N + 1) in *count; then `capacity` floats of room — MPCVR_E_INVALIDARG when
 * that is fewer than the table has, nothing is written.  (Replaces mpcvr_plan_pq_eotf_lut(float[4096]) of round 3, whose table grew in
 * place in round 4: a caller that sizes its buffer from an old header can no longer be overrun.) */
int32_t mpcvr_plan_pq_eotf_table(float *out, int32_t capacity, int32_t *count);
/* DEPRECATED, one more release: the round-3 entry point with its fixed float[4096] — the same function on that 4096-point grid
 * (x = (i/4095)^2), so that a caller built against the old header links and is neither overrun nor handed a table of another size.
 * New code: mpcvr_plan_pq_eotf_table. */
int32_t mpcvr_plan_pq_eotf_lut(float out[4096]);
/* which draws Process() would issue (UpdateTexParams :1143, UpdatePostScaleTexures :2894, ResizeShaderPass :3103) */
int32_t mpcvr_plan_describe(const mpcvr_settings *s, int32_t cformat, int32_t rect_w, int32_t rect_h,
                            const mpcvr_rect *video_rect, int32_t window_w, int32_t window_h,
                            char *buf, size_t buf_size);

/* A measurement aid (bench.py: roofline.empirical_shape_peak_GBps), not part of the video path: one launch that reads src_bytes (a multiple of
 * 16, 16-byte aligned DEVICE buffers) once and writes fan * src_bytes — the fused kernels' traffic shape (a 4K P010 sample in, an 8K
 * B8G8R8A8 target out is 1 : 5.33) with no arithmetic.  dst holds fan * src_bytes bytes; stream = a hipStream_t or NULL. */
int32_t mpcvr_bandwidth_probe(const void *src_dev, void *dst_dev, size_t src_bytes, int32_t fan, void *stream);
/* ... and the exact-2x kernel's own shape over a whole batch in one launch (round 6): n (<= 64) bi-planar 16-bit 4:2:0 samples of src_w x src_h
 * (P010: src_w * src_h * 3 bytes each) and n targets of 2 src_w x 2 src_h x 4 bytes; a wavefront owns a strip of 120 source columns x
 * seg_rows source rows, reads two luma rows and a chroma row per step (240 contiguous bytes each) and writes four target rows (960 contiguous
 * bytes each, one 16-byte piece per lane), like csrc/vp_fused_up2x.h.  mode 0: read + write, 1: write only, 2: read only.  strip_cols (even, <= 128):
 * source columns per wavefront — 120 is the kernel's, 128 the 1 KiB-aligned variant of the same pattern.  The pointer arrays are
 * HOST arrays of DEVICE pointers (16-byte aligned). */
int32_t mpcvr_bandwidth_probe_up2x(int32_t mode, int32_t n, const void *const *srcs_dev, void *const *dsts_dev, int32_t src_w, int32_t src_h,
                                   int32_t seg_rows, int32_t strip_cols, void *stream);

/* A verification aid, not part of the video path: the transcendentals of the pass-per-kernel tier (csrc/vp_crmath.h — HLSL's
 * pow(x, y) = exp2(y * log2(x)) as d3dcompiler lowers it, Shaders/convert/st2084.hlsl:9-25, with every step the correctly rounded fp32
 * function) evaluated on the device over n floats: fn = 0 log2f(x), 1 exp2f(x), 2 expf(x), 3 powf(x, y), 4 sinf(x), 5 cosf(x) (the last two:
 * the windowed sinc / jinc weights of the resize shaders, ps_interpolation_lanczos3.hlsl:50-59, ps_resize_onepass_jinc2.hlsl:44-101); y_dev is read by fn = 3 only.
 * The tests hold the result to a CPU evaluation of the same definition bit for bit.  DEVICE buffers; stream = a hipStream_t or NULL. */
int32_t mpcvr_eval_transcendental(int32_t fn, const float *x_dev, const float *y_dev, float *out_dev, size_t n, void *stream);
/* ... and the pass-per-kernel tier's Dolby Vision tail (csrc/vp_device.h: PQ EOTF -> LMS matrix -> PQ OETF, Shaders.cpp:844-859; level-2 trims
 * :766-773; ST2084ToLinear * scale; Hable; 2020 -> 709; pow 1/2.2, :870-923) over n PQ-coded RGB triples, cut off after `stage` (0 .. 5 in that
 * order): the tests hold every stage to the oracle's bit for bit.  lms9 / l2k5 as mpcvr_plan_dovi returns them.  DEVICE buffers of 3 n floats. */
int32_t mpcvr_eval_dovi_tail(int32_t stage, const float *rgb_dev, float *out_dev, size_t n, const float lms9[9], const float l2k5[5], int32_t l2_enabled,
                             float lum_scale, void *stream);
/* ... and the same functions compiled for the host, over HOST buffers (no GPU needed: the CPU suite's check of the definition) */
int32_t mpcvr_eval_transcendental_host(int32_t fn, const float *x, const float *y, float *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* MPCVR_H */
