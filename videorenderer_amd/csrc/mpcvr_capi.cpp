// mpcvr_capi.cpp — extern "C" surface of libmpcvr.so (include/mpcvr.h) over CHipVideoProcessor.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/mpcvr.h"
#include "hip_video_processor.h"

using mpcvr::CHipVideoProcessor;
using mpcvr::CRect;

struct mpcvr_ctx {
    CHipVideoProcessor vp;
};

static inline CRect ToRect(const mpcvr_rect *r) { return r ? CRect(r->left, r->top, r->right, r->bottom) : CRect(); }

extern "C" {

int32_t mpcvr_settings_default(mpcvr_settings *s)
{   // Settings_t::SetDefault — IVideoRenderer.h:140-185
    if (!s) return MPCVR_E_POINTER;
    std::memset(s, 0, sizeof(*s));
    s->iTexFormat = MPCVR_TEXFMT_AUTOINT;
    s->iChromaScaling = MPCVR_CHROMA_Bilinear;
    s->iUpscaling = MPCVR_UPSCALE_CatmullRom;
    s->iDownscaling = MPCVR_DOWNSCALE_Hamming;
    s->bInterpolateAt50pct = 1;
    s->bUseDither = 1;
    s->bDeintBlend = 0;
    s->bConvertToSdr = 1;
    s->iSDRDisplayNits = 125;
    s->output_format = MPCVR_OUT_BGRA8;
    s->flags = 0;
    return MPCVR_S_OK;
}

int32_t mpcvr_create(const mpcvr_settings *settings, int32_t device, mpcvr_ctx **out)
{
    if (!out) return MPCVR_E_POINTER;
    *out = nullptr;
    mpcvr_settings def;
    mpcvr_settings_default(&def);
    mpcvr_ctx *ctx = new (std::nothrow) mpcvr_ctx();
    if (!ctx) return MPCVR_E_OUTOFMEMORY;
    const int32_t hr = ctx->vp.Init(device, settings ? *settings : def);
    if (hr < 0) {
        std::fprintf(stderr, "mpcvr_create: %s\n", ctx->vp.LastError());
        delete ctx;
        return hr;
    }
    *out = ctx;
    return MPCVR_S_OK;
}

int32_t mpcvr_destroy(mpcvr_ctx *ctx)
{
    if (!ctx) return MPCVR_E_POINTER;
    delete ctx;
    return MPCVR_S_OK;
}

#define CTX_OR_FAIL() do { if (!ctx) return MPCVR_E_POINTER; } while (0)

int32_t mpcvr_set_stream(mpcvr_ctx *ctx, void *hip_stream) { CTX_OR_FAIL(); return ctx->vp.SetStream((hipStream_t)hip_stream); }
int32_t mpcvr_synchronize(mpcvr_ctx *ctx) { CTX_OR_FAIL(); return ctx->vp.Synchronize(); }

int32_t mpcvr_set_input(mpcvr_ctx *ctx, int32_t cformat, int32_t width, int32_t height, int32_t pitch,
                        const mpcvr_rect *src_rect, uint32_t extfmt)
{
    CTX_OR_FAIL();
    const CRect r = ToRect(src_rect);
    return ctx->vp.InitMediaType(cformat, width, height, pitch, src_rect ? &r : nullptr, extfmt);
}

int32_t mpcvr_set_video_rect(mpcvr_ctx *ctx, const mpcvr_rect *r) { CTX_OR_FAIL(); if (!r) return MPCVR_E_POINTER; return ctx->vp.SetVideoRect(ToRect(r)); }
int32_t mpcvr_set_window_rect(mpcvr_ctx *ctx, const mpcvr_rect *r) { CTX_OR_FAIL(); if (!r) return MPCVR_E_POINTER; return ctx->vp.SetWindowRect(ToRect(r)); }
int32_t mpcvr_set_rotation(mpcvr_ctx *ctx, int32_t degrees) { CTX_OR_FAIL(); return ctx->vp.SetRotation(degrees); }
int32_t mpcvr_set_error_diffusion_patience(mpcvr_ctx *ctx, int32_t polls) { CTX_OR_FAIL(); return ctx->vp.SetErrorDiffusionPatience(polls); }
int32_t mpcvr_set_flip(mpcvr_ctx *ctx, int32_t flip) { CTX_OR_FAIL(); return ctx->vp.SetFlip(flip != 0); }
int32_t mpcvr_set_sample_format(mpcvr_ctx *ctx, int32_t frame_format) { CTX_OR_FAIL(); return ctx->vp.SetSampleFormat(frame_format); }
int32_t mpcvr_set_hdr_output(mpcvr_ctx *ctx, int32_t enable, int32_t tone_map_type, float display_max_nits)
{ CTX_OR_FAIL(); return ctx->vp.SetHdrOutput(enable != 0, tone_map_type, display_max_nits); }
int32_t mpcvr_set_hdr_metadata(mpcvr_ctx *ctx, float min_mastering_nits, float max_mastering_nits, float max_cll, float max_fall)
{ CTX_OR_FAIL(); return ctx->vp.SetHdrMetadata(min_mastering_nits, max_mastering_nits, max_cll, max_fall); }

int32_t mpcvr_set_dovi_metadata(mpcvr_ctx *ctx, const mpcvr_dovi_metadata *md)
{ CTX_OR_FAIL(); return ctx->vp.SetDoviMetadata(md); }

int32_t mpcvr_configure(mpcvr_ctx *ctx, const mpcvr_settings *settings)
{
    CTX_OR_FAIL();
    if (!settings) return MPCVR_E_POINTER;
    return ctx->vp.Configure(*settings);
}

int32_t mpcvr_set_procamp(mpcvr_ctx *ctx, uint32_t flags, float brightness, float contrast, float hue, float saturation)
{
    CTX_OR_FAIL();
    return ctx->vp.SetProcAmpValues(flags, brightness, contrast, hue, saturation);
}

int32_t mpcvr_copy_sample(mpcvr_ctx *ctx, const void *data, int32_t pitch, int32_t mem_kind)
{
    CTX_OR_FAIL();
    return ctx->vp.CopySample(data, pitch, mem_kind);
}

int32_t mpcvr_process(mpcvr_ctx *ctx, void *dst_dev, int32_t dst_pitch, const mpcvr_rect *src_rect,
                      const mpcvr_rect *dst_rect, int32_t second_field)
{
    CTX_OR_FAIL();
    const CRect s = ToRect(src_rect), d = ToRect(dst_rect);
    return ctx->vp.Process(dst_dev, dst_pitch, src_rect ? &s : nullptr, dst_rect ? &d : nullptr, second_field != 0);
}

// the reference's call pattern over n frames — CopySample + Process per frame, one after the other (DX11VideoProcessor.cpp:2143-2200 -> :2730) —
// as ONE entry: what a render thread written in C or C++ does anyway; a scripting-language caller (bench.py, the tests) otherwise times its own
// foreign-function calls (two per frame, ~1.5 us each from ctypes: 6 % of a 50 us frame) instead of the path
int32_t mpcvr_process_frames(mpcvr_ctx *ctx, int32_t n, const void *const *samples, int32_t pitch, int32_t mem_kind, void *const *dsts_dev, int32_t dst_pitch)
{
    CTX_OR_FAIL();
    if (n < 0 || (n > 0 && (!samples || !dsts_dev))) return MPCVR_E_POINTER;
    for (int32_t i = 0; i < n; i++) {
        int32_t hr = ctx->vp.CopySample(samples[i], pitch, mem_kind);
        if (hr < 0) return hr;
        hr = ctx->vp.Process(dsts_dev[i], dst_pitch, nullptr, nullptr, false);
        if (hr < 0) return hr;
    }
    return MPCVR_S_OK;
}

int32_t mpcvr_render(mpcvr_ctx *ctx, int32_t field) { CTX_OR_FAIL(); return ctx->vp.Render(field); }

int32_t mpcvr_get_backbuffer(mpcvr_ctx *ctx, void **dev_ptr, int32_t *pitch, int32_t *width, int32_t *height)
{
    CTX_OR_FAIL();
    return ctx->vp.GetBackBuffer(dev_ptr, pitch, width, height);
}

int32_t mpcvr_get_current_image(mpcvr_ctx *ctx, void *host_bgra, size_t *size) { CTX_OR_FAIL(); return ctx->vp.GetCurentImage(host_bgra, size); }
int32_t mpcvr_get_displayed_image(mpcvr_ctx *ctx, void *host_pixels, size_t *size, int32_t deep_color, int32_t *width, int32_t *height, int32_t *bits_per_pixel)
{
    CTX_OR_FAIL();
    return ctx->vp.GetDisplayedImage(host_pixels, size, deep_color != 0, width, height, bits_per_pixel);
}
int32_t mpcvr_flush(mpcvr_ctx *ctx) { CTX_OR_FAIL(); ctx->vp.Flush(); return MPCVR_S_OK; }
int32_t mpcvr_reset(mpcvr_ctx *ctx) { CTX_OR_FAIL(); return ctx->vp.Reset(); }

int32_t mpcvr_process_batch(mpcvr_ctx *ctx, int32_t n, const void *const *srcs, void *const *dsts, int32_t dst_pitch)
{
    CTX_OR_FAIL();
    return ctx->vp.ProcessBatch(n, srcs, dsts, dst_pitch);
}

int32_t mpcvr_process_batch_dovi(mpcvr_ctx *ctx, int32_t n, const void *const *srcs, void *const *dsts, int32_t dst_pitch, const mpcvr_dovi_metadata *rpus)
{
    CTX_OR_FAIL();
    return ctx->vp.ProcessBatchDovi(n, srcs, dsts, dst_pitch, rpus);
}

int32_t mpcvr_get_param_blob(mpcvr_ctx *ctx, void *buf, size_t *size) { CTX_OR_FAIL(); return ctx->vp.GetParamBlob(buf, size); }
int32_t mpcvr_set_param_blob(mpcvr_ctx *ctx, const void *buf, size_t size) { CTX_OR_FAIL(); return ctx->vp.SetParamBlob(buf, size); }
int32_t mpcvr_broadcast_param_blob_begin(mpcvr_ctx *ctx, void *nccl_comm, int32_t root, int32_t rank) { CTX_OR_FAIL(); return ctx->vp.BroadcastParamBlobBegin(nccl_comm, root, rank); }
int32_t mpcvr_broadcast_param_blob_end(mpcvr_ctx *ctx) { CTX_OR_FAIL(); return ctx->vp.BroadcastParamBlobEnd(); }
int32_t mpcvr_broadcast_param_blob(mpcvr_ctx *ctx, void *nccl_comm, int32_t root, int32_t rank)
{
    CTX_OR_FAIL();
    const int32_t hr = ctx->vp.BroadcastParamBlobBegin(nccl_comm, root, rank);
    return hr < 0 ? hr : ctx->vp.BroadcastParamBlobEnd();
}

int32_t mpcvr_get_color_matrix(mpcvr_ctx *ctx, float out12[12]) { CTX_OR_FAIL(); if (!out12) return MPCVR_E_POINTER; return ctx->vp.GetColorMatrix(out12); }
int32_t mpcvr_get_extfmt(mpcvr_ctx *ctx, uint32_t *extfmt) { CTX_OR_FAIL(); if (!extfmt) return MPCVR_E_POINTER; return ctx->vp.GetExtFmt(extfmt); }
int32_t mpcvr_get_frame_bytes(mpcvr_ctx *ctx, size_t *bytes, int32_t *pitch) { CTX_OR_FAIL(); return ctx->vp.GetFrameBytes(bytes, pitch); }

int32_t mpcvr_get_path_info(mpcvr_ctx *ctx, char *buf, size_t buf_size)
{
    CTX_OR_FAIL();
    if (!buf || !buf_size) return MPCVR_E_POINTER;
    const std::string s = ctx->vp.GetPathInfo();
    std::snprintf(buf, buf_size, "%s", s.c_str());
    return MPCVR_S_OK;
}

int32_t mpcvr_get_last_batch_info(mpcvr_ctx *ctx, char *buf, size_t buf_size)
{
    CTX_OR_FAIL();
    if (!buf || !buf_size) return MPCVR_E_POINTER;
    std::snprintf(buf, buf_size, "%s", ctx->vp.GetLastBatchInfo().c_str());
    return MPCVR_S_OK;
}

const char *mpcvr_last_error(mpcvr_ctx *ctx) { return ctx ? ctx->vp.LastError() : "null context"; }
const char *mpcvr_version(void) { return "mpcvr-mi355x 0.1 (gfx950)"; }

int32_t mpcvr_get_last_process_ms(mpcvr_ctx *ctx, float *ms) { CTX_OR_FAIL(); return ctx->vp.GetLastProcessMs(ms); }
int32_t mpcvr_get_last_timings(mpcvr_ctx *ctx, float *copy_host_ms, float *upload_ms, float *process_ms, float *readback_ms)
{
    CTX_OR_FAIL();
    return ctx->vp.GetLastTimings(copy_host_ms, upload_ms, process_ms, readback_ms);
}

}  // extern "C"

// ---- host-side parameter maths without a context (no GPU needed) --------------------------------
// What the reference computes on the CPU before it ever touches the device: format table, extended
// format defaults, colour / gamut matrices, resize weights, tap tables and the pass plan.
#include "vp_plan.h"

extern "C" {

int32_t mpcvr_plan_frame_layout(int32_t cformat, int32_t width, int32_t height, int32_t *pitch, size_t *bytes)
{
    const mpcvr::FmtConvParams *f = mpcvr::GetFmtConvParams(cformat);
    if (!f) return MPCVR_E_NOTIMPL;
    const int p = mpcvr::DefaultPitch(*f, width);
    if (pitch) *pitch = p;
    if (bytes) *bytes = (size_t)p * mpcvr::SourceLines(*f, height);
    return MPCVR_S_OK;
}

int32_t mpcvr_plan_color_matrix(int32_t cformat, int32_t rect_w, int32_t rect_h, uint32_t extfmt,
                                float brightness, float contrast, float hue, float saturation,
                                float out12[12], uint32_t *extfmt_out)
{
    const mpcvr::FmtConvParams *f = mpcvr::GetFmtConvParams(cformat);
    if (!f) return MPCVR_E_NOTIMPL;
    if (!out12) return MPCVR_E_POINTER;
    const mpcvr::ExtFmt ex = mpcvr::SpecifyExtendedFormat(mpcvr::ExtFmt{extfmt}, *f, rect_w, rect_h);
    mpcvr::ProcAmp pa;
    pa.brightness = brightness; pa.contrast = contrast; pa.hue = hue; pa.saturation = saturation;
    mpcvr::ComputeColorMatrix(ex, *f, pa, out12);
    if (extfmt_out) *extfmt_out = ex.value;
    return MPCVR_S_OK;
}

int32_t mpcvr_plan_gamut_2020_to_709(float out9[9])
{
    if (!out9) return MPCVR_E_POINTER;
    mpcvr::ComputeGamut2020to709(out9);
    return MPCVR_S_OK;
}

int32_t mpcvr_plan_pq_lut(float lum_scale, float out4096[4096])
{
    static_assert(mpcvr::kPqLutSize == 4096, "header documents 4096 entries");
    if (!out4096) return MPCVR_E_POINTER;
    mpcvr::BuildPqSdrLut(lum_scale, out4096);
    return MPCVR_S_OK;
}

int32_t mpcvr_plan_final_pass_multiplier(int32_t quant, int32_t maxv, uint32_t *multiplier)
{
    if (!multiplier) return MPCVR_E_POINTER;
    *multiplier = mpcvr::FinalPassMultiplier(quant, maxv);
    return MPCVR_S_OK;
}

int32_t mpcvr_plan_correction_matrices(float fix_bt2020_16[16], float fix_ycgco_16[16], float gamut9[9])
{
    if (!fix_bt2020_16 || !fix_ycgco_16 || !gamut9) return MPCVR_E_POINTER;
    mpcvr::CorrectionMatrices(fix_bt2020_16, fix_ycgco_16, gamut9);
    return MPCVR_S_OK;
}

int32_t mpcvr_correction_pass(int32_t kind, const void *src, int32_t src_pitch, int32_t src_fmt,
                              void *dst, int32_t dst_pitch, int32_t dst_fmt, int32_t w, int32_t h, int32_t sdr_nits, void *stream)
{
    if (!src || !dst) return MPCVR_E_POINTER;
    if (kind < MPCVR_CORR_FIX_BT2020 || kind > MPCVR_CORR_CONVERT_HLG_TO_PQ || w <= 0 || h <= 0 || sdr_nits <= 0 ||
        src_pitch < w * 4 || dst_pitch < w * 4 || (src_pitch & 3) || (dst_pitch & 3) ||
        (src_fmt != MPCVR_OUT_BGRA8 && src_fmt != MPCVR_OUT_RGB10A2) || (dst_fmt != MPCVR_OUT_BGRA8 && dst_fmt != MPCVR_OUT_RGB10A2))
        return MPCVR_E_INVALIDARG;
    float fix2020[16], fixycgco[16], gamut[9];
    mpcvr::CorrectionMatrices(fix2020, fixycgco, gamut);
    const mpcvr::Surface in{const_cast<void *>(src), src_pitch, w, h, src_fmt == MPCVR_OUT_RGB10A2 ? mpcvr::SF_RGB10A2 : mpcvr::SF_BGRA8};
    const mpcvr::Surface out{dst, dst_pitch, w, h, dst_fmt == MPCVR_OUT_RGB10A2 ? mpcvr::SF_RGB10A2 : mpcvr::SF_BGRA8};
    const hipError_t e = mpcvr::LaunchCorrection(kind, in, out, kind == MPCVR_CORR_FIX_YCGCO ? fixycgco : fix2020, gamut,
                                                 10000.0f / (float)sdr_nits, (hipStream_t)stream);
    return e == hipSuccess ? MPCVR_S_OK : MPCVR_E_FAIL;
}

int32_t mpcvr_plan_dovi(const mpcvr_dovi_metadata *md, int32_t display_nits, float *cb705, int32_t *has_mmr,
                        float lms9[9], float l2k[5], int32_t *l2_enabled, uint32_t l1_nits[3], int32_t *l1_present)
{
    if (!md) return MPCVR_E_POINTER;
    if (!mpcvr::CheckDoviCurves(*md)) return MPCVR_E_INVALIDARG;
    if (cb705 || has_mmr) {
        mpcvr::DoviParams P{};
        mpcvr::PackDoviCurves(*md, &P);
        if (has_mmr) *has_mmr = P.has_mmr;
        if (cb705)
            for (int c = 0; c < 3; c++) {
                float *o = cb705 + c * 235;
                const mpcvr::DoviCurve &cv = P.curves[c];
                std::memcpy(o, cv.pivots, sizeof(float) * 7);
                std::memcpy(o + 7, cv.coeffs, sizeof(float) * 32);
                std::memcpy(o + 39, cv.mmr, sizeof(float) * 192);
                o[231] = (float)cv.methods; o[232] = (float)cv.mmr_single; o[233] = (float)cv.min_order; o[234] = (float)cv.max_order;
            }
    }
    if (lms9) mpcvr::DoviLmsMatrix(*md, lms9);
    if (l2k || l2_enabled) {
        float k[5];
        const bool on = mpcvr::DoviL2Constants(*md, display_nits, k);
        if (l2k) std::memcpy(l2k, k, sizeof(k));
        if (l2_enabled) *l2_enabled = on ? 1 : 0;
    }
    if (l1_nits || l1_present) {
        uint32_t v[3];
        const bool on = mpcvr::DoviL1Nits(*md, v);
        if (l1_nits) std::memcpy(l1_nits, v, sizeof(v));
        if (l1_present) *l1_present = on ? 1 : 0;
    }
    return MPCVR_S_OK;
}

int32_t mpcvr_plan_upscale_weights(int32_t iUpscaling, float t, float w6[6])
{
    if (!w6) return MPCVR_E_POINTER;
    return mpcvr::UpscaleWeights(iUpscaling, t, w6);      // tap count (4/6) or 0
}

int32_t mpcvr_plan_axis_taps(int32_t kind, int32_t method, int32_t src_l, int32_t src_len, int32_t n_out,
                             int32_t tex_len, uint32_t flags, int32_t cap_taps, int32_t *idx, float *w,
                             float *wsum, int32_t *ntaps, int32_t *normalise)
{
    if (!ntaps) return MPCVR_E_POINTER;
    if (n_out <= 0 || src_len <= 0 || tex_len <= 0) return MPCVR_E_INVALIDARG;
    mpcvr::HostAxisTaps h;
    if (!mpcvr::BuildAxisTaps(mpcvr::Resizer{kind, method}, src_l, src_len, n_out, tex_len, flags, &h)) return MPCVR_E_NOTIMPL;
    *ntaps = h.ntaps;
    if (normalise) *normalise = h.normalise;
    if (h.ntaps > cap_taps) return MPCVR_S_FALSE;          // caller's buffers too small: only *ntaps is valid
    if (idx) std::memcpy(idx, h.idx.data(), h.idx.size() * sizeof(int32_t));
    if (w) std::memcpy(w, h.w.data(), h.w.size() * sizeof(float));
    if (wsum && h.normalise) std::memcpy(wsum, h.wsum.data(), h.wsum.size() * sizeof(float));
    return MPCVR_S_OK;
}

int32_t mpcvr_plan_strip(int32_t kind_x, int32_t method_x, int32_t kind_y, int32_t method_y, int32_t src_w, int32_t src_h,
                         int32_t out_w, int32_t out_h, uint32_t flags, int32_t out8[8], int32_t *yrange, int32_t *xstrip,
                         int32_t *xi_t, float *xw_t, int32_t *yi, float *yw)
{
    if (!out8) return MPCVR_E_POINTER;
    if (src_w <= 0 || src_h <= 0 || out_w <= 0 || out_h <= 0) return MPCVR_E_INVALIDARG;
    mpcvr::HostAxisTaps hx, hy;
    // the draws as UpdatePlan builds them: X from the convert output (rect at the origin), Y from m_TexResize (src_h rows)
    if (!mpcvr::BuildAxisTaps(mpcvr::Resizer{kind_x, method_x}, 0, src_w, out_w, src_w, flags, &hx)) return MPCVR_E_NOTIMPL;
    if (!mpcvr::BuildAxisTaps(mpcvr::Resizer{kind_y, method_y}, 0, src_h, out_h, src_h, flags, &hy)) return MPCVR_E_NOTIMPL;
    mpcvr::StripPlan sp;
    if (!mpcvr::PlanFusedStrip(hx, hy, out_w, out_h, src_w, src_h, &sp)) return MPCVR_E_NOTIMPL;
    const int strips = (out_w + sp.strip_w - 1) / sp.strip_w;
    const int per_wave = 2 * sp.acols * 8 + sp.ring * 64 * (sp.pxl == 2 ? 12 : 8);
    const int32_t o[8] = {sp.nt, sp.pxl, sp.strip_w, sp.ring, sp.acols, strips, per_wave, 0};
    std::memcpy(out8, o, sizeof(o));
    if (yrange) std::memcpy(yrange, sp.yrange.data(), sp.yrange.size() * sizeof(int32_t));
    if (xstrip) std::memcpy(xstrip, sp.xstrip.data(), sp.xstrip.size() * sizeof(int32_t));
    if (xi_t) std::memcpy(xi_t, sp.xi_t.data(), sp.xi_t.size() * sizeof(int32_t));
    if (xw_t) std::memcpy(xw_t, sp.xw_t.data(), sp.xw_t.size() * sizeof(float));
    if (yi) std::memcpy(yi, sp.yi.data(), sp.yi.size() * sizeof(int32_t));
    if (yw) std::memcpy(yw, sp.yw.data(), sp.yw.size() * sizeof(float));
    return MPCVR_S_OK;
}

int32_t mpcvr_plan_period(int32_t method, int32_t src_w, int32_t src_h, int32_t out_w, int32_t out_h, uint32_t flags,
                          int32_t out6[6], int32_t *xi_t, float *xw_t, float *yw, int32_t *xstrip, int32_t *strip_w)
{
    if (!out6) return MPCVR_E_POINTER;
    if (src_w <= 0 || src_h <= 0 || out_w <= 0 || out_h <= 0) return MPCVR_E_INVALIDARG;
    mpcvr::HostAxisTaps hx, hy;
    if (!mpcvr::BuildAxisTaps(mpcvr::Resizer{mpcvr::RS_UP, method}, 0, src_w, out_w, src_w, flags, &hx)) return MPCVR_E_NOTIMPL;
    if (!mpcvr::BuildAxisTaps(mpcvr::Resizer{mpcvr::RS_UP, method}, 0, src_h, out_h, src_h, flags, &hy)) return MPCVR_E_NOTIMPL;
    const bool q1 = method == MPCVR_UPSCALE_Lanczos3 && !(flags & MPCVR_FLAG_LANCZOS3_FIXED);
    mpcvr::PeriodPlan pp;
    if (!mpcvr::PlanFusedPeriod(hx, hy, out_w, out_h, src_w, src_h, q1, &pp)) return MPCVR_E_NOTIMPL;
    const int32_t o[6] = {pp.P, pp.Q, pp.nt, (out_w + pp.strip_w - 1) / pp.strip_w, pp.acols, 6 * pp.P / pp.Q};
    if (strip_w) *strip_w = pp.strip_w;
    std::memcpy(out6, o, sizeof(o));
    if (xi_t) std::memcpy(xi_t, pp.xi_t.data(), pp.xi_t.size() * sizeof(int32_t));
    if (xw_t) std::memcpy(xw_t, pp.xw_t.data(), pp.xw_t.size() * sizeof(float));
    if (yw) std::memcpy(yw, pp.yw.data(), pp.yw.size() * sizeof(float));
    if (xstrip) std::memcpy(xstrip, pp.xstrip.data(), pp.xstrip.size() * sizeof(int32_t));
    return MPCVR_S_OK;
}

int32_t mpcvr_plan_hdr10_params(float min_mastering, float max_mastering, float max_cll, float max_fall, float display_max,
                                int32_t selection, uint32_t out6[6])
{
    if (!out6) return MPCVR_E_POINTER;
    mpcvr::HdrToneMapParams k{min_mastering, max_mastering, max_cll, max_fall, display_max, selection};
    mpcvr::SanitiseHdr10Params(&k);
    std::memcpy(out6, &k, 20);
    out6[5] = (uint32_t)k.selection;
    return MPCVR_S_OK;
}

int32_t mpcvr_plan_pq_eotf_table(float *out, int32_t capacity, int32_t *count)
{
    constexpr int32_t n = mpcvr::kEotfLutSize + 1;
    if (count) *count = n;
    if (!out) return count ? MPCVR_S_OK : MPCVR_E_POINTER;
    if (capacity < n) return MPCVR_E_INVALIDARG;
    mpcvr::BuildPqEotfLut(out);
    return MPCVR_S_OK;
}

// DEPRECATED (kept so that a caller built against the round-3 header still links and cannot be overrun): the same function on the 4096-entry
// grid that header promised, computed from the table's definition (vp_plan.cpp BuildPqEotfLut), not cut out of the larger table
int32_t mpcvr_plan_pq_eotf_lut(float out[4096])
{
    if (!out) return MPCVR_E_POINTER;
    const double m1 = 2610.0 / (4096.0 * 4.0), m2 = (2523.0 / 4096.0) * 128.0;
    const double c1 = 3424.0 / 4096.0, c2 = (2413.0 / 4096.0) * 32.0, c3 = (2392.0 / 4096.0) * 32.0;
    for (int i = 0; i < 4096; i++) {
        const double t = (double)i / 4095.0;
        double x = std::pow(t * t, 1.0 / m2);
        x = std::fmax(x - c1, 0.0) / (c2 - c3 * x);
        const double l = x > 0.0 ? std::log2(x) / m1 : -1e9;
        out[i] = l > -150.0 ? (float)l : -150.0f;
    }
    return MPCVR_S_OK;
}

int32_t mpcvr_plan_describe(const mpcvr_settings *s, int32_t cformat, int32_t rect_w, int32_t rect_h,
                            const mpcvr_rect *video_rect, int32_t window_w, int32_t window_h,
                            char *buf, size_t buf_size)
{
    if (!s || !video_rect || !buf || !buf_size) return MPCVR_E_POINTER;
    const mpcvr::FmtConvParams *f = mpcvr::GetFmtConvParams(cformat);
    if (!f) return MPCVR_E_NOTIMPL;
    const mpcvr::PlanGeometry g{rect_w, rect_h, video_rect->left, video_rect->top, video_rect->right,
                                video_rect->bottom, window_w, window_h};
    mpcvr::PassPlan plan;
    std::string why;
    if (!mpcvr::DecidePlan(s->iTexFormat, s->iChromaScaling, s->iUpscaling, s->iDownscaling, s->bInterpolateAt50pct,
                           s->bUseDither, s->output_format, s->flags, *f, g, &plan, &why)) {
        std::snprintf(buf, buf_size, "%s", why.c_str());
        return MPCVR_E_NOTIMPL;
    }
    std::snprintf(buf, buf_size, "%s;internal=%d;swap=%d;final=%d", plan.describe().c_str(), plan.internal_fmt,
                  plan.swap_fmt, plan.final_pass ? 1 : 0);
    return MPCVR_S_OK;
}

}  // extern "C"

// verification aid: the plain tier's Dolby Vision tail over an array, stage by stage (csrc/vp_kernels.hip: k_eval_dovi_tail)
int32_t mpcvr_eval_dovi_tail(int32_t stage, const float *rgb_dev, float *out_dev, size_t n, const float lms9[9], const float l2k5[5], int32_t l2_enabled, float lum_scale, void *stream)
{
    if (!rgb_dev || !out_dev || !lms9 || !l2k5) return MPCVR_E_POINTER;
    if (stage < 0 || stage > 5) return MPCVR_E_INVALIDARG;
    float gamut[9];
    mpcvr::ComputeGamut2020to709(gamut);
    return mpcvr::LaunchEvalDoviTail(rgb_dev, out_dev, n, lms9, l2k5, gamut, l2_enabled, lum_scale, stage, (hipStream_t)stream) == hipSuccess ? MPCVR_S_OK : MPCVR_E_FAIL;
}
