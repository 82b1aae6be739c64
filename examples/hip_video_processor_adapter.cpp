// examples/hip_video_processor_adapter.cpp — the third CVideoProcessor backend a maintainer of MPC Video Renderer would add
// (Source/HipVideoProcessor.cpp): every method of the reference's processor interface (Source/VideoProcessor.h:171-236) that belongs to the
// shader video-processor path forwards to include/mpcvr.h; the rest (subtitles, OSD, statistics, display switching) keeps the defaults.
//
// Compiled by tests/test_adapter_compiles.py — and LINKED with a driver and run on the GPU box (tests/adapter_env/build_adapter.py,
// adapter_driver.cpp; tests/test_parity_gpu.py::test_one_frame_through_the_adapter_class) — against the REAL declarations of CVideoProcessor —
// cut out of the reference header at test time — and the REAL Settings_t / enumerators of Source/IVideoRenderer.h, behind stand-ins for the
// Windows / DirectShow types they mention (tests/adapter_env/): a method the reference adds, renames or re-types, a Settings_t field that
// disappears, or an enumerator whose value moves away from mpcvr.h's fails that test.  In the reference tree it is built with the real headers.
#include "VideoProcessor.h"

extern "C" {
#include "mpcvr.h"
}

// the enumerators cross the boundary as plain numbers: mpcvr.h's must be the reference's (IVideoRenderer.h:25-72)
#define SAME(a, b) ((int)(a) == (int)(b))
static_assert(SAME(TEXFMT_AUTOINT, MPCVR_TEXFMT_AUTOINT) && SAME(TEXFMT_8INT, MPCVR_TEXFMT_8INT) && SAME(TEXFMT_10INT, MPCVR_TEXFMT_10INT) && SAME(TEXFMT_16FLOAT, MPCVR_TEXFMT_16FLOAT), "TEXFMT_*");
static_assert(SAME(CHROMA_Nearest, MPCVR_CHROMA_Nearest) && SAME(CHROMA_Bilinear, MPCVR_CHROMA_Bilinear) && SAME(CHROMA_CatmullRom, MPCVR_CHROMA_CatmullRom), "CHROMA_*");
static_assert(SAME(UPSCALE_Nearest, MPCVR_UPSCALE_Nearest) && SAME(UPSCALE_Mitchell, MPCVR_UPSCALE_Mitchell) && SAME(UPSCALE_CatmullRom, MPCVR_UPSCALE_CatmullRom) &&
              SAME(UPSCALE_Lanczos2, MPCVR_UPSCALE_Lanczos2) && SAME(UPSCALE_Lanczos3, MPCVR_UPSCALE_Lanczos3) && SAME(UPSCALE_Jinc2, MPCVR_UPSCALE_Jinc2), "UPSCALE_*");
static_assert((int)UPSCALE_COUNT == (int)MPCVR_UPSCALE_Spline36_EXT, "the Spline36 extension takes the first value the reference does not use");
static_assert(SAME(DOWNSCALE_Box, MPCVR_DOWNSCALE_Box) && SAME(DOWNSCALE_Bilinear, MPCVR_DOWNSCALE_Bilinear) && SAME(DOWNSCALE_Hamming, MPCVR_DOWNSCALE_Hamming) &&
              SAME(DOWNSCALE_Bicubic, MPCVR_DOWNSCALE_Bicubic) && SAME(DOWNSCALE_BicubicSharp, MPCVR_DOWNSCALE_BicubicSharp) && SAME(DOWNSCALE_Lanczos, MPCVR_DOWNSCALE_Lanczos), "DOWNSCALE_*");
static_assert(SDR_NITS_DEF == 125, "iSDRDisplayNits default");
static_assert((int)CF_NV12 == MPCVR_CF_NV12 && (int)CF_P010 == MPCVR_CF_P010 && (int)CF_YV12 == MPCVR_CF_YV12 && (int)CF_YUV420P10 == MPCVR_CF_YUV420P10 &&
              (int)CF_Y16 == MPCVR_CF_Y16, "ColorFormat_t (Helper.h:86-127)");

enum : int { VP_HIP = 12 };           // next to VP_DX9 / VP_DX11 (VideoProcessor.h:28-31)

class CHipVideoProcessorAdapter : public CVideoProcessor
{
    mpcvr_ctx *m_ctx = nullptr;
    mpcvr_settings m_cfg{};
    bool m_bInit = false;
    // displayConfig.HDRSupported() && m_bHdrDisplayModeEnabled (DX11VideoProcessor.cpp:481-493): the display's state, not a setting.  HDR sources
    // pass through / are tone-mapped for an HDR display only when it is set (:1476, :2948); bHdrPassthrough defaults to TRUE in Settings_t, so
    // without this gate every PQ frame on an SDR display would leave unconverted (found by RUNNING the adapter: round 6).
    bool m_bHdrPassthroughSupport = false;
    HRESULT ApplyHdrOutput(const Settings_t &s)
    {
        const bool on = m_bHdrPassthroughSupport && (s.bHdrPassthrough || s.bHdrLocalToneMapping);
        return mpcvr_set_hdr_output(m_ctx, on, s.bHdrLocalToneMapping ? s.iHdrLocalToneMappingType : 0, (float)s.iHdrDisplayMaxNits);
    }

    static mpcvr_settings FromSettings(const Settings_t &s, bool tenBitOutput)
    {
        mpcvr_settings c;
        mpcvr_settings_default(&c);
        c.iTexFormat = s.iTexFormat;
        c.iChromaScaling = s.iChromaScaling;
        c.iUpscaling = s.iUpscaling;
        c.iDownscaling = s.iDownscaling;
        c.bInterpolateAt50pct = s.bInterpolateAt50pct;
        c.bUseDither = s.bUseDither;
        c.bDeintBlend = s.bDeintBlend;
        c.bConvertToSdr = s.bConvertToSdr;
        c.iSDRDisplayNits = s.iSDRDisplayNits;
        c.output_format = tenBitOutput ? MPCVR_OUT_RGB10A2 : MPCVR_OUT_BGRA8;
        return c;
    }
    static mpcvr_rect ToRect(const CRect &r) { return mpcvr_rect{r.left, r.top, r.right, r.bottom}; }

public:
    CHipVideoProcessorAdapter(CMpcVideoRenderer *pFilter, const Settings_t &config, HRESULT &hr) : CVideoProcessor(pFilter)
    {
        m_cfg = FromSettings(config, /*tenBitOutput*/ false);
        hr = mpcvr_create(&m_cfg, /*device*/ 0, &m_ctx);
        if (SUCCEEDED(hr)) hr = ApplyHdrOutput(config);
    }
    ~CHipVideoProcessorAdapter() override { mpcvr_destroy(m_ctx); }

    // SetDisplayInfo's part that matters here (:481-493): the display entered / left HDR10 mode
    void SetHdrDisplay(bool hdrSupportedAndEnabled, const Settings_t &s) { m_bHdrPassthroughSupport = hdrSupportedAndEnabled; (void)ApplyHdrOutput(s); }

    int Type() override { return VP_HIP; }
    HRESULT Init(const HWND hwnd, bool /*displayHdrChanged*/, bool *pChangeDevice = nullptr) override
    {
        m_hWnd = hwnd;
        if (pChangeDevice) *pChangeDevice = false;
        m_bInit = m_ctx != nullptr;
        return m_bInit ? S_OK : E_FAIL;
    }
    bool IsInit() const override { return m_bInit; }

    BOOL VerifyMediaType(const CMediaType *pmt) override { return GetFmtConvParams(pmt).cformat != CF_NONE; }
    BOOL InitMediaType(const CMediaType *pmt) override
    {
        const FmtConvParams_t &fmt = GetFmtConvParams(pmt);                                   // DX11VideoProcessor.cpp:1742-1768
        const VIDEOINFOHEADER2 *vih2 = (const VIDEOINFOHEADER2 *)pmt->pbFormat;
        const BITMAPINFOHEADER &bih = vih2->bmiHeader;
        const mpcvr_rect src{vih2->rcSource.left, vih2->rcSource.top, vih2->rcSource.right, vih2->rcSource.bottom};
        DXVA2_ExtendedFormat ex{};
        if (vih2->dwControlFlags & (AMCONTROL_USED | AMCONTROL_COLORINFO_PRESENT)) {              // :1763-1766 (the decoder's DXVA2_ExtendedFormat bits)
            ex.value = (LONG)vih2->dwControlFlags;
            ex.SampleFormat = AMCONTROL_USED | AMCONTROL_COLORINFO_PRESENT;                       // "ignore other flags"
        }
        const HRESULT hr = mpcvr_set_input(m_ctx, (int32_t)fmt.cformat, bih.biWidth, std::labs(bih.biHeight), /*pitch: the reference's rule*/ 0, &src, (uint32_t)ex.value);
        if (hr != MPCVR_S_OK) return 0;
        m_srcParams = fmt;
        m_srcWidth = (UINT)bih.biWidth; m_srcHeight = (UINT)std::labs(bih.biHeight);
        m_srcRect = CRect(src.left, src.top, src.right, src.bottom);
        m_decExFmt = ex;
        size_t bytes = 0; int32_t pitch = 0;
        if (mpcvr_get_frame_bytes(m_ctx, &bytes, &pitch) == MPCVR_S_OK) m_srcPitch = pitch;
        return 1;
    }
    BOOL GetAlignmentSize(const CMediaType &, SIZE &) override { return 0; }                      // (no allocator-imposed alignment on this path)

    HRESULT ProcessSample(IMediaSample *pSample) override                                         // DX11VideoProcessor.cpp:2143-2200 -> CopySample :2202
    {
        BYTE *data = nullptr;
        HRESULT hr = pSample->GetPointer(&data);
        if (FAILED(hr)) return hr;
        hr = mpcvr_copy_sample(m_ctx, data, m_srcPitch, MPCVR_MEM_HOST);
        if (SUCCEEDED(hr)) hr = Render(1, 0);
        return hr;
    }
    // the Dolby Vision RPU of a sample (IID_MediaSideDataDOVIMetadataV2 branch of CopySample, :2270-2520): a field-by-field copy out of the packed LAV structure
    HRESULT SetDoviMetadata(const MediaSideDataDOVIMetadata *rpu)
    {
        if (!rpu) return mpcvr_set_dovi_metadata(m_ctx, nullptr);
        if (!CheckDoviMetadata(rpu, 1)) return S_FALSE;
        const mpcvr_dovi_metadata md = ToMpcvr(*rpu);
        return mpcvr_set_dovi_metadata(m_ctx, &md);
    }
    HRESULT Render(int field, const REFERENCE_TIME /*frameStartTime*/) override                  // :2599-2813 minus Present
    {
        const HRESULT hr = mpcvr_render(m_ctx, field);
        if (hr == MPCVR_S_OK) {
            void *p = nullptr; int32_t pitch = 0, w = 0, h = 0;
            (void)mpcvr_get_backbuffer(m_ctx, &p, &pitch, &w, &h);        // the adapter presents / interop-copies p
        }
        return hr;
    }
    HRESULT FillBlack() override { return S_OK; }
    void Flush() override { (void)mpcvr_flush(m_ctx); }
    HRESULT Reset(bool /*bDisplayModeChange*/) override { return mpcvr_reset(m_ctx); }

    void SetVideoRect(const CRect &r) override { m_videoRect = r; const mpcvr_rect q = ToRect(r); (void)mpcvr_set_video_rect(m_ctx, &q); }
    HRESULT SetWindowRect(const CRect &r) override { m_windowRect = r; const mpcvr_rect q = ToRect(r); return mpcvr_set_window_rect(m_ctx, &q); }
    void Configure(const Settings_t &s) override
    {
        m_cfg = FromSettings(s, false);
        (void)mpcvr_configure(m_ctx, &m_cfg);
        (void)ApplyHdrOutput(s);
    }
    void SetRotation(int value) override { if (mpcvr_set_rotation(m_ctx, value) == MPCVR_S_OK) m_iRotation = value; }
    void SetFlipForwarded(bool value) { SetFlip(value); (void)mpcvr_set_flip(m_ctx, value); }    // (SetFlip itself is not virtual: VideoProcessor.h:210)

    HRESULT GetCurentImage(long *pDIBImage) override                                              // :3493-3608, two-call size protocol as VideoRenderer.cpp:979-988
    {
        size_t size = 0;
        HRESULT hr = mpcvr_get_current_image(m_ctx, nullptr, &size);
        if (FAILED(hr)) return hr;
        BITMAPINFOHEADER *bih = (BITMAPINFOHEADER *)pDIBImage;
        std::memset(bih, 0, sizeof(*bih));
        LONG w = m_srcRect.Width(), h = m_srcRect.Height();                                       // the snapshot is SOURCE-rect sized (:3495-3502), not the window
        if (m_iRotation == 90 || m_iRotation == 270) { const LONG t = w; w = h; h = t; }
        bih->biSize = sizeof(*bih); bih->biWidth = w; bih->biHeight = -h;
        bih->biBitCount = 32; bih->biPlanes = 1; bih->biSizeImage = (DWORD)size;
        return mpcvr_get_current_image(m_ctx, bih + 1, &size);
    }
    HRESULT GetDisplayedImage(BYTE **ppDib, unsigned *pSize) override                             // :3610-3683: header + the back buffer's pixels in a LocalAlloc block
    {
        size_t size = 0; int32_t w = 0, h = 0, bits = 0;
        HRESULT hr = mpcvr_get_displayed_image(m_ctx, nullptr, &size, m_bAllowDeepColorBitmaps, &w, &h, &bits);
        if (FAILED(hr)) return hr;
        *pSize = (unsigned)(sizeof(BITMAPINFOHEADER) + size);
        BYTE *p = (BYTE *)LocalAlloc(LMEM_FIXED, *pSize);                                         // "only this allocator can be used" (:3659); the caller frees it
        if (!p) return E_OUTOFMEMORY;
        BITMAPINFOHEADER *bih = (BITMAPINFOHEADER *)p;
        std::memset(bih, 0, sizeof(*bih));
        bih->biSize = sizeof(*bih); bih->biWidth = w; bih->biHeight = -h;                         // top-down RGB bitmap
        bih->biBitCount = (decltype(bih->biBitCount))bits; bih->biPlanes = 1; bih->biSizeImage = (DWORD)size;
        hr = mpcvr_get_displayed_image(m_ctx, bih + 1, &size, m_bAllowDeepColorBitmaps, nullptr, nullptr, nullptr);
        if (SUCCEEDED(hr)) *ppDib = p; else LocalFree(p);
        return hr;
    }
    HRESULT GetVPInfo(std::wstring &str) override
    {
        char buf[512];
        const HRESULT hr = mpcvr_get_path_info(m_ctx, buf, sizeof buf);
        if (SUCCEEDED(hr)) { str = L"HIP shader video processor: "; for (const char *p = buf; *p; p++) str += (wchar_t)*p; }
        return hr;
    }
    void CalcStatsParams() override {}

private:
    void UpdateStatsStatic() override {}

    static mpcvr_dovi_metadata ToMpcvr(const MediaSideDataDOVIMetadata &s)                        // Include/IMediaSideData.h:154-330
    {
        mpcvr_dovi_metadata d{};
        d.bl_bit_depth = s.Header.bl_bit_depth; d.coef_log2_denom = s.Header.coef_log2_denom;
        d.source_max_pq = s.ColorMetadata.source_max_pq;
        for (int i = 0; i < 9; i++) { d.ycc_to_rgb_matrix[i] = s.ColorMetadata.ycc_to_rgb_matrix[i]; d.rgb_to_lms_matrix[i] = s.ColorMetadata.rgb_to_lms_matrix[i]; }
        for (int i = 0; i < 3; i++) d.ycc_to_rgb_offset[i] = s.ColorMetadata.ycc_to_rgb_offset[i];
        for (int c = 0; c < 3; c++) {
            const auto &in = s.Mapping.curves[c];
            auto &out = d.curves[c];
            out.num_pivots = in.num_pivots;
            for (int i = 0; i < 9; i++) out.pivots[i] = in.pivots[i];
            for (int i = 0; i < 8; i++) {
                out.mapping_idc[i] = in.mapping_idc[i]; out.poly_order[i] = in.poly_order[i]; out.mmr_order[i] = in.mmr_order[i];
                out.mmr_constant[i] = in.mmr_constant[i];
                for (int k = 0; k < 3; k++) out.poly_coef[i][k] = in.poly_coef[i][k];
                for (int o = 0; o < 3; o++)
                    for (int k = 0; k < 7; k++) out.mmr_coef[i][o][k] = in.mmr_coef[i][o][k];
            }
        }
        bool l1 = false, l3 = false;
        for (const auto &e : s.Extensions) {                                                      // CopySample :2347-2399
            if (e.level == 1 && !l1) { l1 = true; d.l1_present = 1; d.l1_min_pq = e.Level1.min_pq; d.l1_max_pq = e.Level1.max_pq; d.l1_avg_pq = e.Level1.avg_pq; }
            if (e.level == 3 && !l3) { l3 = true; d.l3_present = 1; d.l3_min_pq_offset = e.Level3.min_pq_offset; d.l3_max_pq_offset = e.Level3.max_pq_offset; d.l3_avg_pq_offset = e.Level3.avg_pq_offset; }
            if (e.level == 2 && d.n_l2 < 32) d.l2[d.n_l2++] = {e.Level2.target_max_pq, e.Level2.trim_slope, e.Level2.trim_offset, e.Level2.trim_power, e.Level2.trim_chroma_weight, e.Level2.trim_saturation_gain};
        }
        return d;
    }
};

// VideoRenderer.cpp:285-287, next to the existing two choices:
//     if (m_Sets.bUseHip)        m_VideoProcessor.reset(new CHipVideoProcessorAdapter(this, m_Sets, hr));
//     else if (m_Sets.bUseD3D11) m_VideoProcessor.reset(new CDX11VideoProcessor(this, m_Sets, hr));
CVideoProcessor *CreateHipVideoProcessor(CMpcVideoRenderer *pFilter, const Settings_t &config, HRESULT &hr)
{
    return new CHipVideoProcessorAdapter(pFilter, config, hr);        // (instantiating the class checks that no pure virtual is left)
}
