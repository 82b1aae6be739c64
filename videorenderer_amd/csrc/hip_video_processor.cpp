// hip_video_processor.cpp — pass sequencing and resource management of the shader video processor on
// HIP.  Restates the control flow of CDX11VideoProcessor::{InitMediaType, Configure, CopySample,
// Process, ConvertColorPass, ResizeShaderPass, FinalPass, GetCurentImage}
// (Source/DX11VideoProcessor.cpp) without the D3D11 plumbing.
#include "hip_video_processor.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>

#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstring>

namespace mpcvr {

static const uint16_t kDitherTable[1024] = {
#include "dither_table.inc"
};

// parameter blob exchanged between ranks (mpcvr_get/set_param_blob)
struct ParamBlob {
    uint32_t magic;          // 'MPVB'
    uint32_t version;
    float cm[12];
    float lum_scale;
    float gamut[9];
    int32_t tail;
    float gamma;
    Up2xWeights upx, upy;
    uint16_t dither[1024];
    float pq_lut[kPqLutSize];      // tone-map LUT (valid when tail == PQ->SDR)
};
static const uint32_t kBlobMagic = 0x4256504du;

// ------------------------------------------------------------------------------------------------
hipError_t DevBuffer::CheckCreate(size_t bytes)
{
    if (bytes <= size && ptr) return hipSuccess;
    Release();
    hipError_t e = hipMalloc(&ptr, bytes);
    if (e == hipSuccess) size = bytes; else ptr = nullptr;
    return e;
}
void DevBuffer::Release()
{
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr; size = 0;
}

CHipVideoProcessor::CHipVideoProcessor() { std::memcpy(m_ditherHost, kDitherTable, sizeof(m_ditherHost)); }

CHipVideoProcessor::~CHipVideoProcessor()
{
    // runs after a failed Init as well (the stream / events / dither buffer may exist already): every handle below is guarded
    if (!m_bInit && !m_stream && !m_evStart && !m_evStop && !m_dither.ptr) return;
    (void)hipSetDevice(m_device);
    if (m_stream) (void)hipStreamSynchronize(m_stream);
    for (DevBuffer *b : {&m_batchConv, &m_batchMid, &m_batchPost, &m_edPost, &m_edHandoff, &m_batchTex, &m_jincFirst, &m_jincSecond, &m_jincFused, &m_TexSrcVideo, &m_TexRaw, &m_TexPost, &m_TexConvertOutput, &m_TexResize, &m_BackBuffer, &m_Snapshot, &m_dither,
                         &m_pqLut, &m_hlgLut, &m_eotfLut, &m_stripTab, &m_tapsXi, &m_tapsXw, &m_tapsXs, &m_tapsYi, &m_tapsYw, &m_tapsYs, &m_otherX, &m_otherY, &m_tapsXb, &m_tapsYb})
        b->Release();
    for (UploadSlot &u : m_up) {
        u.dev.Release();
        if (u.pinned) (void)hipHostFree(u.pinned);
        if (u.uploaded) (void)hipEventDestroy(u.uploaded);
        if (u.consumed) (void)hipEventDestroy(u.consumed);
    }
    if (m_edStatus) (void)hipHostFree(m_edStatus);
    if (m_copyStream) (void)hipStreamDestroy(m_copyStream);
    m_doviDev.Release();
    m_bcast.Release();
    for (DoviSlot &d : m_doviSlots) {
        if (d.pinned) (void)hipHostFree(d.pinned);
        if (d.copied) (void)hipEventDestroy(d.copied);
    }
    for (DoviTableSlot &d : m_dvSlots) {
        d.dev.Release();
        if (d.pinned) (void)hipHostFree(d.pinned);
        if (d.done) (void)hipEventDestroy(d.done);
    }
    for (int i = 1; i < kLanes; i++) {
        Lane &l = m_lanes[i];
        l.conv.Release(); l.mid.Release(); l.post.Release();
        if (l.done) (void)hipEventDestroy(l.done);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    if (m_fork) (void)hipEventDestroy(m_fork);
    if (m_evStreamMark) (void)hipEventDestroy(m_evStreamMark);
    for (FrameLane &fl : m_flanes) {
        if (fl.stream) { (void)hipStreamSynchronize(fl.stream); (void)hipStreamDestroy(fl.stream); }
        for (LaneFrame &f : fl.ring) if (f.done) (void)hipEventDestroy(f.done);
        if (fl.batchDone) (void)hipEventDestroy(fl.batchDone);
    }
    for (FrameSlot &fs : m_slots) {
        fs.dev.Release();
        if (fs.pinned) (void)hipHostFree(fs.pinned);
        if (fs.done) (void)hipEventDestroy(fs.done);
    }
    if (m_evStart) (void)hipEventDestroy(m_evStart);
    if (m_evStop) (void)hipEventDestroy(m_evStop);
    for (hipEvent_t *e : {&m_evUp0, &m_evUp1, &m_evRb0, &m_evRb1}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    if (m_ownStream && m_stream) (void)hipStreamDestroy(m_stream);
}

// MPCVR_LOG=1: failures (and, =2, every plan the context settles on) go to stderr as well — the stand-in for the reference's
// DLog() lines (Utils/Util.h); mpcvr_last_error carries the same text to the caller either way.
static int LogLevel()
{
    static const int lvl = [] { const char *e = std::getenv("MPCVR_LOG"); return e && *e ? std::atoi(e) : 0; }();
    return lvl;
}

HRESULT CHipVideoProcessor::Fail(HRESULT hr, const std::string &msg)
{
    m_lastError = msg;
    if (LogLevel() >= 1) std::fprintf(stderr, "mpcvr[%p]: error 0x%08x: %s\n", (void *)this, (unsigned)hr, msg.c_str());
    return hr;
}

HRESULT CHipVideoProcessor::CheckHip(hipError_t e, const char *what)
{
    if (what[0] == 'k' && what[1] == '_') m_launches++;        // (kernel launches are checked under their kernel's name: GetLastBatchInfo counts them)
    if (e == hipSuccess) return MPCVR_S_OK;
    return Fail(e == hipErrorOutOfMemory ? MPCVR_E_OUTOFMEMORY : MPCVR_E_FAIL,
                std::string(what) + ": " + hipGetErrorString(e));
}

static bool ValidSettings(const mpcvr_settings &s, std::string *why)
{
    auto bad = [&](const char *m) { *why = m; return false; };
    if (s.iTexFormat != MPCVR_TEXFMT_AUTOINT && s.iTexFormat != MPCVR_TEXFMT_8INT &&
        s.iTexFormat != MPCVR_TEXFMT_10INT && s.iTexFormat != MPCVR_TEXFMT_16FLOAT) return bad("iTexFormat");
    if (s.iChromaScaling < 0 || s.iChromaScaling > MPCVR_CHROMA_CatmullRom) return bad("iChromaScaling");
    if (s.iUpscaling < 0 || s.iUpscaling > MPCVR_UPSCALE_Spline36_EXT) return bad("iUpscaling");
    if (s.iDownscaling < 0 || s.iDownscaling > MPCVR_DOWNSCALE_Lanczos) return bad("iDownscaling");
    if (s.iSDRDisplayNits < 25 || s.iSDRDisplayNits > 400) return bad("iSDRDisplayNits");   // IVideoRenderer.h:87-90
    if (s.output_format != MPCVR_OUT_BGRA8 && s.output_format != MPCVR_OUT_RGB10A2) return bad("output_format");
    if (s.bUseDither < 0 || s.bUseDither > MPCVR_DITHER_ErrorDiffusion_EXT) return bad("bUseDither");
    return true;
}

HRESULT CHipVideoProcessor::Init(int device, const mpcvr_settings &settings)
{
    std::string why;
    if (!ValidSettings(settings, &why)) return Fail(MPCVR_E_INVALIDARG, "invalid settings: " + why);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return Fail(MPCVR_E_FAIL, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device < 0 || device >= count) return Fail(MPCVR_E_INVALIDARG, "device ordinal out of range");
    m_device = device;
    HRESULT hr;
    if ((hr = CheckHip(hipSetDevice(device), "hipSetDevice"))) return hr;
    // the context's own stream is a BLOCKING stream: it orders itself against the legacy null stream, so a caller that
    // prepares samples / render targets on stream 0 (torch's default stream) and never hands over a stream is still ordered
    if ((hr = CheckHip(hipStreamCreateWithFlags(&m_stream, hipStreamDefault), "hipStreamCreate"))) return hr;
    m_ownStream = true;
    if ((hr = CheckHip(hipEventCreate(&m_evStart), "hipEventCreate"))) return hr;
    if ((hr = CheckHip(hipEventCreate(&m_evStop), "hipEventCreate"))) return hr;
    // dither texture load — DX11VideoProcessor.cpp:1414-1440
    if ((hr = CheckHip(m_dither.CheckCreate(sizeof(m_ditherHost)), "dither alloc"))) return hr;
    if ((hr = CheckHip(hipMemcpy(m_dither.ptr, m_ditherHost, sizeof(m_ditherHost), hipMemcpyHostToDevice), "dither upload"))) return hr;
    m_cfg = settings;
    m_bInit = true;
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::SetStream(hipStream_t s)
{
    if (!m_bInit) return Fail(MPCVR_E_NOT_VALID_STATE, "not initialised");
    (void)hipSetDevice(m_device);
    (void)JoinFrameLanes(true);
    if (m_stream) (void)hipStreamSynchronize(m_stream);
    if (m_ownStream && m_stream) (void)hipStreamDestroy(m_stream);
    m_ownStream = false;
    m_stream = s;
    if (!s) {
        // NULL (which is also the handle of the legacy default stream) = the context's own BLOCKING stream, see Init
        HRESULT hr = CheckHip(hipStreamCreateWithFlags(&m_stream, hipStreamDefault), "hipStreamCreate");
        if (hr) return hr;
        m_ownStream = true;
    }
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::Synchronize()
{
    if (!m_bInit) return Fail(MPCVR_E_NOT_VALID_STATE, "not initialised");
    (void)hipSetDevice(m_device);
    HRESULT hr = JoinFrameLanes(true);
    if (hr) return hr;
    if ((hr = CheckHip(hipStreamSynchronize(m_stream), "hipStreamSynchronize"))) return hr;
    if (m_edStatus && *m_edStatus) { *m_edStatus = 0; return Fail(MPCVR_E_FAIL, "error diffusion: a band gave up waiting for the band above"); }
    return MPCVR_S_OK;
}

// ---- frame lanes (see hip_video_processor.h) ----
// A frame may run beside its predecessor when nothing it touches is shared with it: the context owns its stream (a caller's stream
// promises stream order), the sample is read in place (no repack / copy into m_TexSrcVideo), no per-frame constants are uploaded on
// the context stream (Dolby Vision), and the plan has no intermediate surface (one fused kernel per frame: exact 2x, the strip /
// periodic kernel without the HDR10 tone-mapping step, the same-size block convert).
bool CHipVideoProcessor::FrameLanesUsable() const
{
    static const bool off = [] { const char *e = std::getenv("MPCVR_NO_FRAME_LANES"); return e && *e && *e != '0'; }();
    if (off || !m_ownStream || (m_cfg.flags & MPCVR_FLAG_NO_FRAME_LANES) || m_doviValid || !m_srcParams) return false;
    if (m_srcParams->cformat == MPCVR_CF_V210 || m_srcParams->layout == LAY_RGB || ((uintptr_t)m_curSample & 3) != 0) return false;
    if (m_plan.errdiff) return false;                 // (the error-diffusion pass reads the context's one intermediate)
    if (m_plan.fused_up2x) return true;
    if (m_strip && !m_plan.hdr_tonemap) return true;
    if (m_plan.direct_convert) return true;
    return false;
}

// Four lanes: measured on MI355X with 4K P010 -> 8K frames (bench.py process_per_frame) — one lane 14.4 k frames/s, two 18.2 k, four
// 19.3 k; the kernels size their segments for that many frames side by side (FusedParams::inflight).
int CHipVideoProcessor::FrameLaneCount()
{
    static const int n = [] { const char *e = std::getenv("MPCVR_FRAME_LANES"); const int v = e && *e ? std::atoi(e) : 4; return v < 1 ? 1 : v > kFrameLanes ? kFrameLanes : v; }();
    return n;
}

// the lane of the frame about to be queued: one that still holds a frame into the same render target if there is one (stream order
// then keeps the two writes apart; further lanes holding such a frame are waited for), else the next in turn
CHipVideoProcessor::FrameLane *CHipVideoProcessor::PickFrameLane(const void *rt)
{
    FrameLane *pick = nullptr;
    hipEvent_t also[kFrameLanes];
    int n_also = 0;
    for (int li = 0; li < FrameLaneCount(); li++) {
        FrameLane &fl = m_flanes[li];
        hipEvent_t latest = nullptr;                 // the lane's most recent unfinished frame into rt (the ring is walked oldest first)
        for (int i = 0; i < kLaneDepth; i++) {
            LaneFrame &f = fl.ring[(fl.head + i) % kLaneDepth];
            if (!f.pending || f.rt != rt) continue;          // (only a frame into the same target is worth a driver call)
            if (hipEventQuery(f.done) == hipSuccess) { f.pending = false; continue; }
            latest = f.done;
        }
        if (!latest) continue;
        if (!pick) pick = &fl; else also[n_also++] = latest;
    }
    if (!pick) { pick = &m_flanes[m_flaneNext]; m_flaneNext = (m_flaneNext + 1) % FrameLaneCount(); }
    if (!pick->stream && hipStreamCreateWithFlags(&pick->stream, hipStreamDefault) != hipSuccess) { pick->stream = nullptr; return nullptr; }
    for (int i = 0; i < n_also; i++) (void)hipStreamWaitEvent(pick->stream, also[i], 0);
    // ... and behind a whole batch still in flight on another lane that writes this target (the pick's own batches: stream order)
    for (int li = 0; li < kFrameLanes; li++) {
        FrameLane &bl = m_flanes[li];
        if (!bl.batchPending || &bl == pick) continue;
        if (hipEventQuery(bl.batchDone) == hipSuccess) { bl.batchPending = false; bl.batchRts.clear(); continue; }
        if (std::binary_search(bl.batchRts.begin(), bl.batchRts.end(), rt)) (void)hipStreamWaitEvent(pick->stream, bl.batchDone, 0);
    }
    return pick;
}

// ---- whole batches on the lanes (see FrameLane) ----
// The same rule as for single frames — nothing a batch touches may be shared with the batch beside it — checked on the route the batch will
// take: the exact-2x kernel, the strip / periodic kernel reading the samples themselves, the same-size block convert; default tier only.
bool CHipVideoProcessor::BatchLanesUsable(int n, const void *const *srcs, void *const *dsts, int rtPitch)
{
    static const bool off = [] { const char *e = std::getenv("MPCVR_NO_BATCH_LANES"); return e && *e && *e != '0'; }();
    if (off || !m_ownStream || n < 2 || !m_srcParams || m_doviValid || m_dvFrames) return false;
    if (m_cfg.flags & (MPCVR_FLAG_NO_FRAME_LANES | MPCVR_FLAG_NO_FUSED | MPCVR_FLAG_NO_FAST_CONVERT | MPCVR_FLAG_NO_STRIP)) return false;
    if (m_srcParams->cformat == MPCVR_CF_V210 || m_srcParams->layout == LAY_RGB) return false;
    if (m_plan.errdiff || m_plan.hdr_tonemap || rtPitch < m_windowRect.Width() * 4) return false;
    bool aligned = true;
    for (int i = 0; i < n; i++) {
        if (!srcs[i] || !dsts[i] || ((uintptr_t)srcs[i] & 3) != 0) return false;
        if (((uintptr_t)dsts[i] & 15) != 0) aligned = false;
    }
    // Two launches in flight were measured to pay on every such route (same box, bench.py process_batch_on_lanes against `value`, a quarter of
    // a second of batches each; profiles/r06/bench_batch_lanes_all_routes_call32.txt, bench_batch_lanes_call27.txt): same-size block convert
    // +11-16 %, exact-2x kernel +2-5 % (4K -> 8K, four rounds of waves per batch) to +15 % (1080p -> 4K, one round), strip / periodic kernel
    // +11-29 % (1080p -> 1440p 99.8 k -> 116.4 k frames/s, 720p -> 1080p 190 k -> 246 k), fused Jinc2m +4 %.  (An earlier table that had
    // the strip kernels LOSE was a wall clock around 30 launches of 0.3 ms: it measured the closing synchronize.)
    if (m_plan.fused_up2x) return true;
    if (m_strip) {
        FusedStripParams sp{};
        return FillStripParams((const uint8_t *)srcs[0], dsts[0], rtPitch, MakeStore(dsts[0], rtPitch, m_plan.swap_fmt, true), &sp) && !sp.surface_mode;
    }
    if (m_plan.direct_convert) {
        FusedParams conv{}, direct{};
        m_batchRepacked = false; m_batchSrc16 = false;           // (BatchPlan reads them; ProcessBatchRoutesOn sets them again)
        return BatchPlan((const uint8_t *)srcs[0], dsts[0], rtPitch, aligned, &conv, &direct);
    }
    return false;
}

// the lane of the batch about to be queued (the two take turns), ordered behind everything still in flight on OTHER lanes that writes one
// of its render targets: single frames (their ring entries) and batches
CHipVideoProcessor::FrameLane *CHipVideoProcessor::PickBatchLane(int n, void *const *dsts)
{
    FrameLane *pick = &m_flanes[m_blaneNext];
    if (!pick->stream && hipStreamCreateWithFlags(&pick->stream, hipStreamDefault) != hipSuccess) { pick->stream = nullptr; return nullptr; }
    // (two lanes: MPCVR_BATCH_LANE_COUNT = 2 .. 8 for the A/B — profiles/r06/batch_lane_count_call34.txt)
    static const int count = [] { const char *e = std::getenv("MPCVR_BATCH_LANE_COUNT"); const int v = e && *e ? std::atoi(e) : kBatchLanes; return v < 2 ? 2 : v > kFrameLanes ? kFrameLanes : v; }();
    m_blaneNext = (m_blaneNext + 1) % count;
    std::vector<const void *> rts(dsts, dsts + n);
    std::sort(rts.begin(), rts.end());
    for (FrameLane &fl : m_flanes) {
        if (&fl == pick || !fl.stream) continue;
        for (LaneFrame &f : fl.ring) {
            if (!f.pending) continue;
            if (hipEventQuery(f.done) == hipSuccess) { f.pending = false; continue; }
            if (std::binary_search(rts.begin(), rts.end(), f.rt)) (void)hipStreamWaitEvent(pick->stream, f.done, 0);
        }
        if (!fl.batchPending) continue;
        if (hipEventQuery(fl.batchDone) == hipSuccess) { fl.batchPending = false; fl.batchRts.clear(); continue; }
        bool shared = false;
        for (size_t a = 0, b = 0; a < rts.size() && b < fl.batchRts.size() && !shared;) {
            if (rts[a] == fl.batchRts[b]) shared = true;
            else if (rts[a] < fl.batchRts[b]) a++; else b++;
        }
        if (shared) (void)hipStreamWaitEvent(pick->stream, fl.batchDone, 0);
    }
    return pick;
}

// the batch just queued on `fl` writes dsts[0..n): its completion event, and its targets joined to those of the lane's batches still in flight
void CHipVideoProcessor::NoteLaneBatch(FrameLane *fl, int n, void *const *dsts)
{
    if (!fl->batchDone && hipEventCreateWithFlags(&fl->batchDone, hipEventDisableTiming) != hipSuccess) { fl->batchDone = nullptr; (void)hipStreamSynchronize(fl->stream); return; }
    if (fl->batchPending && hipEventQuery(fl->batchDone) == hipSuccess) fl->batchPending = false;
    if (!fl->batchPending) fl->batchRts.clear();
    fl->batchRts.insert(fl->batchRts.end(), dsts, dsts + n);
    std::sort(fl->batchRts.begin(), fl->batchRts.end());
    fl->batchRts.erase(std::unique(fl->batchRts.begin(), fl->batchRts.end()), fl->batchRts.end());
    (void)hipEventRecord(fl->batchDone, fl->stream);
    fl->batchPending = true;
    fl->last = fl->batchDone;
}

// the frame just queued on `fl` writes `rt`: its completion event takes the ring's oldest slot (whose frame must have completed)
void CHipVideoProcessor::NoteLaneFrame(FrameLane *fl, const void *rt)
{
    LaneFrame &f = fl->ring[fl->head];
    fl->head = (fl->head + 1) % kLaneDepth;
    if (!f.done && hipEventCreateWithFlags(&f.done, hipEventDisableTiming) != hipSuccess) { f.done = nullptr; (void)hipStreamSynchronize(fl->stream); return; }
    if (f.pending) (void)hipEventSynchronize(f.done);
    f.rt = rt; f.pending = true;
    (void)hipEventRecord(f.done, fl->stream);
    fl->last = f.done;
}

// the context stream -> lane edge (see m_streamGen): one event record per generation of context-stream work, one wait per lane
void CHipVideoProcessor::LaneWaitsForStream(FrameLane *fl)
{
    if (fl->seenGen == m_streamGen || !m_stream) return;
    if (m_markGen != m_streamGen) {
        if (!m_evStreamMark && hipEventCreateWithFlags(&m_evStreamMark, hipEventDisableTiming) != hipSuccess) m_evStreamMark = nullptr;
        if (!m_evStreamMark || hipEventRecord(m_evStreamMark, m_stream) != hipSuccess) {       // no event: the host waits instead
            (void)hipStreamSynchronize(m_stream);
            for (FrameLane &l : m_flanes) l.seenGen = m_streamGen;
            return;
        }
        m_markGen = m_streamGen;
    }
    (void)hipStreamWaitEvent(fl->stream, m_evStreamMark, 0);
    fl->seenGen = m_streamGen;
}

// host_wait: block until the lanes are idle; otherwise the context stream waits for them (work queued on it afterwards runs behind
// every frame in flight)
HRESULT CHipVideoProcessor::JoinFrameLanes(bool host_wait)
{
    HRESULT hr = MPCVR_S_OK;
    for (FrameLane &fl : m_flanes) {
        if (!fl.stream || !fl.last) continue;
        if (host_wait) {
            HRESULT h = CheckHip(hipStreamSynchronize(fl.stream), "frame lane sync");
            if (h) hr = h;
            for (LaneFrame &f : fl.ring) f.pending = false;
            fl.batchPending = false; fl.batchRts.clear();
            fl.last = nullptr;
        } else if (m_stream) (void)hipStreamWaitEvent(m_stream, fl.last, 0);
    }
    return hr;
}

// ------------------------------------------------------------------------------------------------
// InitMediaType — DX11VideoProcessor.cpp:1742-1959 (shader-path half: InitializeTexVP :2018-2047)
// ------------------------------------------------------------------------------------------------
HRESULT CHipVideoProcessor::InitMediaType(int cformat, int width, int height, int pitch, const CRect *srcRect, uint32_t extfmt)
{
    if (!m_bInit) return Fail(MPCVR_E_NOT_VALID_STATE, "not initialised");
    const FmtConvParams *f = GetFmtConvParams(cformat);
    if (!f) return Fail(MPCVR_E_NOTIMPL, "colour format not supported by this build");
    if (width <= 0 || height <= 0 || width > 16384 || height > 16384) return Fail(MPCVR_E_INVALIDARG, "bad frame size");
    if ((f->div_w == 2 && (width & 1)) || (f->div_h == 2 && (height & 1)))
        return Fail(MPCVR_E_INVALIDARG, "subsampled formats need even dimensions");
    const int defPitch = DefaultPitch(*f, width);
    if (pitch == 0) pitch = defPitch;
    // a bottom-up RGB DIB arrives with a negative pitch (BI_RGB && biHeight > 0 => m_srcPitch = -m_srcPitch, :1801-1803)
    bool bottomUp = false;
    if (pitch < 0) {
        if (f->layout != LAY_RGB) return Fail(MPCVR_E_INVALIDARG, "a negative pitch (bottom-up) is only defined for the RGB formats");
        bottomUp = true; pitch = -pitch;
    }
    if (pitch < width * f->Packsize) return Fail(MPCVR_E_INVALIDARG, "pitch smaller than a row");
    if (f->cformat == MPCVR_CF_V210 && (pitch < (width + 5) / 6 * 16 || (pitch & 3)))
        return Fail(MPCVR_E_INVALIDARG, "v210 pitch smaller than a row of 16-byte groups");
    if (f->bytes == 2 && (pitch & 1)) return Fail(MPCVR_E_INVALIDARG, "16-bit formats need an even pitch");
    if (f->bytes == 4 && (pitch & 3)) return Fail(MPCVR_E_INVALIDARG, "32-bit texels need a pitch that is a multiple of 4");
    CRect r = srcRect ? *srcRect : CRect();
    if (r.IsRectNull()) r = CRect(0, 0, width, height);                        // :1821-1823
    if (r.left < 0 || r.top < 0 || r.right > width || r.bottom > height || r.Width() <= 0 || r.Height() <= 0)
        return Fail(MPCVR_E_INVALIDARG, "source rect outside the frame");

    m_srcParams = f;
    m_srcWidth = width; m_srcHeight = height;
    m_srcPitch = pitch;
    m_srcBottomUp = bottomUp;
    m_srcLines = SourceLines(*f, height);
    m_srcRect = r;
    m_srcRectWidth = r.Width(); m_srcRectHeight = r.Height();
    m_decExFmt.value = extfmt;
    m_srcExFmt = SpecifyExtendedFormat(m_decExFmt, *f, m_srcRectWidth, m_srcRectHeight);   // :1827
    m_blobOverride = false;
    if (m_videoRect.IsRectNull()) m_videoRect = CRect(0, 0, m_srcRectWidth, m_srcRectHeight);
    if (m_windowRect.IsRectNull()) m_windowRect = CRect(0, 0, m_videoRect.right, m_videoRect.bottom);
    SetShaderConvertColorParams();
    SetShaderLuminanceParams();
    m_curSample = nullptr;
    m_planDirty = true;
    m_texSrcZeroed = m_batchTexZeroed = false;      // another format / size: the RGB48 remainder texels must be cleared again
    return MPCVR_S_OK;
}

void CHipVideoProcessor::SetShaderConvertColorParams()
{
    if (!m_srcParams || m_blobOverride) return;
    if (m_doviValid) DoviColorMatrix(m_doviMd, *m_srcParams, m_procAmp, m_cm);     // :817-834
    else ComputeColorMatrix(m_srcExFmt, *m_srcParams, m_procAmp, m_cm);
    ComputeGamut2020to709(m_gamut);
    SelectTail(m_srcExFmt, m_cfg.bConvertToSdr != 0, &m_tail, &m_gamma, m_hdrOutput, m_doviValid);
}

void CHipVideoProcessor::SetShaderLuminanceParams()
{
    if (m_blobOverride) return;
    m_lumScale = 10000.0f / m_cfg.iSDRDisplayNits;                              // :891
}

HRESULT CHipVideoProcessor::SetVideoRect(const CRect &r)
{
    if (r.Width() <= 0 || r.Height() <= 0) return Fail(MPCVR_E_INVALIDARG, "empty video rect");
    if (r != m_videoRect) { m_videoRect = r; m_planDirty = true; }
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::SetWindowRect(const CRect &r)
{
    if (r.Width() <= 0 || r.Height() <= 0) return Fail(MPCVR_E_INVALIDARG, "empty window rect");
    const CRect w(0, 0, r.Width(), r.Height());
    if (w != m_windowRect) { m_windowRect = w; m_planDirty = true; }
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::SetRotation(int value)
{
    if (value != 0 && value != 90 && value != 180 && value != 270) return Fail(MPCVR_E_INVALIDARG, "rotation must be 0, 90, 180 or 270");
    if (value != m_iRotation) { m_iRotation = value; m_planDirty = true; }
    return MPCVR_S_OK;
}

// HDR output: stands in for m_bHdrPassthroughSupport && (m_bHdrPassthrough || m_bHdrLocalToneMapping) — the display is in
// HDR10 mode, so HDR sources are not converted to SDR (convertType :2948-2950) — plus m_bHdrLocalToneMapping /
// m_iHdrLocalToneMappingType / m_iHdrDisplayMaxNits
HRESULT CHipVideoProcessor::SetHdrOutput(bool enable, int toneMapType, float displayMaxNits)
{
    if (toneMapType < 0 || toneMapType > 6) return Fail(MPCVR_E_INVALIDARG, "tone mapping type must be 0 (off) .. 6");
    m_hdrOutput = enable; m_hdrToneMapType = toneMapType; m_hdrDisplayMaxNits = displayMaxNits;
    UpdateHdrToneMapParams();
    m_blobOverride = false;
    SetShaderConvertColorParams();
    m_planDirty = true;
    if (m_doviValid) {          // the level-2 selection depends on the display peak (:2384)
        const mpcvr_dovi_metadata md = m_doviMd;
        return SetDoviMetadata(&md);
    }
    return MPCVR_S_OK;
}

// the HDR10 metadata Render() hands to SetHDR10ShaderParams (:2716-2727): mastering min/max luminance, MaxCLL, MaxFALL
HRESULT CHipVideoProcessor::SetHdrMetadata(float minMastering, float maxMastering, float maxCLL, float maxFALL)
{
    m_hdrMeta[0] = minMastering; m_hdrMeta[1] = maxMastering; m_hdrMeta[2] = maxCLL; m_hdrMeta[3] = maxFALL;
    m_hdrMetaValid = true;
    UpdateHdrToneMapParams();
    m_planDirty = true;
    return MPCVR_S_OK;
}

// SetHDR10ShaderParams — DX11VideoProcessor.cpp:907-917
void CHipVideoProcessor::UpdateHdrToneMapParams()
{
    HdrToneMapParams k{m_hdrMeta[0], m_hdrMeta[1], m_hdrMeta[2], m_hdrMeta[3], m_hdrDisplayMaxNits, m_hdrToneMapType};
    if (m_doviValid && m_doviL1Present) {       // Render :2716-2720: L1 min, max, max, avg; BT.2390 -> ST 2094-10
        k.min_mastering = (float)m_doviL1[0]; k.max_mastering = (float)m_doviL1[1];
        k.max_cll = (float)m_doviL1[1]; k.max_fall = (float)m_doviL1[2];
        if (k.selection == 5) k.selection = 6;
    }
    k.l2_enabled = (m_doviValid && m_doviL2Present) ? 1 : 0;      // m_pDoViDynamicConstants at b1 (:3362-3364)
    std::memcpy(k.l2k, m_doviL2Raw, sizeof(k.l2k));
    SanitiseHdr10Params(&k);
    m_hdrTm = k;
}

// m_pPSHDR10ToneMapping exists once an HDR10/HLG source is shown in HDR with local tone mapping on and metadata known
bool CHipVideoProcessor::ToneMapActive() const
{
    const unsigned tf = m_srcExFmt.VideoTransferFunction();
    if (m_doviValid) return m_hdrOutput && m_hdrToneMapType > 0 && (m_hdrMetaValid || m_doviL1Present);     // SourceIsHDR()
    return m_hdrOutput && m_hdrToneMapType > 0 && m_hdrMetaValid && (tf == 15 || tf == 16);
}

// CopySample, IID_MediaSideDataDOVIMetadataV2 branch — DX11VideoProcessor.cpp:2270-2520
HRESULT CHipVideoProcessor::SetDoviMetadata(const mpcvr_dovi_metadata *md)
{
    const HRESULT hr = ApplyDoviMetadata(md);
    return (hr || !md) ? hr : UploadDoviParams();
}

// the host side of an RPU: curves, matrices, trims, what the plan depends on — everything but the copy to the device
HRESULT CHipVideoProcessor::ApplyDoviMetadata(const mpcvr_dovi_metadata *md)
{
    if (!m_bInit) return Fail(MPCVR_E_NOT_VALID_STATE, "not initialised");
    if (!md) {
        if (m_doviValid) { m_doviValid = false; m_planDirty = true; }
        m_doviL1Present = m_doviL2Present = false;
        m_blobOverride = false;
        SetShaderConvertColorParams();
        UpdateHdrToneMapParams();
        return MPCVR_S_OK;
    }
    if (!CheckDoviCurves(*md)) return Fail(MPCVR_E_INVALIDARG, "Dolby Vision curves: num_pivots outside [2,9], mapping_idc > 1 or more than 32 level-2 blocks");
    const bool wasValid = m_doviValid, hadToneMap = m_srcParams && ToneMapActive();
    m_doviMd = *md;
    m_doviValid = true;
    uint32_t l1[3];
    if (DoviL1Nits(*md, l1)) { m_doviL1Present = true; std::memcpy(m_doviL1, l1, sizeof(l1)); }
    float k[5];
    if (DoviL2Constants(*md, (int)m_hdrDisplayMaxNits, k)) { m_doviL2Present = true; std::memcpy(m_doviL2Raw, k, sizeof(k)); }
    else if (!m_doviL2Present) std::memcpy(m_doviL2Raw, k, sizeof(k));       // the cbuffer of an absent L2 (:956-960)
    PackDoviCurves(*md, &m_doviHost);
    DoviLmsMatrix(*md, m_doviHost.lms);
    m_doviHost.l2_enabled = m_doviL2Present ? 1 : 0;
    std::memcpy(m_doviHost.l2k, m_doviL2Raw, sizeof(m_doviHost.l2k));
    m_blobOverride = false;
    SetShaderConvertColorParams();
    UpdateHdrToneMapParams();
    if (!wasValid || (m_srcParams && hadToneMap != ToneMapActive())) m_planDirty = true;
    return MPCVR_S_OK;
}

// the curve / trim constant buffers travel through a small pinned ring so per-frame RPUs never stall the stream
HRESULT CHipVideoProcessor::UploadDoviParams()
{
    (void)hipSetDevice(m_device);
    HRESULT hr;
    if ((hr = CheckHip(m_doviDev.CheckCreate(sizeof(DoviParams)), "dovi constants"))) return hr;
    if (!m_eotfLut.ptr) {           // the PQ EOTF table of the block convert's Dolby Vision variants: a constant of the transfer function
        std::vector<float> lut(kPqEncOffset + kPqEncSize + 1, 0.0f);      // the EOTF table, then (16-byte aligned) the PQ encode table of the level-2 variant
        BuildPqEotfLut(lut.data());
        BuildPqEncodeLut(lut.data() + kPqEncOffset);
        if ((hr = CheckHip(m_eotfLut.CheckCreate(lut.size() * sizeof(float)), "pq eotf lut"))) return hr;
        if ((hr = CheckHip(hipMemcpy(m_eotfLut.ptr, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice), "pq eotf lut upload"))) return hr;
    }
    DoviSlot &slot = m_doviSlots[m_doviSlotNext++ % 4];
    if (!slot.pinned) {
        if ((hr = CheckHip(hipHostMalloc((void **)&slot.pinned, sizeof(DoviParams), hipHostMallocDefault), "dovi staging"))) return hr;
        if ((hr = CheckHip(hipEventCreateWithFlags(&slot.copied, hipEventDisableTiming), "dovi event"))) return hr;
    } else if ((hr = CheckHip(hipEventSynchronize(slot.copied), "dovi staging wait"))) return hr;
    *slot.pinned = m_doviHost;
    if ((hr = CheckHip(hipMemcpyAsync(m_doviDev.ptr, slot.pinned, sizeof(DoviParams), hipMemcpyHostToDevice, m_stream), "dovi upload"))) return hr;
    return CheckHip(hipEventRecord(slot.copied, m_stream), "dovi event record");
}

HRESULT CHipVideoProcessor::SetSampleFormat(int frameFormat)
{
    if (frameFormat < 0 || frameFormat > 2) return Fail(MPCVR_E_INVALIDARG, "frame format must be 0 (progressive), 1 (TFF) or 2 (BFF)");
    if (frameFormat != m_SampleFormat) { m_SampleFormat = frameFormat; m_planDirty = true; }
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::SetFlip(bool value)
{
    if (value != m_bFlip) { m_bFlip = value; m_planDirty = true; }
    return MPCVR_S_OK;
}

// Configure — DX11VideoProcessor.cpp:3800-4050: diff, then rebuild only what changed
HRESULT CHipVideoProcessor::Configure(const mpcvr_settings &c)
{
    if (!m_bInit) return Fail(MPCVR_E_NOT_VALID_STATE, "not initialised");
    std::string why;
    if (!ValidSettings(c, &why)) return Fail(MPCVR_E_INVALIDARG, "invalid settings: " + why);
    bool changeConvertShader = false, changeLuminance = false, changePlan = false;
    if (c.iTexFormat != m_cfg.iTexFormat) changePlan = true;
    if (c.iChromaScaling != m_cfg.iChromaScaling) changeConvertShader = true;
    if (c.iUpscaling != m_cfg.iUpscaling || c.iDownscaling != m_cfg.iDownscaling ||
        c.bInterpolateAt50pct != m_cfg.bInterpolateAt50pct) changePlan = true;
    if (c.bUseDither != m_cfg.bUseDither || c.output_format != m_cfg.output_format || c.flags != m_cfg.flags) changePlan = true;
    if (c.bConvertToSdr != m_cfg.bConvertToSdr) changeConvertShader = true;
    if (c.bDeintBlend != m_cfg.bDeintBlend) changePlan = true;
    if (c.iSDRDisplayNits != m_cfg.iSDRDisplayNits) changeLuminance = true;
    m_cfg = c;
    if (changeConvertShader || changeLuminance) m_blobOverride = false;
    if (changeConvertShader) { SetShaderConvertColorParams(); changePlan = true; }
    if (changeLuminance) { SetShaderLuminanceParams(); changePlan = true; }
    if (changePlan) m_planDirty = true;
    return (changeConvertShader || changeLuminance || changePlan) ? MPCVR_S_OK : MPCVR_S_FALSE;
}

// SetProcAmpValues — DX11VideoProcessor.cpp:4506-4537 (clamped to the ranges of Helper.cpp:182-187)
HRESULT CHipVideoProcessor::SetProcAmpValues(uint32_t flags, float b, float c, float h, float s)
{
    auto clampf = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };
    if (flags & MPCVR_PROCAMP_BRIGHTNESS) m_procAmp.brightness = clampf(b, -100.f, 100.f);
    if (flags & MPCVR_PROCAMP_CONTRAST) m_procAmp.contrast = clampf(c, 0.f, 2.f);
    if (flags & MPCVR_PROCAMP_HUE) m_procAmp.hue = clampf(h, -180.f, 180.f);
    if (flags & MPCVR_PROCAMP_SATURATION) m_procAmp.saturation = clampf(s, 0.f, 2.f);
    m_blobOverride = false;
    SetShaderConvertColorParams();
    m_planDirty = true;
    return MPCVR_S_OK;
}

static size_t SurfBytesPerPixel(int fmt) { return fmt == SF_RGBA16F ? 8 : 4; }
static int RgbTexFmt(const FmtConvParams &f);

HRESULT CHipVideoProcessor::UploadTaps(const HostAxisTaps &h, DevBuffer &bi, DevBuffer &bw, DevBuffer &bs, DevBuffer &bb,
                                       const std::vector<int32_t> &other, AxisTaps *out)
{
    HRESULT hr;
    if ((hr = CheckHip(bi.CheckCreate(h.idx.size() * sizeof(int32_t)), "taps alloc"))) return hr;
    if ((hr = CheckHip(bw.CheckCreate(h.w.size() * sizeof(float)), "taps alloc"))) return hr;
    if ((hr = CheckHip(hipMemcpy(bi.ptr, h.idx.data(), h.idx.size() * sizeof(int32_t), hipMemcpyHostToDevice), "taps upload"))) return hr;
    if ((hr = CheckHip(hipMemcpy(bw.ptr, h.w.data(), h.w.size() * sizeof(float), hipMemcpyHostToDevice), "taps upload"))) return hr;
    out->idx = (const int32_t *)bi.ptr; out->w = (const float *)bw.ptr; out->wsum = nullptr;
    if (h.normalise) {
        if ((hr = CheckHip(bs.CheckCreate(h.wsum.size() * sizeof(float)), "taps alloc"))) return hr;
        if ((hr = CheckHip(hipMemcpy(bs.ptr, h.wsum.data(), h.wsum.size() * sizeof(float), hipMemcpyHostToDevice), "taps upload"))) return hr;
        out->wsum = (const float *)bs.ptr;
    }
    out->ntaps = h.ntaps; out->normalise = h.normalise;
    // hints for the folded resize kernels: the source window of every block of 64 outputs, and whether the unfiltered
    // coordinate maps 1:1
    out->blk_lo = nullptr; out->blk_span = 0; out->idx_t = nullptr; out->w_t = nullptr; out->n_out = 0;
    out->blk8_lo = nullptr; out->blk8_span = 0; out->blk32_lo = nullptr; out->blk32_span = 0;
    const size_t nOut = h.ntaps > 0 ? h.idx.size() / (size_t)h.ntaps : 0;
    if (nOut > 0) {
        std::vector<int32_t> lo((nOut + 63) / 64);
        int span = 0;
        for (size_t b = 0; b < lo.size(); b++) {
            const size_t first = b * 64 * (size_t)h.ntaps, last = std::min(nOut, (b + 1) * 64) * (size_t)h.ntaps;
            const auto mm = std::minmax_element(h.idx.begin() + first, h.idx.begin() + last);
            lo[b] = *mm.first;
            span = std::max(span, *mm.second - *mm.first + 1);
        }
        std::vector<int32_t> lo8((nOut + 7) / 8);
        int span8 = 0;
        for (size_t b = 0; b < lo8.size(); b++) {
            const size_t first = b * 8 * (size_t)h.ntaps, last = std::min(nOut, (b + 1) * 8) * (size_t)h.ntaps;
            const auto mm = std::minmax_element(h.idx.begin() + first, h.idx.begin() + last);
            lo8[b] = *mm.first;
            span8 = std::max(span8, *mm.second - *mm.first + 1);
        }
        std::vector<int32_t> lo32((nOut + 31) / 32);
        int span32 = 0;
        for (size_t b = 0; b < lo32.size(); b++) {
            const size_t first = b * 32 * (size_t)h.ntaps, last = std::min(nOut, (b + 1) * 32) * (size_t)h.ntaps;
            const auto mm = std::minmax_element(h.idx.begin() + first, h.idx.begin() + last);
            lo32[b] = *mm.first;
            span32 = std::max(span32, *mm.second - *mm.first + 1);
        }
        // tap-major copies of both tables and the 8- / 32-output block tables behind the block table, in the same buffer
        const size_t off = (lo.size() + 63) / 64 * 64, cnt = h.idx.size();
        std::vector<int32_t> pack(off + 2 * cnt + lo8.size() + lo32.size());
        std::copy(lo8.begin(), lo8.end(), pack.begin() + off + 2 * cnt);
        std::copy(lo32.begin(), lo32.end(), pack.begin() + off + 2 * cnt + lo8.size());
        std::copy(lo.begin(), lo.end(), pack.begin());
        for (size_t f = 0; f < nOut; f++)
            for (int k = 0; k < h.ntaps; k++) {
                pack[off + (size_t)k * nOut + f] = h.idx[f * h.ntaps + k];
                std::memcpy(&pack[off + cnt + (size_t)k * nOut + f], &h.w[f * h.ntaps + k], sizeof(float));
            }
        if ((hr = CheckHip(bb.CheckCreate(pack.size() * sizeof(int32_t)), "taps alloc"))) return hr;
        if ((hr = CheckHip(hipMemcpy(bb.ptr, pack.data(), pack.size() * sizeof(int32_t), hipMemcpyHostToDevice), "taps upload"))) return hr;
        out->blk_lo = (const int32_t *)bb.ptr; out->blk_span = span;
        out->idx_t = out->blk_lo + off; out->w_t = (const float *)(out->idx_t + cnt); out->n_out = (int)nOut;
        out->blk8_lo = out->idx_t + 2 * cnt; out->blk8_span = span8;
        out->blk32_lo = out->blk8_lo + lo8.size(); out->blk32_span = span32;
    }
    out->other_identity = 1;
    for (size_t i = 0; i < other.size(); i++)
        if (other[i] != (int32_t)i) { out->other_identity = 0; break; }
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::UploadIndex(const std::vector<int32_t> &v, DevBuffer &b)
{
    HRESULT hr;
    if ((hr = CheckHip(b.CheckCreate(v.size() * sizeof(int32_t)), "index alloc"))) return hr;
    return CheckHip(hipMemcpy(b.ptr, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice), "index upload");
}

// UpdateTexures (:2869-2892) + UpdatePostScaleTexures (:2894-2912) + the per-axis shader choice of
// ResizeShaderPass (:3103-3133), evaluated once per geometry/settings change instead of per frame.
HRESULT CHipVideoProcessor::UpdatePlan()
{
    if (!m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    (void)hipSetDevice(m_device);
    (void)JoinFrameLanes(true);
    (void)hipStreamSynchronize(m_stream);    // resources below may still be in use
    const int w1 = m_srcRectWidth, h1 = m_srcRectHeight;
    const int w2 = m_videoRect.Width(), h2 = m_videoRect.Height();
    {
        const PlanGeometry g{w1, h1, m_videoRect.left, m_videoRect.top, m_videoRect.right, m_videoRect.bottom,
                             m_windowRect.Width(), m_windowRect.Height(), m_iRotation, m_bFlip ? 1 : 0,
                             ConvertEnabled() ? 1 : 0, ToneMapActive() ? 1 : 0, m_doviValid ? 1 : 0};
        std::string why;
        if (!DecidePlan(m_cfg.iTexFormat, m_cfg.iChromaScaling, m_cfg.iUpscaling, m_cfg.iDownscaling,
                        m_cfg.bInterpolateAt50pct, m_cfg.bUseDither, m_cfg.output_format,
                        m_cfg.flags,
                        *m_srcParams, g, &m_plan, &why))
            return Fail(MPCVR_E_NOTIMPL, why);
    }

    HRESULT hr;
    if (m_plan.hdr_tonemap &&
        (hr = CheckHip(m_TexPost.CheckCreate((size_t)w2 * SurfBytesPerPixel(m_plan.internal_fmt) * h2), "m_TexsPostScale"))) return hr;
    m_postBytes = m_plan.hdr_tonemap ? (size_t)w2 * SurfBytesPerPixel(m_plan.internal_fmt) * h2 : 0;
    m_midBytes = 0;
    // m_TexConvertOutput: srcRect-sized, internal format (:2889-2890)
    const size_t convPitch = (size_t)w1 * SurfBytesPerPixel(m_plan.internal_fmt);
    if ((hr = CheckHip(m_TexConvertOutput.CheckCreate(convPitch * h1), "m_TexConvertOutput"))) return hr;
    m_convBytes = convPitch * h1;

    HostAxisTaps hx, hy;
    std::vector<int32_t> ox, oy;
    if (m_plan.two_pass || m_plan.one_pass) {
        // The rotation-carrying draw (TextureResizeShader / TextureCopyRect with FillVertices' rotation and flip,
        // :130-179): which texture coordinate runs along which screen axis, and in which direction
        //     rot   0: U = l + a(r-l)  V = t + b(bm-t)      rot  90: U = l + b(r-l)  V = bm - a(bm-t)
        //     rot 180: U = r - a(r-l)  V = bm - b(bm-t)     rot 270: U = r - b(r-l)  V = t + a(bm-t)     flip: l <-> r
        const int rot = m_plan.rotation;
        const bool swap = rot == 90 || rot == 270;
        const int tax = swap ? 1 : 0;                               // texture axis run through by screen x
        bool rev_u = rot == 180 || rot == 270;
        const bool rev_v = rot == 90 || rot == 180;
        if (m_plan.flip) rev_u = !rev_u;
        const bool rev_x = tax == 0 ? rev_u : rev_v, rev_y = tax == 0 ? rev_v : rev_u;
        const int len_x = tax == 0 ? w1 : h1, len_y = tax == 0 ? h1 : w1;       // extent of the source rect along x / y
        // source of the draw: the convert output (rect at the origin) or, with the convert draw disabled, the source
        // texture itself with rSrc = srcRect (:3321-3323); clamp addressing covers the whole texture
        const bool fromTex = !m_plan.convert;
        const int ol = fromTex ? m_srcRect.left : 0, ot = fromTex ? m_srcRect.top : 0;
        const int tw = fromTex ? m_srcWidth : w1, th = fromTex ? m_srcHeight : h1;
        const int org_x = tax == 0 ? ol : ot, org_y = tax == 0 ? ot : ol;
        const int tex_x = tax == 0 ? tw : th, tex_y = tax == 0 ? th : tw;
        const int outW = w2, outH = m_plan.two_pass ? m_plan.mid_h : h2;
        const int a = m_plan.first_tex_axis;
        // scale[AXIS] as TextureResizeShader sets it: srcRect/dstRect of the same-named screen dimension (:351-354)
        const float cscale = a == 0 ? (float)w1 / (float)outW : (float)h1 / (float)outH;
        const bool taps_on_x = (a < 0) || (tax == a);               // ps_simple: a 1-tap table along x
        const Resizer rs = a < 0 ? Resizer{RS_NONE, 0} : m_plan.first_rs;
        m_firstJinc = rs.kind == RS_UP && rs.method == MPCVR_UPSCALE_Jinc2;
        m_firstCoords = DrawCoords{org_x, len_x, rev_x ? 1 : 0, (float)len_x / (float)outW,
                                   org_y, len_y, rev_y ? 1 : 0, (float)len_y / (float)outH, swap ? 1 : 0, tex_x, tex_y, outW, outH};
        bool ok = true;
        if (m_firstJinc) {
            // the 2-D shader needs no tables
        } else
        if (taps_on_x) {
            ok = BuildAxisTaps(rs, org_x, len_x, outW, tex_x, m_cfg.flags, &hx, rev_x, a < 0 ? 0.0f : cscale);
            BuildPointIndex(org_y, len_y, outH, tex_y, &ox, rev_y);
        } else {
            ok = BuildAxisTaps(rs, org_y, len_y, outH, tex_y, m_cfg.flags, &hx, rev_y, cscale);
            BuildPointIndex(org_x, len_x, outW, tex_x, &ox, rev_x);
        }
        if (!ok) return Fail(MPCVR_E_NOTIMPL, "resize ratio outside the supported range");
        m_firstAxis = taps_on_x ? 0 : 1;
        m_firstSwap = swap;
        if (!m_firstJinc) {
            if ((hr = UploadTaps(hx, m_tapsXi, m_tapsXw, m_tapsXs, m_tapsXb, ox, &m_tapsX))) return hr;
            if ((hr = UploadIndex(ox, m_otherX))) return hr;
        }
        m_jincFirstTab = nullptr; m_jincFirstCtr = nullptr;
        if (m_firstJinc && (hr = UploadJincPhases(m_firstCoords, m_jincFirst, &m_jincFirstTab, &m_jincFirstCtr))) return hr;
    }
    if (m_plan.two_pass) {
        // m_TexResize: fp16, dst width x (source extent along screen y) (:3143-3160); the second draw is unrotated
        const int mh = m_plan.mid_h;
        if ((hr = CheckHip(m_TexResize.CheckCreate((size_t)w2 * 8 * mh), "m_TexResize"))) return hr;
        m_midBytes = (size_t)w2 * 8 * mh;
        m_secondJinc = m_plan.ry.kind == RS_UP && m_plan.ry.method == MPCVR_UPSCALE_Jinc2;
        m_secondCoords = DrawCoords{0, w2, 0, 1.0f, 0, mh, 0, (float)mh / (float)h2, 0, w2, mh, w2, h2};
        m_jincSecondTab = nullptr; m_jincSecondCtr = nullptr;
        if (m_secondJinc && (hr = UploadJincPhases(m_secondCoords, m_jincSecond, &m_jincSecondTab, &m_jincSecondCtr))) return hr;
        if (!m_secondJinc) {
            if (!BuildAxisTaps(m_plan.ry, 0, mh, h2, mh, m_cfg.flags, &hy))
                return Fail(MPCVR_E_NOTIMPL, "resize ratio outside the supported range");
            BuildPointIndex(0, w2, w2, w2, &oy);     // Y pass: columns map 1:1
            if ((hr = UploadTaps(hy, m_tapsYi, m_tapsYw, m_tapsYs, m_tapsYb, oy, &m_tapsY))) return hr;
            if ((hr = UploadIndex(oy, m_otherY))) return hr;
        }
    }

    // the arbitrary-ratio fused kernel takes an unrotated two-pass resize whose tables fit it: straight from the raw sample for
    // 4:2:0 sources (m_strip, decided below), else from the convert kernel's output / the RGB source texture (m_stripSurf).
    // A horizontal flip (FillVertices swaps src_l and src_r, DX11VideoProcessor.cpp:167-169) is the X draw's table read from the other end — per-column tap
    // indices and weights are what these kernels read anyway — so a flipped frame stays on the fused path.  Rotation 180 also
    // reverses the first draw's ROW map (m_otherX), which the surface variant reads row by row: a frame turned upside down goes
    // convert kernel -> m_TexConvertOutput -> k_fused_strip:surface.  90 / 270 turn the first draw into a Y shader and stay per draw
    m_strip = m_stripSurf = m_stripPlanned = false;
    m_stripRan = -1;
    m_periodPlan.P = 0;
    static const bool no_strip_env = [] { const char *e = std::getenv("MPCVR_NO_STRIP"); return e && *e && *e != '0'; }();
    if (!no_strip_env && m_plan.two_pass && !m_firstJinc && !m_secondJinc && m_firstAxis == 0 && !m_firstSwap && (m_plan.rotation == 0 || m_plan.rotation == 180) &&
        !(m_cfg.flags & (MPCVR_FLAG_NO_FUSED | MPCVR_FLAG_NO_FAST_CONVERT | MPCVR_FLAG_NO_STRIP)) &&
        PlanFusedStrip(hx, hy, w2, h2, m_plan.convert ? w1 : m_srcWidth, m_plan.mid_h, &m_stripPlan)) {
        // one buffer: yrange | xstrip | xi_t | xw_t | yi | yw  (all 4-byte words)
        const StripPlan &sp = m_stripPlan;
        std::vector<int32_t> pack;
        auto put = [&pack](const void *src, size_t words) {
            const size_t at = pack.size();
            pack.resize(at + words);
            std::memcpy(pack.data() + at, src, words * 4);
            return at;
        };
        m_stripOff[0] = put(sp.yrange.data(), sp.yrange.size());
        m_stripOff[1] = put(sp.xstrip.data(), sp.xstrip.size());
        m_stripOff[2] = put(sp.xi_t.data(), sp.xi_t.size());
        m_stripOff[3] = put(sp.xw_t.data(), sp.xw_t.size());
        m_stripOff[4] = put(sp.yi.data(), sp.yi.size());
        m_stripOff[5] = put(sp.yw.data(), sp.yw.size());
        // periodic vertical ratio (1080p -> 1440p, 720p -> 1080p, 4K -> 1440p, 4K -> 1080p ...): the register-window kernel's tables
        m_periodPlan.P = 0;
        const bool q1 = m_plan.rx.kind == RS_UP && m_plan.ry.kind == RS_UP && m_cfg.iUpscaling == MPCVR_UPSCALE_Lanczos3 && !(m_cfg.flags & MPCVR_FLAG_LANCZOS3_FIXED);
        // (an interleaved RGB sample without a convert draw is read in place: the X tables then index the whole texture's columns)
        if (m_plan.rx.kind == RS_UP && m_plan.ry.kind == RS_UP &&
            PlanFusedPeriod(hx, hy, w2, h2, m_plan.convert ? w1 : m_srcWidth, m_plan.mid_h, q1, &m_periodPlan, m_tail == TAIL_PQ_TO_SDR || m_tail == TAIL_HLG_TO_SDR)) {
            const PeriodPlan &pp = m_periodPlan;
            m_periodOff[0] = put(pp.xi_t.data(), pp.xi_t.size());
            m_periodOff[1] = put(pp.xw_t.data(), pp.xw_t.size());
            while (pack.size() & 7) pack.push_back(0);                      // the weight rows (32 bytes each) are read with scalar multi-dword loads
            m_periodOff[2] = put(pp.yw.data(), pp.yw.size());
            m_periodOff[3] = put(pp.xstrip.data(), pp.xstrip.size());
        }
        if ((hr = CheckHip(m_stripTab.CheckCreate(pack.size() * sizeof(int32_t)), "strip tables"))) return hr;
        if ((hr = CheckHip(hipMemcpy(m_stripTab.ptr, pack.data(), pack.size() * sizeof(int32_t), hipMemcpyHostToDevice), "strip tables upload"))) return hr;
        m_stripPlanned = true;
        m_strip = m_plan.convert && !m_doviValid && m_plan.internal_fmt != SF_RGBA16F && m_plan.rotation == 0;
    }

    // PQ -> SDR table: the fused kernel's tone-map stage and the folded convert kernel's
    if (m_tail == TAIL_PQ_TO_SDR) {
        if (!m_blobOverride) BuildPqSdrLut(m_lumScale, m_pqLutHost);
        if ((hr = CheckHip(m_pqLut.CheckCreate(sizeof(m_pqLutHost)), "pq lut"))) return hr;
        if ((hr = CheckHip(hipMemcpy(m_pqLut.ptr, m_pqLutHost, sizeof(m_pqLutHost), hipMemcpyHostToDevice), "pq lut upload"))) return hr;
        m_pqLutValid = true;
    } else {
        m_pqLutValid = false;
    }
    if (m_tail == TAIL_HLG_TO_SDR && !m_hlgLut.ptr) {          // constants only: built once per context
        std::vector<float> t(kPqLutSize);
        BuildHlgInverseLut(t.data());
        if ((hr = CheckHip(m_hlgLut.CheckCreate(t.size() * sizeof(float)), "hlg lut"))) return hr;
        if ((hr = CheckHip(hipMemcpy(m_hlgLut.ptr, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice), "hlg lut upload"))) return hr;
    }
    m_jincFusedTab = nullptr;
    if (m_plan.fused_jinc) {
        // the fused Jinc2m kernel's weights: the phase table of a 2x draw (integer origins drop out of it) in the kernel's reading order
        // (the draw the kernel replaces: the whole m_TexConvertOutput, src-rect sized, onto the video rect; BuildJincPhases checks every
        // output index's tap base against the shader's own texture coordinate)
        DrawCoords dc{};
        dc.step_x = dc.step_y = 0.5f;
        dc.len_x = dc.tex_x = m_srcRectWidth; dc.len_y = dc.tex_y = m_srcRectHeight;
        dc.n_x = m_videoRect.Width(); dc.n_y = m_videoRect.Height();
        std::vector<unsigned char> phases(JincPhasesBytes());
        std::vector<float> tab(FusedJincTableBytes() / sizeof(float));
        if (!BuildJincPhases(dc, phases.data())) m_plan.fused_up2x = m_plan.fused_jinc = false;
        else {
#ifdef MPCVR_DEBUG_HOOKS        // (debug builds only — tools/debug/jinc_diff.py: a one-tap filter, which texel does an output pixel read?)
            if (const char *e = std::getenv("MPCVR_JINC_DBG")) {
                JincPhases &jp = *(JincPhases *)phases.data();
                const int tap = std::atoi(e);
                for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) { for (int k = 0; k < 16; k++) jp.w[a][b][k] = k == tap ? 1.0f : 0.0f; jp.wsum[a][b] = 1.0f; }
            }
#endif
            BuildFusedJincTable(phases.data(), tab.data());
            if ((hr = CheckHip(m_jincFused.CheckCreate(tab.size() * sizeof(float)), "fused jinc table"))) return hr;
            if ((hr = CheckHip(hipMemcpy(m_jincFused.ptr, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice), "fused jinc table upload"))) return hr;
            m_jincFusedTab = (const float *)m_jincFused.ptr;
        }
    }
    if (m_plan.fused_up2x) {
        if (!m_plan.fused_jinc && (!m_blobOverride || m_upX.ntaps == 0)) {
            float w[6];
            const int n = UpscaleWeights(m_cfg.iUpscaling, 0.75f, w);
            m_upX.ntaps = n; std::memset(m_upX.w_even, 0, sizeof(m_upX.w_even)); std::memset(m_upX.w_odd, 0, sizeof(m_upX.w_odd));
            std::memcpy(m_upX.w_even, w, sizeof(float) * n);
            UpscaleWeights(m_cfg.iUpscaling, 0.25f, w);
            std::memcpy(m_upX.w_odd, w, sizeof(float) * n);
            m_upX.q1_quirk = (m_cfg.iUpscaling == MPCVR_UPSCALE_Lanczos3 && !(m_cfg.flags & MPCVR_FLAG_LANCZOS3_FIXED)) ? 1 : 0;
            m_upY = m_upX;
        }
        FusedParams fp{};
        FillFusedParams(nullptr, nullptr, 0, &fp);
        m_plan.fused_up2x = FusedUp2xSupported(fp);
        // the fused Jinc2m kernel wants 114 - 146 KiB of LDS per workgroup: where the device grants less, the convert + k_jinc2 draws (advisor, round 5)
        if (m_plan.fused_jinc && FusedJincLdsBytes(fp) > DeviceLdsLimit()) m_plan.fused_up2x = false;
        if (!m_plan.fused_up2x) m_plan.fused_jinc = false;
        // experiment knob: exact 2x through the arbitrary-ratio kernel instead (DESIGN.md §4.3 compares the two)
        static const bool no_up2x_env = [] { const char *e = std::getenv("MPCVR_NO_UP2X"); return e && *e && *e != '0'; }();
        if (no_up2x_env && m_strip) m_plan.fused_up2x = false;
    }
    m_period = false;
    // the store the resize draws will meet: the render target, or m_TexsPostScale in front of the HDR10 tone-mapping step (:3359-3367)
    const StoreParams probeStore = m_plan.hdr_tonemap ? MakeStore(nullptr, (int)(w2 * SurfBytesPerPixel(m_plan.internal_fmt)), m_plan.internal_fmt, false)
                                                      : MakeStore(nullptr, m_windowRect.Width() * 4, m_plan.swap_fmt, true);
    if (m_strip) {      // the launch-time conditions that do not depend on the frame pointers
        FusedStripParams sp{};
        m_strip = FillStripParams(nullptr, nullptr, probeStore.dst_pitch, probeStore, &sp);
        m_period = m_strip && FusedPeriodTakes(sp);
    }
    if (m_stripPlanned && !m_strip) {
        FusedStripParams sp{};
        const Surface probe = m_plan.convert ? Surface{nullptr, (int)(w1 * SurfBytesPerPixel(m_plan.internal_fmt)), w1, h1, m_plan.internal_fmt}
                                             : Surface{nullptr, TexPitch(), m_srcWidth, m_srcHeight, RgbTexFmt(*m_srcParams)};
        m_stripSurf = FillStripSurfParams(probe, probeStore, &sp);
        m_period = m_stripSurf && FusedPeriodTakes(sp);
    }
    m_planDirty = false;
    UseLane(0);
    if (LogLevel() >= 2)
        std::fprintf(stderr, "mpcvr[%p]: plan %s (%dx%d -> %dx%d in %dx%d)\n", (void *)this, GetPathInfo().c_str(), m_srcRectWidth, m_srcRectHeight,
                     m_videoRect.Width(), m_videoRect.Height(), m_windowRect.Width(), m_windowRect.Height());
    return MPCVR_S_OK;
}

// dyadic, unrotated Jinc2m draws take their 16 weights from a phase table (vp_kernels.hip: k_jinc2_phases)
HRESULT CHipVideoProcessor::UploadJincPhases(const DrawCoords &dc, DevBuffer &buf, const void **tab, const float **ctr)
{
    *tab = nullptr; *ctr = nullptr;
    HRESULT hr;
    std::vector<unsigned char> host(JincPhasesBytes());
    if ((m_cfg.flags & MPCVR_FLAG_NO_FUSED) || !BuildJincPhases(dc, host.data())) {
        // the plain kernel: its per-column / per-row texture coordinates as a table (FillVertices' corner values interpolated in fp64, once per index)
        std::vector<float> c((size_t)dc.n_x + dc.n_y);
        BuildDrawCentres(dc, c.data());
        if ((hr = CheckHip(buf.CheckCreate(c.size() * sizeof(float)), "jinc centres"))) return hr;
        if ((hr = CheckHip(hipMemcpy(buf.ptr, c.data(), c.size() * sizeof(float), hipMemcpyHostToDevice), "jinc centres upload"))) return hr;
        *ctr = (const float *)buf.ptr;
        return MPCVR_S_OK;
    }
    if ((hr = CheckHip(buf.CheckCreate(host.size()), "jinc phases"))) return hr;
    if ((hr = CheckHip(hipMemcpy(buf.ptr, host.data(), host.size(), hipMemcpyHostToDevice), "jinc phases upload"))) return hr;
    *tab = buf.ptr;
    return MPCVR_S_OK;
}

void CHipVideoProcessor::UseLane(int lane)
{
    if (lane == 0) { m_run = m_stream; m_runConv = m_TexConvertOutput.ptr; m_runMid = m_TexResize.ptr; m_runPost = m_TexPost.ptr; return; }
    Lane &l = m_lanes[lane];
    m_run = l.stream; m_runConv = l.conv.ptr; m_runMid = l.mid.ptr; m_runPost = l.post.ptr;
}

HRESULT CHipVideoProcessor::PrepareLanes(int lanes)
{
    HRESULT hr;
    if (!m_fork && (hr = CheckHip(hipEventCreateWithFlags(&m_fork, hipEventDisableTiming), "fork event"))) return hr;
    for (int i = 1; i < lanes; i++) {
        Lane &l = m_lanes[i];
        if (!l.stream && (hr = CheckHip(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking), "lane stream"))) return hr;
        if (!l.done && (hr = CheckHip(hipEventCreateWithFlags(&l.done, hipEventDisableTiming), "lane event"))) return hr;
        if (m_convBytes && (hr = CheckHip(l.conv.CheckCreate(m_convBytes), "lane convert output"))) return hr;
        if (m_midBytes && (hr = CheckHip(l.mid.CheckCreate(m_midBytes), "lane resize texture"))) return hr;
        if (m_postBytes && (hr = CheckHip(l.post.CheckCreate(m_postBytes), "lane post-scale texture"))) return hr;
    }
    return MPCVR_S_OK;
}

// m_PSConvColorData.bEnable — DX11VideoProcessor.cpp:849-853: interleaved RGB skips the convert draw unless brightness
// or contrast are set (hue / saturation do not count)
bool CHipVideoProcessor::ConvertEnabled() const
{
    const FmtConvParams &f = *m_srcParams;
    if (m_doviValid) return true;                                              // :834
    if (f.CSType == CST_YUV || f.CSType == CST_GRAY || (f.CSType == CST_RGB && f.planes == 3)) return true;
    return std::fabs(m_procAmp.brightness / 255) > 1e-4f || std::fabs(m_procAmp.contrast - 1.0f) > 1e-4f;
}

int CHipVideoProcessor::TexPitch() const
{
    if (!m_srcParams) return m_srcPitch;
    if (m_srcParams->cformat == MPCVR_CF_V210) return V210TexPitch(m_srcWidth);
    if (m_srcParams->layout == LAY_RGB) return m_srcWidth * (m_srcParams->bits10 ? 4 : 4 * m_srcParams->bytes);
    return m_srcPitch;
}

// format of the texture an interleaved RGB sample is copied into (Helper.cpp:345-354)
static int RgbTexFmt(const FmtConvParams &f) { return f.bits10 ? SF_RGB10A2 : (f.bytes == 2 ? SF_RGBA16 : SF_BGRA8); }

// GetCopyPlaneFunction (Helper.cpp:377-412): every format handled here is copied as is (the <<6 of CopyPlane10to16 is
// applied when a texel is loaded) except v210, which CopyFrameV210 unpacks into a Y210 texture.
HRESULT CHipVideoProcessor::PrepareSample(const uint8_t *dev_sample, const uint8_t **tex)
{
    HRESULT hr;
    if (m_srcParams->cformat != MPCVR_CF_V210 && m_srcParams->layout != LAY_RGB) {
        if (((uintptr_t)dev_sample & 3) == 0) { *tex = dev_sample; return MPCVR_S_OK; }
        // a device sample that does not start on a dword: the kernels read rows with 4- / 16-byte loads, so it is copied into
        // the context's own texture first — what the reference does with EVERY decoder-owned sample (CopySubresourceRegion,
        // DX11VideoProcessor.cpp:2563-2569)
        const size_t bytes = (size_t)m_srcPitch * m_srcLines;
        if ((hr = CheckHip(m_TexSrcVideo.CheckCreate(bytes), "m_TexSrcVideo"))) return hr;
        m_texSrcZeroed = false;
        // the copy runs on the context stream: behind every lane frame that still reads the texture, and in front of the lane frames to come
        (void)JoinFrameLanes(false);
        NoteStreamWork();
        if ((hr = CheckHip(hipMemcpyAsync(m_TexSrcVideo.ptr, dev_sample, bytes, hipMemcpyDeviceToDevice, m_stream), "sample copy"))) return hr;
        *tex = (const uint8_t *)m_TexSrcVideo.ptr;
        return MPCVR_S_OK;
    }
    const int tp = TexPitch();
    (void)JoinFrameLanes(false);
    NoteStreamWork();                 // (the repacks below run on the context stream)
    const bool fresh = m_TexSrcVideo.size < (size_t)tp * m_srcHeight || !m_TexSrcVideo.ptr || !m_texSrcZeroed;
    if ((hr = CheckHip(m_TexSrcVideo.CheckCreate((size_t)tp * m_srcHeight), "m_TexSrcVideo"))) return hr;
    if (m_srcParams->layout == LAY_RGB) {
        // texels the reference's copy loop never writes (RGB48 remainder) stay zero
        if (fresh && (hr = CheckHip(hipMemsetAsync(m_TexSrcVideo.ptr, 0, (size_t)tp * m_srcHeight, m_stream), "clear texture"))) return hr;
        m_texSrcZeroed = true;
        if ((hr = CheckHip(LaunchRepackRgb(m_srcParams->repack, dev_sample, m_srcBottomUp ? -m_srcPitch : m_srcPitch,
                                           (uint8_t *)m_TexSrcVideo.ptr, tp, m_srcWidth, m_srcHeight, m_stream), "k_repack_rgb"))) return hr;
        *tex = (const uint8_t *)m_TexSrcVideo.ptr;
        return MPCVR_S_OK;
    }
    m_texSrcZeroed = false;
    if ((hr = CheckHip(LaunchRepackV210(dev_sample, m_srcPitch, (uint8_t *)m_TexSrcVideo.ptr, tp, m_srcHeight, m_stream), "k_repack_v210"))) return hr;
    *tex = (const uint8_t *)m_TexSrcVideo.ptr;
    return MPCVR_S_OK;
}

void CHipVideoProcessor::FillConvertParams(const uint8_t *sample, ConvertParams *P) const
{
    const FmtConvParams &f = *m_srcParams;
    std::memset(P, 0, sizeof(*P));
    // plane walk of MemCopyToTexSrcVideo — DX11VideoProcessor.cpp:1213-1252
    const int pitch0 = TexPitch();
    const int cromaH = m_srcHeight / f.div_h;
    const int cromaPitch = (f.planes == 3) ? pitch0 / f.div_w : pitch0;
    P->plane[0] = sample;
    P->plane[1] = (sample && f.planes > 1) ? sample + (size_t)pitch0 * m_srcHeight : nullptr;
    P->plane[2] = (sample && f.planes > 2) ? P->plane[1] + (size_t)cromaPitch * cromaH : nullptr;
    P->pitch[0] = pitch0; P->pitch[1] = cromaPitch; P->pitch[2] = cromaPitch;
    P->tex_w = m_srcWidth; P->tex_h = m_srcHeight;
    P->cw = m_srcWidth / f.div_w; P->ch = cromaH;
    P->rect_l = m_srcRect.left; P->rect_t = m_srcRect.top;
    P->out_w = m_srcRectWidth; P->out_h = m_srcRectHeight;
    P->fmt.planes = f.planes; P->fmt.bytes = f.bytes; P->fmt.div_w = f.div_w; P->fmt.div_h = f.div_h;
    P->fmt.shift = f.shift; P->fmt.v_first = f.v_first; P->fmt.subsampling = f.Subsampling; P->fmt.cdepth = f.CDepth;
    // m_pPSConvertColorDeint: 4:2:0 planar/bi-planar only (:2964), used for interlaced samples when bDeintBlend (:3075)
    P->blend_deint = (m_cfg.bDeintBlend && m_SampleFormat != 0 && f.Subsampling == 420 && f.planes >= 2) ? 1 : 0;
    P->fmt.layout = f.layout; P->fmt.bits10 = f.bits10;
    for (int k = 0; k < 4; k++) P->fmt.ci[k] = f.ci[k];
    P->chroma_scaling = m_cfg.iChromaScaling;
    switch (m_srcExFmt.VideoChromaSubsampling()) {          // Shaders.cpp:121-137
    case 7: P->chroma_loc = CLOC_COSITED; break;
    case 1: P->chroma_loc = CLOC_MPEG1; break;
    default: P->chroma_loc = CLOC_MPEG2; break;
    }
    P->tail = m_tail; P->gamma = m_gamma;
    std::memcpy(P->cm, m_cm, sizeof(m_cm));
    P->lum_scale = m_lumScale;
    std::memcpy(P->gamut, m_gamut, sizeof(m_gamut));
    P->out_fmt = m_plan.internal_fmt;
    P->dovi = !m_doviValid ? nullptr : m_dvTabDev ? m_dvTabDev : (const DoviParams *)m_doviDev.ptr;     // (m_dvTabDev: one RPU per frame, ProcessBatchDovi)
    P->pq_lut = (m_pqLutValid && !(m_cfg.flags & (MPCVR_FLAG_NO_LUT | MPCVR_FLAG_NO_FUSED))) ? (const float *)m_pqLut.ptr : nullptr;
}

StoreParams CHipVideoProcessor::MakeStore(void *dst, int pitch, int dstFmt, bool rt) const
{
    StoreParams s{};
    s.dst = dst; s.dst_pitch = pitch; s.dst_fmt = dstFmt;
    s.mode = ST_SURFACE; s.mid_fmt = m_plan.internal_fmt; s.quant = m_plan.quant;
    s.dither = (const uint16_t *)m_dither.ptr;
    if (rt) {
        s.off_x = m_videoRect.left; s.off_y = m_videoRect.top;
        s.clip_w = m_windowRect.Width(); s.clip_h = m_windowRect.Height();
        if (m_plan.final_pass) s.mode = ST_FINAL;
    }
    return s;
}

void CHipVideoProcessor::FillFusedParams(const uint8_t *sample, void *rt, int rtPitch, FusedParams *fp) const
{
    FillConvertParams(sample, &fp->conv);
    fp->plane_off[0] = 0;
    fp->plane_off[1] = (size_t)m_srcPitch * m_srcHeight;
    fp->plane_off[2] = fp->plane_off[1] + (size_t)fp->conv.pitch[1] * fp->conv.ch;
    fp->wx = m_upX; fp->wy = m_upY;
    fp->out_w = m_videoRect.Width(); fp->out_h = m_videoRect.Height();
    fp->store = MakeStore(rt, rtPitch, m_plan.swap_fmt, true);
    const bool no_lut = (m_cfg.flags & MPCVR_FLAG_NO_LUT) != 0;
    fp->pq_lut = (m_pqLutValid && !no_lut) ? (const float *)m_pqLut.ptr : nullptr;
    fp->literal_tail = no_lut ? 1 : 0;
    fp->hlg_lut = (m_tail == TAIL_HLG_TO_SDR && !no_lut) ? (const float *)m_hlgLut.ptr : nullptr;
    fp->eotf_lut = (m_doviValid && !no_lut) ? (const float *)m_eotfLut.ptr : nullptr;
    fp->dovi_l2 = (m_doviValid && m_doviHost.l2_enabled) ? 1 : 0;
    fp->dovi_cm = (m_doviValid && m_dvTabDev) ? m_dvCmDev : nullptr;
    fp->jinc_tab = m_plan.fused_jinc ? m_jincFusedTab : nullptr;
    fp->exact_wide = m_plan.hdr_tonemap ? 1 : 0;
    fp->inflight = m_inflight;
    fp->dst_aligned16 = (((uintptr_t)rt) & 15) == 0;        // batches: ProcessBatch checks every target
    fp->src_aligned16 = (((uintptr_t)sample) & 15) == 0;
    // vectorised convert: dword loads need 4-byte aligned rows and a source rect starting on a 4-px boundary
    fp->fast_convert = (m_srcRect.left % 4 == 0) && (m_srcRect.top % 2 == 0) && (m_srcPitch % 4 == 0) &&
                       (fp->conv.pitch[1] % 4 == 0) && (fp->plane_off[1] % 4 == 0) && (fp->plane_off[2] % 4 == 0) &&
                       !(m_cfg.flags & MPCVR_FLAG_NO_FAST_CONVERT);
}

// ------------------------------------------------------------------------------------------------
// CopySample — DX11VideoProcessor.cpp:2202-2597 (memory branch -> MemCopyToTexSrcVideo :1213-1252)
// ------------------------------------------------------------------------------------------------
HRESULT CHipVideoProcessor::CopySample(const void *data, int pitch, int memKind)
{
    if (!m_bInit || !m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    if (!data) return Fail(MPCVR_E_POINTER, "null sample");
    if (pitch != (m_srcBottomUp ? -m_srcPitch : m_srcPitch)) return Fail(MPCVR_E_UNEXPECTED, "sample pitch differs from the media type");   // :2545
    (void)hipSetDevice(m_device);
    const size_t bytes = (size_t)m_srcPitch * m_srcLines;
    HRESULT hr;
    const auto t_host0 = std::chrono::steady_clock::now();
    struct HostTimer { CHipVideoProcessor *self; std::chrono::steady_clock::time_point t0;
                       ~HostTimer() { self->m_copyHostMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } host_timer{this, t_host0};
    MarkConsumed();                                  // the previous sample is done with as far as the host is concerned
    m_curSlot = -1;
    m_lastRun = nullptr;
    if (memKind == MPCVR_MEM_DEVICE) {               // zero-copy, cf. the IMediaSampleD3D11 branch :2528-2569
        return PrepareSample((const uint8_t *)data, &m_curSample);
    }
    if (memKind != MPCVR_MEM_HOST && memKind != MPCVR_MEM_HOST_PINNED) return Fail(MPCVR_E_INVALIDARG, "mem_kind");
    if (!m_copyStream && (hr = CheckHip(hipStreamCreateWithFlags(&m_copyStream, hipStreamNonBlocking), "copy stream"))) return hr;
    const int si = m_upNext;
    m_upNext = (m_upNext + 1) % kUploadSlots;
    UploadSlot &u = m_up[si];
    if (!u.uploaded) {
        if ((hr = CheckHip(hipEventCreateWithFlags(&u.uploaded, hipEventDisableTiming), "upload event"))) return hr;
        if ((hr = CheckHip(hipEventCreateWithFlags(&u.consumed, hipEventDisableTiming), "consume event"))) return hr;
    }
    // the slot's staging / device buffers are free once the work that last used them has completed
    if (u.inFlight && (hr = CheckHip(hipEventSynchronize(u.consumedRecorded ? u.consumed : u.uploaded), "upload slot wait"))) return hr;
    if ((hr = CheckHip(u.dev.CheckCreate(bytes), "upload buffer"))) return hr;
    const void *from = data;
    if (memKind == MPCVR_MEM_HOST) {
        if (u.pinnedSize < bytes) {
            if (u.pinned) (void)hipHostFree(u.pinned);
            u.pinned = nullptr; u.pinnedSize = 0;
            if ((hr = CheckHip(hipHostMalloc(&u.pinned, bytes, hipHostMallocDefault), "pinned staging"))) return hr;
            u.pinnedSize = bytes;
        }
        std::memcpy(u.pinned, data, bytes);          // the reference's only per-frame CPU work (MemCopyToTexSrcVideo); the
        from = u.pinned;                             // <<6 / v210 / RGB repacks happen on the device (PrepareSample)
    }
    // the new data must not overtake work of the main stream that still reads this device buffer
    if (!m_evUp0 && ((hr = CheckHip(hipEventCreate(&m_evUp0), "upload timer")) || (hr = CheckHip(hipEventCreate(&m_evUp1), "upload timer")))) return hr;
    (void)hipEventRecord(m_evUp0, m_copyStream);
    if ((hr = CheckHip(hipMemcpyAsync(u.dev.ptr, from, bytes, hipMemcpyHostToDevice, m_copyStream), "upload"))) return hr;
    (void)hipEventRecord(m_evUp1, m_copyStream);
    m_upTimed = true;
    if ((hr = CheckHip(hipEventRecord(u.uploaded, m_copyStream), "upload event"))) return hr;
    if ((hr = CheckHip(hipStreamWaitEvent(m_stream, u.uploaded, 0), "stream wait"))) return hr;
    u.inFlight = true; u.consumedRecorded = false;
    m_curSlot = si;
    return PrepareSample((const uint8_t *)u.dev.ptr, &m_curSample);
}

// a Process / Render that read the current upload slot has been queued: the slot may be recycled after it
void CHipVideoProcessor::MarkConsumed()
{
    if (m_curSlot < 0) return;
    UploadSlot &u = m_up[m_curSlot];
    if (u.consumed && hipEventRecord(u.consumed, m_lastRun ? m_lastRun : m_stream) == hipSuccess) u.consumedRecorded = true;
}

HRESULT CHipVideoProcessor::ConvertColorPass(const uint8_t *sample)
{
    ConvertParams P;
    FillConvertParams(sample, &P);
    Surface out{m_runConv, (int)(m_srcRectWidth * SurfBytesPerPixel(m_plan.internal_fmt)),
                m_srcRectWidth, m_srcRectHeight, m_plan.internal_fmt};
    if (!(m_cfg.flags & MPCVR_FLAG_NO_FUSED)) {           // the fused kernel's block convert, when the source qualifies
        FusedParams fp{};
        FillFusedParams(sample, out.ptr, out.pitch, &fp);
        fp.store = MakeStore(out.ptr, out.pitch, out.fmt, false);
        fp.dst_aligned16 = 1;
        fp.exact_convert = 1;            // m_TexConvertOutput in front of a draw
        if (ConvertBlocksSupported(fp, false))
            return CheckHip(LaunchConvertBlocks(fp, nullptr, FusedFrame{sample, out.ptr}, 1, m_run), "k_convert_blocks");
    }
    return CheckHip(LaunchConvert(P, out, m_run, (m_cfg.flags & MPCVR_FLAG_NO_FUSED) != 0), "k_convert");
}

// ResizeShaderPass (:3103-3187) with FinalPass (:3189-3233) folded into the epilogue of the last draw
HRESULT CHipVideoProcessor::ResizeShaderPass(void *rt, int rtPitch, const uint8_t *sample)
{
    const int w1 = m_srcRectWidth, h1 = m_srcRectHeight, w2 = m_videoRect.Width(), h2 = m_videoRect.Height();
    Surface conv{m_runConv, (int)(w1 * SurfBytesPerPixel(m_plan.internal_fmt)), w1, h1, m_plan.internal_fmt};
    if (!m_plan.convert)      // pInputTexture = &m_TexSrcVideo (:3321-3323)
        conv = Surface{(void *)sample, TexPitch(), m_srcWidth, m_srcHeight, RgbTexFmt(*m_srcParams)};
    const StoreParams final = MakeStore(rt, rtPitch, m_plan.swap_fmt, true);
    // with the HDR10 tone-mapping step the resize draws into a post-scale texture (internal format, video-rect sized)
    // and the step itself writes the render target / runs the final pass (:3359-3367)
    Surface post{m_runPost, (int)(w2 * SurfBytesPerPixel(m_plan.internal_fmt)), w2, h2, m_plan.internal_fmt};
    const StoreParams last = m_plan.hdr_tonemap ? MakeStore(post.ptr, post.pitch, m_plan.internal_fmt, false) : final;
    HRESULT hr = MPCVR_S_OK;
    bool drawn = true;
    const bool plain = (m_cfg.flags & MPCVR_FLAG_NO_FUSED) != 0;      // keep the whole path on the one-kernel-fits-all versions
    const bool jfast = !(m_cfg.flags & (MPCVR_FLAG_NO_FUSED | MPCVR_FLAG_NO_FAST_CONVERT));       // Jinc2m quad kernel: default tier only
    FusedStripParams ssp{};
    if (m_stripSurf && FillStripSurfParams(conv, last, &ssp)) {
        // convert output (or RGB source texture) -> both draws -> final pass in the arbitrary-ratio fused kernel, no convert stage
        hr = CheckHip(LaunchFusedStrip(ssp, nullptr, FusedFrame{(const uint8_t *)conv.ptr, last.dst}, 1, m_run), "k_fused_strip<surface>");
    } else if (m_plan.two_pass && !plain && !m_firstJinc && !m_secondJinc && m_firstAxis == 0 && !m_firstSwap &&
        Resize2DSupported(conv, m_tapsX, m_tapsY, last)) {
        // both draws in one LDS-tiled kernel: m_TexResize stays on chip
        hr = CheckHip(LaunchResize2D(conv, m_tapsX, m_tapsY, (const int32_t *)m_otherX.ptr, m_plan.mid_h, w2, h2, last, m_run), "k_resize_2d");
    } else if (m_plan.two_pass) {
        Surface mid{m_runMid, w2 * 8, w2, m_plan.mid_h, SF_RGBA16F};
        StoreParams st = MakeStore(mid.ptr, mid.pitch, SF_RGBA16F, false);
        if (m_firstJinc) hr = CheckHip(LaunchJinc2(conv, m_firstCoords, w2, m_plan.mid_h, st, m_run, m_jincFirstTab, jfast, nullptr, m_jincFirstCtr), "k_jinc2");
        else hr = CheckHip(LaunchResize(m_firstAxis, m_firstSwap, conv, m_tapsX, (const int32_t *)m_otherX.ptr, w2, m_plan.mid_h, st, m_run, plain), "k_resize<first>");
        if (hr) return hr;
        if (m_secondJinc) hr = CheckHip(LaunchJinc2(mid, m_secondCoords, w2, h2, last, m_run, m_jincSecondTab, jfast, nullptr, m_jincSecondCtr), "k_jinc2");
        else hr = CheckHip(LaunchResize(1, false, mid, m_tapsY, (const int32_t *)m_otherY.ptr, w2, h2, last, m_run, plain), "k_resize<Y>");
    } else if (m_plan.one_pass) {
        if (m_firstJinc) hr = CheckHip(LaunchJinc2(conv, m_firstCoords, w2, h2, last, m_run, m_jincFirstTab, jfast, nullptr, m_jincFirstCtr), "k_jinc2");
        else hr = CheckHip(LaunchResize(m_firstAxis, m_firstSwap, conv, m_tapsX, (const int32_t *)m_otherX.ptr, w2, h2, last, m_run, plain), "k_resize<one>");
    } else {
        drawn = false;
        if (!m_plan.convert) {    // the next step reads the source rect of the texture (pTex = pInputTexture, :3352)
            const int bpp = conv.fmt == SF_RGBA16 ? 8 : 4;
            conv.ptr = (uint8_t *)conv.ptr + (size_t)m_srcRect.top * conv.pitch + (size_t)m_srcRect.left * bpp;
            conv.w = w1; conv.h = h1;
        }
        if (!m_plan.hdr_tonemap) {
            StoreParams direct = final;
            if (!m_plan.convert) direct.mid_fmt = conv.fmt;   // nothing was drawn into m_TexsPostScale: the final pass sees the texture's own precision
            return CheckHip(LaunchCopy(conv, w2, h2, direct, m_run), "k_copy");
        }
    }
    if (hr || !m_plan.hdr_tonemap) return hr;
    return CheckHip(LaunchHdr10ToneMap(drawn ? post : conv, m_hdrTm, w2, h2, final, m_run), "k_hdr10_tonemap");
}

// parameters of the arbitrary-ratio fused kernel for one launch; false: this launch cannot take it (alignment, sizes)
bool CHipVideoProcessor::FillStripParams(const uint8_t *sample, void *dst, int dstPitch, const StoreParams &store, FusedStripParams *sp) const
{
    FillFusedParams(sample, dst, dstPitch, &sp->fp);
    sp->fp.store = store;
    sp->ran_period = &m_stripRan;
    const int32_t *tab = (const int32_t *)m_stripTab.ptr;
    sp->yrange = tab + m_stripOff[0]; sp->xstrip = tab + m_stripOff[1];
    sp->xi_t = tab + m_stripOff[2]; sp->xw_t = tab + m_stripOff[3];
    sp->yi = tab + m_stripOff[4]; sp->yw = tab + m_stripOff[5];
    sp->out_w = m_videoRect.Width(); sp->out_h = m_videoRect.Height();
    sp->nt = m_stripPlan.nt; sp->pxl = m_stripPlan.pxl; sp->strip_w = m_stripPlan.strip_w; sp->ring = m_stripPlan.ring; sp->acols = m_stripPlan.acols;
    sp->per_P = 0;
    if (m_periodPlan.P && !(m_cfg.flags & MPCVR_FLAG_NO_PERIOD)) {
        sp->per_P = m_periodPlan.P; sp->per_Q = m_periodPlan.Q; sp->per_nt = m_periodPlan.nt; sp->per_acols = m_periodPlan.acols; sp->per_strip_w = m_periodPlan.strip_w; sp->per_own = m_periodPlan.own; sp->per_force = (m_cfg.flags & MPCVR_FLAG_FORCE_PERIOD) ? 1 : 0;
        sp->per_xi_t = tab + m_periodOff[0]; sp->per_xw_t = tab + m_periodOff[1]; sp->per_yw = tab + m_periodOff[2]; sp->per_xstrip = tab + m_periodOff[3];
    }
    return FusedStripSupported(*sp) && FusedStripLdsBytes(*sp) <= DeviceLdsLimit();
}

// the same kernel without its convert stage: `src` = m_TexConvertOutput (any convert kernel wrote it) or the RGB source texture
bool CHipVideoProcessor::FillStripSurfParams(const Surface &src, const StoreParams &store, FusedStripParams *sp) const
{
    *sp = FusedStripParams{};
    sp->fp.store = store;
    sp->ran_period = &m_stripRan;
    sp->fp.dst_aligned16 = (((uintptr_t)store.dst) & 15) == 0;      // (batches: the caller knows every target of the table and overrides it)
    const int32_t *tab = (const int32_t *)m_stripTab.ptr;
    sp->yrange = tab + m_stripOff[0]; sp->xstrip = tab + m_stripOff[1];
    sp->xi_t = tab + m_stripOff[2]; sp->xw_t = tab + m_stripOff[3];
    sp->yi = tab + m_stripOff[4]; sp->yw = tab + m_stripOff[5];
    sp->out_w = m_videoRect.Width(); sp->out_h = m_videoRect.Height();
    sp->nt = m_stripPlan.nt; sp->pxl = m_stripPlan.pxl;
    sp->strip_w = m_stripPlan.strip_w; sp->ring = m_stripPlan.ring; sp->acols = m_stripPlan.acols;
    sp->surface_mode = 1;
    sp->surf = src;
    sp->other = m_otherX.ptr && !m_tapsX.other_identity ? (const int32_t *)m_otherX.ptr : nullptr;
    sp->mid_h = m_plan.mid_h;
    sp->per_P = 0;
    if (m_periodPlan.P && !(m_cfg.flags & MPCVR_FLAG_NO_PERIOD)) {      // periodic vertical ratio: the register-window kernel reads the surface as well
        sp->per_P = m_periodPlan.P; sp->per_Q = m_periodPlan.Q; sp->per_nt = m_periodPlan.nt; sp->per_acols = m_periodPlan.acols; sp->per_strip_w = m_periodPlan.strip_w; sp->per_own = m_periodPlan.own; sp->per_force = 1;
        sp->per_xi_t = tab + m_periodOff[0]; sp->per_xw_t = tab + m_periodOff[1]; sp->per_yw = tab + m_periodOff[2]; sp->per_xstrip = tab + m_periodOff[3];
    }
    return FusedStripSupported(*sp) && FusedStripLdsBytes(*sp) <= DeviceLdsLimit();
}

HRESULT CHipVideoProcessor::ProcessOne(const uint8_t *sample, void *rt, int rtPitch)
{
    HRESULT hr;
    if (m_plan.fused_up2x) {
        FusedParams fp{};
        FillFusedParams(sample, rt, rtPitch, &fp);
        const FusedFrame fr{sample, rt};        // a single frame travels by value in the kernel arguments
        return CheckHip(LaunchFusedUp2x(fp, nullptr, fr, 1, m_run), "k_fused_up2x");
    }
    if (m_strip) {
        // with the HDR10 tone-mapping step the resize draws into the post-scale texture and the step writes the target (:3359-3367)
        const int w2 = m_videoRect.Width(), h2 = m_videoRect.Height();
        Surface post{m_runPost, (int)(w2 * SurfBytesPerPixel(m_plan.internal_fmt)), w2, h2, m_plan.internal_fmt};
        const StoreParams final = MakeStore(rt, rtPitch, m_plan.swap_fmt, true);
        const StoreParams last = m_plan.hdr_tonemap ? MakeStore(post.ptr, post.pitch, m_plan.internal_fmt, false) : final;
        FusedStripParams sp{};
        if (FillStripParams(sample, last.dst, last.dst_pitch, last, &sp)) {
            if ((hr = CheckHip(LaunchFusedStrip(sp, nullptr, FusedFrame{sample, last.dst}, 1, m_run), "k_fused_strip"))) return hr;
            if (!m_plan.hdr_tonemap) return MPCVR_S_OK;
            return CheckHip(LaunchHdr10ToneMap(post, m_hdrTm, w2, h2, final, m_run), "k_hdr10_tonemap");
        }
    }
    if (m_plan.direct_convert) {
        FusedParams fp{};
        FillFusedParams(sample, rt, rtPitch, &fp);
        if (ConvertBlocksSupported(fp, true))
            return CheckHip(LaunchConvertBlocks(fp, nullptr, FusedFrame{sample, rt}, 1, m_run), "k_convert_blocks");
        ConvertParams P;
        FillConvertParams(sample, &P);
        return CheckHip(LaunchConvertDirect(P, MakeStore(rt, rtPitch, m_plan.swap_fmt, true), m_run), "k_convert_direct");
    }
    if (m_plan.convert && (hr = ConvertColorPass(sample))) return hr;
    return ResizeShaderPass(rt, rtPitch, sample);
}

// Process — DX11VideoProcessor.cpp:3285-3424
HRESULT CHipVideoProcessor::Process(void *pRenderTarget, int rtPitch, const CRect *srcRect, const CRect *dstRect, bool /*second*/)
{
    if (!m_bInit || !m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    if (!pRenderTarget) return Fail(MPCVR_E_POINTER, "null render target");
    if (!m_curSample) return Fail(MPCVR_E_NOT_VALID_STATE, "no sample: call CopySample first");
    if (srcRect && !srcRect->IsRectNull() && *srcRect != m_srcRect)
        return Fail(MPCVR_E_INVALIDARG, "src_rect must equal the input's source rect");
    (void)hipSetDevice(m_device);
    HRESULT hr;
    if (dstRect && !dstRect->IsRectNull()) { if ((hr = SetVideoRect(*dstRect))) return hr; }
    if (rtPitch < m_windowRect.Width() * 4) return Fail(MPCVR_E_INVALIDARG, "render-target pitch smaller than a row");
    if (m_planDirty && (hr = UpdatePlan())) return hr;
    UseLane(0);
    FrameLane *fl = (m_noLanesOnce || !FrameLanesUsable()) ? nullptr : PickFrameLane(pRenderTarget);
    m_inflight = fl ? FrameLaneCount() : 1;           // the kernels size their segments for that many frames side by side
    if (fl) {
        m_run = fl->stream;
        LaneWaitsForStream(fl);           // behind a batch / an off-lane frame / a sample copy still queued on the context stream
        // the sample's upload (copy stream) was ordered in front of the context stream by CopySample: the lane needs the same edge
        if (m_curSlot >= 0 && m_up[m_curSlot].uploaded) (void)hipStreamWaitEvent(fl->stream, m_up[m_curSlot].uploaded, 0);
        if (m_clearOnRun) (void)hipMemsetAsync(m_BackBuffer.ptr, 0, m_clearOnRun, fl->stream);
    } else {
        // strictly in stream order behind whatever the lanes still hold (a plan that left the lanes, a caller's stream, the snapshot)
        (void)JoinFrameLanes(false);
        NoteStreamWork();
        if (m_clearOnRun) (void)hipMemsetAsync(m_BackBuffer.ptr, 0, m_clearOnRun, m_stream);
    }
    m_clearOnRun = 0;
    // the timing pair (m_RenderStats.paintticks' stand-in): every frame off the lanes; on the lanes one frame in eight — two timestamped
    // events per frame are two more packets in front of and behind a 45 us kernel on each of four queues (same box, timed every frame /
    // every 8th: 4K -> 8K 20.03 k -> 20.35 k frames/s, 1080p same size 111 k -> 121-162 k; profiles/r04/ab_call29_lane_timing.jsonl)
    static const int every = [] { const char *e = std::getenv("MPCVR_LANE_TIMING_EVERY"); const int v = e && *e ? std::atoi(e) : 8; return v < 1 ? 1 : v; }();
    const bool timeIt = !fl || every == 1 || (m_laneFrames++ % (unsigned)every) == 0 || !m_timed;
    if (timeIt) (void)hipEventRecord(m_evStart, m_run);
    if (m_plan.errdiff) {
        // EXTENSION (bUseDither = 2): the draws render into the window-sized R10G10B10A2 intermediate, as for a 10-bit swap chain; the
        // error-diffusion pass takes it to the render target
        if (!(hr = PrepareErrDiff(1)) && !(hr = ProcessOne(m_curSample, m_edBase, m_edPitch)))
            hr = ErrDiffPass(1, nullptr, FusedFrame{m_edBase, pRenderTarget}, &pRenderTarget, rtPitch, m_run);
    } else
        hr = ProcessOne(m_curSample, pRenderTarget, rtPitch);
    if (timeIt) (void)hipEventRecord(m_evStop, m_run);
    m_lastRun = m_run;
    MarkConsumed();
    if (fl) NoteLaneFrame(fl, pRenderTarget);
    m_inflight = 1;
    UseLane(0);
    m_timed = true;
    return hr;
}

HRESULT CHipVideoProcessor::ProcessBatch(int n, const void *const *srcs, void *const *dsts, int rtPitch)
{
    const unsigned before = m_launches;
    HRESULT hr = MPCVR_S_OK;
    if (m_cfg.bUseDither == MPCVR_DITHER_ErrorDiffusion_EXT && m_bInit && m_srcParams && n > 0 && srcs && dsts && m_planDirty) hr = UpdatePlan();
    if (!hr) hr = (m_bInit && m_srcParams && n > 0 && srcs && dsts && !m_planDirty && m_plan.errdiff) ? ProcessBatchErrDiff(n, srcs, dsts, rtPitch)
                                                                                                     : ProcessBatchRoutes(n, srcs, dsts, rtPitch);
    m_lastBatchFrames = n; m_lastBatchLaunches = (int)(m_launches - before);
    return hr;
}

HRESULT CHipVideoProcessor::ProcessBatchRoutes(int n, const void *const *srcs, void *const *dsts, int rtPitch)
{
    // a batch that is one launch with nothing shared runs on one of two lanes, beside the batch before it (FrameLane); everything else —
    // and every batch of a context on a caller's stream — in stream order on the context stream
    FrameLane *bl = nullptr;
    if (m_bInit && m_srcParams && n > 1 && srcs && dsts && !m_keepStart) {
        (void)hipSetDevice(m_device);
        HRESULT hr;
        if (m_planDirty && (hr = UpdatePlan())) return hr;
        if (BatchLanesUsable(n, srcs, dsts, rtPitch)) bl = PickBatchLane(n, dsts);
    }
    m_lastBatchLane = bl ? (int)(bl - m_flanes) : -1;
    if (!bl) return ProcessBatchRoutesOn(n, srcs, dsts, rtPitch);
    LaneWaitsForStream(bl);                  // behind whatever the context stream was given since the lane last looked
    hipStream_t const ctx = m_stream;
    m_stream = bl->stream; m_batchOnLane = true;
    UseLane(0);
    const HRESULT hr = ProcessBatchRoutesOn(n, srcs, dsts, rtPitch);
    m_stream = ctx; m_batchOnLane = false;
    UseLane(0);
    NoteLaneBatch(bl, n, dsts);
    return hr;
}

HRESULT CHipVideoProcessor::ProcessBatchRoutesOn(int n, const void *const *srcs, void *const *dsts, int rtPitch)
{
    if (!m_bInit || !m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    if (n <= 0 || !srcs || !dsts) return Fail(MPCVR_E_INVALIDARG, "empty batch");
    if (rtPitch < m_windowRect.Width() * 4) return Fail(MPCVR_E_INVALIDARG, "render-target pitch smaller than a row");
    (void)hipSetDevice(m_device);
    HRESULT hr;
    if (!m_keepStart) m_startRecorded = false;
    if (m_planDirty && (hr = UpdatePlan())) return hr;
    if (!m_batchOnLane) {
        (void)JoinFrameLanes(false);         // a batch runs on the context stream, behind every single frame still in flight
        NoteStreamWork();                    // ... and single frames queued after it run behind the batch (LaneWaitsForStream)
    }
    for (int i = 0; i < n; i++)
        if (!srcs[i] || !dsts[i]) return Fail(MPCVR_E_POINTER, "null frame in batch");
    // v210 samples are repacked into m_TexSrcVideo's layout first (CopyFrameV210, Helper.cpp:709-748): a batch gets one repack launch
    // per 32 frames into the slots of a batch texture, and the whole-batch launches below read the slots as if they were the samples
    std::vector<const void *> slots;
    m_batchRepacked = false;
    if (m_srcParams->cformat == MPCVR_CF_V210 && n > 1 && !(m_cfg.flags & (MPCVR_FLAG_NO_FUSED | MPCVR_FLAG_NO_FAST_CONVERT))) {
        const int tp = TexPitch();
        const size_t texBytes = ((size_t)tp * m_srcHeight + 255) & ~(size_t)255;
        if (texBytes * (size_t)n <= ((size_t)1 << 30)) {
            if ((hr = CheckHip(m_batchTex.CheckCreate(texBytes * n), "batch source texture"))) return hr;
            slots.resize(n);
            if (!m_startRecorded) (void)hipEventRecord(m_evStart, m_stream);      // the repack is part of the batch's process time, as on the other branches
            m_startRecorded = true;
            if ((hr = CheckHip(LaunchRepackV210(nullptr, m_srcPitch, (uint8_t *)m_batchTex.ptr, tp, m_srcHeight, m_stream, srcs, n, texBytes), "k_repack_v210"))) return hr;
            m_batchTexZeroed = false;                       // (the RGB batches' zeroed remainder columns are gone)
            for (int i = 0; i < n; i++) slots[i] = (uint8_t *)m_batchTex.ptr + (size_t)i * texBytes;
            srcs = slots.data();
            m_batchRepacked = true;
        }
    }
    bool aligned = true, src4 = true;
    m_batchSrc16 = true;
    for (int i = 0; i < n; i++) {
        if (((uintptr_t)srcs[i] & 15) != 0) m_batchSrc16 = false;
        if (((uintptr_t)srcs[i] & 3) != 0) src4 = false;
        if (((uintptr_t)dsts[i] & 15) != 0) aligned = false;
    }
    // samples that are repacked into m_TexSrcVideo first (v210, interleaved RGB) cannot be read in place by a whole-batch launch:
    // they take the frame-by-frame branch below like samples that do not start on a dword (v210 became a fused-2x / strip
    // candidate when packed 4:2:2 joined the block convert)
    if ((m_srcParams->cformat == MPCVR_CF_V210 && !m_batchRepacked) || m_srcParams->layout == LAY_RGB) src4 = false;
    // pass-per-kernel path, whole batch per launch: possible when every stage has a kernel with a frame dimension
    // the arbitrary-ratio fused kernel takes the whole batch in one launch, like the 2x kernel
    FusedStripParams strip_sp{};
    const bool strip = m_strip && !m_plan.fused_up2x && !m_plan.hdr_tonemap && src4 && !m_dvFrames &&
                       FillStripParams((const uint8_t *)srcs[0], dsts[0], rtPitch, MakeStore(dsts[0], rtPitch, m_plan.swap_fmt, true), &strip_sp);
    bool batchable = false;
    if (!m_plan.fused_up2x && !strip && n > 1 && src4 && !(m_cfg.flags & (MPCVR_FLAG_NO_FUSED | MPCVR_FLAG_NO_FAST_CONVERT))) {
        FusedParams a{}, b{};
        batchable = BatchPlan((const uint8_t *)srcs[0], dsts[0], rtPitch, aligned, &a, &b);
        // one RPU per frame (ProcessBatchDovi): the block convert's Dolby Vision variants index the run's tables by the frame; the HDR10
        // tone-mapping step takes its level-1 constants by value, so such a run goes frame by frame
        if (m_dvFrames && (!m_dvTabReady || m_plan.hdr_tonemap)) batchable = false;
        if (batchable && m_dvFrames) { m_dvTabDev = m_dvTabReady; m_dvCmDev = m_dvCmReady; }
    }
    if (batchable && m_plan.direct_convert && n <= kHostTableMax) {
        // same-size frames: one convert launch with the frame table in its kernel arguments (32 frames for the block convert, 128 where the
        // streaming kernel takes the launch — it answers hipErrorInvalidValue otherwise and the table is uploaded below)
        FusedParams conv{}, direct{};
        if (!BatchPlan((const uint8_t *)srcs[0], dsts[0], rtPitch, aligned, &conv, &direct)) return Fail(MPCVR_E_UNEXPECTED, "batch plan changed");
        FusedFrame tab[kHostTableMax];
        for (int i = 0; i < n; i++) tab[i] = FusedFrame{(const uint8_t *)srcs[i], dsts[i]};
        if (!m_startRecorded) { (void)hipEventRecord(m_evStart, m_stream); m_startRecorded = true; }
        const hipError_t e = LaunchConvertBlocks(direct, nullptr, FusedFrame{nullptr, nullptr}, n, m_stream, 0, tab);
        if (e != hipErrorInvalidValue || n <= 32) {
            hr = CheckHip(e, "k_convert_blocks");
            (void)hipEventRecord(m_evStop, m_stream);
            m_timed = true;
            return hr;
        }
    }
    // Interleaved RGB without a convert draw (m_PSConvColorData.bEnable false, :849-853): every frame is repacked into its own slot of a
    // batch texture (the reference's CopyFrame* upload: one repack launch per 32 frames, the sample pointers in its arguments) and ONE
    // k_fused_strip:surface launch resizes the whole chunk from there, instead of a repack + a resize launch per frame
    if (m_srcParams->layout == LAY_RGB && !m_plan.convert && m_stripSurf && m_plan.two_pass && !m_plan.hdr_tonemap && n > 1 &&
        !(m_cfg.flags & (MPCVR_FLAG_NO_FUSED | MPCVR_FLAG_NO_FAST_CONVERT | MPCVR_FLAG_NO_STRIP))) {
        const int tp = TexPitch();
        const size_t texBytes = (size_t)tp * m_srcHeight;
        const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)1 << 30) / std::max<size_t>(texBytes, 1)));
        const bool fresh = m_batchTex.size < texBytes * chunk || !m_batchTex.ptr || !m_batchTexZeroed;     // (not by size alone: a v210 batch or another media type may have used it since)
        if ((hr = CheckHip(m_batchTex.CheckCreate(texBytes * chunk), "batch source texture"))) return hr;
        const Surface cs{m_batchTex.ptr, tp, m_srcWidth, m_srcHeight, RgbTexFmt(*m_srcParams)};
        FusedStripParams ssp{};
        if (FillStripSurfParams(cs, MakeStore(dsts[0], rtPitch, m_plan.swap_fmt, true), &ssp)) {
            // texels the reference's copy loop never writes (RGB48 remainder) stay zero, as in PrepareSample
            if (fresh && (hr = CheckHip(hipMemsetAsync(m_batchTex.ptr, 0, texBytes * chunk, m_stream), "clear batch texture"))) return hr;
            m_batchTexZeroed = true;
            FrameSlot &slot = m_slots[m_slotNext];
            m_slotNext = (m_slotNext + 1) % kFrameSlots;
            if (!slot.done && (hr = CheckHip(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming), "slot event"))) return hr;
            if (slot.used && (hr = CheckHip(hipEventSynchronize(slot.done), "slot wait"))) return hr;
            if ((size_t)n > slot.cap) {
                if (slot.pinned) (void)hipHostFree(slot.pinned);
                slot.pinned = nullptr; slot.cap = 0;
                const size_t cap = n < 64 ? 64 : (size_t)n;
                if ((hr = CheckHip(hipHostMalloc(&slot.pinned, sizeof(FusedFrame) * cap, hipHostMallocDefault), "frames pinned"))) return hr;
                if ((hr = CheckHip(slot.dev.CheckCreate(sizeof(FusedFrame) * cap), "frames"))) return hr;
                slot.cap = cap;
            }
            FusedFrame *fr = (FusedFrame *)slot.pinned;
            for (int i = 0; i < n; i++) { fr[i].src = (const uint8_t *)srcs[i]; fr[i].dst = dsts[i]; }
            if ((hr = CheckHip(hipMemcpyAsync(slot.dev.ptr, fr, sizeof(FusedFrame) * n, hipMemcpyHostToDevice, m_stream), "frame table"))) return hr;
            bool aligned8 = true;
            for (int i = 0; i < n; i++)
                if (((uintptr_t)dsts[i] & 7) != 0) aligned8 = false;
            ssp.surf_stride = texBytes;
            ssp.fp.dst_aligned16 = aligned8 ? 1 : 0;
            if (!m_startRecorded) (void)hipEventRecord(m_evStart, m_stream);
            for (int at = 0; at < n && !hr; at += chunk) {
                const int m = std::min(chunk, n - at);
                hr = CheckHip(LaunchRepackRgb(m_srcParams->repack, nullptr, m_srcBottomUp ? -m_srcPitch : m_srcPitch, (uint8_t *)m_batchTex.ptr, tp,
                                              m_srcWidth, m_srcHeight, m_stream, srcs + at, m, texBytes), "k_repack_rgb");
                if (!hr) hr = CheckHip(LaunchFusedStrip(ssp, (const FusedFrame *)slot.dev.ptr + at, FusedFrame{nullptr, nullptr}, m, m_stream), "k_fused_strip<surface>");
            }
            (void)hipEventRecord(m_evStop, m_stream);
            (void)hipEventRecord(slot.done, m_stream);
            slot.used = true;
            m_timed = true;
            return hr;
        }
    }
    // HDR10 tone-mapping step behind the one-kernel strip path (what a single frame of this plan runs, ProcessOne): the strip kernel draws
    // every frame of a chunk into its slot of m_batchPost (a second frame table: same samples, the slots as targets) and ONE
    // k_hdr10_tonemap launch writes the render targets (:3359-3367)
    if (m_plan.hdr_tonemap && m_strip && !m_plan.fused_up2x && src4 && n > 1 && !(m_cfg.flags & (MPCVR_FLAG_NO_FUSED | MPCVR_FLAG_NO_FAST_CONVERT | MPCVR_FLAG_NO_STRIP))) {
        const int w2 = m_videoRect.Width(), h2 = m_videoRect.Height();
        const size_t postStride = PostStride();
        const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)4 << 30) / std::max<size_t>(postStride, 1)));
        if ((hr = CheckHip(m_batchPost.CheckCreate(postStride * chunk), "batch post-scale textures"))) return hr;
        const Surface post{m_batchPost.ptr, (int)(w2 * SurfBytesPerPixel(m_plan.internal_fmt)), w2, h2, m_plan.internal_fmt};
        const StoreParams last = MakeStore(post.ptr, post.pitch, m_plan.internal_fmt, false);
        FusedStripParams sp{};
        if (FillStripParams((const uint8_t *)srcs[0], post.ptr, post.pitch, last, &sp)) {
            sp.fp.dst_aligned16 = 1;                         // the slots start on 256-byte boundaries
            if (!m_startRecorded) (void)hipEventRecord(m_evStart, m_stream);
            for (int at = 0; at < n; at += chunk) {
                const int m = std::min(chunk, n - at);
                const FusedFrame *drawTab = nullptr, *realTab = nullptr;
                hipEvent_t d1 = nullptr, d2 = nullptr;
                if ((hr = UploadFrameTable(m, srcs + at, nullptr, (uint8_t *)m_batchPost.ptr, postStride, &drawTab, &d1))) return hr;
                if ((hr = UploadFrameTable(m, srcs + at, dsts + at, nullptr, 0, &realTab, &d2))) return hr;
                if ((hr = CheckHip(LaunchFusedStrip(sp, drawTab, FusedFrame{nullptr, nullptr}, m, m_stream), "k_fused_strip"))) return hr;
                ResizeBatch tb; tb.n = m; tb.in_stride = postStride; tb.frames = realTab;
                if ((hr = CheckHip(LaunchHdr10ToneMap(post, m_hdrTm, w2, h2, MakeStore(dsts[at], rtPitch, m_plan.swap_fmt, true), m_stream, &tb), "k_hdr10_tonemap"))) return hr;
                (void)hipEventRecord(d1, m_stream);
                (void)hipEventRecord(d2, m_stream);
            }
            (void)hipEventRecord(m_evStop, m_stream);
            m_timed = true;
            return MPCVR_S_OK;
        }
    }
    // (a batch of one needs no frame table: the frame travels in the kernel arguments, like mpcvr_process)
    if ((!m_plan.fused_up2x && !strip && !batchable) || !src4 || n == 1) {
        // samples that are repacked (or, not starting on a dword, copied) first share m_TexSrcVideo: those batches stay on the
        // context stream, frame by frame
        const bool repack = (m_srcParams->cformat == MPCVR_CF_V210 && !m_batchRepacked) || m_srcParams->layout == LAY_RGB || !src4;
        // MPCVR_BATCH_LANES=2..4 deals the frames to that many streams with private intermediates.  Measured on MI355X:
        // +5..10 % on the two-pass resize geometries, -15 % on 1080p same-size (fork/join events cost more than the
        // overlap returns), so one lane is the default.
        static const int want = [] { const char *e = std::getenv("MPCVR_BATCH_LANES"); return e ? std::atoi(e) : 1; }();
        const int lanes = (repack || n < 2 || want < 2 || m_dvFrames) ? 1 : std::min(std::min(n, want), (int)kLanes);
        if (lanes > 1 && (hr = PrepareLanes(lanes))) return hr;
        if (!m_startRecorded) (void)hipEventRecord(m_evStart, m_stream);
        if (lanes > 1) {
            if ((hr = CheckHip(hipEventRecord(m_fork, m_stream), "fork"))) return hr;
            for (int l = 1; l < lanes; l++)
                if ((hr = CheckHip(hipStreamWaitEvent(m_lanes[l].stream, m_fork, 0), "lane fork"))) return hr;
        }
        for (int i = 0; i < n && !hr; i++) {
            const uint8_t *tex;
            UseLane(i % lanes);
            if (m_batchRepacked) tex = (const uint8_t *)srcs[i];           // (already in m_TexSrcVideo's layout: a slot of the batch texture)
            else if ((hr = PrepareSample((const uint8_t *)srcs[i], &tex))) break;
            if (m_dvFrames && (hr = ApplyDoviFrame(m_dvFrames[i]))) break;          // this frame's RPU: constants, matrix, tone-mapping metadata
            hr = ProcessOne(tex, dsts[i], rtPitch);
        }
        UseLane(0);
        for (int l = 1; l < lanes; l++) {        // join, also on the error path: the context stream stays the only handle
            (void)hipEventRecord(m_lanes[l].done, m_lanes[l].stream);
            (void)hipStreamWaitEvent(m_stream, m_lanes[l].done, 0);
        }
        if (hr) return hr;
        (void)hipEventRecord(m_evStop, m_stream);
        m_timed = true;
        return MPCVR_S_OK;
    }
    // one launch for the whole batch
    // The frame table travels through a small ring of pinned/device slots so the host can queue several
    // batches ahead; a slot is reused only after the launch that read it has completed (its event).
    FrameSlot &slot = m_slots[m_slotNext];
    m_slotNext = (m_slotNext + 1) % kFrameSlots;
    if (!slot.done && (hr = CheckHip(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming), "slot event"))) return hr;
    if (slot.used && (hr = CheckHip(hipEventSynchronize(slot.done), "slot wait"))) return hr;
    if ((size_t)n > slot.cap) {
        if (slot.pinned) (void)hipHostFree(slot.pinned);
        slot.pinned = nullptr; slot.cap = 0;
        const size_t cap = n < 64 ? 64 : (size_t)n;
        if ((hr = CheckHip(hipHostMalloc(&slot.pinned, sizeof(FusedFrame) * cap, hipHostMallocDefault), "frames pinned"))) return hr;
        if ((hr = CheckHip(slot.dev.CheckCreate(sizeof(FusedFrame) * cap), "frames"))) return hr;
        slot.cap = cap;
    }
    FusedFrame *fr = (FusedFrame *)slot.pinned;
    for (int i = 0; i < n; i++) { fr[i].src = (const uint8_t *)srcs[i]; fr[i].dst = dsts[i]; }
    if ((hr = CheckHip(hipMemcpyAsync(slot.dev.ptr, fr, sizeof(FusedFrame) * n, hipMemcpyHostToDevice, m_stream), "frame table"))) return hr;
    if (batchable) {
        if (!m_startRecorded) (void)hipEventRecord(m_evStart, m_stream);
        hr = ProcessBatchLaunches(n, (const FusedFrame *)slot.dev.ptr, (const uint8_t *)srcs[0], dsts[0], rtPitch, aligned);
        (void)hipEventRecord(m_evStop, m_stream);
        (void)hipEventRecord(slot.done, m_stream);
        slot.used = true;
        m_timed = true;
        return hr;
    }
    if (strip) {
        bool aligned8 = true;
        for (int i = 0; i < n; i++)
            if (((uintptr_t)dsts[i] & 7) != 0) aligned8 = false;
        strip_sp.fp.dst_aligned16 = aligned8 ? 1 : 0;
        if (!m_startRecorded) (void)hipEventRecord(m_evStart, m_stream);
        hr = CheckHip(LaunchFusedStrip(strip_sp, (const FusedFrame *)slot.dev.ptr, FusedFrame{nullptr, nullptr}, n, m_stream), "k_fused_strip");
        (void)hipEventRecord(m_evStop, m_stream);
        (void)hipEventRecord(slot.done, m_stream);
        slot.used = true;
        m_timed = true;
        return hr;
    }
    FusedParams fp{};
    FillFusedParams((const uint8_t *)srcs[0], nullptr, rtPitch, &fp);
    fp.dst_aligned16 = aligned ? 1 : 0;
    if (!m_startRecorded) (void)hipEventRecord(m_evStart, m_stream);
    hr = CheckHip(LaunchFusedUp2x(fp, (const FusedFrame *)slot.dev.ptr, FusedFrame{nullptr, nullptr}, n, m_stream), "k_fused_up2x");
    (void)hipEventRecord(m_evStop, m_stream);
    (void)hipEventRecord(slot.done, m_stream);
    slot.used = true;
    m_timed = true;
    return hr;
}

// ---- EXTENSION: error-diffusion final pass (bUseDither = 2; no reference counterpart, include/mpcvr.h) -----------------------------
// The plan of such a context is the 10-bit swap chain's (DecidePlan: swap_fmt = SF_RGB10A2, no final pass for UNORM internal formats),
// so every kernel of the library — fused, batched, Dolby Vision — runs as it does for a 10-bit target; only the target differs: a
// window-sized intermediate per frame, from which k_error_diffusion writes the B8G8R8A8 render target inside video rect ∩ window.
HRESULT CHipVideoProcessor::PrepareErrDiff(int frames)
{
    m_edPitch = (m_windowRect.Width() * 4 + 255) & ~255;
    m_edStride = (size_t)m_edPitch * (size_t)m_windowRect.Height();
    // (256 bytes in front and behind: the pass reads its rows in 16-byte pieces that may start two pixels in front of a row and end three
    // behind it — inside the image that is the neighbouring row's padding, at its two ends it is this margin)
    const HRESULT hr = CheckHip(m_edPost.CheckCreate(m_edStride * (size_t)frames + 512), "error-diffusion intermediates");
    m_edBase = hr ? nullptr : (uint8_t *)m_edPost.ptr + 256;
    return hr;
}

HRESULT CHipVideoProcessor::ErrDiffPass(int n, const FusedFrame *table, FusedFrame single, void *const *dsts, int rtPitch, hipStream_t s)
{
    ErrDiffParams P{};
    P.x0 = std::max((int)m_videoRect.left, 0); P.y0 = std::max((int)m_videoRect.top, 0);
    P.x1 = std::min((int)m_videoRect.right, m_windowRect.Width()); P.y1 = std::min((int)m_videoRect.bottom, m_windowRect.Height());
    if (P.x1 <= P.x0 || P.y1 <= P.y0) return MPCVR_S_OK;          // the video rect lies outside the window: nothing is drawn
    P.src_pitch = m_edPitch; P.dst_pitch = rtPitch;
    (void)dsts;
    // band-major ticket order: same box, 32 frames 4K -> 8K: 3.85 k frames/s against 3.28 k frame-major (profiles/r04/ab_call24_errdiff_order.jsonl)
    static const int order = [] { const char *e = std::getenv("MPCVR_ERRDIFF_ORDER"); return e ? std::atoi(e) : 1; }();
    P.order = order;
    // (tests: a band that never publishes and a short patience, read per call — the give-up path must end in an error, not in a hang)
    const char *stall = std::getenv("MPCVR_ERRDIFF_TEST_STALL"), *spin = std::getenv("MPCVR_ERRDIFF_SPIN");
    P.test_stall = stall && *stall && *stall != '0' ? 1 : 0;
    P.spin_limit = spin && *spin ? std::atoi(spin) : m_edPatience;      // (0: the launcher's default, 2^21 polls of about a microsecond)
    HRESULT hr;
    if (!m_edStatus) {
        if ((hr = CheckHip(hipHostMalloc((void **)&m_edStatus, sizeof(int), hipHostMallocDefault), "error-diffusion status word"))) return hr;
        *m_edStatus = 0;
    }
    if (*m_edStatus) { *m_edStatus = 0; return Fail(MPCVR_E_FAIL, "error diffusion: a band of an earlier pass gave up waiting for the band above"); }
    // the hand-off rows are cleared and rewritten by every launch: launches of one context run in stream order on one buffer
    if (s != m_stream) (void)hipStreamSynchronize(m_stream);
    const size_t need = ErrorDiffusionHandoffBytes(P, n);
    if ((hr = CheckHip(m_edHandoff.CheckCreate(need), "error-diffusion hand-off rows"))) return hr;
    P.handoff = (uint32_t *)m_edHandoff.ptr; P.status = m_edStatus;
    // the hand-off words carry the launch's generation: rows of the same layout need no clearing from launch to launch (600 MB for a 32-frame
    // batch of 8K frames); another layout, another buffer or a wrapped count: gen = 0 = the launcher clears them and starts at 1
    // (the layout is compared field by field — round 5 hashed it into 64 bits, and two layouts whose hashes met would have shared uncleared rows)
    const EdLayout key{P.x0, P.x1, P.y0, P.y1, n, (const void *)P.handoff};
    if (!(key == m_edKey) || m_edGen >= 4095 || m_edGen <= 0 || P.test_stall) { P.gen = 0; m_edGen = 1; m_edKey = P.test_stall ? EdLayout{} : key; }
    else P.gen = ++m_edGen;
    return CheckHip(LaunchErrorDiffusion(P, table, single, n, s), "k_error_diffusion");
}

HRESULT CHipVideoProcessor::ProcessBatchErrDiff(int n, const void *const *srcs, void *const *dsts, int rtPitch)
{
    if (rtPitch < m_windowRect.Width() * 4) return Fail(MPCVR_E_INVALIDARG, "render-target pitch smaller than a row");
    for (int i = 0; i < n; i++)
        if (!srcs[i] || !dsts[i]) return Fail(MPCVR_E_POINTER, "null frame in batch");
    (void)hipSetDevice(m_device);
    HRESULT hr;
    const size_t one = ((size_t)((m_windowRect.Width() * 4 + 255) & ~255)) * (size_t)m_windowRect.Height();
    // intermediates for up to ~4 GiB of frames at a time, in chunks of equal size: the pass is a chain of dependent steps per frame and only
    // many frames side by side fill the chip (a 33-frame batch as 32 + 1 took 13.8 ms where 32 take 8.5: the odd frame ran alone)
    static const int cap = [] { const char *e = std::getenv("MPCVR_ERRDIFF_CHUNK"); return e ? std::atoi(e) : 0; }();      // (tests: chunks of a few small frames)
    int most = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)4 << 30) / std::max<size_t>(one, 1)));
    if (cap > 0) most = std::min(most, cap);
    const int chunks = (n + most - 1) / most;
    const int chunk = (n + chunks - 1) / chunks;
    if ((hr = PrepareErrDiff(chunk))) return hr;
    bool usedTables = false;         // (ProcessBatchDovi reports per run whether the per-frame tables were read)
    std::vector<void *> mids(chunk);
    for (int i = 0; i < chunk; i++) mids[i] = m_edBase + (size_t)i * m_edStride;
    // the batch's process time runs from in front of the first chunk to behind the last chunk's pass
    (void)JoinFrameLanes(false);
    (void)hipEventRecord(m_evStart, m_stream);
    m_startRecorded = true; m_keepStart = true;
    struct KeepStartGuard { bool &f; ~KeepStartGuard() { f = false; } } keepGuard{m_keepStart};
    for (int at = 0; at < n; at += chunk) {
        const int m = std::min(chunk, n - at);
        // the whole-batch routes of the 10-bit plan, into the intermediates (the previous chunk's pass reads them in stream order); a run of
        // ProcessBatchDovi hands its per-frame RPU state over by frame index: the chunk sees its own slice
        const DoviFrameState *const dvFrames = m_dvFrames;
        const DoviParams *const dvTab = m_dvTabReady;
        const float *const dvCm = m_dvCmReady;
        if (m_dvFrames) m_dvFrames += at;
        if (m_dvTabReady) { m_dvTabReady += at; m_dvCmReady += (size_t)12 * at; }
        if (m_dvFrames) { m_dvTabDev = nullptr; m_dvCmDev = nullptr; }      // (a chunk that goes frame by frame must not read the tables the chunk before it took)
        hr = ProcessBatchRoutes(m, srcs + at, mids.data(), m_edPitch);
        m_dvFrames = dvFrames; m_dvTabReady = dvTab; m_dvCmReady = dvCm;
        if (hr) return hr;
        usedTables = usedTables || m_dvTabDev != nullptr;
        const FusedFrame *tab = nullptr;
        hipEvent_t done = nullptr;
        if ((hr = UploadFrameTable(m, (const void *const *)mids.data(), dsts + at, nullptr, 0, &tab, &done))) return hr;
        hr = ErrDiffPass(m, tab, FusedFrame{nullptr, nullptr}, dsts + at, rtPitch, m_stream);
        (void)hipEventRecord(done, m_stream);
        if (hr) return hr;
    }
    if (m_dvFrames && usedTables && !m_dvTabDev) { m_dvTabDev = m_dvTabReady; m_dvCmDev = m_dvCmReady; }
    (void)hipEventRecord(m_evStop, m_stream);       // (the batch's process time includes the pass)
    m_timed = true;
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::UploadFrameTable(int n, const void *const *srcs, void *const *dsts, uint8_t *dst_base, size_t dst_stride, const FusedFrame **dev, hipEvent_t *done)
{
    HRESULT hr;
    FrameSlot &slot = m_slots[m_slotNext];
    m_slotNext = (m_slotNext + 1) % kFrameSlots;
    if (!slot.done && (hr = CheckHip(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming), "slot event"))) return hr;
    if (slot.used && (hr = CheckHip(hipEventSynchronize(slot.done), "slot wait"))) return hr;
    if ((size_t)n > slot.cap) {
        if (slot.pinned) (void)hipHostFree(slot.pinned);
        slot.pinned = nullptr; slot.cap = 0;
        const size_t cap = n < 64 ? 64 : (size_t)n;
        if ((hr = CheckHip(hipHostMalloc(&slot.pinned, sizeof(FusedFrame) * cap, hipHostMallocDefault), "frames pinned"))) return hr;
        if ((hr = CheckHip(slot.dev.CheckCreate(sizeof(FusedFrame) * cap), "frames"))) return hr;
        slot.cap = cap;
    }
    FusedFrame *fr = (FusedFrame *)slot.pinned;
    for (int i = 0; i < n; i++) { fr[i].src = srcs ? (const uint8_t *)srcs[i] : nullptr; fr[i].dst = dsts ? dsts[i] : (void *)(dst_base + (size_t)i * dst_stride); }
    if ((hr = CheckHip(hipMemcpyAsync(slot.dev.ptr, fr, sizeof(FusedFrame) * n, hipMemcpyHostToDevice, m_stream), "frame table"))) return hr;
    slot.used = true;        // the caller records *done behind the last launch that reads the table
    *dev = (const FusedFrame *)slot.dev.ptr;
    *done = slot.done;
    return MPCVR_S_OK;
}

// Can this plan run as whole-batch launches?  *conv: the block convert into the (batched) convert output; *direct: the block
// convert straight into the render targets (same-size frames).  Exactly one of them is used.
bool CHipVideoProcessor::BatchPlan(const uint8_t *sample0, void *rt0, int rtPitch, bool aligned, FusedParams *conv, FusedParams *direct) const
{
    // every draw kernel has a frame dimension (round 4: the one-kernel-fits-all k_resize / k_jinc2 too — quarter turns, flips outside the
    // strip kernels' reach, the two-draw Jinc2m), so what decides is the convert stage: the 2x2-block kernel must take the sample
    if (!m_plan.convert) return false;
    if ((m_srcParams->cformat == MPCVR_CF_V210 && !m_batchRepacked) || m_srcParams->layout == LAY_RGB) return false;
    const int w1 = m_srcRectWidth, h1 = m_srcRectHeight, w2 = m_videoRect.Width();
    if (m_plan.direct_convert) {
        FillFusedParams(sample0, rt0, rtPitch, direct);
        direct->dst_aligned16 = aligned ? 1 : 0;
        direct->src_aligned16 = m_batchSrc16 ? 1 : 0;
        return ConvertBlocksSupported(*direct, true);
    }
    // with the HDR10 tone-mapping step (:3359-3367) the draws go into the frames' post-scale textures (m_batchPost, internal format)
    // and one tone-mapping launch writes the render targets; without a resize the step reads the convert outputs
    const bool hdr = m_plan.hdr_tonemap;
    if (!m_plan.two_pass && !m_plan.one_pass && !hdr) return false;
    const int convPitch = (int)(w1 * SurfBytesPerPixel(m_plan.internal_fmt));
    FillFusedParams(sample0, m_batchConv.ptr, convPitch, conv);
    conv->store = MakeStore(m_batchConv.ptr, convPitch, m_plan.internal_fmt, false);
    conv->dst_aligned16 = 1;
    conv->src_aligned16 = m_batchSrc16 ? 1 : 0;
    conv->exact_convert = 1;
    if (!ConvertBlocksSupported(*conv, false)) return false;
    const Surface cs{nullptr, convPitch, w1, h1, m_plan.internal_fmt};
    const StoreParams final = hdr ? MakeStore((void *)(uintptr_t)4096, (int)(w2 * SurfBytesPerPixel(m_plan.internal_fmt)), m_plan.internal_fmt, false)
                                  : MakeStore(rt0, rtPitch, m_plan.swap_fmt, true);
    (void)cs; (void)final; (void)w2;
    return true;            // the draws: ProcessBatchLaunches picks the kernel per chunk exactly as ResizeShaderPass does per frame
}

// convert all -> first draw all -> second draw all, a frame dimension in every grid; the intermediates hold `chunk` frames
HRESULT CHipVideoProcessor::ProcessBatchLaunches(int n, const FusedFrame *table, const uint8_t *sample0, void *rt0, int rtPitch, bool aligned)
{
    HRESULT hr;
    const int w1 = m_srcRectWidth, h1 = m_srcRectHeight, w2 = m_videoRect.Width(), h2 = m_videoRect.Height();
    FusedParams conv{}, direct{};
    if (m_plan.direct_convert) {
        if (!BatchPlan(sample0, rt0, rtPitch, aligned, &conv, &direct)) return Fail(MPCVR_E_UNEXPECTED, "batch plan changed");
        return CheckHip(LaunchConvertBlocks(direct, table, FusedFrame{nullptr, nullptr}, n, m_stream), "k_convert_blocks");
    }
    // intermediates for up to `chunk` frames (at most ~4 GiB)
    const bool hdr = m_plan.hdr_tonemap;
    const size_t postStride = hdr ? PostStride() : 0;
    const size_t per = m_convBytes + m_midBytes + postStride;
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)4 << 30) / std::max<size_t>(per, 1)));
    if ((hr = CheckHip(m_batchConv.CheckCreate(m_convBytes * chunk), "batch convert output"))) return hr;
    if (m_midBytes && (hr = CheckHip(m_batchMid.CheckCreate(m_midBytes * chunk), "batch resize texture"))) return hr;
    if (hdr && (hr = CheckHip(m_batchPost.CheckCreate(postStride * chunk), "batch post-scale textures"))) return hr;
    if (!BatchPlan(sample0, rt0, rtPitch, aligned, &conv, &direct)) return Fail(MPCVR_E_UNEXPECTED, "batch plan changed");
    const int convPitch = (int)(w1 * SurfBytesPerPixel(m_plan.internal_fmt));
    const Surface cs{m_batchConv.ptr, convPitch, w1, h1, m_plan.internal_fmt};
    const StoreParams target = MakeStore(rt0, rtPitch, m_plan.swap_fmt, true);
    // HDR10 tone-mapping step: the draws write frame z's post-scale texture (a second frame table whose targets are the slots of
    // m_batchPost), then ONE k_hdr10_tonemap launch per chunk writes the render targets (:3359-3367)
    const Surface post{m_batchPost.ptr, (int)(w2 * SurfBytesPerPixel(m_plan.internal_fmt)), w2, h2, m_plan.internal_fmt};
    const StoreParams final = hdr ? MakeStore(post.ptr, post.pitch, m_plan.internal_fmt, false) : target;
    const FusedFrame *postTab = nullptr;
    hipEvent_t postDone = nullptr;
    if (hdr) {
        if ((hr = UploadFrameTable(chunk, nullptr, nullptr, (uint8_t *)m_batchPost.ptr, postStride, &postTab, &postDone))) return hr;
        aligned = true;             // the slots of m_batchPost start on 256-byte boundaries
    }
    const bool drawn = m_plan.two_pass || m_plan.one_pass;
    for (int at = 0; at < n; at += chunk) {
        const int m = std::min(chunk, n - at);
        const FusedFrame *tab = hdr ? postTab : table + at;
        // frame z of the chunk: sample from the table, output at m_batchConv + z * m_convBytes
        conv.store.dst = m_batchConv.ptr;
        if (m_dvTabDev) { conv.conv.dovi = m_dvTabDev + at; conv.dovi_cm = m_dvCmDev + (size_t)12 * at; }       // (the chunk's slice of the per-frame RPU tables)
        if ((hr = CheckHip(LaunchConvertBlocks(conv, table + at, FusedFrame{nullptr, nullptr}, m, m_stream, m_convBytes), "k_convert_blocks"))) return hr;
        // the draws, kernel by kernel as ResizeShaderPass picks them for one frame (default tier), each with the chunk as its frame dimension
        ResizeBatch b1; b1.n = m; b1.in_stride = m_convBytes;
        FusedStripParams ssp{};
        if (m_stripSurf && FillStripSurfParams(cs, final, &ssp)) {
            ssp.surf_stride = m_convBytes;
            ssp.fp.dst_aligned16 = aligned ? 1 : 0;
            if ((hr = CheckHip(LaunchFusedStrip(ssp, tab, FusedFrame{nullptr, nullptr}, m, m_stream), "k_fused_strip<surface>"))) return hr;
        } else if (m_plan.two_pass && !m_firstJinc && !m_secondJinc && m_firstAxis == 0 && !m_firstSwap && Resize2DSupported(cs, m_tapsX, m_tapsY, final)) {
            b1.frames = tab;
            if ((hr = CheckHip(LaunchResize2D(cs, m_tapsX, m_tapsY, (const int32_t *)m_otherX.ptr, m_plan.mid_h, w2, h2, final, m_stream, &b1), "k_resize_2d"))) return hr;
        } else if (m_plan.two_pass) {
            const Surface mid{m_batchMid.ptr, w2 * 8, w2, m_plan.mid_h, SF_RGBA16F};
            b1.dst_stride = m_midBytes;
            const StoreParams st = MakeStore(mid.ptr, mid.pitch, SF_RGBA16F, false);
            if (m_firstJinc) hr = CheckHip(LaunchJinc2(cs, m_firstCoords, w2, m_plan.mid_h, st, m_stream, m_jincFirstTab, true, &b1, m_jincFirstCtr), "k_jinc2");
            else hr = CheckHip(LaunchResize(m_firstAxis, m_firstSwap, cs, m_tapsX, (const int32_t *)m_otherX.ptr, w2, m_plan.mid_h, st, m_stream, false, &b1), "k_resize<first>");
            if (hr) return hr;
            ResizeBatch b2; b2.n = m; b2.in_stride = m_midBytes; b2.frames = tab; b2.dst_aligned8 = aligned ? 1 : 0;
            if (m_secondJinc) hr = CheckHip(LaunchJinc2(mid, m_secondCoords, w2, h2, final, m_stream, m_jincSecondTab, true, &b2, m_jincSecondCtr), "k_jinc2");
            else hr = CheckHip(LaunchResize(1, false, mid, m_tapsY, (const int32_t *)m_otherY.ptr, w2, h2, final, m_stream, false, &b2), "k_resize<Y>");
            if (hr) return hr;
        } else if (m_firstJinc) {
            b1.frames = tab; b1.dst_aligned8 = aligned ? 1 : 0;
            if ((hr = CheckHip(LaunchJinc2(cs, m_firstCoords, w2, h2, final, m_stream, m_jincFirstTab, true, &b1, m_jincFirstCtr), "k_jinc2"))) return hr;
        } else if (drawn) {
            b1.frames = tab;
            if ((hr = CheckHip(LaunchResize(m_firstAxis, m_firstSwap, cs, m_tapsX, (const int32_t *)m_otherX.ptr, w2, h2, final, m_stream, false, &b1), "k_resize<one>"))) return hr;
        }
        if (hdr) {
            ResizeBatch tb; tb.n = m; tb.in_stride = drawn ? postStride : m_convBytes; tb.frames = table + at;
            if ((hr = CheckHip(LaunchHdr10ToneMap(drawn ? post : cs, m_hdrTm, w2, h2, target, m_stream, &tb), "k_hdr10_tonemap"))) return hr;
        }
    }
    if (postDone) (void)hipEventRecord(postDone, m_stream);
    return MPCVR_S_OK;
}

// ---- a batch with one Dolby Vision RPU per frame -------------------------------------------------------------------------------
// The reference reads the RPU of every sample in CopySample (IID_MediaSideDataDOVIMetadataV2, :2270-2520) and rebuilds the constant
// buffers when it differs from the previous one.  Here: rpus[i] is applied in front of frame i exactly as SetDoviMetadata would
// (level-1 / level-2 blocks stay as last seen until Flush), the frames are cut into runs that share a plan and a kernel variant
// (level-2 trims present or not), and a run goes through ProcessBatch — ONE launch per stage where the block convert's Dolby Vision
// variants take it (they index a table of DoviParams and colour matrices by the frame), frame by frame with the RPU's constants
// uploaded in stream order otherwise.  The context is left as after the last frame's SetDoviMetadata.
void CHipVideoProcessor::SaveDoviWalk(DoviWalkState *s) const
{
    s->valid = m_doviValid; s->l1Present = m_doviL1Present; s->l2Present = m_doviL2Present; s->blobOverride = m_blobOverride; s->planDirty = m_planDirty;
    s->md = m_doviMd; s->host = m_doviHost;
    std::memcpy(s->l1, m_doviL1, sizeof(m_doviL1)); std::memcpy(s->l2raw, m_doviL2Raw, sizeof(m_doviL2Raw)); std::memcpy(s->cm, m_cm, sizeof(m_cm));
    s->tail = m_tail; s->gamma = m_gamma; s->tm = m_hdrTm;
}
void CHipVideoProcessor::RestoreDoviWalk(const DoviWalkState &s)
{
    m_doviValid = s.valid; m_doviL1Present = s.l1Present; m_doviL2Present = s.l2Present; m_blobOverride = s.blobOverride; m_planDirty = s.planDirty;
    m_doviMd = s.md; m_doviHost = s.host;
    std::memcpy(m_doviL1, s.l1, sizeof(m_doviL1)); std::memcpy(m_doviL2Raw, s.l2raw, sizeof(m_doviL2Raw)); std::memcpy(m_cm, s.cm, sizeof(m_cm));
    m_tail = s.tail; m_gamma = s.gamma; m_hdrTm = s.tm;
}

HRESULT CHipVideoProcessor::ApplyDoviFrame(const DoviFrameState &f)
{
    m_doviHost = f.p;
    std::memcpy(m_cm, f.cm, sizeof(m_cm));
    m_hdrTm = f.tm;
    return UploadDoviParams();          // through the pinned ring, in stream order: the frames before this one read their own copy
}

// DoviParams[n] followed by cm[12 n], staged through one of two pinned / device slots (a slot is rewritten only after the launches
// that read it have completed: `done`, recorded by ProcessBatchDovi behind the run)
HRESULT CHipVideoProcessor::UploadDoviTables(int n, hipEvent_t *done)
{
    HRESULT hr;
    DoviTableSlot &slot = m_dvSlots[m_dvSlotNext++ % 2];
    if (!slot.done && (hr = CheckHip(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming), "dovi table event"))) return hr;
    if (slot.used && (hr = CheckHip(hipEventSynchronize(slot.done), "dovi table wait"))) return hr;
    const size_t need = (size_t)n * (sizeof(DoviParams) + 12 * sizeof(float));
    if (need > slot.cap) {
        if (slot.pinned) (void)hipHostFree(slot.pinned);
        slot.pinned = nullptr; slot.cap = 0;
        const size_t cap = std::max<size_t>(need, 64 * (sizeof(DoviParams) + 12 * sizeof(float)));
        if ((hr = CheckHip(hipHostMalloc(&slot.pinned, cap, hipHostMallocDefault), "dovi tables pinned"))) return hr;
        if ((hr = CheckHip(slot.dev.CheckCreate(cap), "dovi tables"))) return hr;
        slot.cap = cap;
    }
    DoviParams *tp = (DoviParams *)slot.pinned;
    float *tc = (float *)(tp + n);
    for (int i = 0; i < n; i++) {
        tp[i] = m_dvFrames[i].p;
        std::memcpy(tc + (size_t)12 * i, m_dvFrames[i].cm, 12 * sizeof(float));
    }
    if ((hr = CheckHip(hipMemcpyAsync(slot.dev.ptr, slot.pinned, need, hipMemcpyHostToDevice, m_stream), "dovi tables upload"))) return hr;
    m_dvTabReady = (const DoviParams *)slot.dev.ptr;
    m_dvCmReady = (const float *)((const DoviParams *)slot.dev.ptr + n);
    slot.used = true;
    *done = slot.done;
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::ProcessBatchDovi(int n, const void *const *srcs, void *const *dsts, int rtPitch, const mpcvr_dovi_metadata *rpus)
{
    if (!m_bInit || !m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    if (n <= 0 || !srcs || !dsts || !rpus) return Fail(MPCVR_E_INVALIDARG, "empty batch");
    for (int i = 0; i < n; i++)         // all or nothing: no frame is drawn when one RPU of the batch is malformed
        if (!CheckDoviCurves(rpus[i])) return Fail(MPCVR_E_INVALIDARG, "Dolby Vision curves: num_pivots outside [2,9], mapping_idc > 1 or more than 32 level-2 blocks");
    (void)hipSetDevice(m_device);
    HRESULT hr = MPCVR_S_OK;
    std::vector<DoviFrameState> fs((size_t)n);
    m_dvLastInfo.clear();
    int launches = 0;
    auto collect = [&](int i) { fs[i].p = m_doviHost; std::memcpy(fs[i].cm, m_cm, sizeof(m_cm)); fs[i].tm = m_hdrTm; };
    DoviWalkState back;
    for (int i = 0; i < n && !hr;) {
        // frame i opens a run: its RPU may change the plan (the first RPU of a stream, level-1 data switching the tone mapping on)
        if ((hr = ApplyDoviMetadata(&rpus[i]))) break;
        if (m_planDirty && (hr = UpdatePlan())) break;
        collect(i);
        if ((hr = UploadDoviParams())) break;           // (the context's own copy and, at the first RPU of a stream, the PQ EOTF table exist from here on)
        int j = i + 1;
        for (; j < n; j++) {
            SaveDoviWalk(&back);
            if ((hr = ApplyDoviMetadata(&rpus[j]))) break;
            if (m_planDirty || m_doviHost.l2_enabled != fs[i].p.l2_enabled) { RestoreDoviWalk(back); break; }      // frame j opens the next run
            collect(j);
        }
        if (hr) break;
        const int len = j - i;
        m_dvFrames = fs.data() + i; m_dvCount = len;
        m_dvTabReady = nullptr; m_dvCmReady = nullptr; m_dvTabDev = nullptr; m_dvCmDev = nullptr;
        hipEvent_t done = nullptr;
        if (len > 1 && !m_plan.hdr_tonemap) hr = UploadDoviTables(len, &done);
        if (!hr) hr = ProcessBatch(len, srcs + i, dsts + i, rtPitch);
        const bool tables = m_dvTabDev != nullptr;
        launches += m_lastBatchLaunches;
        m_dvLastInfo += (m_dvLastInfo.empty() ? "" : ",") + std::to_string(len) + (tables ? ":tables" : ":frames");
        m_dvFrames = nullptr; m_dvCount = 0;
        m_dvTabReady = nullptr; m_dvCmReady = nullptr; m_dvTabDev = nullptr; m_dvCmDev = nullptr;
        if (done) (void)hipEventRecord(done, m_stream);
        // the context's own copy of the constants: the run's last frame (a whole-batch route did not touch it)
        if (!hr && tables) hr = UploadDoviParams();
        i = j;
    }
    m_lastBatchFrames = n; m_lastBatchLaunches = launches;
    return hr;
}

// how the last mpcvr_process_batch[_dovi] call ran: "frames=<n>;launches=<kernel launches>[;dovi_runs=<frames>:<tables|frames>,...]" —
// a batch on a whole-batch route launches a handful of kernels whatever n is, a frame-by-frame one at least n
std::string CHipVideoProcessor::GetLastBatchInfo() const
{
    std::string s = "frames=" + std::to_string(m_lastBatchFrames) + ";launches=" + std::to_string(m_lastBatchLaunches) + ";lane=" + std::to_string(m_lastBatchLane);
    if (!m_dvLastInfo.empty()) s += ";dovi_runs=" + m_dvLastInfo;
    return s;
}

// Render minus Present — DX11VideoProcessor.cpp:2599-2813
HRESULT CHipVideoProcessor::Render(int /*field*/)
{
    if (!m_bInit || !m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    if (!m_curSample) return MPCVR_S_FALSE;          // nothing to draw (cf. :2603-2606)
    (void)hipSetDevice(m_device);
    const int w = m_windowRect.Width(), h = m_windowRect.Height();
    const size_t bytes = (size_t)w * 4 * h;
    HRESULT hr;
    const bool fresh = m_BackBuffer.size < bytes || !m_BackBuffer.ptr;
    if ((hr = CheckHip(m_BackBuffer.CheckCreate(bytes), "back buffer"))) return hr;
    // ClearRenderTargetView to black (:2622) — only the letterbox area survives Process
    m_clearOnRun = (fresh || m_videoRect != CRect(0, 0, w, h)) ? bytes : 0;     // queued by Process on the stream the frame runs on
    hr = Process(m_BackBuffer.ptr, w * 4, nullptr, nullptr, false);
    m_clearOnRun = 0;
    if (hr >= 0) { m_backW = w; m_backH = h; m_backFmt = m_cfg.output_format; }        // what GetDisplayedImage will find there
    return hr;
}

HRESULT CHipVideoProcessor::GetBackBuffer(void **ptr, int *pitch, int *w, int *h)
{
    if (!m_BackBuffer.ptr) return Fail(MPCVR_E_NOT_VALID_STATE, "Render has not been called");
    if (ptr) *ptr = m_BackBuffer.ptr;
    if (pitch) *pitch = m_windowRect.Width() * 4;
    if (w) *w = m_windowRect.Width();
    if (h) *h = m_windowRect.Height();
    return MPCVR_S_OK;
}

// GetDisplayedImage — DX11VideoProcessor.cpp:3610-3683: the back buffer as it was last rendered (no new draw), copied to host memory as the
// pixels of a top-down DIB: B8G8R8A8 as it is (CopyPlaneAsIs); R10G10B10A2 as BGR32 through ConvertR10G10B10A2toBGR32 (Helper.cpp:805-834:
// the top eight bits of each channel, X = 0xff) or, with m_bAllowDeepColorBitmaps, as BGR48 (ConvertR10G10B10A2toBGR48, :836-857: the ten
// bits in the top of each 16-bit word).  Rows are CalcDibRowPitch(width, bits) apart.  The BITMAPINFOHEADER and the LocalAlloc block the
// reference puts around the pixels are the caller's (the adapter's): host_pixels == NULL reports the size and the header's fields.
HRESULT CHipVideoProcessor::GetDisplayedImage(void *hostPixels, size_t *size, bool deepColor, int *width, int *height, int *bits)
{
    if (!size) return Fail(MPCVR_E_POINTER, "null size");
    if (!m_BackBuffer.ptr || m_backW <= 0 || m_backH <= 0) return Fail(MPCVR_E_NOT_VALID_STATE, "Render has not been called");     // (E_ABORT without a swap chain, :3612-3614)
    const int w = m_backW, h = m_backH;
    const bool ten = m_backFmt == MPCVR_OUT_RGB10A2;
    const int bpp = (ten && deepColor) ? 48 : 32;
    const size_t dibPitch = (((size_t)w * bpp + 31) & ~(size_t)31) / 8, need = dibPitch * h;      // CalcDibRowPitch
    if (width) *width = w;
    if (height) *height = h;
    if (bits) *bits = bpp;
    if (!hostPixels) { *size = need; return MPCVR_S_OK; }
    if (*size < need) { *size = need; return Fail(MPCVR_E_INVALIDARG, "buffer too small"); }
    HRESULT hr = Synchronize();                     // frames still on the lanes / the context stream write the buffer
    if (hr) return hr;
    const size_t srcPitch = (size_t)w * 4;
    if (!ten) {                                     // 32 bits per pixel: the DIB pitch is the back buffer's
        if ((hr = CheckHip(hipMemcpy(hostPixels, m_BackBuffer.ptr, need, hipMemcpyDeviceToHost), "displayed image read-back"))) return hr;
    } else {
        std::vector<uint32_t> staging((size_t)w * h);
        if ((hr = CheckHip(hipMemcpy(staging.data(), m_BackBuffer.ptr, srcPitch * h, hipMemcpyDeviceToHost), "displayed image read-back"))) return hr;
        for (int y = 0; y < h; y++) {
            const uint32_t *src = staging.data() + (size_t)y * w;
            uint8_t *row = (uint8_t *)hostPixels + (size_t)y * dibPitch;
            if (bpp == 32) {
                uint32_t *d = (uint32_t *)row;
                for (int x = 0; x < w; x++) {
                    const uint32_t t = src[x];
                    d[x] = ((t & 0x3fc00000u) >> 22) | ((t & 0x000ff000u) >> 4) | ((t & 0x000003fcu) << 14) | 0xff000000u;
                }
            } else {
                uint16_t *d = (uint16_t *)row;
                for (int x = 0; x < w; x++) {
                    const uint32_t t = src[x];
                    *d++ = (uint16_t)((t & 0x3ff00000u) >> 14); *d++ = (uint16_t)((t & 0x000ffc00u) >> 4); *d++ = (uint16_t)((t & 0x000003ffu) << 6);
                }
            }
        }
    }
    *size = need;
    return MPCVR_S_OK;
}

// GetCurentImage — DX11VideoProcessor.cpp:3493-3608
HRESULT CHipVideoProcessor::GetCurentImage(void *hostBGRA, size_t *size)
{
    if (!size) return Fail(MPCVR_E_POINTER, "null size");
    if (!m_bInit || !m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    int w = m_srcRectWidth, h = m_srcRectHeight;            // no anamorphic sources here (m_srcAnamorphic :3497-3499)
    if (m_iRotation == 90 || m_iRotation == 270) std::swap(w, h);     // :3500-3502
    const size_t need = (size_t)w * 4 * h;
    if (!hostBGRA) { *size = need; return MPCVR_S_OK; }
    if (*size < need) { *size = need; return Fail(MPCVR_E_INVALIDARG, "buffer too small"); }
    if (!m_curSample) return Fail(MPCVR_E_NOT_VALID_STATE, "no sample");
    (void)hipSetDevice(m_device);
    HRESULT hr;
    if ((hr = CheckHip(m_Snapshot.CheckCreate(need), "snapshot"))) return hr;
    // temporarily point video/window rect at the image (:3549-3553), B8G8R8X8 target (:3518)
    const CRect backupVid = m_videoRect, backupWnd = m_windowRect;
    const int backupOut = m_cfg.output_format;
    // an HDR source shown in HDR is snapshot as SDR: m_bHdrPassthrough / m_bHdrLocalToneMapping are cleared around the draw and
    // the convert shader rebuilt with the PQ / HLG -> SDR tail (:3530-3545, restored :3562-3580)
    const bool backupHdr = m_hdrOutput;
    const bool backupOverride = m_blobOverride;
    // a context running on rank 0's broadcast parameter blob keeps it: the whole block is saved and put back verbatim (recomputing
    // the parameters locally and then claiming "overridden" again would silently drop the blob's matrices, tail and tables)
    struct { float cm[12]; float lum; float gamut[9]; int tail; float gamma; Up2xWeights ux, uy; } keep{};
    std::vector<uint16_t> keepDither;
    std::vector<float> keepLut;
    if (m_hdrOutput && backupOverride) {
        std::memcpy(keep.cm, m_cm, sizeof(m_cm)); keep.lum = m_lumScale; std::memcpy(keep.gamut, m_gamut, sizeof(m_gamut));
        keep.tail = m_tail; keep.gamma = m_gamma; keep.ux = m_upX; keep.uy = m_upY;
        keepDither.assign(m_ditherHost, m_ditherHost + sizeof(m_ditherHost) / sizeof(m_ditherHost[0]));
        keepLut.assign(m_pqLutHost, m_pqLutHost + sizeof(m_pqLutHost) / sizeof(m_pqLutHost[0]));
    }
    if (m_hdrOutput) { m_hdrOutput = false; m_blobOverride = false; SetShaderConvertColorParams(); UpdateHdrToneMapParams(); }
    m_videoRect = CRect(0, 0, w, h); m_windowRect = m_videoRect; m_cfg.output_format = MPCVR_OUT_BGRA8;
    m_planDirty = true;
    m_noLanesOnce = true;                    // the read-back below follows on the context stream
    hr = Process(m_Snapshot.ptr, w * 4, nullptr, nullptr, false);
    m_noLanesOnce = false;
    m_videoRect = backupVid; m_windowRect = backupWnd; m_cfg.output_format = backupOut;
    if (backupHdr) {
        m_hdrOutput = true; SetShaderConvertColorParams(); UpdateHdrToneMapParams();
        if (backupOverride) {
            std::memcpy(m_cm, keep.cm, sizeof(m_cm)); m_lumScale = keep.lum; std::memcpy(m_gamut, keep.gamut, sizeof(m_gamut));
            m_tail = keep.tail; m_gamma = keep.gamma; m_upX = keep.ux; m_upY = keep.uy;
            std::memcpy(m_ditherHost, keepDither.data(), sizeof(m_ditherHost));
            std::memcpy(m_pqLutHost, keepLut.data(), sizeof(m_pqLutHost));
            m_blobOverride = true;
        }
    }
    m_planDirty = true;
    if (hr) return hr;
    if (!m_evRb0 && ((hr = CheckHip(hipEventCreate(&m_evRb0), "read-back timer")) || (hr = CheckHip(hipEventCreate(&m_evRb1), "read-back timer")))) return hr;
    (void)hipEventRecord(m_evRb0, m_stream);
    if ((hr = CheckHip(hipMemcpyAsync(hostBGRA, m_Snapshot.ptr, need, hipMemcpyDeviceToHost, m_stream), "readback"))) return hr;
    (void)hipEventRecord(m_evRb1, m_stream);
    m_rbTimed = true;
    if ((hr = CheckHip(hipStreamSynchronize(m_stream), "readback sync"))) return hr;
    // (the error-diffusion pass's give-up flag: a snapshot is a result handed back, it must not carry a broken frame with S_OK)
    if (m_edStatus && *m_edStatus) { *m_edStatus = 0; return Fail(MPCVR_E_FAIL, "error diffusion: a band gave up waiting for the band above"); }
    *size = need;
    return MPCVR_S_OK;
}

void CHipVideoProcessor::Flush()
{
    if (m_bInit) { (void)hipSetDevice(m_device); (void)JoinFrameLanes(true); (void)hipStreamSynchronize(m_stream); }
    m_curSample = nullptr;
    // m_DoviExtensionMetadata = {} (:4082): L1 / L2 are forgotten; the uploaded constants change with the next RPU
    m_doviL1Present = m_doviL2Present = false;
    std::memset(m_doviL1, 0, sizeof(m_doviL1));
}

HRESULT CHipVideoProcessor::Reset()
{
    Flush();
    m_planDirty = true;
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::GetParamBlob(void *buf, size_t *size)
{
    if (!size) return Fail(MPCVR_E_POINTER, "null size");
    if (!m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    if (!buf) { *size = sizeof(ParamBlob); return MPCVR_S_OK; }
    if (*size < sizeof(ParamBlob)) { *size = sizeof(ParamBlob); return Fail(MPCVR_E_INVALIDARG, "buffer too small"); }
    HRESULT hr;
    if (m_planDirty && (hr = UpdatePlan())) return hr;
    ParamBlob b{};
    b.magic = kBlobMagic; b.version = 1;
    std::memcpy(b.cm, m_cm, sizeof(m_cm));
    b.lum_scale = m_lumScale;
    std::memcpy(b.gamut, m_gamut, sizeof(m_gamut));
    b.tail = m_tail; b.gamma = m_gamma;
    b.upx = m_upX; b.upy = m_upY;
    std::memcpy(b.dither, m_ditherHost, sizeof(m_ditherHost));
    if (m_tail == TAIL_PQ_TO_SDR) BuildPqSdrLut(m_lumScale, b.pq_lut);
    std::memcpy(buf, &b, sizeof(b));
    *size = sizeof(b);
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::SetParamBlob(const void *buf, size_t size)
{
    if (!buf) return Fail(MPCVR_E_POINTER, "null blob");
    if (size < sizeof(ParamBlob)) return Fail(MPCVR_E_INVALIDARG, "blob too small");
    if (!m_bInit || !m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    ParamBlob b;
    std::memcpy(&b, buf, sizeof(b));
    if (b.magic != kBlobMagic || b.version != 1) return Fail(MPCVR_E_INVALIDARG, "bad blob magic/version");
    // the blob crosses a process boundary (rank 0's broadcast): nothing in it is trusted to index or select kernels unchecked
    if (b.tail < TAIL_NONE || b.tail > TAIL_HLG_TO_PQ) return Fail(MPCVR_E_INVALIDARG, "blob: tail kind out of range");
    for (const Up2xWeights *w : {&b.upx, &b.upy})
        if ((w->ntaps != 0 && w->ntaps != 4 && w->ntaps != 6) || (w->q1_quirk != 0 && w->q1_quirk != 1))
            return Fail(MPCVR_E_INVALIDARG, "blob: phase-weight table malformed");
    if (!(b.lum_scale > 0.0f) || !(b.gamma > 0.0f || b.tail != TAIL_GAMMA_GAMUT)) return Fail(MPCVR_E_INVALIDARG, "blob: luminance scale / gamma");
    (void)hipSetDevice(m_device);
    std::memcpy(m_cm, b.cm, sizeof(m_cm));
    m_lumScale = b.lum_scale;
    std::memcpy(m_gamut, b.gamut, sizeof(m_gamut));
    m_tail = b.tail; m_gamma = b.gamma;
    m_upX = b.upx; m_upY = b.upy;
    std::memcpy(m_ditherHost, b.dither, sizeof(m_ditherHost));
    std::memcpy(m_pqLutHost, b.pq_lut, sizeof(m_pqLutHost));
    HRESULT hr;
    if ((hr = CheckHip(hipStreamSynchronize(m_stream), "sync"))) return hr;
    if ((hr = CheckHip(hipMemcpy(m_dither.ptr, m_ditherHost, sizeof(m_ditherHost), hipMemcpyHostToDevice), "dither upload"))) return hr;
    m_blobOverride = true;
    m_planDirty = true;
    return MPCVR_S_OK;
}

// ---- RCCL ----
// The library does not link librccl: a host that never broadcasts never loads it.  The communicator comes from the host, so the
// host's RCCL is already mapped when these entry points are called; it is looked up (RTLD_NOLOAD first: torch ships its own copy)
// and only loaded from the default search path when the process has none yet.  Prototypes as in rccl.h (ncclResult_t = int, ncclChar = 0).
namespace {
struct RcclApi {
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
const RcclApi &Rccl()
{
    static const RcclApi api = [] {
        RcclApi a;
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.Broadcast = (decltype(a.Broadcast))dlsym(h, "ncclBroadcast");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
        a.ok = a.Broadcast != nullptr;
        return a;
    }();
    return api;
}
}  // namespace

HRESULT CHipVideoProcessor::BroadcastParamBlobBegin(void *ncclComm, int root, int rank)
{
    if (!m_bInit || !m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    if (!ncclComm) return Fail(MPCVR_E_POINTER, "null RCCL communicator");
    if (m_bcastPending) return Fail(MPCVR_E_NOT_VALID_STATE, "a parameter-blob broadcast is already in flight");
    const RcclApi &R = Rccl();
    if (!R.ok) return Fail(MPCVR_E_FAIL, std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "ncclBroadcast not found"));
    (void)hipSetDevice(m_device);
    HRESULT hr;
    if ((hr = CheckHip(m_bcast.CheckCreate(sizeof(ParamBlob)), "broadcast buffer"))) return hr;
    m_bcastIsRoot = rank == root;
    if (m_bcastIsRoot) {
        m_bcastHost.resize(sizeof(ParamBlob));
        size_t size = m_bcastHost.size();
        if ((hr = GetParamBlob(m_bcastHost.data(), &size))) return hr;
        if ((hr = CheckHip(hipMemcpyAsync(m_bcast.ptr, m_bcastHost.data(), sizeof(ParamBlob), hipMemcpyHostToDevice, m_stream), "blob upload"))) return hr;
    }
    const int rc = R.Broadcast(m_bcast.ptr, m_bcast.ptr, sizeof(ParamBlob), /* ncclChar */ 0, root, ncclComm, m_stream);
    if (rc != 0) return Fail(MPCVR_E_FAIL, std::string("ncclBroadcast: ") + (R.GetErrorString ? R.GetErrorString(rc) : "error"));
    m_bcastPending = true;
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::BroadcastParamBlobEnd()
{
    if (!m_bcastPending) return Fail(MPCVR_E_NOT_VALID_STATE, "no parameter-blob broadcast in flight");
    (void)hipSetDevice(m_device);
    m_bcastPending = false;
    HRESULT hr;
    if (m_bcastIsRoot) return CheckHip(hipStreamSynchronize(m_stream), "broadcast sync");
    m_bcastHost.resize(sizeof(ParamBlob));
    if ((hr = CheckHip(hipMemcpyAsync(m_bcastHost.data(), m_bcast.ptr, sizeof(ParamBlob), hipMemcpyDeviceToHost, m_stream), "blob download"))) return hr;
    if ((hr = CheckHip(hipStreamSynchronize(m_stream), "broadcast sync"))) return hr;
    return SetParamBlob(m_bcastHost.data(), m_bcastHost.size());
}

HRESULT CHipVideoProcessor::GetColorMatrix(float out[12])
{
    if (!m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    std::memcpy(out, m_cm, sizeof(m_cm));
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::GetExtFmt(uint32_t *v)
{
    if (!m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    *v = m_srcExFmt.value;
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::GetFrameBytes(size_t *bytes, int *pitch)
{
    if (!m_srcParams) return Fail(MPCVR_E_NOT_VALID_STATE, "InitMediaType has not been called");
    if (bytes) *bytes = (size_t)m_srcPitch * m_srcLines;
    if (pitch) *pitch = m_srcPitch;
    return MPCVR_S_OK;
}

std::string CHipVideoProcessor::GetPathInfo()
{
    if (!m_srcParams) return "uninitialised";
    if (m_planDirty && UpdatePlan() != MPCVR_S_OK) return "error: " + m_lastError;
    if ((!m_strip && !m_stripSurf) || m_plan.fused_up2x) return m_plan.describe();
    if ((m_strip || m_stripSurf) && (m_stripRan >= 0 ? m_stripRan == 1 : m_period))
        return m_plan.describe() + (m_strip ? ";kernel=fused_period(rows=" : ";kernel=fused_period:surface(rows=") + std::to_string(m_periodPlan.P) + ":" + std::to_string(m_periodPlan.Q) + ",taps=" + std::to_string(m_periodPlan.nt) + ",px_per_lane=2,strip=" + std::to_string(m_periodPlan.strip_w) + ",window=6 rows in registers)";
    return m_plan.describe() + (m_strip ? ";kernel=fused_strip(taps=" : ";kernel=fused_strip:surface(taps=") + std::to_string(m_stripPlan.nt) + ",px_per_lane=" + std::to_string(m_stripPlan.pxl) +
           ",strip=" + std::to_string(m_stripPlan.strip_w) + ",ring=" + std::to_string(m_stripPlan.ring) + ")";
}

// FrameStats.h:145-173: copyticks (:2594), paintticks (:2790) and the snapshot's read-back; -1 = not timed yet
HRESULT CHipVideoProcessor::GetLastTimings(float *copy_host_ms, float *upload_ms, float *process_ms, float *readback_ms)
{
    (void)hipSetDevice(m_device);
    auto elapsed = [](bool on, hipEvent_t a, hipEvent_t b) {
        float ms = -1.0f;
        if (on && hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) != hipSuccess) ms = -1.0f;
        return ms;
    };
    if (copy_host_ms) *copy_host_ms = m_copyHostMs;
    if (upload_ms) *upload_ms = elapsed(m_upTimed, m_evUp0, m_evUp1);
    if (process_ms) *process_ms = elapsed(m_timed, m_evStart, m_evStop);
    if (readback_ms) *readback_ms = elapsed(m_rbTimed, m_evRb0, m_evRb1);
    return MPCVR_S_OK;
}

HRESULT CHipVideoProcessor::GetLastProcessMs(float *ms)
{
    if (!ms) return Fail(MPCVR_E_POINTER, "null");
    if (!m_timed) return Fail(MPCVR_E_NOT_VALID_STATE, "nothing timed yet");
    (void)hipSetDevice(m_device);
    HRESULT hr;
    if ((hr = CheckHip(hipEventSynchronize(m_evStop), "event sync"))) return hr;
    return CheckHip(hipEventElapsedTime(ms, m_evStart, m_evStop), "event elapsed");
}

}  // namespace mpcvr
