// hip_video_processor.h — CHipVideoProcessor: the MI355X-native stand-in for the shader path of
// CDX11VideoProcessor (Source/DX11VideoProcessor.h:256-384).  Method names follow the reference's
// so the call sites in CMpcVideoRenderer map one-to-one; the D3D11 device/swap-chain/OSD/subtitle
// members have no counterpart here (out of scope).
#pragma once
#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "../../include/mpcvr.h"
#include "vp_launch.h"
#include "vp_params.h"
#include "vp_plan.h"

namespace mpcvr {

typedef int32_t HRESULT;

struct CRect {
    int left = 0, top = 0, right = 0, bottom = 0;
    CRect() = default;
    CRect(int l, int t, int r, int b) : left(l), top(t), right(r), bottom(b) {}
    int Width() const { return right - left; }
    int Height() const { return bottom - top; }
    bool IsRectNull() const { return !left && !top && !right && !bottom; }
    bool operator==(const CRect &o) const { return left == o.left && top == o.top && right == o.right && bottom == o.bottom; }
    bool operator!=(const CRect &o) const { return !(*this == o); }
};

// device allocation that only grows (CheckCreate analogue of Tex2D_t, DX11Helper.h:37-90)
struct DevBuffer {
    void *ptr = nullptr;
    size_t size = 0;
    hipError_t CheckCreate(size_t bytes);
    void Release();
};

class CHipVideoProcessor {
public:
    CHipVideoProcessor();
    ~CHipVideoProcessor();

    HRESULT Init(int device, const mpcvr_settings &settings);                 // ctor + Init (:381,547)
    HRESULT SetStream(hipStream_t s);
    HRESULT Synchronize();

    HRESULT InitMediaType(int cformat, int width, int height, int pitch, const CRect *srcRect, uint32_t extfmt); // :1742
    HRESULT SetVideoRect(const CRect &videoRect);                             // :3426
    HRESULT SetWindowRect(const CRect &windowRect);                           // :3433
    HRESULT SetRotation(int value);                                           // :4052
    // (extension) polls a band of the error-diffusion pass grants the band above before the launch is flagged failed; <= 0: the default (2^21,
    // about two seconds) — a host that shares the GPU, or runs under a debugger, raises it; one that wants a hard deadline lowers it
    HRESULT SetErrorDiffusionPatience(int polls) { m_edPatience = polls > 0 ? polls : 0; return MPCVR_S_OK; }
    HRESULT SetFlip(bool value);                                              // VideoProcessor.h:210
    HRESULT SetSampleFormat(int frameFormat);                                 // m_SampleFormat, :2209-2219
    HRESULT SetHdrOutput(bool enable, int toneMapType, float displayMaxNits);  // m_bHdrPassthrough / m_bHdrLocalToneMapping
    HRESULT SetHdrMetadata(float minMastering, float maxMastering, float maxCLL, float maxFALL);   // SetHDR10ShaderParams :907
    HRESULT SetDoviMetadata(const mpcvr_dovi_metadata *md);                   // CopySample :2270-2520 (IID_MediaSideDataDOVIMetadataV2)
    HRESULT Configure(const mpcvr_settings &config);                          // :3800
    HRESULT SetProcAmpValues(uint32_t flags, float b, float c, float h, float s); // :4506

    HRESULT CopySample(const void *data, int pitch, int memKind);             // :2202 / MemCopyToTexSrcVideo :1213
    HRESULT Process(void *pRenderTarget, int rtPitch, const CRect *srcRect, const CRect *dstRect, bool second); // :3285
    HRESULT Render(int field);                                                // :2599 minus Present
    HRESULT GetBackBuffer(void **ptr, int *pitch, int *w, int *h);
    HRESULT GetCurentImage(void *hostBGRA, size_t *size);                     // :3493
    HRESULT GetDisplayedImage(void *hostPixels, size_t *size, bool deepColor, int *width, int *height, int *bits);     // :3610
    HRESULT ProcessBatch(int n, const void *const *srcs, void *const *dsts, int rtPitch);
    std::string GetLastBatchInfo() const;
    HRESULT ProcessBatchDovi(int n, const void *const *srcs, void *const *dsts, int rtPitch, const mpcvr_dovi_metadata *rpus);
    void Flush();                                                             // :4074
    HRESULT Reset();                                                          // :3453

    HRESULT GetParamBlob(void *buf, size_t *size);
    HRESULT SetParamBlob(const void *buf, size_t size);
    // the one collective of the path — rank `root`'s parameter blob to every rank of an RCCL communicator the host created
    // (ncclCommInitRank / ncclCommInitAll), on the context's stream.  Begin queues the broadcast (inside the host's ncclGroupStart /
    // ncclGroupEnd when one process drives several devices), End waits for it and adopts the blob on the other ranks.
    HRESULT BroadcastParamBlobBegin(void *ncclComm, int root, int rank);
    HRESULT BroadcastParamBlobEnd();
    HRESULT GetColorMatrix(float out[12]);
    HRESULT GetExtFmt(uint32_t *v);
    HRESULT GetFrameBytes(size_t *bytes, int *pitch);
    std::string GetPathInfo();
    HRESULT GetLastProcessMs(float *ms);
    HRESULT GetLastTimings(float *copy_host_ms, float *upload_ms, float *process_ms, float *readback_ms);
    const char *LastError() const { return m_lastError.c_str(); }

private:
    HRESULT Fail(HRESULT hr, const std::string &msg);
    HRESULT CheckHip(hipError_t e, const char *what);
    bool IsInit() const { return m_bInit; }

    // mirrors of the reference's private helpers
    void SetShaderConvertColorParams();                  // :813
    void SetShaderLuminanceParams();                     // :889
    HRESULT UpdatePlan();                                // UpdateTexures/UpdatePostScaleTexures/Update*scalingShaders
    HRESULT ConvertColorPass(const uint8_t *sample);     // :3048
    HRESULT ResizeShaderPass(void *rt, int rtPitch, const uint8_t *sample);     // :3103 (+ FinalPass :3189 fused into the last draw)
    HRESULT ProcessOne(const uint8_t *sample, void *rt, int rtPitch);
    HRESULT UploadTaps(const HostAxisTaps &h, DevBuffer &bi, DevBuffer &bw, DevBuffer &bs, DevBuffer &bb, const std::vector<int32_t> &other, AxisTaps *out);
    HRESULT UploadIndex(const std::vector<int32_t> &v, DevBuffer &b);
    bool ConvertEnabled() const;                       // m_PSConvColorData.bEnable (:849-853)
    int TexPitch() const;                              // row pitch of the source texture (differs from the sample's for v210)
    HRESULT PrepareSample(const uint8_t *dev_sample, const uint8_t **tex);   // device sample -> source texture
    void FillConvertParams(const uint8_t *sample, ConvertParams *P) const;
    StoreParams MakeStore(void *dst, int pitch, int dstFmt, bool rt) const;
    void FillFusedParams(const uint8_t *sample, void *rt, int rtPitch, FusedParams *fp) const;

    bool m_bInit = false;
    int m_device = 0;
    hipStream_t m_stream = nullptr;
    bool m_ownStream = false;
    hipEvent_t m_evStart = nullptr, m_evStop = nullptr;
    bool m_timed = false;
    // FrameStats.h:145-173 beside paintticks: the last CopySample (host wall time = copyticks :2594, and its H2D transfer on the copy
    // stream) and the last GetCurentImage read-back
    hipEvent_t m_evUp0 = nullptr, m_evUp1 = nullptr, m_evRb0 = nullptr, m_evRb1 = nullptr;
    bool m_upTimed = false, m_rbTimed = false;
    float m_copyHostMs = -1.0f;
    std::string m_lastError;

    // settings (Settings_t mirror)
    mpcvr_settings m_cfg{};
    ProcAmp m_procAmp;
    int m_iRotation = 0;
    bool m_bFlip = false;
    int m_SampleFormat = 0;        // 0 progressive, 1 TFF, 2 BFF
    bool m_hdrOutput = false, m_hdrMetaValid = false;
    int m_hdrToneMapType = 0;
    float m_hdrDisplayMaxNits = 1000.0f, m_hdrMeta[4] = {0, 0, 0, 0};
    HdrToneMapParams m_hdrTm{};
    void UpdateHdrToneMapParams();
    // m_Dovi / m_DoviExtensionMetadata: the RPU of the current sample; L1 / L2 stay as last seen until Flush (:4082)
    struct DoviSlot { DoviParams *pinned = nullptr; hipEvent_t copied = nullptr; };
    bool m_doviValid = false;
    mpcvr_dovi_metadata m_doviMd{};
    DoviParams m_doviHost{};
    bool m_doviL1Present = false, m_doviL2Present = false;
    uint32_t m_doviL1[3] = {0, 0, 0};
    float m_doviL2Raw[5] = {0, 0, 0, 0, 0};       // cbuffer values for the last level-2 selection
    DevBuffer m_doviDev;
    DoviSlot m_doviSlots[4];
    unsigned m_doviSlotNext = 0;
    HRESULT UploadDoviParams();
    HRESULT ApplyDoviMetadata(const mpcvr_dovi_metadata *md);       // SetDoviMetadata without the upload
    // ProcessBatchDovi: one RPU per frame of a batch.  What the kernels of frame k read that its RPU decides ...
    struct DoviFrameState { DoviParams p; float cm[12]; HdrToneMapParams tm; };
    // ... and everything an RPU changes in the context (to take a step back when frame j turns out to open the next run)
    struct DoviWalkState {
        bool valid, l1Present, l2Present, blobOverride, planDirty;
        mpcvr_dovi_metadata md; DoviParams host; uint32_t l1[3]; float l2raw[5]; float cm[12]; int tail; float gamma; HdrToneMapParams tm;
    };
    void SaveDoviWalk(DoviWalkState *s) const;
    void RestoreDoviWalk(const DoviWalkState &s);
    const DoviFrameState *m_dvFrames = nullptr;       // set around ProcessBatch by ProcessBatchDovi: the frames of the run in flight
    int m_dvCount = 0;
    const DoviParams *m_dvTabReady = nullptr;         // device tables of the run, uploaded: DoviParams[n], then cm[12 n] ...
    const float *m_dvCmReady = nullptr;
    const DoviParams *m_dvTabDev = nullptr;           // ... and in use: ProcessBatch took a whole-batch route (FillConvertParams / FillFusedParams point the kernels at them)
    const float *m_dvCmDev = nullptr;
    struct DoviTableSlot { void *pinned = nullptr; size_t cap = 0; DevBuffer dev; hipEvent_t done = nullptr; bool used = false; };
    DoviTableSlot m_dvSlots[2];
    unsigned m_dvSlotNext = 0;
    HRESULT UploadDoviTables(int n, hipEvent_t *done);
    HRESULT ApplyDoviFrame(const DoviFrameState &f);
    std::string m_dvLastInfo;                         // the runs of the last ProcessBatchDovi call (GetLastBatchInfo: ";dovi_runs=3:tables,1:frames")
    unsigned m_laneFrames = 0;                        // frames queued on the frame lanes (the timing pair is recorded on every n-th)
    unsigned m_launches = 0;                          // kernel launches so far (CheckHip) ...
    int m_lastBatchLane = -1;                                // the lane the last batch ran on (-1: the context stream)
    int m_lastBatchFrames = 0, m_lastBatchLaunches = 0;      // ... and what the last batch call used
    HRESULT ProcessBatchRoutes(int n, const void *const *srcs, void *const *dsts, int rtPitch);
    bool ToneMapActive() const;
    int m_firstAxis = 0;           // screen axis the first draw's tap table runs along
    bool m_firstSwap = false;      // rotation 90/270: taps address the other texture axis
    bool m_firstJinc = false, m_secondJinc = false;    // the draw runs the 2-D Jinc2m shader
    DrawCoords m_firstCoords{}, m_secondCoords{};

    // input
    const FmtConvParams *m_srcParams = nullptr;
    int m_srcWidth = 0, m_srcHeight = 0, m_srcPitch = 0, m_srcLines = 0;
    bool m_srcBottomUp = false;    // RGB DIB stored bottom-up (negative m_srcPitch in the reference)
    CRect m_srcRect;
    int m_srcRectWidth = 0, m_srcRectHeight = 0;
    ExtFmt m_decExFmt{0}, m_srcExFmt{0};
    CRect m_videoRect, m_windowRect;

    // constants (PS_COLOR_TRANSFORM, PS_PARAMETERS, matrix_conv_prim)
    float m_cm[12] = {0};
    float m_lumScale = 80.0f;
    float m_gamut[9] = {0};
    int m_tail = TAIL_NONE;
    float m_gamma = 1.0f;
    bool m_blobOverride = false;
    DevBuffer m_bcast;             // the blob in device memory while an RCCL broadcast is in flight
    bool m_bcastPending = false, m_bcastIsRoot = false;
    std::vector<unsigned char> m_bcastHost;      // host side of that copy: alive until BroadcastParamBlobEnd

    // plan
    bool m_planDirty = true;
    PassPlan m_plan;
    Up2xWeights m_upX{}, m_upY{};

    // device resources
    DevBuffer m_TexSrcVideo;       // uploaded sample (for v210: the Y210 texture CopyFrameV210 fills)
    DevBuffer m_TexRaw;            // v210 only: the raw sample before the unpack
    DevBuffer m_TexPost;           // m_TexsPostScale stand-in: input of the HDR10 tone-mapping step
    // upload ring (N3): pinned staging + device buffer per slot, copies on their own stream so that the upload of the
    // next sample overlaps the processing of the current one
    struct UploadSlot {
        void *pinned = nullptr; size_t pinnedSize = 0;
        DevBuffer dev;
        hipEvent_t uploaded = nullptr, consumed = nullptr;
        bool inFlight = false;         // an upload into this slot has been queued
        bool consumedRecorded = false; // a Process that read it has been queued after that upload
    };
    static constexpr int kUploadSlots = 3;
    UploadSlot m_up[kUploadSlots];
    int m_upNext = 0, m_curSlot = -1;
    hipStream_t m_copyStream = nullptr;
    void MarkConsumed();
    const uint8_t *m_curSample = nullptr;   // device pointer of the current sample (own buffer or zero-copy)
    DevBuffer m_TexConvertOutput, m_TexResize, m_BackBuffer, m_Snapshot;
    int m_backW = 0, m_backH = 0, m_backFmt = 0;       // the last frame Render put into m_BackBuffer: size and format (GetDisplayedImage)
    DevBuffer m_dither;
    DevBuffer m_pqLut;             // kPqLutSize floats (fused path tone-map table)
    DevBuffer m_hlgLut;            // kPqLutSize floats: per-channel inverse HLG OETF (fused kernels' HLG -> SDR tail)
    DevBuffer m_eotfLut;           // kEotfLutSize + 1 floats: PQ EOTF (Dolby Vision block convert), uploaded with the first RPU
    float m_pqLutHost[kPqLutSize];
    bool m_pqLutValid = false;
    DevBuffer m_tapsXi, m_tapsXw, m_tapsXs, m_tapsYi, m_tapsYw, m_tapsYs, m_otherX, m_otherY, m_tapsXb, m_tapsYb;
    AxisTaps m_tapsX{}, m_tapsY{};
    // ring of frame-table slots for mpcvr_process_batch (pinned host copy + device copy + completion event)
    static constexpr int kFrameSlots = 4;
    struct FrameSlot { void *pinned = nullptr; DevBuffer dev; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; };
    FrameSlot m_slots[kFrameSlots];
    int m_slotNext = 0;
    uint16_t m_ditherHost[1024];
    // mpcvr_process_batch on the pass-per-kernel path: the frames of a batch are independent, so they can be dealt to a
    // few lanes (stream + private intermediates) whose kernels overlap (opt-in, MPCVR_BATCH_LANES; see ProcessBatch).
    // Lane 0 is the context stream with m_TexConvertOutput / m_TexResize / m_TexPost.
    static constexpr int kLanes = 4;
    struct Lane { hipStream_t stream = nullptr; DevBuffer conv, mid, post; hipEvent_t done = nullptr; };
    Lane m_lanes[kLanes];
    hipEvent_t m_fork = nullptr;
    // mpcvr_process frame after frame (the reference's own call pattern, Render -> Process, DX11VideoProcessor.cpp:2730): a single 4K
    // frame is one round of waves on this part, so a kernel's ramp-up and drain cost a third of its time when frames run strictly one
    // after the other.  Frames are independent (a D3D11 driver overlaps draws into different render targets as well): a context that
    // owns its stream deals consecutive frames to four lanes whose kernels overlap; everything that can observe a result
    // (mpcvr_synchronize, the snapshot, a batch, a plan change, a new stream) joins them first.  The lane streams are BLOCKING streams
    // like the context's own (Init), so work on the legacy default stream stays ordered against them.
    static constexpr int kFrameLanes = 8;         // built; FrameLaneCount() of them are used (4 unless MPCVR_FRAME_LANES says otherwise)
    static int FrameLaneCount();
    // every frame queued on a lane leaves (render target, completion event) in the lane's ring; a slot is reused only after its frame has
    // completed, which also bounds how far the host runs ahead (kFrameLanes x kLaneDepth frames)
    static constexpr int kLaneDepth = 8;
    struct LaneFrame { const void *rt = nullptr; hipEvent_t done = nullptr; bool pending = false; };
    // (round 6) WHOLE BATCHES take turns on the first two lanes as well (ProcessBatch on a context that owns its stream, a plan that is one
    // launch per batch with no intermediate surface): two launches in flight fill each other's ramp-up and tail — same box, 32-frame batches:
    // 4K -> 8K 22.6 k -> 23.5 k frames/s, 1080p -> 1440p 99.4 k -> 115.6 k (profiles/r06/final4/bench_workloads.jsonl).  batchRts: the render targets of
    // the lane's batches still in flight (sorted), batchDone: the event behind the last of them.
    struct FrameLane { hipStream_t stream = nullptr; LaneFrame ring[kLaneDepth]; int head = 0; hipEvent_t last = nullptr; unsigned seenGen = 0;
                       std::vector<const void *> batchRts; hipEvent_t batchDone = nullptr; bool batchPending = false; };
    static constexpr int kBatchLanes = 2;
    int m_blaneNext = 0;
    bool m_batchOnLane = false;               // ProcessBatchRoutesOn runs with m_stream = a lane's stream
    bool BatchLanesUsable(int n, const void *const *srcs, void *const *dsts, int rtPitch);
    FrameLane *PickBatchLane(int n, void *const *dsts);
    void NoteLaneBatch(FrameLane *fl, int n, void *const *dsts);
    HRESULT ProcessBatchRoutesOn(int n, const void *const *srcs, void *const *dsts, int rtPitch);
    // work queued on the CONTEXT stream (a batch, a frame that ran off the lanes, a sample copy / repack, a read-back) since a lane last
    // waited for it: every such call bumps m_streamGen; a lane whose seenGen is behind waits for an event recorded on the context stream
    // (m_evStreamMark, recorded once per generation) before its next frame — a lane frame into the render target, or out of the sample, that
    // the context stream is still writing or reading can then neither overtake nor overlap it (mpcvr.h: frames into the same target stay in order)
    unsigned m_streamGen = 0, m_markGen = ~0u;
    hipEvent_t m_evStreamMark = nullptr;
    void NoteStreamWork() { m_streamGen++; }
    void LaneWaitsForStream(FrameLane *fl);
    FrameLane m_flanes[kFrameLanes];
    int m_flaneNext = 0;
    int m_inflight = 1;                       // FusedParams::inflight of the frame being queued
    size_t m_clearOnRun = 0;                  // Render: bytes of the back buffer to clear in front of the frame, on whatever stream it runs
    bool m_noLanesOnce = false;               // the snapshot's Process stays on the context stream
    hipStream_t m_lastRun = nullptr;          // the stream the current sample's last Process ran on (MarkConsumed records there)
    bool FrameLanesUsable() const;
    FrameLane *PickFrameLane(const void *rt);
    HRESULT JoinFrameLanes(bool host_wait);
    void NoteLaneFrame(FrameLane *fl, const void *rt);
    size_t m_convBytes = 0, m_midBytes = 0, m_postBytes = 0;
    // resources of the frame being processed (lane 0 outside ProcessBatch)
    hipStream_t m_run = nullptr;
    void *m_runConv = nullptr, *m_runMid = nullptr, *m_runPost = nullptr;
    void UseLane(int lane);
    // whole-batch launches of the pass-per-kernel path (block convert + folded resize kernels with a frame dimension)
    DevBuffer m_batchConv, m_batchMid;
    DevBuffer m_batchPost;         // HDR10 tone-mapping step of a batch: the frames' m_TexsPostScale copies side by side
    // EXTENSION (bUseDither = 2, m_plan.errdiff): the frames as a 10-bit swap chain would receive them, window geometry, side by side;
    // the error-diffusion pass (vp_errdiff.hip) reads them and writes the real render targets
    DevBuffer m_edPost;
    DevBuffer m_edHandoff;         // the pass's hand-off rows between bands of 64 rows (vp_errdiff.hip)
    struct EdLayout {              // what the hand-off rows of the last pass were laid out for (ErrDiffPass): compared field by field
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0, n = 0; const void *rows = nullptr;
        bool operator==(const EdLayout &o) const { return x0 == o.x0 && x1 == o.x1 && y0 == o.y0 && y1 == o.y1 && n == o.n && rows == o.rows; }
    };
    int m_edGen = 0; EdLayout m_edKey;       // generation of the hand-off words of the last pass, and the layout they belong to
    int m_edPatience = 0;          // SetErrorDiffusionPatience: polls per group before a band gives up (0: the launcher's default)
    int *m_edStatus = nullptr;     // pinned host word the pass sets when a band gave up waiting (checked at the next pass and in Synchronize)
    uint8_t *m_edBase = nullptr;   // first intermediate (m_edPost.ptr + a margin)
    int m_edPitch = 0;             // bytes per row of an intermediate (a multiple of 256)
    size_t m_edStride = 0;         // bytes per intermediate
    HRESULT PrepareErrDiff(int frames);
    HRESULT ErrDiffPass(int n, const FusedFrame *table, FusedFrame single, void *const *dsts, int rtPitch, hipStream_t s);
    HRESULT ProcessBatchErrDiff(int n, const void *const *srcs, void *const *dsts, int rtPitch);
    size_t PostStride() const { return (m_postBytes + 255) & ~(size_t)255; }
    // a frame table in a slot of the ring (pinned copy + device copy): frame i = {srcs ? srcs[i] : null, dsts ? dsts[i] : dst_base + i * dst_stride}
    HRESULT UploadFrameTable(int n, const void *const *srcs, void *const *dsts, uint8_t *dst_base, size_t dst_stride, const FusedFrame **dev, hipEvent_t *done);
    DevBuffer m_batchTex;          // interleaved RGB / v210 batches: the frames' m_TexSrcVideo copies side by side (ProcessBatch)
    bool m_texSrcZeroed = false, m_batchTexZeroed = false;     // the texels the RGB copy loops never write have been cleared for the current media type
    bool m_startRecorded = false;  // ProcessBatch: m_evStart already sits in front of a repack launch
    bool m_keepStart = false;      // ProcessBatchErrDiff: m_evStart sits in front of the FIRST chunk; the chunks' ProcessBatchRoutes calls leave it there
    bool m_batchRepacked = false;  // the batch at hand reads v210 samples already repacked into m_batchTex
    bool m_batchSrc16 = false;     // every sample of the batch being planned starts on a 16-byte boundary
    // Jinc2m phase tables of the first / second draw (null: weights per pixel)
    DevBuffer m_jincFirst, m_jincSecond, m_jincFused;
    const float *m_jincFusedTab = nullptr;                                  // the fused Jinc2m kernel's weight table (BuildFusedJincTable), PassPlan::fused_jinc
    const void *m_jincFirstTab = nullptr, *m_jincSecondTab = nullptr;
    const float *m_jincFirstCtr = nullptr, *m_jincSecondCtr = nullptr;     // the plain kernel's texcoord tables (no phase table: BuildDrawCentres), in the same buffers
    HRESULT UploadJincPhases(const DrawCoords &dc, DevBuffer &buf, const void **tab, const float **ctr);
    // arbitrary-ratio fused kernel (vp_fused_strip.hip): geometry planned with the tap tables (UpdatePlan)
    bool m_strip = false;          // raw 4:2:0 sample -> render target in one kernel
    bool m_stripSurf = false;      // any other source: the convert kernel's output (or the RGB source texture) -> render target through the same kernel, no convert stage
    bool m_stripPlanned = false;
    bool FillStripSurfParams(const Surface &src, const StoreParams &store, FusedStripParams *sp) const;
    StripPlan m_stripPlan;
    DevBuffer m_stripTab;          // yrange | xstrip | xi_t | xw_t | yi | yw, word offsets in m_stripOff
    size_t m_stripOff[6] = {0, 0, 0, 0, 0, 0};
    bool FillStripParams(const uint8_t *sample, void *dst, int dstPitch, const StoreParams &store, FusedStripParams *sp) const;
    // periodic-phase variant of the same launch (vp_fused_period.h): vertical ratio 4:3 / 3:2 / 2:3 / 1:2 / 3:1, tables behind the strip kernel's in m_stripTab
    PeriodPlan m_periodPlan;       // P == 0: not a periodic geometry
    mutable int m_stripRan = -1;   // which kernel the last strip launch of this plan really ran (1 = k_fused_period, 0 = k_fused_strip, -1 = none yet): the plan-time
                                   // probe uses a null, aligned target — a real target with an odd pitch or offset sends the launch to k_fused_strip (GetPathInfo reports what ran)
    size_t m_periodOff[4] = {0, 0, 0, 0};     // xi_t | xw_t | yw | xstrip
    bool m_period = false;         // the planned launch (window-sized target) takes the periodic kernel: what GetVPInfo reports
    bool BatchPlan(const uint8_t *sample0, void *rt0, int rtPitch, bool aligned, FusedParams *conv, FusedParams *direct) const;
    HRESULT ProcessBatchLaunches(int n, const FusedFrame *table, const uint8_t *sample0, void *rt0, int rtPitch, bool aligned);
    HRESULT PrepareLanes(int lanes);
};

}  // namespace mpcvr
