// vp_fused_strip.hip — host side of the arbitrary-ratio fused kernel (vp_fused_strip.h): preconditions, work decomposition, dispatch to
// the per-tap-count translation units (vp_fused_strip_nt{4,6,8,16}.hip hold the kernels).
#include "vp_fused_strip.h"

#include <cstdio>
#include <map>
#include <mutex>

namespace mpcvr {

extern template hipError_t LaunchFusedStripNT<4>(const FusedArgs &, const StripArgs &, const StoreParams &, int, int, int, int, bool, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
extern template hipError_t LaunchFusedStripNT<6>(const FusedArgs &, const StripArgs &, const StoreParams &, int, int, int, int, bool, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
extern template hipError_t LaunchFusedStripNT<8>(const FusedArgs &, const StripArgs &, const StoreParams &, int, int, int, int, bool, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
extern template hipError_t LaunchFusedStripNT<16>(const FusedArgs &, const StripArgs &, const StoreParams &, int, int, int, int, bool, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);

// host side -------------------------------------------------------------------------------------------------------------

bool FusedStripSupported(const FusedStripParams &S)
{
    const FusedParams &P = S.fp;
    const ConvertParams &c = P.conv;
    // the last tap of the Y draw saturates to [0, 1] (UNORM targets store that way anyway): an fp16 destination — m_TexsPostScale in front
    // of the HDR10 tone-mapping step with iTexFormat = 16FLOAT — keeps the filter's over- and undershoot in the reference, so it is not
    // this kernel's case (k_resize* store fp16 unclamped)
    if (P.store.mode == ST_SURFACE && P.store.dst_fmt == SF_RGBA16F) return false;
    if (S.surface_mode) {
        if (S.surf.fmt != SF_BGRA8 && S.surf.fmt != SF_RGB10A2 && S.surf.fmt != SF_RGBA16F) return false;
        if (S.mid_h < 1 || S.surf.w < 1 || S.surf.pitch <= 0 || (S.surf.pitch & 3)) return false;
        if ((uint64_t)S.surf.pitch * (uint64_t)S.surf.h >= (1ull << 32)) return false;
        if (P.store.off_x + S.out_w > 0 && (uint64_t)P.store.dst_pitch * (uint64_t)(std::max(P.store.off_y, 0) + S.out_h) >= (1ull << 32)) return false;
        if (S.nt != 4 && S.nt != 6 && S.nt != 8 && S.nt != 16) return false;
        if (!S.xi_t || !S.xw_t || !S.yi || !S.yw || !S.yrange || !S.xstrip) return false;
        if ((S.ring != 8 && S.ring != 16 && S.ring != 32) || (S.pxl != 1 && S.pxl != 2) || (S.nt == 16 && S.pxl != 1)) return false;
        return S.strip_w >= S.pxl && S.strip_w <= 64 * S.pxl && (S.strip_w % S.pxl) == 0;
    }
    if (c.out_fmt != SF_BGRA8 && c.out_fmt != SF_RGB10A2) return false;
    if (!BlockConvertLayout(P, false) || c.blend_deint || c.dovi) return false;
    if (c.out_w < 8 || c.out_h < 2 || (c.out_w & 1) || (c.out_h & 1)) return false;
    if (!P.fast_convert) return false;
    if ((uint64_t)c.pitch[0] * (uint64_t)(c.rect_t + c.out_h + 2) >= (1ull << 32)) return false;
    if ((uint64_t)P.plane_off[1] >= (1ull << 31) || (uint64_t)P.plane_off[2] >= (1ull << 31)) return false;
    if (P.store.off_x + S.out_w > 0 && (uint64_t)P.store.dst_pitch * (uint64_t)(std::max(P.store.off_y, 0) + S.out_h) >= (1ull << 32)) return false;
    if (S.nt != 4 && S.nt != 6 && S.nt != 8 && S.nt != 16) return false;
    if (!S.xi_t || !S.xw_t || !S.yi || !S.yw || !S.yrange || !S.xstrip) return false;
    if (S.ring != 8 && S.ring != 16 && S.ring != 32) return false;
    if ((S.pxl != 1 && S.pxl != 2) || (S.nt == 16 && S.pxl != 1)) return false;
    if (S.strip_w < S.pxl || S.strip_w > 64 * S.pxl || (S.strip_w % S.pxl)) return false;
    return true;
}

// LDS per workgroup of a configuration with `waves` strips per workgroup
static size_t StripLds(const FusedStripParams &S, bool fastepi, bool lut, int waves)
{
    return (fastepi ? LDS_DB : 0) + (lut ? LDS_T : 0) + (size_t)waves * (2 * (size_t)S.acols * 8 + (size_t)S.ring * 64 * (S.pxl == 2 ? 12 : 8));
}
// The tables (dither, tone-map LUT) exist once per workgroup, the A slice and the ring once per wave: the more waves share a
// workgroup the more of them a CU's 160 KiB hold — occupancy is what hides this kernel's LDS and memory round trips.
static int StripWaves(const FusedStripParams &S, bool fastepi, bool lut)
{
    static const int env = EnvInt("MPCVR_STRIP_WAVES", 0);
    if (env >= 1 && env <= 16 && StripLds(S, fastepi, lut, env) <= DeviceLdsLimit()) return env;
    int best = 1, best_per_cu = 0;
    for (int w : {4, 6, 8, 10, 12, 14, 16}) {
        const size_t lds = StripLds(S, fastepi, lut, w);
        if (lds > DeviceLdsLimit()) break;
        const int per_cu = std::min((int)(DeviceLdsLimit() / lds) * w, 32);
        if (per_cu > best_per_cu) { best_per_cu = per_cu; best = w; }
    }
    return best;
}

// dynamic LDS above the default limit needs the function attribute once per kernel (and device); remembered per (kernel, device)
size_t DeviceLdsLimit()
{
    static std::mutex mu;
    static std::map<int, size_t> limit;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 160 * 1024;
    std::lock_guard<std::mutex> lock(mu);
    auto it = limit.find(dev);
    if (it != limit.end()) return it->second;
    int optin = 0, plain = 0;
    (void)hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev);
    (void)hipDeviceGetAttribute(&plain, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    size_t v = (size_t)std::max(std::max(optin, plain), 0);
    // MPCVR_LDS_LIMIT=<bytes>: plan as if the device granted no more (tests: the planners' fallbacks for parts / partitions with less LDS
    // than an MI355X's 160 KiB run on this box) — it can only lower the limit
    if (const int cap = EnvInt("MPCVR_LDS_LIMIT", 0); cap >= 32 * 1024 && (size_t)cap < v) v = (size_t)cap;
    if (EnvInt("MPCVR_LOG", 0) >= 2)
        std::fprintf(stderr, "mpcvr: device %d LDS per workgroup: opt-in %d B, default %d B\n", dev, optin, plain);
    return limit[dev] = v >= 32 * 1024 ? v : 160 * 1024;
}

hipError_t AllowLargeLds(const void *kern, size_t lds)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> granted;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    size_t &g = granted[{kern, dev}];
    if (g >= lds) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) g = lds;
    return e;
}

hipError_t LaunchFusedStrip(const FusedStripParams &S, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s)
{
    if (!frames_dev && n_frames != 1) return hipErrorInvalidValue;
    const FusedParams &P = S.fp;
    FusedArgs a;
    if (S.surface_mode) {       // store-side constants only: there is no convert stage
        std::memset(&a, 0, sizeof(a));
        a.maxv = S.surf.fmt == SF_RGB10A2 ? 1023.0f : 255.0f;       // format of m_TexsPostScale = the internal format
        if (P.store.mode == ST_FINAL) a.maxv = P.store.mid_fmt == SF_RGB10A2 ? 1023.0f : 255.0f;
        a.inv_maxv = 1.0f / a.maxv;
        a.q_over_maxv = (float)P.store.quant / a.maxv;
        a.epi_mul = FinalPassMultiplier(P.store.quant, (int)a.maxv);
        a.dst_pitch = P.store.dst_pitch; a.off_x = P.store.off_x; a.off_y = P.store.off_y;
        a.final_pass = P.store.mode == ST_FINAL; a.out10 = P.store.dst_fmt == SF_RGB10A2;
        a.quant = (float)P.store.quant;
        a.dither = P.store.dither;
        a.H = S.mid_h; a.W = S.surf.w;
    } else FillFusedArgs(P, a, 1);
    if (S.per_P) {       // a periodic vertical ratio: the register-window kernel when the launch meets its preconditions
        const hipError_t ep = LaunchFusedPeriod(S, a, frames_dev, single, n_frames, s);
        if (ep != hipErrorNotSupported) { if (S.ran_period) *S.ran_period = 1; return ep; }
    }
    if (S.ran_period) *S.ran_period = 0;
    StripArgs q{};
    q.xi_t = (const int32_t *)S.xi_t; q.xw_t = (const float *)S.xw_t;
    q.yi = (const int32_t *)S.yi; q.yw = (const float *)S.yw;
    q.yrange = (const int32_t *)S.yrange; q.xstrip = (const int32_t *)S.xstrip;
    q.out_w = S.out_w; q.out_h = S.out_h;
    q.strip_w = S.strip_w; q.n_strips = (S.out_w + S.strip_w - 1) / S.strip_w;
    q.ring_mask = S.ring - 1;
    q.acols = S.acols;
    q.a_scale = 16777216.0f / a.maxv;
    if (S.surface_mode) {
        q.surf = n_frames > 1 || !single.src ? (const uint8_t *)S.surf.ptr : nullptr;      // a batch reads surf + z * stride; one frame: single.src
        q.other = S.other; q.surf_fmt = S.surf.fmt; q.surf_pitch = S.surf.pitch; q.surf_w = S.surf.w; q.surf_stride = S.surf_stride;
        q.a_scale = S.surf.fmt == SF_RGBA16F ? 1.0f : 16777216.0f / (S.surf.fmt == SF_RGB10A2 ? 1023.0f : 255.0f);
    }
    // segment height: long segments recompute less (the taps' span of source rows each), short ones fill the chip
    static const int seg_env = EnvInt("MPCVR_STRIP_SEG", 0);
    int seg = seg_env;
    if (seg <= 0) {
        seg = 16;
        const long side = (long)n_frames * (P.inflight > 1 ? P.inflight : 1);       // frames sharing the chip: a batch, or single frames on the context's lanes
        const long want = n_frames > 1 ? 12288 : 4096;
        for (int cand : {192, 128, 96, 64, 48, 32, 24, 16})
            if ((long)q.n_strips * ((S.out_h + cand - 1) / cand) * side >= want || cand == 16) { seg = cand; break; }
    }
    q.seg_rows = std::min(seg, S.out_h);

    const StoreParams &st = P.store;
    const bool inside = st.off_x >= 0 && st.off_y >= 0 && (st.clip_w <= 0 || (st.off_x + S.out_w <= st.clip_w && st.off_y + S.out_h <= st.clip_h));
    const bool fastepi = inside && st.mode == ST_FINAL && st.dst_fmt == SF_BGRA8 && st.quant == 255 && a.epi_mul != 0 &&
                         (S.surface_mode || P.conv.out_fmt == SF_RGB10A2) && st.mid_fmt == SF_RGB10A2 && (st.dst_pitch & 3) == 0;
    // straight UNORM store of the Y result (no final pass) into a B8G8R8A8 / R10G10B10A2 target or post-scale texture
    const bool direct = !fastepi && inside && st.mode == ST_SURFACE && (st.dst_fmt == SF_BGRA8 || st.dst_fmt == SF_RGB10A2) && (st.dst_pitch & 3) == 0;
    const int tailk = S.surface_mode ? TAILK_NONE : FusedTailKind(P), srck = S.surface_mode ? SRC_SURFACE : FusedSourceKind(P);
    const int n_segs = (S.out_h + q.seg_rows - 1) / q.seg_rows;
    // waves per workgroup: as many as LDS allows for a launch that fills the chip several times over; a small launch (one frame) is
    // spread over all CUs instead — 1,800 work items in 16-wave workgroups occupy 113 of 256 CUs, four waves deep
    int waves = StripWaves(S, fastepi, tail_has_table(tailk));
    const long items = (long)q.n_strips * n_segs * n_frames;
    if (items < 512L * waves) waves = (int)std::max<long>(1, std::min<long>(waves, items / 512));
    const size_t lds = StripLds(S, fastepi, tail_has_table(tailk), waves);
    const dim3 grid((q.n_strips * n_segs + waves - 1) / waves, 1, n_frames), block(64 * waves, 1, 1);
    const int epi = fastepi ? STRIP_EPI_FAST : direct ? STRIP_EPI_DIRECT : STRIP_EPI_GENERIC;
    switch (S.nt) {
    case 4: return LaunchFusedStripNT<4>(a, q, st, S.pxl, tailk, srck, epi, S.surface_mode != 0, grid, block, lds, frames_dev, single, s);
    case 6: return LaunchFusedStripNT<6>(a, q, st, S.pxl, tailk, srck, epi, S.surface_mode != 0, grid, block, lds, frames_dev, single, s);
    case 8: return LaunchFusedStripNT<8>(a, q, st, S.pxl, tailk, srck, epi, S.surface_mode != 0, grid, block, lds, frames_dev, single, s);
    case 16: return LaunchFusedStripNT<16>(a, q, st, S.pxl, tailk, srck, epi, S.surface_mode != 0, grid, block, lds, frames_dev, single, s);
    }
    return hipErrorNotSupported;
}

size_t FusedStripLdsBytes(const FusedStripParams &S) { return StripLds(S, true, true, 1); }

}  // namespace mpcvr
