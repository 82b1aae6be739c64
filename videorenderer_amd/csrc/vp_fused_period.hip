// vp_fused_period.hip — host side of the periodic-phase fused kernel (vp_fused_period.h): preconditions, work decomposition,
// dispatch to the per-ratio translation units (vp_fused_period_{4_3,3_2,2_3,1_2,3_1}.hip hold the kernels).
#include "vp_fused_period.h"

namespace mpcvr {

extern template hipError_t LaunchFusedPeriodPQ<4, 3>(const FusedArgs &, const PeriodArgs &, int, int, int, int, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
extern template hipError_t LaunchFusedPeriodPQ<3, 2>(const FusedArgs &, const PeriodArgs &, int, int, int, int, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
extern template hipError_t LaunchFusedPeriodPQ<2, 3>(const FusedArgs &, const PeriodArgs &, int, int, int, int, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
extern template hipError_t LaunchFusedPeriodPQ<1, 2>(const FusedArgs &, const PeriodArgs &, int, int, int, int, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);
extern template hipError_t LaunchFusedPeriodPQ<3, 1>(const FusedArgs &, const PeriodArgs &, int, int, int, int, dim3, dim3, size_t, const FusedFrame *, FusedFrame, hipStream_t);

namespace {

// which epilogue the launch would run: EPI_DITHER8 (10-bit internal -> final pass -> B8G8R8A8), EPI_DIRECT8 (straight UNORM store), or
// -1 (window clipping, odd alignment, RGB10A2 behind a final pass ...: k_fused_strip's generic epilogue serves those)
int PeriodEpilogue(const FusedStripParams &S, uint32_t epi_mul)
{
    const FusedParams &P = S.fp;
    const StoreParams &st = P.store;
    const bool inside = st.off_x >= 0 && st.off_y >= 0 && (st.clip_w <= 0 || (st.off_x + S.out_w <= st.clip_w && st.off_y + S.out_h <= st.clip_h));
    // one 8-byte store and one 8-byte dither read per lane and row: even window column, 8-byte aligned rows and targets
    if (!inside || (st.off_x & 1) || (st.dst_pitch & 7) || !P.dst_aligned16) return -1;
    const int in_fmt = S.surface_mode ? S.surf.fmt : P.conv.out_fmt;        // format of m_TexConvertOutput (its UNORM scale is the window's)
    if (st.mode == ST_FINAL && st.dst_fmt == SF_BGRA8 && st.quant == 255 && epi_mul != 0 && in_fmt == SF_RGB10A2 && st.mid_fmt == SF_RGB10A2) return EPI_DITHER8;
    if (st.mode == ST_SURFACE && (st.dst_fmt == SF_BGRA8 || st.dst_fmt == SF_RGB10A2) && st.dst_fmt == in_fmt) return EPI_DIRECT8;
    return -1;
}

size_t PeriodLds(const FusedStripParams &S, bool fastepi, bool lut, int waves)
{
    return (fastepi ? LDS_DB : 0) + (lut ? LDS_T : 0) + (size_t)waves * (size_t)S.per_acols * 24;
}

}  // namespace

bool FusedPeriodTakes(const FusedStripParams &S)
{
    if (!S.per_P || !S.per_xi_t || !S.per_xw_t || !S.per_yw || !S.per_xstrip) return false;
    static const int off = EnvInt("MPCVR_NO_PERIOD", 0);
    if (off) return false;
    const FusedParams &P = S.fp;
    // surface mode (the convert output of another kernel, or an interleaved RGB source texture, feeds the X draw): UNORM texels read row
    // for row, through the draw's row map where there is one (a source rect, rotation 180); an fp16 internal format stays with k_fused_strip
    if (S.surface_mode && ((S.surf.fmt != SF_BGRA8 && S.surf.fmt != SF_RGB10A2) || (S.surf.pitch & 3))) return false;
    const int tailk = S.surface_mode ? TAILK_NONE : FusedTailKind(P);
    if (tailk == TAILK_ALU) return false;                         // the literal tails stay with k_fused_strip
    if (S.per_nt < 4 || S.per_nt > 5 || S.per_acols < 2 || (S.per_acols & 1)) return false;      // (4 taps, or Lanczos3 as Direct3D 11 draws it; six taps: k_fused_strip)
    if (S.per_strip_w < 2 || S.per_strip_w > kPeriodStripMax || (S.per_strip_w & 1)) return false;
    // measured (profiles/r03): without a table tail the 4-tap filters run as fast or faster through k_fused_strip (SDR 1080p -> 1440p
    // Catmull-Rom 120.6 k against 115.4 k frames/s: the convert is light, and that is where the register window pays)
    // (round 6: those instantiations — 35 kernels, a tenth of the library's build time, reachable through MPCVR_FLAG_FORCE_PERIOD only — left the build)
    if (!S.surface_mode && tailk == TAILK_NONE && S.per_nt == 4) return false;
    const uint32_t epi_mul = FinalPassMultiplier(P.store.quant, (S.surface_mode ? S.surf.fmt : P.conv.out_fmt) == SF_RGB10A2 ? 1023 : 255);
    const int epik = PeriodEpilogue(S, epi_mul);
    if (epik < 0) return false;
    return PeriodLds(S, epik == EPI_DITHER8, tail_has_table(tailk), 1) <= DeviceLdsLimit();
}

// the same launch contract as LaunchFusedStrip (which calls this first); hipErrorNotSupported = not this kernel's case
hipError_t LaunchFusedPeriod(const FusedStripParams &S, const FusedArgs &a_in, const FusedFrame *frames_dev, FusedFrame single, int n_frames, hipStream_t s)
{
    if (!FusedPeriodTakes(S)) return hipErrorNotSupported;
    const FusedParams &P = S.fp;
    const int tailk = S.surface_mode ? TAILK_NONE : FusedTailKind(P), srck0 = S.surface_mode ? SRC_SURFACE : FusedSourceKind(P);
    const int epik = PeriodEpilogue(S, a_in.epi_mul);
    // the straight store writes the TARGET's codes: StoreParams::quant is the swap chain's, which is not the target when the HDR10
    // tone-mapping step follows (the draw then goes into m_TexsPostScale, internal format — 8-bit texels in front of a 10-bit swap chain)
    FusedArgs a = a_in;
    if (epik == EPI_DIRECT8) a.quant = a.out10 ? 1023.0f : 255.0f;
    // built source specialisations: P01x and NV12 (and a surface); everything else reads its layout at run time
    const int srck = (srck0 == SRC_SURFACE || srck0 == SRC_P01X || (srck0 == SRC_NV12 && epik == EPI_DIRECT8)) ? srck0 : SRC_GENERIC;
    const int PB = 6 * S.per_P / S.per_Q;
    PeriodArgs q{};
    q.xi_t = (const int32_t *)S.per_xi_t; q.xw_t = (const float *)S.per_xw_t; q.yw = (const float *)S.per_yw; q.xstrip = (const int32_t *)S.per_xstrip;
    q.out_w = S.out_w; q.out_h = S.out_h;
    q.strip_w = S.per_strip_w;
    static const int own_env = EnvInt("MPCVR_PERIOD_OWN", -1);       // A/B knob: force the lane -> column ownership (0, 1, 2)
    q.own = own_env >= 0 && own_env <= 2 ? own_env : S.per_own;
    q.n_strips = (S.out_w + q.strip_w - 1) / q.strip_w;
    q.acols = S.per_acols;
    if (S.surface_mode) {
        q.surf = n_frames > 1 || !single.src ? (const uint8_t *)S.surf.ptr : nullptr;      // a batch reads surf + z * stride; one frame: single.src
        q.surf_fmt = S.surf.fmt; q.surf_pitch = S.surf.pitch; q.surf_w = S.surf.w; q.surf_stride = S.surf_stride; q.other = S.other;
    }
    // segment height (a multiple of the body's PB output rows): long segments recompute less (the taps' span each), short ones fill the chip
    static const int seg_env = EnvInt("MPCVR_PERIOD_SEG", 0);
    int seg = seg_env;
    if (seg <= 0) {
        seg = 2 * PB;
        const long side = (long)n_frames * (P.inflight > 1 ? P.inflight : 1);       // frames sharing the chip: a batch, or single frames on the context's lanes
        // two rounds of the 4096 resident waves are enough: longer segments recompute less than a third round balances (round 4, same box:
        // 1080p -> 1440p 93.3 k frames/s at 64 rows (14,720 waves), 99.2 k at 96 (9,600), 85.3 k at 192 (5,120); 4K -> 1440p flat from 48 to 192)
        const long want = n_frames > 1 ? 8192 : 4096;
        for (int cand : {24, 16, 12, 8, 6, 4, 3, 2})
            if ((long)q.n_strips * ((S.out_h + cand * PB - 1) / (cand * PB)) * side >= want || cand == 2) { seg = cand * PB; break; }
    }
    seg = std::max(PB, (seg / PB) * PB);
    q.seg_rows = seg;
    const int n_segs = (S.out_h + seg - 1) / seg;
    const bool fastepi = epik == EPI_DITHER8, lut = tail_has_table(tailk);
    // waves per workgroup: the tables exist once per workgroup, the A slice once per wave.  8 measured best where LDS leaves the choice
    // (up1440: 84.9 k frames/s against 83.2 k at 12 and 78.6 k at 16 — a workgroup's slot frees only when its slowest wave is done, and two
    // 8-wave workgroups fill a CU's SIMDs like one 16-wave workgroup); wide source windows (downscales) take what puts most waves on a CU
    static const int waves_env = EnvInt("MPCVR_PERIOD_WAVES", 0);
    int waves = 8;
    if (waves_env >= 1 && waves_env <= kPeriodMaxThreads / 64) waves = waves_env;
    else {
        int best_per_cu = 0;
        for (int w : {8, 7, 6, 5, 4}) {
            const size_t l = PeriodLds(S, fastepi, lut, w);
            if (l > DeviceLdsLimit()) continue;
            const int per_cu = std::min((int)(DeviceLdsLimit() / l) * w, 16);
            if (per_cu > best_per_cu) { best_per_cu = per_cu; waves = w; }
        }
    }
    while (waves > 1 && PeriodLds(S, fastepi, lut, waves) > DeviceLdsLimit()) waves--;
    const long items = (long)q.n_strips * n_segs * n_frames;
    if (items < 512L * waves) waves = (int)std::max<long>(1, std::min<long>(waves, items / 512));
    const size_t lds = PeriodLds(S, fastepi, lut, waves);
    const dim3 grid((q.n_strips * n_segs + waves - 1) / waves, 1, n_frames), block(64 * waves, 1, 1);
    if (S.per_P == 4 && S.per_Q == 3) return LaunchFusedPeriodPQ<4, 3>(a, q, S.per_nt, tailk, srck, epik, grid, block, lds, frames_dev, single, s);
    if (S.per_P == 3 && S.per_Q == 2) return LaunchFusedPeriodPQ<3, 2>(a, q, S.per_nt, tailk, srck, epik, grid, block, lds, frames_dev, single, s);
    if (S.per_P == 2 && S.per_Q == 3) return LaunchFusedPeriodPQ<2, 3>(a, q, S.per_nt, tailk, srck, epik, grid, block, lds, frames_dev, single, s);
    if (S.per_P == 1 && S.per_Q == 2) return LaunchFusedPeriodPQ<1, 2>(a, q, S.per_nt, tailk, srck, epik, grid, block, lds, frames_dev, single, s);
    if (S.per_P == 3 && S.per_Q == 1) return LaunchFusedPeriodPQ<3, 1>(a, q, S.per_nt, tailk, srck, epik, grid, block, lds, frames_dev, single, s);
    return hipErrorNotSupported;
}

}  // namespace mpcvr
