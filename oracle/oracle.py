"""ctypes loader for the CPU oracle (oracle/mpcvr_oracle.c) and, when built, the real-reference
csputils library (oracle/_ref/libref_csputils.so).

TEST INFRASTRUCTURE ONLY: imported by tests/ (the suite and its checker scripts under tests/tools/: fuzz, diagnostics),
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package (videorenderer_amd/) or tools/.
"""
import ctypes as C
import os
import subprocess

import numpy as np



def host_cpus():
    """CPUs this process may actually use: logical CPUs, capped by the affinity mask and the cgroup's CPU quota (a 256-CPU box may
    grant 16; OpenMP's default of one thread per logical CPU is then throttled to a fraction of a core each)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:                                            # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, -(-int(q) // int(per))))
    except Exception:
        try:                                        # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = max(1, min(n, -(-q // per)))
        except Exception:
            pass
    return n


# the oracle and the reference-text harness are OpenMP code: tell their runtime before it starts (the libraries are loaded later)
os.environ.setdefault("OMP_NUM_THREADS", str(host_cpus()))

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libmpcvr_oracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libref_csputils.so")
REFERENCE_ROOT = "/root/reference"
DITHER_PATH = os.path.join(os.path.dirname(HERE), "videorenderer_amd", "data", "dither32x32float16.bin")

# ColorFormat_t values (Source/Helper.h:86-127)
CF = dict(NV12=1, P010=2, P016=3, P210=6, P216=7, YV12=14, YV16=15, YV24=16,
          YUV420P8=17, YUV422P8=18, YUV444P8=19, YUV420P10=20, YUV420P16=21,
          YUV422P10=22, YUV422P16=23, YUV444P10=24, YUV444P16=25,
          YUY2=4, UYVY=5, Y210=8, Y216=9, V210=10, AYUV=11, Y410=12, Y416=13,
          GBRP8=26, GBRP10=27, GBRP16=28, Y8=37, Y10=38, Y16=39,
          RGB24=29, XRGB32=30, ARGB32=31, r210=32, RGB48=33, BGR48=34, BGRA64=35, B64A=36)


class OrcDoviCurve(C.Structure):
    _fields_ = [
        ("num_pivots", C.c_uint8), ("mapping_idc", C.c_uint8 * 8), ("poly_order", C.c_uint8 * 8),
        ("mmr_order", C.c_uint8 * 8), ("pivots", C.c_uint16 * 9),
        ("poly_coef", (C.c_int64 * 3) * 8), ("mmr_constant", C.c_int64 * 8), ("mmr_coef", ((C.c_int64 * 7) * 3) * 8),
    ]


class OrcDoviL2(C.Structure):
    _fields_ = [(n, C.c_uint16) for n in ("target_max_pq", "trim_slope", "trim_offset", "trim_power",
                                          "trim_chroma_weight", "trim_saturation_gain")]


class OrcDovi(C.Structure):
    _fields_ = [
        ("bl_bit_depth", C.c_uint8), ("coef_log2_denom", C.c_uint8), ("source_max_pq", C.c_uint16),
        ("l1_present", C.c_uint8), ("l3_present", C.c_uint8),
        ("l1_min_pq", C.c_uint16), ("l1_max_pq", C.c_uint16), ("l1_avg_pq", C.c_uint16),
        ("l3_min_pq_offset", C.c_uint16), ("l3_max_pq_offset", C.c_uint16), ("l3_avg_pq_offset", C.c_uint16),
        ("n_l2", C.c_uint32), ("l2", OrcDoviL2 * 32),
        ("ycc_to_rgb_matrix", C.c_double * 9), ("ycc_to_rgb_offset", C.c_double * 3), ("rgb_to_lms_matrix", C.c_double * 9),
        ("curves", OrcDoviCurve * 3),
    ]


class OrcDoviCb(C.Structure):
    _fields_ = [("pivots", C.c_float * 7), ("coeffs", (C.c_float * 4) * 8), ("mmr", (C.c_float * 4) * 48),
                ("methods", C.c_uint32), ("mmr_single", C.c_uint32), ("min_order", C.c_uint32), ("max_order", C.c_uint32)]


def fill_dovi(st, d):
    """Fill a (Orc|mpcvr) dovi ctypes struct from the plain dict videorenderer_amd.synth.dovi_metadata() returns."""
    for k in ("bl_bit_depth", "coef_log2_denom", "source_max_pq", "l1_present", "l3_present", "l1_min_pq", "l1_max_pq",
              "l1_avg_pq", "l3_min_pq_offset", "l3_max_pq_offset", "l3_avg_pq_offset"):
        setattr(st, k, int(d.get(k, 0)))
    l2 = d.get("l2", [])
    st.n_l2 = len(l2)
    for i, e in enumerate(l2):
        for k, v in e.items():
            setattr(st.l2[i], k, int(v))
    for k in ("ycc_to_rgb_matrix", "ycc_to_rgb_offset", "rgb_to_lms_matrix"):
        getattr(st, k)[:] = [float(x) for x in d[k]]
    for c, cv in enumerate(d["curves"]):
        o = st.curves[c]
        o.num_pivots = len(cv["pivots"])
        o.pivots[:len(cv["pivots"])] = [int(x) for x in cv["pivots"]]
        for i, piece in enumerate(cv["pieces"]):
            if "poly" in piece:
                o.mapping_idc[i] = 0
                o.poly_order[i] = piece["order"]
                for j, x in enumerate(piece["poly"]):
                    o.poly_coef[i][j] = int(x)
            else:
                o.mapping_idc[i] = 1
                o.mmr_order[i] = piece["order"]
                o.mmr_constant[i] = int(piece["constant"])
                for j, row in enumerate(piece["mmr"]):
                    for k, x in enumerate(row):
                        o.mmr_coef[i][j][k] = int(x)
    return st


class OrcParams(C.Structure):
    _fields_ = [
        ("cformat", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
        ("src_rect", C.c_int32 * 4), ("exfmt", C.c_uint32),
        ("iTexFormat", C.c_int32), ("iChromaScaling", C.c_int32),
        ("iUpscaling", C.c_int32), ("iDownscaling", C.c_int32),
        ("bInterpolateAt50pct", C.c_int32), ("bUseDither", C.c_int32),
        ("bConvertToSdr", C.c_int32), ("iSDRDisplayNits", C.c_int32),
        ("output_format", C.c_int32),
        ("brightness", C.c_float), ("contrast", C.c_float), ("hue", C.c_float), ("saturation", C.c_float),
        ("window_w", C.c_int32), ("window_h", C.c_int32), ("video_rect", C.c_int32 * 4),
        ("flags", C.c_uint32),
        ("blend_deint", C.c_int32),
        ("rotation", C.c_int32), ("flip", C.c_int32),
        ("hdr_output", C.c_int32), ("hdr_tonemap_type", C.c_int32),
        ("hdr_display_max_nits", C.c_float), ("hdr_min_mastering", C.c_float), ("hdr_max_mastering", C.c_float),
        ("hdr_max_cll", C.c_float), ("hdr_max_fall", C.c_float),
        ("dovi", C.POINTER(OrcDovi)),
    ]


def make_extfmt(chroma=0, nominal_range=0, matrix=0, lighting=0, primaries=0, transfer=0, sample_format=0):
    """Pack a DXVA2_ExtendedFormat.value (dxva2api.h bit layout, LSB first)."""
    return ((sample_format & 0xff) | ((chroma & 0xf) << 8) | ((nominal_range & 0x7) << 12) |
            ((matrix & 0x7) << 15) | ((lighting & 0xf) << 18) | ((primaries & 0x1f) << 22) |
            ((transfer & 0x1f) << 27))


def build(ref=True, quiet=True):
    """Compile the oracle (always) and oracle/_ref (only when /root/reference is mounted)."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.check_call(["make", "-C", HERE], stdout=out)
    if ref and os.path.isdir(os.path.join(REFERENCE_ROOT, "Source")):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=out)
        # the reference's own HLSL (Shaders/*.hlsl + the text Source/Shaders.cpp generates) compiled for the CPU
        subprocess.check_call(["make", "-C", HERE, "ref-hlsl"], stdout=out)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.orc_params_default.argtypes = [C.POINTER(OrcParams)]
        L.orc_specify_extfmt.restype = C.c_uint32
        L.orc_specify_extfmt.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.orc_color_matrix.argtypes = [C.POINTER(OrcParams), fp]
        L.orc_csp_matrix.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_int, fp, fp]
        L.orc_gamut_matrix.argtypes = [C.c_int, C.c_int, fp]
        L.orc_gamut_2020_to_709.argtypes = [fp]
        L.orc_luminance_scale.restype = C.c_float
        L.orc_luminance_scale.argtypes = [C.c_int]
        for name in ("orc_st2084_to_linear", "orc_linear_to_st2084"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_float, C.c_float]
        L.orc_hable.restype = C.c_float
        L.orc_hable.argtypes = [C.c_float]
        L.orc_hlg_to_linear.argtypes = [fp]
        L.orc_tonemap_hable.argtypes = [fp]
        L.orc_hdr_tail.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_float]
        L.orc_upscale_weights.restype = C.c_int
        L.orc_upscale_weights.argtypes = [C.c_int, C.c_float, fp]
        L.orc_downscale_filter.restype = C.c_float
        L.orc_downscale_filter.argtypes = [C.c_int, C.c_float, fp]
        L.orc_axis_taps.restype = C.c_int
        L.orc_axis_taps.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int,
                                    C.POINTER(C.c_int32), fp, fp]
        L.orc_half_round.restype = C.c_float
        L.orc_half_round.argtypes = [C.c_float]
        L.orc_frame_bytes.restype = C.c_size_t
        L.orc_frame_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_process.restype = C.c_int
        L.orc_process.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_convert_only.restype = C.c_int
        L.orc_convert_only.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_int, C.c_void_p]
        L.orc_dovi_pack_curves.argtypes = [C.POINTER(OrcDovi), C.POINTER(OrcDoviCb), C.POINTER(C.c_int)]
        L.orc_dovi_reshape.argtypes = [C.POINTER(OrcDoviCb), C.c_int, fp]
        L.orc_dovi_lms_matrix.argtypes = [C.POINTER(OrcDovi), fp]
        L.orc_dovi_l2_constants.restype = C.c_int
        L.orc_dovi_l2_constants.argtypes = [C.POINTER(OrcDovi), C.c_int, fp]
        L.orc_dovi_l1_nits.restype = C.c_int
        L.orc_dovi_l1_nits.argtypes = [C.POINTER(OrcDovi), C.POINTER(C.c_uint32)]
        L.orc_set_pow_ulp_bias.argtypes = [C.c_int]
        L.orc_set_pow_ulp_noise.argtypes = [C.c_int, C.c_uint32]
        L.orc_set_tex_ulp_bias.argtypes = [C.c_int]
        L.orc_set_tonemap_input_bias.argtypes = [C.c_int, C.c_int, C.c_uint32]
        L.orc_set_pow_log2_noise.argtypes = [C.c_int, C.c_uint32]
        L.orc_set_convert_output_bias.argtypes = [C.c_int, C.c_int, C.c_uint32]
        L.orc_eval_transcendental.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_eval_dovi_tail.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, fp, fp, C.c_int, C.c_float]
        L.orc_hdr10_params.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_uint32)]
        L.orc_specify_extfmt.restype = C.c_uint32
        L.orc_specify_extfmt.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.orc_correction_pass.restype = C.c_int
        L.orc_correction_pass.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_correction_matrices.argtypes = [fp, fp, fp]
        L.orc_error_diffusion.restype = C.c_int
        L.orc_error_diffusion.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads(min(int(L.orc_num_threads()), host_cpus()))
        _lib = L
    return _lib


def ref():
    """The real csputils.cpp (compiled from /root/reference) or None when not built."""
    global _ref
    if _ref is None and os.path.exists(REF_PATH):
        R = C.CDLL(REF_PATH)
        fp = C.POINTER(C.c_float)
        R.ref_csp_matrix.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_int, fp, fp]
        R.ref_gamut_matrix.argtypes = [C.c_int, C.c_int, fp]
        _ref = R
    return _ref


def default_params(**kw):
    p = OrcParams()
    lib().orc_params_default(C.byref(p))
    set_params(p, **kw)
    return p


def set_params(p, **kw):
    for k, v in kw.items():
        if k in ("src_rect", "video_rect"):
            getattr(p, k)[:] = list(v)
        elif k == "dovi":
            if v is None:
                p.dovi = None
                p._dovi_keep = None
            else:
                st = fill_dovi(OrcDovi(), v) if isinstance(v, dict) else v
                p._dovi_keep = st               # keep the pointee alive
                p.dovi = C.pointer(st)
        else:
            setattr(p, k, v)
    return p


def dither_table():
    return np.fromfile(DITHER_PATH, dtype=np.uint16)


def frame_bytes(cformat, w, h):
    pitch = C.c_int(0)
    n = lib().orc_frame_bytes(cformat, w, h, C.byref(pitch))
    return int(n), pitch.value


def color_matrix(p):
    out = (C.c_float * 12)()
    rc = lib().orc_color_matrix(C.byref(p), out)
    assert rc == 0
    return np.array(out, dtype=np.float32)


def correction_pass(kind, src, src_fmt, dst_fmt, sdr_nits=125):
    """One m_pPSCorrection shader over a (h, w) uint32 surface; fmt 8 = B8G8R8A8, 10 = R10G10B10A2.  Returns (h, w) uint32."""
    src = np.ascontiguousarray(src, dtype=np.uint32)
    h, w = src.shape
    dst = np.zeros((h, w), dtype=np.uint32)
    rc = lib().orc_correction_pass(kind, src.ctypes.data, w * 4, src_fmt, dst.ctypes.data, w * 4, dst_fmt, w, h, sdr_nits)
    assert rc == 0, rc
    return dst


def correction_matrices():
    a, b, g = (C.c_float * 16)(), (C.c_float * 16)(), (C.c_float * 9)()
    lib().orc_correction_matrices(a, b, g)
    return np.array(a, np.float32), np.array(b, np.float32), np.array(g, np.float32)


def process(p, frame, pitch, dither=None, dst=None):
    """Run the whole path; returns (window_h, window_w, 4) uint8 (or (h,w) uint32 for RGB10A2)."""
    frame = np.ascontiguousarray(frame).view(np.uint8).ravel()
    if dither is None:
        dither = dither_table()
    if dst is None:
        dst = np.zeros((p.window_h, p.window_w, 4), dtype=np.uint8)
    rc = lib().orc_process(C.byref(p), frame.ctypes.data, pitch, dither.ctypes.data,
                           dst.ctypes.data, p.window_w * 4)
    if rc != 0:
        raise RuntimeError(f"orc_process failed: {rc}")
    return dst


def error_diffusion(img10, rect=None, dst=None):
    """EXTENSION (bUseDither = 2; no reference counterpart): the serial error-diffusion model over rect = (x0, y0, x1, y1) of an
    (h, w) uint32 R10G10B10A2 image; returns (h, w, 4) uint8 B8G8R8A8, untouched (zero) outside the rect."""
    img10 = np.ascontiguousarray(img10, dtype=np.uint32)
    h, w = img10.shape
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, w, h)
    if dst is None:
        dst = np.zeros((h, w, 4), dtype=np.uint8)
    rc = lib().orc_error_diffusion(img10.ctypes.data, w * 4, dst.ctypes.data, w * 4, int(x0), int(y0), int(x1), int(y1))
    if rc != 0:
        raise RuntimeError(f"orc_error_diffusion failed: {rc}")
    return dst


def process_errdiff(p, frame, pitch):
    """bUseDither = 2 as the product defines it: the frame as a 10-bit swap chain would receive it (no final pass), then the serial
    error diffusion inside video rect ∩ window."""
    import copy
    keep = (p.output_format,)
    p.output_format = 1             # ORC_OUT_RGB10A2
    try:
        img10 = process(p, frame, pitch).view(np.uint32)[:, :, 0]
    finally:
        p.output_format = keep[0]
    x0, y0 = max(p.video_rect[0], 0), max(p.video_rect[1], 0)
    x1, y1 = min(p.video_rect[2], p.window_w), min(p.video_rect[3], p.window_h)
    return error_diffusion(img10, (x0, y0, x1, y1))


def process_with_pow_bias(p, frame, pitch, bias, dst=None, seed=0):
    """process() with every pow() of the HDR / Dolby Vision chains answering `bias` ulps off (the sensitivity probe); seed != 0: each
    call by its own hash-drawn amount in [-bias, +bias] instead."""
    if seed:
        lib().orc_set_pow_ulp_noise(int(abs(bias)), int(seed))
    else:
        lib().orc_set_pow_ulp_bias(int(bias))
    try:
        return process(p, frame, pitch, dst=dst)
    finally:
        lib().orc_set_pow_ulp_bias(0)


def process_with_tonemap_input_bias(p, frame, pitch, bias, channel=-1, seed=0, dst=None):
    """process() with the texture in front of the HDR10 tone-mapping step `bias` codes off (channel 0..2 or -1 = all; seed != 0: every channel
    of every texel by its own draw in [-|bias|, +|bias|]) — what a one-code difference of that intermediate becomes behind the operator."""
    lib().orc_set_tonemap_input_bias(int(bias), int(channel), int(seed))
    try:
        return process(p, frame, pitch, dst=dst)
    finally:
        lib().orc_set_tonemap_input_bias(0, -1, 0)


def process_with_log2_noise(p, frame, pitch, amplitude, seed=0, dst=None):
    """process() with log2(x) inside every pow() up to `amplitude` ulps off (seed 0: every call by that amount, signed) — where a GPU's pow errs."""
    lib().orc_set_pow_log2_noise(int(amplitude), int(seed))
    try:
        return process(p, frame, pitch, dst=dst)
    finally:
        lib().orc_set_pow_log2_noise(0, 0)


def process_with_convert_output_bias(p, frame, pitch, bias, channel=-1, seed=0, dst=None):
    """process() with m_TexConvertOutput `bias` codes off (channel 0..2, -1 = all; seed != 0: per texel and channel, drawn in [-|bias|, +|bias|])."""
    lib().orc_set_convert_output_bias(int(bias), int(channel), int(seed))
    try:
        return process(p, frame, pitch, dst=dst)
    finally:
        lib().orc_set_convert_output_bias(0, -1, 0)


def eval_transcendental(fn, x, y=None):
    """The oracle's defined transcendentals (crmath.h) over float32 arrays: fn = 'log2' | 'exp2' | 'exp' | 'pow' | 'sin' | 'cos'."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y if y is not None else x, dtype=np.float32)
    out = np.empty_like(x)
    lib().orc_eval_transcendental({"log2": 0, "exp2": 1, "exp": 2, "pow": 3, "sin": 4, "cos": 5}[fn], x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size)
    return out


def process_with_tex_bias(p, frame, pitch, bias):
    """process() with every draw's interpolated texture coordinate `bias` ulps off the modelled rasteriser's (sensitivity probe)."""
    lib().orc_set_tex_ulp_bias(int(bias))
    try:
        return process(p, frame, pitch)
    finally:
        lib().orc_set_tex_ulp_bias(0)


def convert_only(p, frame, pitch):
    frame = np.ascontiguousarray(frame).view(np.uint8).ravel()
    r = list(p.src_rect)
    if not any(r):
        r = [0, 0, p.width, p.height]
    out = np.zeros((r[3] - r[1], r[2] - r[0], 4), dtype=np.float32)
    fmt = lib().orc_convert_only(C.byref(p), frame.ctypes.data, pitch, out.ctypes.data)
    if fmt < 0:
        raise RuntimeError(f"orc_convert_only failed: {fmt}")
    return out, fmt
