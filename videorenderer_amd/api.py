"""Host-side mirror of the reference's video-processor interface over the C-ABI (include/mpcvr.h).

`VideoProcessor` follows CVideoProcessor / CDX11VideoProcessor (Source/VideoProcessor.h:171-236,
Source/DX11VideoProcessor.h:256-384): same method names, argument meaning and HRESULT-style error
behaviour, so the parity tests read like calls into the reference.  All arithmetic happens in
libmpcvr.so (hand-written HIP for gfx950); there is NO CPU fallback — a missing library or a missing
GPU raises.  PyTorch only supplies device memory and streams.
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmpcvr.so")

S_OK, S_FALSE = 0, 1
E_FAIL = -2147467259          # 0x80004005
E_POINTER = -2147467261       # 0x80004003
E_INVALIDARG = -2147024809    # 0x80070057
E_UNEXPECTED = -2147418113    # 0x8000FFFF
E_NOTIMPL = -2147467263       # 0x80004001
E_OUTOFMEMORY = -2147024882   # 0x8007000E
E_NOT_VALID_STATE = -2147019873  # 0x8007139F

# ColorFormat_t (Source/Helper.h:86-127)
CF_NV12, CF_P010, CF_P016 = 1, 2, 3
CF_YUY2, CF_UYVY = 4, 5
CF_P210, CF_P216 = 6, 7
CF_Y210, CF_Y216, CF_V210, CF_AYUV, CF_Y410, CF_Y416 = 8, 9, 10, 11, 12, 13
CF_GBRP8, CF_GBRP10, CF_GBRP16 = 26, 27, 28
CF_Y8, CF_Y10, CF_Y16 = 37, 38, 39
CF_RGB24, CF_XRGB32, CF_ARGB32, CF_r210, CF_RGB48, CF_BGR48, CF_BGRA64, CF_B64A = 29, 30, 31, 32, 33, 34, 35, 36
CF_YV12, CF_YV16, CF_YV24 = 14, 15, 16
CF_YUV420P8, CF_YUV422P8, CF_YUV444P8 = 17, 18, 19
CF_YUV420P10, CF_YUV420P16, CF_YUV422P10, CF_YUV422P16, CF_YUV444P10, CF_YUV444P16 = 20, 21, 22, 23, 24, 25

# Settings enums (Source/IVideoRenderer.h:25-72)
TEXFMT_AUTOINT, TEXFMT_8INT, TEXFMT_10INT, TEXFMT_16FLOAT = 0, 8, 10, 16
CHROMA_Nearest, CHROMA_Bilinear, CHROMA_CatmullRom = 0, 1, 2
UPSCALE_Nearest, UPSCALE_Mitchell, UPSCALE_CatmullRom, UPSCALE_Lanczos2, UPSCALE_Lanczos3, UPSCALE_Jinc2 = range(6)
UPSCALE_Spline36_EXT = 6          # extension: not a reference setting (IVideoRenderer.h:54-62)
DOWNSCALE_Box, DOWNSCALE_Bilinear, DOWNSCALE_Hamming, DOWNSCALE_Bicubic, DOWNSCALE_BicubicSharp, DOWNSCALE_Lanczos = range(6)
OUT_BGRA8, OUT_RGB10A2 = 0, 1
DITHER_None, DITHER_Ordered, DITHER_ErrorDiffusion_EXT = 0, 1, 2     # bUseDither; 2 is an extension (error diffusion, include/mpcvr.h)
FLAG_LANCZOS3_FIXED, FLAG_NO_FUSED, FLAG_NO_LUT, FLAG_NO_FAST_CONVERT, FLAG_FUSED_VALU, FLAG_FUSED_MFMA, FLAG_NO_STRIP = 1, 2, 4, 8, 16, 32, 64
FLAG_NO_PERIOD = 128
FLAG_FORCE_PERIOD = 256
FLAG_NO_FRAME_LANES = 512     # mpcvr_process strictly frame after frame (default: two overlapping lanes when the context owns its stream)
MEM_HOST, MEM_DEVICE, MEM_HOST_PINNED = 0, 1, 2
PROCAMP_BRIGHTNESS, PROCAMP_CONTRAST, PROCAMP_HUE, PROCAMP_SATURATION = 1, 2, 4, 8

# DXVA2_ExtendedFormat codes used by the reference (Helper.cpp:1215-1223)
CHROMA_LOC_MPEG1, CHROMA_LOC_MPEG2, CHROMA_LOC_COSITED = 1, 5, 7
RANGE_0_255, RANGE_16_235 = 1, 2
MATRIX_BT709, MATRIX_BT601, MATRIX_SMPTE240M, MATRIX_BT2020, MATRIX_YCGCO = 1, 2, 3, 4, 7
PRIM_BT709, PRIM_BT2020 = 2, 9
TRC_22, TRC_709, TRC_SRGB, TRC_PQ, TRC_HLG = 4, 5, 7, 15, 16


def make_extfmt(chroma=0, nominal_range=0, matrix=0, lighting=0, primaries=0, transfer=0, sample_format=0):
    """Pack DXVA2_ExtendedFormat.value (dxva2api.h bit layout, LSB first)."""
    return ((sample_format & 0xff) | ((chroma & 0xf) << 8) | ((nominal_range & 0x7) << 12) |
            ((matrix & 0x7) << 15) | ((lighting & 0xf) << 18) | ((primaries & 0x1f) << 22) |
            ((transfer & 0x1f) << 27))


class Settings(C.Structure):
    """mpcvr_settings == the subset of Settings_t (IVideoRenderer.h:104-135) that reaches this path."""
    _fields_ = [("iTexFormat", C.c_int32), ("iChromaScaling", C.c_int32), ("iUpscaling", C.c_int32),
                ("iDownscaling", C.c_int32), ("bInterpolateAt50pct", C.c_int32), ("bUseDither", C.c_int32),
                ("bDeintBlend", C.c_int32), ("bConvertToSdr", C.c_int32), ("iSDRDisplayNits", C.c_int32),
                ("output_format", C.c_int32), ("flags", C.c_uint32)]

    def copy(self, **kw):
        s = Settings()
        C.memmove(C.byref(s), C.byref(self), C.sizeof(Settings))
        for k, v in kw.items():
            setattr(s, k, v)
        return s


class Rect(C.Structure):
    _fields_ = [("left", C.c_int32), ("top", C.c_int32), ("right", C.c_int32), ("bottom", C.c_int32)]


class DoviCurve(C.Structure):          # mpcvr_dovi_curve
    _fields_ = [
        ("num_pivots", C.c_uint8), ("mapping_idc", C.c_uint8 * 8), ("poly_order", C.c_uint8 * 8),
        ("mmr_order", C.c_uint8 * 8), ("pivots", C.c_uint16 * 9),
        ("poly_coef", (C.c_int64 * 3) * 8), ("mmr_constant", C.c_int64 * 8), ("mmr_coef", ((C.c_int64 * 7) * 3) * 8),
    ]


class DoviL2(C.Structure):             # mpcvr_dovi_l2
    _fields_ = [(n, C.c_uint16) for n in ("target_max_pq", "trim_slope", "trim_offset", "trim_power",
                                          "trim_chroma_weight", "trim_saturation_gain")]


class DoviMetadata(C.Structure):       # mpcvr_dovi_metadata
    _fields_ = [
        ("bl_bit_depth", C.c_uint8), ("coef_log2_denom", C.c_uint8), ("source_max_pq", C.c_uint16),
        ("l1_present", C.c_uint8), ("l3_present", C.c_uint8),
        ("l1_min_pq", C.c_uint16), ("l1_max_pq", C.c_uint16), ("l1_avg_pq", C.c_uint16),
        ("l3_min_pq_offset", C.c_uint16), ("l3_max_pq_offset", C.c_uint16), ("l3_avg_pq_offset", C.c_uint16),
        ("n_l2", C.c_uint32), ("l2", DoviL2 * 32),
        ("ycc_to_rgb_matrix", C.c_double * 9), ("ycc_to_rgb_offset", C.c_double * 3), ("rgb_to_lms_matrix", C.c_double * 9),
        ("curves", DoviCurve * 3),
    ]

    @classmethod
    def from_dict(cls, d):
        """Build from the plain-dict form synth.dovi_metadata() produces:
        {bl_bit_depth, coef_log2_denom, source_max_pq, [l1_*, l3_*], l2: [{target_max_pq, trim_*}...],
         ycc_to_rgb_matrix[9], ycc_to_rgb_offset[3], rgb_to_lms_matrix[9],
         curves: 3 x {pivots: [...], pieces: [{order, poly: [c0,c1,c2]} | {order, constant, mmr: [[7]...]}]}}"""
        st = cls()
        for k in ("bl_bit_depth", "coef_log2_denom", "source_max_pq", "l1_present", "l3_present", "l1_min_pq",
                  "l1_max_pq", "l1_avg_pq", "l3_min_pq_offset", "l3_max_pq_offset", "l3_avg_pq_offset"):
            setattr(st, k, int(d.get(k, 0)))
        l2 = d.get("l2", [])
        st.n_l2 = len(l2)
        for i, e in enumerate(l2[:32]):
            for k, v in e.items():
                setattr(st.l2[i], k, int(v))
        for k in ("ycc_to_rgb_matrix", "ycc_to_rgb_offset", "rgb_to_lms_matrix"):
            getattr(st, k)[:] = [float(x) for x in d[k]]
        for c, cv in enumerate(d["curves"]):
            o = st.curves[c]
            o.num_pivots = cv.get("num_pivots", len(cv["pivots"]))
            o.pivots[:len(cv["pivots"])] = [int(x) for x in cv["pivots"]]
            for i, piece in enumerate(cv["pieces"]):
                if "poly" in piece:
                    o.mapping_idc[i] = piece.get("idc", 0)
                    o.poly_order[i] = piece["order"]
                    for j, x in enumerate(piece["poly"]):
                        o.poly_coef[i][j] = int(x)
                else:
                    o.mapping_idc[i] = piece.get("idc", 1)
                    o.mmr_order[i] = piece["order"]
                    o.mmr_constant[i] = int(piece["constant"])
                    for j, row in enumerate(piece["mmr"]):
                        for k, x in enumerate(row):
                            o.mmr_coef[i][j][k] = int(x)
        return st


class MpcvrError(RuntimeError):
    def __init__(self, hr, msg):
        super().__init__(f"HRESULT 0x{hr & 0xffffffff:08X}: {msg}")
        self.hr = hr


EXPORTS = [
    "mpcvr_settings_default", "mpcvr_create", "mpcvr_destroy", "mpcvr_set_stream", "mpcvr_synchronize",
    "mpcvr_set_input", "mpcvr_set_video_rect", "mpcvr_set_window_rect", "mpcvr_set_rotation", "mpcvr_set_error_diffusion_patience", "mpcvr_set_flip", "mpcvr_set_sample_format", "mpcvr_set_hdr_output", "mpcvr_set_hdr_metadata",
    "mpcvr_set_dovi_metadata", "mpcvr_plan_dovi", "mpcvr_correction_pass", "mpcvr_plan_correction_matrices",
    "mpcvr_configure", "mpcvr_set_procamp", "mpcvr_copy_sample", "mpcvr_process", "mpcvr_process_frames", "mpcvr_render",
    "mpcvr_get_backbuffer", "mpcvr_get_current_image", "mpcvr_get_displayed_image", "mpcvr_flush", "mpcvr_reset", "mpcvr_process_batch", "mpcvr_process_batch_dovi",
    "mpcvr_get_param_blob", "mpcvr_set_param_blob", "mpcvr_broadcast_param_blob_begin", "mpcvr_broadcast_param_blob_end",
    "mpcvr_broadcast_param_blob", "mpcvr_get_color_matrix", "mpcvr_get_extfmt",
    "mpcvr_get_frame_bytes", "mpcvr_get_path_info", "mpcvr_get_last_batch_info", "mpcvr_last_error", "mpcvr_version",
    "mpcvr_get_last_process_ms", "mpcvr_get_last_timings",
    "mpcvr_plan_frame_layout", "mpcvr_plan_color_matrix", "mpcvr_plan_gamut_2020_to_709", "mpcvr_plan_pq_lut",
    "mpcvr_plan_upscale_weights", "mpcvr_plan_axis_taps", "mpcvr_plan_describe", "mpcvr_plan_final_pass_multiplier",
    "mpcvr_plan_strip", "mpcvr_plan_pq_eotf_table", "mpcvr_plan_pq_eotf_lut", "mpcvr_bandwidth_probe", "mpcvr_plan_period", "mpcvr_plan_hdr10_params",
    "mpcvr_eval_transcendental", "mpcvr_eval_transcendental_host", "mpcvr_bandwidth_probe_up2x", "mpcvr_eval_dovi_tail",
]

_lib = None


def load_library():
    """dlopen libmpcvr.so.  Raises (never falls back) when it has not been built.
    Side effect: imports torch first when it is installed (one HIP runtime for both, see below) unless MPCVR_NO_TORCH_IMPORT=1."""
    global _lib
    if _lib is not None:
        return _lib
    # MPCVR_LIB=<path>: load another build of the library instead (same-box A/B runs of tools/ against an older libmpcvr.so; entry
    # points that build lacks are skipped).  Never set in normal use.
    lib_path = os.environ.get("MPCVR_LIB") or LIB_PATH
    if not os.path.exists(lib_path):
        raise RuntimeError(f"{lib_path} is missing: run `python -m videorenderer_amd.build` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    # In a process that also uses torch, torch's bundled HIP runtime must be the one both share: libmpcvr.so resolves
    # libamdhip64 through the dynamic loader, and a second copy of the runtime (loaded first from /opt/rocm) sees no device once
    # torch has initialised its own.  Importing torch first — only if it is installed; the library itself does not need it —
    # makes the load order irrelevant.  A host that never touches torch can skip the (slow) import with MPCVR_NO_TORCH_IMPORT=1.
    if "torch" not in sys.modules and not os.environ.get("MPCVR_NO_TORCH_IMPORT"):
        try:
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(lib_path)
    vp, i32, u32, f = C.c_void_p, C.c_int32, C.c_uint32, C.c_float
    P = C.POINTER
    sig = {
        "mpcvr_settings_default": [P(Settings)],
        "mpcvr_create": [P(Settings), i32, P(vp)],
        "mpcvr_destroy": [vp],
        "mpcvr_set_stream": [vp, vp],
        "mpcvr_synchronize": [vp],
        "mpcvr_set_input": [vp, i32, i32, i32, i32, P(Rect), u32],
        "mpcvr_set_video_rect": [vp, P(Rect)],
        "mpcvr_set_window_rect": [vp, P(Rect)],
        "mpcvr_set_rotation": [vp, i32],
        "mpcvr_set_error_diffusion_patience": [vp, i32],
        "mpcvr_set_flip": [vp, i32],
        "mpcvr_set_sample_format": [vp, i32],
        "mpcvr_set_hdr_output": [vp, i32, i32, f],
        "mpcvr_set_hdr_metadata": [vp, f, f, f, f],
        "mpcvr_set_dovi_metadata": [vp, P(DoviMetadata)],
        "mpcvr_plan_dovi": [P(DoviMetadata), i32, P(f), P(i32), P(f), P(f), P(i32), P(u32), P(i32)],
        "mpcvr_correction_pass": [i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp],
        "mpcvr_plan_correction_matrices": [P(f), P(f), P(f)],
        "mpcvr_configure": [vp, P(Settings)],
        "mpcvr_set_procamp": [vp, u32, f, f, f, f],
        "mpcvr_copy_sample": [vp, vp, i32, i32],
        "mpcvr_process": [vp, vp, i32, P(Rect), P(Rect), i32],
        "mpcvr_render": [vp, i32],
        "mpcvr_process_frames": [vp, i32, P(C.c_void_p), i32, i32, P(C.c_void_p), i32],
        "mpcvr_get_backbuffer": [vp, P(vp), P(i32), P(i32), P(i32)],
        "mpcvr_get_displayed_image": [vp, C.c_void_p, P(C.c_size_t), i32, P(i32), P(i32), P(i32)],
        "mpcvr_get_current_image": [vp, vp, P(C.c_size_t)],
        "mpcvr_flush": [vp],
        "mpcvr_reset": [vp],
        "mpcvr_process_batch": [vp, i32, P(vp), P(vp), i32],
        "mpcvr_process_batch_dovi": [vp, i32, P(vp), P(vp), i32, P(DoviMetadata)],
        "mpcvr_get_param_blob": [vp, vp, P(C.c_size_t)],
        "mpcvr_set_param_blob": [vp, vp, C.c_size_t],
        "mpcvr_broadcast_param_blob_begin": [vp, vp, i32, i32],
        "mpcvr_broadcast_param_blob_end": [vp],
        "mpcvr_broadcast_param_blob": [vp, vp, i32, i32],
        "mpcvr_get_color_matrix": [vp, P(f)],
        "mpcvr_get_extfmt": [vp, P(u32)],
        "mpcvr_get_frame_bytes": [vp, P(C.c_size_t), P(i32)],
        "mpcvr_get_path_info": [vp, C.c_char_p, C.c_size_t],
        "mpcvr_get_last_batch_info": [vp, C.c_char_p, C.c_size_t],
        "mpcvr_get_last_process_ms": [vp, P(f)],
        "mpcvr_get_last_timings": [vp, P(f), P(f), P(f), P(f)],
        "mpcvr_plan_frame_layout": [i32, i32, i32, P(i32), P(C.c_size_t)],
        "mpcvr_plan_color_matrix": [i32, i32, i32, u32, f, f, f, f, P(f), P(u32)],
        "mpcvr_plan_gamut_2020_to_709": [P(f)],
        "mpcvr_plan_pq_lut": [f, P(f)],
        "mpcvr_plan_final_pass_multiplier": [i32, i32, P(u32)],
        "mpcvr_plan_upscale_weights": [i32, f, P(f)],
        "mpcvr_plan_axis_taps": [i32, i32, i32, i32, i32, i32, u32, i32, P(i32), P(f), P(f), P(i32), P(i32)],
        "mpcvr_plan_strip": [i32, i32, i32, i32, i32, i32, i32, i32, u32, P(i32), P(i32), P(i32), P(i32), P(f), P(i32), P(f)],
        "mpcvr_plan_hdr10_params": [f, f, f, f, f, i32, P(u32)],
        "mpcvr_plan_period": [i32, i32, i32, i32, i32, u32, P(i32), P(i32), P(f), P(f), P(i32), P(i32)],
        "mpcvr_plan_pq_eotf_table": [P(f), i32, P(i32)],
        "mpcvr_plan_pq_eotf_lut": [P(f)],
        "mpcvr_bandwidth_probe": [C.c_void_p, C.c_void_p, C.c_size_t, i32, C.c_void_p],
        "mpcvr_bandwidth_probe_up2x": [i32, i32, P(C.c_void_p), P(C.c_void_p), i32, i32, i32, i32, C.c_void_p],
        "mpcvr_eval_transcendental": [i32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
        "mpcvr_eval_dovi_tail": [i32, C.c_void_p, C.c_void_p, C.c_size_t, P(f), P(f), i32, f, C.c_void_p],
        "mpcvr_eval_transcendental_host": [i32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
        "mpcvr_plan_describe": [P(Settings), i32, i32, i32, P(Rect), i32, i32, C.c_char_p, C.c_size_t],
    }
    for name, args in sig.items():
        if lib_path != LIB_PATH and not hasattr(L, name):
            continue
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i32
    L.mpcvr_last_error.argtypes = [vp]
    L.mpcvr_last_error.restype = C.c_char_p
    L.mpcvr_version.argtypes = []
    L.mpcvr_version.restype = C.c_char_p
    _lib = L
    return L


def default_settings(**kw):
    s = Settings()
    load_library().mpcvr_settings_default(C.byref(s))
    for k, v in kw.items():
        setattr(s, k, v)
    return s


# ---- host-side parameter maths (no context, no GPU) ------------------------------------------------
def plan_frame_layout(cformat, width, height):
    pitch, nbytes = C.c_int32(), C.c_size_t()
    hr = load_library().mpcvr_plan_frame_layout(cformat, width, height, C.byref(pitch), C.byref(nbytes))
    if hr < 0:
        raise MpcvrError(hr, "mpcvr_plan_frame_layout")
    return nbytes.value, pitch.value


def plan_color_matrix(cformat, rect_w, rect_h, extfmt=0, brightness=0.0, contrast=1.0, hue=0.0, saturation=1.0):
    out = (C.c_float * 12)()
    ex = C.c_uint32()
    hr = load_library().mpcvr_plan_color_matrix(cformat, rect_w, rect_h, extfmt, brightness, contrast, hue,
                                                saturation, out, C.byref(ex))
    if hr < 0:
        raise MpcvrError(hr, "mpcvr_plan_color_matrix")
    return list(out), ex.value


def plan_gamut_2020_to_709():
    out = (C.c_float * 9)()
    load_library().mpcvr_plan_gamut_2020_to_709(out)
    return list(out)


def plan_pq_lut(lum_scale):
    out = (C.c_float * 4096)()
    load_library().mpcvr_plan_pq_lut(lum_scale, out)
    return list(out)


def plan_final_pass_multiplier(quant, maxv):
    """M of the fused path's integer final pass, (k*M + (j << 14)) >> 24; 0 when not representable."""
    m = C.c_uint32(0)
    load_library().mpcvr_plan_final_pass_multiplier(quant, maxv, C.byref(m))
    return m.value


def plan_dovi(md, display_nits=1000):
    """Host maths of the Dolby Vision path (SetShaderDoviCurves, LMS matrix, level-2 selection, level-1 nits)."""
    st = md if isinstance(md, DoviMetadata) else DoviMetadata.from_dict(md)
    cb = (C.c_float * 705)(); lms = (C.c_float * 9)(); l2k = (C.c_float * 5)(); l1 = (C.c_uint32 * 3)()
    has_mmr, l2on, l1on = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    hr = load_library().mpcvr_plan_dovi(C.byref(st), int(display_nits), cb, C.byref(has_mmr), lms, l2k, C.byref(l2on),
                                        l1, C.byref(l1on))
    if hr:
        raise MpcvrError(hr, "mpcvr_plan_dovi")
    import numpy as np
    return dict(cb=np.array(cb, dtype=np.float32).reshape(3, 235), has_mmr=has_mmr.value,
                lms=np.array(lms, dtype=np.float32), l2k=np.array(l2k, dtype=np.float32), l2_enabled=l2on.value,
                l1_nits=np.array(l1, dtype=np.uint32), l1_present=l1on.value)


CORR_FIX_BT2020, CORR_FIX_YCGCO, CORR_FIXCONVERT_PQ_TO_SDR, CORR_FIXCONVERT_HLG_TO_SDR, CORR_CONVERT_PQ_TO_SDR, CORR_CONVERT_HLG_TO_PQ = range(1, 7)


def plan_correction_matrices():
    """fix_bt2020_matrix, fix_ycgco_matrix (4x4) and convert_matrix_2020_to_709 (3x3) of the correction shaders, in fp32."""
    a, b, g = (C.c_float * 16)(), (C.c_float * 16)(), (C.c_float * 9)()
    hr = load_library().mpcvr_plan_correction_matrices(a, b, g)
    if hr:
        raise MpcvrError(hr, "mpcvr_plan_correction_matrices")
    return list(a), list(b), list(g)


def correction_pass(kind, src, src_pitch, src_fmt, dst, dst_pitch, dst_fmt, w, h, sdr_nits=125, stream=None):
    """One m_pPSCorrection shader (CORR_*) over a device surface of 32-bit texels; fmt: 0 = BGRA8, 1 = RGB10A2."""
    hr = load_library().mpcvr_correction_pass(int(kind), C.c_void_p(_ptr(src)), int(src_pitch), int(src_fmt), C.c_void_p(_ptr(dst)),
                                              int(dst_pitch), int(dst_fmt), int(w), int(h), int(sdr_nits),
                                              C.c_void_p(stream) if stream else None)
    if hr:
        raise MpcvrError(hr, "mpcvr_correction_pass")


def plan_upscale_weights(method, t):
    w = (C.c_float * 6)()
    n = load_library().mpcvr_plan_upscale_weights(method, t, w)
    return list(w)[:n]


def plan_strip(kind_x, method_x, kind_y, method_y, src_w, src_h, out_w, out_h, flags=0):
    """PlanFusedStrip through the C-ABI (no device): dict with the kernel's geometry and its tables as numpy arrays, or None when
    the resize does not fit the arbitrary-ratio fused kernel."""
    import numpy as np
    L = load_library()
    out8 = (C.c_int32 * 8)()
    hr = L.mpcvr_plan_strip(kind_x, method_x, kind_y, method_y, src_w, src_h, out_w, out_h, flags, out8, None, None, None, None, None, None)
    if hr == E_NOTIMPL:
        return None
    if hr != 0:
        raise MpcvrError(hr, "mpcvr_plan_strip")
    nt, pxl, strip_w, ring, acols, strips, lds_wave, _ = list(out8)
    yr = np.zeros((out_h, 2), np.int32); xs = np.zeros((strips, 2), np.int32)
    xi = np.zeros((nt, out_w), np.int32); xw = np.zeros((nt, out_w), np.float32)
    yi = np.zeros((out_h, nt), np.int32); yw = np.zeros((out_h, nt), np.float32)
    as_p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    hr = L.mpcvr_plan_strip(kind_x, method_x, kind_y, method_y, src_w, src_h, out_w, out_h, flags, out8, as_p(yr, C.c_int32), as_p(xs, C.c_int32),
                            as_p(xi, C.c_int32), as_p(xw, C.c_float), as_p(yi, C.c_int32), as_p(yw, C.c_float))
    if hr != 0:
        raise MpcvrError(hr, "mpcvr_plan_strip")
    return dict(taps=nt, px_per_lane=pxl, strip_w=strip_w, ring=ring, acols=acols, strips=strips, lds_per_wave=lds_wave,
                yrange=yr, xstrip=xs, xi_t=xi, xw_t=xw, yi=yi, yw=yw)


def plan_period(method, src_w, src_h, out_w, out_h, flags=0):
    """PlanFusedPeriod through the C-ABI (no device): the periodic-phase kernel's geometry and tables, or None when the vertical
    ratio is not one of 4:3 / 3:2 / 2:3 / 1:2 / 3:1 (or the tap rows are not the pattern the kernel hard-codes)."""
    import numpy as np
    L = load_library()
    out6 = (C.c_int32 * 6)()
    sw = C.c_int32(0)
    hr = L.mpcvr_plan_period(method, src_w, src_h, out_w, out_h, flags, out6, None, None, None, None, C.byref(sw))
    if hr == E_NOTIMPL:
        return None
    if hr != 0:
        raise MpcvrError(hr, "mpcvr_plan_period")
    Pn, Qn, nt, strips, acols, pb = list(out6)
    xi = np.zeros((nt, out_w), np.int32); xw = np.zeros((nt, out_w), np.float32)
    yw = np.zeros((out_h, 8), np.float32); xs = np.zeros((strips, 2), np.int32)
    as_p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    hr = L.mpcvr_plan_period(method, src_w, src_h, out_w, out_h, flags, out6, as_p(xi, C.c_int32), as_p(xw, C.c_float), as_p(yw, C.c_float), as_p(xs, C.c_int32), C.byref(sw))
    if hr != 0:
        raise MpcvrError(hr, "mpcvr_plan_period")
    return dict(strip_w=sw.value, P=Pn, Q=Qn, taps=nt, strips=strips, acols=acols, rows_per_body=pb, xi_t=xi, xw_t=xw, yw=yw, xstrip=xs)


def plan_pq_eotf_lut():
    import numpy as np
    L = load_library()
    n = C.c_int32(0)
    hr = L.mpcvr_plan_pq_eotf_table(None, 0, C.byref(n))
    if hr:
        raise MpcvrError(hr, "mpcvr_plan_pq_eotf_table")
    out = np.zeros(n.value, np.float32)
    hr = L.mpcvr_plan_pq_eotf_table(out.ctypes.data_as(C.POINTER(C.c_float)), n.value, None)
    if hr:
        raise MpcvrError(hr, "mpcvr_plan_pq_eotf_table")
    return out


def plan_axis_taps(kind, method, src_l, src_len, n_out, tex_len, flags=0, cap_taps=160):
    """Returns (idx[n_out][ntaps], w[n_out][ntaps], wsum[n_out] or None) as nested lists."""
    idx = (C.c_int32 * (n_out * cap_taps))()
    w = (C.c_float * (n_out * cap_taps))()
    ws = (C.c_float * n_out)()
    nt, norm = C.c_int32(), C.c_int32()
    hr = load_library().mpcvr_plan_axis_taps(kind, method, src_l, src_len, n_out, tex_len, flags, cap_taps,
                                             idx, w, ws, C.byref(nt), C.byref(norm))
    if hr != 0:
        raise MpcvrError(hr, "mpcvr_plan_axis_taps")
    n = nt.value
    I = [list(idx[i * n:(i + 1) * n]) for i in range(n_out)]
    W = [list(w[i * n:(i + 1) * n]) for i in range(n_out)]
    return I, W, (list(ws) if norm.value else None)


def plan_describe(settings, cformat, rect_w, rect_h, video_rect, window_w, window_h):
    buf = C.create_string_buffer(256)
    hr = load_library().mpcvr_plan_describe(C.byref(settings), cformat, rect_w, rect_h, C.byref(Rect(*video_rect)),
                                            window_w, window_h, buf, 256)
    if hr < 0:
        raise MpcvrError(hr, buf.value.decode())
    return buf.value.decode()


def _ptr(x):
    """Device/host address of a torch tensor, numpy array or int."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if hasattr(x, "ctypes"):
        return x.ctypes.data
    return int(x)


class PreparedBatch:
    __slots__ = ("n", "sa", "da", "keep")

    def __init__(self, n, sa, da, keep):
        self.n, self.sa, self.da, self.keep = n, sa, da, keep


class VideoProcessor:
    """One processor context on one GPU (not re-entrant, like the reference under m_RendererLock)."""

    def __init__(self, settings=None, device=0, use_torch_stream=True):
        self._L = load_library()
        self._ctx = C.c_void_p()
        self.settings = settings.copy() if settings is not None else default_settings()
        hr = self._L.mpcvr_create(C.byref(self.settings), device, C.byref(self._ctx))
        if hr < 0:
            self._ctx = C.c_void_p()
            raise MpcvrError(hr, "mpcvr_create failed (is a HIP device visible?)")
        self.device = device
        self._keep = []
        if use_torch_stream:
            import torch
            with torch.cuda.device(device):
                self.SetStream(torch.cuda.current_stream().cuda_stream)

    # -- plumbing --------------------------------------------------------------------------------
    def _check(self, hr, allow_false=True):
        if hr < 0:
            raise MpcvrError(hr, self._L.mpcvr_last_error(self._ctx).decode())
        return hr

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._L.mpcvr_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def SetStream(self, hip_stream):
        return self._check(self._L.mpcvr_set_stream(self._ctx, C.c_void_p(hip_stream or 0)))

    def Synchronize(self):
        return self._check(self._L.mpcvr_synchronize(self._ctx))

    # -- CVideoProcessor surface -------------------------------------------------------------------
    def InitMediaType(self, cformat, width, height, pitch=0, src_rect=None, extfmt=0):
        """VerifyMediaType + InitMediaType (DX11VideoProcessor.cpp:1569,1742)."""
        r = Rect(*src_rect) if src_rect is not None else None
        return self._check(self._L.mpcvr_set_input(self._ctx, cformat, width, height, pitch,
                                                   C.byref(r) if r is not None else None, extfmt))

    def SetVideoRect(self, rect):
        return self._check(self._L.mpcvr_set_video_rect(self._ctx, C.byref(Rect(*rect))))

    def SetWindowRect(self, rect):
        return self._check(self._L.mpcvr_set_window_rect(self._ctx, C.byref(Rect(*rect))))

    def SetRotation(self, value):
        return self._check(self._L.mpcvr_set_rotation(self._ctx, value))

    def SetErrorDiffusionPatience(self, polls):
        """(extension) polls a band of the error-diffusion pass waits for the band above before the pass fails; <= 0: the default."""
        return self._check(self._L.mpcvr_set_error_diffusion_patience(self._ctx, int(polls)))

    def SetFlip(self, value):
        return self._check(self._L.mpcvr_set_flip(self._ctx, int(bool(value))))

    def SetHdrOutput(self, enable, tone_map_type=0, display_max_nits=1000.0):
        """HDR10 display mode (m_bHdrPassthrough / m_bHdrLocalToneMapping, m_iHdrLocalToneMappingType, m_iHdrDisplayMaxNits)."""
        return self._check(self._L.mpcvr_set_hdr_output(self._ctx, int(bool(enable)), int(tone_map_type), float(display_max_nits)))

    def SetHdrMetadata(self, min_mastering, max_mastering, max_cll, max_fall):
        """The values Render() hands to SetHDR10ShaderParams (DX11VideoProcessor.cpp:907-917)."""
        return self._check(self._L.mpcvr_set_hdr_metadata(self._ctx, float(min_mastering), float(max_mastering), float(max_cll), float(max_fall)))

    def SetDoviMetadata(self, md):
        """The Dolby Vision RPU of the next sample(s) (CopySample, DX11VideoProcessor.cpp:2270-2520); None ends DoVi mode."""
        if md is None:
            return self._check(self._L.mpcvr_set_dovi_metadata(self._ctx, None))
        st = md if isinstance(md, DoviMetadata) else DoviMetadata.from_dict(md)
        return self._check(self._L.mpcvr_set_dovi_metadata(self._ctx, C.byref(st)))

    def SetSampleFormat(self, frame_format):
        """m_SampleFormat (DX11VideoProcessor.cpp:2209-2219): 0 progressive, 1 interlaced TFF, 2 interlaced BFF."""
        return self._check(self._L.mpcvr_set_sample_format(self._ctx, int(frame_format)))

    def Configure(self, settings):
        hr = self._check(self._L.mpcvr_configure(self._ctx, C.byref(settings)))
        self.settings = settings.copy()
        return hr

    def SetProcAmpValues(self, brightness=None, contrast=None, hue=None, saturation=None):
        flags = 0
        vals = []
        for bit, v, d in ((1, brightness, 0.0), (2, contrast, 1.0), (4, hue, 0.0), (8, saturation, 1.0)):
            if v is not None:
                flags |= bit
            vals.append(d if v is None else float(v))
        return self._check(self._L.mpcvr_set_procamp(self._ctx, flags, *vals))

    def CopySample(self, data, pitch=None, mem_kind=None):
        """CopySample (DX11VideoProcessor.cpp:2202): host array => upload; CUDA/HIP tensor => zero-copy."""
        if pitch is None:
            pitch = self.GetFrameBytes()[1]
        if mem_kind is None:
            mem_kind = MEM_DEVICE if (hasattr(data, "is_cuda") and data.is_cuda) else MEM_HOST
        self._keep = [data]
        return self._check(self._L.mpcvr_copy_sample(self._ctx, C.c_void_p(_ptr(data)), pitch, mem_kind))

    def Process(self, render_target, rt_pitch, src_rect=None, dst_rect=None, second=False):
        """Process (DX11VideoProcessor.cpp:3285). render_target: device tensor/pointer."""
        s = Rect(*src_rect) if src_rect is not None else None
        d = Rect(*dst_rect) if dst_rect is not None else None
        return self._check(self._L.mpcvr_process(self._ctx, C.c_void_p(_ptr(render_target)), rt_pitch,
                                                 C.byref(s) if s is not None else None,
                                                 C.byref(d) if d is not None else None, int(second)))

    def Render(self, field=1):
        return self._check(self._L.mpcvr_render(self._ctx, field))

    def GetBackBuffer(self):
        p, pitch, w, h = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._L.mpcvr_get_backbuffer(self._ctx, C.byref(p), C.byref(pitch), C.byref(w), C.byref(h)))
        return p.value, pitch.value, w.value, h.value

    def GetCurentImage(self):
        """GetCurentImage (DX11VideoProcessor.cpp:3493): BGRX snapshot at source-rect size -> numpy."""
        import numpy as np
        size = C.c_size_t(0)
        self._check(self._L.mpcvr_get_current_image(self._ctx, None, C.byref(size)))
        buf = np.empty(size.value, dtype=np.uint8)
        self._check(self._L.mpcvr_get_current_image(self._ctx, C.c_void_p(buf.ctypes.data), C.byref(size)))
        return buf

    def GetDisplayedImage(self, deep_color=False):
        """GetDisplayedImage (DX11VideoProcessor.cpp:3610): the back buffer of the last Render as DIB pixels -> (numpy bytes, width, height, bits)."""
        import numpy as np
        size, w, h, bits = C.c_size_t(0), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._L.mpcvr_get_displayed_image(self._ctx, None, C.byref(size), int(deep_color), C.byref(w), C.byref(h), C.byref(bits)))
        buf = np.empty(size.value, dtype=np.uint8)
        self._check(self._L.mpcvr_get_displayed_image(self._ctx, C.c_void_p(buf.ctypes.data), C.byref(size), int(deep_color), C.byref(w), C.byref(h), C.byref(bits)))
        return buf, w.value, h.value, bits.value

    def ProcessBatch(self, srcs, dsts, rt_pitch):
        if isinstance(srcs, PreparedBatch):          # pointer arrays built once (PrepareBatch): nothing per call but the C call
            return self._check(self._L.mpcvr_process_batch(self._ctx, srcs.n, srcs.sa, srcs.da, rt_pitch))
        b = self.PrepareBatch(srcs, dsts)
        self._keep = b
        return self._check(self._L.mpcvr_process_batch(self._ctx, b.n, b.sa, b.da, rt_pitch))

    def ProcessBatchDovi(self, srcs, dsts, rt_pitch, rpus):
        """ProcessBatch with one Dolby Vision RPU per frame (mpcvr_process_batch_dovi): rpus[i] is a DoviMetadata or the dict
        DoviMetadata.from_dict takes."""
        b = srcs if isinstance(srcs, PreparedBatch) else self.PrepareBatch(srcs, dsts)
        self._keep = b
        assert len(rpus) == b.n
        arr = (DoviMetadata * b.n)(*[m if isinstance(m, DoviMetadata) else DoviMetadata.from_dict(m) for m in rpus])
        return self._check(self._L.mpcvr_process_batch_dovi(self._ctx, b.n, b.sa, b.da, rt_pitch, arr))

    @staticmethod
    def PrepareBatch(srcs, dsts):
        """The pointer arrays of a batch, for callers that submit the same buffers again and again (a ring): building them
        costs ~2 us of interpreter time per frame, more than a 1080p frame costs the GPU."""
        n = len(srcs)
        assert n == len(dsts) and n > 0
        return PreparedBatch(n, (C.c_void_p * n)(*[_ptr(s) for s in srcs]), (C.c_void_p * n)(*[_ptr(d) for d in dsts]), (srcs, dsts))

    def Flush(self):
        return self._check(self._L.mpcvr_flush(self._ctx))

    def Reset(self):
        return self._check(self._L.mpcvr_reset(self._ctx))

    # -- introspection ---------------------------------------------------------------------------
    def GetParamBlob(self):
        size = C.c_size_t(0)
        self._check(self._L.mpcvr_get_param_blob(self._ctx, None, C.byref(size)))
        buf = (C.c_uint8 * size.value)()
        self._check(self._L.mpcvr_get_param_blob(self._ctx, buf, C.byref(size)))
        return bytes(buf)

    def SetParamBlob(self, blob):
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        return self._check(self._L.mpcvr_set_param_blob(self._ctx, buf, len(blob)))

    def GetColorMatrix(self):
        out = (C.c_float * 12)()
        self._check(self._L.mpcvr_get_color_matrix(self._ctx, out))
        return list(out)

    def GetExtFmt(self):
        v = C.c_uint32()
        self._check(self._L.mpcvr_get_extfmt(self._ctx, C.byref(v)))
        return v.value

    def GetFrameBytes(self):
        n, p = C.c_size_t(), C.c_int32()
        self._check(self._L.mpcvr_get_frame_bytes(self._ctx, C.byref(n), C.byref(p)))
        return n.value, p.value

    def GetVPInfo(self):
        buf = C.create_string_buffer(256)
        self._check(self._L.mpcvr_get_path_info(self._ctx, buf, 256))
        return buf.value.decode()

    def GetLastBatchInfo(self):
        """'frames=<n>;launches=<kernel launches>;lane=<0 | 1: the batch ran on that lane beside its predecessor, -1: on the context stream>[;dovi_runs=...]' of the last ProcessBatch / ProcessBatchDovi call, as a dict."""
        buf = C.create_string_buffer(512)
        self._check(self._L.mpcvr_get_last_batch_info(self._ctx, buf, 512))
        d = dict(kv.split("=", 1) for kv in buf.value.decode().split(";"))
        return dict(frames=int(d["frames"]), launches=int(d["launches"]), dovi_runs=d.get("dovi_runs", ""), lane=int(d.get("lane", -1)))

    def GetLastProcessMs(self):
        ms = C.c_float()
        self._check(self._L.mpcvr_get_last_process_ms(self._ctx, C.byref(ms)))
        return ms.value

    def GetLastTimings(self):
        """{copy_host_ms, upload_ms, process_ms, readback_ms} (FrameStats.h:145-173); -1 = not timed yet."""
        v = [C.c_float(-1.0) for _ in range(4)]
        self._check(self._L.mpcvr_get_last_timings(self._ctx, *(C.byref(x) for x in v)))
        return dict(zip(("copy_host_ms", "upload_ms", "process_ms", "readback_ms"), (x.value for x in v)))
