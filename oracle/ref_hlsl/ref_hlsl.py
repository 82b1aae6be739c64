"""ref_hlsl.py — build + ctypes access to the REFERENCE's own pixel shaders compiled for the CPU.  TEST INFRASTRUCTURE ONLY.

Two libraries under oracle/_ref/ (git-ignored; they travel to the GPU box like every other built .so):

  libref_shadergen.so  the real Source/Shaders.cpp + csputils.cpp behind Win32 stand-ins (stub_shadergen/): hands back the HLSL
                       text GetShaderConvertColor would give D3DCompile for a format / colourimetry / chroma mode;
  libref_hlsl.so       every fixed shader of the path (Shaders/d3d11/ps_interpolation_*, ps_convolution, ps_final_pass,
                       ps_hdr10_tonemap, the six correction shaders, examples/ps_resize_onepass_jinc2) in its fxc macro variants
                       (Shaders/compile_shaders.cmd:80-109), plus the generated convert shaders of the requested configurations,
                       each rewritten mechanically by hlsl2cpp.py and compiled against hlsl_shim.h.

Convert shaders asked for later (a new size or format) are compiled on demand into oracle/_ref/cv/ when /root/reference is
mounted; on the GPU box only the prebuilt ones exist.
"""
import ctypes as C
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.dirname(HERE)
REF_DIR = os.path.join(ORACLE, "_ref")
GEN_DIR = os.path.join(REF_DIR, "gen")
CV_DIR = os.path.join(REF_DIR, "cv")
REFERENCE_ROOT = os.environ.get("MPCVR_REFERENCE_ROOT", "/root/reference")     # tests point it elsewhere to rehearse the GPU box
LIB_HLSL = os.path.join(REF_DIR, "libref_hlsl.so")
LIB_GEN = os.path.join(REF_DIR, "libref_shadergen.so")
CXXFLAGS = ["-std=c++17", "-O2", "-msse2", "-ffp-contract=off", "-fno-math-errno", "-fopenmp", "-fPIC", "-w"]

sys.path.insert(0, HERE)
import hlsl2cpp  # noqa: E402

D3D11 = "Shaders/d3d11/"
# name -> (file relative to the reference root, fxc /D macros)   — Shaders/compile_shaders.cmd:80-109
FIXED_SHADERS = {
    "mitchell4_x": (D3D11 + "ps_interpolation_spline4.hlsl", {"METHOD": "0", "AXIS": "0"}),
    "mitchell4_y": (D3D11 + "ps_interpolation_spline4.hlsl", {"METHOD": "0", "AXIS": "1"}),
    "catmull4_x": (D3D11 + "ps_interpolation_spline4.hlsl", {"METHOD": "1", "AXIS": "0"}),
    "catmull4_y": (D3D11 + "ps_interpolation_spline4.hlsl", {"METHOD": "1", "AXIS": "1"}),
    "lanczos2_x": (D3D11 + "ps_interpolation_lanczos2.hlsl", {"AXIS": "0"}),
    "lanczos2_y": (D3D11 + "ps_interpolation_lanczos2.hlsl", {"AXIS": "1"}),
    "lanczos3_x": (D3D11 + "ps_interpolation_lanczos3.hlsl", {"AXIS": "0"}),
    "lanczos3_y": (D3D11 + "ps_interpolation_lanczos3.hlsl", {"AXIS": "1"}),
    "jinc2": ("Shaders/examples/ps_resize_onepass_jinc2.hlsl", {}),
    "final_pass": (D3D11 + "ps_final_pass.hlsl", {}),
    "final_pass_10": (D3D11 + "ps_final_pass.hlsl", {"QUANTIZATION": "1023"}),
    "hdr10_tonemap": (D3D11 + "ps_hdr10_tonemap.hlsl", {}),
    "fix_bt2020": (D3D11 + "ps_fix_bt2020.hlsl", {}),
    "fix_ycgco": (D3D11 + "ps_fix_ycgco.hlsl", {}),
    "fixconvert_pq_to_sdr": (D3D11 + "ps_fixconvert_pq_to_sdr.hlsl", {}),
    "fixconvert_hlg_to_sdr": (D3D11 + "ps_fixconvert_hlg_to_sdr.hlsl", {}),
    "convert_pq_to_sdr": (D3D11 + "ps_convert_pq_to_sdr.hlsl", {}),
    "convert_hlg_to_pq": (D3D11 + "ps_convert_hlg_to_pq.hlsl", {}),
    "simple": (D3D11 + "ps_simple.hlsl", {}),
}
for _i, (_n, _extra) in enumerate([("box", {}), ("bilinear", {}), ("hamming", {}), ("bicubic05", {"A": "-0.5"}),
                                   ("bicubic15", {"A": "-1.5"}), ("lanczos", {})]):
    _f = {"box": 0, "bilinear": 1, "hamming": 2, "bicubic05": 3, "bicubic15": 3, "lanczos": 4}[_n]
    for _ax, _axn in ((0, "x"), (1, "y")):
        FIXED_SHADERS["convol_%s_%s" % (_n, _axn)] = (D3D11 + "ps_convolution.hlsl", dict(FILTER=str(_f), AXIS=str(_ax), **_extra))


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "Shaders", "d3d11"))


def _run(cmd, **kw):
    subprocess.check_call(cmd, **kw)


def _compile(cpp, obj):
    _run(["g++"] + CXXFLAGS + ["-I", HERE, "-c", cpp, "-o", obj])


def _float_defs(d):
    return {k: hlsl2cpp._FLOAT_LIT.sub(lambda m: m.group(1) + "f", v) for k, v in d.items()}


def build_shadergen():
    """libref_shadergen.so = the real Shaders.cpp + csputils.cpp + our capture shim (compiled where they lie)."""
    os.makedirs(REF_DIR, exist_ok=True)
    stub = os.path.join(HERE, "stub_shadergen")
    src = os.path.join(REFERENCE_ROOT, "Source")
    objs = []
    for name, stdin_file, extra in (("shaders", os.path.join(src, "Shaders.cpp"), ["-iquote", stub, "-I", stub]),
                                    ("csputils_g", os.path.join(src, "csputils.cpp"), ["-iquote", os.path.join(ORACLE, "stub")])):
        o = os.path.join(REF_DIR, name + ".o")
        with open(stdin_file, "rb") as f:
            _run(["g++", "-std=c++17", "-O2", "-msse2", "-ffp-contract=off", "-fPIC", "-w", "-x", "c++", "-c", "-o", o] + extra +
                 ["-I", src, "-"], stdin=f)
        objs.append(o)
    o = os.path.join(REF_DIR, "shadergen_shim.o")
    _run(["g++", "-std=c++17", "-O2", "-fPIC", "-w", "-c", "-o", o, "-iquote", stub, "-I", stub, "-I", src,
          os.path.join(HERE, "ref_shadergen_shim.cpp")])
    objs.append(o)
    _run(["g++", "-shared", "-o", LIB_GEN] + objs)
    for o in objs:
        os.remove(o)


_gen = None


def shadergen():
    global _gen
    if _gen is None:
        if not os.path.exists(LIB_GEN):
            if not have_reference():
                return None
            build_shadergen()
        L = C.CDLL(LIB_GEN)
        L.ref_convert_shader_text.restype = C.c_int
        L.ref_convert_shader_text.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        L.ref_shadergen_set_root.argtypes = [C.c_char_p]
        L.ref_shadergen_set_root(REFERENCE_ROOT.encode())
        _gen = L
    return _gen


def convert_shader_text(cformat, planes, subsampling, width, texw, texh, exfmt, chroma_scaling, convert_type, blend_deint,
                        dovi=0, rgb_to_lms=None):
    """The text GetShaderConvertColor (Source/Shaders.cpp:593-930) emits — produced by the real function."""
    L = shadergen()
    if L is None or not have_reference():          # the generator reads the .hlsl includes from the reference tree
        return None
    buf = C.create_string_buffer(1 << 17)
    lms = None
    if rgb_to_lms is not None:
        lms = (C.c_double * 9)(*[float(x) for x in rgb_to_lms])
    n = L.ref_convert_shader_text(cformat, planes, subsampling, width, texw, texh, exfmt & 0xffffffff, chroma_scaling, convert_type,
                                  int(blend_deint), dovi, lms, buf, len(buf))
    assert 0 < n < len(buf), n
    return buf.value.decode("utf-8", errors="replace")


KEYS_JSON = os.path.join(REF_DIR, "convert_keys.json")
_keys = None


def _args_id(args):
    """Stable identifier of a GetShaderConvertColor argument set (the Dolby Vision matrix enters the text, so it is part of it)."""
    a = list(args)
    lms = a[-1]
    a[-1] = None if lms is None else [float(x).hex() for x in lms]
    return repr(a)


def convert_keys():
    global _keys
    if _keys is None:
        _keys = {}
        if os.path.exists(KEYS_JSON):
            import json
            with open(KEYS_JSON) as f:
                _keys = json.load(f)
    return _keys


def convert_fn(*args):
    """Draw function of the convert shader GetShaderConvertColor emits for these arguments (see convert_shader_text).
    With the reference mounted the text is generated (and compiled on demand); without it, only configurations recorded in
    convert_keys.json at build time are available.  Returns (fn, text-or-None); fn is None when unavailable."""
    text = convert_shader_text(*args)
    if text is not None:
        return convert_shader(text), text
    key = convert_keys().get(_args_id(args))
    return (find_shader(key) if key else None), None


def text_key(text):
    return "cv_" + hashlib.sha1(text.encode()).hexdigest()[:16]


def _gen_fixed(name):
    rel, defs = FIXED_SHADERS[name]
    path = os.path.join(REFERENCE_ROOT, rel)
    with open(path, encoding="utf-8-sig", errors="replace") as f:
        cpp = hlsl2cpp.transform(f.read(), "ps_" + name, os.path.dirname(path), _float_defs(defs))
    out = os.path.join(GEN_DIR, "ps_%s.cpp" % name)
    with open(out, "w") as f:
        f.write(cpp)
    obj = out[:-4] + ".o"
    _compile(out, obj)
    return obj


def _gen_convert(text):
    key = text_key(text)
    out = os.path.join(GEN_DIR, key + ".cpp")
    with open(out, "w") as f:
        f.write(hlsl2cpp.transform(text, key))
    obj = out[:-4] + ".o"
    _compile(out, obj)
    return obj


def build(convert_args=(), jobs=None):
    """Build libref_hlsl.so: all fixed shaders + the convert shaders of the given GetShaderConvertColor argument sets
    (tuples as convert_shader_text takes them); records args -> key in convert_keys.json.  Needs /root/reference."""
    import json
    assert have_reference(), "the reference tree is not mounted"
    os.makedirs(GEN_DIR, exist_ok=True)
    if not os.path.exists(LIB_GEN):
        build_shadergen()
    keys = {}
    texts = set()
    for a in convert_args:
        t = convert_shader_text(*a)
        keys[_args_id(a)] = text_key(t)
        texts.add(t)
    texts = sorted(texts)
    with open(KEYS_JSON, "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    global _keys
    _keys = None
    with ThreadPoolExecutor(jobs or os.cpu_count() or 4) as ex:
        objs = list(ex.map(_gen_fixed, sorted(FIXED_SHADERS)))
        objs += list(ex.map(_gen_convert, texts))
    rt = os.path.join(GEN_DIR, "ref_runtime.o")
    _compile(os.path.join(HERE, "ref_runtime.cpp"), rt)
    _run(["g++", "-shared", "-fopenmp", "-o", LIB_HLSL, rt] + objs)
    for o in objs + [rt]:
        os.remove(o)
    import shutil
    shutil.rmtree(GEN_DIR, ignore_errors=True)      # the rewritten sources are build intermediates
    global _lib
    _lib = None
    return len(objs)


class RefSurface(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("w", C.c_int32), ("h", C.c_int32)]


class RefDraw(C.Structure):
    _fields_ = [("tex", RefSurface * 4), ("samp_filter", C.c_int32 * 4), ("samp_address", C.c_int32 * 4),
                ("cb", C.POINTER(C.c_uint32) * 4), ("cb_words", C.c_int32 * 4),
                ("rt", RefSurface), ("rt_fmt", C.c_int32),
                ("vp_x", C.c_int32), ("vp_y", C.c_int32), ("vp_w", C.c_int32), ("vp_h", C.c_int32),
                ("uv", (C.c_float * 2) * 3), ("threads", C.c_int32)]


_lib = None
_cv_libs = {}


def lib():
    global _lib
    if _lib is None and os.path.exists(LIB_HLSL):
        _lib = C.CDLL(LIB_HLSL)
        _lib.ref_half_round.restype = C.c_float
        _lib.ref_half_round.argtypes = [C.c_float]
    return _lib


def available():
    return lib() is not None


def find_shader(name):
    """ctypes function of a compiled shader: 'ps_<fixed name>' or a convert-shader key; None when it is not built."""
    L = lib()
    fn = None
    if L is not None:
        try:
            fn = getattr(L, name)
        except AttributeError:
            fn = None
    if fn is None and name in _cv_libs:
        fn = getattr(_cv_libs[name], name)
    if fn is None:
        so = os.path.join(CV_DIR, name + ".so")
        if os.path.exists(so):
            _cv_libs[name] = C.CDLL(so)
            fn = getattr(_cv_libs[name], name)
    if fn is not None:
        fn.argtypes = [C.POINTER(RefDraw)]
        fn.restype = None
    return fn


def convert_shader(text):
    """Compiled draw function of a generated convert shader; compiled on demand when the reference is mounted."""
    key = text_key(text)
    fn = find_shader(key)
    if fn is None and have_reference() and lib() is not None:
        os.makedirs(CV_DIR, exist_ok=True)
        os.makedirs(GEN_DIR, exist_ok=True)
        cpp = os.path.join(GEN_DIR, key + ".cpp")
        with open(cpp, "w") as f:
            f.write(hlsl2cpp.transform(text, key))
        so = os.path.join(CV_DIR, key + ".so")
        # g_cb / rt_round live in libref_hlsl.so (already loaded RTLD_GLOBAL below)
        C.CDLL(LIB_HLSL, mode=C.RTLD_GLOBAL)
        _run(["g++"] + CXXFLAGS + ["-I", HERE, "-shared", "-o", so, cpp, "-L", REF_DIR, "-l:libref_hlsl.so",
                                     "-Wl,-rpath," + REF_DIR])
        fn = find_shader(key)
    return fn


def surface(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 3 and a.shape[2] == 4
    s = RefSurface(a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1], a.shape[0])
    s._keep = a
    return s


def words(*vals):
    """cbuffer words: python floats -> fp32 bits, ints (np.uint32 / int) -> as is; arrays are flattened."""
    out = []
    for v in vals:
        if isinstance(v, np.ndarray):
            out.append(np.ascontiguousarray(v).view(np.uint32).ravel() if v.dtype != np.uint32 else v.ravel())
        elif isinstance(v, (int, np.integer)) and not isinstance(v, bool):
            out.append(np.array([v], dtype=np.uint32))
        else:
            out.append(np.array([v], dtype=np.float32).view(np.uint32))
    return np.concatenate(out).astype(np.uint32) if out else np.zeros(0, np.uint32)


def draw(fn, textures, rt, rt_fmt, viewport, uv, samplers=((0, 0),), cbs=(), threads=0):
    """One Draw(4,0): textures = [(h,w,4) fp32 arrays] for t0.., samplers = [(filter, address)] for s0.., cbs = [uint32 arrays]
    for b0.., rt = (H,W,4) fp32 array written in place, viewport = (x, y, w, h), uv = ((u,v) top-left, top-right, bottom-left)."""
    d = RefDraw()
    keep = []
    for i, t in enumerate(textures):
        if t is None:
            continue
        s = surface(t)
        keep.append(s)
        d.tex[i] = s
    for i, (f, a) in enumerate(samplers):
        d.samp_filter[i] = f
        d.samp_address[i] = a
    for i, cb in enumerate(cbs):
        if cb is None:
            continue
        cb = np.ascontiguousarray(cb, dtype=np.uint32)
        keep.append(cb)
        d.cb[i] = cb.ctypes.data_as(C.POINTER(C.c_uint32))
        d.cb_words[i] = cb.size
    assert rt.dtype == np.float32 and rt.flags["C_CONTIGUOUS"]
    d.rt = RefSurface(rt.ctypes.data_as(C.POINTER(C.c_float)), rt.shape[1], rt.shape[0])
    d.rt_fmt = rt_fmt
    d.vp_x, d.vp_y, d.vp_w, d.vp_h = [int(v) for v in viewport]
    for i in range(3):
        d.uv[i][0] = float(uv[i][0])
        d.uv[i][1] = float(uv[i][1])
    d.threads = threads
    fn(C.byref(d))
    return rt
