"""ref_pipeline.py — CDX11VideoProcessor::Process (shader path) executed with the REFERENCE's own pixel shaders.

TEST INFRASTRUCTURE ONLY.  Every Draw of the path runs the reference shader text compiled by ref_hlsl.py; this file only does
what the C++ host and the Direct3D runtime do around the shaders: decode the sample into textures (UNORM), pick the shaders and
constants (DX11VideoProcessor.cpp:3285-3424, :3103-3187, :332-377, :3189-3233, :3048-3101), set viewports and vertex texcoords
(FillVertices :130-179).  It takes the same orc_params the oracle takes, so a test can run both on one input.

Covered: every ColorFormat_t — planar / bi-planar YUV, planar RGB, gray, packed 4:2:2 / 4:4:4 YUV, interleaved RGB (uploaded by the
reference's own CopyFrame* functions, with and without the convert draw); all chroma modes, HDR tails, Dolby Vision reshaping, every
scaler incl. Jinc2, rotation / flip, post-scale tone mapping, final pass.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_hlsl as R  # noqa: E402
import ref_hostmath as RH  # noqa: E402
from oracle import oracle as O  # noqa: E402

F32 = np.float32

# s_FmtConvMapping rows (Helper.cpp:309-359): cformat -> (layout, planes, bytes, subsampling, CDepth, CSType)
# layout: P planar/bi-planar, G gray, K422 packed 4:2:2, K444 packed 4:4:4
_FMT = {}
for cf, pl, by, sub, cd in ((1, 2, 1, 420, 8), (2, 2, 2, 420, 16), (3, 2, 2, 420, 16), (6, 2, 2, 422, 16), (7, 2, 2, 422, 16),
                            (14, 3, 1, 420, 8), (15, 3, 1, 422, 8), (16, 3, 1, 444, 8), (17, 3, 1, 420, 8), (18, 3, 1, 422, 8),
                            (19, 3, 1, 444, 8), (20, 3, 2, 420, 10), (21, 3, 2, 420, 16), (22, 3, 2, 422, 10), (23, 3, 2, 422, 16),
                            (24, 3, 2, 444, 10), (25, 3, 2, 444, 16)):
    _FMT[cf] = ("P", pl, by, sub, cd, "yuv")
for cf, by, cd in ((26, 1, 8), (27, 2, 10), (28, 2, 16)):
    _FMT[cf] = ("P", 3, by, 444, cd, "rgb")
for cf, by, cd in ((37, 1, 8), (38, 2, 10), (39, 2, 16)):
    _FMT[cf] = ("G", 1, by, 400, cd, "gray")
_FMT[4] = ("K422", 1, 1, 422, 8, "yuv")       # YUY2  -> R8G8B8A8 texture, w/2 texels
_FMT[5] = ("K422", 1, 1, 422, 8, "yuv")       # UYVY
_FMT[8] = ("K422", 1, 2, 422, 10, "yuv")      # Y210  -> R16G16B16A16
_FMT[9] = ("K422", 1, 2, 422, 16, "yuv")      # Y216
_FMT[10] = ("K422", 1, 2, 422, 10, "yuv")     # v210 (CopyFrameV210 -> Y210 texture)
_FMT[11] = ("K444", 1, 1, 444, 8, "yuv")      # AYUV  -> R8G8B8A8
_FMT[12] = ("K444", 1, 4, 444, 10, "yuv")     # Y410  -> R10G10B10A2
_FMT[13] = ("K444", 1, 2, 444, 16, "yuv")     # Y416  -> R16G16B16A16
# interleaved RGB (Helper.cpp:344-354): cformat -> (oracle RPK_* = ref_copy_frame_rgb kind, sample bytes per pixel, texture format, CDepth);
# texture formats: "bgrx8" DXGI_FORMAT_B8G8R8X8/A8_UNORM, "rgb10a2" R10G10B10A2_UNORM, "rgba16" R16G16B16A16_UNORM
_RGB = {29: (1, 3, "bgrx8", 8), 30: (0, 4, "bgrx8", 8), 31: (0, 4, "bgrx8", 8), 32: (2, 4, "rgb10a2", 10),
        33: (3, 6, "rgba16", 16), 34: (4, 6, "rgba16", 16), 35: (5, 8, "rgba16", 16), 36: (6, 8, "rgba16", 16)}
for cf, (_k, _pk, _tf, cd) in _RGB.items():
    _FMT[cf] = ("RGB", 1, 1, 444, cd, "rgb")
V_FIRST = (14, 15, 16)                         # YV12 / YV16 / YV24: texV is t1 (Shaders.cpp:159-162), and V is stored first


def supported(cformat):
    return cformat in _FMT


# ---- host-side constants: the reference's OWN functions (libref_hostmath.so: SpecifyExtendedFormat, CopyFrameV210,
# SetShaderDoviCurves[Poly], the level 1 / 3 / 2 block of CopySample, SetHDR10ShaderParams).  The oracle supplies the colour matrix
# only (pinned bit for bit to the real csputils.cpp) and plain data (the Dolby Vision RPU struct, the v210 texture's pitch). ----
def _H():
    L = RH.lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libref_hostmath.so is not built (and /root/reference is not mounted)")
    return L


def specify_extfmt(exfmt, cformat, w, h):
    """SpecifyExtendedFormat (Helper.cpp:1169-1211) with the format's CSType / Subsampling (s_FmtConvMapping)."""
    cstype = {"yuv": 0, "rgb": 1, "gray": 2}[_FMT[cformat][5]]
    return int(_H().ref_specify_extfmt(int(exfmt) & 0xffffffff, cstype, _FMT[cformat][3], int(w), int(h)))


def dovi_curves(dovi_ptr):
    """SetShaderDoviCurves / SetShaderDoviCurvesPoly (:990-1141) chosen as :2305-2318 does (any MMR piece -> the full cbuffer):
    (cbuffer words, has_mmr)."""
    has_mmr = any(dovi_ptr.contents.curves[c].mapping_idc[i] == 1 for c in range(3) for i in range(max(0, dovi_ptr.contents.curves[c].num_pivots - 1)))
    buf = (C.c_uint8 * 3072)()
    n = _H().ref_dovi_curves(dovi_ptr, 0 if has_mmr else 1, buf)
    raw = np.frombuffer(bytes(buf)[:n], dtype=np.uint32).reshape(3, -1)
    # The host struct spends a float4 on every pivot because that is how HLSL lays out `float pivots_data[7]` in a cbuffer (one
    # register per array element, Shaders.cpp:718 "sizeof(float) == sizeof(float4)"); hlsl_shim.h's cbuffers are plain word streams
    # in declaration order, so the seven .x words are handed over back to back — Direct3D's packing rule, applied here.
    stream = np.concatenate([np.concatenate([r[0:28:4], r[28:]]) for r in raw])
    return stream.astype(np.uint32), has_mmr


def dovi_levels(dovi_ptr, display_nits):
    """the level 1 / 3 / 2 block of CopySample + SetDolbyVisionDynamicParams: (k5 floats, enabled, l1 nits, l1 present)."""
    k5 = (C.c_float * 5)(); en = C.c_int(0); l1 = (C.c_uint * 3)()
    present = _H().ref_dovi_levels(dovi_ptr, int(display_nits), k5, C.byref(en), l1)
    return np.array(k5, F32), int(en.value), [int(x) for x in l1], bool(present)


def _rgba(h, w, *ch):
    t = np.zeros((h, w, 4), F32)
    t[..., 3] = 1.0
    for i, c in enumerate(ch):
        t[..., i] = c
    return t


def _unorm(a, maxv):
    return a.astype(F32) / F32(maxv)


def source_textures(p, frame, pitch):
    """Decode the media sample into the textures MemCopyToTexSrcVideo fills (DX11VideoProcessor.cpp:1213-1252,
    Tex11Video_t::CreateEx DX11Helper.h:94-167).  Returns [t0, t1, t2] fp32 RGBA arrays (None where absent)."""
    lay, planes, by, sub, cdepth, _ = _FMT[p.cformat]
    w, h = p.width, p.height
    buf = np.frombuffer(np.ascontiguousarray(frame).view(np.uint8).tobytes(), dtype=np.uint8)
    dt = np.uint8 if by == 1 else np.uint16
    maxv = 255 if by == 1 else 65535

    def plane(off, rows, cols, pit, comps=1):
        a = np.lib.stride_tricks.as_strided(buf[off:], shape=(rows, pit), strides=(pit, 1))[:, :cols * comps * by]
        return np.ascontiguousarray(a).view(dt).reshape(rows, cols, comps) if comps > 1 else np.ascontiguousarray(a).view(dt).reshape(rows, cols)

    if lay == "RGB":
        # MemCopyToTexSrcVideo (:1213-1252) with the format's upload function (GetCopyPlaneFunction, Helper.cpp:377-412) — the
        # reference's own CopyFrame* code (libref_hostmath.so) — into a mapped texture row; a bottom-up DIB (negative pitch) is walked
        # from its last row (:1243-1248).  The row is `width` texels, widened when the sample's pitch makes the reference's loops copy
        # more pixels than that (they land in the padding of the mapped row).
        kind, pack, tfmt, _cd = _RGB[p.cformat]
        tbpp = 8 if tfmt == "rgba16" else 4
        ap = abs(pitch)
        row_px = max(w, ap // pack + 1)
        tp = row_px * tbpp
        dst = np.zeros(tp * h + 16, np.uint8)
        src = np.ascontiguousarray(buf)
        s0 = src.ctypes.data + (pitch * (1 - h) if pitch < 0 else 0)
        _H().ref_copy_frame_rgb(kind, h, dst.ctypes.data, tp, s0, pitch)
        rows = np.lib.stride_tricks.as_strided(dst, shape=(h, tp), strides=(tp, 1))[:, :w * tbpp]
        rows = np.ascontiguousarray(rows)
        t = np.zeros((h, w, 4), F32)
        if tfmt == "bgrx8":
            px = rows.reshape(h, w, 4)
            t[..., 0] = _unorm(px[..., 2], 255); t[..., 1] = _unorm(px[..., 1], 255); t[..., 2] = _unorm(px[..., 0], 255)
            t[..., 3] = _unorm(px[..., 3], 255) if p.cformat == 31 else 1.0
        elif tfmt == "rgb10a2":
            d = rows.view(np.uint32).reshape(h, w)
            for k in range(3):
                t[..., k] = _unorm((d >> (10 * k)) & 0x3ff, 1023)
            t[..., 3] = _unorm(d >> 30, 3)
        else:
            px = rows.view(np.uint16).reshape(h, w, 4)
            for k in range(4):
                t[..., k] = _unorm(px[..., k], 65535)
        return [t, None, None]
    if lay in ("P", "G"):
        shift = 6 if (cdepth == 10 and planes != 2) else 0            # CopyPlane10to16 (Helper.cpp:789-803); P010 is MSB-aligned already
        y = plane(0, h, w, pitch)
        if by == 2:
            y = (y.astype(np.uint32) << shift).astype(np.uint16)
        t0 = _rgba(h, w, _unorm(y, maxv))
        if lay == "G":
            t0[..., 1] = 0
            t0[..., 2] = 0
            return [t0, None, None]
        dw = 1 if sub == 444 else 2
        dh = 2 if sub == 420 else 1
        cw, ch = w // dw, h // dh
        off = pitch * h
        if planes == 2:
            uv = plane(off, ch, cw, pitch, 2)
            return [t0, _rgba(ch, cw, _unorm(uv[..., 0], maxv), _unorm(uv[..., 1], maxv)), None]
        cp = pitch // dw
        a = plane(off, ch, cw, cp)
        b = plane(off + cp * ch, ch, cw, cp)
        if by == 2:
            a = (a.astype(np.uint32) << shift).astype(np.uint16)
            b = (b.astype(np.uint32) << shift).astype(np.uint16)
        # planes are bound in storage order to t1, t2; for YV12 the first chroma plane is V and the shader names t1 texV
        return [t0, _rgba(ch, cw, _unorm(a, maxv)), _rgba(ch, cw, _unorm(b, maxv))]
    if p.cformat == 10:                                                # v210 -> Y210 words: the reference's own CopyFrameV210 (Helper.cpp:709-748)
        tp = O.lib().orc_v210_tex_pitch(w)                             # (the mapped texture's row pitch: a property of the resource, not arithmetic)
        dst = np.zeros(tp * h + 16, np.uint8)
        buf = np.ascontiguousarray(buf)
        _H().ref_copy_frame_v210(h, dst.ctypes.data, tp, buf.ctypes.data, pitch)
        buf, pitch = dst, tp
    if lay == "K422":
        tw = w // 2
        a = plane(0, h, tw, pitch, 4)
        return [np.ascontiguousarray(_unorm(a, maxv)), None, None]
    if p.cformat == 12:                                                # Y410: R10G10B10A2
        d = np.lib.stride_tricks.as_strided(buf, shape=(h, pitch), strides=(pitch, 1))[:, :w * 4]
        d = np.ascontiguousarray(d).view(np.uint32).reshape(h, w)
        t = np.zeros((h, w, 4), F32)
        for k in range(3):
            t[..., k] = _unorm((d >> (10 * k)) & 0x3ff, 1023)
        t[..., 3] = _unorm(d >> 30, 3)
        return [t, None, None]
    a = plane(0, h, w, pitch, 4)
    return [np.ascontiguousarray(_unorm(a, maxv)), None, None]


def convert_args(p):
    """Arguments UpdateConvertColorShader hands GetShaderConvertColor (DX11VideoProcessor.cpp:2942-2975) for these params."""
    lay, planes, by, sub, cdepth, _ = _FMT[p.cformat]
    rect = list(p.src_rect)
    if not any(rect):
        rect = [0, 0, p.width, p.height]
    exfmt = specify_extfmt(p.exfmt, p.cformat, rect[2] - rect[0], rect[3] - rect[1])
    trc = (exfmt >> 27) & 0x1f
    hdr_out = bool(p.hdr_output)
    convert_type = 1 if (p.bConvertToSdr and not hdr_out) else (2 if (hdr_out and trc == 16) else 0)    # :2948-2950
    texw = p.width // 2 if lay == "K422" else p.width
    dv_kind, lms = 0, None
    if p.dovi:
        dv_kind = 2 if dovi_curves(p.dovi)[1] else 1
        lms = tuple(p.dovi.contents.rgb_to_lms_matrix)
    blend = bool(p.blend_deint) and sub == 420 and lay == "P"
    return (p.cformat, planes if lay == "P" else 1, sub, p.width, texw, p.height, exfmt, p.iChromaScaling, convert_type, blend, dv_kind, lms)


def fill_vertices(tex_w, tex_h, rect, rotation, flip):
    """FillVertices (DX11VideoProcessor.cpp:130-179): texcoords at the quad's screen corners -> (TL, TR, BL)."""
    dx, dy = F32(1.0) / F32(tex_w), F32(1.0) / F32(tex_h)
    l, r = dx * F32(rect[0]), dx * F32(rect[2])
    t, b = dy * F32(rect[1]), dy * F32(rect[3])
    if flip:
        l, r = r, l
    v = [(l, b), (l, t), (r, b), (r, t)]
    pts = {90: [(-1, 1), (1, 1), (-1, -1), (1, -1)], 180: [(1, 1), (1, -1), (-1, 1), (-1, -1)],
           270: [(1, -1), (-1, -1), (1, 1), (-1, 1)]}.get(rotation, [(-1, -1), (-1, 1), (1, -1), (1, 1)])
    at = dict(zip(pts, v))
    return (at[(-1, 1)], at[(1, 1)], at[(-1, -1)])


_UP = {1: "mitchell4", 2: "catmull4", 3: "lanczos2", 4: "lanczos3"}
_DOWN = {0: "box", 1: "bilinear", 2: "hamming", 3: "bicubic05", 4: "bicubic15", 5: "lanczos"}
_FMT_RT = {8: 8, 10: 10, 16: 16}


def _shader(name):
    fn = R.find_shader("ps_" + name)
    assert fn is not None, name
    return fn


def _texture_resize(tex, rt, rt_fmt, src_rect, dst_rect, shader, rotation, flip):
    """TextureResizeShader (DX11VideoProcessor.cpp:332-377) / TextureCopyRect with ps_simple when shader is None."""
    th, tw = tex.shape[:2]
    sw, sh = src_rect[2] - src_rect[0], src_rect[3] - src_rect[1]
    dw, dh = dst_rect[2] - dst_rect[0], dst_rect[3] - dst_rect[1]
    cb = R.words(F32(tw), F32(th), F32(1.0) / F32(tw), F32(1.0) / F32(th), F32(sw) / F32(dw), F32(sh) / F32(dh))
    uv = fill_vertices(tw, th, src_rect, rotation, flip)
    R.draw(_shader(shader or "simple"), [tex], rt, rt_fmt, (dst_rect[0], dst_rect[1], dw, dh), uv, samplers=[(0, 0)], cbs=[cb])


def process(p, frame, pitch, dither=None, background=0, stages=None):
    """Whole Process(); returns (window_h, window_w, 4) uint8 (BGRA8) or (window_h, window_w) uint32 (RGB10A2).
    `stages`, when a dict, receives the intermediate surfaces ('convert', 'resize_x', 'post')."""
    assert supported(p.cformat), p.cformat
    L = O.lib()
    lay, planes, by, sub, cdepth, cstype = _FMT[p.cformat]
    rect = list(p.src_rect)
    if not any(rect):
        rect = [0, 0, p.width, p.height]
    rw, rh = rect[2] - rect[0], rect[3] - rect[1]
    exfmt = specify_extfmt(p.exfmt, p.cformat, rw, rh)
    trc, prim = (exfmt >> 27) & 0x1f, (exfmt >> 22) & 0x1f
    hdr_out = bool(p.hdr_output)
    dovi = p.dovi.contents if p.dovi else None

    # UpdateTexParams :1143-1155
    internal = {8: 8, 10: 10, 16: 16}.get(p.iTexFormat, 10 if cdepth > 8 else 8)
    swap = 10 if p.output_format == 1 else 8
    need_dither = (swap == 8 and internal != 8) or (swap == 10 and internal == 16)
    final_pass = bool(p.bUseDither) and need_dither
    tonemap = hdr_out and p.hdr_tonemap_type > 0 and (trc in (15, 16) or dovi is not None)
    has_steps = final_pass or tonemap

    # ---- ConvertColorPass :3048-3101 ----
    texs = source_textures(p, frame, pitch)
    cbs = [None, None, None, None]
    cm = O.color_matrix(p)                                              # cm_r, cm_g, cm_b, cm_c (pinned to the real csputils.cpp)
    cbs[0] = R.words(np.asarray(cm, F32))
    cbs[1] = R.words(F32(10000.0) / F32(p.iSDRDisplayNits), F32(0.0))   # SetShaderLuminanceParams :889-905
    if dovi is not None:                                                # cbuffers b2 / b3 as the reference's own code fills them (:990-1141, :953-960)
        cbs[2], _ = dovi_curves(p.dovi)
        k5, l2, _, _ = dovi_levels(p.dovi, int(p.hdr_display_max_nits))
        cbs[3] = R.words(k5, np.uint32(l2), F32(0), F32(0))
    # m_PSConvColorData.bEnable (:849-853): interleaved RGB with default brightness / contrast has no convert draw — the source
    # texture itself feeds the resize / final pass with rSrc = m_srcRect (:3321-3323)
    enable = (cstype in ("yuv", "gray") or (cstype == "rgb" and lay == "P") or
              abs(F32(p.brightness) / F32(255)) > F32(1e-4) or abs(F32(p.contrast) - F32(1)) > F32(1e-4))
    if enable:
        fn, text = R.convert_fn(*convert_args(p))
        if fn is None:
            raise RuntimeError("convert shader for this configuration is not built (and /root/reference is not mounted)")
        conv = _rgba(rh, rw)
        uv = fill_vertices(p.width, p.height, rect, 0, False)           # CreateVertexBuffer(m_srcWidth, m_srcHeight, m_srcRect) :2042
        R.draw(fn, texs, conv, _FMT_RT[internal], (0, 0, rw, rh), uv, samplers=[(0, 0), (1, 0)], cbs=cbs)
        if stages is not None:
            stages["convert"] = conv.copy()
            stages["convert_text"] = text
    else:
        conv = texs[0]

    # ---- Process :3285-3424 ----
    ww, wh = p.window_w, p.window_h
    dst = list(p.video_rect)
    w2, h2 = dst[2] - dst[0], dst[3] - dst[1]
    rt_final = np.zeros((wh, ww, 4), F32)
    rt_final[...] = np.nan                                              # untouched pixels are reported as `background`
    rsrc = [0, 0, rw, rh] if enable else list(rect)
    rot, flip = p.rotation, bool(p.flip)
    k = 2 if p.bInterpolateAt50pct else 1

    def resize_pass(tex, rt, rt_fmt):                                   # ResizeShaderPass :3103-3187
        up = None if p.iUpscaling == 0 else (("jinc2", "jinc2") if p.iUpscaling == 5 else (_UP[p.iUpscaling] + "_x", _UP[p.iUpscaling] + "_y"))
        dn = ("convol_" + _DOWN[p.iDownscaling] + "_x", "convol_" + _DOWN[p.iDownscaling] + "_y")

        def pick(a, b, axis):
            if a == b:
                return None
            s = dn if a > k * b else up
            return "simple" if s is None else s[axis]                   # UPSCALE_Nearest: ps_simple serves as the "shader" (:94)
        if rot in (90, 270):
            w1, h1 = rsrc[3] - rsrc[1], rsrc[2] - rsrc[0]
            rx = pick(w1, w2, 1)
            ry = pick(h1, h2, 1) if rx else pick(h1, h2, 0)
        else:
            w1, h1 = rsrc[2] - rsrc[0], rsrc[3] - rsrc[1]
            rx, ry = pick(w1, w2, 0), pick(h1, h2, 1)
        if rx and ry:
            if rx == ry:
                return _texture_resize(tex, rt, rt_fmt, rsrc, dst, rx, rot, flip)
            mid = _rgba(h1, rt.shape[1])                                # m_TexResize: RT width x h1, RGBA16F (:3143-3160)
            rr = [dst[0], 0, dst[2], h1]
            _texture_resize(tex, mid, 16, rsrc, rr, rx, rot, flip)
            if stages is not None:
                stages["resize_x"] = mid.copy()
            return _texture_resize(mid, rt, rt_fmt, rr, dst, ry, 0, False)
        return _texture_resize(tex, rt, rt_fmt, rsrc, dst, rx or ry, rot, flip)

    if has_steps:
        steps = int(final_pass) + int(tonemap)                          # GetPostScaleSteps :779-795
        ring = [np.zeros((wh, ww, 4), F32) for _ in range(steps)]      # m_TexsPostScale: window size, internal format
        st = dict(step=0, ring=0, inp=conv, ptex=ring[0], prt=ring[0], prt_fmt=_FMT_RT[internal])

        def step_setting():                                             # StepSetting :3323-3333
            st["step"] += 1
            st["inp"] = st["ptex"]
            if st["step"] < steps:
                st["ring"] += 1
                st["ptex"] = ring[st["ring"]]
                st["prt"], st["prt_fmt"] = st["ptex"], _FMT_RT[internal]
            else:
                st["prt"], st["prt_fmt"] = rt_final, swap
        rect_i = [max(dst[0], 0), max(dst[1], 0), min(dst[2], ww), min(dst[3], wh)]      # dstRect ∩ texture :3336-3337
        vp_i = (rect_i[0], rect_i[1], rect_i[2] - rect_i[0], rect_i[3] - rect_i[1])
        if rsrc != dst or rot != 0:
            resize_pass(st["inp"], st["prt"], st["prt_fmt"])
        else:
            st["ptex"] = st["inp"]                                      # "Hmm" :3352
        if tonemap:                                                     # :3359-3367, TextureCopyRect :300-330
            step_setting()
            tm = _hdr_tm_constants(p, L)
            l2 = cbs[3] if dovi is not None else R.words(F32(0), F32(0), F32(0), F32(0), F32(0), np.uint32(0), F32(0), F32(0))
            inp = st["inp"]
            R.draw(_shader("hdr10_tonemap"), [inp], st["prt"], st["prt_fmt"], vp_i, fill_vertices(inp.shape[1], inp.shape[0], rect_i, 0, False),
                   samplers=[(0, 0)], cbs=[tm, l2])
        if final_pass:                                                  # FinalPass(*pTex, pRT, rect, rect) :3411-3415, :3189-3233
            step_setting()
            tex = st["ptex"]
            if dither is None:
                dither = O.dither_table()
            d = np.asarray(dither, np.uint16).view(np.float16).astype(F32).reshape(32, 32)
            dtex = np.repeat(d[:, :, None], 4, axis=2).copy()          # replicated to 4 channels :1414-1440
            cb = R.words(F32(tex.shape[1]) / F32(32), F32(tex.shape[0]) / F32(32))
            R.draw(_shader("final_pass_10" if swap == 10 else "final_pass"), [tex, dtex], rt_final, swap, vp_i,
                   fill_vertices(tex.shape[1], tex.shape[0], rect_i, 0, False), samplers=[(0, 0), (0, 1)], cbs=[cb])
        if stages is not None:
            stages["post"] = [t.copy() for t in ring]
    else:
        resize_pass(conv, rt_final, swap)

    return pack_output(rt_final, swap, background)


def _hdr_tm_constants(p, L):
    """HDRParamsConstantBuffer_t from the reference's own SetHDR10ShaderParams (:907-923); with Dolby Vision level-1 data the caller
    passes the L1 nits and turns type 5 into 6 (Render :2716-2720)."""
    mn, mx, cll, fall, disp, sel = p.hdr_min_mastering, p.hdr_max_mastering, p.hdr_max_cll, p.hdr_max_fall, p.hdr_display_max_nits, p.hdr_tonemap_type
    if p.dovi:
        _, _, l1, present = dovi_levels(p.dovi, int(p.hdr_display_max_nits))
        if present:
            mn, mx, cll, fall = float(l1[0]), float(l1[1]), float(l1[1]), float(l1[2])
            if sel == 5:
                sel = 6
    out = (C.c_uint32 * 6)()
    _H().ref_hdr10_params(mn, mx, cll, fall, disp, int(sel), out)
    return np.concatenate([np.array(out, np.uint32), R.words(F32(0), F32(0))])


def pack_output(rt, swap, background=0):
    """Render-target contents (already rounded to the format by the draw) -> B8G8R8A8 bytes / R10G10B10A2 dwords."""
    mask = np.isnan(rt[..., 0])
    v = np.nan_to_num(rt, nan=0.0)
    if swap == 10:
        q = np.floor(v[..., :3] * F32(1023) + F32(0.5)).astype(np.uint32)
        a = np.floor(v[..., 3] * F32(3) + F32(0.5)).astype(np.uint32)
        out = q[..., 0] | (q[..., 1] << 10) | (q[..., 2] << 20) | (a << 30)
        out[mask] = np.uint32(background) * np.uint32(0x01010101)
        return out
    q = np.floor(v * F32(255) + F32(0.5)).astype(np.uint8)
    out = np.stack([q[..., 2], q[..., 1], q[..., 0], q[..., 3]], axis=2)
    out[mask] = background
    return out


def build_all(extra_params=()):
    """Build oracle/_ref/libref_hlsl.so with the convert shader of every golden / pinning case (tests/golden/cases.py) and of the
    full-size BASELINE configurations, so that the live checks also run where /root/reference does not exist (the GPU box)."""
    from tests.golden import cases
    from tests.golden.make_ref_hlsl_golden import all_cases, comparable
    args = []
    for c in all_cases().values():
        if comparable(c):
            args.append(convert_args(cases.oracle_params(O, c)))
    for cf, w, h, ex in ((2, 3840, 2160, cases.HDR10), (2, 3840, 2160, cases.HLG), (2, 3840, 2160, cases.ext(matrix=cases.M709)),
                         (20, 1920, 1080, cases.ext(matrix=cases.M709)), (1, 1920, 1080, cases.ext(matrix=cases.M709)),
                         (2, 1920, 1080, cases.HDR10), (2, 256, 144, cases.HDR10)):
        args.append(convert_args(O.default_params(cformat=cf, width=w, height=h, exfmt=ex, window_w=w, window_h=h, video_rect=(0, 0, w, h))))
    for c in cases.FULL_SIZE_CASES.values():
        args.append(convert_args(cases.oracle_params(O, c)))
    for p in extra_params:
        args.append(convert_args(p))
    return R.build(args)


if __name__ == "__main__":
    print("built", build_all(), "shaders into", R.LIB_HLSL)
