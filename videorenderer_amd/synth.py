"""Deterministic synthetic frames: SplitMix64 noise in the legal code range and a
structure frame (ramps, checker, zone plate, PQ staircase), laid out exactly like the media samples
the reference receives (planes back to back, MemCopyToTexSrcVideo DX11VideoProcessor.cpp:1213-1252)."""
import numpy as np

SEED_BASE = 0x4D50435652

# cformat -> (planes, bytes, div_w, div_h, bits, msb_aligned, v_first)
FORMATS = {
    1: (2, 1, 2, 2, 8, False, False),    # NV12
    2: (2, 2, 2, 2, 10, True, False),    # P010 (10 bits in the MSBs)
    3: (2, 2, 2, 2, 16, True, False),    # P016
    6: (2, 2, 2, 1, 10, True, False),    # P210
    7: (2, 2, 2, 1, 16, True, False),    # P216
    14: (3, 1, 2, 2, 8, False, True),    # YV12
    15: (3, 1, 2, 1, 8, False, True),    # YV16
    16: (3, 1, 1, 1, 8, False, True),    # YV24
    17: (3, 1, 2, 2, 8, False, False),   # YUV420P8
    18: (3, 1, 2, 1, 8, False, False),
    19: (3, 1, 1, 1, 8, False, False),
    20: (3, 2, 2, 2, 10, False, False),  # YUV420P10 (raw 0..1023)
    21: (3, 2, 2, 2, 16, False, False),
    22: (3, 2, 2, 1, 10, False, False),
    23: (3, 2, 2, 1, 16, False, False),
    24: (3, 2, 1, 1, 10, False, False),
    25: (3, 2, 1, 1, 16, False, False),
}


# one-plane formats (Helper.cpp:315-324,341-343,356-358): cformat -> (kind, bytes per component, code bits, msb_aligned)
#   p422: two pixels per RGBA texel; p444: one pixel per texel; gbrp: planar G,B,R; gray: luma only
PACKED = {
    4: ("yuy2", 1, 8, False), 5: ("uyvy", 1, 8, False), 8: ("y210", 2, 10, True), 9: ("y210", 2, 16, True),
    10: ("v210", 0, 10, False),
    11: ("ayuv", 1, 8, False), 12: ("y410", 4, 10, False), 13: ("y416", 2, 16, True),
    26: ("gbrp", 1, 8, False), 27: ("gbrp", 2, 10, False), 28: ("gbrp", 2, 16, False),
    37: ("gray", 1, 8, False), 38: ("gray", 2, 10, False), 39: ("gray", 2, 16, False),
    # interleaved RGB (Helper.cpp:345-354); the "Y,U,V" floats stand in for R,G,B
    29: ("rgb24", 1, 8, False), 30: ("rgb32", 1, 8, False), 31: ("rgb32", 1, 8, False), 32: ("r210", 4, 10, False),
    33: ("rgb48", 2, 16, False), 34: ("bgr48", 2, 16, False), 35: ("bgra64", 2, 16, False), 36: ("b64a", 2, 16, False),
}
RGB_FAMILIES = ("rgb24", "rgb32", "r210", "rgb48", "bgr48", "bgra64", "b64a")


def splitmix64(n, seed):
    """n uint64 values of the SplitMix64 stream started at `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def default_pitch(cformat, w):
    if cformat in PACKED:
        kind, nbytes = PACKED[cformat][0], PACKED[cformat][1]
        if kind in ("yuy2", "uyvy", "y210"):
            return w * 2 * nbytes
        if kind == "v210":
            return (((w + 5) // 6 * 16) + 127) & ~127           # DX11VideoProcessor.cpp:1798-1799
        if kind in ("ayuv", "y410"):
            return w * 4
        if kind == "y416":
            return w * 8
        if kind in RGB_FAMILIES:
            pitch = w * {"rgb24": 3, "rgb32": 4, "r210": 4, "rgb48": 6, "bgr48": 6, "bgra64": 8, "b64a": 8}[kind]
            return (pitch + 3) & ~3 if kind in ("rgb24", "bgr48") else pitch       # :1792-1796
        pitch = w * nbytes                                       # gbrp, gray
        return (pitch + 3) & ~3 if cformat == 37 else pitch      # Y8: ALIGN(pitch, 4) (:1792-1796)
    planes, nbytes = FORMATS[cformat][0], FORMATS[cformat][1]
    pitch = w * nbytes
    if cformat == 1:
        pitch = (pitch + 3) & ~3
    return pitch


def _planes_float(kind, w, h, cw, ch, seed):
    """Y, U, V as float in [0,1] of the *legal* range (0 = black/min chroma, 1 = white/max chroma)."""
    if kind == "noise":
        r = splitmix64(w * h + 2 * cw * ch, seed)
        u01 = (r >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        y = u01[: w * h].reshape(h, w)
        u = u01[w * h: w * h + cw * ch].reshape(ch, cw)
        v = u01[w * h + cw * ch:].reshape(ch, cw)
        return y, u, v
    if kind in ("structure", "hdr"):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        ramp = 0.5 * (xx / max(w - 1, 1)) + 0.5 * (yy / max(h - 1, 1))
        checker = (((xx // 8) + (yy // 8)) % 2)
        cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
        r2 = ((xx - cx) ** 2 + (yy - cy) ** 2) / float(max(w, h))
        zone = 0.5 + 0.5 * np.cos(np.pi * r2 / 8.0)
        y = np.where(yy < h / 3, ramp, np.where(yy < 2 * h / 3, 0.15 + 0.7 * checker, zone))
        if kind == "hdr":       # PQ grey staircase 0..1 in 64 steps across the top rows
            stair = np.floor(xx / max(w / 64.0, 1.0)) / 63.0
            y = np.where(yy < h / 8, np.clip(stair, 0, 1), y)
        cyy, cxx = np.mgrid[0:ch, 0:cw].astype(np.float64)
        u = cxx / max(cw - 1, 1)
        v = 1.0 - cyy / max(ch - 1, 1)
        if kind == "hdr":
            u = np.where(cyy < ch / 8, 0.5, u)
            v = np.where(cyy < ch / 8, 0.5, v)
        return y, u, v
    raise ValueError(kind)


def _quantise(y, u, v, code_bits, full_range, rgb=False):
    scale = 1 << (code_bits - 8)
    if full_range or rgb:
        ylo, yhi, clo, chi = 0, (1 << code_bits) - 1, 0, (1 << code_bits) - 1
    else:
        ylo, yhi, clo, chi = 16 * scale, 235 * scale, 16 * scale, 240 * scale
    q = lambda a, lo, hi: np.clip(np.floor(lo + a * (hi - lo) + 0.5), lo, hi).astype(np.uint32)
    return q(y, ylo, yhi), q(u, clo, chi), q(v, clo, chi)


def _make_packed(cformat, w, h, kind, seed, pitch, full_range):
    fam, nbytes, bits, msb = PACKED[cformat]
    if pitch is None:
        pitch = default_pitch(cformat, w)
    if fam in ("yuy2", "uyvy", "y210", "v210"):
        cw, ch = w // 2, h
    elif fam == "gray":
        cw, ch = 1, 1
    else:
        cw, ch = w, h          # 4:4:4 and RGB: full-resolution "chroma"
    y, u, v = _planes_float(kind, w, h, cw, ch, SEED_BASE + seed)
    yq, uq, vq = _quantise(y, u, v, bits, full_range, rgb=(fam == "gbrp" or fam in RGB_FAMILIES))
    if msb and bits == 10:
        yq, uq, vq = yq << 6, uq << 6, vq << 6
    buf = np.zeros(pitch * h * (3 if fam == "gbrp" else 1), dtype=np.uint8)
    rows = buf[: pitch * h].reshape(h, pitch)
    if fam in ("yuy2", "uyvy", "y210"):
        dt = np.uint8 if nbytes == 1 else np.uint16
        t = rows.view(dt)[:, : 2 * w].reshape(h, w // 2, 4)
        iy0, iu, iy1, iv = (1, 0, 3, 2) if fam == "uyvy" else (0, 1, 2, 3)
        t[:, :, iy0] = yq[:, 0::2]; t[:, :, iy1] = yq[:, 1::2]; t[:, :, iu] = uq; t[:, :, iv] = vq
    elif fam == "v210":
        # SMPTE v210: 6 pixels in four little-endian dwords of three 10-bit fields (low, mid, high):
        #   (Cb0,Y0,Cr0) (Y1,Cb1,Y2) (Cr1,Y3,Cb2) (Y4,Cr2,Y5); rows padded to whole groups
        groups = (w + 5) // 6
        yp = np.zeros((h, groups * 6), np.uint32); yp[:, :w] = yq
        up = np.zeros((h, groups * 3), np.uint32); up[:, : w // 2] = uq
        vp = np.zeros((h, groups * 3), np.uint32); vp[:, : w // 2] = vq
        yg, ug, vg = yp.reshape(h, groups, 6), up.reshape(h, groups, 3), vp.reshape(h, groups, 3)
        d = rows.view(np.uint32)[:, : groups * 4].reshape(h, groups, 4)
        d[:, :, 0] = ug[:, :, 0] | (yg[:, :, 0] << 10) | (vg[:, :, 0] << 20)
        d[:, :, 1] = yg[:, :, 1] | (ug[:, :, 1] << 10) | (yg[:, :, 2] << 20)
        d[:, :, 2] = vg[:, :, 1] | (yg[:, :, 3] << 10) | (ug[:, :, 2] << 20)
        d[:, :, 3] = yg[:, :, 4] | (vg[:, :, 2] << 10) | (yg[:, :, 5] << 20)
    elif fam == "ayuv":
        t = rows[:, : 4 * w].reshape(h, w, 4)
        t[:, :, 0] = vq; t[:, :, 1] = uq; t[:, :, 2] = yq; t[:, :, 3] = 255
    elif fam == "y410":
        rows.view(np.uint32)[:, :w] = uq | (yq << 10) | (vq << 20) | np.uint32(3 << 30)
    elif fam == "y416":
        t = rows.view(np.uint16)[:, : 4 * w].reshape(h, w, 4)
        t[:, :, 0] = uq; t[:, :, 1] = yq; t[:, :, 2] = vq; t[:, :, 3] = 65535
    elif fam == "gbrp":      # planes G, B, R (the "Y,U,V" floats stand in for G,B,R)
        dt = np.uint8 if nbytes == 1 else np.uint16
        for i, q in enumerate((yq, uq, vq)):
            buf[i * pitch * h: (i + 1) * pitch * h].reshape(h, pitch).view(dt)[:, :w] = q.astype(dt)
    elif fam == "rgb24":     # Windows RGB24: B,G,R bytes
        t = rows[:, : 3 * w].reshape(h, w, 3)
        t[:, :, 0] = vq; t[:, :, 1] = uq; t[:, :, 2] = yq
    elif fam == "rgb32":     # B,G,R,X
        t = rows[:, : 4 * w].reshape(h, w, 4)
        t[:, :, 0] = vq; t[:, :, 1] = uq; t[:, :, 2] = yq; t[:, :, 3] = 255
    elif fam == "r210":      # big-endian dword: 2 pad bits, R10, G10, B10
        word = (yq << 20) | (uq << 10) | vq
        rows.view(np.uint32)[:, :w] = word.astype(">u4").view(np.uint32)
    elif fam in ("rgb48", "bgr48"):
        t = rows.view(np.uint16)[:, : 3 * w].reshape(h, w, 3)
        a, b_, c_ = (yq, uq, vq) if fam == "rgb48" else (vq, uq, yq)
        t[:, :, 0] = a; t[:, :, 1] = b_; t[:, :, 2] = c_
    elif fam == "bgra64":
        t = rows.view(np.uint16)[:, : 4 * w].reshape(h, w, 4)
        t[:, :, 0] = vq; t[:, :, 1] = uq; t[:, :, 2] = yq; t[:, :, 3] = 65535
    elif fam == "b64a":      # big-endian words A,R,G,B
        t = rows.view(np.uint16)[:, : 4 * w].reshape(h, w, 4)
        be = lambda a: a.astype(np.uint16).byteswap()
        t[:, :, 0] = 65535; t[:, :, 1] = be(yq); t[:, :, 2] = be(uq); t[:, :, 3] = be(vq)
    else:                    # gray
        dt = np.uint8 if nbytes == 1 else np.uint16
        rows.view(dt)[:, :w] = yq.astype(dt)
    return buf, pitch


def make_frame(cformat, w, h, kind="noise", seed=0, pitch=None, full_range=False):
    """Return (uint8 buffer in the reference sample layout, pitch)."""
    if cformat in PACKED:
        return _make_packed(cformat, w, h, kind, seed, pitch, full_range)
    planes, nbytes, dw, dh, bits, msb, v_first = FORMATS[cformat]
    if pitch is None:
        pitch = default_pitch(cformat, w)
    cw, ch = w // dw, h // dh
    y, u, v = _planes_float(kind, w, h, cw, ch, SEED_BASE + seed)
    code_bits = 8 if nbytes == 1 else (10 if bits == 10 else 16)
    scale = 1 << (code_bits - 8)
    if full_range:
        ylo, yhi, clo, chi = 0, (1 << code_bits) - 1, 0, (1 << code_bits) - 1
    else:
        ylo, yhi, clo, chi = 16 * scale, 235 * scale, 16 * scale, 240 * scale
    yq = np.clip(np.floor(ylo + y * (yhi - ylo) + 0.5), ylo, yhi).astype(np.uint32)
    uq = np.clip(np.floor(clo + u * (chi - clo) + 0.5), clo, chi).astype(np.uint32)
    vq = np.clip(np.floor(clo + v * (chi - clo) + 0.5), clo, chi).astype(np.uint32)
    if msb and code_bits == 10:
        yq, uq, vq = yq << 6, uq << 6, vq << 6
    dt = np.uint8 if nbytes == 1 else np.uint16
    cpitch = pitch // dw if planes == 3 else pitch
    total = pitch * h + (cpitch * ch * (2 if planes == 3 else 1))
    buf = np.zeros(total, dtype=np.uint8)
    yview = buf[: pitch * h].reshape(h, pitch).view(dt)
    yview[:, :w] = yq.astype(dt)
    off = pitch * h
    if planes == 2:
        cview = buf[off: off + cpitch * ch].reshape(ch, cpitch).view(dt)
        cview[:, 0:2 * cw:2] = uq.astype(dt)
        cview[:, 1:2 * cw:2] = vq.astype(dt)
    else:
        first, second = (vq, uq) if v_first else (uq, vq)
        p1 = buf[off: off + cpitch * ch].reshape(ch, cpitch).view(dt)
        p1[:, :cw] = first.astype(dt)
        off2 = off + cpitch * ch
        p2 = buf[off2: off2 + cpitch * ch].reshape(ch, cpitch).view(dt)
        p2[:, :cw] = second.astype(dt)
    return buf, pitch


# ------------------------------------------------------------------------------------------------
# synthetic Dolby Vision RPUs — plain dicts in the shape api.DoviMetadata.from_dict() takes
# (fields of MediaSideDataDOVIMetadata, Include/IMediaSideData.h:154-330).  The numbers are made up but
# plausible for a profile-5 (IPTPQc2) stream: fixed-point coefficients with 23 fractional bits, 10-bit base layer.
# ------------------------------------------------------------------------------------------------
def _nits_to_pq12(nits):
    m1, m2 = 2610 / 16384, 2523 / 4096 * 128
    c1, c2, c3 = 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
    y = (nits / 10000.0) ** m1
    return int(round(((c1 + c2 * y) / (1 + c3 * y)) ** m2 * 4095))


def dovi_metadata(kind="poly", l1=False, l2=(), l3=False):
    """kind: 'poly' (8 luma pieces of order 1/2, linear chroma), 'mmr' (chroma through one order-3 MMR piece each),
    'mixed' (a curve mixing polynomial and two MMR pieces of different order), 'identity'.
    l2: iterable of target nits for which level-2 trim blocks are generated."""
    den = 23
    one = 1 << den

    def fx(x):
        return int(round(x * one))

    ident = dict(pivots=[0, 1023], pieces=[dict(order=1, poly=[0, one, 0])])
    if kind == "identity":
        curves = [ident, ident, ident]
    else:
        # luma: a gentle S-curve split in 8 pieces, continuous at the pivots (order 0 quirks included: piece 3 is order 1)
        piv = [0, 64, 160, 300, 460, 620, 780, 900, 1023]
        pieces = []
        for i in range(8):
            x0 = piv[i] / 1023.0
            a = 0.004 * (i - 3.5)                  # curvature changes sign across the range
            b = 1.0 - 0.03 * (i - 3.5)
            c = x0 - (a * x0 + b) * x0 + 0.002 * i
            if i == 3:
                pieces.append(dict(order=1, poly=[fx(c), fx(b), fx(0.75)]))      # x^2 term ignored for order 1
            else:
                pieces.append(dict(order=2, poly=[fx(c), fx(b), fx(a)]))
        luma = dict(pivots=piv, pieces=pieces)
        if kind == "poly":
            cu = dict(pivots=[0, 512, 1023], pieces=[dict(order=1, poly=[fx(0.01), fx(0.98), 0]),
                                                     dict(order=2, poly=[fx(-0.015), fx(1.06), fx(-0.04)])])
            cv = dict(pivots=[0, 1023], pieces=[dict(order=1, poly=[fx(-0.005), fx(1.01), 0])])
            curves = [luma, cu, cv]
        else:
            def mmr_piece(comp, order, bias):
                rows = []
                for o in range(order):
                    w = [0.0] * 7
                    if o == 0:
                        w[comp] = 0.97
                        w[0] += 0.02
                        w[3] = 0.015; w[4] = -0.01; w[5] = 0.02; w[6] = -0.03
                    elif o == 1:
                        w[comp] = 0.04; w[3] = -0.02; w[5] = 0.01; w[6] = 0.05
                    else:
                        w[comp] = -0.015; w[4] = 0.03; w[6] = -0.02
                    rows.append([fx(x) for x in w])
                return dict(order=order, constant=fx(bias), mmr=rows)
            if kind == "mmr":
                cu = dict(pivots=[0, 1023], pieces=[mmr_piece(1, 3, -0.004)])
                cv = dict(pivots=[0, 1023], pieces=[mmr_piece(2, 3, 0.003)])
            elif kind == "mixed":
                cu = dict(pivots=[0, 300, 700, 1023],
                          pieces=[dict(order=1, poly=[fx(0.004), fx(0.99), 0]), mmr_piece(1, 1, 0.002), mmr_piece(1, 3, -0.003)])
                cv = dict(pivots=[0, 512, 1023], pieces=[mmr_piece(2, 2, 0.001), mmr_piece(2, 2, -0.002)])
            else:
                raise ValueError(kind)
            curves = [luma, cu, cv]
    # IPT -> L'M'S' (scaled by 1/8192 in real RPUs) and the crosstalk-removing "rgb_to_lms" of profile 5
    # (chroma gains reduced to 0.3x so that noise frames, whose chroma spans the whole code range, stay mostly unclipped)
    ycc = [x / 8192.0 for x in (8192, 240, 504, 8192, -280, 327, 8192, 80, -1663)]
    c = 0.02
    xt = np.array([[1 - 2 * c, c, c], [c, 1 - 2 * c, c], [c, c, 1 - 2 * c]])
    lms = (np.round(np.linalg.inv(xt) * 16384) / 16384).reshape(-1).tolist()
    md = dict(bl_bit_depth=10, coef_log2_denom=den, source_max_pq=_nits_to_pq12(4000),
              ycc_to_rgb_matrix=ycc, ycc_to_rgb_offset=[0.0, 0.5, 0.5], rgb_to_lms_matrix=lms, curves=curves, l2=[])
    for i, nits in enumerate(l2):
        md["l2"].append(dict(target_max_pq=_nits_to_pq12(nits), trim_slope=2048 + 60 - 25 * i, trim_offset=2048 - 12 + 7 * i,
                             trim_power=2048 + 40 - 30 * i, trim_chroma_weight=2048 + 100 - 60 * i,
                             trim_saturation_gain=2048 - 50 + 45 * i))
    if l1:
        md.update(l1_present=1, l1_min_pq=62, l1_max_pq=_nits_to_pq12(1400), l1_avg_pq=_nits_to_pq12(60))
    if l3:
        md.update(l3_present=1, l3_min_pq_offset=2048 - 10, l3_max_pq_offset=2048 + 90, l3_avg_pq_offset=2048 + 35)
    return md
