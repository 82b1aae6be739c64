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


def splitmix64(n, seed):
    """n uint64 values of the SplitMix64 stream started at `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def default_pitch(cformat, w):
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


def make_frame(cformat, w, h, kind="noise", seed=0, pitch=None, full_range=False):
    """Return (uint8 buffer in the reference sample layout, pitch)."""
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
