"""ref_hostmath.py — the REFERENCE's own host-side parameter maths, compiled for the CPU.  TEST INFRASTRUCTURE ONLY.

oracle/_ref/libref_hostmath.so holds, compiled from the text of /root/reference where it lies (nothing is copied into the repo):

  SetShaderDoviCurves / SetShaderDoviCurvesPoly   DX11VideoProcessor.cpp:990-1141   the PS_DOVI_CURVE / PS_DOVI_POLY_CURVE cbuffers
  SetHDR10ShaderParams                            DX11VideoProcessor.cpp:907-923    HDRParamsConstantBuffer_t (defaults and clamps)
  the level 1 / 3 / 2 block of CopySample         DX11VideoProcessor.cpp:2326-2469  L1 nits, level-2 trim selection / interpolation
  SetDolbyVisionDynamicParams                     DX11VideoProcessor.cpp:953-960    DoViDynamicConstantsBuffer_t
  SpecifyExtendedFormat                           Helper.cpp:1169-1211
  CopyFrameV210                                   Helper.cpp:709-748
  CopyPlaneAsIs, CopyFrameRGB24 / R210 / RGB48 / BGR48 / BGRA64 / B64A   Helper.cpp:414-483,541-566,600-677,769-787 (interleaved RGB uploads)

The extraction is MECHANICAL, like hlsl2cpp.py's: this file knows where a region starts and ends (an anchor line each) and what
class members it mentions, not what it computes.  Member functions of CDX11VideoProcessor are cut at the point where the Direct3D
buffer plumbing begins and become methods of a stand-in class that owns exactly the members the text names; struct definitions
come out of the reference headers by the same cut.  ours: the anchors, the stand-in class, the extern "C" wrappers that translate
the oracle's plain `orc_dovi` into the reference's MediaSideDataDOVIMetadata (the real Include/IMediaSideData.h) field by field.
"""
import ctypes as C
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.dirname(HERE)
REF_DIR = os.path.join(ORACLE, "_ref")
GEN_DIR = os.path.join(REF_DIR, "gen")
LIB = os.path.join(REF_DIR, "libref_hostmath.so")
REFERENCE_ROOT = os.environ.get("MPCVR_REFERENCE_ROOT", "/root/reference")


def have_reference():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "Source", "DX11VideoProcessor.cpp"))


def available():
    return os.path.exists(LIB) or have_reference()


def _cut(text, start, end, include_end=False, after=0):
    """text[start anchor .. end anchor): both are literal substrings; the search for `end` begins behind `start`."""
    a = text.index(start, after)
    b = text.index(end, a + len(start))
    return text[a:b + (len(end) if include_end else 0)]


def _braces(text, start):
    """The brace-balanced block that begins at the first '{' at or behind the literal anchor `start` (anchor included)."""
    a = text.index(start)
    i = text.index("{", a)
    depth = 0
    for j in range(i, len(text)):
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return text[a:j + 1]
    raise ValueError(start)


def generate():
    src = os.path.join(REFERENCE_ROOT, "Source")
    vp = open(os.path.join(src, "DX11VideoProcessor.cpp"), encoding="utf-8", errors="replace").read()
    vph = open(os.path.join(src, "DX11VideoProcessor.h"), encoding="utf-8", errors="replace").read()
    hp = open(os.path.join(src, "Helper.cpp"), encoding="utf-8", errors="replace").read()

    structs = "\n".join([
        _braces(vph, "struct HDRParamsConstantBuffer_t") + ";",
        _braces(vph, "struct DoViDynamicConstantsBuffer_t") + ";",
        _braces(vph, "struct DoviExtensionMetadata_t") + ";",
    ])
    free_fns = "\n".join([
        _braces(hp, "void CopyFrameV210("),
        _braces(hp, "DXVA2_ExtendedFormat SpecifyExtendedFormat("),
        # the upload repacks of the interleaved RGB formats (GetCopyPlaneFunction, Helper.cpp:377-412; the plain C versions)
        _braces(hp, "void CopyPlaneAsIs("),
        _braces(hp, "void CopyFrameRGB24("),
        _braces(hp, "void CopyFrameRGB48("),
        _braces(hp, "void CopyFrameBGR48("),
        _braces(hp, "void CopyFrameBGRA64("),
        _braces(hp, "void CopyFrameB64A("),
        _braces(hp, "void CopyFrameR210("),
    ])
    # member functions, cut where the D3D buffer plumbing starts
    poly = _cut(vp, "PS_DOVI_POLY_CURVE polyCurves[3] = {};", "HRESULT hr;", after=vp.index("CDX11VideoProcessor::SetShaderDoviCurvesPoly()"))
    full = _cut(vp, "PS_DOVI_CURVE cbuffer[3] = {};", "HRESULT hr;", after=vp.index("HRESULT CDX11VideoProcessor::SetShaderDoviCurves()"))
    hdr_sig = re.search(r"void CDX11VideoProcessor::SetHDR10ShaderParams\((.*?)\)\s*\{", vp, re.S)
    hdr_body = _cut(vp, hdr_sig.group(0), "if (!m_pPSHDR10ToneMapping)")[len(hdr_sig.group(0)):]
    dyn = _cut(vp, "const DoViDynamicConstantsBuffer_t cbuffer = {", "};", include_end=True,
               after=vp.index("void CDX11VideoProcessor::SetDolbyVisionDynamicParams()"))
    levels = _cut(vp, "// based on libplacebo source code", "SetDolbyVisionDynamicParams();",
                  after=vp.index("const bool bMasteringLuminanceChanged"))

    out = f"""// GENERATED by oracle/ref_hlsl/ref_hostmath.py from {REFERENCE_ROOT}/Source — reference text between the markers, do not edit.
#include "stdafx.h"
#include <algorithm>
#include <cmath>
#include "Helper.h"
#include "Shaders.h"
#include "hostmath_stub.h"
extern "C" {{
#include "mpcvr_oracle.h"
}}

// ---- DX11VideoProcessor.h ----
{structs}

// ---- Helper.cpp ----
{free_fns}

struct RefHost {{
    struct {{ bool bValid = true; bool bHasMMR = false; MediaSideDataDOVIMetadata msd = {{}}; }} m_Dovi;
    DoviExtensionMetadata_t m_DoviExtensionMetadata;
    int m_iHdrDisplayMaxNits = 1000;
    bool m_bHdrPassthroughSupport = false, m_bHdrLocalToneMapping = false;
    void UpdateStatsStatic() {{}}

    void curves_poly(void* out__) {{
        // ---- CDX11VideoProcessor::SetShaderDoviCurvesPoly ----
        {poly}
        memcpy(out__, &polyCurves, sizeof(polyCurves));
    }}
    void curves(void* out__) {{
        // ---- CDX11VideoProcessor::SetShaderDoviCurves ----
        {full}
        memcpy(out__, &cbuffer, sizeof(cbuffer));
    }}
    void hdr10({hdr_sig.group(1)}, void* out__) {{
        // ---- CDX11VideoProcessor::SetHDR10ShaderParams ----
        {hdr_body}
        memcpy(out__, &cbuffer, sizeof(cbuffer));
    }}
    void levels(const MediaSideDataDOVIMetadata* pDOVIMetadata) {{
        // ---- CDX11VideoProcessor::CopySample, the level 1 / 3 / 2 block ----
        {levels}
    }}
    void dynamic(void* out__) {{
        // ---- CDX11VideoProcessor::SetDolbyVisionDynamicParams ----
        {dyn}
        memcpy(out__, &cbuffer, sizeof(cbuffer));
    }}
}};

// ---- ours from here: orc_dovi -> MediaSideDataDOVIMetadata, field by field (what the adapter does with LAV's side data) ----
static void fill_msd(const orc_dovi* d, MediaSideDataDOVIMetadata* m)
{{
    memset(m, 0, sizeof(*m));
    m->Header.bl_bit_depth = d->bl_bit_depth;
    m->Header.coef_log2_denom = d->coef_log2_denom;
    for (int c = 0; c < 3; c++) {{
        auto& o = m->Mapping.curves[c];
        const orc_dovi_curve& s = d->curves[c];
        o.num_pivots = s.num_pivots;
        for (int i = 0; i < 9; i++) o.pivots[i] = s.pivots[i];
        for (int i = 0; i < 8; i++) {{
            o.mapping_idc[i] = s.mapping_idc[i]; o.poly_order[i] = s.poly_order[i]; o.mmr_order[i] = s.mmr_order[i];
            o.mmr_constant[i] = s.mmr_constant[i];
            for (int k = 0; k < 3; k++) o.poly_coef[i][k] = s.poly_coef[i][k];
            for (int j = 0; j < 3; j++) for (int k = 0; k < 7; k++) o.mmr_coef[i][j][k] = s.mmr_coef[i][j][k];
        }}
    }}
    for (int i = 0; i < 9; i++) {{ m->ColorMetadata.ycc_to_rgb_matrix[i] = d->ycc_to_rgb_matrix[i]; m->ColorMetadata.rgb_to_lms_matrix[i] = d->rgb_to_lms_matrix[i]; }}
    for (int i = 0; i < 3; i++) m->ColorMetadata.ycc_to_rgb_offset[i] = d->ycc_to_rgb_offset[i];
    m->ColorMetadata.source_max_pq = d->source_max_pq;
    unsigned n = 0;
    if (d->l1_present) {{ auto& e = m->Extensions[n++]; e.level = 1; e.Level1.min_pq = d->l1_min_pq; e.Level1.max_pq = d->l1_max_pq; e.Level1.avg_pq = d->l1_avg_pq; }}
    for (unsigned i = 0; i < d->n_l2 && n < LAV_DOVI_MAX_EXTENSIONS - 1; i++) {{
        auto& e = m->Extensions[n++]; e.level = 2;
        e.Level2.target_max_pq = d->l2[i].target_max_pq; e.Level2.trim_slope = d->l2[i].trim_slope; e.Level2.trim_offset = d->l2[i].trim_offset;
        e.Level2.trim_power = d->l2[i].trim_power; e.Level2.trim_chroma_weight = d->l2[i].trim_chroma_weight; e.Level2.trim_saturation_gain = d->l2[i].trim_saturation_gain;
    }}
    if (d->l3_present) {{ auto& e = m->Extensions[n++]; e.level = 3; e.Level3.min_pq_offset = d->l3_min_pq_offset; e.Level3.max_pq_offset = d->l3_max_pq_offset; e.Level3.avg_pq_offset = d->l3_avg_pq_offset; }}
}}

extern "C" {{
// out: 3 x PS_DOVI_CURVE (has_mmr) or 3 x PS_DOVI_POLY_CURVE as the reference would upload them; returns the byte count
int ref_dovi_curves(const orc_dovi* d, int poly, void* out)
{{
    RefHost h; fill_msd(d, &h.m_Dovi.msd);
    if (poly) {{ h.curves_poly(out); return (int)(3 * sizeof(PS_DOVI_POLY_CURVE)); }}
    h.curves(out); return (int)(3 * sizeof(PS_DOVI_CURVE));
}}
// k5 = DoViDynamicConstantsBuffer_t's five floats, *enabled its flag; l1 = {{min, max, avg}} nits as stored, returns L1.present
int ref_dovi_levels(const orc_dovi* d, int display_nits, float k5[5], int* enabled, unsigned l1[3])
{{
    RefHost h; fill_msd(d, &h.m_Dovi.msd);
    h.m_iHdrDisplayMaxNits = display_nits;
    h.levels(&h.m_Dovi.msd);
    DoViDynamicConstantsBuffer_t cb; h.dynamic(&cb);
    k5[0] = cb.trim_chroma_weight; k5[1] = cb.trim_saturation_gain; k5[2] = cb.trim_slope; k5[3] = cb.trim_offset; k5[4] = cb.trim_power;
    *enabled = (int)cb.enabled;
    l1[0] = h.m_DoviExtensionMetadata.L1.min_pq; l1[1] = h.m_DoviExtensionMetadata.L1.max_pq; l1[2] = h.m_DoviExtensionMetadata.L1.avg_pq;
    return h.m_DoviExtensionMetadata.L1.present ? 1 : 0;
}}
// out6 = HDRParamsConstantBuffer_t's first six words (five floats and the selection)
void ref_hdr10_params(float mn, float mx, float cll, float fall, float disp, int sel, void* out6)
{{
    RefHost h; HDRParamsConstantBuffer_t cb; h.hdr10(mn, mx, cll, fall, disp, sel, &cb);
    memcpy(out6, &cb, 24);
}}
unsigned ref_specify_extfmt(unsigned exfmt, int cstype, int subsampling, unsigned w, unsigned hgt)
{{
    DXVA2_ExtendedFormat e; e.value = (LONG)exfmt;
    FmtConvParams_t f = {{}};
    f.CSType = (ColorSystem_t)cstype; f.Subsampling = subsampling;
    return (unsigned)SpecifyExtendedFormat(e, f, w, hgt).value;
}}
void ref_copy_frame_v210(unsigned lines, unsigned char* dst, unsigned dst_pitch, const unsigned char* src, int src_pitch)
{{
    CopyFrameV210(lines, dst, dst_pitch, src, src_pitch);
}}
// kind: 0 CopyPlaneAsIs, 1 CopyFrameRGB24, 2 CopyFrameR210, 3 CopyFrameRGB48, 4 CopyFrameBGR48, 5 CopyFrameBGRA64, 6 CopyFrameB64A
// (the order of the oracle's RPK_* codes)
void ref_copy_frame_rgb(int kind, unsigned lines, unsigned char* dst, unsigned dst_pitch, const unsigned char* src, int src_pitch)
{{
    switch (kind) {{
    case 0: CopyPlaneAsIs(lines, dst, dst_pitch, src, src_pitch); break;
    case 1: CopyFrameRGB24(lines, dst, dst_pitch, src, src_pitch); break;
    case 2: CopyFrameR210(lines, dst, dst_pitch, src, src_pitch); break;
    case 3: CopyFrameRGB48(lines, dst, dst_pitch, src, src_pitch); break;
    case 4: CopyFrameBGR48(lines, dst, dst_pitch, src, src_pitch); break;
    case 5: CopyFrameBGRA64(lines, dst, dst_pitch, src, src_pitch); break;
    case 6: CopyFrameB64A(lines, dst, dst_pitch, src, src_pitch); break;
    }}
}}
}}
"""
    os.makedirs(GEN_DIR, exist_ok=True)
    path = os.path.join(GEN_DIR, "hostmath.cpp")
    with open(path, "w") as f:
        f.write(out)
    return path


def build():
    """oracle/_ref/libref_hostmath.so from the reference tree (needs it mounted)."""
    cpp = generate()
    stub = os.path.join(HERE, "stub_shadergen")
    src = os.path.join(REFERENCE_ROOT, "Source")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-msse2", "-ffp-contract=off", "-fPIC", "-w", "-shared", "-o", LIB,
                           "-iquote", stub, "-I", stub, "-I", HERE, "-I", src, "-I", ORACLE, cpp])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            if not have_reference():
                return None
            build()
        L = C.CDLL(LIB)
        L.ref_dovi_curves.restype = C.c_int
        L.ref_dovi_curves.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ref_dovi_levels.restype = C.c_int
        L.ref_dovi_levels.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint)]
        L.ref_hdr10_params.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]
        L.ref_specify_extfmt.restype = C.c_uint
        L.ref_specify_extfmt.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_uint, C.c_uint]
        L.ref_copy_frame_v210.argtypes = [C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
        L.ref_copy_frame_rgb.argtypes = [C.c_int, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
        _lib = L
    return _lib


if __name__ == "__main__":
    print("built", build())
