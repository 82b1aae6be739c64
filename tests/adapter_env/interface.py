"""tests/adapter_env/interface.py — TEST INFRASTRUCTURE: `class CVideoProcessor`'s method list cut out of the reference header.

What is REAL: every line of the public method list of Source/VideoProcessor.h (from the destructor to the protected helpers) and the protected
data members the adapter touches, verbatim, at the moment of the call; Source/IVideoRenderer.h and Include/IMediaSideData.h are included as they are.
What is a stand-in: the Windows / DirectShow / ATL types those lines mention (win32_standins.h, dxva2api.h beside this file).  Used by
tests/test_adapter_compiles.py (syntax and overrides) and by build_adapter.py (the adapter LINKED and run: oracle/_ref/adapter_driver).
Needs /root/reference; nothing of the reference is stored in the repo."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

# protected members of CVideoProcessor the adapter reads or writes: their declarations are taken from the reference header, not restated
MEMBERS = ["m_srcParams", "m_srcWidth", "m_srcHeight", "m_srcPitch", "m_srcRect", "m_decExFmt", "m_videoRect", "m_windowRect", "m_iRotation", "m_bFlip", "m_hWnd", "m_Dovi",
           "m_nCurrentAdapter", "m_rtStart", "m_bAllowDeepColorBitmaps", "m_bDoubleFrames"]      # (the last four: read by inline methods inside the cut)


def available():
    return os.path.isdir(os.path.join(REF, "Source"))


def cut_interface():
    text = open(os.path.join(REF, "Source", "VideoProcessor.h"), encoding="utf-8", errors="replace").read()
    lines = text.split("\n")
    cls = next(i for i, l in enumerate(lines) if re.match(r"\s*class CVideoProcessor\b", l))
    first = next(i for i in range(cls, len(lines)) if "virtual ~CVideoProcessor()" in lines[i])
    last = next(i for i in range(first, len(lines)) if "inline bool SourceIsHDR10orHLG()" in lines[i])
    methods = lines[first:last]
    while methods and methods[-1].strip() in ("", "protected:"):
        methods.pop()
    assert any("virtual HRESULT Render(" in l for l in methods) and any("virtual HRESULT GetCurentImage(" in l for l in methods)
    priv = [l for l in lines[last:] if "virtual void UpdateStatsStatic()" in l]
    assert len(priv) == 1
    members = []
    for m in MEMBERS:
        hit = [i for i in range(cls, first) if re.search(r"\b%s\b" % m, lines[i])]
        assert hit, f"CVideoProcessor no longer has a member {m}"
        i = hit[0]
        if m == "m_Dovi":           # struct DOVIMetadata { ... } m_Dovi;
            j = max(k for k in range(cls, i + 1) if "struct DOVIMetadata" in lines[k])
            members += lines[j:i + 1]
        else:
            members.append(lines[i])
    check = next(l for l in lines[cls:first] if "bool CheckDoviMetadata(" in l)
    ctor = next(l for l in lines[cls:first + 1] if re.search(r"CVideoProcessor\(CMpcVideoRenderer\* pFilter\)", l))
    n_pure = sum(1 for l in methods if re.search(r"=\s*0\s*;", l))
    return methods, priv, members, check, ctor, n_pure


def header_text(methods=None):
    m, priv, members, check, ctor, _ = cut_interface()
    if methods is None:
        methods = m
    hdr = ["// generated from /root/reference/Source/VideoProcessor.h (tests/adapter_env/interface.py): every declaration below the stand-in includes is the reference's own line",
           "#pragma once", '#include "win32_standins.h"', '#include "IVideoRenderer.h"', '#include "../Include/IMediaSideData.h"',
           "enum : int { VP_DX9 = 9, VP_DX11 = 11 };", "class CMpcVideoRenderer;", "class CVideoProcessor", "{", "protected:",
           "\tCMpcVideoRenderer* m_pFilter = nullptr;"] + members + [check, ctor, "public:"] + methods + ["private:"] + priv + ["};", ""]
    return "\n".join(hdr)


def include_dirs(gen_dir, root):
    return ["-I", str(gen_dir), "-I", HERE, "-I", os.path.join(REF, "Source"), "-I", os.path.join(root, "include")]
