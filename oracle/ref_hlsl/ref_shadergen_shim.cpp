// ref_shadergen_shim.cpp — runs the REAL convert-shader generator (Source/Shaders.cpp:593-930 GetShaderConvertColor,
// :82-529 ShaderGetPixels, :531-589 the Dolby Vision emitters) and hands back the HLSL text it would give D3DCompile.
// TEST INFRASTRUCTURE ONLY.  This file is ours; Shaders.cpp and csputils.cpp are compiled from /root/reference where they lie
// (oracle/Makefile ref-hlsl).  CompileShader() (Shaders.cpp:29-63) loads "d3dcompiler_47.dll" and calls its D3DCompile: the
// LoadLibraryW / GetProcAddress below hand it a function that keeps the source text instead of compiling it.
#include "stdafx.h"
#include <D3Dcompiler.h>
#include "Helper.h"
#include "resource.h"
#include "IVideoRenderer.h"
#include "Shaders.h"
#include <cstdio>
#include <vector>

static std::string g_ref_root = "/root/reference";

struct TextBlob : ID3DBlob {
    std::string s;
    void* GetBufferPointer() override { return (void*)s.data(); }
    size_t GetBufferSize() override { return s.size(); }
    void Release() override { delete this; }
};

static HRESULT CaptureCompile(const void* src, size_t n, const char*, const D3D_SHADER_MACRO*, void*, const char*, const char*,
                              UINT, UINT, ID3DBlob** code, ID3DBlob**)
{
    TextBlob* b = new TextBlob;
    b->s.assign((const char*)src, n);
    *code = b;
    return S_OK;
}
HMODULE LoadLibraryW(const wchar_t*) { return (HMODULE)1; }
void* GetProcAddress(HMODULE, const char*) { return (void*)&CaptureCompile; }

// the HLSL includes are RCDATA resources of the DLL (Source/res/MpcVideoRenderer.rc2); here they are read from the tree
HRESULT GetDataFromResource(LPVOID& data, DWORD& size, UINT resid)
{
    static std::vector<char> keep[3];
    const char* rel; int slot;
    switch (resid) {
    case IDF_HLSL_ST2084:           rel = "/Shaders/convert/st2084.hlsl"; slot = 0; break;
    case IDF_HLSL_HLG:              rel = "/Shaders/convert/hlg.hlsl"; slot = 1; break;
    case IDF_HLSL_HDR_TONE_MAPPING: rel = "/Shaders/convert/hdr_tone_mapping.hlsl"; slot = 2; break;
    default: return E_FAIL;
    }
    FILE* f = fopen((g_ref_root + rel).c_str(), "rb");
    if (!f) return E_FAIL;
    keep[slot].clear();
    char buf[4096]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) keep[slot].insert(keep[slot].end(), buf, buf + n);
    fclose(f);
    data = keep[slot].data(); size = (DWORD)keep[slot].size();
    return S_OK;
}

extern "C" {

void ref_shadergen_set_root(const char* root) { g_ref_root = root; }

// planes / subsampling: the DX11 plane count and Subsampling of the format's s_FmtConvMapping row (Helper.cpp:309-359).
// dovi: 0 none, 1 polynomial curves only, 2 with an MMR piece; rgb_to_lms: ColorMetadata.rgb_to_lms_matrix.
// Returns the text length (0 on failure); the text is copied to out (cap bytes, NUL-terminated).
int ref_convert_shader_text(int cformat, int planes, int subsampling, unsigned width, int texW, int texH, unsigned exfmt,
                            int chroma_scaling, int convert_type, int blend_deint, int dovi, const double* rgb_to_lms,
                            char* out, int cap)
{
    DX11PlaneConfig_t pc = {1, planes >= 2 ? 1 : 0, planes >= 3 ? 1 : 0, 1, 1};
    FmtConvParams_t fp = {};
    fp.cformat = (ColorFormat_t)cformat;
    fp.str = L"";
    fp.pDX11Planes = &pc;
    fp.Subsampling = subsampling;
    DXVA2_ExtendedFormat ex; ex.value = (LONG)exfmt;
    static MediaSideDataDOVIMetadata md;
    const MediaSideDataDOVIMetadata* pmd = nullptr;
    if (dovi) {
        md = MediaSideDataDOVIMetadata();
        for (auto& c : md.Mapping.curves) c.num_pivots = 2;
        if (dovi == 2) md.Mapping.curves[1].mapping_idc[0] = 1;
        for (int i = 0; i < 9; i++) md.ColorMetadata.rgb_to_lms_matrix[i] = rgb_to_lms ? rgb_to_lms[i] : (i % 4 == 0);
        pmd = &md;
    }
    const RECT rc = {0, 0, texW, texH};
    ID3DBlob* blob = nullptr;
    HRESULT hr = GetShaderConvertColor(true, width, texW, texH, rc, fp, ex, pmd, chroma_scaling, convert_type, blend_deint != 0, &blob);
    if (FAILED(hr) || !blob) return 0;
    const int n = (int)blob->GetBufferSize();
    if (out && cap > 0) {
        const int m = n < cap - 1 ? n : cap - 1;
        memcpy(out, blob->GetBufferPointer(), m);
        out[m] = 0;
    }
    blob->Release();
    return n;
}

}  // extern "C"
