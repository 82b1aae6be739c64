// Helper.h — stand-in (TEST INFRASTRUCTURE ONLY; ours) for Source/Helper.h, which needs dxva2api.h / mfobjects.h / DirectShow.
// Restated: ColorFormat_t (Helper.h:86-127, same order), the fields of FmtConvParams_t Shaders.cpp reads (Helper.h:152-166),
// DXVA2_ExtendedFormat's bit layout and the enumerators used (Windows SDK dxva2api.h / mfobjects.h, not in the repository).
// Real reference headers are still used for everything they can give: csputils.h and Include/IMediaSideData.h.
#pragma once
#include "csputils.h"
#include "../Include/IMediaSideData.h"

typedef int D3DFORMAT;
typedef int DXGI_FORMAT;

enum ColorFormat_t {
    CF_NONE = 0, CF_NV12, CF_P010, CF_P016, CF_YUY2, CF_UYVY, CF_P210, CF_P216, CF_Y210, CF_Y216, CF_V210, CF_AYUV, CF_Y410, CF_Y416,
    CF_YV12, CF_YV16, CF_YV24, CF_YUV420P8, CF_YUV422P8, CF_YUV444P8, CF_YUV420P10, CF_YUV420P16, CF_YUV422P10, CF_YUV422P16,
    CF_YUV444P10, CF_YUV444P16, CF_GBRP8, CF_GBRP10, CF_GBRP16, CF_RGB24, CF_XRGB32, CF_ARGB32, CF_r210, CF_RGB48, CF_BGR48,
    CF_BGRA64, CF_B64A, CF_Y8, CF_Y10, CF_Y16,
};
enum ColorSystem_t { CS_YUV, CS_RGB, CS_GRAY };
struct DX9PlaneConfig { D3DFORMAT FmtPlane1, FmtPlane2, FmtPlane3; UINT div_chroma_w, div_chroma_h; };
struct DX11PlaneConfig_t { DXGI_FORMAT FmtPlane1, FmtPlane2, FmtPlane3; UINT div_chroma_w, div_chroma_h; };
struct FmtConvParams_t {
    ColorFormat_t      cformat;
    const wchar_t*     str;
    D3DFORMAT          DXVA2Format;
    D3DFORMAT          D3DFormat;
    DX9PlaneConfig*    pDX9Planes;
    DXGI_FORMAT        VP11Format;
    DXGI_FORMAT        DX11Format;
    DX11PlaneConfig_t* pDX11Planes;
    int                Packsize;
    int                PitchCoeff;
    ColorSystem_t      CSType;
    int                Subsampling;
    int                CDepth;
};

struct DXVA2_ExtendedFormat {
    union {
        struct {
            UINT SampleFormat : 8;
            UINT VideoChromaSubsampling : 4;
            UINT NominalRange : 3;
            UINT VideoTransferMatrix : 3;
            UINT VideoLighting : 4;
            UINT VideoPrimaries : 5;
            UINT VideoTransferFunction : 5;
        };
        LONG value;
    };
};
enum {
    DXVA2_VideoChromaSubsampling_MPEG1 = 1, DXVA2_VideoChromaSubsampling_MPEG2 = 5, DXVA2_VideoChromaSubsampling_Cosited = 7,
    DXVA2_VideoTransFunc_10 = 1, DXVA2_VideoTransFunc_18 = 2, DXVA2_VideoTransFunc_20 = 3, DXVA2_VideoTransFunc_22 = 4,
    DXVA2_VideoTransFunc_709 = 5, DXVA2_VideoTransFunc_240M = 6, DXVA2_VideoTransFunc_sRGB = 7, DXVA2_VideoTransFunc_28 = 8,
    MFVideoTransFunc_26 = 14, MFVideoTransFunc_2084 = 15, MFVideoTransFunc_HLG = 16,
    MFVideoPrimaries_BT2020 = 9,
};
