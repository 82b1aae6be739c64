// tests/adapter_env/dxva2api.h — TEST INFRASTRUCTURE ONLY (ours): what Source/IVideoRenderer.h needs from the Windows SDK header (nothing
// but the header's existence) and the DXVA2_ExtendedFormat bitfield the adapter fills (layout: SURVEY.md appendix B).
#pragma once
#include "win32_standins.h"
struct DXVA2_ExtendedFormat {
    union {
        struct { UINT SampleFormat : 8; UINT VideoChromaSubsampling : 4; UINT NominalRange : 3; UINT VideoTransferMatrix : 3; UINT VideoLighting : 4; UINT VideoPrimaries : 5; UINT VideoTransferFunction : 5; };
        LONG value;
    };
};
inline bool IsWindows8OrGreater() { return true; }
inline bool IsWindows10OrGreater() { return true; }
