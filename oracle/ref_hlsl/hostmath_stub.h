// hostmath_stub.h — stand-in (TEST INFRASTRUCTURE ONLY; ours) for the Windows SDK enumerators Source/Helper.cpp's
// SpecifyExtendedFormat names (dxva2api.h; values as documented for DXVA2_ExtendedFormat) and for the few library calls the
// extracted host code makes that libstdc++ 11 / the stub stdafx.h do not provide.
#pragma once
#include <cmath>
enum {
    DXVA2_VideoChromaSubsampling_Unknown = 0,
    DXVA2_NominalRange_Unknown = 0, DXVA2_NominalRange_0_255 = 1, DXVA2_NominalRange_16_235 = 2,
    DXVA2_VideoTransferMatrix_Unknown = 0, DXVA2_VideoTransferMatrix_BT709 = 1, DXVA2_VideoTransferMatrix_BT601 = 2,
    DXVA2_VideoLighting_Unknown = 0, DXVA2_VideoLighting_dim = 3,
    DXVA2_VideoPrimaries_Unknown = 0, DXVA2_VideoPrimaries_BT709 = 2,
    DXVA2_VideoTransFunc_Unknown = 0,
};
