// d3dcommon.h — stand-in (TEST INFRASTRUCTURE ONLY; ours) for what Source/Shaders.h needs.
#pragma once
#include "D3Dcompiler.h"
namespace DirectX { struct XMFLOAT4 { float x, y, z, w; }; }
