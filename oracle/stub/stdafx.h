// Stub precompiled header for building the reference's csputils.cpp outside Windows.
// (The real Source/stdafx.h pulls in atlbase/d3d11; csputils.cpp itself needs only libm.)
#pragma once
#include <cmath>
#include <cstdlib>
