// IVideoRenderer.h — stand-in (TEST INFRASTRUCTURE ONLY; ours).  The real header declares COM interfaces; Shaders.cpp reads
// only the chroma-scaling enum (Source/IVideoRenderer.h:47-52), restated here with the same values.
#pragma once
enum : int { CHROMA_Nearest = 0, CHROMA_Bilinear = 1, CHROMA_CatmullRom = 2 };
