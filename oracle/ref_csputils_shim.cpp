// ref_csputils_shim.cpp — extern "C" access to the REAL reference parameter maths.
// TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with
// /root/reference/Source/csputils.cpp (read where it lies, never copied) into oracle/_ref/.
// Wraps: mp_get_csp_matrix (csputils.cpp:392-509), GetColorspaceGamutConversionMatrix (:549-557).
#include <cmath>
#include <cstdlib>
#include "csputils.h"   // resolved via -I/root/reference/Source

extern "C" {

// space/levels: mp_csp / mp_csp_levels numeric values; bits = input_bits = texture_bits
// (DX11VideoProcessor.cpp:845); brightness/contrast/hue/saturation already in csp_params units.
void ref_csp_matrix(int space, int levels, int bits, float brightness, float contrast,
                    float hue, float saturation, int gray, float m[9], float c[3])
{
    mp_csp_params p = {};          // levels_out = AUTO -> PC, is_float = false, as SetShaderConvertColorParams leaves them
    p.color = {};
    p.color.space = (mp_csp)space;
    p.color.levels = (mp_csp_levels)levels;
    p.brightness = brightness;
    p.contrast = contrast;
    p.hue = hue;
    p.saturation = saturation;
    p.gray = gray != 0;
    p.input_bits = p.texture_bits = bits;
    mp_cmat cm;
    mp_get_csp_matrix(&p, &cm);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) m[i * 3 + j] = cm.m[i][j];
        c[i] = cm.c[i];
    }
}

void ref_gamut_matrix(int prim_in, int prim_out, float m[9])
{
    float mm[3][3];
    GetColorspaceGamutConversionMatrix(mm, (mp_csp_prim)prim_in, (mp_csp_prim)prim_out);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) m[i * 3 + j] = mm[i][j];
}

}  // extern "C"
