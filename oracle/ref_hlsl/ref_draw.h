// ref_draw.h — the draw loop that executes a compiled reference pixel shader over a viewport.  TEST INFRASTRUCTURE ONLY.
// Models what Direct3D does around the shader (ours, not reference code): one invocation per pixel centre of the viewport,
// TEXCOORD interpolated linearly between the quad's corner values (evaluated in double, rounded once to fp32), the returned
// colour converted to the render-target format (UNORM: round to nearest after clamping; fp16: round to nearest even).
#pragma once
#include <omp.h>
#include "hlsl_shim.h"

namespace hlsl {

float rt_round(float x, int fmt, int channel);      // ref_runtime.cpp

template <class S> void run_draw(const RefDraw& d)
{
    const double ax = d.uv[0][0], ay = d.uv[0][1];
    const double bx = (double)d.uv[1][0] - ax, by = (double)d.uv[1][1] - ay;      // along screen x
    const double cx = (double)d.uv[2][0] - ax, cy = (double)d.uv[2][1] - ay;      // along screen y
#pragma omp parallel num_threads(d.threads > 0 ? d.threads : omp_get_max_threads())
    {
        for (int i = 0; i < 4; i++) { g_cb[i].p = d.cb[i]; g_cb[i].n = d.cb_words[i]; g_cb[i].pos = 0; }
        S sh;
        sh.bind__(d);
#pragma omp for schedule(static)
        for (int py = 0; py < d.vp_h; py++) {
            const double b = ((double)py + 0.5) / (double)d.vp_h;
            for (int px = 0; px < d.vp_w; px++) {
                const double a = ((double)px + 0.5) / (double)d.vp_w;
                typename S::PS_INPUT in;
                in.Pos = float4((float)(d.vp_x + px) + 0.5f, (float)(d.vp_y + py) + 0.5f, 0.0f, 1.0f);
                in.Tex = float2((float)(ax + bx * a + cx * b), (float)(ay + by * a + cy * b));
                const float4 o = sh.main(in);
                const int X = d.vp_x + px, Y = d.vp_y + py;
                if (X < 0 || Y < 0 || X >= d.rt.w || Y >= d.rt.h) continue;
                float* q = d.rt.data + ((size_t)Y * d.rt.w + X) * 4;
                for (int c = 0; c < 4; c++) q[c] = rt_round(o.v[c], d.rt_fmt, c);
            }
        }
    }
}

}  // namespace hlsl
