// mfma_owncol.hip — evidence for the matrix-core taps of vp_fused_mx.hip (v_mfma_f32_16x16x32_f16, gfx950).
//   (1) "own column" operand layout: with a block-diagonal weight fragment A, lane L's four results depend only on lane L's
//       eight B values (checked against a per-lane scalar reference, asymmetric data);
//   (2) fp16 subnormal operands: are they flushed by the matrix core?  (decides the power-of-two scales of the kernel);
//   (3) hi/lo weight split against the fp32 FMA chain of the shader on UNORM10 codes: max error and fp16-rounding flips;
//   (4) how MFMA and VALU work share a SIMD: the same instruction counts issued (a) by ONE wave, (b) by DIFFERENT waves of a
//       SIMD (an MFMA-only wave beside VALU-only waves), against each kind alone.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_owncol mfma_owncol.hip && ./mfma_owncol
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// every lane: its own 8 data values b[], the weight rows w[4][8] it wants applied; A fragment built block-diagonally
__global__ void k_owncol(const float *data, const float *wrow, float *out, int split)
{
    const int l = threadIdx.x;
    const bool diag = ((l & 15) >> 2) == (l >> 4);
    const int q = l & 3;
    h8 ah, al, b;
    for (int t = 0; t < 8; t++) {
        const float w = diag ? wrow[q * 8 + t] : 0.0f;
        const _Float16 hi = (_Float16)w;
        ah[t] = hi;
        al[t] = split ? (_Float16)(w - (float)hi) : (_Float16)0.0f;
        b[t] = (_Float16)data[l * 8 + t];
    }
    f4v d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b, f4v{0, 0, 0, 0}, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b, d, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[l * 4 + r] = d[r];
}

// KIND: 0 = every wave issues NM mfma + NV pk_fma per iteration; 1 = wave 0 of each SIMD's group issues the MFMAs of all, the
// others the VALU of all (same totals per workgroup); NM or NV may be 0
template <int NM, int NV, int MUL, int PLAIN>
__global__ __launch_bounds__(256) void k_mix(float *out, int iters, float s, int split_roles)
{
    const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(l * 0.001f + i); b[i] = (_Float16)(i * 0.5f - l * 0.002f); }
    f4v acc[8];
    for (int i = 0; i < 8; i++) acc[i] = f4v{0, 0, 0, 0};
    f2 p[16];
    for (int i = 0; i < 16; i++) p[i] = f2{(float)l, (float)i};
    const f2 s2 = f2{s, s * 0.5f};
    // split_roles: 4 waves of a workgroup sit on the 4 SIMDs of a CU (one each); with 2 workgroups per CU, workgroup parity picks
    // the role so that each SIMD hosts one MFMA wave and one VALU wave
    // which two workgroups share a CU is the dispatcher's business: role bit = b & 1, (b >> 8) & 1 or (b >> 3) & 1 (split_roles 1, 2, 3)
    const int role = split_roles == 1 ? (blockIdx.x & 1) : split_roles == 2 ? ((blockIdx.x >> 8) & 1) : ((blockIdx.x >> 3) & 1);
    const bool do_m = !split_roles || role == 0, do_v = !split_roles || role == 1;
    for (int it = 0; it < iters; it++) {
        if (do_m) {
#pragma unroll
            for (int m = 0; m < NM * MUL; m++) { acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 7], 0, 0, 0); }
        }
        if (do_v) {
#pragma unroll
            for (int j = 0; j < NV * MUL; j++) {
                if (PLAIN) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(p[j & 15].x) : "v"(p[(j + 1) & 15].y), "v"(p[(j + 2) & 15].y)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(p[j & 15].y) : "v"(p[(j + 1) & 15].x), "v"(p[(j + 2) & 15].x)); }
                else p[j & 15] = __builtin_elementwise_fma(p[j & 15], s2, s2);
            }
        }
    }
    float r = 0;
    for (int m = 0; m < 8; m++) for (int i = 0; i < 4; i++) r += acc[m][i];
    for (int i = 0; i < 16; i++) r += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + wave;
}

template <int NM, int NV, int PLAIN = 0>
static double run_mix(const char *name, float *d_out, int blocks, int split)
{
    const int iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (split) hipLaunchKernelGGL((k_mix<NM, NV, 2, PLAIN>), dim3(blocks), dim3(256), 0, 0, d_out, 10, 0.999f, split);
    else hipLaunchKernelGGL((k_mix<NM, NV, 1, PLAIN>), dim3(blocks), dim3(256), 0, 0, d_out, 10, 0.999f, split);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    if (split) hipLaunchKernelGGL((k_mix<NM, NV, 2, PLAIN>), dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, split);
    else hipLaunchKernelGGL((k_mix<NM, NV, 1, PLAIN>), dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, split);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wps = blocks / 256.0;                         // waves per SIMD
    const double ns = ms * 1e6 / iters / wps;                  // SIMD time per (one wave's share of an) iteration
    printf("  %-44s blocks=%4d  %8.3f ms  %7.2f ns per wave-iteration per SIMD\n", name, blocks, ms, ns);
    return ns;
}

int main()
{
    srand(7);
    float *d_data, *d_w, *d_out;
    CK(hipMalloc(&d_data, 64 * 8 * 4)); CK(hipMalloc(&d_w, 32 * 4)); CK(hipMalloc(&d_out, 4096 * 256 * 4));
    // ---- 1. own-column layout ----
    {
        std::vector<float> data(64 * 8), w(32), out(256);
        for (auto &x : data) x = (float)(rand() % 1024) / 1024.0f;                       // exact fp16
        for (int i = 0; i < 32; i++) w[i] = (float)((rand() % 4001) - 2000) / 64.0f;     // exact fp16, asymmetric
        CK(hipMemcpy(d_data, data.data(), data.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_owncol, dim3(1), dim3(64), 0, 0, d_data, d_w, d_out, 0);
        CK(hipMemcpy(out.data(), d_out, 256 * 4, hipMemcpyDeviceToHost));
        double me = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
            double ref = 0;
            for (int t = 0; t < 8; t++) ref += (double)w[r * 8 + t] * data[l * 8 + t];
            me = fmax(me, fabs(ref - out[l * 4 + r]));
        }
        printf("1. own-column layout: max |D - per-lane reference| = %g  (%s)\n", me, me < 1e-3 ? "OK" : "WRONG LAYOUT");
    }
    // ---- 2. fp16 subnormal operands ----
    {
        std::vector<float> data(64 * 8, 0.0f), w(32, 0.0f), out(256);
        for (int l = 0; l < 64; l++) data[l * 8] = 1024.0f;
        w[0] = 3.0e-5f;                              // subnormal in fp16 (< 6.1e-5): exact value 3.0040e-05 after rounding to 2^-24 steps
        w[8] = 1.0f; for (int l = 0; l < 64; l++) data[l * 8 + 1] = 0.0f;
        CK(hipMemcpy(d_data, data.data(), data.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_owncol, dim3(1), dim3(64), 0, 0, d_data, d_w, d_out, 0);
        CK(hipMemcpy(out.data(), d_out, 256 * 4, hipMemcpyDeviceToHost));
        printf("2. subnormal fp16 weight 3.0e-5 x 1024: D = %g (expected %g; 0 => subnormals flushed)\n", out[0], (double)(float)(_Float16)3.0e-5f * 1024.0);
        for (int l = 0; l < 64; l++) data[l * 8] = 3.0e-5f;
        w[0] = 1024.0f;
        CK(hipMemcpy(d_data, data.data(), data.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_owncol, dim3(1), dim3(64), 0, 0, d_data, d_w, d_out, 0);
        CK(hipMemcpy(out.data(), d_out, 256 * 4, hipMemcpyDeviceToHost));
        printf("   subnormal fp16 DATA   3.0e-5 x 1024: D = %g\n", out[0]);
    }
    // ---- 3. hi/lo split vs the shader's fp32 chain: X pass on UNORM10 codes, scales as in the kernel ----
    {
        const float wt[5] = {-0.01083153f - 0.08472481f, 0.89105344f, 0.23991315f, -0.01790517f, -0.01750513f};   // Lanczos3 t=.25, Q1-folded
        std::vector<float> data(64 * 8), w(32, 0.0f), out(256);
        std::vector<int> code(64 * 8);
        for (int i = 0; i < 64 * 8; i++) { code[i] = rand() % 1024; data[i] = code[i] / 1024.0f; }
        const float sc = 4096.0f * 1024.0f / 1023.0f;
        const int pos[4][5] = {{0, 2, 3, 4, 5}, {1, 3, 4, 5, 6}, {1, 3, 4, 5, 6}, {2, 4, 5, 6, 7}};
        for (int q = 0; q < 4; q++) for (int t = 0; t < 5; t++) w[q * 8 + pos[q][t]] = wt[t] * sc;
        CK(hipMemcpy(d_data, data.data(), data.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
        for (int split = 0; split < 2; split++) {
            hipLaunchKernelGGL(k_owncol, dim3(1), dim3(64), 0, 0, d_data, d_w, d_out, split);
            CK(hipMemcpy(out.data(), d_out, 256 * 4, hipMemcpyDeviceToHost));
            double me = 0; int flips = 0;
            for (int l = 0; l < 64; l++) for (int q = 0; q < 4; q++) {
                float ref = wt[0] * ((float)code[l * 8 + pos[q][0]] / 1023.0f);
                for (int t = 1; t < 5; t++) ref = fmaf(wt[t], (float)code[l * 8 + pos[q][t]] / 1023.0f, ref);
                const float got = out[l * 4 + q] / 4096.0f;
                me = fmax(me, fabs((double)ref - got));
                if ((_Float16)ref != (_Float16)got) flips++;
            }
            printf("3. X taps on the matrix core, %s: max |diff| vs fp32 chain = %.3g, fp16-rounding flips %d / 256\n",
                   split ? "hi+lo weights" : "hi weights only", me, flips);
        }
    }
    // ---- 4. MFMA beside VALU ----
    printf("4. issue model (ns of SIMD time per wave-iteration; 36 MFMA / 360 pk_fma is the kernel's own ratio x1)\n");
    for (int blocks : {512, 768}) {
        const double m = run_mix<36, 0>("36 mfma alone", d_out, blocks, 0);
        const double v = run_mix<0, 90>("90 pk_fma alone", d_out, blocks, 0);
        const double b = run_mix<36, 90>("36 mfma + 90 pk_fma, same wave", d_out, blocks, 0);
        printf("     -> same wave: %.2f vs sum %.2f vs max %.2f\n", b, m + v, fmax(m, v));
    }
    printf("   plain v_fmac_f32 (VOP2, VGPR operands; 2 per packed FMA) instead of v_pk_fma_f32\n");
    for (int blocks : {512, 768}) {
        const double m = run_mix<36, 0, 1>("36 mfma alone", d_out, blocks, 0);
        const double v = run_mix<0, 90, 1>("180 v_fmac alone", d_out, blocks, 0);
        const double b = run_mix<36, 90, 1>("36 mfma + 180 v_fmac, same wave", d_out, blocks, 0);
        printf("     -> same wave: %.2f vs sum %.2f vs max %.2f\n", b, m + v, fmax(m, v));
    }
    {
        const double m = run_mix<36, 0, 1>("72 mfma per MFMA wave, alone (split, role bit 2)", d_out, 512, 2);
        const double v = run_mix<0, 90, 1>("360 v_fmac per VALU wave, alone (split)", d_out, 512, 2);
        const double b = run_mix<36, 90, 1>("MFMA waves beside plain-VALU waves", d_out, 512, 2);
        printf("     -> separate waves, plain VALU: %.2f vs sum %.2f vs max %.2f\n", b, m + v, fmax(m, v));
    }
    for (int roles = 1; roles <= 3; roles++) {   // different waves: 512 blocks = 2 workgroups per CU, half MFMA-only (2x the MFMAs), half VALU-only (2x)
        printf("   role bit %d\n", roles);
        const double m = run_mix<36, 0>("72 mfma per MFMA wave, alone (split)", d_out, 512, roles);
        const double v = run_mix<0, 90>("180 pk_fma per VALU wave, alone (split)", d_out, 512, roles);
        const double b = run_mix<36, 90>("MFMA waves beside VALU waves", d_out, 512, roles);
        printf("     -> separate waves: %.2f vs sum %.2f vs max %.2f\n", b, m + v, fmax(m, v));
    }
    return 0;
}
