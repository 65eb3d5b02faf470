// build + run (gfx950 box): hipcc --offload-arch=gfx950 -O3 -o probe_mfma_f32_16x16x4 probe_mfma_f32_16x16x4.hip && ./probe_mfma_f32_16x16x4
// v_mfma_f32_16x16x4_f32: (1) the lane layout of A / B / D, decoded from products of small integers; (2) its issue rate
// (cycles per instruction, one wave per SIMD and two); (3) whether fp32 / fp64 VALU work of ANOTHER wave on the same SIMD
// runs beside a stream of these MFMAs (the fp64 forms share one pipe with fp64 VALU: tools/microbench_fp64).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void layout(float *out, int ksel)
{
    const int l = threadIdx.x;
    // assumed operand layout: lane l holds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]
    const int k = l >> 4;
    const float a = k == ksel ? (float)((l & 15) + 1) : 0.0f, b = k == ksel ? 100.0f * (float)((l & 15) + 1) : 0.0f;
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
template <int WHO>   // waves 0-3: MFMA stream; waves 4-7 (same SIMDs): WHO = 0 idle, 1 fp32 FMA chains, 2 fp64 FMA chains
__global__ void timing(float *out, int iters, long long *cycles)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w < 4) {
        float a = 1.0f + l * 1e-3f, b = 0.5f;
        v4f acc[8];
        for (int k = 0; k < 8; ++k) acc[k] = v4f{0, 0, 0, 0};
        const long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
        const long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][3];
        out[threadIdx.x] = s;
        if (l == 0 && blockIdx.x == 0) cycles[w] = t1 - t0;
    } else if (WHO == 1) {
        float y[8];
        for (int k = 0; k < 8; ++k) y[k] = 1.0f + l * 1e-6f + k;
        const long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 32; ++k) y[k & 7] = fmaf(0.999999f, y[k & 7], 1e-7f);
        const long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int k = 0; k < 8; ++k) s += y[k];
        out[threadIdx.x] = s;
        if (l == 0 && blockIdx.x == 0) cycles[w] = t1 - t0;
    } else if (WHO == 2) {
        double y[8];
        for (int k = 0; k < 8; ++k) y[k] = 1.0 + l * 1e-9 + k;
        const long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 32; ++k) y[k & 7] = fma(0.9999999, y[k & 7], 1e-9);
        const long long t1 = __builtin_amdgcn_s_memtime();
        double s = 0;
        for (int k = 0; k < 8; ++k) s += y[k];
        out[threadIdx.x] = (float)s;
        if (l == 0 && blockIdx.x == 0) cycles[w] = t1 - t0;
    }
}
int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    std::vector<float> h(256);
    for (int ksel = 0; ksel < 4; ksel += 3) {
        hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out, ksel);
        hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
        printf("k = %d contributes: D element (i, j) held by (lane, reg):\n", ksel);
        bool std_layout = true, alt_layout = true;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int v = (int)(h[l * 4 + r] + 0.5f), j = v / 100 > 0 ? 0 : 0;
                (void)j;
                // v = (i + 1) * 100 * (j + 1): recover by trial
                int fi = -1, fj = -1;
                for (int i = 0; i < 16 && fi < 0; ++i)
                    for (int jj = 0; jj < 16; ++jj)
                        if ((i + 1) * 100 * (jj + 1) == v && jj == (l & 15)) { fi = i; fj = jj; break; }
                if (fi != 4 * (l >> 4) + r) std_layout = false;
                if (fi != 4 * r + (l >> 4)) alt_layout = false;
                if (l == 0 || l == 17 || l == 35 || l == 63) printf("  lane %2d reg %d: value %6d -> (i %d, j %d)\n", l, r, v, fi, fj);
            }
        printf("  D[i = 4 (lane >> 4) + reg][j = lane & 15]: %s;  D[i = 4 reg + (lane >> 4)][j = lane & 15]: %s\n",
               std_layout ? "YES" : "no", alt_layout ? "YES" : "no");
    }
    const int iters = 4000;
    long long c[8];
    const char *names[3] = {"MFMA waves alone (4 waves, one per SIMD)", "beside fp32 FMA chains on the same SIMDs", "beside fp64 FMA chains on the same SIMDs"};
    for (int who = 0; who < 3; ++who) {
        hipMemset(cyc, 0, 64);
        if (who == 0) hipLaunchKernelGGL(timing<0>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
        if (who == 1) hipLaunchKernelGGL(timing<1>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
        if (who == 2) hipLaunchKernelGGL(timing<2>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
        hipDeviceSynchronize();
        hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
        printf("%s: s_memtime ticks per MFMA %.2f (wave 0), per 32 VALU FMAs of wave 4: %.1f\n", names[who], (double)c[0] / (iters * 8.0),
               (double)c[4] / iters);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int who = 0; who < 3; ++who) {
        hipEventRecord(e0);
        if (who == 0) hipLaunchKernelGGL(timing<0>, dim3(2560), dim3(512), 0, 0, out, iters, cyc);
        if (who == 1) hipLaunchKernelGGL(timing<1>, dim3(2560), dim3(512), 0, 0, out, iters, cyc);
        if (who == 2) hipLaunchKernelGGL(timing<2>, dim3(2560), dim3(512), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2560.0 * 4 * iters * 8 * 2048;
        printf("%s: %.3f ms, %.1f TFLOP/s of f32 MFMA (2560 workgroups)\n", names[who], ms, flop / ms / 1e9);
    }
    return 0;
}
