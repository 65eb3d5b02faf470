// build + run (gfx950 box): hipcc --offload-arch=gfx950 -O3 -o probe_mfma_f32 probe_mfma_f32.hip && ./probe_mfma_f32
// v_mfma_f32_4x4x1_16b_f32: lane layout of A / B / D and the CBSZ / ABID broadcast of A, and whether fp64 VALU work
// overlaps with a stream of these MFMAs (separate pipes?).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int ABID> __global__ void probe(float *out)
{
    const int l = threadIdx.x;
    const float a = 100.0f + l, b = (float)(l + 1);
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
__global__ void probe_plain(float *out)
{
    const int l = threadIdx.x;
    const float a = 100.0f + l, b = (float)(l + 1);
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
// timing: NM MFMAs on 16 accumulators per iteration, optionally NV fp64 FMAs (+ 2 cvt per pair) interleaved
template <int NV> __global__ void timing(float *out, int iters, long long *cycles)
{
    const int l = threadIdx.x;
    float a = 1.0f + l * 1e-3f, b = 0.5f;
    v4f acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = v4f{0, 0, 0, 0};
    double y0 = 1.0 + l * 1e-9, y1 = 0.999, kk = 1.9999;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define STEP(k) do { if (NV) { _Pragma("unroll") for (int v = 0; v < NV; ++v) { const double y2 = fma(kk, y1, -y0); y0 = y1; y1 = y2; } b = (float)y1; } \
        acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 4, k, 0); } while (0)
        STEP(0); STEP(1); STEP(2); STEP(3); STEP(4); STEP(5); STEP(6); STEP(7);
        STEP(8); STEP(9); STEP(10); STEP(11); STEP(12); STEP(13); STEP(14); STEP(15);
#undef STEP
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int k = 0; k < 16; ++k) s += acc[k][0] + acc[k][3];
    out[blockIdx.x * blockDim.x + l] = s + (float)y1;
    if (l == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}
// independent streams in ONE wave: 16 MFMAs and NV fp64 FMAs (8 independent chains) + NC f64->f32 conversions per iteration
template <int NV, int NC, bool MF> __global__ void overlap(float *out, int iters, long long *cycles)
{
    const int l = threadIdx.x;
    float a = 1.0f + l * 1e-3f, b = 0.5f;
    v4f acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = v4f{0, 0, 0, 0};
    double y[8];
    for (int k = 0; k < 8; ++k) y[k] = 1.0 + l * 1e-9 + k;
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double kk = 0.9999999;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define STEP(k) do { if (MF) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 4, k, 0); \
        if ((k) < NV) y[(k) & 7] = fma(kk, y[(k) & 7], 1e-9); if ((k) < NC) f[(k) & 7] += (float)y[(k) & 7]; } while (0)
        STEP(0); STEP(1); STEP(2); STEP(3); STEP(4); STEP(5); STEP(6); STEP(7);
        STEP(8); STEP(9); STEP(10); STEP(11); STEP(12); STEP(13); STEP(14); STEP(15);
#undef STEP
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int k = 0; k < 16; ++k) s += acc[k][0] + acc[k][3];
    for (int k = 0; k < 8; ++k) s += (float)y[k] + f[k];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}
int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    std::vector<float> h(256);
    auto show = [&](const char *name) {
        hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
        printf("%s\n  lane 0: %g %g %g %g | lane 1: %g %g %g %g | lane 5: %g %g %g %g | lane 63: %g %g %g %g\n", name, h[0], h[1], h[2],
               h[3], h[4], h[5], h[6], h[7], h[20], h[21], h[22], h[23], h[252], h[253], h[254], h[255]);
    };
    hipLaunchKernelGGL(probe_plain, dim3(1), dim3(64), 0, 0, out); show("cbsz 0 (no broadcast): expect lane l reg i = (100 + 4 (l/4) + i) * (l + 1)");
    hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, out); show("cbsz 4 abid 0: expect lane l reg i = (100 + i) * (l + 1)");
    hipLaunchKernelGGL(probe<5>, dim3(1), dim3(64), 0, 0, out); show("cbsz 4 abid 5: expect lane l reg i = (120 + i) * (l + 1)");
    const int iters = 20000;
    long long c;
    // one wave per SIMD (256 threads), then two (512)
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(timing<0>, dim3(256), dim3(threads), 0, 0, out, iters, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%d threads/block: MFMA only: %.2f cycles per MFMA per wave\n", threads, (double)c / iters / 16);
        hipLaunchKernelGGL(timing<1>, dim3(256), dim3(threads), 0, 0, out, iters, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%d threads/block: MFMA + 1 fp64 FMA + cvt each: %.2f cycles per MFMA per wave\n", threads, (double)c / iters / 16);
        hipLaunchKernelGGL(timing<2>, dim3(256), dim3(threads), 0, 0, out, iters, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%d threads/block: MFMA + 2 fp64 FMA + cvt each: %.2f cycles per MFMA per wave\n", threads, (double)c / iters / 16);
        hipLaunchKernelGGL(timing<4>, dim3(256), dim3(threads), 0, 0, out, iters, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%d threads/block: MFMA + 4 fp64 FMA + cvt each: %.2f cycles per MFMA per wave\n", threads, (double)c / iters / 16);
    }
    auto run = [&](auto kern, const char *name, int threads) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%d threads: %s: %.1f cycles per iteration\n", threads, name, (double)c / iters);
    };
    for (int threads : {256, 512, 768}) {
        run(overlap<0, 0, true>, "16 MFMA", threads);
        run(overlap<16, 0, false>, "16 fp64 FMA", threads);
        run(overlap<16, 0, true>, "16 MFMA + 16 fp64 FMA", threads);
        run(overlap<0, 16, false>, "16 cvt+add", threads);
        run(overlap<16, 16, false>, "16 fp64 FMA + 16 cvt+add", threads);
        run(overlap<16, 16, true>, "16 MFMA + 16 fp64 FMA + 16 cvt+add", threads);
    }
    return 0;
}
