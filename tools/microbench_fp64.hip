// fp64 issue-rate microbenchmark for gfx950: v_fma_f64 (VALU) against the f64 MFMA forms,
// alone and co-issued.  Prints achieved TFLOP/s so DESIGN.md's roofline denominators are
// measured, not assumed.  Build: hipcc -O3 --offload-arch=gfx950 microbench_fp64.hip -o microbench_fp64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;

__global__ __launch_bounds__(256) void fma_kernel(double *out, double a, double b)
{
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void mfma16_kernel(double *out, double a, double b)
{
    double4_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = {0, 0, 0, 0};
    double av = a + threadIdx.x, bv = b - threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void mfma4_kernel(double *out, double a, double b)
{
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0;
    double av = a + threadIdx.x, bv = b - threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same wave alternates 1 MFMA 4x4x4 with NV v_fma_f64: do the pipes overlap within a wave?
template <int NV>
__global__ __launch_bounds__(256) void mixed_kernel(double *out, double a, double b)
{
    double acc[8], x[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
    double av = a + threadIdx.x, bv = b - threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) x[(i * NV + j) & 15] = fma(x[(i * NV + j) & 15], a, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// v_fmac_f64 with a 64-bit DPP row_newbcast source: acc += lane_k_of_row(g) * y
__global__ __launch_bounds__(256) void dpp_kernel(double *out, double a, double b)
{
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
    double g = a + (threadIdx.x & 15) * 1e-9, y = b;
    for (int it = 0; it < ITERS; ++it) {
#define DPPSTEP(i, k) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(g), "v"(y));
        DPPSTEP(0, 0) DPPSTEP(1, 1) DPPSTEP(2, 2) DPPSTEP(3, 3) DPPSTEP(4, 4) DPPSTEP(5, 5) DPPSTEP(6, 6) DPPSTEP(7, 7)
        DPPSTEP(8, 8) DPPSTEP(9, 9) DPPSTEP(10, 10) DPPSTEP(11, 11) DPPSTEP(12, 12) DPPSTEP(13, 13) DPPSTEP(14, 14) DPPSTEP(15, 15)
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// semantics check: out[lane] = 0 + g[lane k of the lane's row of 16] * 1.0
__global__ void dpp_check_kernel(double *out)
{
    double g = 100.0 + threadIdx.x, y = 1.0, acc = 0.0;
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(g), "v"(y));
    out[threadIdx.x] = acc;
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s arch %s CUs %d clock %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * 8192));
    {
        double h[64];
        hipLaunchKernelGGL(dpp_check_kernel, dim3(1), dim3(64), 0, 0, out);
        CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        int ok = 1;
        for (int i = 0; i < 64; ++i) ok &= (h[i] == 100.0 + (i / 16) * 16 + 5);
        printf("row_newbcast:5 semantics %s (lane0 %.0f lane17 %.0f lane63 %.0f)\n", ok ? "OK" : "MISMATCH", h[0], h[17], h[63]);
    }
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
        int blocks = p.multiProcessorCount * wps;  // 256 threads = 4 waves = 1 wave per SIMD
        double lanes = (double)blocks * 256;
        double ms;
        ms = time_ms([&] { hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9); }, 5);
        printf("waves/SIMD %d  v_fma_f64         : %8.3f ms  %7.2f TFLOP/s\n", wps, ms, lanes * ITERS * 16 * 2 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(dpp_kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9); }, 5);
        printf("waves/SIMD %d  v_fmac_f64_dpp    : %8.3f ms  %7.2f TFLOP/s\n", wps, ms, lanes * ITERS * 16 * 2 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(mfma16_kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9); }, 5);
        printf("waves/SIMD %d  mfma_f64_16x16x4  : %8.3f ms  %7.2f TFLOP/s\n", wps, ms, (double)blocks * 4 * ITERS * 4 * (16 * 16 * 4 * 2) / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(mfma4_kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9); }, 5);
        printf("waves/SIMD %d  mfma_f64_4x4x4_4b : %8.3f ms  %7.2f TFLOP/s\n", wps, ms, (double)blocks * 4 * ITERS * 8 * (4 * 4 * 4 * 4 * 2) / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((mixed_kernel<2>), dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9); }, 5);
        printf("waves/SIMD %d  mixed 1 mfma4:2 fma: %8.3f ms  mfma %7.2f + valu %7.2f TFLOP/s\n", wps, ms,
               (double)blocks * 4 * ITERS * 8 * 512 / ms / 1e9, lanes * ITERS * 16 * 2 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((mixed_kernel<4>), dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9); }, 5);
        printf("waves/SIMD %d  mixed 1 mfma4:4 fma: %8.3f ms  mfma %7.2f + valu %7.2f TFLOP/s\n", wps, ms,
               (double)blocks * 4 * ITERS * 8 * 512 / ms / 1e9, lanes * ITERS * 32 * 2 / ms / 1e9);
    }
    CHECK(hipFree(out));
    return 0;
}
