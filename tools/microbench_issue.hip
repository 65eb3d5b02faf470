// Single-wave-per-SIMD issue cadence of fp64 VALU and MFMA on gfx950 (cycles per instruction from
// the shader clock): independent FMAs, dependent chains of various widths, and MFMA/VALU mixes.
// Build: hipcc -O3 --offload-arch=gfx950 microbench_issue.hip -o microbench_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

constexpr int ITERS = 2000;

// NCH independent chains, each FMA depends on the previous of its chain; 32 FMAs per iteration
template <int NCH>
__global__ __launch_bounds__(512) void valu_chains(double *out, long long *cyc, double a, double b)
{
    double x[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) x[i] = threadIdx.x * 1e-3 + i;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / NCH; ++r)
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// per iteration: NM MFMAs (independent accumulators) then NV independent-chain FMAs (8 chains)
template <int NM, int NV>
__global__ __launch_bounds__(512) void mix(double *out, long long *cyc, double a, double b)
{
    double acc[16], x[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    double av = a + threadIdx.x, bv = b - threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i)
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[i & 15]) : "v"(av), "v"(bv));
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(a), "v"(b));
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K>
int run(const char *name, K kernel, int blocks, int threads, int ninstr, double *d_out, long long *d_cyc)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 1.0000001, 1e-9);  // warm
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 1.0000001, 1e-9);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long c;
    CHECK(hipMemcpy(&c, d_cyc, sizeof(c), hipMemcpyDeviceToHost));
    const double ns_per_iter = ms / 5 * 1e6 / ITERS;          // whole kernel: slowest wave
    const int wps = threads / 256;
    printf("%-46s kernel %8.1f ns/iter = %6.2f ns per instr per SIMD | oldest wave %7.1f ticks/iter\n", name, ns_per_iter,
           ns_per_iter / (ninstr * wps), c / (double)ITERS);
    return 0;
}

int main()
{
    double *d_out; long long *d_cyc;
    CHECK(hipMalloc(&d_out, 256 * 1024 * 8)); CHECK(hipMalloc(&d_cyc, 8));
    // 256 blocks x 256 threads = 1 wave per SIMD; x 512 threads = 2 waves per SIMD
    run("1 wave/SIMD  32 FMA, 1 chain (fully dependent)", valu_chains<1>, 256, 256, 32, d_out, d_cyc);
    run("1 wave/SIMD  32 FMA, 2 chains", valu_chains<2>, 256, 256, 32, d_out, d_cyc);
    run("1 wave/SIMD  32 FMA, 4 chains", valu_chains<4>, 256, 256, 32, d_out, d_cyc);
    run("1 wave/SIMD  32 FMA, 8 chains", valu_chains<8>, 256, 256, 32, d_out, d_cyc);
    run("1 wave/SIMD  32 FMA, 16 chains", valu_chains<16>, 256, 256, 32, d_out, d_cyc);
    run("2 waves/SIMD 32 FMA, 8 chains", valu_chains<8>, 256, 512, 32, d_out, d_cyc);
    run("1 wave/SIMD  16 MFMA", mix<16, 0>, 256, 256, 16, d_out, d_cyc);
    run("1 wave/SIMD  16 MFMA + 16 FMA", mix<16, 16>, 256, 256, 32, d_out, d_cyc);
    run("1 wave/SIMD  16 MFMA + 32 FMA", mix<16, 32>, 256, 256, 48, d_out, d_cyc);
    run("2 waves/SIMD 16 MFMA + 16 FMA", mix<16, 16>, 256, 512, 32, d_out, d_cyc);
    run("2 waves/SIMD 16 MFMA", mix<16, 0>, 256, 512, 16, d_out, d_cyc);
    run("2 waves/SIMD 16 MFMA + 32 FMA", mix<16, 32>, 256, 512, 48, d_out, d_cyc);
    run("2 waves/SIMD 2 MFMA + 2 FMA", mix<2, 2>, 256, 512, 4, d_out, d_cyc);
    run("1 wave/SIMD  2 MFMA + 2 FMA", mix<2, 2>, 256, 256, 4, d_out, d_cyc);
    run("1 wave/SIMD  4 MFMA + 4 FMA", mix<4, 4>, 256, 256, 8, d_out, d_cyc);
    return 0;
}
