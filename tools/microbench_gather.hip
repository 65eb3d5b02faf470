// How fast can a CU consume 16-byte loads whose lanes sit in different cache lines?  (the access pattern of the
// lane-per-cell kernels: predict_vis with DDEs, calibration consumers)
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_gather.hip -o tools/microbench_gather && tools/microbench_gather
// stream: N cells of RB bytes (HBM, read once);  gains: a small table (L2 resident) of RB-byte records gathered
// per cell by a pseudo-random antenna index.  Variants:
//   cell     lane = cell, RB/16 dwordx4 loads per record at lane stride RB (each instruction touches 64 lines)
//   coop     G = RB/16 lanes per record, each instruction touches 64/G full lines; no transposition (sums only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int RB, int NG>
__global__ __launch_bounds__(256) void k_cell(const double2 *__restrict__ stream, const double2 *__restrict__ gains,
                                              const int *__restrict__ ant, long n, double2 *__restrict__ out)
{
    const long cell = (long)blockIdx.x * 256 + threadIdx.x;
    if (cell >= n) return;
    constexpr int G = RB / 16;
    double2 acc = make_double2(0, 0);
    const double2 *p = stream + cell * G;
#pragma unroll
    for (int h = 0; h < G; ++h) { double2 v = p[h]; acc.x += v.x; acc.y += v.y; }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const double2 *q = gains + (long)ant[cell * NG + g] * G;
#pragma unroll
        for (int h = 0; h < G; ++h) { double2 v = q[h]; acc.x += v.x; acc.y -= v.y; }
    }
    out[cell] = acc;
}

template <int RB, int NG>
__global__ __launch_bounds__(256) void k_coop(const double2 *__restrict__ stream, const double2 *__restrict__ gains,
                                              const int *__restrict__ ant, long n, double2 *__restrict__ out)
{
    constexpr int G = RB / 16, CPI = 64 / G;   // lanes per record, cells per instruction
    const int lane = threadIdx.x & 63;
    const long wave_cell0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (wave_cell0 >= n) return;
    double2 acc = make_double2(0, 0);
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const long cell = wave_cell0 + k * CPI + lane / G;
        double2 v = stream[cell * G + (lane % G)];
        acc.x += v.x; acc.y += v.y;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double2 u = gains[(long)ant[cell * NG + g] * G + (lane % G)];
            acc.x += u.x; acc.y -= u.y;
        }
    }
    out[wave_cell0 + lane] = acc;
}

int main()
{
    const long n = 1L << 24;   // cells
    constexpr int RB = 128, NG = 2;
    const int nant = 64 * 64;  // table of 4096 records (512 KB)
    double2 *stream, *gains, *out; int *ant;
    CK(hipMalloc(&stream, n * RB)); CK(hipMalloc(&gains, (size_t)nant * RB)); CK(hipMalloc(&out, n * 16));
    CK(hipMalloc(&ant, n * NG * 4));
    CK(hipMemset(stream, 0, n * RB)); CK(hipMemset(gains, 0, (size_t)nant * RB));
    int *h = (int *)malloc(n * NG * 4);
    for (long i = 0; i < n; ++i) {   // like (row, chan) cells: the antenna of a cell changes every 64 cells
        h[i * NG] = (int)(((i >> 6) * 7 % 64) * 64 + (i & 63));
        h[i * NG + 1] = (int)(((i >> 6) * 13 % 64) * 64 + (i & 63));
    }
    CK(hipMemcpy(ant, h, n * NG * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch, double bytes) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%-44s %7.3f ms  %6.2f TB/s of HBM bytes\n", name, ms, bytes / ms / 1e9);
    };
    const double hbm = (double)n * (RB + 16 + NG * 4);
    run("cell: stream only", [&] { hipLaunchKernelGGL((k_cell<RB, 0>), dim3(n / 256), dim3(256), 0, 0, stream, gains, ant, n, out); }, (double)n * (RB + 16));
    run("coop: stream only", [&] { hipLaunchKernelGGL((k_coop<RB, 0>), dim3(n / 256), dim3(256), 0, 0, stream, gains, ant, n, out); }, (double)n * (RB + 16));
    run("cell: stream + 2 gathered records", [&] { hipLaunchKernelGGL((k_cell<RB, NG>), dim3(n / 256), dim3(256), 0, 0, stream, gains, ant, n, out); }, hbm);
    run("coop: stream + 2 gathered records", [&] { hipLaunchKernelGGL((k_coop<RB, NG>), dim3(n / 256), dim3(256), 0, 0, stream, gains, ant, n, out); }, hbm);
    return 0;
}
