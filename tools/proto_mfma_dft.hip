// Prototype / microbenchmark of the MFMA-accumulator direct transform for gfx950:
// a wave owns 16 rows, each MFMA step contracts 4 sources against the 4 correlations
// (v_mfma_f64_4x4x4_4b: blocks = 4 row groups), accumulators for CT channels live in the
// MFMA C/D registers (AGPR-capable), the per-lane VALU work is the phasor recurrence only.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../codex_africanus_amd/csrc proto_mfma_dft.hip -o proto_mfma_dft
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "af_sincos.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

void af_set_error(const char *, ...) {}
int af_hip_fail(hipError_t, const char *, const char *, int) { return 1; }

constexpr int ANCHOR = 16;

template <int CT, int VAR>
__global__ __launch_bounds__(256) void dft_mfma_kernel(const double *__restrict__ uvw, const double *__restrict__ recB,
                                                      double F0, double FD, double *__restrict__ out, int64_t nrow,
                                                      int nit, int64_t nchan, long long *dbg)
{
    // stage = [16 doubles (l,m,n,0) x 4 sources][CT/2 channel pairs][16 (k, corr)][2 channels]
    constexpr int STAGE = (CT + 1) * 16;        // doubles
    constexpr int UNITS = STAGE / 2;            // 16-byte units
    __shared__ double smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = lane >> 4;
    int64_t row = (int64_t)blockIdx.x * 64 + wave * 16 + (lane & 15);
    if (row >= nrow) row = nrow - 1;
    const double u = uvw[3 * row], v = uvw[3 * row + 1], w = uvw[3 * row + 2];
    const int boff = (k * 4 + (lane & 3)) * 2;

    double are[CT], aim[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) are[j] = aim[j] = 0.0;

    auto stage_load = [&](int it, int buf) {
        const double *src = recB + (int64_t)it * STAGE;
#pragma unroll
        for (int e0 = 0; e0 < UNITS; e0 += 256) {
            const int ebase = e0 + wave * 64;          // wave-uniform
            if (ebase + lane < UNITS) {
                const double *g = src + (ebase + lane) * 2;
                const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(smem + buf * STAGE + ebase * 2));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
            }
        }
    };
    stage_load(0, 0);
    asm volatile("" :: "v"(u), "v"(v), "v"(w));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < nit; ++it) {
        const int cur = it & 1;
        if (it + 1 < nit) stage_load(it + 1, cur ^ 1);
        const double *S = smem + cur * STAGE;
        const double2 lm_ = *reinterpret_cast<const double2 *>(S + 4 * k);
        const double n = S[4 * k + 2];
        const double q = fma(n, w, fma(lm_.y, v, lm_.x * u));
        double dr, di, y0r, y0i;
        if (VAR == 1) { dr = q * 1e-9; di = q * 2e-9; y0r = q * 3e-9; y0i = q * 4e-9; }
        else {
            sincos_quarter_turns<7>(q * FD, dr, di);
            sincos_quarter_turns<7>(q * F0, y0r, y0i);
        }
        double ar = dr, ai = di;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double nr = fma(ar, ar, -(ai * ai)), ni = (ar + ar) * ai;
            ar = nr; ai = ni;
        }
        const double kk = dr + dr;
        const double2 *B = reinterpret_cast<const double2 *>(S + 16 + boff);   // + (j/2)*16 double2
        constexpr int GP = 4;                      // channel pairs per register group
        constexpr int NGRP = CT / 2 / GP;
        double2 bg[2][GP];
#pragma unroll
        for (int p = 0; p < GP; ++p) bg[0][p] = B[p * 16];
        if (VAR >= 2) {
            // phasors of a group of 8 channels are produced one group ahead of the MFMAs that use them;
            // four independent recurrence chains (re/im x even/odd channels, step 2 delta)
            const double k2 = fma(kk, kk, -2.0);
            double yr[2][8], yi[2][8];
            double anr = y0r, ani = y0i;
            auto start_segment = [&](double (&Yr)[8], double (&Yi)[8]) {
                Yr[0] = anr; Yi[0] = ani;
                Yr[1] = fma(anr, dr, -(ani * di)); Yi[1] = fma(anr, di, ani * dr);
                Yr[2] = fma(kk, Yr[1], -Yr[0]); Yi[2] = fma(kk, Yi[1], -Yi[0]);
                Yr[3] = fma(kk, Yr[2], -Yr[1]); Yi[3] = fma(kk, Yi[2], -Yi[1]);
#pragma unroll
                for (int t = 4; t < 8; ++t) { Yr[t] = fma(k2, Yr[t - 2], -Yr[t - 4]); Yi[t] = fma(k2, Yi[t - 2], -Yi[t - 4]); }
            };
            auto continue_segment = [&](double (&Yr)[8], double (&Yi)[8], const double (&Pr)[8], const double (&Pi)[8]) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const double ar2 = t >= 2 ? Yr[t - 2] : Pr[t + 6], ar4 = t >= 4 ? Yr[t - 4] : Pr[t + 4];
                    const double ai2 = t >= 2 ? Yi[t - 2] : Pi[t + 6], ai4 = t >= 4 ? Yi[t - 4] : Pi[t + 4];
                    Yr[t] = fma(k2, ar2, -ar4); Yi[t] = fma(k2, ai2, -ai4);
                }
            };
            start_segment(yr[0], yi[0]);
#pragma unroll
            for (int g = 0; g < NGRP; ++g) {
                if (g + 1 < NGRP) {
#pragma unroll
                    for (int p = 0; p < GP; ++p) bg[(g + 1) & 1][p] = B[((g + 1) * GP + p) * 16];
                    if (VAR == 3) {
                    } else if (((g + 1) * 8) % ANCHOR == 0) {
                        const double tr = fma(anr, ar, -(ani * ai)), ti = fma(anr, ai, ani * ar);
                        anr = tr; ani = ti;
                        start_segment(yr[(g + 1) & 1], yi[(g + 1) & 1]);
                    } else {
                        continue_segment(yr[(g + 1) & 1], yi[(g + 1) & 1], yr[g & 1], yi[g & 1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = g * 8 + jj;
                    const double b = (jj & 1) ? bg[g & 1][jj >> 1].y : bg[g & 1][jj >> 1].x;
                    if (VAR == 4) { are[j] += yr[g & 1][jj] * b; aim[j] += yi[g & 1][jj] * b; continue; }
                    are[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(yr[VAR == 3 ? 0 : (g & 1)][jj], b, are[j], 0, 0, 0);
                    aim[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(yi[VAR == 3 ? 0 : (g & 1)][jj], b, aim[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        double y1r = fma(y0r, dr, -(y0i * di)), y1i = fma(y0r, di, y0i * dr);
        double anr = y0r, ani = y0i;
#pragma unroll
        for (int g = 0; g < CT / 2 / GP; ++g) {
            if (g + 1 < CT / 2 / GP) {
#pragma unroll
                for (int p = 0; p < GP; ++p) bg[(g + 1) & 1][p] = B[((g + 1) * GP + p) * 16];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < 2 * GP; ++jj) {
                const int j = g * 2 * GP + jj;
                double yr, yi;
                if (j % ANCHOR == 0) {
                    if (j > 0) {
                        const double tr = fma(anr, ar, -(ani * ai)), ti = fma(anr, ai, ani * ar);
                        anr = tr; ani = ti;
                        y0r = anr; y0i = ani;
                        y1r = fma(y0r, dr, -(y0i * di)); y1i = fma(y0r, di, y0i * dr);
                    }
                    yr = y0r; yi = y0i;
                } else if (j % ANCHOR == 1) {
                    yr = y1r; yi = y1i;
                } else {
                    yr = fma(kk, y1r, -y0r);
                    yi = fma(kk, y1i, -y0i);
                    y0r = y1r; y0i = y1i; y1r = yr; y1i = yi;
                }
                const double b = (jj & 1) ? bg[g & 1][jj >> 1].y : bg[g & 1][jj >> 1].x;
                are[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(yr, b, are[j], 0, 0, 0);
                aim[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(yi, b, aim[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (tid == 0 && blockIdx.x < 512) { dbg[2 * blockIdx.x] = t1 - t0; dbg[2 * blockIdx.x + 1] = r1 - r0; }
    const int orow = 4 * ((lane >> 2) & 3) + (lane >> 4), ocorr = lane & 3;
    const int64_t r = (int64_t)blockIdx.x * 64 + wave * 16 + orow;
    if (r < nrow) {
#pragma unroll
        for (int j = 0; j < CT; ++j)
            reinterpret_cast<double2 *>(out)[(r * nchan + j) * 4 + ocorr] = make_double2(are[j], aim[j]);
    }
}

int main(int argc, char **argv)
{
    const int64_t nrow = argc > 1 ? atoll(argv[1]) : 262144;
    const int nsrc = 1000, nit = nsrc / 4;
    constexpr int CT = 64;
    const int64_t nchan = CT;
    std::vector<double> uvw(nrow * 3), lm(nsrc * 2), img((size_t)nsrc * nchan * 4), rec((size_t)nit * (CT + 1) * 16, 0.0);
    srand(1);
    auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
    for (auto &x : uvw) x = rnd() * 8000.0;
    for (auto &x : lm) x = rnd() * 0.1;
    for (auto &x : img) x = rnd() * 2.0;
    const double c = 2.99792458e8, f0 = 0.856e9, df = 0.856e9 / 63.0;
    for (int it = 0; it < nit; ++it)
        for (int k = 0; k < 4; ++k) {
            const int s = 4 * it + k;
            const double l = lm[2 * s], m = lm[2 * s + 1];
            double *R = rec.data() + (size_t)it * (CT + 1) * 16;
            R[4 * k] = l; R[4 * k + 1] = m; R[4 * k + 2] = sqrt(1 - l * l - m * m) - 1;
            for (int j = 0; j < CT; ++j)
                for (int n = 0; n < 4; ++n) R[16 + (j / 2) * 32 + (k * 4 + n) * 2 + (j & 1)] = img[((size_t)s * nchan + j) * 4 + n];
        }
    double *d_uvw, *d_rec, *d_out;
    CHECK(hipMalloc(&d_uvw, uvw.size() * 8)); CHECK(hipMalloc(&d_rec, rec.size() * 8));
    CHECK(hipMalloc(&d_out, (size_t)nrow * nchan * 4 * 16));
    CHECK(hipMemcpy(d_uvw, uvw.data(), uvw.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rec, rec.data(), rec.size() * 8, hipMemcpyHostToDevice));
    const double F0 = -4.0 * f0 / c, FD = -4.0 * df / c;
    long long *d_dbg, h_dbg[1024];
    CHECK(hipMalloc(&d_dbg, sizeof(h_dbg)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    dim3 grid((unsigned)((nrow + 63) / 64));
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        if (rep == 2 || rep == 3) hipLaunchKernelGGL((dft_mfma_kernel<CT, 3>), grid, dim3(256), 0, 0, d_uvw, d_rec, F0, FD, d_out, nrow, nit, nchan, d_dbg);
        else if (rep >= 4) hipLaunchKernelGGL((dft_mfma_kernel<CT, 2>), grid, dim3(256), 0, 0, d_uvw, d_rec, F0, FD, d_out, nrow, nit, nchan, d_dbg);
        else
        hipLaunchKernelGGL((dft_mfma_kernel<CT, 0>), grid, dim3(256), 0, 0, d_uvw, d_rec, F0, FD, d_out, nrow, nit, nchan, d_dbg);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("rep %d CT=%d rows=%lld: %.3f ms  -> %.1f Mvis/s, scaled to 1e6 rows: %.2f ms\n", rep, CT, (long long)nrow, ms,
               nrow * nchan / ms / 1e3, ms * 1e6 / nrow);
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h_dbg, d_dbg, sizeof(h_dbg), hipMemcpyDeviceToHost));
    for (int b = 0; b < 512; b += 173)
        printf("block %d: main loop %lld cycles (%.0f per iteration), %lld wall ticks (100 MHz) -> %.3f GHz\n", b, h_dbg[2 * b],
               h_dbg[2 * b] / (double)nit, h_dbg[2 * b + 1], h_dbg[2 * b] / (h_dbg[2 * b + 1] * 10.0));
    // check 48 rows against a direct evaluation
    std::vector<double> out(48 * nchan * 8);
    double maxerr = 0;
    for (int t = 0; t < 48; ++t) {
        const int64_t r = (nrow - 1) * t / 47;
        CHECK(hipMemcpy(out.data(), d_out + r * nchan * 8, nchan * 8 * 8, hipMemcpyDeviceToHost));
        for (int j = 0; j < nchan; ++j)
            for (int n = 0; n < 4; ++n) {
                double re = 0, im = 0;
                for (int s = 0; s < nsrc; ++s) {
                    const double l = lm[2 * s], m = lm[2 * s + 1], nn = sqrt(1 - l * l - m * m) - 1;
                    const double p = -2 * M_PI / c * (l * uvw[3 * r] + m * uvw[3 * r + 1] + nn * uvw[3 * r + 2]) * (f0 + j * df);
                    re += cos(p) * img[((size_t)s * nchan + j) * 4 + n];
                    im += sin(p) * img[((size_t)s * nchan + j) * 4 + n];
                }
                maxerr = fmax(maxerr, fmax(fabs(re - out[(j * 4 + n) * 2]), fabs(im - out[(j * 4 + n) * 2 + 1])));
            }
    }
    printf("max abs err on 48 rows: %.3e\n", maxerr);
    return 0;
}
