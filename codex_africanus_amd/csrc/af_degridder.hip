// Convolutional degridder and gridder for gfx950 (BASELINE configs[4]).
//
// Replaces africanus/gridding/perleypolyhedron/degridder.py:15-175 with the gather convolution policies
// (policies/convolution_policies.py:188-323), the 16 Stokes -> correlation policies
// (policies/stokes_conversion_policies.py:8-137, passed as per-correlation complex factors) and the optional
// facet phase rotation (policies/phase_transform_policies.py:9-35):
//     vis[r,c,:] = factor[:] * phase(r,c) * sum_{tv,tu} grid[band(c), dv+tv-W/2, du+tu-W/2] K[v tap] K[u tap]
//                  / (sum of the in-bounds tap weights + 1e-8)
// One lane per visibility (row, chan), channel fastest: neighbouring channels of a row land on neighbouring
// grid cells, so a wave's W x W gathers share cache lines; the grid (268 MB at 4096^2) lives in L2 / Infinity
// Cache, the kernel is gather-bound.  The tap weights of a visibility are 2 W values read once per lane.
#include <stdlib.h>

#include <type_traits>

#include "af_common.h"
#include "af_sincos.h"

namespace {

constexpr int MAXW = 15;
constexpr int NBIN = 4096;  // 64 x 64 uv tiles, Morton ordered

// ---- uv-tile binning of the rows --------------------------------------------------------------------
// A row's channels lie on a ray through the uv origin; rows arrive in no useful order (Measurement-Set order is
// time-major: consecutive rows are different baselines).  Processing the rows tile by tile of their mid-band uv
// position keeps the W x W gathers of concurrently running waves inside one L2-sized neighbourhood of the grid.
// Counting sort: histogram -> exclusive scan -> scatter (order within a tile is arbitrary: rows are independent).
__device__ __forceinline__ unsigned morton6(unsigned x, unsigned y)
{
    unsigned k = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) k |= ((x >> b) & 1u) << (2 * b) | ((y >> b) & 1u) << (2 * b + 1);
    return k;
}

__global__ void degrid_bin_kernel_dev(const double *__restrict__ uvw, int64_t nrow,
                                      const double *__restrict__ wavelengths, int64_t nchan, double scale_factor,
                                      int64_t npix, unsigned short *__restrict__ key, int *__restrict__ hist)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    const double scale_over_lambda_mid = scale_factor / wavelengths[nchan / 2];
    const double tile = (double)npix / 64.0;
    double x = (uvw[3 * r] * scale_over_lambda_mid + (double)(npix / 2)) / tile;
    double y = (uvw[3 * r + 1] * scale_over_lambda_mid + (double)(npix / 2)) / tile;
    x = x < 0.0 ? 0.0 : (x > 63.0 ? 63.0 : x);   // NaN -> 63 via the comparisons' false branches is fine: any bin works
    y = y < 0.0 ? 0.0 : (y > 63.0 ? 63.0 : y);
    const unsigned k = morton6((unsigned)x & 63u, (unsigned)y & 63u);
    key[r] = (unsigned short)k;
    atomicAdd(&hist[k], 1);
}

__global__ __launch_bounds__(1024) void degrid_scan_kernel(int *__restrict__ hist)  // in place: counts -> starts
{
    __shared__ int part[1024];
    const int t = threadIdx.x;
    int v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = hist[4 * t + i]; s += v[i]; }
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int add = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int base = part[t] - s;
#pragma unroll
    for (int i = 0; i < 4; ++i) { hist[4 * t + i] = base; base += v[i]; }
}

__global__ void degrid_scatter_kernel(const unsigned short *__restrict__ key, int64_t nrow, int *__restrict__ start,
                                      int *__restrict__ perm)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    perm[atomicAdd(&start[key[r]], 1)] = (int)r;
}

// grid: ceil(nrow*nchan / 256)
template <int WT>  // compile-time tap count (0: runtime W)
__global__ __launch_bounds__(256) void degrid_kernel(const double *__restrict__ uvw, const double2 *__restrict__ grid,
                                                     const double *__restrict__ wavelengths,
                                                     const int64_t *__restrict__ chanmap,
                                                     const double *__restrict__ kernel, int Wrt, int os, int packed,
                                                     int ncorr, const double2 *__restrict__ coef, double scale_factor,
                                                     int phase_rotate, double ll, double mm, double nn, int64_t nrow,
                                                     int64_t nchan, int64_t npix, const int *__restrict__ perm,
                                                     double2 *__restrict__ out)
{
    const int W = WT ? WT : Wrt;
    const int64_t lane_idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (lane_idx >= nrow * nchan) return;
    const int64_t p = lane_idx / nchan, c = lane_idx - p * nchan;
    const int64_t r = perm ? perm[p] : p;     // rows in uv-tile order
    const int64_t idx = r * nchan + c;
    const double u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
    const double lam = wavelengths[c];
    const double su = u * scale_factor / lam, sv = v * scale_factor / lam;
    const double offset_u = su + (double)(npix / 2), offset_v = sv + (double)(npix / 2);
    const int64_t disc_u = (int64_t)rint(offset_u), disc_v = (int64_t)rint(offset_v);  // np.round: half to even
    const int frac_u = (int)((-offset_u + (double)disc_u) * os), frac_v = (int)((-offset_v + (double)disc_v) * os);
    const int klen = os * (W + 2);
    // tap weights (convolution_policies.py:228-247 packed, :303-310 unpacked; negative packed indices wrap)
    double ku[WT ? WT : MAXW], kv[WT ? WT : MAXW];
#pragma unroll
    for (int t = 0; t < (WT ? WT : MAXW); ++t) {
        if (t < W) {
            int iu = packed ? t + (frac_u < 0 ? 0 : 1) + frac_u * (W + 2) : (t + 1) * os + frac_u;
            int iv = packed ? t + (frac_v < 0 ? 0 : 1) + frac_v * (W + 2) : (t + 1) * os + frac_v;
            if (iu < 0) iu += klen;
            if (iv < 0) iv += klen;
            ku[t] = kernel[iu];
            kv[t] = kernel[iv];
        }
    }
    const double2 *__restrict__ gb = grid + chanmap[c] * npix * npix;
    double are = 0.0, aim = 0.0, cw = 0.0;
    const int64_t u0 = disc_u - W / 2, v0 = disc_v - W / 2;
    const bool interior = u0 >= 0 && u0 + W <= npix && v0 >= 0 && v0 + W <= npix;
    if (interior) {
#pragma unroll
        for (int tv = 0; tv < (WT ? WT : MAXW); ++tv) {
            if (tv < W) {
                const double2 *__restrict__ row = gb + (v0 + tv) * npix + u0;
                double rre = 0.0, rim = 0.0, rw = 0.0;
#pragma unroll
                for (int tu = 0; tu < (WT ? WT : MAXW); ++tu) {
                    if (tu < W) {
                        const double2 x = row[tu];
                        rre = fma(x.x, ku[tu], rre);
                        rim = fma(x.y, ku[tu], rim);
                        rw += ku[tu];
                    }
                }
                are = fma(rre, kv[tv], are);
                aim = fma(rim, kv[tv], aim);
                cw = fma(rw, kv[tv], cw);
            }
        }
    } else {
        for (int tv = 0; tv < W; ++tv) {
            const int64_t gv = v0 + tv;
            if (gv < 0 || gv >= npix) continue;
            for (int tu = 0; tu < W; ++tu) {
                const int64_t gu = u0 + tu;
                if (gu < 0 || gu >= npix) continue;
                const double2 x = gb[gv * npix + gu];
                const double wgt = kv[tv] * ku[tu];
                are = fma(x.x, wgt, are);
                aim = fma(x.y, wgt, aim);
                cw += wgt;
            }
        }
    }
    const double inv = 1.0 / (cw + 1.0e-8);
    are *= inv; aim *= inv;
    if (phase_rotate) {  // vis *= exp(-2 pi i (u ll + v mm + w nn) / lambda)   (phase_transform_policies.py:33-35)
        const double turns = -(u * ll + v * mm + w * nn) / lam;
        double pc, ps;
        sincos_quarter_turns<7>(4.0 * turns, pc, ps);
        const double tr = are * pc - aim * ps, ti = are * ps + aim * pc;
        are = tr; aim = ti;
    }
    double2 *o = out + idx * ncorr;
    for (int k = 0; k < ncorr; ++k) {
        const double2 f = coef[k];
        o[k] = make_double2(f.x * are - f.y * aim, f.x * aim + f.y * are);
    }
}

template <int N, int I = 0, typename F> __device__ __forceinline__ void static_for8(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for8<N, I + 1>(f);
    }
}

// value of lane (group base | K) in every lane of its 8-lane group: ds_swizzle bit-mask mode, and = 0x18, or = K
template <int K> __device__ __forceinline__ int group8_bcast(int x)
{
    return __builtin_amdgcn_ds_swizzle(x, (K << 5) | 0x18);
}
template <int K> __device__ __forceinline__ double group8_bcast(double x)
{
    return __hiloint2double(group8_bcast<K>(__double2hiint(x)), group8_bcast<K>(__double2loint(x)));
}

// Cooperative variant for W <= 8 taps.  With one lane per visibility every 16-byte gather instruction touches 64
// different cache lines, and the CU's address / tag path -- not L2 or HBM -- bounds the kernel
// (tools/microbench_gather.hip; 49 such instructions per visibility).  Here a lane still OWNS one visibility (its
// geometry is worked out once, by the owner), but the taps are read by eight lanes at a time: in round k the 8-lane
// group g samples the visibility of its lane k -- lane t takes tap column t, so a tap row's 7 adjacent cells are
// ONE instruction (1-2 lines per visibility, ~10 per instruction) and a visibility costs 7 gather instructions
// instead of 49.  The owner's geometry and the row weights reach the group by ds_swizzle broadcasts, the columns
// are summed across the group by a butterfly, and after 8 rounds every lane finishes its own visibility.  The sums
// run column-first instead of row-first: the same taps and weights, different rounding order (the reference
// compiles this loop with fastmath).  grid: ceil(nrow*nchan / 256).
template <int WT>  // compile-time tap count (0: runtime W <= 8)
__global__ __launch_bounds__(256) void degrid_coop_kernel(const double *__restrict__ uvw, const double2 *__restrict__ grid,
                                                          const double *__restrict__ wavelengths,
                                                          const int64_t *__restrict__ chanmap,
                                                          const double *__restrict__ kernel, int Wrt, int os, int packed,
                                                          int ncorr, const double2 *__restrict__ coef,
                                                          double scale_factor, int phase_rotate, double ll, double mm,
                                                          double nn, int64_t nrow, int64_t nchan, int64_t npix,
                                                          const int *__restrict__ perm, double2 *__restrict__ out)
{
    const int W = WT ? WT : Wrt;
    const int64_t vis_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = vis_raw < nrow * nchan;
    const int64_t vis_idx = live ? vis_raw : nrow * nchan - 1;
    const int64_t p = vis_idx / nchan, c = vis_idx - p * nchan;
    const int64_t r = perm ? perm[p] : p;          // rows in uv-tile order
    const double u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
    const double lam = wavelengths[c];
    const double su = u * scale_factor / lam, sv = v * scale_factor / lam;
    const double offset_u = su + (double)(npix / 2), offset_v = sv + (double)(npix / 2);
    const double du = rint(offset_u), dv = rint(offset_v);   // np.round: half to even
    // owner-side geometry handed to the group: first tap cell, oversampling phases, grid band
    const int own_fu = (int)((-offset_u + du) * os), own_fv = (int)((-offset_v + dv) * os);
    // far outside the grid every tap is invalid anyway: clamp so that the cell index fits 32 bits
    const double lim = (double)npix + 16.0;
    const int own_u0 = (int)fmin(fmax(du, -lim), lim) - W / 2, own_v0 = (int)fmin(fmax(dv, -lim), lim) - W / 2;
    const int own_band = (int)chanmap[c];
    const int klen = os * (W + 2);
    const int t = threadIdx.x & 7;                 // this lane's tap column (and the tap row it looks the weight up for)
    const int inp = (int)npix;
    double my_re = 0.0, my_im = 0.0, my_cw = 0.0;
    static_for8<8>([&](auto kc) {
        constexpr int K = decltype(kc)::value;
        const int u0 = group8_bcast<K>(own_u0), v0 = group8_bcast<K>(own_v0);
        const int fu = group8_bcast<K>(own_fu), fv = group8_bcast<K>(own_fv);
        const int band = group8_bcast<K>(own_band);
        const int gu = u0 + t;
        const bool col_ok = t < W && gu >= 0 && gu < inp;
        double ku = 0.0, kv_own = 0.0;
        if (col_ok) {
            int iu = packed ? t + (fu < 0 ? 0 : 1) + fu * (W + 2) : (t + 1) * os + fu;
            if (iu < 0) iu += klen;
            ku = kernel[iu];
        }
        if (t < W && v0 + t >= 0 && v0 + t < inp) {
            int iv = packed ? t + (fv < 0 ? 0 : 1) + fv * (W + 2) : (t + 1) * os + fv;
            if (iv < 0) iv += klen;
            kv_own = kernel[iv];
        }
        const double2 *__restrict__ gb = grid + (int64_t)band * npix * npix + gu;
        double cre = 0.0, cim = 0.0, skv = 0.0;   // column sums over the tap rows; sum of the valid row weights
        static_for8<(WT ? WT : 8)>([&](auto tvc) {
            constexpr int tv = decltype(tvc)::value;
            const double kv = group8_bcast<tv>(kv_own);
            const int gv = v0 + tv;
            skv += kv;
            if (col_ok && tv < W && gv >= 0 && gv < inp) {
                const double2 x = gb[(int64_t)gv * npix];
                cre = fma(x.x, kv, cre);
                cim = fma(x.y, kv, cim);
            }
        });
        double are = cre * ku, aim = cim * ku, sku = ku;
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            are += __shfl_xor(are, m, 64);
            aim += __shfl_xor(aim, m, 64);
            sku += __shfl_xor(sku, m, 64);
        }
        if (t == K) { my_re = are; my_im = aim; my_cw = sku * skv; }   // sum over the valid taps of kv[tv] * ku[tu]
    });
    if (!live) return;
    const double inv = 1.0 / (my_cw + 1.0e-8);
    double are = my_re * inv, aim = my_im * inv;
    if (phase_rotate) {  // vis *= exp(-2 pi i (u ll + v mm + w nn) / lambda)   (phase_transform_policies.py:33-35)
        const double turns = -(u * ll + v * mm + w * nn) / lam;
        double pc, ps;
        sincos_quarter_turns<7>(4.0 * turns, pc, ps);
        const double tr = are * pc - aim * ps, ti = are * ps + aim * pc;
        are = tr; aim = ti;
    }
    double2 *o = out + (r * nchan + c) * ncorr;
    for (int k = 0; k < ncorr; ++k) {
        const double2 f = coef[k];
        o[k] = make_double2(f.x * are - f.y * aim, f.x * aim + f.y * are);
    }
}

// ---- gridder (the adjoint): scatter with fp64 hardware atomics ------------------------------------------------
// One lane per visibility (row, chan): Stokes value = sum_k factor[k] * vis[k] * phase, then W x W atomic adds of
// weight * value into the band's grid; the visibility's weight sum over ALL taps (on or off the grid, as the
// reference counts them) is reduced per workgroup and added to the band's total.  Rows in uv-tile order, as in the
// degridder: the atomics of concurrent waves then fall into one cache-sized neighbourhood.  The order of the adds is
// not fixed, so results are reproducible to rounding (~1e-16 relative), not bit for bit.
template <int WT>
__global__ __launch_bounds__(256) void grid_kernel(const double *__restrict__ uvw, const double2 *__restrict__ vis,
                                                   const double *__restrict__ wavelengths,
                                                   const int64_t *__restrict__ chanmap, const double *__restrict__ kernel,
                                                   int Wrt, int os, int conv_policy, int ncorr,
                                                   const double2 *__restrict__ coef, double scale_factor, int phase_rotate,
                                                   double ll, double mm, double nn, int64_t nrow, int64_t nchan,
                                                   int64_t npix, int nband, const int *__restrict__ perm,
                                                   double *__restrict__ grid, double *__restrict__ wt)
{
    const int W = WT ? WT : Wrt;
    __shared__ double wsum[64];
    for (int b = threadIdx.x; b < 64; b += 256) wsum[b] = 0.0;
    __syncthreads();
    const int64_t lane_idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (lane_idx < nrow * nchan) {
        const int64_t p = lane_idx / nchan, c = lane_idx - p * nchan;
        const int64_t r = perm ? perm[p] : p;
        const double u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
        const double lam = wavelengths[c];
        double pc = 1.0, ps = 0.0;
        if (phase_rotate) sincos_quarter_turns<7>(4.0 * ((u * ll + v * mm + w * nn) / lam), pc, ps);
        double sre = 0.0, sim = 0.0;
        const double2 *x = vis + (r * nchan + c) * ncorr;
        for (int k = 0; k < ncorr; ++k) {
            const double xr = x[k].x * pc - x[k].y * ps, xi = x[k].x * ps + x[k].y * pc;
            sre += coef[k].x * xr - coef[k].y * xi;
            sim += coef[k].x * xi + coef[k].y * xr;
        }
        const double su = u * scale_factor / lam, sv = v * scale_factor / lam;
        const double offset_u = su + (double)(npix / 2), offset_v = sv + (double)(npix / 2);
        const int64_t disc_u = (int64_t)rint(offset_u), disc_v = (int64_t)rint(offset_v);
        const int band = (int)chanmap[c];
        double *gb = grid + (int64_t)band * npix * npix * 2;
        double cw = 0.0;
        if (conv_policy == 2) {  // nearest neighbour (convolution_policies.py:146-185; off-grid points are dropped)
            if (disc_u >= 0 && disc_u < npix && disc_v >= 0 && disc_v < npix) {
                unsafeAtomicAdd(gb + (disc_v * npix + disc_u) * 2, sre);
                unsafeAtomicAdd(gb + (disc_v * npix + disc_u) * 2 + 1, sim);
            }
            cw = 1.0;
        } else {
            const int frac_u = (int)((-offset_u + (double)disc_u) * os), frac_v = (int)((-offset_v + (double)disc_v) * os);
            const int klen = os * (W + 2);
            const bool packed = conv_policy == 1;
            double ku[WT ? WT : MAXW], kv[WT ? WT : MAXW];
#pragma unroll
            for (int t = 0; t < (WT ? WT : MAXW); ++t) {
                if (t < W) {
                    int iu = packed ? t + (frac_u < 0 ? 0 : 1) + frac_u * (W + 2) : (t + 1) * os + frac_u;
                    int iv = packed ? t + (frac_v < 0 ? 0 : 1) + frac_v * (W + 2) : (t + 1) * os + frac_v;
                    if (iu < 0) iu += klen;
                    if (iv < 0) iv += klen;
                    ku[t] = kernel[iu];
                    kv[t] = kernel[iv];
                }
            }
            const int64_t u0 = disc_u - W / 2, v0 = disc_v - W / 2;
#pragma unroll
            for (int tv = 0; tv < (WT ? WT : MAXW); ++tv) {
                if (tv < W) {
                    const int64_t gv = v0 + tv;
#pragma unroll
                    for (int tu = 0; tu < (WT ? WT : MAXW); ++tu) {
                        if (tu < W) {
                            const int64_t gu = u0 + tu;
                            const double wgt = kv[tv] * ku[tu];
                            if (gv >= 0 && gv < npix && gu >= 0 && gu < npix) {
                                unsafeAtomicAdd(gb + (gv * npix + gu) * 2, wgt * sre);
                                unsafeAtomicAdd(gb + (gv * npix + gu) * 2 + 1, wgt * sim);
                            }
                            cw += wgt;
                        }
                    }
                }
            }
        }
        if (band < 64) unsafeAtomicAdd(&wsum[band], cw);
        else unsafeAtomicAdd(&wt[band], cw);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 64 && b < nband; b += 256)
        if (wsum[b] != 0.0) unsafeAtomicAdd(&wt[b], wsum[b]);
}

// W = 7 wave-cooperative scatter.  Atomics are fast when one wave instruction covers a few contiguous row segments
// and ~17x slower when its 64 lanes hit 64 unrelated rows (MI355X_MICROARCH.md, atomics): every lane first sets up
// ITS visibility (Stokes value, tap weights, grid origin) into LDS, then the wave walks its 64 visibilities and
// issues, per visibility, two atomic instructions whose lanes are (tap row, tap column, re/im): 4 + 3 row segments
// of 14 contiguous doubles.
__global__ __launch_bounds__(256) void grid_wave7_kernel(const double *__restrict__ uvw, const double2 *__restrict__ vis,
                                                         const double *__restrict__ wavelengths,
                                                         const int64_t *__restrict__ chanmap,
                                                         const double *__restrict__ kernel, int os, int packed, int ncorr,
                                                         const double2 *__restrict__ coef, double scale_factor,
                                                         int phase_rotate, double ll, double mm, double nn, int64_t nrow,
                                                         int64_t nchan, int64_t npix, int nband,
                                                         const int *__restrict__ perm, double *__restrict__ grid,
                                                         double *__restrict__ wt)
{
    constexpr int W = 7;
    __shared__ double wsum[64];
    __shared__ double sk[4][64][2 * W + 2];   // per wave, per visibility: ku[7], kv[7], re, im
    __shared__ int so[4][64][4];              // u0, v0, band, live
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = threadIdx.x; b < 64; b += 256) wsum[b] = 0.0;
    __syncthreads();
    const int64_t lane_idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int live = 0;
    if (lane_idx < nrow * nchan) {
        const int64_t p = lane_idx / nchan, c = lane_idx - p * nchan;
        const int64_t r = perm ? perm[p] : p;
        const double u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
        const double lam = wavelengths[c];
        double pc = 1.0, ps = 0.0;
        if (phase_rotate) sincos_quarter_turns<7>(4.0 * ((u * ll + v * mm + w * nn) / lam), pc, ps);
        double sre = 0.0, sim = 0.0;
        const double2 *x = vis + (r * nchan + c) * ncorr;
        for (int k = 0; k < ncorr; ++k) {
            const double xr = x[k].x * pc - x[k].y * ps, xi = x[k].x * ps + x[k].y * pc;
            sre += coef[k].x * xr - coef[k].y * xi;
            sim += coef[k].x * xi + coef[k].y * xr;
        }
        const double offset_u = u * scale_factor / lam + (double)(npix / 2), offset_v = v * scale_factor / lam + (double)(npix / 2);
        // non-finite coordinates: the reference's int() of NaN is undefined; such visibilities are dropped here
        if (isfinite(offset_u) && isfinite(offset_v) && fabs(offset_u) < 1e9 && fabs(offset_v) < 1e9) {
            const int64_t disc_u = (int64_t)rint(offset_u), disc_v = (int64_t)rint(offset_v);
            const int frac_u = (int)((-offset_u + (double)disc_u) * os), frac_v = (int)((-offset_v + (double)disc_v) * os);
            const int klen = os * (W + 2);
            double su = 0.0, sv = 0.0;
#pragma unroll
            for (int t = 0; t < W; ++t) {
                int iu = packed ? t + (frac_u < 0 ? 0 : 1) + frac_u * (W + 2) : (t + 1) * os + frac_u;
                int iv = packed ? t + (frac_v < 0 ? 0 : 1) + frac_v * (W + 2) : (t + 1) * os + frac_v;
                if (iu < 0) iu += klen;
                if (iv < 0) iv += klen;
                const double a = kernel[iu], b = kernel[iv];
                sk[wave][lane][t] = a;
                sk[wave][lane][W + t] = b;
                su += a; sv += b;
            }
            sk[wave][lane][2 * W] = sre;
            sk[wave][lane][2 * W + 1] = sim;
            const int band = (int)chanmap[c];
            so[wave][lane][0] = (int)(disc_u - W / 2);
            so[wave][lane][1] = (int)(disc_v - W / 2);
            so[wave][lane][2] = band;
            live = 1;
            const double cw = su * sv;  // sum over all 49 taps of kv ku (the reference counts off-grid taps too)
            if (band < 64) unsafeAtomicAdd(&wsum[band], cw);
            else unsafeAtomicAdd(&wt[band], cw);
        }
    }
    so[wave][lane][3] = live;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // lane -> (row segment, position): 14 doubles per tap row = 7 taps x (re, im); 4 rows per instruction
    const int seg = lane / 14, pos = lane - seg * 14;       // seg 0..4 (lanes 56..63: seg 4, idle)
    const int tu = pos >> 1, part = pos & 1;
    for (int i = 0; i < 64; ++i) {
        if (!so[wave][i][3]) continue;                      // wave-uniform
        const int u0 = so[wave][i][0], v0 = so[wave][i][1];
        double *gb = grid + (int64_t)so[wave][i][2] * npix * npix * 2;
        const double val = sk[wave][i][2 * W + part] * sk[wave][i][tu];
        const int64_t gu = (int64_t)u0 + tu;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tv = half * 4 + seg;
            const int64_t gv = (int64_t)v0 + tv;
            if (seg < 4 && tv < W && gv >= 0 && gv < npix && gu >= 0 && gu < npix)
                unsafeAtomicAdd(gb + (gv * npix + gu) * 2 + part, val * sk[wave][i][W + tv]);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 64 && b < nband; b += 256)
        if (wsum[b] != 0.0) unsafeAtomicAdd(&wt[b], wsum[b]);
}

// gridder.py:114-116: every band divided by its weight sum + 1e-8
__global__ void grid_normalize_kernel(double2 *__restrict__ grid, const double *__restrict__ wt, int64_t npix2, int64_t total)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double d = wt[i / npix2] + 1.0e-8;
    grid[i] = make_double2(grid[i].x / d, grid[i].y / d);
}

}  // namespace

AF_EXPORT size_t af_degridder_workspace_bytes(int64_t nrow)
{
    if (nrow < 0) return 0;
    return af_align_up(NBIN * sizeof(int), 256) + af_align_up((size_t)nrow * sizeof(int), 256) +
           af_align_up((size_t)nrow * sizeof(unsigned short), 256);
}

AF_EXPORT int af_degridder_c128(const double *uvw, const double *gridstack, const double *wavelengths,
                                const int64_t *chanmap, double cell, const double *image_centre_host,
                                const double *phase_centre_host, const double *convolution_kernel, int64_t kernel_width,
                                int64_t kernel_oversampling, int phase_rotate, const double *corr_factors, int ncorr,
                                int packed, int64_t nrow, int64_t nchan, int64_t npix, double *out, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(nrow >= 0 && nchan >= 0 && npix >= 1, "af_degridder_c128: bad extents");
    AF_REQUIRE(kernel_width >= 1 && kernel_width <= MAXW && (kernel_width & 1), "af_degridder_c128: kernel width must be odd and <= %d",
               MAXW);
    AF_REQUIRE(kernel_oversampling >= 1, "af_degridder_c128: oversampling must be >= 1");
    AF_REQUIRE(ncorr == 2 || ncorr == 4, "Invalid stokes conversion");
    if (nrow == 0 || nchan == 0) return AF_OK;
    AF_REQUIRE(uvw && gridstack && wavelengths && chanmap && convolution_kernel && corr_factors && out &&
                   image_centre_host && phase_centre_host,
               "af_degridder_c128: NULL array");
    AF_REQUIRE(af_cdiv(nrow * nchan, 256) < (1LL << 31), "af_degridder_c128: problem too large for one launch");
    // degridder.py:131 and phase_transform_policies.py:21-32 (host scalars)
    const double scale_factor = npix * cell / 3600.0 * 3.141592653589793 / 180.0;
    const double ra0 = phase_centre_host[0], dec0 = phase_centre_host[1], ra = image_centre_host[0], dec = image_centre_host[1];
    const double d_ra = ra - ra0;
    const double ll = cos(dec) * sin(d_ra), mm = sin(dec) * cos(dec0) - cos(dec) * sin(dec0) * cos(d_ra);
    const double nn = -(1 - sqrt(1 - ll * ll - mm * mm));
    const dim3 grid((unsigned)af_cdiv(nrow * nchan, 256)), block(256);
    hipStream_t st = af_stream(stream);
    // rows in uv-tile order (skipped for small calls and when the caller passes no workspace)
    const int *perm = nullptr;
    const size_t need = af_degridder_workspace_bytes(nrow);
    if (workspace != nullptr && workspace_bytes >= need && nrow >= 4096 && nrow < (1LL << 31)) {
        AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_degridder_c128: workspace must be 256-byte aligned");
        char *ws = static_cast<char *>(workspace);
        int *hist = reinterpret_cast<int *>(ws);
        int *pm = reinterpret_cast<int *>(ws + af_align_up(NBIN * sizeof(int), 256));
        unsigned short *key = reinterpret_cast<unsigned short *>(ws + af_align_up(NBIN * sizeof(int), 256) +
                                                                 af_align_up((size_t)nrow * sizeof(int), 256));
        AF_HIP(hipMemsetAsync(hist, 0, NBIN * sizeof(int), st));
        hipLaunchKernelGGL(degrid_bin_kernel_dev, dim3((unsigned)af_cdiv(nrow, 256)), dim3(256), 0, st, uvw, nrow,
                           wavelengths, nchan, scale_factor, npix, key, hist);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL(degrid_scan_kernel, dim3(1), dim3(1024), 0, st, hist);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL(degrid_scatter_kernel, dim3((unsigned)af_cdiv(nrow, 256)), dim3(256), 0, st, key, nrow, hist, pm);
        AF_LAUNCH_CHECK();
        perm = pm;
    }
    const double2 *g = reinterpret_cast<const double2 *>(gridstack), *cf = reinterpret_cast<const double2 *>(corr_factors);
    double2 *o = reinterpret_cast<double2 *>(out);
    af_prof_begin(st);
    static const int coop_env = getenv("AFHIP_DEGRID_COOP") ? atoi(getenv("AFHIP_DEGRID_COOP")) : 1;
    const int64_t coop_blocks = af_cdiv(nrow * nchan, 256);
    if (coop_env && kernel_width <= 8 && coop_blocks < (1LL << 31) && npix < (1LL << 30)) {
        const dim3 cgrid((unsigned)coop_blocks);
        if (kernel_width == 7)
            hipLaunchKernelGGL((degrid_coop_kernel<7>), cgrid, block, 0, st, uvw, g, wavelengths, chanmap, convolution_kernel,
                               7, (int)kernel_oversampling, packed, ncorr, cf, scale_factor, phase_rotate, ll, mm, nn, nrow,
                               nchan, npix, perm, o);
        else
            hipLaunchKernelGGL((degrid_coop_kernel<0>), cgrid, block, 0, st, uvw, g, wavelengths, chanmap, convolution_kernel,
                               (int)kernel_width, (int)kernel_oversampling, packed, ncorr, cf, scale_factor, phase_rotate, ll,
                               mm, nn, nrow, nchan, npix, perm, o);
    } else if (kernel_width == 7)
        hipLaunchKernelGGL((degrid_kernel<7>), grid, block, 0, st, uvw, g, wavelengths, chanmap, convolution_kernel, 7,
                           (int)kernel_oversampling, packed, ncorr, cf, scale_factor, phase_rotate, ll, mm, nn, nrow, nchan,
                           npix, perm, o);
    else
        hipLaunchKernelGGL((degrid_kernel<0>), grid, block, 0, st, uvw, g, wavelengths, chanmap, convolution_kernel,
                           (int)kernel_width, (int)kernel_oversampling, packed, ncorr, cf, scale_factor, phase_rotate, ll,
                           mm, nn, nrow, nchan, npix, perm, o);
    af_prof_end(st);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

AF_EXPORT size_t af_gridder_workspace_bytes(int64_t nrow, int64_t nband)
{
    if (nrow < 0 || nband < 0) return 0;
    return af_degridder_workspace_bytes(nrow) + af_align_up((size_t)(nband > 0 ? nband : 1) * sizeof(double), 256);
}

AF_EXPORT int af_gridder_c128(const double *uvw, const double *vis, const double *wavelengths, const int64_t *chanmap,
                              int64_t npix, double cell, const double *image_centre_host, const double *phase_centre_host,
                              const double *convolution_kernel, int64_t kernel_width, int64_t kernel_oversampling,
                              int phase_rotate, const double *corr_factors, int ncorr, int conv_policy, int do_normalize,
                              int64_t nrow, int64_t nchan, int64_t nband, double *gridstack, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(nrow >= 0 && nchan >= 0 && npix >= 1 && nband >= 0, "af_gridder_c128: bad extents");
    AF_REQUIRE(conv_policy >= 0 && conv_policy <= 2, "Invalid convolution policy type");
    AF_REQUIRE(conv_policy == 2 || (kernel_width >= 1 && kernel_width <= MAXW && (kernel_width & 1)),
               "af_gridder_c128: kernel width must be odd and <= %d", MAXW);
    AF_REQUIRE(kernel_oversampling >= 1, "af_gridder_c128: oversampling must be >= 1");
    AF_REQUIRE(ncorr >= 1 && ncorr <= 4, "Invalid stokes conversion");
    hipStream_t st = af_stream(stream);
    if (nband == 0) return AF_OK;
    AF_REQUIRE(gridstack != nullptr, "af_gridder_c128: gridstack is NULL");
    AF_HIP(hipMemsetAsync(gridstack, 0, sizeof(double) * 2 * (size_t)(nband * npix * npix), st));
    if (nrow == 0 || nchan == 0) return AF_OK;
    AF_REQUIRE(uvw && vis && wavelengths && chanmap && corr_factors && image_centre_host && phase_centre_host &&
                   (convolution_kernel || conv_policy == 2),
               "af_gridder_c128: NULL array");
    const size_t need = af_gridder_workspace_bytes(nrow, nband);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= need, "af_gridder_c128: workspace too small (%zu < %zu)",
               workspace_bytes, need);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_gridder_c128: workspace must be 256-byte aligned");
    AF_REQUIRE(af_cdiv(nrow * nchan, 256) < (1LL << 31), "af_gridder_c128: problem too large for one launch");
    const double scale_factor = npix * cell / 3600.0 * 3.141592653589793 / 180.0;
    const double ra0 = phase_centre_host[0], dec0 = phase_centre_host[1], ra = image_centre_host[0], dec = image_centre_host[1];
    const double d_ra = ra - ra0;
    const double ll = cos(dec) * sin(d_ra), mm = sin(dec) * cos(dec0) - cos(dec) * sin(dec0) * cos(d_ra);
    const double nn = -(1 - sqrt(1 - ll * ll - mm * mm));
    char *ws = static_cast<char *>(workspace);
    double *wt = reinterpret_cast<double *>(ws + af_degridder_workspace_bytes(nrow));
    AF_HIP(hipMemsetAsync(wt, 0, (size_t)nband * sizeof(double), st));
    const int *perm = nullptr;
    if (nrow >= 4096 && nrow < (1LL << 31)) {
        int *hist = reinterpret_cast<int *>(ws);
        int *pm = reinterpret_cast<int *>(ws + af_align_up(NBIN * sizeof(int), 256));
        unsigned short *key = reinterpret_cast<unsigned short *>(ws + af_align_up(NBIN * sizeof(int), 256) +
                                                                 af_align_up((size_t)nrow * sizeof(int), 256));
        AF_HIP(hipMemsetAsync(hist, 0, NBIN * sizeof(int), st));
        hipLaunchKernelGGL(degrid_bin_kernel_dev, dim3((unsigned)af_cdiv(nrow, 256)), dim3(256), 0, st, uvw, nrow,
                           wavelengths, nchan, scale_factor, npix, key, hist);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL(degrid_scan_kernel, dim3(1), dim3(1024), 0, st, hist);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL(degrid_scatter_kernel, dim3((unsigned)af_cdiv(nrow, 256)), dim3(256), 0, st, key, nrow, hist, pm);
        AF_LAUNCH_CHECK();
        perm = pm;
    }
    const dim3 grid((unsigned)af_cdiv(nrow * nchan, 256)), block(256);
    const double2 *vs = reinterpret_cast<const double2 *>(vis), *cf = reinterpret_cast<const double2 *>(corr_factors);
    af_prof_begin(st);
    if (kernel_width == 7 && conv_policy != 2)
        hipLaunchKernelGGL(grid_wave7_kernel, grid, block, 0, st, uvw, vs, wavelengths, chanmap, convolution_kernel,
                           (int)kernel_oversampling, conv_policy == 1, ncorr, cf, scale_factor, phase_rotate, ll, mm, nn,
                           nrow, nchan, npix, (int)nband, perm, gridstack, wt);
    else
        hipLaunchKernelGGL((grid_kernel<0>), grid, block, 0, st, uvw, vs, wavelengths, chanmap, convolution_kernel,
                           (int)kernel_width, (int)kernel_oversampling, conv_policy, ncorr, cf, scale_factor, phase_rotate,
                           ll, mm, nn, nrow, nchan, npix, (int)nband, perm, gridstack, wt);
    af_prof_end(st);
    AF_LAUNCH_CHECK();
    if (do_normalize) {
        const int64_t total = nband * npix * npix;
        AF_REQUIRE(af_cdiv(total, 256) < (1LL << 31), "af_gridder_c128: grid too large for one launch");
        hipLaunchKernelGGL(grid_normalize_kernel, dim3((unsigned)af_cdiv(total, 256)), dim3(256), 0, st,
                           reinterpret_cast<double2 *>(gridstack), wt, npix * npix, total);
        AF_LAUNCH_CHECK();
    }
    return AF_OK;
}
