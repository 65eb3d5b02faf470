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

// The 4096 counters share 128 cache lines: with every row's atomic on ONE histogram, same-line atomics serialised
// (130 us for 1e6 rows, twice: count and scatter).  Eight histograms (block b uses copy b % 8, counters [bin][copy]): an
// eighth of the contention; the counting atomic's return value is kept as the row's rank inside its (bin, copy) slice, and
// one exclusive scan over the [bin][copy] array gives every slice its start -- the scatter needs no atomics.
constexpr int NCOPY = 8;
__global__ void degrid_bin_kernel_dev(const double *__restrict__ uvw, int64_t nrow,
                                      const double *__restrict__ wavelengths, int64_t nchan, double scale_factor,
                                      int64_t npix, unsigned short *__restrict__ key, int *__restrict__ hist,
                                      int *__restrict__ rank)
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
    rank[r] = atomicAdd(&hist[k * NCOPY + (blockIdx.x & (NCOPY - 1))], 1);
}

__global__ __launch_bounds__(1024) void degrid_scan_kernel(int *__restrict__ hist)  // in place: counts -> starts
{
    constexpr int PER = NBIN * NCOPY / 1024;
    __shared__ int part[1024];
    const int t = threadIdx.x;
    int v[PER], s = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { v[i] = hist[PER * t + i]; s += v[i]; }
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
    for (int i = 0; i < PER; ++i) { hist[PER * t + i] = base; base += v[i]; }
}

__global__ void degrid_scatter_kernel(const unsigned short *__restrict__ key, int64_t nrow, const int *__restrict__ start,
                                      const int *__restrict__ rank, int *__restrict__ perm)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    perm[start[(int)key[r] * NCOPY + (blockIdx.x & (NCOPY - 1))] + rank[r]] = (int)r;
}

// rows in uv-tile order: perm (nrow) in the workspace `ws` (af_degridder_workspace_bytes(nrow)); enqueued on st
const int *degrid_sort_rows(char *ws, const double *uvw, int64_t nrow, const double *wavelengths, int64_t nchan,
                            double scale_factor, int64_t npix, hipStream_t st)
{
    int *hist = reinterpret_cast<int *>(ws);
    size_t o = af_align_up((size_t)NBIN * NCOPY * sizeof(int), 256);
    int *pm = reinterpret_cast<int *>(ws + o);
    o += af_align_up((size_t)nrow * sizeof(int), 256);
    unsigned short *key = reinterpret_cast<unsigned short *>(ws + o);
    o += af_align_up((size_t)nrow * sizeof(unsigned short), 256);
    int *rank = reinterpret_cast<int *>(ws + o);
    if (hipMemsetAsync(hist, 0, (size_t)NBIN * NCOPY * sizeof(int), st) != hipSuccess) return nullptr;
    const dim3 g((unsigned)af_cdiv(nrow, 256));
    hipLaunchKernelGGL(degrid_bin_kernel_dev, g, dim3(256), 0, st, uvw, nrow, wavelengths, nchan, scale_factor, npix, key, hist, rank);
    hipLaunchKernelGGL(degrid_scan_kernel, dim3(1), dim3(1024), 0, st, hist);
    hipLaunchKernelGGL(degrid_scatter_kernel, g, dim3(256), 0, st, key, nrow, hist, rank, pm);
    return pm;
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
                                                          const int *__restrict__ perm, double2 *__restrict__ out,
                                                          int xcd_per)
{
    const int W = WT ? WT : Wrt;
    // rows arrive in uv-tile order: neighbouring blocks gather from the same neighbourhood of the grid.  With xcd_per > 0
    // each XCD (block i lives on XCD i % 8) takes a contiguous eighth of the blocks (xcd_per of them), so that its L2
    // holds one neighbourhood instead of a share of eight
    int64_t lb = blockIdx.x;
    if (xcd_per > 0) {
        lb = (int64_t)(blockIdx.x & 7) * xcd_per + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= xcd_per) return;
    }
    const int64_t vis_raw = lb * 256 + threadIdx.x;
    if (lb * 256 >= nrow * nchan) return;
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

// ======================================================================================================================
// NOT ON THE SURVEY 8 HOT PATH from here to af_degridder_c128's host code: the Perley GRIDDER (visibilities -> grid:
// grid_kernel, grid_wave7_kernel, grid_tile_*; entry af_gridder_c128), built in round 2 as the degridder's adjoint
// cross-check (gridding/perleypolyhedron/gridder.py:12-117).  Not tuned since, not a roofline row; the degridder above
// (BASELINE configs[4]) uses none of it.
// ======================================================================================================================
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


// ---- gridder, tile-binned: visibilities sorted by uv tile, accumulated in LDS, one flush per tile chunk ----------
// Global fp64 atomics run at ~1.3 TB/s on this chip whatever their locality (MI355X_MICROARCH.md, "Global float
// atomics": they execute at the memory side): 49 taps x 16 B x 6.4e7 visibilities = 50 GB of adds at configs[4] is
// ~40 ms on any global-atomic design.  Here the W x W scatter of a visibility goes into an LDS image of its uv tile
// instead, and only the finished tile image is added to the grid, once per chunk of visibilities:
//   1. grid_tile_count:   every visibility (row, chan) -> bin = (band, 32 x 32-cell tile of its nearest cell); counts
//   2. grid_tile_scan:    bin starts, and the prefix of the bins' chunk counts (a chunk = up to TILE_CHUNK visibilities
//                         of one bin = one workgroup: a crowded tile is shared by several workgroups)
//   3. grid_tile_scatter: the visibility indices of every bin, contiguous (counting sort; order inside a bin arbitrary)
//   4. grid_tile_kernel:  workgroup = ONE WAVE = one chunk.  Its LDS holds a private image of the 32 x 32 tile plus a
//                         halo of W/2 cells (23 KB at W = 7: six waves per CU); passes of 64 visibilities: every lane
//                         sets up ONE visibility (Stokes value, tap weights, origin) in LDS from operands fetched a pass
//                         ahead, then the wave walks the 64 with lanes = (tap row, tap column, re / im): a tap row is 2W
//                         contiguous doubles, 64 / 2W rows per instruction (W = 7: two read-add-write steps per
//                         visibility, row pitch chosen so that the rows of one instruction do not share banks) --
//                         plain LDS accesses, no atomics: measured, ds_add_f64 retires about one lane per clock per CU
//                         (13.9 ms for this pass with a shared 64 x 64 image and LDS atomics);
//                         finally the image is added to the grid (contiguous row segments: the fast atomic shape).
// No host synchronisation: the number of chunks is bounded by nvis / TILE_CHUNK + nbins, surplus workgroups exit.
// The order of the adds inside a tile follows the (arbitrary) order of the counting sort: reproducible to rounding.
constexpr int TILE = 16;
constexpr int TILE_CHUNK = 8192;
constexpr int TILE_THREADS = 64;       // one wave per workgroup: the LDS image is private to it

struct GridGeom {
    double scale_factor;
    int64_t npix, nchan;
    int ntx;        // tiles per axis
    int nbins;      // nband * ntx * ntx
};

// nearest cell of a visibility and whether it takes part at all (finite coordinates): THE definition used by the
// count, scatter and accumulate passes alike
__device__ __forceinline__ bool grid_vis_cell(const double *__restrict__ uvw, const double *__restrict__ wavelengths,
                                              const GridGeom &g, int64_t r, int64_t c, double &offset_u, double &offset_v,
                                              int64_t &disc_u, int64_t &disc_v)
{
    const double lam = wavelengths[c];
    offset_u = uvw[3 * r] * g.scale_factor / lam + (double)(g.npix / 2);
    offset_v = uvw[3 * r + 1] * g.scale_factor / lam + (double)(g.npix / 2);
    if (!(isfinite(offset_u) && isfinite(offset_v) && fabs(offset_u) < 1e9 && fabs(offset_v) < 1e9)) return false;
    disc_u = (int64_t)rint(offset_u);
    disc_v = (int64_t)rint(offset_v);
    return true;
}

__device__ __forceinline__ int grid_tile_of(int64_t disc, int ntx)
{
    int64_t t = disc >= 0 ? disc / TILE : -1;
    return (int)(t < 0 ? 0 : (t >= ntx ? ntx - 1 : t));
}

// key of visibility i, or -1 (dropped)
__device__ __forceinline__ int grid_vis_key(const double *__restrict__ uvw, const double *__restrict__ wavelengths,
                                            const int64_t *__restrict__ chanmap, const GridGeom &g, int64_t i)
{
    const int64_t r = i / g.nchan, c = i - r * g.nchan;
    double ou, ov;
    int64_t du, dv;
    if (!grid_vis_cell(uvw, wavelengths, g, r, c, ou, ov, du, dv)) return -1;
    return ((int)chanmap[c] * g.ntx + grid_tile_of(dv, g.ntx)) * g.ntx + grid_tile_of(du, g.ntx);
}

// Adjacent lanes are adjacent channels of a row: runs of equal keys.  One atomic per run (its first lane), the other
// lanes take their slot from the run's base: returns this lane's slot in its bin, or -1.
__device__ __forceinline__ int grid_run_atomic(int *__restrict__ counter, int key, bool want_slot)
{
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1, 64);
    const bool leader = lane == 0 || prev != key;
    const unsigned long long leaders = __ballot(leader);
    const unsigned long long above = lane == 63 ? 0ULL : (leaders >> (lane + 1));
    const int run = above ? __ffsll((long long)above) : 64 - lane;   // lanes up to the next leader
    int base = 0;
    if (leader && key >= 0) base = atomicAdd(&counter[key], run);
    if (!want_slot) return 0;
    // the run's first lane: highest leader bit at or below this lane
    const unsigned long long below = leaders & (~0ULL >> (63 - lane));
    const int first = 63 - __clzll((long long)below);
    base = __shfl(base, first, 64);
    return key >= 0 ? base + (lane - first) : -1;
}

__global__ __launch_bounds__(256) void grid_tile_count(const double *__restrict__ uvw, const double *__restrict__ wavelengths,
                                                       const int64_t *__restrict__ chanmap, GridGeom g, int64_t nvis,
                                                       int *__restrict__ count)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < nvis; i0 += stride) {   // whole waves stay together
        const int64_t i = i0 + threadIdx.x;
        const int key = i < nvis ? grid_vis_key(uvw, wavelengths, chanmap, g, i) : -1;
        grid_run_atomic(count, key, false);
    }
}

// one block: start[b] = exclusive prefix of count (start[nbins] = total), cursor = start, cstart[b] = exclusive prefix of
// the bins' chunk counts (cstart[nbins] = number of chunks)
__global__ __launch_bounds__(1024) void grid_tile_scan(const int *__restrict__ count, int nbins, int *__restrict__ start,
                                                       int *__restrict__ cursor, int *__restrict__ cstart)
{
    __shared__ int part[1024], cpart[1024];
    const int t = threadIdx.x;
    const int per = (nbins + 1023) / 1024;
    const int lo = t * per, hi = (lo + per < nbins) ? lo + per : nbins;
    int s = 0, cs = 0;
    for (int b = lo; b < hi; ++b) { s += count[b]; cs += (count[b] + TILE_CHUNK - 1) / TILE_CHUNK; }
    part[t] = s; cpart[t] = cs;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int a = t >= off ? part[t - off] : 0, ca = t >= off ? cpart[t - off] : 0;
        __syncthreads();
        part[t] += a; cpart[t] += ca;
        __syncthreads();
    }
    int base = part[t] - s, cbase = cpart[t] - cs;
    for (int b = lo; b < hi; ++b) {
        start[b] = base; cursor[b] = base; cstart[b] = cbase;
        base += count[b]; cbase += (count[b] + TILE_CHUNK - 1) / TILE_CHUNK;
    }
    if (t == 1023) { start[nbins] = part[1023]; cstart[nbins] = cpart[1023]; }
}

__global__ __launch_bounds__(256) void grid_tile_scatter(const double *__restrict__ uvw, const double *__restrict__ wavelengths,
                                                         const int64_t *__restrict__ chanmap, GridGeom g, int64_t nvis,
                                                         int *__restrict__ cursor, unsigned *__restrict__ idx)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < nvis; i0 += stride) {
        const int64_t i = i0 + threadIdx.x;
        const int key = i < nvis ? grid_vis_key(uvw, wavelengths, chanmap, g, i) : -1;
        const int slot = grid_run_atomic(cursor, key, true);
        if (slot >= 0) idx[slot] = (unsigned)i;
    }
}

template <int W>
__global__ __launch_bounds__(TILE_THREADS) void grid_tile_kernel(
    const double *__restrict__ uvw, const double2 *__restrict__ vis, const double *__restrict__ wavelengths,
    const double *__restrict__ kernel, int os, int packed, int ncorr, const double2 *__restrict__ coef, GridGeom g,
    int phase_rotate, double ll, double mm, double nn, const int *__restrict__ start, const int *__restrict__ cstart,
    const unsigned *__restrict__ idx, double *__restrict__ grid, double *__restrict__ wt)
{
    constexpr int H = W / 2;
    constexpr int REG = TILE + 2 * H;                       // cells per side of the LDS image
    // doubles per image row: >= 2 REG and = 16 mod 32, so that consecutive tap rows start 32 banks apart
    constexpr int PITCH = ((2 * REG - 16 + 31) / 32) * 32 + 16;
    constexpr int ROWLEN = 2 * W;                           // doubles of one tap row
    constexpr int RPI = 64 / ROWLEN < W ? 64 / ROWLEN : W;  // tap rows per instruction
    constexpr int NINSTR = (W + RPI - 1) / RPI;
    constexpr int DUMMY = REG * PITCH;                      // one spare cell per lane: the target of idle lanes
    __shared__ double tile[REG * PITCH + 64];               // this WAVE's private image of its tile (+ halo)
    extern __shared__ double ktab[];                        // the convolution kernel: os * (W + 2) taps
    const int lane = threadIdx.x;

    // which (bin, chunk) am I: largest bin with cstart[bin] <= blockIdx.x
    const int nchunks = cstart[g.nbins];
    if ((int)blockIdx.x >= nchunks) return;
    int lo = 0, hi = g.nbins;                               // invariant: cstart[lo] <= w < cstart[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cstart[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const int bin = lo, chunk = (int)blockIdx.x - cstart[bin];
    const int first = start[bin] + chunk * TILE_CHUNK;
    const int last = (first + TILE_CHUNK < start[bin + 1]) ? first + TILE_CHUNK : start[bin + 1];
    const int tx = bin % g.ntx, ty = (bin / g.ntx) % g.ntx, band = bin / (g.ntx * g.ntx);
    const int64_t reg_u0 = (int64_t)tx * TILE - H, reg_v0 = (int64_t)ty * TILE - H;   // grid cell of image (0, 0)
    double *gb = grid + (int64_t)band * g.npix * g.npix * 2;

    const int klen = os * (W + 2);
    for (int e = lane; e < REG * PITCH + 64; e += 64) tile[e] = 0.0;
    for (int e = lane; e < klen; e += 64) ktab[e] = kernel[e];

    // this lane's role in the wave's walk: (tap row of the instruction, tap column, re / im).  Everything that
    // depends on the lane only is folded into constants, so that a visibility costs the lane one multiply-add per
    // address: byte address = on1 * (image offset of the visibility) + off8, with on1 = 0 and off8 = the lane's spare
    // cell for the lanes that have no tap in instruction k (they add garbage to a cell nobody reads).
    const int seg = lane / ROWLEN, pos = lane - seg * ROWLEN, tu = pos >> 1, part = pos & 1;
    const int kstep = packed ? 1 : os;                      // kernel index step between consecutive taps
    unsigned on1[NINSTR], off8[NINSTR], kv8[NINSTR];
#pragma unroll
    for (int k = 0; k < NINSTR; ++k) {
        const int tv = k * RPI + seg;
        const bool on = seg < RPI && tv < W;
        on1[k] = on ? 1u : 0u;
        off8[k] = 8u * (unsigned)(on ? tv * PITCH + tu * 2 + part : DUMMY + lane);
        kv8[k] = 8u * (unsigned)((on ? tv : 0) * kstep);
    }
    const unsigned ku8 = 8u * (unsigned)((seg < RPI ? tu : 0) * kstep);
    const double is_im = part ? 1.0 : 0.0, is_re = part ? 0.0 : 1.0;
    const char *tile_b = reinterpret_cast<const char *>(tile);
    const char *ktab_b = reinterpret_cast<const char *>(ktab);
    double cw_sum = 0.0;
    // raw operands of a visibility, fetched one pass ahead of their use (and its index two passes ahead): the
    // dependent global loads index -> (uvw, vis) never sit between two accumulation loops
    struct Raw { double u, v, w, lam; double2 x[4]; bool have; };
    auto fetch = [&](int j, unsigned i32) {
        Raw R;
        R.have = j < last;
        const int64_t i = R.have ? (int64_t)i32 : 0;
        const int64_t r = i / g.nchan, c = i - r * g.nchan;
        R.u = uvw[3 * r]; R.v = uvw[3 * r + 1]; R.w = uvw[3 * r + 2];
        R.lam = wavelengths[c];
        const double2 *x = vis + (r * g.nchan + c) * ncorr;
#pragma unroll
        for (int k = 0; k < 4; ++k) R.x[k] = k < ncorr ? x[k] : make_double2(0.0, 0.0);
        return R;
    };
    unsigned i_next = first + lane < last ? idx[first + lane] : 0u;
    Raw nxt = fetch(first + lane, i_next);
    i_next = first + 64 + lane < last ? idx[first + 64 + lane] : 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int base = first; base < last; base += 64) {
        // ---- every lane sets up one visibility (operands fetched during the previous pass), in registers
        const Raw cur = nxt;
        nxt = fetch(base + 64 + lane, i_next);
        i_next = base + 128 + lane < last ? idx[base + 128 + lane] : 0u;
        int img = -1, iu = 0, iv = 0;           // image offset of the footprint's first cell (-1: not for the walk)
        double sre = 0.0, sim = 0.0;
        if (cur.have) {
            const double offset_u = cur.u * g.scale_factor / cur.lam + (double)(g.npix / 2);
            const double offset_v = cur.v * g.scale_factor / cur.lam + (double)(g.npix / 2);
            const int64_t disc_u = (int64_t)rint(offset_u), disc_v = (int64_t)rint(offset_v);   // binned: finite
            double pc = 1.0, ps = 0.0;
            if (phase_rotate) sincos_quarter_turns<7>(4.0 * ((cur.u * ll + cur.v * mm + cur.w * nn) / cur.lam), pc, ps);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < ncorr) {
                    const double xr = cur.x[k].x * pc - cur.x[k].y * ps, xi = cur.x[k].x * ps + cur.x[k].y * pc;
                    sre += coef[k].x * xr - coef[k].y * xi;
                    sim += coef[k].x * xi + coef[k].y * xr;
                }
            }
            const int frac_u = (int)((-offset_u + (double)disc_u) * os), frac_v = (int)((-offset_v + (double)disc_v) * os);
            // kernel index of tap t = first index + t * kstep (convolution_policies.py:228-247 packed: the wrap of
            // negative indices is the same for every tap of a visibility; :303-310 unpacked: never negative)
            iu = packed ? (frac_u < 0 ? 0 : 1) + frac_u * (W + 2) : os + frac_u;
            iv = packed ? (frac_v < 0 ? 0 : 1) + frac_v * (W + 2) : os + frac_v;
            if (iu < 0) iu += klen;
            if (iv < 0) iv += klen;
            double su = 0.0, sv = 0.0;
#pragma unroll
            for (int t = 0; t < W; ++t) { su += ktab[iu + t * kstep]; sv += ktab[iv + t * kstep]; }
            cw_sum += su * sv;      // all W x W taps, on or off the grid, as the reference counts them
            const int u0 = (int)(disc_u - H - reg_u0), v0 = (int)(disc_v - H - reg_v0);   // footprint origin in the image
            if (u0 >= 0 && u0 + W <= REG && v0 >= 0 && v0 + W <= REG) {
                img = v0 * PITCH + u0 * 2;      // whole footprint inside the image: every centre inside the tile
            } else {
                // centre beyond the grid (clamped into an edge tile): its few in-grid taps go straight to the grid
                for (int tv = 0; tv < W; ++tv)
                    for (int t = 0; t < W; ++t) {
                        const int64_t cu = disc_u - H + t, cv = disc_v - H + tv;
                        if (cu >= 0 && cu < g.npix && cv >= 0 && cv < g.npix) {
                            const double wgt = ktab[iv + tv * kstep] * ktab[iu + t * kstep];
                            unsafeAtomicAdd(gb + (cv * g.npix + cu) * 2, wgt * sre);
                            unsafeAtomicAdd(gb + (cv * g.npix + cu) * 2 + 1, wgt * sim);
                        }
                    }
            }
        }
        // ---- walk the visibilities that have their footprint in the image: lane j's registers hold visibility j,
        // read by v_readlane; the tap weights come from the kernel table in LDS; the image update is a plain
        // read-add-write (the image is private to the wave, the lanes of an instruction hit distinct cells, the LDS
        // executes a wave's accesses in order).  The operands of the next visibility are staged between the image reads
        // and the image writes of the current one, so a visibility costs one LDS round trip.
        unsigned long long todo = __ballot(img >= 0);
        const int kidx = iu | (iv << 16);                    // both below 2^15 (checked on the host)
        struct Stage { unsigned a[NINSTR]; double x[NINSTR]; };
        auto stage = [&](int i) {
            Stage S;
            const unsigned img_i = 8u * (unsigned)__builtin_amdgcn_readlane(img, i);
            const int kidx_i = __builtin_amdgcn_readlane(kidx, i);
            const unsigned iu_i = 8u * (unsigned)(kidx_i & 0xffff), iv_i = 8u * (unsigned)(kidx_i >> 16);
            const double re_i = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sre), i),
                                                 __builtin_amdgcn_readlane(__double2loint(sre), i));
            const double im_i = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sim), i),
                                                 __builtin_amdgcn_readlane(__double2loint(sim), i));
            const double ku = *reinterpret_cast<const double *>(ktab_b + (iu_i + ku8));
            double kv[NINSTR];
#pragma unroll
            for (int k = 0; k < NINSTR; ++k) kv[k] = *reinterpret_cast<const double *>(ktab_b + (iv_i + kv8[k]));
            const double val = fma(is_im, im_i, is_re * re_i) * ku;
#pragma unroll
            for (int k = 0; k < NINSTR; ++k) {
                S.a[k] = on1[k] * img_i + off8[k];          // v_mad_u32_u24: img_i < 2^24
                S.x[k] = val * kv[k];
            }
            return S;
        };
        if (todo) {
            int i = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            Stage S = stage(i);
            while (true) {
                double c[NINSTR];
#pragma unroll
                for (int k = 0; k < NINSTR; ++k) c[k] = *reinterpret_cast<const double *>(tile_b + S.a[k]);
                const bool more = todo != 0;
                i = more ? __ffsll((long long)todo) - 1 : i;
                todo &= todo - 1;
                Stage N = stage(i);                           // its table reads queue behind the image reads ...
#pragma unroll
                for (int k = 0; k < NINSTR; ++k) asm volatile("" : "+v"(N.x[k]));   // ... and are issued before the writes
#pragma unroll
                for (int k = 0; k < NINSTR; ++k)
                    *reinterpret_cast<double *>(const_cast<char *>(tile_b) + S.a[k]) = c[k] + S.x[k];
                if (!more) break;
                S = N;
            }
        }
    }
    // ---- weight sum of the chunk's visibilities
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cw_sum += __shfl_xor(cw_sum, o, 64);
    if (lane == 0 && cw_sum != 0.0) unsafeAtomicAdd(&wt[band], cw_sum);
    // ---- flush: image rows are contiguous on the grid; image cells beyond the grid's edge are dropped here
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < REG * REG * 2; e += 64) {
        const int gv = e / (2 * REG), rem = e - gv * (2 * REG);
        const double x = tile[gv * PITCH + rem];
        const int64_t cell_v = reg_v0 + gv, cell_u = reg_u0 + (rem >> 1);
        if (x != 0.0 && cell_v >= 0 && cell_v < g.npix && cell_u >= 0 && cell_u < g.npix)
            unsafeAtomicAdd(gb + (cell_v * g.npix + cell_u) * 2 + (rem & 1), x);
    }
}

struct GridTileWs { size_t count, start, cursor, cstart, idx, total; };
GridTileWs grid_tile_ws(int64_t nvis, int64_t nbins)
{
    GridTileWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    w.count = take((size_t)(nbins + 1) * sizeof(int));
    w.start = take((size_t)(nbins + 1) * sizeof(int));
    w.cursor = take((size_t)(nbins + 1) * sizeof(int));
    w.cstart = take((size_t)(nbins + 1) * sizeof(int));
    w.idx = take((size_t)nvis * sizeof(unsigned));
    w.total = o;
    return w;
}

// the tile path serves kernel widths 3 / 5 / 7 / 9 with problem sizes whose indices fit 32 bits
bool grid_tile_ok(int64_t nrow, int64_t nchan, int64_t nband, int64_t npix, int64_t W, int conv_policy)
{
    if (conv_policy == 2 || !(W == 3 || W == 5 || W == 7 || W == 9)) return false;
    const int64_t ntx = af_cdiv(npix, TILE);
    return nrow * nchan < (1LL << 31) && nband * ntx * ntx < (1LL << 22) && nrow * nchan >= 4096;
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
    return af_align_up((size_t)NBIN * NCOPY * sizeof(int), 256) + 2 * af_align_up((size_t)nrow * sizeof(int), 256) +
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
        perm = degrid_sort_rows(static_cast<char *>(workspace), uvw, nrow, wavelengths, nchan, scale_factor, npix, st);
        AF_REQUIRE(perm != nullptr, "af_degridder_c128: hipMemsetAsync failed");
        AF_LAUNCH_CHECK();
    }
    const double2 *g = reinterpret_cast<const double2 *>(gridstack), *cf = reinterpret_cast<const double2 *>(corr_factors);
    double2 *o = reinterpret_cast<double2 *>(out);
    af_prof_begin(st);
    static const int coop_env = getenv("AFHIP_DEGRID_COOP") ? atoi(getenv("AFHIP_DEGRID_COOP")) : 1;
    const int64_t coop_blocks = af_cdiv(nrow * nchan, 256);
    if (coop_env && kernel_width <= 8 && coop_blocks < (1LL << 31) && npix < (1LL << 30)) {
        const int xcd_on = getenv("AFHIP_DEGRID_XCD") ? atoi(getenv("AFHIP_DEGRID_XCD")) : 1;    // read per call (A/B)
        const int xcd_per = (xcd_on && perm != nullptr && coop_blocks + 8 < (1LL << 31)) ? (int)af_cdiv(coop_blocks, 8) : 0;
        const dim3 cgrid((unsigned)(xcd_per ? 8 * (int64_t)xcd_per : coop_blocks));
        if (kernel_width == 7)
            hipLaunchKernelGGL((degrid_coop_kernel<7>), cgrid, block, 0, st, uvw, g, wavelengths, chanmap, convolution_kernel,
                               7, (int)kernel_oversampling, packed, ncorr, cf, scale_factor, phase_rotate, ll, mm, nn, nrow,
                               nchan, npix, perm, o, xcd_per);
        else
            hipLaunchKernelGGL((degrid_coop_kernel<0>), cgrid, block, 0, st, uvw, g, wavelengths, chanmap, convolution_kernel,
                               (int)kernel_width, (int)kernel_oversampling, packed, ncorr, cf, scale_factor, phase_rotate, ll,
                               mm, nn, nrow, nchan, npix, perm, o, xcd_per);
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

AF_EXPORT size_t af_gridder_workspace_bytes(int64_t nrow, int64_t nchan, int64_t nband, int64_t npix)
{
    if (nrow < 0 || nchan < 0 || nband < 0 || npix < 0) return 0;
    size_t b = af_degridder_workspace_bytes(nrow) + af_align_up((size_t)(nband > 0 ? nband : 1) * sizeof(double), 256);
    const int64_t ntx = af_cdiv(npix > 0 ? npix : 1, TILE);
    if (nrow * nchan < (1LL << 31) && nband * ntx * ntx < (1LL << 22))   // the tile path's index arrays
        b += grid_tile_ws(nrow * nchan, nband * ntx * ntx).total;
    return b;
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
    const size_t need = af_gridder_workspace_bytes(nrow, nchan, nband, npix);
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
    const double2 *vs = reinterpret_cast<const double2 *>(vis), *cf = reinterpret_cast<const double2 *>(corr_factors);
    static const int tile_env = getenv("AFHIP_GRID_TILES") ? atoi(getenv("AFHIP_GRID_TILES")) : 1;   // A/B hook
    if (tile_env && grid_tile_ok(nrow, nchan, nband, npix, kernel_width, conv_policy)) {
        // ---- tile-binned LDS accumulation
        GridGeom g;
        g.scale_factor = scale_factor; g.npix = npix; g.nchan = nchan;
        g.ntx = (int)af_cdiv(npix, TILE); g.nbins = (int)(nband * g.ntx * g.ntx);
        const int64_t nvis = nrow * nchan;
        const GridTileWs T = grid_tile_ws(nvis, g.nbins);
        char *tw = ws + af_degridder_workspace_bytes(nrow) + af_align_up((size_t)nband * sizeof(double), 256);
        int *count = reinterpret_cast<int *>(tw + T.count), *start = reinterpret_cast<int *>(tw + T.start);
        int *cursor = reinterpret_cast<int *>(tw + T.cursor), *cstart = reinterpret_cast<int *>(tw + T.cstart);
        unsigned *idx = reinterpret_cast<unsigned *>(tw + T.idx);
        AF_HIP(hipMemsetAsync(count, 0, (size_t)(g.nbins + 1) * sizeof(int), st));
        int64_t blocks = af_cdiv(nvis, 256);
        if (blocks > 16384) blocks = 16384;
        af_prof_begin(st);
        hipLaunchKernelGGL(grid_tile_count, dim3((unsigned)blocks), dim3(256), 0, st, uvw, wavelengths, chanmap, g, nvis, count);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL(grid_tile_scan, dim3(1), dim3(1024), 0, st, count, g.nbins, start, cursor, cstart);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL(grid_tile_scatter, dim3((unsigned)blocks), dim3(256), 0, st, uvw, wavelengths, chanmap, g, nvis,
                           cursor, idx);
        AF_LAUNCH_CHECK();
        const int64_t max_chunks = nvis / TILE_CHUNK + g.nbins;
        const size_t ktab_bytes = (size_t)kernel_oversampling * (kernel_width + 2) * sizeof(double);
        AF_REQUIRE(ktab_bytes <= 32 * 1024, "af_gridder_c128: convolution kernel table too large for LDS");
        auto run = [&](auto kernel) -> int {
            hipLaunchKernelGGL(kernel, dim3((unsigned)max_chunks), dim3(TILE_THREADS), ktab_bytes, st, uvw, vs, wavelengths,
                               convolution_kernel, (int)kernel_oversampling, (int)(conv_policy == 1), ncorr, cf, g,
                               phase_rotate, ll, mm, nn, start, cstart, idx, gridstack, wt);
            AF_LAUNCH_CHECK();
            return AF_OK;
        };
        int rc;
        switch (kernel_width) {
        case 3: rc = run(grid_tile_kernel<3>); break;
        case 5: rc = run(grid_tile_kernel<5>); break;
        case 7: rc = run(grid_tile_kernel<7>); break;
        default: rc = run(grid_tile_kernel<9>); break;
        }
        if (rc != AF_OK) return rc;
        af_prof_end(st);
    } else {
        // ---- global fp64 atomics (other kernel widths, nearest-neighbour policy, small calls), rows in uv-tile order
        const int *perm = nullptr;
        if (nrow >= 4096 && nrow < (1LL << 31)) {
            perm = degrid_sort_rows(ws, uvw, nrow, wavelengths, nchan, scale_factor, npix, st);
            AF_REQUIRE(perm != nullptr, "af_gridder_c128: hipMemsetAsync failed");
            AF_LAUNCH_CHECK();
        }
        const dim3 grid((unsigned)af_cdiv(nrow * nchan, 256)), block(256);
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
    }
    if (do_normalize) {
        const int64_t total = nband * npix * npix;
        AF_REQUIRE(af_cdiv(total, 256) < (1LL << 31), "af_gridder_c128: grid too large for one launch");
        hipLaunchKernelGGL(grid_normalize_kernel, dim3((unsigned)af_cdiv(total, 256)), dim3(256), 0, st,
                           reinterpret_cast<double2 *>(gridstack), wt, npix * npix, total);
        AF_LAUNCH_CHECK();
    }
    return AF_OK;
}
