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
    if (kernel_width == 7)
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
