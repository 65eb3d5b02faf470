// Convolutional degridder for gfx950 (BASELINE configs[4]).
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
