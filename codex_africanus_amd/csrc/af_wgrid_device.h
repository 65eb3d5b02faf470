// Device helpers and constants shared by the two translation units of the wgridder: af_wgridder.hip (image -> visibilities,
// the SURVEY 8 row, and the host section of both directions) and af_wgridder_adjoint.hip (visibilities -> image: not on the
// hot path, kept as the forward operator's cross-check).  Both directions take their tap weights, the w fold and the first
// cell of a visibility's support from the functions below, so that the transpose relation between them holds to rounding.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "af_common.h"

// per-tap polynomials of the kernel psi (wg_taps below; fitted on the host by wg_fit_poly in af_wgridder.hip)
constexpr int WG_POLYW = 10, WG_POLYD = WG_POLYW + 2;
struct WgPoly { double c[WG_POLYW][WG_POLYD + 1]; };

// the gridding (adjoint) direction's tiling: shared with the host section, which sizes the sort and the workspace by it
constexpr int WG_GCHUNK = 4096;
constexpr int WG_GKB = 256;          // buckets (first planes) of one exact sort: calls with more planes sort per batch of planes
__host__ __device__ constexpr int wg_gtile(int W) { return W <= 8 ? 12 : (W <= 12 ? 8 : 4); }   // ring + table <= 160 KB

// The adjoint's device passes (af_wgridder_adjoint.hip), enqueued on `st` by the host section of af_wgridder.hip.  Hidden
// symbols (-fvisibility=hidden): not part of the C ABI.
struct WgSpreadArgs {
    const double *uvw, *freq;
    int64_t nrow, nchan_b, chan0, nchan_total;
    double2 *grids;
    int64_t nu, nv;
    double cellx, celly, beta, w0, dw;
    int pk0, pk1, do_w;
    const unsigned char *mask;      // small calls (one lane per visibility)
    const double *wgt;
    const double2 *vis;
    const unsigned *idx;            // large calls (sorted by tile and first plane): the sort's outputs
    const int *start;
    int kb;
    const int2 *chunks;
    const int *nchunks;
};
// the visibilities of planes pk0 .. pk1 - 1 spread onto the grids; tiled: `blocks` = the chunk bound, else ceil(nvis / 256)
void wg_adjoint_spread(int kernel_width, bool tiled, unsigned blocks, hipStream_t st, const WgSpreadArgs &a, const WgPoly &poly);
void wg_adjoint_gather_rows(const double2 *G, int64_t nx, int64_t nu, int64_t nv, double2 *S, hipStream_t st);
void wg_adjoint_add_plane(const double2 *S, const double *A, const double *nm1, int64_t nx, int64_t ny, int64_t nv, double wk,
                          int first, double *image, hipStream_t st);

namespace {

__device__ __forceinline__ double es_kernel(double t, double inv_half_w, double beta)
{
    const double x = t * inv_half_w;           // [-1, 1] inside the support
    const double s = 1.0 - x * x;
    return s > 0.0 ? exp(beta * (sqrt(s) - 1.0)) : 0.0;
}

// The W taps of a visibility along one axis: psi at offsets f, f + 1, ..., f + W - 1 from the visibility, f = (first cell)
// - (position) in [-W/2, -W/2 + 1).  An exp and a sqrt in fp64 per tap cost ~150 instructions; for W <= 10 the taps come
// from per-tap polynomials in u = 2 (f + W/2) - 1 instead (degree W + 2, Horner, coefficients in the kernel arguments =
// scalar operands), as ducc0 does.  psi has a square-root singularity at the ends of its support, where it is ~10^-W:
// the fit stalls at an absolute error of ~5 10^-(W+1) = epsilon / 200, which is what the accuracy contract can ignore;
// the deconvolution keeps the exact psihat.  Every kernel of both directions takes its weights from this one function
// (the transpose relation between `model` and `dirty` holds to rounding only if they do).
template <int W>
__device__ __forceinline__ void wg_taps(const WgPoly &P, double f, double beta, double (&out)[W])
{
    if constexpr (W <= WG_POLYW) {
        const double u = 2.0 * (f + 0.5 * (double)W) - 1.0;
#pragma unroll
        for (int a = 0; a < W; ++a) {
            double acc = P.c[a][W + 2];
#pragma unroll
            for (int d = W + 1; d >= 0; --d) acc = fma(acc, u, P.c[a][d]);
            out[a] = acc;
        }
    } else {
        constexpr double inv_half_w = 2.0 / (double)W;
#pragma unroll
        for (int a = 0; a < W; ++a) out[a] = es_kernel(f + (double)a, inv_half_w, beta);
    }
}
// The three axes' taps of one visibility in ONE walk over the coefficients (same Horner steps per polynomial as wg_taps:
// the same bits).  The coefficients are kernel arguments -- W (W + 3) doubles in scalar registers, 140 registers at W = 7,
// more than a wave has: three separate walks made the compiler keep them all and park them in vector-register lanes
// (v_writelane / v_readlane: ~400 instructions per chunk of the tile kernel); walked once, each is loaded, used three times
// and forgotten.
template <int W>
__device__ __forceinline__ void wg_taps3(const WgPoly &P, double fu, double fv, double fw, double beta, double (&ku)[W],
                                         double (&kv)[W], double (&kw)[W])
{
    if constexpr (W <= WG_POLYW) {
        const double uu = 2.0 * (fu + 0.5 * (double)W) - 1.0, uv = 2.0 * (fv + 0.5 * (double)W) - 1.0,
                     uw = 2.0 * (fw + 0.5 * (double)W) - 1.0;
#pragma unroll
        for (int a = 0; a < W; ++a) {
            double au = P.c[a][W + 2], av = au, aw = au;
#pragma unroll
            for (int d = W + 1; d >= 0; --d) {
                const double c = P.c[a][d];
                au = fma(au, uu, c);
                av = fma(av, uv, c);
                aw = fma(aw, uw, c);
            }
            ku[a] = au; kv[a] = av; kw[a] = aw;
        }
    } else {
        wg_taps<W>(P, fu, beta, ku);
        wg_taps<W>(P, fv, beta, kv);
        wg_taps<W>(P, fw, beta, kw);
    }
}
// the weight of plane k0 + a, a = k - k0 in 0 .. W - 1 (lane-dependent): a chain of selects, no indexed registers
template <int W>
__device__ __forceinline__ double wg_pick(const double (&kw)[W], int a)
{
    double r = 0.0;
#pragma unroll
    for (int t = 0; t < W; ++t) r = a == t ? kw[t] : r;
    return r;
}

// (in everything below "u" is the SLOW axis of the stored planes and "v" the fast one: the planes are v-major, so the
// host hands uvw's v as this code's u -- component 1 -- and u as its v)
constexpr int WG_CU = 1, WG_CV = 0;
// The w fold.  The image is real, so V(-u, -v, -w) = conj V(u, v, w): with w-stacking every visibility with w < 0 is
// evaluated (or, in the adjoint, gridded) at the mirrored point and conjugated.  The planes then cover [min |w|, max |w|]
// instead of [min w, max w] -- about half as many for an array whose baselines point either way (ducc0's wgridder
// treats w < 0 the same way).  Every kernel takes a row's sign from here, so they agree on it to the last bit; the
// products with +-1.0 are exact.
__device__ __forceinline__ double wg_fold_sign(const double *__restrict__ uvw_row, int do_w)
{
    return do_w && uvw_row[2] < 0.0 ? -1.0 : 1.0;
}
// first cell of a visibility's support along one axis, wrapped onto the grid: the sort key and the tile kernel must
// agree on it to the last bit, so both call this
__device__ __forceinline__ int wg_first_cell(double g, int W, int n)
{
    const double t = ceil(g - 0.5 * W);                     // |t| < 1e15: exact in double
    const double m = t - (double)n * floor(t / (double)n);
    int p = (int)m;
    p = p < 0 ? p + n : p;                                  // (rounding of t / n at multiples of n)
    return p >= n ? p - n : p;
}

}  // namespace
