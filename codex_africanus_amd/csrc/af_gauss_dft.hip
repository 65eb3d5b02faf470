// Gaussian (and point) sources WITHOUT direction-dependent terms:
//
//     V[r,nu] = sum_s  shape(r,s,nu) K(r,s,nu) X_s(nu)
//
// the reference's  phase_delay -> gaussian_shape -> einsum("srf,srf,sfij->srfij") -> predict_vis(source_coh)
// (africanus/rime/phase.py:28-61, africanus/model/shape/gaussian_shape.py:21-62,
// africanus/rime/examples/predict.py:107-134,525) as a direct transform whose phasor carries the envelope.  The
// direct-transform kernels of af_im_to_vis*.hip cannot do this -- their per-source operand is a (chan, corr) pixel,
// the shape depends on the row -- and until round 4 such calls ran through the fused beam kernel with an identity cube
// (the full 2 x 2 Jones algebra per (row, chan, source) plus a sampled beam of ones).
//
// lane = row; a block walks the sources for one tile of CT channels, accumulators in registers (CT x 4 complex).
// Per (row, source, tile): q = l u + m v + n w (phase_delay's clamped n), the anchor phasor and the channel-step phasor
// (quarter-turn polynomial, ~1e-16), and for an extended source the envelope
//     shape(nu) = exp(-a nu^2),  a = u1^2 + v1^2,  u1 = (u em - v el) er,  v1 = u el + v em      (gaussian_shape.py:45-60)
// which on a uniformly spaced tile nu_j = nu_0 + j d is a second-order PRODUCT recurrence:
//     e_0 = exp(-a nu_0^2),  r_0 = exp(-a (2 nu_0 d + d^2)),  c = exp(-2 a d^2):   e_(j+1) = e_j r_j,  r_(j+1) = r_j c
// (three exponentials per (row, source, tile) instead of one per channel; relative error ~ j ulp).  Per channel:
// 2 FMA (three-term phasor recurrence) + 2 products (envelope) + 2 (envelope x phasor) + 16 FMA (4 complex MACs).
// Point sources (major = minor = 0) skip the envelope: their shape factor is exactly 1.
// Non-uniform tiles (decided on the device, as in af_im_to_vis.hip): one sincos and one exponential per channel.
// Source coordinates, shape parameters and the tile's brightness matrices of a batch of GD_SB sources are staged in LDS
// by the whole block and read back by every lane from ONE address per instruction (LDS broadcasts; through scalar
// loads each source iteration waited out eight SMEM round trips: 159 ms at 1e6 x 64 x 1000 against 127 ms now; the
// identity-cube route through the beam kernel takes 212 ms).
// fp64-VALU bound.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "af_dft_mfma.h"
#include "af_fused_device.h"

namespace {

constexpr int GD_CT = 8, GD_ROWS = 256, GD_SB = 16;   // channels per tile, rows per block, sources per LDS batch

struct GaussWs {
    size_t lmn, gauss, tilef, flags, gauss_mfma, srcbad, mtilef, mflags, mfma, total;
};
// the MFMA-accumulator form (af_im_to_vis_mfma.hip, GAUSS) needs a 16-channel tile at least
bool gauss_mfma_eligible(int64_t nchan) { return nchan >= 14 && nchan / 32 + 1 <= 65535; }
GaussWs gauss_ws(int64_t nsrc, int64_t nchan)
{
    GaussWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    const int64_t ntile = (nchan + GD_CT - 1) / GD_CT;
    w.lmn = take((size_t)nsrc * 4 * sizeof(double));
    w.gauss = take((size_t)nsrc * 4 * sizeof(double));
    w.tilef = take((size_t)ntile * 2 * sizeof(double));
    w.flags = take(64 * sizeof(int));
    w.gauss_mfma = take((size_t)nsrc * 4 * sizeof(double));
    w.srcbad = take((size_t)nsrc * sizeof(int));
    w.mtilef = take(4 * sizeof(double));
    w.mflags = take(64 * sizeof(int));
    w.mfma = o;
    if (gauss_mfma_eligible(nchan)) o += af_dft_mfma_workspace_bytes(af_cdiv(nsrc > 0 ? nsrc : 1, 4) * 4, nchan, true, true);
    w.total = o;
    return w;
}

// The MFMA kernels evaluate a whole tile (tiles start at multiples of 64 channels) as nu[c0] + j d with ONE d for the band:
// mflags[0] = mflags[1] = every channel within 2 ulp of that; mtilef[1] = d in quarter turns per metre.  One block.
__global__ void gauss_band_prep(const double *__restrict__ freq, int64_t nchan, int sign, double *__restrict__ mtilef,
                                int *__restrict__ mflags)
{
    __shared__ int ok;
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();
    const double f0 = freq[0], df = nchan > 1 ? (freq[nchan - 1] - f0) / (double)(nchan - 1) : 0.0;
    bool good = isfinite(f0) && isfinite(df);
    for (int64_t j = threadIdx.x; j < nchan; j += blockDim.x) {
        const int64_t c0 = (j / 64) * 64;
        const double pred = fma((double)(j - c0), df, freq[c0]), f = freq[j];
        const double tol = 2.0 * 2.220446049250313e-16 * fmax(fabs(f), fabs(pred));
        if (!(fabs(f - pred) <= tol)) good = false;
    }
    if (!good) atomicAnd(&ok, 0);
    __syncthreads();
    if (threadIdx.x == 0) {
        mflags[0] = ok; mflags[1] = ok; mflags[2] = 0;
        mtilef[0] = 4.0 * (double)sign * f0 / AF_LIGHTSPEED;
        mtilef[1] = 4.0 * (double)sign * df / AF_LIGHTSPEED;
    }
}

// per tile: (nu_0, d) with d from the tile's own end points; flags[0] &= every channel within 2 ulp of that progression
__global__ void gauss_prep_freq(const double *__restrict__ freq, int64_t nchan, int64_t ntile, double *__restrict__ tilef,
                                int *__restrict__ flags)
{
    const int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= ntile) return;
    const int64_t c0 = tile * GD_CT;
    const int64_t n = nchan - c0 < GD_CT ? nchan - c0 : GD_CT;
    const double f0 = freq[c0], df = n > 1 ? (freq[c0 + n - 1] - f0) / (double)(n - 1) : 0.0;
    tilef[2 * tile] = f0;
    tilef[2 * tile + 1] = df;
    bool uniform = isfinite(f0) && isfinite(df);
    for (int64_t j = 0; j < n; ++j) {
        const double pred = fma((double)j, df, f0), f = freq[c0 + j];
        const double tol = 2.0 * 2.220446049250313e-16 * fabs(f);
        if (!(fabs(f - pred) <= tol)) uniform = false;
    }
    if (!uniform) atomicAnd(&flags[0], 0);
}

template <bool UNIFORM>
__global__ __launch_bounds__(GD_ROWS) void gauss_dft_kernel(const double *__restrict__ uvw, const double *__restrict__ lmn,
                                                           const double *__restrict__ gp, const double *__restrict__ freq,
                                                           const double *__restrict__ tilef, const int *__restrict__ flags,
                                                           const double2 *__restrict__ brightness, int nsrc, int64_t nrow,
                                                           int64_t nchan, double sign4_over_c, double2 *__restrict__ out,
                                                           const int *__restrict__ mflags)
{
    if (mflags != nullptr && mflags[0] == 1) return;   // the MFMA kernels own this band (gauss_band_prep)
    if ((flags[0] != 0) != UNIFORM) return;       // decided on the device by gauss_prep_freq
    const int64_t tile = blockIdx.y, c0 = tile * GD_CT;
    const int nt = (int)(nchan - c0 < GD_CT ? nchan - c0 : GD_CT);
    int64_t row = (int64_t)blockIdx.x * GD_ROWS + threadIdx.x;
    const bool live = row < nrow;
    if (!live) row = nrow - 1;
    const double u = uvw[3 * row], v = uvw[3 * row + 1], w = uvw[3 * row + 2];
    const double nu0 = tilef[2 * tile], dnu = tilef[2 * tile + 1];
    const double F0 = __dmul_rn(sign4_over_c, nu0), FD = __dmul_rn(sign4_over_c, dnu);   // quarter turns per metre
    double fq[GD_CT], f2[GD_CT];
    if constexpr (!UNIFORM) {
#pragma unroll
        for (int j = 0; j < GD_CT; ++j) {
            const double f = freq[c0 + (j < nt ? j : 0)];
            fq[j] = __dmul_rn(sign4_over_c, f);
            f2[j] = __dmul_rn(f, f);
        }
    }
    C2 acc[GD_CT][4];
#pragma unroll
    for (int j = 0; j < GD_CT; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[j][c].re = acc[j][c].im = 0.0;

    __shared__ double2 sX[GD_SB][GD_CT * 4];      // brightness of the batch's sources on this tile
    __shared__ double sL[GD_SB][4], sG[GD_SB][4]; // (l, m, n, 0) and (el, em, er, extended)
    for (int s0 = 0; s0 < nsrc; s0 += GD_SB) {
        const int nb = nsrc - s0 < GD_SB ? nsrc - s0 : GD_SB;
        __syncthreads();                          // the previous batch has been consumed
        for (int i = threadIdx.x; i < nb * GD_CT * 4; i += GD_ROWS) {
            const int sl = i / (GD_CT * 4), e = i - sl * (GD_CT * 4);
            sX[sl][e] = (e >> 2) < nt ? brightness[((int64_t)(s0 + sl) * nchan + c0) * 4 + e] : make_double2(0.0, 0.0);
        }
        for (int i = threadIdx.x; i < nb * 4; i += GD_ROWS) {
            sL[i >> 2][i & 3] = lmn[4 * (int64_t)s0 + i];
            sG[i >> 2][i & 3] = gp[4 * (int64_t)s0 + i];
        }
        __syncthreads();
#pragma unroll 1
        for (int sl = 0; sl < nb; ++sl) {
            const double l = sL[sl][0], m = sL[sl][1], n = sL[sl][2];
            const double gel = sG[sl][0], gem = sG[sl][1], ger = sG[sl][2];
            const bool extended = sG[sl][3] != 0.0;     // block-uniform
            const double q = fma(n, w, fma(m, v, __dmul_rn(l, u)));
            double a = 0.0;
            if (extended) {
                const double u1 = (u * gem - v * gel) * ger, v1 = u * gel + v * gem;
                a = u1 * u1 + v1 * v1;
            }
            const double2 *X = sX[sl];
            if constexpr (UNIFORM) {
                C2 y0, d, y1;
                sincos_quarter_turns<7>(__dmul_rn(q, F0), y0.re, y0.im);
                sincos_quarter_turns<7>(__dmul_rn(q, FD), d.re, d.im);
                y1 = cmul(y0, d);
                const double k2 = __dadd_rn(d.re, d.re);
                // (channels beyond the band's last -- a short last tile -- multiply zero pixels and are never stored)
                // one block-uniform branch per source, not one per channel: each arm is a single scheduling region
                auto channels = [&](auto with_envelope, double e, double r, double cc) {
                    constexpr bool ENV = decltype(with_envelope)::value;
#pragma unroll
                    for (int j = 0; j < GD_CT; ++j) {
                        C2 y;
                        if (j == 0) y = y0;
                        else if (j == 1) y = y1;
                        else {
                            y.re = fma(k2, y1.re, -y0.re);
                            y.im = fma(k2, y1.im, -y0.im);
                            y0 = y1; y1 = y;
                        }
                        C2 sy = y;
                        if constexpr (ENV) {
                            sy.re = __dmul_rn(y.re, e); sy.im = __dmul_rn(y.im, e);
                            e = __dmul_rn(e, r);
                            r = __dmul_rn(r, cc);
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const double2 x = X[j * 4 + c];
                            C2 xx;
                            xx.re = x.x; xx.im = x.y;
                            cmac(acc[j][c], sy, xx);
                        }
                    }
                };
                // (a falling band has r_0 > 1; beyond e^700 the envelope is 0 on the whole tile, and 0 x inf must not appear)
                if (extended)
                    channels(std::true_type{}, exp_neg(a * nu0 * nu0), exp_neg(fmax(a * (2.0 * nu0 * dnu + dnu * dnu), -700.0)),
                             exp_neg(2.0 * a * dnu * dnu));
                else
                    channels(std::false_type{}, 1.0, 1.0, 1.0);
            } else {
#pragma unroll
                for (int j = 0; j < GD_CT; ++j) {
                    C2 sy;
                    sincos_quarter_turns<7>(__dmul_rn(q, fq[j]), sy.re, sy.im);
                    if (extended) {
                        const double e = exp_neg(a * f2[j]);
                        sy.re = __dmul_rn(sy.re, e); sy.im = __dmul_rn(sy.im, e);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double2 x = X[j * 4 + c];
                        C2 xx;
                        xx.re = x.x; xx.im = x.y;
                        cmac(acc[j][c], sy, xx);
                    }
                }
            }
        }
    }
    if (!live) return;
    double2 *o = out + (row * nchan + c0) * 4;
#pragma unroll
    for (int j = 0; j < GD_CT; ++j)
        if (j < nt) {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[j * 4 + c] = make_double2(acc[j][c].re, acc[j][c].im);
        }
}

}  // namespace

AF_EXPORT size_t af_gauss_predict_workspace_bytes(int64_t nsrc, int64_t nchan)
{
    if (nsrc < 0 || nchan < 0) return 0;
    return gauss_ws(nsrc, nchan).total;
}

// Replaces the chain phase_delay (africanus/rime/phase.py:11-63) x gaussian shape
// (africanus/model/shape/gaussian_shape.py:11-62) x brightness summed over sources
// (africanus/rime/examples/predict.py:107-134 + predict_vis with source_coh only, africanus/rime/predict.py:229-246).
// lm (nsrc,2), uvw (nrow,3), frequency (nchan), brightness (nsrc,nchan,2,2) complex128, gauss_shape (nsrc,3) =
// (major, minor, orientation) [rad] or NULL (all point sources); out (nrow,nchan,2,2) complex128.  DEVICE pointers.
int af_chi2_launch(const double *model, const double *data, const double *weight, int64_t nrow, int64_t nchan, int64_t ncorr,
                   double *chi2_per_chan, const int *skip, hipStream_t st);   // af_chi2.hip

namespace {
int gauss_predict_impl(const double *lm, const double *uvw, const double *frequency, const double *brightness,
                       const double *gauss_shape, int64_t nsrc, int64_t nrow, int64_t nchan, int convention, double *out,
                       void *workspace, size_t workspace_bytes, void *stream, const AfDftChi2 *chi)
{
    AF_REQUIRE(convention == AF_CONVENTION_FOURIER || convention == AF_CONVENTION_CASA,
               "convention not in ('fourier', 'casa')");
    AF_REQUIRE(nsrc >= 0 && nrow >= 0 && nchan >= 0 && nsrc < (1LL << 31), "af_gauss_predict_c128: bad extents");
    hipStream_t st = af_stream(stream);
    if (nrow == 0 || nchan == 0) {
        if (chi && nchan > 0) AF_HIP(hipMemsetAsync(chi->chi2, 0, sizeof(double) * (size_t)nchan, st));
        return AF_OK;
    }
    AF_REQUIRE(out != nullptr, "af_gauss_predict_c128: out is NULL");
    if (nsrc == 0) {
        AF_HIP(hipMemsetAsync(out, 0, sizeof(double) * 8 * (size_t)(nrow * nchan), st));
        return chi ? af_chi2_launch(out, chi->data, chi->weight, nrow, nchan, 4, chi->chi2, nullptr, st) : AF_OK;
    }
    AF_REQUIRE(lm && uvw && frequency && brightness, "af_gauss_predict_c128: NULL array");
    const GaussWs W = gauss_ws(nsrc, nchan);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= W.total, "af_gauss_predict_c128: workspace too small (%zu < %zu)",
               workspace_bytes, W.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_gauss_predict_c128: workspace must be 256-byte aligned");
    char *ws = static_cast<char *>(workspace);
    double *lmn = reinterpret_cast<double *>(ws + W.lmn), *gp = reinterpret_cast<double *>(ws + W.gauss);
    double *tilef = reinterpret_cast<double *>(ws + W.tilef);
    int *flags = reinterpret_cast<int *>(ws + W.flags);
    const int64_t ntile = af_cdiv(nchan, GD_CT);
    AF_REQUIRE(ntile <= 65535, "af_gauss_predict_c128: more than %d channels", 65535 * GD_CT);
    hipLaunchKernelGGL(fused_prep_src, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st, lm, nsrc, lmn);
    AF_LAUNCH_CHECK();
    const double fwhm = 2.0 * sqrt(2.0 * log(2.0));  // gaussian_shape.py:23-25
    const double gs = (1.0 / fwhm) * sqrt(2.0) * 3.141592653589793 / AF_LIGHTSPEED;
    hipLaunchKernelGGL(fused_prep_gauss, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st, gauss_shape, nsrc, gs, gp);
    AF_LAUNCH_CHECK();
    AF_HIP(hipMemsetAsync(flags, 0, 64 * sizeof(int), st));
    AF_HIP(hipMemsetAsync(flags, 1, 1, st));          // flags[0] = 1: uniform until a tile says otherwise
    hipLaunchKernelGGL(gauss_prep_freq, dim3((unsigned)af_cdiv(ntile, 64)), dim3(64), 0, st, frequency, nchan, ntile, tilef,
                       flags);
    AF_LAUNCH_CHECK();
    const double s4c = 4.0 * (double)convention / AF_LIGHTSPEED;
    // The MFMA-accumulator form: the brightness matrices as the complex image of dft_mfma_kernel, the envelope in the
    // lane's phasor.  Whether it or the lane = row kernels below do the work is decided on the device (one channel
    // spacing for the band); AFHIP_GAUSS_MFMA=0 keeps the band on the lane = row kernels (A/B measurements).
    static const bool mfma_on = !(getenv("AFHIP_GAUSS_MFMA") && atoi(getenv("AFHIP_GAUSS_MFMA")) == 0);
    const bool mfma = mfma_on && gauss_mfma_eligible(nchan);
    const int *mflags = nullptr;
    if (mfma) {
        double *gpk = reinterpret_cast<double *>(ws + W.gauss_mfma), *mtilef = reinterpret_cast<double *>(ws + W.mtilef);
        int *srcbad = reinterpret_cast<int *>(ws + W.srcbad), *mf = reinterpret_cast<int *>(ws + W.mflags);
        // shape parameters per unit of the kernels' frequency (1/256 turns per metre): nu = F c / 256
        hipLaunchKernelGGL(fused_prep_gauss, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st, gauss_shape, nsrc,
                           gs * (AF_LIGHTSPEED / 256.0), gpk);
        AF_LAUNCH_CHECK();
        AF_HIP(hipMemsetAsync(srcbad, 0, (size_t)nsrc * sizeof(int), st));
        hipLaunchKernelGGL(gauss_band_prep, dim3(1), dim3(256), 0, st, frequency, nchan, convention, mtilef, mf);
        AF_LAUNCH_CHECK();
        if (chi) AF_HIP(hipMemsetAsync(chi->chi2, 0, sizeof(double) * (size_t)nchan, st));
        const int rc = af_gauss_mfma_run(brightness, gpk, uvw, frequency, lmn, srcbad, mtilef, mf, convention, out, nrow, nsrc,
                                         af_cdiv(nsrc, 4) * 4, nchan, ws + W.mfma, st, chi);
        if (rc != AF_OK) return rc;
        mflags = mf;
    }
    const dim3 grid((unsigned)af_cdiv(nrow, GD_ROWS), (unsigned)ntile);
    if (!mfma) af_prof_begin(st);
    hipLaunchKernelGGL(gauss_dft_kernel<true>, grid, dim3(GD_ROWS), 0, st, uvw, lmn, gp, frequency, tilef, flags,
                       reinterpret_cast<const double2 *>(brightness), (int)nsrc, nrow, nchan, s4c,
                       reinterpret_cast<double2 *>(out), mflags);
    if (!mfma) af_prof_end(st);
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gauss_dft_kernel<false>, grid, dim3(GD_ROWS), 0, st, uvw, lmn, gp, frequency, tilef, flags,
                       reinterpret_cast<const double2 *>(brightness), (int)nsrc, nrow, nchan, s4c,
                       reinterpret_cast<double2 *>(out), mflags);
    AF_LAUNCH_CHECK();
    // chi^2: summed in the MFMA kernels' epilogue when they own the band (device flags (1, 1, 0)), else by the separate pass
    if (chi) return af_chi2_launch(out, chi->data, chi->weight, nrow, nchan, 4, chi->chi2, mflags, st);
    return AF_OK;
}
}  // namespace

AF_EXPORT int af_gauss_predict_c128(const double *lm, const double *uvw, const double *frequency, const double *brightness,
                                    const double *gauss_shape, int64_t nsrc, int64_t nrow, int64_t nchan, int convention,
                                    double *out, void *workspace, size_t workspace_bytes, void *stream)
{
    return gauss_predict_impl(lm, uvw, frequency, brightness, gauss_shape, nsrc, nrow, nchan, convention, out, workspace,
                              workspace_bytes, stream, nullptr);
}

// The same predict and  chi2[nu] = sum_{row, corr} [weight] |data - out|^2  (af_chi2_c128's quantity) in one call: the step of
// the row-sharded predict.  On bands the MFMA-accumulator kernels own, chi^2 is summed in their epilogue
// (af_im_to_vis_chi2_f64's scheme); elsewhere the call falls back, on the device, to the separate pass.
AF_EXPORT int af_gauss_predict_chi2_c128(const double *lm, const double *uvw, const double *frequency, const double *brightness,
                                         const double *gauss_shape, int64_t nsrc, int64_t nrow, int64_t nchan, int convention,
                                         double *out, const double *data, const double *weight, double *chi2_per_chan,
                                         void *workspace, size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(nchan == 0 || chi2_per_chan != nullptr, "af_gauss_predict_chi2_c128: chi2_per_chan is NULL");
    AF_REQUIRE(data != nullptr || nrow == 0 || nchan == 0, "af_gauss_predict_chi2_c128: data is NULL");
    AfDftChi2 chi;
    chi.data = data; chi.weight = weight; chi.chi2 = chi2_per_chan;
    return gauss_predict_impl(lm, uvw, frequency, brightness, gauss_shape, nsrc, nrow, nchan, convention, out, workspace,
                              workspace_bytes, stream, &chi);
}
