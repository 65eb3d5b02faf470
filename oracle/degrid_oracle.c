/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the convolutional degridder (BASELINE
 * configs[4]).  Plain-C restatement of africanus/gridding/perleypolyhedron:
 *   degridder.py:15-76 (degridder_row_kernel), :79-175 (degridder)
 *   policies/convolution_policies.py:188-260 (packed gather), :263-323 (unpacked gather)
 *   policies/baseline_transform_policies.py:56-81 (wlinapprox), policies/phase_transform_policies.py:9-35
 *   policies/stokes_conversion_policies.py:8-137 (stokes2corr: the 16 policies as per-correlation factors)
 * The reference compiles with fastmath=True: its own results are reproducible only to rounding, so parity
 * against it is to a relative tolerance (tests state it), not bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { double re, im; } cplx;

/* np.round: round half to even; int(): truncation toward zero */
static inline int64_t py_round_i(double x) { return (int64_t)nearbyint(x); }

/* coef (ncorr,2) = per-correlation complex factor of the policy; baseline_policy 0 None, 1 wlinapprox;
 * phase_policy 0 None, 1 phase_rotate; conv_policy 0 packed gather, 1 unpacked gather.
 * uvw (nrow,3); grid (nband,npix,npix) complex128; wavelengths, chanmap (nchan); kernel (os*(W+2)) real;
 * out (nrow,nchan,ncorr) complex128. */
int orc_degridder_c128(const double *uvw, const double *grid, const double *wavelengths, const int64_t *chanmap,
                       double cell, const double *image_centre, const double *phase_centre, const double *kernel,
                       int64_t W, int64_t os, int baseline_policy, int phase_policy, const double *coef, int ncorr,
                       int conv_policy, int64_t nrow, int64_t nchan, int64_t npix, double *out)
{
    const cplx *g = (const cplx *)grid;
    cplx *o = (cplx *)out;
    const double ra0 = phase_centre[0], dec0 = phase_centre[1], ra = image_centre[0], dec = image_centre[1];
    const double scale_factor = npix * cell / 3600.0 * 3.141592653589793 / 180.0;
    const int64_t klen = os * (W + 2);
    /* constants of the two transforms (baseline_transform_policies.py:67-79, phase_transform_policies.py:21-32) */
    const double d_ra = ra - ra0, c_d_ra = cos(d_ra), s_d_ra = sin(d_ra);
    const double c_new = cos(dec), c_old = cos(dec0), s_new = sin(dec), s_old = sin(dec0);
    const double li0 = c_new * s_d_ra, mi0 = s_new * c_old - c_new * s_old * c_d_ra, ni0 = s_new * s_old + c_new * c_old * c_d_ra;
    const double ll = c_new * s_d_ra, mm = s_new * c_old - c_new * s_old * c_d_ra;
    const double nn = -(1 - sqrt(1 - ll * ll - mm * mm));
    memset(out, 0, sizeof(double) * 2 * (size_t)(nrow * nchan * ncorr));
    for (int64_t r = 0; r < nrow; ++r) {
        double u = uvw[3 * r], v = uvw[3 * r + 1];
        const double w = uvw[3 * r + 2];
        if (baseline_policy == 1) {
            u = u - w * li0 / ni0;
            v = v - w * mi0 / ni0;
        }
        for (int64_t c = 0; c < nchan; ++c) {
            const double su = u * scale_factor / wavelengths[c], sv = v * scale_factor / wavelengths[c];
            const cplx *gb = g + chanmap[c] * npix * npix;
            const double offset_u = su + npix / 2, offset_v = sv + npix / 2;
            const int64_t disc_u = py_round_i(offset_u), disc_v = py_round_i(offset_v);
            const int64_t frac_u = (int64_t)((-offset_u + disc_u) * os), frac_v = (int64_t)((-offset_v + disc_v) * os);
            const int64_t fo_u = frac_u < 0 ? 0 : 1, fo_v = frac_v < 0 ? 0 : 1;
            cplx acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
            double cw = 0.0;
            for (int64_t tv = 0; tv < W; ++tv) {
                int64_t iv = conv_policy == 0 ? tv + fo_v + frac_v * (W + 2) : (tv + 1) * os + frac_v;
                if (iv < 0) iv += klen;                       /* numpy negative indexing */
                const double conv_v = kernel[iv];
                const int64_t gv = disc_v + tv - W / 2;
                for (int64_t tu = 0; tu < W; ++tu) {
                    int64_t iu = conv_policy == 0 ? tu + fo_u + frac_u * (W + 2) : (tu + 1) * os + frac_u;
                    if (iu < 0) iu += klen;
                    const double conv_u = kernel[iu];
                    const int64_t gu = disc_u + tu - W / 2;
                    if (gv >= 0 && gv < npix && gu >= 0 && gu < npix) {
                        const cplx x = gb[gv * npix + gu];
                        const cplx t = { x.re * conv_v * conv_u, x.im * conv_v * conv_u };
                        for (int k = 0; k < ncorr; ++k) {
                            acc[k].re += coef[2 * k] * t.re - coef[2 * k + 1] * t.im;
                            acc[k].im += coef[2 * k] * t.im + coef[2 * k + 1] * t.re;
                        }
                        cw += conv_v * conv_u;
                    }
                }
            }
            double pr = 1.0, pi_ = 0.0;
            if (phase_policy == 1) {
                const double x = -1.0 * 2 * 3.141592653589793 * (u * ll + v * mm + w * nn) / wavelengths[c];
                pr = cos(x); pi_ = sin(x);
            }
            for (int k = 0; k < ncorr; ++k) {
                const double ar = acc[k].re / (cw + 1.0e-8), ai = acc[k].im / (cw + 1.0e-8);
                o[(r * nchan + c) * ncorr + k].re = ar * pr - ai * pi_;
                o[(r * nchan + c) * ncorr + k].im = ar * pi_ + ai * pr;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------
 * gridder: africanus/gridding/perleypolyhedron/gridder.py:12-117 with the scatter policies
 * (policies/convolution_policies.py:6-185: unpacked 0, packed 1, nearest neighbour 2) and corr2stokes given as
 * per-correlation complex factors (policies/stokes_conversion_policies.py:143-180).  The facet phase rotation is
 * applied with phasesign +1 before gridding (:81-91); the tap weights of a visibility are summed over ALL taps,
 * on or off the grid (:57-67), per band; do_normalize divides every band by its weight sum + 1e-8 (:114-116).
 * vis (nrow,nchan,ncorr) complex128; out grid (nband,npix,npix) complex128.
 * ---------------------------------------------------------------------- */
int orc_gridder_c128(const double *uvw, const double *vis, const double *wavelengths, const int64_t *chanmap,
                     int64_t npix, double cell, const double *image_centre, const double *phase_centre,
                     const double *kernel, int64_t W, int64_t os, int phase_policy, const double *coef, int ncorr,
                     int conv_policy, int do_normalize, int64_t nrow, int64_t nchan, int64_t nband, double *out,
                     double *wt_out)
{
    const cplx *vs = (const cplx *)vis;
    cplx *g = (cplx *)out;
    const double ra0 = phase_centre[0], dec0 = phase_centre[1], ra = image_centre[0], dec = image_centre[1];
    const double scale_factor = npix * cell / 3600.0 * 3.141592653589793 / 180.0;
    const int64_t klen = os * (W + 2);
    const double d_ra = ra - ra0, c_d_ra = cos(d_ra), s_d_ra = sin(d_ra);
    const double c_new = cos(dec), c_old = cos(dec0), s_new = sin(dec), s_old = sin(dec0);
    const double ll = c_new * s_d_ra, mm = s_new * c_old - c_new * s_old * c_d_ra;
    const double nn = -(1 - sqrt(1 - ll * ll - mm * mm));
    memset(out, 0, sizeof(double) * 2 * (size_t)(nband * npix * npix));
    for (int64_t b = 0; b < nband; ++b) wt_out[b] = 0.0;
    for (int64_t r = 0; r < nrow; ++r) {
        const double u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
        for (int64_t c = 0; c < nchan; ++c) {
            double pr = 1.0, pi_ = 0.0;
            if (phase_policy == 1) {
                const double x = 1.0 * 2 * 3.141592653589793 * (u * ll + v * mm + w * nn) / wavelengths[c];
                pr = cos(x); pi_ = sin(x);
            }
            cplx s = {0.0, 0.0};
            for (int k = 0; k < ncorr; ++k) {
                const cplx x = vs[(r * nchan + c) * ncorr + k];
                const double xr = x.re * pr - x.im * pi_, xi = x.re * pi_ + x.im * pr;   /* vis *= phase */
                s.re += coef[2 * k] * xr - coef[2 * k + 1] * xi;
                s.im += coef[2 * k] * xi + coef[2 * k + 1] * xr;
            }
            const double su = u * scale_factor / wavelengths[c], sv = v * scale_factor / wavelengths[c];
            cplx *gb = g + chanmap[c] * npix * npix;
            const double offset_u = su + npix / 2, offset_v = sv + npix / 2;
            const int64_t disc_u = py_round_i(offset_u), disc_v = py_round_i(offset_v);
            if (conv_policy == 2) {
                if (disc_u >= 0 && disc_u < npix && disc_v >= 0 && disc_v < npix) {
                    gb[disc_v * npix + disc_u].re += s.re;
                    gb[disc_v * npix + disc_u].im += s.im;
                }
                wt_out[chanmap[c]] += 1.0;
                continue;
            }
            const int64_t frac_u = (int64_t)((-offset_u + disc_u) * os), frac_v = (int64_t)((-offset_v + disc_v) * os);
            const int64_t fo_u = frac_u < 0 ? 0 : 1, fo_v = frac_v < 0 ? 0 : 1;
            double cw = 0.0;
            for (int64_t tv = 0; tv < W; ++tv) {
                int64_t iv = conv_policy == 1 ? tv + fo_v + frac_v * (W + 2) : (tv + 1) * os + frac_v;
                if (iv < 0) iv += klen;
                const double conv_v = kernel[iv];
                const int64_t gv = disc_v + tv - W / 2;
                for (int64_t tu = 0; tu < W; ++tu) {
                    int64_t iu = conv_policy == 1 ? tu + fo_u + frac_u * (W + 2) : (tu + 1) * os + frac_u;
                    if (iu < 0) iu += klen;
                    const double conv_u = kernel[iu];
                    const int64_t gu = disc_u + tu - W / 2;
                    if (gv >= 0 && gv < npix && gu >= 0 && gu < npix) {
                        gb[gv * npix + gu].re += conv_v * conv_u * s.re;
                        gb[gv * npix + gu].im += conv_v * conv_u * s.im;
                    }
                    cw += conv_v * conv_u;
                }
            }
            wt_out[chanmap[c]] += cw;
        }
    }
    if (do_normalize)
        for (int64_t b = 0; b < nband; ++b)
            for (int64_t i = 0; i < npix * npix; ++i) {
                g[b * npix * npix + i].re /= wt_out[b] + 1.0e-8;
                g[b * npix * npix + i].im /= wt_out[b] + 1.0e-8;
            }
    return 0;
}
