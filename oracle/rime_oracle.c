/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the RIME predict hot path.
 *
 * Plain-C restatement of the reference's numba kernels (codex-africanus v0.4.4):
 *   africanus/rime/phase.py:28-61          -> orc_phase_delay_{f64,f32}
 *   africanus/rime/predict.py:56-373,574-617 -> orc_predict_vis_{f64,f32}
 *   africanus/dft/kernels.py:33-67         -> orc_im_to_vis_f64
 *   africanus/dft/kernels.py:104-146       -> orc_vis_to_im_f64
 *   africanus/rime/wsclean_predict.py:11-84 -> orc_wsclean_predict_f64
 *   africanus/rime/feeds.py:14-47           -> orc_feed_rotation_f64
 *   africanus/model/shape/gaussian_shape.py:21-62 -> orc_gaussian_shape_f64
 *   africanus/rime/fast_beam_cubes.py:10-54  -> orc_freq_grid_interp_{f64,f32}
 *   africanus/rime/fast_beam_cubes.py:57-240 -> orc_beam_cube_dde_{f64,f32}
 *   africanus/constants/consts.py:6-9      -> ORC_*_TWO_PI_OVER_C
 *
 * Parity pinned: tests/test_oracle_golden.py checks every function here against
 * golden vectors captured from the real reference (tests/golden/make_golden.py)
 * and against the reference tests' own known-answer values.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product package must never import it.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_EINVAL 1
#define ORC_ENOMEM 2
#define ORC_MAX_CORR 16

/* africanus/constants/consts.py:6-9: c = 2.99792458e8; 2*math.pi/c */
#define ORC_LIGHTSPEED 2.99792458e8
#define ORC_TWO_PI_OVER_C (2 * 3.141592653589793 / ORC_LIGHTSPEED)
#define ORC_MINUS_TWO_PI_OVER_C (-ORC_TWO_PI_OVER_C)

#define REAL double
#define SUF f64
#define SQRT sqrt
#define COS cos
#define SIN sin
#define FLOOR floor
#define HYPOT hypot
#include "rime_oracle_impl.h"
#undef REAL
#undef SUF
#undef SQRT
#undef COS
#undef SIN
#undef FLOOR
#undef HYPOT

#define REAL float
#define SUF f32
#define SQRT sqrtf
#define COS cosf
#define SIN sinf
#define FLOOR floorf
#define HYPOT hypotf
#include "rime_oracle_impl.h"
#undef REAL
#undef SUF
#undef SQRT
#undef COS
#undef SIN
#undef FLOOR
#undef HYPOT

/* ------------------------------------------------------------------------
 * im_to_vis: africanus/dft/kernels.py:33-67.
 *   constants are Python floats (float64) whatever the input dtype (:35-37);
 *   n = sqrt(1 - l^2 - m^2) - 1, NOT clamped -> NaN outside the unit disc (:54)
 *   real_phase = C*(l*u + m*v + n*w) (:57);  p = real_phase*nu*1j (:61)
 *   if image[s,nu,c]: vis[r,nu,c] += exp(p)*image[s,nu,c]   (:63-65)
 * numba's complex exp gives exp(0)*(cos y + i sin y) = (cos y, sin y) here.
 * image: (nsrc, nchan, ncorr) real (image_is_complex=0) or complex interleaved.
 * out: (nrow, nchan, ncorr) complex128 interleaved.  out_c64 != 0 restates the
 * reference's behaviour when the output array is complex64: every += rounds
 * the running sum to float (the array element is c64, the right-hand side c128).
 * sign: -1 'fourier' (minus_two_pi_over_c), +1 'casa' (:34-39).
 * Rows are independent, so an OpenMP build may split them; per-row arithmetic
 * and order are unchanged.
 * ---------------------------------------------------------------------- */
int orc_im_to_vis_f64(const double *image, int image_is_complex,
                      const double *uvw, const double *lm, const double *frequency,
                      int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr,
                      int sign, int out_c64, double *out)
{
    if (sign != 1 && sign != -1) return ORC_EINVAL;
    const double constant = sign < 0 ? ORC_MINUS_TWO_PI_OVER_C : ORC_TWO_PI_OVER_C;
    const int64_t istride = image_is_complex ? 2 : 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t r = 0; r < nrow; ++r) {
        double u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
        double *vr = out + 2 * r * nchan * ncorr;
        for (int64_t k = 0; k < 2 * nchan * ncorr; ++k) vr[k] = 0.0;
        for (int64_t s = 0; s < nsrc; ++s) {
            double l = lm[2 * s], m = lm[2 * s + 1];
            double n = sqrt(1.0 - l * l - m * m) - 1.0;
            double real_phase = constant * (l * u + m * v + n * w);
            for (int64_t nu = 0; nu < nchan; ++nu) {
                double p = real_phase * frequency[nu];
                double cp = 0.0, sp = 0.0;
                int have_exp = 0;
                for (int64_t c = 0; c < ncorr; ++c) {
                    const double *px = image + ((s * nchan + nu) * ncorr + c) * istride;
                    double ire = px[0], iim = image_is_complex ? px[1] : 0.0;
                    /* Python truthiness: nonzero (NaN is truthy) */
                    if (ire == 0.0 && iim == 0.0) continue;
                    if (!have_exp) { cp = cos(p); sp = sin(p); have_exp = 1; }
                    /* exp(p) * image: full complex multiply, real image widened to (x + 0j) */
                    double tre = cp * ire - sp * iim;
                    double tim = cp * iim + sp * ire;
                    double *o = vr + 2 * (nu * ncorr + c);
                    if (out_c64) {
                        o[0] = (double)(float)(o[0] + tre);
                        o[1] = (double)(float)(o[1] + tim);
                    } else {
                        o[0] += tre;
                        o[1] += tim;
                    }
                }
            }
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------
 * vis_to_im: africanus/dft/kernels.py:104-146 (adjoint of im_to_vis).
 *   constant: 'fourier' -> +two_pi_over_c, 'casa' -> minus (:113-118: the opposite of im_to_vis)
 *   n = sqrt(1 - l^2 - m^2) - 1 unclamped (:125)
 *   for s: for r: real_phase = C*(l*u + m*v + n*w) (:131); for nu: p = real_phase*nu (:135);
 *     if any(flags[r,nu]): continue (:139-140)
 *     for c: im[s,nu,c] += cos(p)*vis.re - sin(p)*vis.im   (:142-146)
 * vis: (nrow, nchan, ncorr) complex128 interleaved (a real vis has zero imaginary parts);
 * flags: (nrow, nchan, ncorr) bytes; out: (nsrc, nchan, ncorr) float64.
 * sign: -1 'fourier', +1 'casa' (same labels as the other entry points).
 * Sources are independent, so an OpenMP build may split them.
 * ---------------------------------------------------------------------- */
int orc_vis_to_im_f64(const double *vis, const double *uvw, const double *lm, const double *frequency,
                      const unsigned char *flags, int64_t nsrc, int64_t nrow, int64_t nchan,
                      int64_t ncorr, int sign, double *out)
{
    if (sign != 1 && sign != -1) return ORC_EINVAL;
    const double constant = sign < 0 ? ORC_TWO_PI_OVER_C : ORC_MINUS_TWO_PI_OVER_C;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t s = 0; s < nsrc; ++s) {
        double l = lm[2 * s], m = lm[2 * s + 1];
        double n = sqrt(1.0 - l * l - m * m) - 1.0;
        double *im = out + s * nchan * ncorr;
        for (int64_t k = 0; k < nchan * ncorr; ++k) im[k] = 0.0;
        for (int64_t r = 0; r < nrow; ++r) {
            double u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
            double real_phase = constant * (l * u + m * v + n * w);
            for (int64_t nu = 0; nu < nchan; ++nu) {
                double p = real_phase * frequency[nu];
                const unsigned char *fl = flags + (r * nchan + nu) * ncorr;
                int flagged = 0;
                for (int64_t c = 0; c < ncorr; ++c) flagged |= fl[c];
                if (flagged) continue;
                double cp = cos(p), sp = sin(p);
                const double *vv = vis + 2 * (r * nchan + nu) * ncorr;
                for (int64_t c = 0; c < ncorr; ++c)
                    im[nu * ncorr + c] += cp * vv[2 * c] - sp * vv[2 * c + 1];
            }
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------
 * wsclean_predict_main: africanus/rime/wsclean_predict.py:11-84.
 *   fwhm = 2 sqrt(2 ln 2); gauss_scale = (1/fwhm) sqrt(2) pi / c (:13-15)
 *   n = sqrt(1 - l^2 - m^2) - 1 (:32); real_phase = two_pi_over_c*(u*l + v*m + w*n) (:40,61): CASA sign
 *   POINT:    vis[r,f] += cos(p)*spec + i sin(p)*spec                       (:42-47)
 *   GAUSSIAN: el = emaj sin(angle), em = emaj cos(angle), er = emin/(emaj or 1) (:49-54)
 *             u1 = (u*em - v*el)*er, v1 = u*el + v*em (:64-65)
 *             shape = exp(-(fu1^2 + fv1^2)), fu1 = u1*scaled_freq[f] (:73-75); re,im *= shape (:76-77)
 * is_gaussian: (nsrc) bytes; gauss_shape (nsrc,3); spectrum (nsrc,nchan); out (nrow,nchan) complex128.
 * ---------------------------------------------------------------------- */
/* numba's float ** int lowering: exponentiation by squaring (numba/cpython/numbers.py int_power_impl) */
static double orc_ipow(double a, int64_t b)
{
    double r = 1.0;
    int invert = b < 0;
    int64_t e = invert ? -b : b;
    if (e > 0x10000) return pow(a, (double)b);
    while (e != 0) {
        if (e & 1) r *= a;
        e >>= 1;
        a *= a;
    }
    return invert ? 1.0 / r : r;
}

/* spectra: africanus/model/wsclean/spec_model.py:70-122.
 *   log_poly:  I * exp(sum_c coeffs[c] * log(nu/rf)^(c+1))   (:101-112)
 *   ordinary:  I + sum_c coeffs[c] * (nu/rf - 1)^(c+1)       (:113-124) */
int orc_spectra_f64(const double *I, const double *coeffs, const unsigned char *log_poly, const double *ref_freq,
                    const double *frequency, int64_t nsrc, int64_t ncoeffs, int64_t nchan, double *out)
{
    for (int64_t s = 0; s < nsrc; ++s) {
        double rf = ref_freq[s];
        if (log_poly[s]) {
            for (int64_t f = 0; f < nchan; ++f) {
                double nu = frequency[f];
                double acc = 0.0;
                for (int64_t c = 0; c < ncoeffs; ++c) acc += coeffs[s * ncoeffs + c] * orc_ipow(log(nu / rf), c + 1);
                out[s * nchan + f] = I[s] * exp(acc);
            }
        } else {
            for (int64_t f = 0; f < nchan; ++f) {
                double nu = frequency[f];
                double acc = I[s];
                for (int64_t c = 0; c < ncoeffs; ++c) {
                    double term = coeffs[s * ncoeffs + c];
                    term *= orc_ipow(nu / rf - 1.0, c + 1);
                    acc += term;
                }
                out[s * nchan + f] = acc;
            }
        }
    }
    return ORC_OK;
}

int orc_wsclean_predict_f64(const double *uvw, const double *lm, const unsigned char *is_gaussian,
                            const double *gauss_shape, const double *frequency, const double *spectrum,
                            int64_t nsrc, int64_t nrow, int64_t nchan, double *out)
{
    const double fwhm = 2.0 * sqrt(2.0 * log(2.0));
    const double fwhminv = 1.0 / fwhm;
    const double gauss_scale = fwhminv * sqrt(2.0) * 3.141592653589793 / ORC_LIGHTSPEED;
    double *scaled_freq = (double *)malloc(sizeof(double) * (size_t)(nchan > 0 ? nchan : 1));
    if (!scaled_freq) return ORC_ENOMEM;
    for (int64_t f = 0; f < nchan; ++f) scaled_freq[f] = frequency[f] * gauss_scale;
    memset(out, 0, sizeof(double) * 2 * (size_t)(nrow * nchan));
    for (int64_t s = 0; s < nsrc; ++s) {
        double l = lm[2 * s], m = lm[2 * s + 1];
        double n = sqrt(1.0 - l * l - m * m) - 1.0;
        if (!is_gaussian[s]) {
            for (int64_t r = 0; r < nrow; ++r) {
                double u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
                double real_phase = ORC_TWO_PI_OVER_C * (u * l + v * m + w * n);
                for (int64_t f = 0; f < nchan; ++f) {
                    double p = real_phase * frequency[f];
                    double re = cos(p) * spectrum[s * nchan + f];
                    double im = sin(p) * spectrum[s * nchan + f];
                    out[2 * (r * nchan + f)] += re;
                    out[2 * (r * nchan + f) + 1] += im;
                }
            }
        } else {
            double emaj = gauss_shape[3 * s], emin = gauss_shape[3 * s + 1], angle = gauss_shape[3 * s + 2];
            double el = emaj * sin(angle), em = emaj * cos(angle);
            double er = emin / (emaj == 0.0 ? 1.0 : emaj);
            for (int64_t r = 0; r < nrow; ++r) {
                double u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
                double real_phase = ORC_TWO_PI_OVER_C * (u * l + v * m + w * n);
                double u1 = (u * em - v * el) * er;
                double v1 = u * el + v * em;
                for (int64_t f = 0; f < nchan; ++f) {
                    double p = real_phase * frequency[f];
                    double re = cos(p) * spectrum[s * nchan + f];
                    double im = sin(p) * spectrum[s * nchan + f];
                    double fu1 = u1 * scaled_freq[f], fv1 = v1 * scaled_freq[f];
                    double shape = exp(-(fu1 * fu1 + fv1 * fv1));
                    re *= shape;
                    im *= shape;
                    out[2 * (r * nchan + f)] += re;
                    out[2 * (r * nchan + f) + 1] += im;
                }
            }
        }
    }
    free(scaled_freq);
    return ORC_OK;
}

/* ------------------------------------------------------------------------
 * feed_rotation: africanus/rime/feeds.py:14-47.  feed_type 0 = linear [[c, s], [-s, c]] (:21-32),
 * 1 = circular diag(e^{-i pa}, e^{+i pa}) (:35-45).  out (n,2,2) complex128.
 * ---------------------------------------------------------------------- */
int orc_feed_rotation_f64(const double *pa, int64_t n, int feed_type, double *out)
{
    if (feed_type != 0 && feed_type != 1) return ORC_EINVAL;
    for (int64_t i = 0; i < n; ++i) {
        const double c = cos(pa[i]), s = sin(pa[i]);
        double *o = out + 8 * i;
        if (feed_type == 0) {
            o[0] = c;  o[1] = 0.0; o[2] = s; o[3] = 0.0;
            o[4] = -s; o[5] = 0.0; o[6] = c; o[7] = 0.0;
        } else {
            o[0] = c;   o[1] = -s;  o[2] = 0.0; o[3] = 0.0;
            o[4] = 0.0; o[5] = 0.0; o[6] = c;   o[7] = s;
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------
 * gaussian shape: africanus/model/shape/gaussian_shape.py:21-62.
 *   shape[s,r,f] = exp(-(fu1^2 + fv1^2)), fu1 = u1*scaled_freq[f], u1 = (u*em - v*el)*er, v1 = u*el + v*em,
 *   el = emaj sin(angle), em = emaj cos(angle), er = emin / (emaj or 1)   (:45-60)
 * out (nsrc,nrow,nchan) float64.
 * ---------------------------------------------------------------------- */
int orc_gaussian_shape_f64(const double *uvw, const double *frequency, const double *shape_params, int64_t nsrc,
                           int64_t nrow, int64_t nchan, double *out)
{
    const double fwhm = 2.0 * sqrt(2.0 * log(2.0));
    const double fwhminv = 1.0 / fwhm;
    const double gauss_scale = fwhminv * sqrt(2.0) * 3.141592653589793 / ORC_LIGHTSPEED;
    double *scaled_freq = (double *)malloc(sizeof(double) * (size_t)(nchan > 0 ? nchan : 1));
    if (!scaled_freq) return ORC_ENOMEM;
    for (int64_t f = 0; f < nchan; ++f) scaled_freq[f] = frequency[f] * gauss_scale;
    for (int64_t s = 0; s < nsrc; ++s) {
        const double emaj = shape_params[3 * s], emin = shape_params[3 * s + 1], angle = shape_params[3 * s + 2];
        const double el = emaj * sin(angle), em = emaj * cos(angle);
        const double er = emin / (emaj == 0.0 ? 1.0 : emaj);
        for (int64_t r = 0; r < nrow; ++r) {
            const double u = uvw[3 * r], v = uvw[3 * r + 1];
            const double u1 = (u * em - v * el) * er;
            const double v1 = u * el + v * em;
            for (int64_t f = 0; f < nchan; ++f) {
                const double fu1 = u1 * scaled_freq[f], fv1 = v1 * scaled_freq[f];
                out[(s * nrow + r) * nchan + f] = exp(-(fu1 * fu1 + fv1 * fv1));
            }
        }
    }
    free(scaled_freq);
    return ORC_OK;
}

/* ------------------------------------------------------------------------
 * spectral_model: africanus/model/spectral/spec_model.py:102-236.  stokes (nsrc,npol), spi (nsrc,nspi,npol),
 * base (npol) ints: 0 std  I * prod_i (nu/nu0)^spi_i (:173-185), 1 log  I * exp(sum_i spi_i ln(nu/nu0)^(i+1))
 * (:187-199), 2 log10  I * 10^(sum_i spi_i log10(nu/nu0)^(i+1)) (:201-213); out (nsrc,nchan,npol).
 * float ** float -> pow(); float ** int -> exponentiation by squaring (orc_ipow); 10 ** float -> pow(10, x).
 * ---------------------------------------------------------------------- */
int orc_spectral_model_f64(const double *stokes, const double *spi, const double *ref_freq, const double *frequency,
                           const int *base, int64_t nsrc, int64_t nspi, int64_t npol, int64_t nchan, double *out)
{
    for (int64_t p = 0; p < npol; ++p) {
        if (base[p] < 0 || base[p] > 2) return ORC_EINVAL;
        for (int64_t s = 0; s < nsrc; ++s) {
            const double rf = ref_freq[s];
            for (int64_t f = 0; f < nchan; ++f) {
                double v;
                if (base[p] == 0) {
                    const double ratio = frequency[f] / rf;
                    v = stokes[s * npol + p];
                    for (int64_t si = 0; si < nspi; ++si) v *= pow(ratio, spi[(s * nspi + si) * npol + p]);
                } else {
                    const double lr = base[p] == 1 ? log(frequency[f] / rf) : log10(frequency[f] / rf);
                    double acc = 0.0;
                    for (int64_t si = 0; si < nspi; ++si) acc += spi[(s * nspi + si) * npol + p] * orc_ipow(lr, si + 1);
                    v = stokes[s * npol + p] * (base[p] == 1 ? exp(acc) : pow(10.0, acc));
                }
                out[(s * nchan + f) * npol + p] = v;
            }
        }
    }
    return ORC_OK;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
