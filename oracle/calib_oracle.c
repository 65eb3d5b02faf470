/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the calibration consumers of the predict path
 * (SURVEY 8(f) rank 4).  Plain-C restatement of the reference's numba loops, complex128:
 *   africanus/calibration/utils/corrupt_vis.py:10-101   -> orc_corrupt_vis_c128
 *   africanus/calibration/utils/residual_vis.py:11-119  -> orc_residual_vis_c128
 *   africanus/calibration/utils/correct_vis.py:10-115   -> orc_correct_vis_c128
 * mode (africanus/calibration/utils/utils.py:6-8,11-45): 0 DIAG_DIAG (jones and vis carry ncorr values),
 * 1 DIAG (jones (2,), vis (2,2)), 2 FULL (jones (2,2), vis (2,2)).
 * Arithmetic as numba emits it (compile with -ffp-contract=off): complex multiply = four rounded
 * products, one subtract, one add; complex division = CPython _Py_c_quot as numba lowers it (numba/cpython/numbers.py complex_div).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double re, im; } cplx;
static inline cplx cmul(cplx a, cplx b) { cplx z = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; return z; }
static inline cplx cadd(cplx a, cplx b) { cplx z = { a.re + b.re, a.im + b.im }; return z; }
static inline cplx csub(cplx a, cplx b) { cplx z = { a.re - b.re, a.im - b.im }; return z; }
static inline cplx cconj(cplx a) { cplx z = { a.re, -a.im }; return z; }
static inline cplx cneg(cplx a) { cplx z = { -a.re, -a.im }; return z; }
/* numba/cpython/numbers.py complex_div: CPython _Py_c_quot */
static inline cplx cdiv(cplx a, cplx b)
{
    cplx z;
    if (fabs(b.re) >= fabs(b.im)) {
        if (b.re == 0.0) { z.re = z.im = NAN; return z; }
        const double rat = b.im / b.re, den = b.re + b.im * rat;
        z.re = (a.re + a.im * rat) / den;
        z.im = (a.im - a.re * rat) / den;
        return z;
    }
    const double rat = b.re / b.im, den = b.re * rat + b.im;
    z.re = (a.re * rat + a.im) / den;
    z.im = (a.im * rat - a.re) / den;
    return z;
}

/* sum over directions of a1j[s] . model[s] . a2j[s]^H into acc (2x2 row-major or ncorr diagonal values);
 * sign +1: acc += term (corrupt_vis.py:10-56), -1: acc -= term (residual_vis.py:11-60) */
static void jones_term(int mode, int ncorr, int64_t ndir, const cplx *a1j, const cplx *model, const cplx *a2j, int sign,
                       cplx *acc)
{
    for (int64_t s = 0; s < ndir; ++s) {
        if (mode == 0) {
            for (int c = 0; c < ncorr; ++c) {
                cplx t = cmul(cmul(a1j[s * ncorr + c], model[s * ncorr + c]), cconj(a2j[s * ncorr + c]));
                acc[c] = sign > 0 ? cadd(acc[c], t) : csub(acc[c], t);
            }
        } else if (mode == 1) {
            const cplx *g = a1j + 2 * s, *h = a2j + 2 * s, *m = model + 4 * s;
            const int gi[4] = {0, 0, 1, 1}, hi[4] = {0, 1, 0, 1};
            for (int c = 0; c < 4; ++c) {
                cplx t = cmul(cmul(g[gi[c]], m[c]), cconj(h[hi[c]]));
                acc[c] = sign > 0 ? cadd(acc[c], t) : csub(acc[c], t);
            }
        } else {
            const cplx *g = a1j + 4 * s, *h = a2j + 4 * s, *m = model + 4 * s;
            /* tmp = conj(a2j.T): tmp[i][j] = conj(h[j][i]) */
            const cplx tmp00 = cconj(h[0]), tmp01 = cconj(h[2]), tmp10 = cconj(h[1]), tmp11 = cconj(h[3]);
            for (int i = 0; i < 2; ++i) {
                const cplx t1 = cmul(g[2 * i], m[0]), t2 = cmul(g[2 * i + 1], m[2]);
                const cplx t3 = cmul(g[2 * i], m[1]), t4 = cmul(g[2 * i + 1], m[3]);
                const cplx o0 = cadd(cadd(cadd(cmul(t1, tmp00), cmul(t2, tmp00)), cmul(t3, tmp10)), cmul(t4, tmp10));
                const cplx o1 = cadd(cadd(cadd(cmul(t1, tmp01), cmul(t2, tmp01)), cmul(t3, tmp11)), cmul(t4, tmp11));
                acc[2 * i] = sign > 0 ? cadd(acc[2 * i], o0) : csub(acc[2 * i], o0);
                acc[2 * i + 1] = sign > 0 ? cadd(acc[2 * i + 1], o1) : csub(acc[2 * i + 1], o1);
            }
        }
    }
}

/* compute_and_corrupt_vis.py:11-71: source_vis = model[s] * exp(1j * real_phase) / n per direction, then the
 * jones_mul body; model (ndir, V) of the row's time bin, lm (ndir, 2) of the time bin */
static void jones_term_computed(int mode, int ncorr, int64_t ndir, const cplx *a1j, const cplx *model, const cplx *a2j,
                                const double *uvw, double freq, const double *lm, cplx *acc)
{
    const double m2pioc = -2 * 3.141592653589793 / 2.99792458e8;
    const int V = mode == 0 ? ncorr : 4;
    for (int64_t s = 0; s < ndir; ++s) {
        const double l = lm[2 * s], m = lm[2 * s + 1];
        const double n = sqrt(1 - l * l - m * m);
        const double real_phase = m2pioc * freq * (uvw[0] * l + uvw[1] * m + uvw[2] * (n - 1));
        /* 1.0j * real_phase = (0*x - 1*0, 0*0 + 1*x); np.exp of it = exp(re) * (cos, sin) */
        const double ere = exp(0.0 * real_phase - 1.0 * 0.0), eim = 0.0 * 0.0 + 1.0 * real_phase;
        const cplx ph = { ere * cos(eim), ere * sin(eim) };
        const cplx nn = { n, 0.0 };
        cplx sv[4];
        for (int c = 0; c < V; ++c) sv[c] = cdiv(cmul(model[s * V + c], ph), nn);
        jones_term(mode, ncorr, 1, a1j + s * (mode == 0 ? ncorr : (mode == 1 ? 2 : 4)), sv,
                   a2j + s * (mode == 0 ? ncorr : (mode == 1 ? 2 : 4)), +1, acc);
    }
}

static int jones_elems(int mode, int ncorr) { return mode == 0 ? ncorr : (mode == 1 ? 2 : 4); }
static int vis_elems(int mode, int ncorr) { return mode == 0 ? ncorr : 4; }

/* jones (ntime,nant,nchan,ndir,J), model (nrow,nchan,ndir,V), out (nrow,nchan,V); tbin_idx must start at 0 */
int orc_corrupt_vis_c128(const int64_t *tbin_idx, const int64_t *tbin_counts, int64_t ntime, const int64_t *ant1,
                         const int64_t *ant2, const double *jones, const double *model, int64_t nrow, int64_t nant,
                         int64_t nchan, int64_t ndir, int mode, int ncorr, double *out)
{
    const int J = jones_elems(mode, ncorr), V = vis_elems(mode, ncorr);
    const cplx *jn = (const cplx *)jones, *md = (const cplx *)model;
    cplx *o = (cplx *)out;
    memset(out, 0, sizeof(double) * 2 * (size_t)(nrow * nchan * V));
    for (int64_t t = 0; t < ntime; ++t)
        for (int64_t row = tbin_idx[t]; row < tbin_idx[t] + tbin_counts[t]; ++row) {
            const int64_t p = ant1[row], q = ant2[row];
            for (int64_t nu = 0; nu < nchan; ++nu)
                jones_term(mode, ncorr, ndir, jn + ((t * nant + p) * nchan + nu) * ndir * J,
                           md + (row * nchan + nu) * ndir * V, jn + ((t * nant + q) * nchan + nu) * ndir * J, +1,
                           o + (row * nchan + nu) * V);
        }
    return 0;
}

/* vis / flag (nrow,nchan,V); residual = vis - sum_dir term where no correlation of (row,chan) is flagged, 0 elsewhere */
int orc_residual_vis_c128(const int64_t *tbin_idx, const int64_t *tbin_counts, int64_t ntime, const int64_t *ant1,
                          const int64_t *ant2, const double *jones, const double *vis, const unsigned char *flag,
                          const double *model, int64_t nrow, int64_t nant, int64_t nchan, int64_t ndir, int mode,
                          int ncorr, double *out)
{
    const int J = jones_elems(mode, ncorr), V = vis_elems(mode, ncorr);
    const cplx *jn = (const cplx *)jones, *md = (const cplx *)model, *vs = (const cplx *)vis;
    cplx *o = (cplx *)out;
    memset(out, 0, sizeof(double) * 2 * (size_t)(nrow * nchan * V));
    for (int64_t t = 0; t < ntime; ++t)
        for (int64_t row = tbin_idx[t]; row < tbin_idx[t] + tbin_counts[t]; ++row) {
            const int64_t p = ant1[row], q = ant2[row];
            for (int64_t nu = 0; nu < nchan; ++nu) {
                int any = 0;
                for (int c = 0; c < V; ++c) any |= flag[(row * nchan + nu) * V + c] != 0;
                if (any) continue;
                cplx *acc = o + (row * nchan + nu) * V;
                for (int c = 0; c < V; ++c) acc[c] = vs[(row * nchan + nu) * V + c];
                jones_term(mode, ncorr, ndir, jn + ((t * nant + p) * nchan + nu) * ndir * J,
                           md + (row * nchan + nu) * ndir * V, jn + ((t * nant + q) * nchan + nu) * ndir * J, -1, acc);
            }
        }
    return 0;
}

/* correct_vis.py:10-115: ndir must be 1; corrected = G_p^-1 vis G_q^-H where unflagged, 0 elsewhere */
int orc_correct_vis_c128(const int64_t *tbin_idx, const int64_t *tbin_counts, int64_t ntime, const int64_t *ant1,
                         const int64_t *ant2, const double *jones, const double *vis, const unsigned char *flag,
                         int64_t nrow, int64_t nant, int64_t nchan, int mode, int ncorr, double *out)
{
    const int J = jones_elems(mode, ncorr), V = vis_elems(mode, ncorr);
    const cplx *jn = (const cplx *)jones, *vs = (const cplx *)vis;
    cplx *o = (cplx *)out;
    memset(out, 0, sizeof(double) * 2 * (size_t)(nrow * nchan * V));
    for (int64_t t = 0; t < ntime; ++t)
        for (int64_t row = tbin_idx[t]; row < tbin_idx[t] + tbin_counts[t]; ++row) {
            const int64_t p = ant1[row], q = ant2[row];
            for (int64_t nu = 0; nu < nchan; ++nu) {
                int any = 0;
                for (int c = 0; c < V; ++c) any |= flag[(row * nchan + nu) * V + c] != 0;
                if (any) continue;
                const cplx *a1 = jn + ((t * nant + p) * nchan + nu) * J, *a2 = jn + ((t * nant + q) * nchan + nu) * J;
                const cplx *b = vs + (row * nchan + nu) * V;
                cplx *r = o + (row * nchan + nu) * V;
                if (mode == 0) {
                    for (int c = 0; c < ncorr; ++c) r[c] = cdiv(b[c], cmul(a1[c], cconj(a2[c])));
                } else if (mode == 1) {
                    r[0] = cdiv(b[0], cmul(a1[0], cconj(a2[0])));
                    r[1] = cdiv(b[1], cmul(a1[0], cconj(a2[1])));
                    r[2] = cdiv(b[2], cmul(a1[1], cconj(a2[0])));
                    r[3] = cdiv(b[3], cmul(a1[1], cconj(a2[1])));
                } else {
                    const cplx det1 = csub(cmul(a1[0], a1[3]), cmul(a1[1], a1[2]));
                    const cplx a00 = cdiv(a1[3], det1), a01 = cdiv(cneg(a1[1]), det1);
                    const cplx a10 = cdiv(cneg(a1[2]), det1), a11 = cdiv(a1[0], det1);
                    const cplx c0 = cconj(a2[0]), c1 = cconj(a2[1]), c2 = cconj(a2[2]), c3 = cconj(a2[3]);
                    const cplx det2 = csub(cmul(c0, c3), cmul(c1, c2));
                    const cplx b00 = cdiv(c3, det2), b01 = cdiv(cneg(c2), det2);
                    const cplx b10 = cdiv(cneg(c1), det2), b11 = cdiv(c0, det2);
                    cplx t1 = cmul(a00, b[0]), t2 = cmul(a01, b[2]), t3 = cmul(a00, b[1]), t4 = cmul(a01, b[3]);
                    r[0] = cadd(cadd(cadd(cmul(t1, b00), cmul(t2, b00)), cmul(t3, b10)), cmul(t4, b10));
                    r[1] = cadd(cadd(cadd(cmul(t1, b01), cmul(t2, b01)), cmul(t3, b11)), cmul(t4, b11));
                    t1 = cmul(a10, b[0]); t2 = cmul(a11, b[2]); t3 = cmul(a10, b[1]); t4 = cmul(a11, b[3]);
                    r[2] = cadd(cadd(cadd(cmul(t1, b00), cmul(t2, b00)), cmul(t3, b10)), cmul(t4, b10));
                    r[3] = cadd(cadd(cadd(cmul(t1, b01), cmul(t2, b01)), cmul(t3, b11)), cmul(t4, b11));
                }
            }
        }
    return 0;
}

/* compute_and_corrupt_vis.py:73-152: model (ntime,nchan,ndir,V), lm (ntime,ndir,2), uvw (nrow,3), freq (nchan) */
int orc_compute_and_corrupt_vis_c128(const int64_t *tbin_idx, const int64_t *tbin_counts, int64_t ntime,
                                     const int64_t *ant1, const int64_t *ant2, const double *jones, const double *model,
                                     const double *uvw, const double *freq, const double *lm, int64_t nrow, int64_t nant,
                                     int64_t nchan, int64_t ndir, int mode, int ncorr, double *out)
{
    const int J = jones_elems(mode, ncorr), V = vis_elems(mode, ncorr);
    const cplx *jn = (const cplx *)jones, *md = (const cplx *)model;
    cplx *o = (cplx *)out;
    memset(out, 0, sizeof(double) * 2 * (size_t)(nrow * nchan * V));
    for (int64_t t = 0; t < ntime; ++t)
        for (int64_t row = tbin_idx[t]; row < tbin_idx[t] + tbin_counts[t]; ++row) {
            const int64_t p = ant1[row], q = ant2[row];
            for (int64_t nu = 0; nu < nchan; ++nu)
                jones_term_computed(mode, ncorr, ndir, jn + ((t * nant + p) * nchan + nu) * ndir * J,
                                    md + (t * nchan + nu) * ndir * V, jn + ((t * nant + q) * nchan + nu) * ndir * J,
                                    uvw + 3 * row, freq[nu], lm + t * ndir * 2, o + (row * nchan + nu) * V);
        }
    return 0;
}
