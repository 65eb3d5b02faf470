// im_to_vis: direct Fourier transform image -> visibilities for gfx950.
//
// Replaces africanus/dft/kernels.py:33-67 (nb_im_to_vis.impl):
//     vis[r,nu,c] = sum_s exp(i*C*(l_s u_r + m_s v_r + n_s w_r)*nu) * image[s,nu,c]
// with n = sqrt(1-l^2-m^2)-1 unclamped (:54) and zero pixels skipped (:64).
//
// Design ("im_to_vis"):
//   * one lane owns one row and a tile of CT channels x NC correlations of complex
//     accumulators in VGPRs (CT=13, NC=4 -> 104 doubles = 208 VGPRs); the wave walks all sources.
//   * everything that is uniform across the wave (l,m,n of the source, the image
//     pixels of the (source, channel tile), tile frequency constants) is read through
//     the scalar data cache into SGPRs and used directly as v_fma_f64 operands, so the
//     inner loop has no LDS or vector-memory traffic at all: it is pure fp64 VALU.
//   * recurrence kernel (uniformly spaced channels): per (row, source, tile) two
//     quarter-turn-reduced polynomial sincos give the phasor at the tile's first
//     channel and the channel-to-channel rotation; the remaining channels follow from
//     the three-term recurrence y[j+1] = 2cos(d)*y[j] - y[j-1] (1 FMA per component).
//   * exact kernel: the reference's operation order (no contraction) and a full
//     accuracy sincos per (row, source, channel); used for non-uniform frequencies and
//     on request (AF_DFT_EXACT).
//   * a prep pass (tiny kernels, no host sync) computes n per source, repacks the image
//     into zero-padded channel tiles, decides uniformity on the device and records the
//     reference's zero-pixel / NaN-source semantics per (channel, corr) column.
#include "af_common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 256;
constexpr int MAXNC = 4;         // correlations per launch (ncorr is processed in chunks)

// ---- workspace layout ------------------------------------------------------------
struct WsLayout {
    size_t flags;     // int[64]        [0] uniform
    size_t lmn;       // double[nsrc*4] (l, m, n, 0), non-finite sources zeroed
    size_t srcbad;    // int[nsrc]      1 if (l,m,n) is not finite
    size_t tilef;     // double[ntile*4] (F0_4, FD_4, 0, 0): quarter-turns per metre
    size_t freq;      // double[ntile*CT] sign*nu/c scaled (exact kernel: nu itself)
    size_t colstate;  // int[nchan_pad*ncorr] 0 normal, 1 force zero, 2 force NaN
    size_t tilestate; // int[ntile*nchunk] OR of colstate in the tile/chunk
    size_t image;     // packed image, chunk-major: [chunk][tile][src][CT][nc][W]
    size_t total;
    int64_t ntile, nchunk;
    int ct;
};

WsLayout ws_layout(int64_t nsrc, int64_t nchan, int64_t ncorr, int is_complex, int CT)
{
    WsLayout L;
    L.ct = CT;
    L.ntile = af_cdiv(nchan > 0 ? nchan : 1, CT);
    L.nchunk = af_cdiv(ncorr > 0 ? ncorr : 1, MAXNC);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    L.flags = take(64 * sizeof(int));
    L.lmn = take((size_t)nsrc * 4 * sizeof(double));
    L.srcbad = take((size_t)nsrc * sizeof(int));
    L.tilef = take((size_t)L.ntile * 4 * sizeof(double));
    L.freq = take((size_t)L.ntile * CT * sizeof(double));
    L.colstate = take((size_t)L.ntile * CT * ncorr * sizeof(int));
    L.tilestate = take((size_t)L.ntile * L.nchunk * sizeof(int));
    L.image = take((size_t)nsrc * L.ntile * CT * ncorr * (is_complex ? 2 : 1) * sizeof(double));
    L.total = o;
    return L;
}

// ---- prep kernels ------------------------------------------------------------------
// n = sqrt(1 - l^2 - m^2) - 1 in the reference's operation order (kernels.py:54).
__global__ void dft_prep_src(const double *__restrict__ lm, int64_t nsrc, double *__restrict__ lmn,
                             int *__restrict__ srcbad)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    double l = lm[2 * s], m = lm[2 * s + 1];
    double n = __dsub_rn(__dsub_rn(1.0, __dmul_rn(l, l)), __dmul_rn(m, m));
    n = __dsub_rn(__dsqrt_rn(n), 1.0);
    bool bad = !(isfinite(l) && isfinite(m) && isfinite(n));
    srcbad[s] = bad ? 1 : 0;
    lmn[4 * s + 0] = bad ? 0.0 : l;
    lmn[4 * s + 1] = bad ? 0.0 : m;
    lmn[4 * s + 2] = bad ? 0.0 : n;
    lmn[4 * s + 3] = 0.0;
}

// Per channel tile: quarter-turn rates F0_4 = 4*sign*nu[c0]/c and FD_4 = 4*sign*dnu/c,
// dnu from the tile's own end points; uniform iff every channel of every tile sits within
// 2 ulp of the tile's arithmetic progression.  One thread per tile; flags[0] &= uniform.
__global__ void dft_prep_freq(const double *__restrict__ freq, int64_t nchan, int64_t ntile, int CT, int sign,
                              double *__restrict__ tilef, double *__restrict__ freq_pad,
                              int *__restrict__ flags)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntile) return;
    const int64_t c0 = t * CT;
    const int64_t nc = (nchan - c0 < CT) ? (nchan - c0) : CT;
    const double f0 = freq[c0];
    const double df = (nc > 1) ? (freq[c0 + nc - 1] - f0) / (double)(nc - 1) : 0.0;
    bool uniform = isfinite(f0) && isfinite(df);
    for (int64_t j = 0; j < CT; ++j) {
        double f = (j < nc) ? freq[c0 + j] : (f0 + (double)j * df);
        freq_pad[c0 + j] = f;
        double pred = f0 + (double)j * df;
        double tol = 2.0 * 2.220446049250313e-16 * fmax(fabs(f), fabs(pred));
        if (!(fabs(f - pred) <= tol)) uniform = false;
    }
    const double s4 = 4.0 * (double)sign;
    tilef[4 * t + 0] = s4 * f0 / AF_LIGHTSPEED;
    tilef[4 * t + 1] = s4 * df / AF_LIGHTSPEED;
    tilef[4 * t + 2] = 0.0;
    tilef[4 * t + 3] = 0.0;
    if (!uniform) atomicAnd(&flags[0], 0);
}

// Repack image (nsrc, nchan, ncorr[, 2]) into [chunk][tile][src][CT][nc][W], zero padded;
// sources whose (l,m,n) is not finite are zeroed (their effect is applied via colstate).
__global__ void dft_pack_image(const double *__restrict__ image, int W, int64_t nsrc, int64_t nchan,
                               int64_t ncorr, int64_t ntile, int CT, const int *__restrict__ srcbad,
                               double *__restrict__ packed)
{
    const int64_t total = nsrc * ntile * CT * ncorr;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        // source order of the scan: (s, chan_padded, corr)
        int64_t c = i % ncorr;
        int64_t ch = (i / ncorr) % (ntile * CT);
        int64_t s = i / (ncorr * ntile * CT);
        int64_t chunk = c / MAXNC, cc = c % MAXNC;
        int64_t nc = (ncorr - chunk * MAXNC < MAXNC) ? (ncorr - chunk * MAXNC) : MAXNC;
        int64_t tile = ch / CT, j = ch % CT;
        // elements before this chunk: chunk*MAXNC correlations over all (tile, src, CT)
        int64_t base = chunk * MAXNC * (ntile * nsrc * CT);
        int64_t dst = base + ((tile * nsrc + s) * CT + j) * nc + cc;
        bool live = (ch < nchan) && !srcbad[s];
        for (int k = 0; k < W; ++k)
            packed[dst * W + k] = live ? image[((s * nchan + ch) * ncorr + c) * W + k] : 0.0;
    }
}

// Column state per (chan, corr): reference semantics of `if image[s,nu,c]:` (kernels.py:64)
//   all pixels zero                       -> the output stays exactly 0 (state 1)
//   a non-finite source has a nonzero pixel -> every row gets NaN there (state 2)
__global__ void dft_colstate(const double *__restrict__ image, int W, int64_t nsrc, int64_t nchan,
                             int64_t ncorr, int64_t ntile, int CT, const int *__restrict__ srcbad,
                             int *__restrict__ colstate, int *__restrict__ tilestate, int nchunk)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ncol = ntile * CT * ncorr;
    if (i >= ncol) return;
    int64_t c = i % ncorr, ch = i / ncorr;
    int state = 0;
    if (ch < nchan) {
        bool any_nz = false, poison = false;
        for (int64_t s = 0; s < nsrc; ++s) {
            const double *px = image + ((s * nchan + ch) * ncorr + c) * W;
            bool nz = (px[0] != 0.0) || (W == 2 && px[1] != 0.0);
            any_nz |= nz;
            poison |= (nz && srcbad[s]);
        }
        state = poison ? 2 : (any_nz ? 0 : 1);
    }
    colstate[i] = state;
    if (state) atomicOr(&tilestate[(ch / CT) * nchunk + c / MAXNC], state);
}

// ---- quarter-turn sincos ---------------------------------------------------------
// (cos, sin)(2*pi*t) for t given in QUARTER turns t4 = 4*t.  Reduction: r = rint(t4) by the
// 1.5*2^52 magic-number add (also yields the quadrant in the low dword), f = t4 - r exact in
// [-0.5, 0.5]; sin(pi/2 f) = f*S(f^2), cos(pi/2 f) = C(f^2), Chebyshev-node fits on
// f^2 in [0, 0.25]: |err| <= 7e-15 (S), 6e-14 (C) with 6 terms; ~1e-16 with 7 terms (and exactly
// (1, 0) at f = 0, so a source at the phase centre gets a unit phasor).  The kernels use 7: an
// error e in cos of the channel-step angle is amplified by the recurrence to ~j^2*e at channel j.
template <int NTERM>
__device__ __forceinline__ void sincos_quarter_turns(double t4, double &c_out, double &s_out)
{
    static_assert(NTERM == 6 || NTERM == 7, "6 or 7 polynomial terms");
    constexpr double S6[7] = {0x1.921fb54442cfap+0, -0x1.4abbce6257a2ap-1, 0x1.466bc67123fa1p-4,
                              -0x1.32d2c644adc0bp-8, 0x1.5071ce4b47930p-13, -0x1.dd54805f3f706p-19, 0.0};
    constexpr double C6[7] = {0x1.ffffffffffe0bp-1, -0x1.3bd3cc9bd2c35p+0, 0x1.03c1f074ded21p-2,
                              -0x1.55d3ba300cd50p-6, 0x1.e1e7ccccb387ap-11, -0x1.a0ee132c60c1fp-16, 0.0};
    constexpr double S7[7] = {0x1.921fb54442d18p+0, -0x1.4abbce625be41p-1, 0x1.466bc677587f8p-4,
                              -0x1.32d2cce2e5b19p-8, 0x1.50782fda12d96p-13, -0x1.e30071afc3e59p-19,
                              0x1.e3f38399551bfp-25};
    constexpr double C7[7] = {0x1.0000000000000p+0, -0x1.3bd3cc9be458bp+0, 0x1.03c1f081b0780p-2,
                              -0x1.55d3c7dbfd139p-6, 0x1.e1f4fb60281f6p-11, -0x1.a6c9c1be9eb49p-16,
                              0x1.f3dbcea61b1a4p-22};
    const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
    double a = __dadd_rn(t4, MAGIC);
    int q = __double2loint(a);                // low 32 bits = rint(t4) mod 2^32
    double r = __dsub_rn(a, MAGIC);
    double f = __dsub_rn(t4, r);
    double z = __dmul_rn(f, f);
    double ps = NTERM == 6 ? S6[5] : S7[6], pc = NTERM == 6 ? C6[5] : C7[6];
#pragma unroll
    for (int i = NTERM - 2; i >= 0; --i) {
        ps = fma(ps, z, NTERM == 6 ? S6[i] : S7[i]);
        pc = fma(pc, z, NTERM == 6 ? C6[i] : C7[i]);
    }
    ps = __dmul_rn(ps, f);
    // quadrant: 0 (c,s)  1 (-s,c)  2 (-c,-s)  3 (s,-c)
    const bool swap = q & 1;
    double cc = swap ? ps : pc;
    double ss = swap ? pc : ps;
    int chi = __double2hiint(cc) ^ (((q + 1) & 2) << 30);
    int shi = __double2hiint(ss) ^ ((q & 2) << 30);
    c_out = __hiloint2double(chi, __double2loint(cc));
    s_out = __hiloint2double(shi, __double2loint(ss));
}

// ---- epilogue shared by both kernels ------------------------------------------------
template <int CT, int NC>
__device__ __forceinline__ void store_tile(const double (&acc)[CT][NC][2], double *__restrict__ out,
                                           int64_t row, bool valid, int64_t nchan, int64_t ncorr,
                                           int64_t c0, int64_t corr0, const int *__restrict__ colstate,
                                           int tstate)
{
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        if (c0 + j < nchan) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double re = acc[j][c][0], im = acc[j][c][1];
                if (tstate) {  // wave-uniform: rare zero-column / NaN-source semantics
                    int st = colstate[(c0 + j) * ncorr + corr0 + c];
                    if (st == 1) { re = 0.0; im = 0.0; }
                    if (st == 2) { re = __longlong_as_double(0x7ff8000000000000LL); im = re; }
                }
                double2 v = make_double2(re, im);
                *reinterpret_cast<double2 *>(out + 2 * ((row * nchan + c0 + j) * ncorr + corr0 + c)) = v;
            }
        }
    }
}

// ---- recurrence kernel ---------------------------------------------------------------
// grid: (ceil(nrow/256), ntile); block 256 = 4 waves, each wave 64 consecutive rows.
template <int CT, int NC, bool CPLX, int NTERM>
__global__ __launch_bounds__(ROWS_PER_BLOCK) void dft_recurrence_kernel(
    const double *__restrict__ uvw, const double *__restrict__ lmn, const double *__restrict__ packed,
    const double *__restrict__ tilef, const int *__restrict__ flags, const int *__restrict__ colstate,
    const int *__restrict__ tilestate, double *__restrict__ out, int64_t nrow, int nsrc, int64_t nchan,
    int64_t ncorr, int64_t corr0, int chunk, int nchunk, int want_uniform)
{
    if (flags[0] != want_uniform) return;  // decided on the device by dft_prep_freq
    constexpr int W = CPLX ? 2 : 1;
    const int tile = blockIdx.y;
    const int64_t c0 = (int64_t)tile * CT;
    int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x;
    const bool valid = row < nrow;
    if (!valid) row = nrow - 1;
    const double u = uvw[3 * row], v = uvw[3 * row + 1], w = uvw[3 * row + 2];
    const double F0 = tilef[4 * tile], FD = tilef[4 * tile + 1];

    double acc[CT][NC][2];
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[j][c][0] = acc[j][c][1] = 0.0;

    const double *__restrict__ img = packed + (int64_t)tile * nsrc * (CT * NC * W);
    // (l, m, n) of the NEXT source is fetched one iteration ahead, so the scalar loads of this
    // source's image pixels can be issued at the top of the iteration and land while the
    // sincos setup (which only needs l, m, n) runs.
    // hipcc sinks a plain prefetch load back to its use, so the three scalar loads are issued
    // from an asm statement (not counted by hipcc) and retired by the explicit lgkmcnt(0) at the
    // bottom of the iteration, by which time every scalar load of the iteration has landed.
    double l = lmn[0], m = lmn[1], n = lmn[2];
#pragma unroll 1
    for (int s = 0; s < nsrc; ++s) {
        const int sn = (s + 1 < nsrc) ? s + 1 : s;
        const double *lmn_next = lmn + 4 * sn;
        double ln, mn, nn;
        asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %3, 0x8\n\ts_load_dwordx2 %2, %3, 0x10"
                     : "=&s"(ln), "=&s"(mn), "=&s"(nn)
                     : "s"(lmn_next));
        // path difference in metres; FMA-contracted (one rounding less than the reference)
        const double q = fma(n, w, fma(m, v, __dmul_rn(l, u)));
        double c0r, c0i, dr, di;
        sincos_quarter_turns<NTERM>(__dmul_rn(q, F0), c0r, c0i);
        sincos_quarter_turns<NTERM>(__dmul_rn(q, FD), dr, di);
        const double k = __dadd_rn(dr, dr);
        double y0r = c0r, y0i = c0i;
        double y1r = fma(c0r, dr, -__dmul_rn(c0i, di));
        double y1i = fma(c0r, di, __dmul_rn(c0i, dr));
        const double *__restrict__ g = img + (int64_t)s * (CT * NC * W);
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            double yr, yi;
            if (j == 0) { yr = y0r; yi = y0i; }
            else if (j == 1) { yr = y1r; yi = y1i; }
            else {
                yr = fma(k, y1r, -y0r);
                yi = fma(k, y1i, -y0i);
                y0r = y1r; y0i = y1i; y1r = yr; y1i = yi;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (CPLX) {
                    const double gr = g[(j * NC + c) * 2], gi = g[(j * NC + c) * 2 + 1];
                    acc[j][c][0] = fma(gr, yr, acc[j][c][0]);
                    acc[j][c][0] = fma(-gi, yi, acc[j][c][0]);
                    acc[j][c][1] = fma(gi, yr, acc[j][c][1]);
                    acc[j][c][1] = fma(gr, yi, acc[j][c][1]);
                } else {
                    const double gr = g[j * NC + c];
                    acc[j][c][0] = fma(gr, yr, acc[j][c][0]);
                    acc[j][c][1] = fma(gr, yi, acc[j][c][1]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ln), "+s"(mn), "+s"(nn));
        l = ln; m = mn; n = nn;
    }
    store_tile<CT, NC>(acc, out, row, valid, nchan, ncorr, c0, corr0, colstate,
                   tilestate[tile * nchunk + chunk]);
}

// ---- exact kernel -----------------------------------------------------------------------
// Reference operation order, no contraction (kernels.py:57,61,65), full-accuracy sincos per
// (row, source, channel).  Channel-outermost inside the lane so only NC complex accumulators
// are live; real_phase is recomputed per channel (6 ops against a ~100-op sincos).
template <int CT, int NC, bool CPLX>
__global__ __launch_bounds__(ROWS_PER_BLOCK) void dft_exact_kernel(
    const double *__restrict__ uvw, const double *__restrict__ lmn, const double *__restrict__ packed,
    const double *__restrict__ freq_pad, const int *__restrict__ flags, const int *__restrict__ colstate,
    const int *__restrict__ tilestate, double *__restrict__ out, int64_t nrow, int nsrc, int64_t nchan,
    int64_t ncorr, int64_t corr0, int chunk, int nchunk, int want_uniform, double constant)
{
    if (want_uniform >= 0 && flags[0] != want_uniform) return;
    constexpr int W = CPLX ? 2 : 1;
    const int tile = blockIdx.y;
    const int64_t c0 = (int64_t)tile * CT;
    int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x;
    const bool valid = row < nrow;
    if (!valid) row = nrow - 1;
    const double u = uvw[3 * row], v = uvw[3 * row + 1], w = uvw[3 * row + 2];
    const double *__restrict__ img = packed + (int64_t)tile * nsrc * (CT * NC * W);
    const int tstate = tilestate[tile * nchunk + chunk];

    for (int j = 0; j < CT; ++j) {
        if (c0 + j >= nchan) break;
        const double nu = freq_pad[c0 + j];
        double acc[NC][2];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c][0] = acc[c][1] = 0.0;
        for (int s = 0; s < nsrc; ++s) {
            const double l = lmn[4 * s], m = lmn[4 * s + 1], n = lmn[4 * s + 2];
            const double real_phase = __dmul_rn(
                constant, __dadd_rn(__dadd_rn(__dmul_rn(l, u), __dmul_rn(m, v)), __dmul_rn(n, w)));
            double yr, yi;
            sincos(__dmul_rn(real_phase, nu), &yi, &yr);
            const double *__restrict__ g = img + ((int64_t)s * CT + j) * (NC * W);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (CPLX) {
                    const double gr = g[c * 2], gi = g[c * 2 + 1];
                    // exp(p)*image, then += : (yr*gr - yi*gi, yr*gi + yi*gr)
                    acc[c][0] = __dadd_rn(acc[c][0], __dsub_rn(__dmul_rn(yr, gr), __dmul_rn(yi, gi)));
                    acc[c][1] = __dadd_rn(acc[c][1], __dadd_rn(__dmul_rn(yr, gi), __dmul_rn(yi, gr)));
                } else {
                    const double gr = g[c];
                    acc[c][0] = __dadd_rn(acc[c][0], __dmul_rn(yr, gr));
                    acc[c][1] = __dadd_rn(acc[c][1], __dmul_rn(yi, gr));
                }
            }
        }
        if (valid) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double re = acc[c][0], im = acc[c][1];
                if (tstate) {
                    int st = colstate[(c0 + j) * ncorr + corr0 + c];
                    if (st == 1) { re = 0.0; im = 0.0; }
                    if (st == 2) { re = __longlong_as_double(0x7ff8000000000000LL); im = re; }
                }
                *reinterpret_cast<double2 *>(out + 2 * ((row * nchan + c0 + j) * ncorr + corr0 + c)) =
                    make_double2(re, im);
            }
        }
    }
}

// ---- host-side dispatch ---------------------------------------------------------------
// Tile width: the lane's accumulators (CT*NC*2 doubles) plus ~45 working VGPRs must fit the
// 256 architectural VGPRs a VALU instruction can address; per (row, source, tile) the
// recurrence kernel spends ~51 fp64 ops of setup plus CT*(2 + 2*NC*(CPLX?2:1)) in the
// channel loop, so the widest tile that fits wins unless padding the last tile costs more.
int choose_ct(int64_t nchan, int nc_max, bool cplx)
{
    const int cands[3] = {13, 12, 8};
    const int per_chan = 2 + 2 * nc_max * (cplx ? 2 : 1);
    int best = cands[0];
    int64_t best_cost = -1;
    for (int k = 0; k < 3; ++k) {
        int ct = cands[k];
        int64_t cost = af_cdiv(nchan, ct) * (51 + (int64_t)ct * per_chan);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ct; }
    }
    return best;
}

struct Args {
    const WsLayout *L;
    char *ws;
    const double *uvw;
    double *out;
    int64_t nrow, nsrc, nchan, ncorr;
    int chunk, mode;
    double constant;
    hipStream_t st;
};

template <int CT, int NC, bool CPLX>
int launch_chunk(const Args &a)
{
    constexpr int W = CPLX ? 2 : 1;
    const WsLayout &L = *a.L;
    char *ws = a.ws;
    const double *lmn = reinterpret_cast<const double *>(ws + L.lmn);
    const double *tilef = reinterpret_cast<const double *>(ws + L.tilef);
    const double *freq_pad = reinterpret_cast<const double *>(ws + L.freq);
    const int *flags = reinterpret_cast<const int *>(ws + L.flags);
    const int *colstate = reinterpret_cast<const int *>(ws + L.colstate);
    const int *tilestate = reinterpret_cast<const int *>(ws + L.tilestate);
    const double *packed = reinterpret_cast<const double *>(ws + L.image) +
                           (int64_t)a.chunk * MAXNC * (L.ntile * a.nsrc * CT) * W;
    dim3 grid((unsigned)af_cdiv(a.nrow, ROWS_PER_BLOCK), (unsigned)L.ntile), block(ROWS_PER_BLOCK);
    const int64_t corr0 = (int64_t)a.chunk * MAXNC;
    if (a.mode == AF_DFT_AUTO || a.mode == AF_DFT_RECURRENCE) {
        // runs iff flags[0] == 1 (set by dft_prep_freq, or forced for AF_DFT_RECURRENCE)
        hipLaunchKernelGGL((dft_recurrence_kernel<CT, NC, CPLX, 7>), grid, block, 0, a.st, a.uvw, lmn, packed,
                           tilef, flags, colstate, tilestate, a.out, a.nrow, (int)a.nsrc, a.nchan, a.ncorr,
                           corr0, a.chunk, (int)L.nchunk, 1);
        AF_LAUNCH_CHECK();
    }
    if (a.mode == AF_DFT_AUTO || a.mode == AF_DFT_EXACT) {
        // AUTO: runs iff flags[0] == 0 (non-uniform frequencies); EXACT: always (-1)
        hipLaunchKernelGGL((dft_exact_kernel<CT, NC, CPLX>), grid, block, 0, a.st, a.uvw, lmn, packed, freq_pad,
                           flags, colstate, tilestate, a.out, a.nrow, (int)a.nsrc, a.nchan, a.ncorr, corr0,
                           a.chunk, (int)L.nchunk, a.mode == AF_DFT_EXACT ? -1 : 0, a.constant);
        AF_LAUNCH_CHECK();
    }
    return AF_OK;
}

template <int CT, bool CPLX>
int launch_chunk_nc(int nc, const Args &a)
{
    switch (nc) {
    case 1: return launch_chunk<CT, 1, CPLX>(a);
    case 2: return launch_chunk<CT, 2, CPLX>(a);
    case 3: return launch_chunk<CT, 3, CPLX>(a);
    default: return launch_chunk<CT, 4, CPLX>(a);
    }
}

template <bool CPLX>
int launch_chunk_ct(int ct, int nc, const Args &a)
{
    switch (ct) {
    case 8: return launch_chunk_nc<8, CPLX>(nc, a);
    case 12: return launch_chunk_nc<12, CPLX>(nc, a);
    default: return launch_chunk_nc<13, CPLX>(nc, a);
    }
}

}  // namespace

AF_EXPORT size_t af_im_to_vis_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t ncorr, int image_is_complex)
{
    if (nsrc < 0 || nchan < 0 || ncorr < 0) return 0;
    // sized for the widest padding any tile width can produce
    size_t m = 0;
    const int cands[3] = {8, 12, 13};
    for (int k = 0; k < 3; ++k) {
        size_t t = ws_layout(nsrc, nchan, ncorr, image_is_complex, cands[k]).total;
        if (t > m) m = t;
    }
    return m;
}

AF_EXPORT int af_im_to_vis_f64(const double *image, int image_is_complex, const double *uvw, const double *lm,
                               const double *frequency, int64_t nsrc, int64_t nrow, int64_t nchan,
                               int64_t ncorr, int convention, int mode, double *out, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(convention == AF_CONVENTION_FOURIER || convention == AF_CONVENTION_CASA,
               "convention not in ('fourier', 'casa')");
    AF_REQUIRE(mode == AF_DFT_AUTO || mode == AF_DFT_EXACT || mode == AF_DFT_RECURRENCE,
               "af_im_to_vis_f64: unknown mode %d", mode);
    AF_REQUIRE(nsrc >= 0 && nrow >= 0 && nchan >= 0 && ncorr >= 0, "af_im_to_vis_f64: negative extent");
    AF_REQUIRE(nsrc < (1LL << 31), "af_im_to_vis_f64: nsrc too large");
    hipStream_t st = af_stream(stream);
    if (nrow == 0 || nchan == 0 || ncorr == 0) return AF_OK;
    AF_REQUIRE(out != nullptr && uvw != nullptr && frequency != nullptr, "af_im_to_vis_f64: NULL array");
    if (nsrc == 0) {  // np.zeros output (kernels.py:45)
        AF_HIP(hipMemsetAsync(out, 0, sizeof(double) * 2 * (size_t)(nrow * nchan * ncorr), st));
        return AF_OK;
    }
    AF_REQUIRE(image != nullptr && lm != nullptr, "af_im_to_vis_f64: NULL array");
    const bool cplx = image_is_complex != 0;
    const int ct = choose_ct(nchan, (int)(ncorr < MAXNC ? ncorr : MAXNC), cplx);
    const WsLayout L = ws_layout(nsrc, nchan, ncorr, image_is_complex, ct);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= L.total,
               "af_im_to_vis_f64: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_im_to_vis_f64: workspace must be 256-byte aligned");
    AF_REQUIRE(L.ntile <= 65535, "af_im_to_vis_f64: too many channels");
    char *ws = static_cast<char *>(workspace);
    const int W = cplx ? 2 : 1;

    // flags[0] = 1: uniform until a tile says otherwise (1-byte memset of the low byte); tilestate = 0
    AF_HIP(hipMemsetAsync(ws + L.tilestate, 0, (size_t)L.ntile * L.nchunk * sizeof(int), st));
    AF_HIP(hipMemsetAsync(ws + L.flags, 0, 64 * sizeof(int), st));
    AF_HIP(hipMemsetAsync(ws + L.flags, 1, 1, st));
    hipLaunchKernelGGL(dft_prep_src, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st, lm, nsrc,
                       reinterpret_cast<double *>(ws + L.lmn), reinterpret_cast<int *>(ws + L.srcbad));
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL(dft_prep_freq, dim3((unsigned)af_cdiv(L.ntile, 64)), dim3(64), 0, st, frequency, nchan,
                       L.ntile, ct, convention, reinterpret_cast<double *>(ws + L.tilef),
                       reinterpret_cast<double *>(ws + L.freq), reinterpret_cast<int *>(ws + L.flags));
    AF_LAUNCH_CHECK();
    if (mode == AF_DFT_RECURRENCE)  // caller asserts uniform spacing
        AF_HIP(hipMemsetAsync(ws + L.flags, 1, 1, st));
    {
        int64_t total = nsrc * L.ntile * ct * ncorr;
        int64_t blocks = af_cdiv(total, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(dft_pack_image, dim3((unsigned)blocks), dim3(256), 0, st, image, W, nsrc, nchan, ncorr,
                           L.ntile, ct, reinterpret_cast<const int *>(ws + L.srcbad),
                           reinterpret_cast<double *>(ws + L.image));
        AF_LAUNCH_CHECK();
        int64_t ncol = L.ntile * ct * ncorr;
        hipLaunchKernelGGL(dft_colstate, dim3((unsigned)af_cdiv(ncol, 64)), dim3(64), 0, st, image, W, nsrc, nchan,
                           ncorr, L.ntile, ct, reinterpret_cast<const int *>(ws + L.srcbad),
                           reinterpret_cast<int *>(ws + L.colstate), reinterpret_cast<int *>(ws + L.tilestate),
                           (int)L.nchunk);
        AF_LAUNCH_CHECK();
    }
    Args a;
    a.L = &L; a.ws = ws; a.uvw = uvw; a.out = out;
    a.nrow = nrow; a.nsrc = nsrc; a.nchan = nchan; a.ncorr = ncorr;
    a.mode = mode; a.st = st;
    a.constant = convention == AF_CONVENTION_FOURIER ? AF_MINUS_TWO_PI_OVER_C : AF_TWO_PI_OVER_C;
    af_prof_begin(st);
    for (int chunk = 0; chunk < (int)L.nchunk; ++chunk) {
        int nc = (int)((ncorr - (int64_t)chunk * MAXNC < MAXNC) ? (ncorr - (int64_t)chunk * MAXNC) : MAXNC);
        a.chunk = chunk;
        int rc = cplx ? launch_chunk_ct<true>(ct, nc, a) : launch_chunk_ct<false>(ct, nc, a);
        if (rc != AF_OK) return rc;
    }
    af_prof_end(st);
    return AF_OK;
}
