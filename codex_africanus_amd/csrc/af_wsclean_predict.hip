// wsclean_predict: fused single-correlation predict from a WSClean component list for gfx950.
//
// Replaces africanus/rime/wsclean_predict.py:11-84 (wsclean_predict_main) together with the
// spectral model africanus/model/wsclean/spec_model.py:70-126 (spectra):
//     vis[r,f] = sum_s spectrum[s,f] * shape_s(r,f) * exp(+i 2pi/c (u l + v m + w n) nu_f)
//     n = sqrt(1 - l^2 - m^2) - 1 (unclamped), CASA sign
//     shape = 1 for POINT components; for GAUSSIAN components
//     shape = exp(-(u1^2 + v1^2) (nu gs)^2),  u1 = (u em - v el) er,  v1 = u el + v em,
//     el = emaj sin(pa), em = emaj cos(pa), er = emin / (emaj or 1), gs = sqrt(2) pi / (fwhm c)
//
// Same machine mapping as af_im_to_vis.hip (lane = row, a tile of CT channels of complex
// accumulators in VGPRs, per-(tile, source) records consumed through DPP row_newbcast operands and
// refreshed by counted asm loads), specialised for one correlation:
//   * records are [l, m, n, 0, el*gs, em*gs, er, 0, spectrum[c0 .. c0+CT)], CT up to 40;
//   * the prep pass orders the components points first, Gaussians second (stable), so the row loop
//     runs two branch-free passes; the point pass never evaluates an envelope;
//   * on uniformly spaced channels the phasor follows the three-term recurrence and the Gaussian
//     envelope E_f = exp(-A nu_f^2) follows the second-order product recurrence
//     E_{f+1} = E_f rho_f, rho_{f+1} = rho_f kappa (3 exp per (row, Gaussian, tile));
//   * exact kernel (non-uniform channels, AF_DFT_EXACT): the reference's operation order with a
//     full-accuracy sincos / exp per (row, component, channel), components in their given order.
#include <type_traits>

#include "af_common.h"
#include "af_sincos.h"
#include "af_dft_device.h"

namespace {

constexpr int ROWS_PER_BLOCK = 256;
constexpr int HDR = 8;  // header doubles of a record

__host__ __device__ constexpr int w_groups(int ct) { return (HDR + ct + GROUP - 1) / GROUP; }
__host__ __device__ constexpr int w_last_chan(int g, int nslot)
{
    int last = g * GROUP + GROUP - 1 < nslot - 1 ? g * GROUP + GROUP - 1 : nslot - 1;
    return last < HDR ? -1 : last - HDR;
}
__host__ __device__ constexpr int w_first_chan(int g) { return g == 0 ? -1 : g * GROUP - HDR; }
// loads that may stay in flight when group g is first used (see af_dft_device.h)
__host__ __device__ constexpr int w_wait_count(int g, int ng, int nslot)
{
    int n = 0;
    for (int h = 0; h < ng; ++h) {
        int ph = w_last_chan(h, nslot), pg = w_last_chan(g, nslot);
        if (ph > pg || (ph == pg && h > g)) ++n;
        if (ph < w_first_chan(g)) ++n;
    }
    return n;
}

struct WsLayout {
    size_t flags;     // int[64]: [0] channels uniformly spaced within every tile, [1] number of point components
    size_t order;     // int[nsrc]: component index in processing order (points, then Gaussians)
    size_t params;    // double[nsrc*8]: (l, m, n, is_gauss, el, em, er, 0) in the GIVEN order (exact kernel)
    size_t spectrum;  // double[nsrc*nchan]
    size_t tilef;     // double[ntile*8]: F0_4, FD_4, nu0^2, 2 nu0 dnu + dnu^2, 2 dnu^2, 0, 0, 0
    size_t records;   // double[ntile][nsrc][groups*16], processing order
    size_t total;
    int64_t ntile;
    int ct, groups;
};

void ws_layout(WsLayout &L, int64_t nsrc, int64_t nchan, int CT)
{
    L.ct = CT;
    L.groups = w_groups(CT);
    L.ntile = af_cdiv(nchan > 0 ? nchan : 1, CT);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    L.flags = take(64 * sizeof(int));
    L.order = take((size_t)nsrc * sizeof(int));
    L.params = take((size_t)nsrc * 8 * sizeof(double));
    L.spectrum = take((size_t)nsrc * nchan * sizeof(double));
    L.tilef = take((size_t)L.ntile * 8 * sizeof(double));
    L.records = take((size_t)L.ntile * nsrc * L.groups * GROUP * sizeof(double));
    L.total = o;
}

// numba lowers float ** int to exponentiation by squaring; same multiplication order here
__device__ inline double ipow(double a, int e)
{
    double r = 1.0;
    while (e != 0) {
        if (e & 1) r = __dmul_rn(r, a);
        e >>= 1;
        a = __dmul_rn(a, a);
    }
    return r;
}

// ---- spectra (spec_model.py:99-124): one thread per (component, channel) ------------------------
__global__ void wsc_spectra_kernel(const double *__restrict__ flux, const double *__restrict__ coeffs,
                                   const unsigned char *__restrict__ log_poly, const double *__restrict__ ref_freq,
                                   const double *__restrict__ freq, int64_t nsrc, int64_t ncoeffs, int64_t nchan,
                                   double *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsrc * nchan) return;
    const int64_t s = i / nchan, f = i - s * nchan;
    const double ratio = freq[f] / ref_freq[s];
    double acc;
    if (log_poly[s]) {
        const double lr = log(ratio);
        acc = 0.0;
        for (int64_t c = 0; c < ncoeffs; ++c) acc = __dadd_rn(acc, __dmul_rn(coeffs[s * ncoeffs + c], ipow(lr, (int)c + 1)));
        acc = __dmul_rn(flux[s], exp(acc));
    } else {
        const double x = __dsub_rn(ratio, 1.0);
        acc = flux[s];
        for (int64_t c = 0; c < ncoeffs; ++c) acc = __dadd_rn(acc, __dmul_rn(coeffs[s * ncoeffs + c], ipow(x, (int)c + 1)));
    }
    out[i] = acc;
}

// ---- component parameters (wsclean_predict.py:32, :50-54) ----------------------------------------
__global__ void wsc_prep_params(const double *__restrict__ lm, const unsigned char *__restrict__ is_gauss,
                                const double *__restrict__ gauss_shape, int64_t nsrc, double *__restrict__ params)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    const double l = lm[2 * s], m = lm[2 * s + 1];
    const double n = __dsub_rn(__dsqrt_rn(__dsub_rn(__dsub_rn(1.0, __dmul_rn(l, l)), __dmul_rn(m, m))), 1.0);
    double el = 0.0, em = 0.0, er = 0.0;
    const bool g = is_gauss[s] != 0;
    if (g) {
        const double emaj = gauss_shape[3 * s], emin = gauss_shape[3 * s + 1], angle = gauss_shape[3 * s + 2];
        el = __dmul_rn(emaj, sin(angle));
        em = __dmul_rn(emaj, cos(angle));
        er = emin / (emaj == 0.0 ? 1.0 : emaj);
    }
    double *p = params + 8 * s;
    p[0] = l; p[1] = m; p[2] = n; p[3] = g ? 1.0 : 0.0;
    p[4] = el; p[5] = em; p[6] = er; p[7] = 0.0;
}

// ---- processing order: points first, Gaussians second, both stable; one workgroup ------------------
__global__ __launch_bounds__(1024) void wsc_order_kernel(const unsigned char *__restrict__ is_gauss, int nsrc,
                                                         int *__restrict__ order, int *__restrict__ flags)
{
    __shared__ int wave_tot[16];
    __shared__ int base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // pass 0 places the points, pass 1 the Gaussians behind them
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        for (int s0 = 0; s0 < nsrc; s0 += 1024) {
            const int s = s0 + threadIdx.x;
            const bool mine = s < nsrc && ((is_gauss[s] != 0) == (pass == 1));
            const unsigned long long b = __ballot(mine);
            if (lane == 0) wave_tot[wave] = __popcll(b);
            __syncthreads();
            int off = base;
            for (int k = 0; k < wave; ++k) off += wave_tot[k];
            if (mine) order[off + __popcll(b & ((1ULL << lane) - 1ULL))] = s;
            __syncthreads();
            if (threadIdx.x == 0) {
                int t = 0;
                for (int k = 0; k < 16; ++k) t += wave_tot[k];
                base += t;
            }
            __syncthreads();
        }
        if (pass == 0 && threadIdx.x == 0) flags[1] = base;
    }
}

// ---- per-tile frequency constants; flags[0] &= every tile is an arithmetic progression (2 ulp) -----
__global__ void wsc_prep_freq(const double *__restrict__ freq, int64_t nchan, int64_t ntile, int CT,
                              double *__restrict__ tilef, int *__restrict__ flags)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntile) return;
    const int64_t c0 = t * CT;
    const int64_t nc = (nchan - c0 < CT) ? (nchan - c0) : CT;
    const double f0 = freq[c0];
    const double df = (nc > 1) ? (freq[c0 + nc - 1] - f0) / (double)(nc - 1) : 0.0;
    bool uniform = isfinite(f0) && isfinite(df);
    for (int64_t j = 0; j < nc; ++j) {
        const double f = freq[c0 + j], pred = f0 + (double)j * df;
        if (!(fabs(f - pred) <= 2.0 * 2.220446049250313e-16 * fmax(fabs(f), fabs(pred)))) uniform = false;
    }
    double *p = tilef + 8 * t;
    p[0] = 4.0 * f0 / AF_LIGHTSPEED;  // quarter turns per metre at the tile's first channel
    p[1] = 4.0 * df / AF_LIGHTSPEED;  // ... per channel step
    p[2] = f0 * f0;
    p[3] = 2.0 * f0 * df + df * df;
    p[4] = 2.0 * df * df;
    p[5] = p[6] = p[7] = 0.0;
    if (!uniform) atomicAnd(&flags[0], 0);
}

// ---- records, processing order: [l, m, n, 0, el*gs, em*gs, er, 0, spectrum of the tile, 0...] ---------
__global__ void wsc_pack_records(const double *__restrict__ params, const double *__restrict__ spectrum,
                                 const int *__restrict__ order, int64_t nsrc, int64_t nchan, int64_t ntile, int CT,
                                 int groups, double gauss_scale, double *__restrict__ rec)
{
    const int64_t per = (int64_t)groups * GROUP;
    const int64_t total = ntile * nsrc * per;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t slot = i % per;
        const int64_t k = (i / per) % nsrc;
        const int64_t tile = i / (per * nsrc);
        const int64_t s = order[k];
        double v = 0.0;
        if (slot < 3) v = params[8 * s + slot];
        else if (slot == 4 || slot == 5) v = params[8 * s + slot] * gauss_scale;
        else if (slot == 6) v = params[8 * s + 6];
        else if (slot >= HDR && slot < HDR + CT) {
            const int64_t ch = tile * CT + (slot - HDR);
            if (ch < nchan) v = spectrum[s * nchan + ch];
        }
        rec[i] = v;
    }
}

// ---- one component's pass over the lane's channel tile -----------------------------------------------
template <int CT, bool GAUSS, int NG>
__device__ __forceinline__ void component_pass(double (&acc)[CT][2], double (&R)[NG], double u, double v, double w,
                                               double F0, double FD, double NU0SQ, double C1, double C2,
                                               unsigned lane_off, const double *rec_next, const double2 *ptab)
{
    constexpr int NSLOT = HDR + CT;
    group_wait<w_wait_count(0, NG, NSLOT)>(R[0]);
    double q = 0.0;  // path difference in metres (slots 0..2)
    fmac_bcast<0>(q, R[0], u);
    fmac_bcast<1>(q, R[0], v);
    fmac_bcast<2>(q, R[0], w);
    double E = 1.0, rho = 1.0, kappa = 1.0;
    if constexpr (GAUSS) {
        double u1 = 0.0, u1e = 0.0, v1 = 0.0;  // already scaled by gs (slots 4..6)
        fmac_bcast<5>(u1, R[0], u);
        fmac_bcast<4, true>(u1, R[0], v);
        fmac_bcast<6>(u1e, R[0], u1);
        fmac_bcast<4>(v1, R[0], u);
        fmac_bcast<5>(v1, R[0], v);
        const double A = fma(u1e, u1e, __dmul_rn(v1, v1));
        E = exp(-__dmul_rn(A, NU0SQ));
        rho = exp(-__dmul_rn(A, C1));
        kappa = exp(-__dmul_rn(A, C2));
    }
    // channel-step and tile-start phasors from the block's table (af_sincos.h; F0, FD in 1/256 turns per metre)
    double dr, di, c0r, c0i;
    {
        TablePhasorStage sd, s0;
        table_phasor_reduce(sd, ptab, __dmul_rn(q, FD));
        table_phasor_reduce(s0, ptab, __dmul_rn(q, F0));
        table_phasor_sin(sd); table_phasor_sin(s0);
        table_phasor_cos(sd); table_phasor_cos(s0);
        table_phasor_finish(sd, dr, di);
        table_phasor_finish(s0, c0r, c0i);
    }
    const double k = __dadd_rn(dr, dr);
    double y0r = c0r, y0i = c0i;
    double y1r = fma(c0r, dr, -__dmul_rn(c0i, di));
    double y1i = fma(c0r, di, __dmul_rn(c0i, dr));
    static_for<0, CT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        double yr, yi;
        if constexpr (j == 0) { yr = y0r; yi = y0i; }
        else if constexpr (j == 1) { yr = y1r; yi = y1i; }
        else {
            yr = fma(k, y1r, -y0r);
            yi = fma(k, y1i, -y0i);
            y0r = y1r; y0i = y1i; y1r = yr; y1i = yi;
        }
        static_for<1, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (w_first_chan(g) == j) group_wait<w_wait_count(g, NG, NSLOT)>(R[g]);
        });
        constexpr int slot = HDR + j;
        if constexpr (GAUSS) {
            fmac_bcast<slot % GROUP>(acc[j][0], R[slot / GROUP], __dmul_rn(yr, E));
            fmac_bcast<slot % GROUP>(acc[j][1], R[slot / GROUP], __dmul_rn(yi, E));
            if constexpr (j + 1 < CT) {
                E = __dmul_rn(E, rho);
                rho = __dmul_rn(rho, kappa);
            }
        } else {
            fmac_bcast<slot % GROUP>(acc[j][0], R[slot / GROUP], yr);
            fmac_bcast<slot % GROUP>(acc[j][1], R[slot / GROUP], yi);
        }
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (w_last_chan(g, NSLOT) == j)
                group_refresh<g * GROUP * (int)sizeof(double)>(R[g], lane_off, rec_next);
        });
    });
}

// grid: (ceil(nrow/256), ntile); block 256 = 4 waves of 64 consecutive rows, all on tile blockIdx.y
template <int CT>
__global__ __launch_bounds__(ROWS_PER_BLOCK) void wsc_recurrence_kernel(
    const double *__restrict__ uvw, const double *__restrict__ records, const double *__restrict__ tilef,
    const int *__restrict__ flags, double *__restrict__ out, int64_t nrow, int nsrc, int64_t nchan)
{
    if (flags[0] != 1) return;  // decided on the device by wsc_prep_freq
    constexpr int NG = w_groups(CT);
    constexpr int STRIDE = CT + 1;
    __shared__ double2 stage[64 * STRIDE];
    __shared__ double2 ptab[PHASOR_TABLE];
    table_phasor_init(ptab, threadIdx.x, ROWS_PER_BLOCK);
    __syncthreads();
    const int tile = blockIdx.y;
    const int64_t c0 = (int64_t)tile * CT;
    int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x;
    if (row >= nrow) row = nrow - 1;
    const double u = uvw[3 * row], v = uvw[3 * row + 1], w = uvw[3 * row + 2];
    const double *tf = tilef + 8 * tile;
    const double F0 = 64.0 * tf[0], FD = 64.0 * tf[1];   // quarter turns -> 1/256 turns per metre (exact)
    const double NU0SQ = tf[2], C1 = tf[3], C2 = tf[4];
    const int npoint = flags[1];

    double acc[CT][2];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j][0] = acc[j][1] = 0.0;

    const unsigned lane_off = (threadIdx.x & (GROUP - 1)) * (unsigned)sizeof(double);
    const double *__restrict__ rec = records + (int64_t)tile * nsrc * (NG * GROUP);
    double R[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) R[g] = rec[g * GROUP + (threadIdx.x & (GROUP - 1))];
    // make hipcc's own waits for the loads above land here, not inside the loops (af_im_to_vis.hip)
    asm volatile("" :: "v"(u), "v"(v), "v"(w), "s"(F0), "s"(FD), "s"(NU0SQ), "s"(C1), "s"(C2), "s"(npoint));
#pragma unroll
    for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(R[g]));

#pragma unroll 1
    for (int s = 0; s < npoint; ++s) {
        const int sn = (s + 1 < nsrc) ? s + 1 : s;
        component_pass<CT, false, NG>(acc, R, u, v, w, F0, FD, NU0SQ, C1, C2, lane_off,
                                      rec + (int64_t)sn * (NG * GROUP), ptab);
    }
#pragma unroll 1
    for (int s = npoint; s < nsrc; ++s) {
        const int sn = (s + 1 < nsrc) ? s + 1 : s;
        component_pass<CT, true, NG>(acc, R, u, v, w, F0, FD, NU0SQ, C1, C2, lane_off,
                                     rec + (int64_t)sn * (NG * GROUP), ptab);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // one wave's 64 rows go through LDS per pass and leave as coalesced row segments of CT complex values
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int seg_len = (int)((nchan - c0 < CT) ? (nchan - c0) : CT);
    for (int pass = 0; pass < ROWS_PER_BLOCK / 64; ++pass) {
        if (wave == pass) {
            double2 *dst = stage + lane * STRIDE;
#pragma unroll
            for (int j = 0; j < CT; ++j) dst[j] = make_double2(acc[j][0], acc[j][1]);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * CT; e += ROWS_PER_BLOCK) {
            const int rl = e / CT, col = e - rl * CT;
            const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + pass * 64 + rl;
            if (r < nrow && col < seg_len) reinterpret_cast<double2 *>(out)[r * nchan + c0 + col] = stage[rl * STRIDE + col];
        }
        __syncthreads();
    }
}

// ---- exact kernel: reference operation order (wsclean_predict.py:36-82), given component order ------
// grid: (ceil(nrow/256), ceil(nchan/CHB)); lane = row; channel-outermost so one accumulator pair is live.
constexpr int CHB = 8;
__global__ __launch_bounds__(ROWS_PER_BLOCK) void wsc_exact_kernel(
    const double *__restrict__ uvw, const double *__restrict__ params, const double *__restrict__ spectrum,
    const double *__restrict__ freq, const int *__restrict__ flags, double *__restrict__ out, int64_t nrow,
    int64_t nsrc, int64_t nchan, double gauss_scale, int want_uniform)
{
    if (want_uniform >= 0 && flags[0] != want_uniform) return;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x;
    if (row >= nrow) return;
    const double u = uvw[3 * row], v = uvw[3 * row + 1], w = uvw[3 * row + 2];
    const int64_t f0 = (int64_t)blockIdx.y * CHB;
    for (int64_t f = f0; f < f0 + CHB && f < nchan; ++f) {
        const double nu = freq[f], sf = __dmul_rn(nu, gauss_scale);
        double re_acc = 0.0, im_acc = 0.0;
        for (int64_t s = 0; s < nsrc; ++s) {
            const double *__restrict__ p = params + 8 * s;
            const double real_phase = __dmul_rn(
                AF_TWO_PI_OVER_C, __dadd_rn(__dadd_rn(__dmul_rn(u, p[0]), __dmul_rn(v, p[1])), __dmul_rn(w, p[2])));
            double c, sn;
            sincos_radians(__dmul_rn(real_phase, nu), c, sn);
            const double sp = spectrum[s * nchan + f];
            double re = __dmul_rn(c, sp), im = __dmul_rn(sn, sp);
            if (p[3] != 0.0) {  // wave-uniform
                const double u1 = __dmul_rn(__dsub_rn(__dmul_rn(u, p[5]), __dmul_rn(v, p[4])), p[6]);
                const double v1 = __dadd_rn(__dmul_rn(u, p[4]), __dmul_rn(v, p[5]));
                const double fu1 = __dmul_rn(u1, sf), fv1 = __dmul_rn(v1, sf);
                const double shape = exp(-__dadd_rn(__dmul_rn(fu1, fu1), __dmul_rn(fv1, fv1)));
                re = __dmul_rn(re, shape);
                im = __dmul_rn(im, shape);
            }
            re_acc = __dadd_rn(re_acc, re);
            im_acc = __dadd_rn(im_acc, im);
        }
        reinterpret_cast<double2 *>(out)[row * nchan + f] = make_double2(re_acc, im_acc);
    }
}

// Tile width: setup per (row, component, tile) is ~75 fp64 ops (two sincos) against 4 per channel
// (8 for a Gaussian); 2*CT accumulators + 3 record groups fit the VGPR file comfortably up to 40.
int choose_ct(int64_t nchan)
{
    const int cands[5] = {40, 32, 24, 16, 8};
    int best = cands[0];
    int64_t best_cost = -1;
    for (int k = 0; k < 5; ++k) {
        const int64_t cost = af_cdiv(nchan, cands[k]) * (75 + 4 * (int64_t)cands[k]);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cands[k]; }
    }
    return best;
}

template <int CT>
void launch_recurrence(const WsLayout &L, char *ws, const double *uvw, double *out, int64_t nrow, int64_t nsrc,
                       int64_t nchan, hipStream_t st)
{
    hipLaunchKernelGGL((wsc_recurrence_kernel<CT>), dim3((unsigned)af_cdiv(nrow, ROWS_PER_BLOCK), (unsigned)L.ntile),
                       dim3(ROWS_PER_BLOCK), 0, st, uvw, reinterpret_cast<const double *>(ws + L.records),
                       reinterpret_cast<const double *>(ws + L.tilef), reinterpret_cast<const int *>(ws + L.flags), out,
                       nrow, (int)nsrc, nchan);
}

double gauss_scale_value()
{
    // wsclean_predict.py:13-15
    const double fwhm = 2.0 * sqrt(2.0 * log(2.0));
    const double fwhminv = 1.0 / fwhm;
    return fwhminv * sqrt(2.0) * 3.141592653589793 / AF_LIGHTSPEED;
}

}  // namespace

AF_EXPORT int af_wsclean_spectra_f64(const double *flux, const double *coeffs, const unsigned char *log_poly,
                                     const double *ref_freq, const double *frequency, int64_t nsrc, int64_t ncoeffs,
                                     int64_t nchan, double *out, void *stream)
{
    AF_REQUIRE(nsrc >= 0 && ncoeffs >= 0 && nchan >= 0, "af_wsclean_spectra_f64: negative extent");
    if (nsrc == 0 || nchan == 0) return AF_OK;
    AF_REQUIRE(flux && log_poly && ref_freq && frequency && out && (coeffs || ncoeffs == 0),
               "af_wsclean_spectra_f64: NULL array");
    hipLaunchKernelGGL(wsc_spectra_kernel, dim3((unsigned)af_cdiv(nsrc * nchan, 256)), dim3(256), 0, af_stream(stream),
                       flux, coeffs, log_poly, ref_freq, frequency, nsrc, ncoeffs, nchan, out);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

AF_EXPORT size_t af_wsclean_predict_workspace_bytes(int64_t nsrc, int64_t nchan)
{
    if (nsrc < 0 || nchan < 0) return 0;
    size_t m = 0;
    const int cands[5] = {40, 32, 24, 16, 8};
    for (int k = 0; k < 5; ++k) {
        WsLayout L;
        ws_layout(L, nsrc, nchan, cands[k]);
        if (L.total > m) m = L.total;
    }
    return m;
}

AF_EXPORT int af_wsclean_predict_f64(const double *uvw, const double *lm, const unsigned char *is_gaussian,
                                     const double *flux, const double *coeffs, const unsigned char *log_poly,
                                     const double *ref_freq, const double *gauss_shape, const double *frequency,
                                     int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncoeffs, int mode, double *out,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(mode == AF_DFT_AUTO || mode == AF_DFT_EXACT || mode == AF_DFT_RECURRENCE,
               "af_wsclean_predict_f64: unknown mode %d", mode);
    AF_REQUIRE(nsrc >= 0 && nrow >= 0 && nchan >= 0 && ncoeffs >= 0, "af_wsclean_predict_f64: negative extent");
    AF_REQUIRE(nsrc < (1LL << 31), "af_wsclean_predict_f64: nsrc too large");
    hipStream_t st = af_stream(stream);
    if (nrow == 0 || nchan == 0) return AF_OK;
    AF_REQUIRE(out != nullptr && uvw != nullptr && frequency != nullptr, "af_wsclean_predict_f64: NULL array");
    if (nsrc == 0) {  // np.zeros output (wsclean_predict.py:27)
        AF_HIP(hipMemsetAsync(out, 0, sizeof(double) * 2 * (size_t)(nrow * nchan), st));
        return AF_OK;
    }
    AF_REQUIRE(lm && is_gaussian && flux && log_poly && ref_freq && gauss_shape && (coeffs || ncoeffs == 0),
               "af_wsclean_predict_f64: NULL array");
    const int ct = choose_ct(nchan);
    WsLayout L;
    ws_layout(L, nsrc, nchan, ct);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= L.total,
               "af_wsclean_predict_f64: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_wsclean_predict_f64: workspace must be 256-byte aligned");
    AF_REQUIRE(L.ntile <= 65535 && af_cdiv(nchan, CHB) <= 65535, "af_wsclean_predict_f64: too many channels");
    char *ws = static_cast<char *>(workspace);
    int *flags = reinterpret_cast<int *>(ws + L.flags);
    double *params = reinterpret_cast<double *>(ws + L.params);
    double *spectrum = reinterpret_cast<double *>(ws + L.spectrum);
    const double gs = gauss_scale_value();

    AF_HIP(hipMemsetAsync(flags, 0, 64 * sizeof(int), st));
    AF_HIP(hipMemsetAsync(flags, 1, 1, st));  // flags[0] = 1 until a tile says otherwise
    hipLaunchKernelGGL(wsc_spectra_kernel, dim3((unsigned)af_cdiv(nsrc * nchan, 256)), dim3(256), 0, st, flux, coeffs,
                       log_poly, ref_freq, frequency, nsrc, ncoeffs, nchan, spectrum);
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL(wsc_prep_params, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st, lm, is_gaussian,
                       gauss_shape, nsrc, params);
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL(wsc_prep_freq, dim3((unsigned)af_cdiv(L.ntile, 64)), dim3(64), 0, st, frequency, nchan, L.ntile,
                       ct, reinterpret_cast<double *>(ws + L.tilef), flags);
    AF_LAUNCH_CHECK();
    if (mode == AF_DFT_RECURRENCE) AF_HIP(hipMemsetAsync(flags, 1, 1, st));  // caller asserts uniform spacing

    af_prof_begin(st);
    if (mode != AF_DFT_EXACT) {
        hipLaunchKernelGGL(wsc_order_kernel, dim3(1), dim3(1024), 0, st, is_gaussian, (int)nsrc,
                           reinterpret_cast<int *>(ws + L.order), flags);
        AF_LAUNCH_CHECK();
        int64_t blocks = af_cdiv(L.ntile * nsrc * L.groups * GROUP, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(wsc_pack_records, dim3((unsigned)blocks), dim3(256), 0, st, params, spectrum,
                           reinterpret_cast<const int *>(ws + L.order), nsrc, nchan, L.ntile, ct, L.groups, gs,
                           reinterpret_cast<double *>(ws + L.records));
        AF_LAUNCH_CHECK();
        switch (ct) {
        case 40: launch_recurrence<40>(L, ws, uvw, out, nrow, nsrc, nchan, st); break;
        case 32: launch_recurrence<32>(L, ws, uvw, out, nrow, nsrc, nchan, st); break;
        case 24: launch_recurrence<24>(L, ws, uvw, out, nrow, nsrc, nchan, st); break;
        case 16: launch_recurrence<16>(L, ws, uvw, out, nrow, nsrc, nchan, st); break;
        default: launch_recurrence<8>(L, ws, uvw, out, nrow, nsrc, nchan, st); break;
        }
        AF_LAUNCH_CHECK();
    }
    if (mode != AF_DFT_RECURRENCE) {
        hipLaunchKernelGGL(wsc_exact_kernel, dim3((unsigned)af_cdiv(nrow, ROWS_PER_BLOCK), (unsigned)af_cdiv(nchan, CHB)),
                           dim3(ROWS_PER_BLOCK), 0, st, uvw, params, spectrum, frequency, flags, out, nrow, nsrc, nchan,
                           gs, mode == AF_DFT_EXACT ? -1 : 0);
        AF_LAUNCH_CHECK();
    }
    af_prof_end(st);
    return AF_OK;
}
