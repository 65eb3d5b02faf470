// vis_to_im: adjoint direct Fourier transform visibilities -> image for gfx950.
//
// Replaces africanus/dft/kernels.py:104-146 (nb_vis_to_im.impl):
//     im[s,nu,c] = sum_r [no corr of (r,nu) flagged] cos(p)*vis[r,nu,c].re - sin(p)*vis[r,nu,c].im,
//     p = C*(l_s u_r + m_s v_r + n_s w_r)*nu,  n = sqrt(1-l^2-m^2)-1 unclamped (:125),
//     C = +2pi/c for 'fourier', -2pi/c for 'casa' (:113-118: the opposite of im_to_vis).
//
// cos(p) v.re - sin(p) v.im = Re(v * e^{ip}) and the phase is symmetric in (l,m,n) <-> (u,v,w), so
// this is the real part of the im_to_vis machinery with the roles of rows and sources swapped
// (DESIGN.md "vis_to_im"):
//   * one lane owns one SOURCE and a tile of CT=16 channels x NC=4 correlations of REAL
//     accumulators (64 doubles = 128 VGPRs); the wave walks ROWS.
//   * per (channel tile, row) the prep pass builds a record of 16-double groups
//     [u, v, w, 0, (re, im) of the tile's visibilities ...] with every flagged (row, chan) zeroed
//     (the reference skips a (row, chan) when any of its correlations is flagged, :139-140);
//     groups are read into VGPR lanes and consumed through DPP row_newbcast operands, refreshed
//     in place for the next row with counted vmcnt waits -- exactly as in af_im_to_vis.hip.
//   * phasors by the channel recurrence (uniformly spaced channels) or, for non-uniform
//     frequencies / AF_DFT_EXACT, the reference's operation order with a full-accuracy sincos.
//   * the sum over rows is split into P row partitions (grid.z) for parallelism; each writes a
//     partial image and a small second kernel adds the partials in partition order, so the result
//     is deterministic (no atomics).  A channel none of whose rows is unflagged is forced to
//     exactly 0 (the reference never touches it, even for a NaN source).
#include "af_common.h"
#include "af_sincos.h"
#include "af_dft_device.h"
#include "af_v2i_mfma.h"

namespace {

constexpr int SRC_PER_BLOCK = 256;
constexpr int MAXNC = 4;

struct VWs {
    size_t flags, lmn, tilef, freq, chan_any, records, partial, total;
    int64_t ntile, nchunk, npart, rows_per_part;
    int64_t chunk_off[64];
    int chunk_nc[64], chunk_groups[64];
    int ct;
};

int choose_ct_v(int64_t nchan, int64_t ncorr)
{
    // 16 channels x 4 corr x 8 B = 128 accumulator VGPRs + 18 of record groups + ~58 working: two
    // waves per SIMD (22 channels would need 258 registers and drop to one wave).  One / two correlations
    // afford 64 / 32 channels in the same registers: the ~50-operation set-up per (source, row, tile) is
    // amortised over four / two times as many channels.
    const int cands4[2] = {16, 8}, cands2[3] = {32, 16, 8}, cands1[4] = {64, 32, 16, 8};
    const int *cands = ncorr == 1 ? cands1 : ncorr == 2 ? cands2 : cands4;
    const int ncand = ncorr == 1 ? 4 : ncorr == 2 ? 3 : 2;
    const int64_t per_chan = 2 + 2 * (ncorr < MAXNC ? ncorr : MAXNC);
    int best = cands[0];
    int64_t best_cost = -1;
    for (int k = 0; k < ncand; ++k) {
        int64_t cost = af_cdiv(nchan, cands[k]) * (50 + (int64_t)cands[k] * per_chan);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cands[k]; }
    }
    return best;
}

bool vws_layout(VWs &L, int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr, int CT)
{
    L.ct = CT;
    L.ntile = af_cdiv(nchan > 0 ? nchan : 1, CT);
    L.nchunk = af_cdiv(ncorr > 0 ? ncorr : 1, MAXNC);
    if (L.nchunk > 64) return false;
    // row partitions: ~4096 waves in flight, at least 64 rows each
    const int64_t waves = af_cdiv(nsrc > 0 ? nsrc : 1, 64) * L.ntile;
    int64_t p = af_cdiv(4096, waves);
    const int64_t pmax = af_cdiv(nrow > 0 ? nrow : 1, 64);
    if (p > pmax) p = pmax;
    if (p < 1) p = 1;
    L.rows_per_part = af_cdiv(af_cdiv(nrow > 0 ? nrow : 1, p), 4) * 4;  // whole 4-row steps (MFMA path)
    L.npart = af_cdiv(nrow > 0 ? nrow : 1, L.rows_per_part);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    L.flags = take(64 * sizeof(int));
    L.lmn = take((size_t)nsrc * 4 * sizeof(double));
    L.tilef = take((size_t)L.ntile * 4 * sizeof(double));
    L.freq = take((size_t)L.ntile * CT * sizeof(double));
    L.chan_any = take((size_t)L.ntile * CT * sizeof(int));
    int64_t rec = 0;
    for (int k = 0; k < (int)L.nchunk; ++k) {
        int nc = (int)((ncorr - (int64_t)k * MAXNC < MAXNC) ? (ncorr - (int64_t)k * MAXNC) : MAXNC);
        L.chunk_nc[k] = nc;
        L.chunk_groups[k] = record_groups(CT, nc, 2);
        L.chunk_off[k] = rec;
        rec += L.ntile * nrow * (int64_t)L.chunk_groups[k] * GROUP;
    }
    // one record region: either these records or the MFMA path's (af_vis_to_im_mfma.hip) are built
    size_t rec_bytes = (size_t)rec * sizeof(double);
    if (af_v2i_mfma_eligible(nchan, ncorr)) {
        const size_t mb = af_v2i_mfma_workspace_bytes(nrow, nchan);
        if (mb > rec_bytes) rec_bytes = mb;
    }
    L.records = take(rec_bytes);
    L.partial = take((size_t)L.npart * nsrc * nchan * ncorr * sizeof(double));
    L.total = o;
    return true;
}

// n = sqrt(1 - l^2 - m^2) - 1 in the reference's operation order, unclamped (kernels.py:125)
__global__ void v2i_prep_src(const double *__restrict__ lm, int64_t nsrc, double *__restrict__ lmn)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    double l = lm[2 * s], m = lm[2 * s + 1];
    double n = __dsub_rn(__dsub_rn(1.0, __dmul_rn(l, l)), __dmul_rn(m, m));
    lmn[4 * s + 0] = l;
    lmn[4 * s + 1] = m;
    lmn[4 * s + 2] = __dsub_rn(__dsqrt_rn(n), 1.0);
    lmn[4 * s + 3] = 0.0;
}

// per tile: quarter-turn rates (F0_4, FD_4) and the uniformity flag (see dft_prep_freq)
__global__ void v2i_prep_freq(const double *__restrict__ freq, int64_t nchan, int64_t ntile, int CT, int sign,
                              double *__restrict__ tilef, double *__restrict__ freq_pad, int *__restrict__ flags)
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

// records of one correlation chunk: [tile][row][groups*16] = [u, v, w, 0, (re,im)(j=0,c=0), ...];
// a (row, chan) with ANY flagged correlation (over all ncorr) contributes zeros (kernels.py:139-140).
// chan_any[chan] is set when at least one row of the channel is unflagged.
__global__ void v2i_pack_records(const double2 *__restrict__ vis, const unsigned char *__restrict__ vflags,
                                 const double *__restrict__ uvw, int64_t nrow, int64_t nchan, int64_t ncorr,
                                 int64_t ntile, int CT, int corr0, int nc, int groups, double *__restrict__ rec,
                                 int *__restrict__ chan_any, const int *__restrict__ flags)
{
    if (flags[3] == 1) return;  // the MFMA path owns this call and has filled the record region
    const int64_t per = (int64_t)groups * GROUP;
    const int64_t total = ntile * nrow * per;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t slot = i % per;
        const int64_t r = (i / per) % nrow;
        const int64_t tile = i / (per * nrow);
        double v = 0.0;
        if (slot < 3) {
            v = uvw[3 * r + slot];
        } else if (slot >= 4 && slot < 4 + (int64_t)CT * nc * 2) {
            const int64_t e = slot - 4;
            const int64_t k = e & 1, c = (e >> 1) % nc, j = (e >> 1) / nc;
            const int64_t ch = tile * CT + j;
            if (ch < nchan) {
                bool flagged = false;
                const unsigned char *fl = vflags + (r * nchan + ch) * ncorr;
                for (int64_t cc = 0; cc < ncorr; ++cc) flagged |= (fl[cc] != 0);
                if (!flagged) {
                    const double2 x = vis[(r * nchan + ch) * ncorr + corr0 + c];
                    v = k ? x.y : x.x;
                    if (k == 0 && c == 0 && corr0 == 0) chan_any[ch] = 1;  // benign race: all writers store 1
                }
            }
        }
        rec[i] = v;
    }
}

// ---- recurrence kernel ---------------------------------------------------------------------------
// grid: (ceil(nsrc/256), ntile, npart); block 256 = 4 waves of 64 sources, all on tile blockIdx.y
// and rows [blockIdx.z*rows_per_part, ...).  partial: [part][src][chan][corr] float64.
template <int CT, int NC, int NTERM>
__global__ __launch_bounds__(SRC_PER_BLOCK) void v2i_recurrence_kernel(
    const double *__restrict__ lmn, const double *__restrict__ records, const double *__restrict__ tilef,
    const int *__restrict__ flags, double *__restrict__ partial, int64_t nsrc, int64_t nrow,
    int64_t rows_per_part, int64_t nchan, int64_t ncorr, int64_t corr0, int want_uniform)
{
    if (flags[0] != want_uniform || flags[3] == 1) return;  // flags[3]: the MFMA path did the work
    constexpr int NG = record_groups(CT, NC, 2);
    constexpr int NSLOT = 4 + CT * NC * 2;
    constexpr int PER_CHAN = NC * 2;
    const int tile = blockIdx.y;
    const int64_t c0 = (int64_t)tile * CT;
    int64_t src = (int64_t)blockIdx.x * SRC_PER_BLOCK + threadIdx.x;
    const bool valid = src < nsrc;
    if (!valid) src = nsrc - 1;
    const double l = lmn[4 * src], m = lmn[4 * src + 1], n = lmn[4 * src + 2];
    const double F0 = 64.0 * tilef[4 * tile], FD = 64.0 * tilef[4 * tile + 1];   // quarter turns -> 1/256 turns
    __shared__ double2 ptab[PHASOR_TABLE];
    table_phasor_init(ptab, threadIdx.x, SRC_PER_BLOCK);
    __syncthreads();
    const int64_t r_begin = (int64_t)blockIdx.z * rows_per_part;
    const int64_t r_end = (r_begin + rows_per_part < nrow) ? r_begin + rows_per_part : nrow;

    double acc[CT][NC];
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[j][c] = 0.0;

    const unsigned lane_off = (threadIdx.x & (GROUP - 1)) * (unsigned)sizeof(double);
    const double *__restrict__ rec = records + (int64_t)tile * nrow * (NG * GROUP);
    double R[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) R[g] = rec[r_begin * (NG * GROUP) + g * GROUP + (threadIdx.x & (GROUP - 1))];
    asm volatile("" :: "v"(l), "v"(m), "v"(n), "s"(F0), "s"(FD));
#pragma unroll
    for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(R[g]));

#pragma unroll 1
    for (int64_t r = r_begin; r < r_end; ++r) {
        const int64_t rn = (r + 1 < r_end) ? r + 1 : r;
        const double *rec_next = rec + rn * (NG * GROUP);
        // path difference q = u*l + v*m + w*n: (u,v,w) are slots 0..2 of group 0
        group_wait<group_wait_count(0, NG, NSLOT, PER_CHAN)>(R[0]);
        double q = 0.0;
        fmac_bcast<0>(q, R[0], l);
        fmac_bcast<1>(q, R[0], m);
        fmac_bcast<2>(q, R[0], n);
        if constexpr (group_last_chan(0, NSLOT, PER_CHAN) < 0) group_refresh<0>(R[0], lane_off, rec_next);
        double c0r, c0i, dr, di;
        table_phasor(ptab, __dmul_rn(q, F0), c0r, c0i);   // F0, FD in 1/256 turns per metre (af_sincos.h)
        table_phasor(ptab, __dmul_rn(q, FD), dr, di);
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
                if constexpr (group_first_chan(g, PER_CHAN) == j)
                    group_wait<group_wait_count(g, NG, NSLOT, PER_CHAN)>(R[g]);
            });
            static_for<0, NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int sr = 4 + (j * NC + c) * 2, si = sr + 1;
                // acc += cos(p)*v.re - sin(p)*v.im   (kernels.py:142-146)
                fmac_bcast<sr % GROUP>(acc[j][c], R[sr / GROUP], yr);
                fmac_bcast<si % GROUP, true>(acc[j][c], R[si / GROUP], yi);
            });
            static_for<0, NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if constexpr (group_last_chan(g, NSLOT, PER_CHAN) == j)
                    group_refresh<g * GROUP * (int)sizeof(double)>(R[g], lane_off, rec_next);
            });
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (valid) {
        double *o = partial + (((int64_t)blockIdx.z * nsrc + src) * nchan + c0) * ncorr + corr0;
#pragma unroll
        for (int j = 0; j < CT; ++j)
            if (c0 + j < nchan) {
#pragma unroll
                for (int c = 0; c < NC; ++c) o[j * ncorr + c] = acc[j][c];
            }
    }
}

// ---- exact kernel ------------------------------------------------------------------------------------
// reference operation order without contraction (kernels.py:131,135,142-146), sincos_radians.
template <int CT, int NC>
__global__ __launch_bounds__(SRC_PER_BLOCK) void v2i_exact_kernel(
    const double *__restrict__ lmn, const double *__restrict__ records, const double *__restrict__ freq_pad,
    const int *__restrict__ flags, double *__restrict__ partial, int64_t nsrc, int64_t nrow,
    int64_t rows_per_part, int64_t nchan, int64_t ncorr, int64_t corr0, int want_uniform, double constant)
{
    if (flags[3] == 1 || (want_uniform >= 0 && flags[0] != want_uniform)) return;
    constexpr int NG = record_groups(CT, NC, 2);
    const int tile = blockIdx.y;
    const int64_t c0 = (int64_t)tile * CT;
    int64_t src = (int64_t)blockIdx.x * SRC_PER_BLOCK + threadIdx.x;
    const bool valid = src < nsrc;
    if (!valid) src = nsrc - 1;
    const double l = lmn[4 * src], m = lmn[4 * src + 1], n = lmn[4 * src + 2];
    const int64_t r_begin = (int64_t)blockIdx.z * rows_per_part;
    const int64_t r_end = (r_begin + rows_per_part < nrow) ? r_begin + rows_per_part : nrow;
    const double *__restrict__ rec = records + (int64_t)tile * nrow * (NG * GROUP);
    for (int j = 0; j < CT; ++j) {
        if (c0 + j >= nchan) break;
        const double nu = freq_pad[c0 + j];
        double acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = 0.0;
        for (int64_t r = r_begin; r < r_end; ++r) {
            const double *__restrict__ rs = rec + r * (NG * GROUP);
            const double u = rs[0], v = rs[1], w = rs[2];
            const double real_phase = __dmul_rn(
                constant, __dadd_rn(__dadd_rn(__dmul_rn(l, u), __dmul_rn(m, v)), __dmul_rn(n, w)));
            double cp, sp;
            sincos_radians(__dmul_rn(real_phase, nu), cp, sp);
            const double *__restrict__ g = rs + 4 + j * (NC * 2);
#pragma unroll
            for (int c = 0; c < NC; ++c)
                acc[c] = __dadd_rn(acc[c], __dsub_rn(__dmul_rn(cp, g[2 * c]), __dmul_rn(sp, g[2 * c + 1])));
        }
        if (valid) {
            double *o = partial + (((int64_t)blockIdx.z * nsrc + src) * nchan + c0 + j) * ncorr + corr0;
#pragma unroll
            for (int c = 0; c < NC; ++c) o[c] = acc[c];
        }
    }
}

// im = sum over partitions, in partition order; channels with no unflagged row stay exactly 0
__global__ void v2i_reduce_kernel(const double *__restrict__ partial, const int *__restrict__ chan_any,
                                  int64_t npart, int64_t nsrc, int64_t nchan, int64_t ncorr,
                                  double *__restrict__ out)
{
    const int64_t total = nsrc * nchan * ncorr;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t ch = (i / ncorr) % nchan;
    double s = 0.0;
    for (int64_t p = 0; p < npart; ++p) s += partial[p * total + i];
    out[i] = chan_any[ch] ? s : 0.0;
}

struct VArgs {
    const VWs *L;
    char *ws;
    int64_t nsrc, nrow, nchan, ncorr;
    int chunk, mode;
    double constant;
    hipStream_t st;
};

template <int CT, int NC>
int v2i_launch(const VArgs &a)
{
    const VWs &L = *a.L;
    char *ws = a.ws;
    const double *lmn = reinterpret_cast<const double *>(ws + L.lmn);
    const double *tilef = reinterpret_cast<const double *>(ws + L.tilef);
    const double *freq_pad = reinterpret_cast<const double *>(ws + L.freq);
    const int *flags = reinterpret_cast<const int *>(ws + L.flags);
    const double *records = reinterpret_cast<const double *>(ws + L.records) + L.chunk_off[a.chunk];
    double *partial = reinterpret_cast<double *>(ws + L.partial);
    dim3 grid((unsigned)af_cdiv(a.nsrc, SRC_PER_BLOCK), (unsigned)L.ntile, (unsigned)L.npart), block(SRC_PER_BLOCK);
    const int64_t corr0 = (int64_t)a.chunk * MAXNC;
    if (a.mode == AF_DFT_AUTO || a.mode == AF_DFT_RECURRENCE) {
        if (a.chunk == 0) af_prof_begin(a.st);
        hipLaunchKernelGGL((v2i_recurrence_kernel<CT, NC, 7>), grid, block, 0, a.st, lmn, records, tilef, flags,
                           partial, a.nsrc, a.nrow, L.rows_per_part, a.nchan, a.ncorr, corr0, 1);
        if (a.chunk == 0) af_prof_end(a.st);
        AF_LAUNCH_CHECK();
    }
    if (a.mode == AF_DFT_AUTO || a.mode == AF_DFT_EXACT) {
        hipLaunchKernelGGL((v2i_exact_kernel<CT, NC>), grid, block, 0, a.st, lmn, records, freq_pad, flags, partial,
                           a.nsrc, a.nrow, L.rows_per_part, a.nchan, a.ncorr, corr0,
                           a.mode == AF_DFT_EXACT ? -1 : 0, a.constant);
        AF_LAUNCH_CHECK();
    }
    return AF_OK;
}

template <int CT>
int v2i_launch_nc(int nc, const VArgs &a)
{
    switch (nc) {
    case 1: return v2i_launch<CT, 1>(a);
    case 2: return v2i_launch<CT, 2>(a);
    case 3: return v2i_launch<CT, 3>(a);
    default: return v2i_launch<CT, 4>(a);
    }
}

}  // namespace

AF_EXPORT size_t af_vis_to_im_workspace_bytes(int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr)
{
    if (nsrc < 0 || nrow < 0 || nchan < 0 || ncorr < 0) return 0;
    size_t m = 0;
    const int cands[4] = {8, 16, 32, 64};
    for (int k = 0; k < 4; ++k) {
        VWs L;
        if (!vws_layout(L, nsrc, nrow, nchan, ncorr, cands[k])) return 0;
        if (L.total > m) m = L.total;
    }
    return m;
}

AF_EXPORT int af_vis_to_im_f64(const double *vis, const double *uvw, const double *lm, const double *frequency,
                               const unsigned char *flags, int64_t nsrc, int64_t nrow, int64_t nchan,
                               int64_t ncorr, int convention, int mode, double *out, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(convention == AF_CONVENTION_FOURIER || convention == AF_CONVENTION_CASA,
               "convention not in ('fourier', 'casa')");
    AF_REQUIRE(mode == AF_DFT_AUTO || mode == AF_DFT_EXACT || mode == AF_DFT_RECURRENCE,
               "af_vis_to_im_f64: unknown mode %d", mode);
    AF_REQUIRE(nsrc >= 0 && nrow >= 0 && nchan >= 0 && ncorr >= 0, "af_vis_to_im_f64: negative extent");
    AF_REQUIRE(ncorr <= 64 * MAXNC, "af_vis_to_im_f64: more than %d correlations", 64 * MAXNC);
    hipStream_t st = af_stream(stream);
    if (nsrc == 0 || nchan == 0 || ncorr == 0) return AF_OK;
    AF_REQUIRE(out != nullptr, "af_vis_to_im_f64: out is NULL");
    if (nrow == 0) {  // np.zeros output (kernels.py:120)
        AF_HIP(hipMemsetAsync(out, 0, sizeof(double) * (size_t)(nsrc * nchan * ncorr), st));
        return AF_OK;
    }
    AF_REQUIRE(vis && uvw && lm && frequency && flags, "af_vis_to_im_f64: NULL array");
    const int ct = choose_ct_v(nchan, ncorr);
    VWs L;
    vws_layout(L, nsrc, nrow, nchan, ncorr, ct);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= L.total,
               "af_vis_to_im_f64: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_vis_to_im_f64: workspace must be 256-byte aligned");
    AF_REQUIRE(L.ntile <= 65535 && L.npart <= 65535, "af_vis_to_im_f64: problem too large for one launch");
    char *ws = static_cast<char *>(workspace);

    AF_HIP(hipMemsetAsync(ws + L.flags, 0, 64 * sizeof(int), st));
    AF_HIP(hipMemsetAsync(ws + L.flags, 1, 1, st));
    AF_HIP(hipMemsetAsync(ws + L.chan_any, 0, (size_t)L.ntile * ct * sizeof(int), st));
    hipLaunchKernelGGL(v2i_prep_src, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st, lm, nsrc,
                       reinterpret_cast<double *>(ws + L.lmn));
    AF_LAUNCH_CHECK();
    // vis_to_im's 'fourier' is exp(+i...): the opposite sign of im_to_vis (kernels.py:113-118)
    hipLaunchKernelGGL(v2i_prep_freq, dim3((unsigned)af_cdiv(L.ntile, 64)), dim3(64), 0, st, frequency, nchan,
                       L.ntile, ct, -convention, reinterpret_cast<double *>(ws + L.tilef),
                       reinterpret_cast<double *>(ws + L.freq), reinterpret_cast<int *>(ws + L.flags));
    AF_LAUNCH_CHECK();
    if (mode == AF_DFT_RECURRENCE) AF_HIP(hipMemsetAsync(ws + L.flags, 1, 1, st));
    if (mode != AF_DFT_EXACT && af_v2i_mfma_eligible(nchan, ncorr)) {
        // flags[3] = 1: the MFMA path owns the call unless its frequency test clears the flag on the device
        AF_HIP(hipMemsetAsync(ws + L.flags + 3 * sizeof(int), 1, 1, st));
        int rc = af_v2i_mfma_run(vis, flags, uvw, frequency, reinterpret_cast<const double *>(ws + L.lmn),
                                 reinterpret_cast<int *>(ws + L.flags), reinterpret_cast<int *>(ws + L.chan_any),
                                 -convention, reinterpret_cast<double *>(ws + L.partial), nsrc, nrow, nchan, L.npart,
                                 L.rows_per_part, mode == AF_DFT_RECURRENCE, ws + L.records, st);
        if (rc != AF_OK) return rc;
    }
    for (int chunk = 0; chunk < (int)L.nchunk; ++chunk) {
        const int64_t total = L.ntile * nrow * (int64_t)L.chunk_groups[chunk] * GROUP;
        int64_t blocks = af_cdiv(total, 256);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(v2i_pack_records, dim3((unsigned)blocks), dim3(256), 0, st,
                           reinterpret_cast<const double2 *>(vis), flags, uvw, nrow, nchan, ncorr, L.ntile, ct,
                           chunk * MAXNC, L.chunk_nc[chunk], L.chunk_groups[chunk],
                           reinterpret_cast<double *>(ws + L.records) + L.chunk_off[chunk],
                           reinterpret_cast<int *>(ws + L.chan_any), reinterpret_cast<const int *>(ws + L.flags));
        AF_LAUNCH_CHECK();
    }
    VArgs a;
    a.L = &L; a.ws = ws; a.nsrc = nsrc; a.nrow = nrow; a.nchan = nchan; a.ncorr = ncorr; a.mode = mode; a.st = st;
    a.constant = convention == AF_CONVENTION_FOURIER ? AF_TWO_PI_OVER_C : AF_MINUS_TWO_PI_OVER_C;
    for (int chunk = 0; chunk < (int)L.nchunk; ++chunk) {
        a.chunk = chunk;
        int rc = ct == 64 ? v2i_launch<64, 1>(a)   // the wide tiles exist for the correlation counts choose_ct_v offers them to
                 : ct == 32 ? (ncorr == 1 ? v2i_launch<32, 1>(a) : v2i_launch<32, 2>(a))
                 : ct == 8 ? v2i_launch_nc<8>(L.chunk_nc[chunk], a) : v2i_launch_nc<16>(L.chunk_nc[chunk], a);
        if (rc != AF_OK) return rc;
    }
    {
        const int64_t total = nsrc * nchan * ncorr;
        hipLaunchKernelGGL(v2i_reduce_kernel, dim3((unsigned)af_cdiv(total, 256)), dim3(256), 0, st,
                           reinterpret_cast<const double *>(ws + L.partial),
                           reinterpret_cast<const int *>(ws + L.chan_any), L.npart, nsrc, nchan, ncorr, out);
        AF_LAUNCH_CHECK();
    }
    return AF_OK;
}
