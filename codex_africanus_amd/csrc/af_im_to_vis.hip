// im_to_vis: direct Fourier transform image -> visibilities for gfx950.
//
// Replaces africanus/dft/kernels.py:33-67 (nb_im_to_vis.impl):
//     vis[r,nu,c] = sum_s exp(i*C*(l_s u_r + m_s v_r + n_s w_r)*nu) * image[s,nu,c]
// with n = sqrt(1-l^2-m^2)-1 unclamped (:54) and zero pixels skipped (:64).
//
// Design ("im_to_vis"):
//   * one lane owns one row and a tile of CT channels x NC correlations of complex
//     accumulators in VGPRs (CT=13, NC=4 -> 104 doubles = 208 VGPRs); the wave walks all sources.
//     The kernel is fp64-VALU bound (nsrc phasors per 64-byte visibility), so everything is
//     organised to keep the inner loop pure v_fma_f64 / v_fmac_f64.
//   * per (channel tile, source) the prep pass builds one RECORD of 16-double groups:
//     [l, m, n, 0, image pixels of the tile...].  A wave reads a record with one
//     global_load_dwordx2 per group (lane i fetches double i%16, so every row of 16 lanes
//     holds the group) and feeds the values to the FMAs through the 64-bit DPP operand
//     `row_newbcast:k` of v_fmac_f64 -- a wave-uniform scalar operand at zero instruction
//     cost, without scalar loads (whose only wait is lgkmcnt(0)) and without LDS.  Each
//     group register is refreshed for the NEXT source right after its last use, so the
//     vector loads have most of an iteration (~600 cycles) to land.
//   * recurrence (uniformly spaced channels): per (row, source, tile) two quarter-turn
//     reduced polynomial sincos give the phasor at the tile's first channel and the
//     channel-to-channel rotation; the other channels follow from the three-term
//     recurrence y[j+1] = 2cos(d)*y[j] - y[j-1] (1 FMA per component).
//   * when the whole band has one channel spacing, tiles are processed four at a time by the four
//     waves of a workgroup on the SAME 64 rows: q and the channel-step phasor of a (row, source)
//     are tile-independent, so each wave computes them for one source of a batch of four and
//     publishes them in LDS (one barrier per four sources).
//   * exact kernel: the reference's operation order (no contraction) and a full-accuracy
//     sincos per (row, source, channel); for non-uniform frequencies and AF_DFT_EXACT.
//   * the prep pass (tiny kernels, no host sync) computes n per source, builds the records,
//     decides uniformity on the device and records the reference's zero-pixel / NaN-source
//     semantics per (channel, corr) column.
#include <type_traits>

#include "af_common.h"
#include "af_sincos.h"
#include "af_dft_device.h"
#include "af_dft_mfma.h"

namespace {

constexpr int ROWS_PER_BLOCK = 256;
#ifndef AF_WIDE_WAVES
#define AF_WIDE_WAVES 3
#endif
constexpr int WIDE_WAVES = AF_WIDE_WAVES;
constexpr int MAXNC = 4;  // correlations per launch (ncorr is processed in chunks)

// ---- workspace layout ------------------------------------------------------------
struct WsLayout {
    size_t flags;     // int[64]          [0] uniform
    size_t lmn;       // double[nsrc*4]   (l, m, n, 0), non-finite sources zeroed
    size_t srcbad;    // int[nsrc]        1 if (l,m,n) is not finite
    size_t tilef;     // double[ntile*4]  (F0_4, FD_4, 0, 0): quarter-turns per metre
    size_t freq;      // double[ntile*CT] channel frequencies, padded per tile
    size_t colstate;  // int[ntile*CT*ncorr] 0 normal, 1 force zero, 2 force NaN
    size_t tilestate; // int[ntile*nchunk] OR of colstate in the tile/chunk
    size_t records;   // double: chunk-major [chunk][tile][src_pad][groups(chunk)*16]
    size_t lgroups;   // double[nsrc_pad/4][16]: [l,m,n,0] of the 4 sources of a batch
    size_t total;
    int64_t ntile, nchunk, nsrc_pad;
    int ct, w;
    int64_t chunk_off[64];  // offset (in doubles) of each chunk's records
    int chunk_nc[64], chunk_groups[64];
};

bool ws_layout(WsLayout &L, int64_t nsrc, int64_t nchan, int64_t ncorr, int is_complex, int CT)
{
    L.ct = CT;
    L.w = is_complex ? 2 : 1;
    L.ntile = af_cdiv(nchan > 0 ? nchan : 1, CT);
    L.nsrc_pad = af_cdiv(nsrc > 0 ? nsrc : 1, 4) * 4;  // whole batches of 4 sources (zero records)
    L.nchunk = af_cdiv(ncorr > 0 ? ncorr : 1, MAXNC);
    if (L.nchunk > 64) return false;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    L.flags = take(64 * sizeof(int));
    L.lmn = take((size_t)nsrc * 4 * sizeof(double));
    L.srcbad = take((size_t)nsrc * sizeof(int));
    L.tilef = take((size_t)L.ntile * 4 * sizeof(double));
    L.freq = take((size_t)L.ntile * CT * sizeof(double));
    L.colstate = take((size_t)L.ntile * CT * ncorr * sizeof(int));
    L.tilestate = take((size_t)L.ntile * L.nchunk * sizeof(int));
    int64_t rec = 0;
    for (int k = 0; k < (int)L.nchunk; ++k) {
        int nc = (int)((ncorr - (int64_t)k * MAXNC < MAXNC) ? (ncorr - (int64_t)k * MAXNC) : MAXNC);
        L.chunk_nc[k] = nc;
        L.chunk_groups[k] = record_groups(CT, nc, L.w);
        L.chunk_off[k] = rec;
        rec += L.ntile * L.nsrc_pad * (int64_t)L.chunk_groups[k] * GROUP;
    }
    L.records = take((size_t)rec * sizeof(double));
    L.lgroups = take((size_t)L.nsrc_pad * 4 * sizeof(double));
    L.total = o;
    return true;
}

// ---- prep kernels ------------------------------------------------------------------
// n = sqrt(1 - l^2 - m^2) - 1 in the reference's operation order (kernels.py:54).
__global__ void dft_prep_src(const double *__restrict__ lm, int64_t nsrc, int clamp_n,
                             double *__restrict__ lmn, int *__restrict__ srcbad)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    double l = lm[2 * s], m = lm[2 * s + 1];
    double n = __dsub_rn(__dsub_rn(1.0, __dmul_rn(l, l)), __dmul_rn(m, m));
    if (clamp_n && n < 0.0) n = 0.0;  // phase_delay's form (africanus/rime/phase.py:43)
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
//
// flags[1] carries two promises.  (a) every tile has tile 0's channel spacing, so the channel-step phasor of a
// (row, source) can be shared by all tiles (dpp4 kernel).  (b) when the MFMA-accumulator kernels own the band
// (mfma_ct = their tile width, else 0): every MFMA tile is ONE arithmetic progression -- dft_mfma_kernel
// evaluates a whole tile as nu[c0] + j * dnu, so a jump between two of THESE tiles that falls inside an MFMA
// tile (concatenated spectral windows with equal channel widths) must clear the flag as well; a jump on an
// MFMA tile boundary is harmless, every MFMA tile takes its own first frequency.
__global__ void dft_prep_freq(const double *__restrict__ freq, int64_t nchan, int64_t ntile, int CT, int sign,
                              int mfma_ct, double *__restrict__ tilef, double *__restrict__ freq_pad,
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
    // flags[1]: every tile has tile 0's channel spacing (4 ulp), so the channel-step phasor of a
    // (row, source) can be shared by all tiles
    const int64_t n0 = nchan < CT ? nchan : CT;
    const double df0 = (n0 > 1) ? (freq[n0 - 1] - freq[0]) / (double)(n0 - 1) : 0.0;
    if (nc > 1 && !(fabs(df - df0) <= 4.0 * 2.220446049250313e-16 * fmax(fabs(df), fabs(df0)) * (double)CT))
        atomicAnd(&flags[1], 0);
    if (mfma_ct > 0 && t > 0 && (c0 % mfma_ct) != 0) {
        // the step INTO this tile continues the progression (differences of ~1e9 Hz values: 4 ulp of nu)
        const double step = f0 - freq[c0 - 1];
        const double tol = 4.0 * 2.220446049250313e-16 * fmax(fabs(f0), fabs(freq[c0 - 1]));
        if (!(fabs(step - df0) <= tol)) atomicAnd(&flags[1], 0);
    }
}

// Build the records of one correlation chunk: for every (tile, source) a block of
// groups*16 doubles = [l, m, n, 0, pixel(j=0,c=0)[.re,.im], pixel(0,1), ..., zero padding].
// Pixels beyond nchan and pixels of sources whose (l,m,n) is not finite are zero (the
// effect of such sources is applied through colstate).
__global__ void dft_pack_records(const double *__restrict__ image, int W, int64_t nsrc, int64_t nsrc_pad,
                                 int64_t nchan, int64_t ncorr, int64_t ntile, int CT, int corr0, int nc,
                                 int groups, const double *__restrict__ lmn, const int *__restrict__ srcbad,
                                 double *__restrict__ rec)
{
    const int64_t per = (int64_t)groups * GROUP;
    const int64_t total = ntile * nsrc_pad * per;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t slot = i % per;
        const int64_t s = (i / per) % nsrc_pad;
        const int64_t tile = i / (per * nsrc_pad);
        double v = 0.0;
        if (s >= nsrc) {
            // padding source: zero (l,m,n) and zero pixels contribute exactly nothing
        } else if (slot < 3) {
            v = lmn[4 * s + slot];
        } else if (slot >= 4 && slot < 4 + (int64_t)CT * nc * W) {
            const int64_t e = slot - 4;
            const int64_t k = e % W, c = (e / W) % nc, j = e / (W * nc);
            const int64_t ch = tile * CT + j;
            if (ch < nchan && !srcbad[s]) v = image[((s * nchan + ch) * ncorr + corr0 + c) * W + k];
        }
        rec[i] = v;
    }
}

// (l,m,n,0) of the 4 sources of every batch, contiguous (16 doubles per batch)
__global__ void dft_pack_lgroups(const double *__restrict__ lmn, int64_t nsrc, int64_t nsrc_pad,
                                 double *__restrict__ lg)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nsrc_pad * 4) lg[i] = (i < nsrc * 4) ? lmn[i] : 0.0;
}

// Column state per (chan, corr): reference semantics of `if image[s,nu,c]:` (kernels.py:64)
//   all pixels zero                         -> the output stays exactly 0 (state 1)
//   a non-finite source has a nonzero pixel -> every row gets NaN there (state 2)
// One wave per column: lanes stride the sources, then a wave-wide OR.
__global__ __launch_bounds__(64) void dft_colstate(const double *__restrict__ image, int W, int64_t nsrc,
                                                   int64_t nchan, int64_t ncorr, int64_t ntile, int CT,
                                                   const int *__restrict__ srcbad, int *__restrict__ colstate,
                                                   int *__restrict__ tilestate, int nchunk, int *__restrict__ flags)
{
    const int64_t i = blockIdx.x;  // column (padded chan, corr)
    const int64_t c = i % ncorr, ch = i / ncorr;
    int state = 0;
    if (ch < nchan) {
        bool any_nz = false, poison = false;
        for (int64_t s = threadIdx.x; s < nsrc; s += 64) {
            const double *px = image + ((s * nchan + ch) * ncorr + c) * W;
            bool nz = (px[0] != 0.0) || (W == 2 && px[1] != 0.0);
            any_nz |= nz;
            poison |= (nz && srcbad[s]);
        }
        const bool w_nz = __ballot(any_nz) != 0ULL, w_poison = __ballot(poison) != 0ULL;
        state = w_poison ? 2 : (w_nz ? 0 : 1);
    }
    if (threadIdx.x == 0) {
        colstate[i] = state;
        if (state) {
            atomicOr(&tilestate[(ch / CT) * nchunk + c / MAXNC], state);
            atomicOr(&flags[2], state);  // any special column at all
        }
    }
}

// ---- epilogue of the recurrence kernel --------------------------------------------
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

// ---- block-cooperative epilogue of the four-tile kernel ------------------------------------------
// The block's 64 rows x 4 tiles form, per row, one contiguous segment of 4*CT*NC complex values
// (3328 B = 26 full 128-byte lines at CT=13, NC=4, line-aligned because tile quads start at
// multiples of it).  Per-lane 16-byte stores at a 4 KiB lane stride leave L2 with partially written
// lines (measured 2.3x write amplification); instead the rows go through LDS 16 at a time and
// leave as fully coalesced stores.  `stage` rows are padded by one double2 so the 16 active lanes
// of a wave hit distinct banks.
constexpr int EPI_ROWS = 16;
template <int CT, int NC>
__device__ __forceinline__ void store_quad_coop(const double (&acc)[CT][NC][2], double2 *stage, double *out,
                                                int64_t row0, int64_t nrow, int64_t nchan, int64_t ncorr,
                                                int64_t c0_block, int wave, int lane)
{
    constexpr int PER_TILE = CT * NC;           // complex values per (row, tile)
    constexpr int PER_ROW = 4 * PER_TILE;       // complex values per row segment
    constexpr int STRIDE = PER_ROW + 1;         // padded LDS row
    const int64_t seg_chans = (nchan - c0_block < 4 * CT) ? (nchan - c0_block) : 4 * CT;
    const int seg_len = (int)(seg_chans * NC);  // valid complex values of the segment (last quad may be short)
    for (int pass = 0; pass < 64 / EPI_ROWS; ++pass) {
        if ((lane / EPI_ROWS) == pass) {
            double2 *dst = stage + (lane % EPI_ROWS) * STRIDE + wave * PER_TILE;
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) dst[j * NC + c] = make_double2(acc[j][c][0], acc[j][c][1]);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < EPI_ROWS * PER_ROW; e += ROWS_PER_BLOCK) {
            const int rl = e / PER_ROW, col = e - rl * PER_ROW;
            const int64_t row = row0 + pass * EPI_ROWS + rl;
            if (row < nrow && col < seg_len)
                reinterpret_cast<double2 *>(out)[(row * nchan + c0_block) * ncorr + col] = stage[rl * STRIDE + col];
        }
        __syncthreads();
    }
}

// The same for the one-tile kernel: the block's 4 waves hold 64 rows each; one wave's rows go
// through LDS per pass (every lane writes its CT*NC values), then all 256 lanes store the 64 row
// segments of CT*NC*16 bytes (832 B at CT=13, NC=4) coalesced.
template <int CT, int NC>
__device__ __forceinline__ void store_tile_coop(const double (&acc)[CT][NC][2], double2 *stage, double *out,
                                                int64_t row0, int64_t nrow, int64_t nchan, int64_t ncorr,
                                                int64_t c0, int wave, int lane)
{
    constexpr int PER_ROW = CT * NC;
    constexpr int STRIDE = PER_ROW + 1;
    const int64_t seg_chans = (nchan - c0 < CT) ? (nchan - c0) : CT;
    const int seg_len = (int)(seg_chans * NC);
    for (int pass = 0; pass < ROWS_PER_BLOCK / 64; ++pass) {
        if (wave == pass) {
            double2 *dst = stage + lane * STRIDE;
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) dst[j * NC + c] = make_double2(acc[j][c][0], acc[j][c][1]);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * PER_ROW; e += ROWS_PER_BLOCK) {
            const int rl = e / PER_ROW, col = e - rl * PER_ROW;
            const int64_t row = row0 + pass * 64 + rl;
            if (row < nrow && col < seg_len)
                reinterpret_cast<double2 *>(out)[(row * nchan + c0) * ncorr + col] = stage[rl * STRIDE + col];
        }
        __syncthreads();
    }
}

// ---- one source's pass over the lane's channel tile ------------------------------------------
// Given the source's path difference q (metres) and channel-step phasor (dr, di) for this row:
// phasor at the tile's first channel, three-term recurrence over the tile, accumulate through the
// DPP operands, and refresh every record group right after its last use.  EXTRA = vector loads
// (besides the record refreshes) issued between the previous source's refreshes and this pass;
// WAIT0 = group 0 has not been retired by the caller yet.
template <int CT, int NC, bool CPLX, int NTERM, int EXTRA, bool WAIT0, int NG, bool TABLE>
__device__ __forceinline__ void channel_pass(double (&acc)[CT][NC][2], double (&R)[NG], double q, double dr,
                                             double di, double F0, unsigned lane_off, const double *rec_next,
                                             const double2 *ptab)
{
    constexpr int W = CPLX ? 2 : 1;
    constexpr int NSLOT = 4 + CT * NC * W;
    constexpr int PER_CHAN = NC * W;
    double c0r, c0i;
    // TABLE: F0 in 1/256 turns per metre, phasor from the block's LDS table (af_sincos.h); else quarter turns and
    // the polynomial pair (the four-tile kernel is at its register limit with 104 accumulators: the table's extra
    // live values would push it to one wave per SIMD)
    if constexpr (TABLE) table_phasor(ptab, __dmul_rn(q, F0), c0r, c0i);
    else sincos_quarter_turns<NTERM>(__dmul_rn(q, F0), c0r, c0i);
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
        // groups first touched by this channel: retire their refresh of the previous source
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr bool first_here = (g == 0) ? (WAIT0 && j == 0) : (group_first_chan(g, PER_CHAN) == j);
            if constexpr (first_here) group_wait<group_wait_count(g, NG, NSLOT, PER_CHAN) + EXTRA>(R[g]);
        });
        static_for<0, NC>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (CPLX) {
                constexpr int sr = 4 + (j * NC + c) * 2, si = sr + 1;
                fmac_bcast<sr % GROUP>(acc[j][c][0], R[sr / GROUP], yr);
                fmac_bcast<si % GROUP, true>(acc[j][c][0], R[si / GROUP], yi);
                fmac_bcast<si % GROUP>(acc[j][c][1], R[si / GROUP], yr);
                fmac_bcast<sr % GROUP>(acc[j][c][1], R[sr / GROUP], yi);
            } else {
                constexpr int sr = 4 + j * NC + c;
                fmac_bcast<sr % GROUP>(acc[j][c][0], R[sr / GROUP], yr);
                fmac_bcast<sr % GROUP>(acc[j][c][1], R[sr / GROUP], yi);
            }
        });
        // refresh, for the next source, every group whose last slot this channel consumed
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (group_last_chan(g, NSLOT, PER_CHAN) == j)
                group_refresh<g * GROUP * (int)sizeof(double)>(R[g], lane_off, rec_next);
        });
    });
}

// ---- recurrence kernel: one channel tile per workgroup ----------------------------------------------
// grid: (ceil(nrow/256), tiles); block 256 = 4 waves, each wave 64 consecutive rows, all on tile
// tile0 + blockIdx.y.  Every lane computes q and both sincos itself.  Runs iff flags[0] ==
// want_uniform and (want_global < 0 or flags[1] == want_global).
template <int CT, int NC, bool CPLX, int NTERM>
__global__ __launch_bounds__(ROWS_PER_BLOCK) void dft_recurrence_dpp_kernel(
    const double *__restrict__ uvw, const double *__restrict__ records, const double *__restrict__ tilef,
    const int *__restrict__ flags, const int *__restrict__ colstate, const int *__restrict__ tilestate,
    double *__restrict__ out, int64_t nrow, int nsrc, int nsrc_pad, int64_t nchan, int64_t ncorr, int64_t corr0,
    int chunk, int nchunk, int tile0, int want_uniform, int want_global)
{
    if (flags[0] != want_uniform) return;  // decided on the device by dft_prep_freq
    if (want_global >= 0 && flags[1] != want_global) return;
    __shared__ double2 stage[64 * (CT * NC + 1)];
    __shared__ double2 ptab[PHASOR_TABLE];
    table_phasor_init(ptab, threadIdx.x, ROWS_PER_BLOCK);
    __syncthreads();
    constexpr int W = CPLX ? 2 : 1;
    constexpr int NG = record_groups(CT, NC, W);
    constexpr int NSLOT = 4 + CT * NC * W;
    constexpr int PER_CHAN = NC * W;
    const int tile = tile0 + blockIdx.y;
    const int64_t c0 = (int64_t)tile * CT;
    int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x;
    const bool valid = row < nrow;
    if (!valid) row = nrow - 1;
    const double u = uvw[3 * row], v = uvw[3 * row + 1], w = uvw[3 * row + 2];
    const double F0 = 64.0 * tilef[4 * tile], FD = 64.0 * tilef[4 * tile + 1];   // quarter turns -> 1/256 turns

    double acc[CT][NC][2];
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[j][c][0] = acc[j][c][1] = 0.0;

    // this lane's double of every group of the (tile, source) record
    const unsigned lane_off = (threadIdx.x & (GROUP - 1)) * (unsigned)sizeof(double);
    const double *__restrict__ rec = records + (int64_t)tile * nsrc_pad * (NG * GROUP);
    double R[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) R[g] = rec[g * GROUP + (threadIdx.x & (GROUP - 1))];
    // Touch everything loaded so far, so that hipcc's own s_waitcnt for these (counted) loads
    // lands here and not at their first use inside the loop, where a vmcnt(0) would drain the
    // asm-issued refreshes every iteration.
    asm volatile("" :: "v"(u), "v"(v), "v"(w), "s"(F0), "s"(FD));
#pragma unroll
    for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(R[g]));

#pragma unroll 1
    for (int s = 0; s < nsrc; ++s) {
        const int sn = (s + 1 < nsrc) ? s + 1 : s;
        const double *rec_next = rec + (int64_t)sn * (NG * GROUP);
        // path difference in metres, q = l*u + m*v + n*w  (slots 0, 1, 2 of group 0)
        group_wait<group_wait_count(0, NG, NSLOT, PER_CHAN)>(R[0]);
        double q = 0.0;
        fmac_bcast<0>(q, R[0], u);
        fmac_bcast<1>(q, R[0], v);
        fmac_bcast<2>(q, R[0], w);
        if constexpr (group_last_chan(0, NSLOT, PER_CHAN) < 0) group_refresh<0>(R[0], lane_off, rec_next);
        double dr, di;
        table_phasor(ptab, __dmul_rn(q, FD), dr, di);
        channel_pass<CT, NC, CPLX, NTERM, 0, false, NG, true>(acc, R, q, dr, di, F0, lane_off, rec_next, ptab);
    }
    // retire the (redundant) refreshes of the last iteration before the registers die
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int tstate = tilestate[tile * nchunk + chunk];
    if (NC == ncorr && tstate == 0) {
        store_tile_coop<CT, NC>(acc, stage, out, (int64_t)blockIdx.x * ROWS_PER_BLOCK, nrow, nchan, ncorr, c0,
                                __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63);
    } else {
        store_tile<CT, NC>(acc, out, row, valid, nchan, ncorr, c0, corr0, colstate, tstate);
    }
}

// ---- recurrence kernel: four channel tiles per workgroup, shared channel-step phasor -------------------
// grid: (ceil(nrow/64), tiles/4); block 256 = 4 waves on the SAME 64 rows, wave w on tile
// 4*blockIdx.y + w.  With one channel spacing for the whole band (flags[1]) the path difference q and
// the channel-step phasor (dr, di) of a (row, source) are the same for every tile, so the four waves
// split that work: per batch of 4 sources wave w computes them for source 4b+w and publishes them
// in LDS; every wave then runs the four sources of the batch on its own tile.  One barrier per batch.
// (l,m,n) of a batch come from `lgroups` (16 doubles per batch: [l,m,n,0] x 4), loaded rotated by
// 4*w lanes so that row_newbcast:0..2 hands wave w its own source.
// (the wide tiles of 1 / 2 correlations -- 32 x 1, 16 x 2 -- need 174 registers: asked to fit 168, three waves share a
// SIMD instead of two; measured at C2's counts, same box: 2 correlations 18.87 -> 18.01 ms, 1 correlation 11.38 -> 11.40)
template <int CT, int NC, bool CPLX, int NTERM>
__global__ __launch_bounds__(ROWS_PER_BLOCK, (!CPLX && CT >= 16 && CT * NC <= 32) ? WIDE_WAVES : 1) void dft_recurrence_dpp4_kernel(
    const double *__restrict__ uvw, const double *__restrict__ records, const double *__restrict__ lgroups,
    const double *__restrict__ tilef, const int *__restrict__ flags, const int *__restrict__ colstate,
    const int *__restrict__ tilestate, double *__restrict__ out, int64_t nrow, int nsrc_pad, int64_t nchan,
    int64_t ncorr, int64_t corr0, int chunk, int nchunk)
{
    if (flags[0] != 1 || flags[1] != 1) return;
    constexpr int W = CPLX ? 2 : 1;
    constexpr int NG = record_groups(CT, NC, W);
    __shared__ double xch[2][4][3][64];
    __shared__ double2 stage[EPI_ROWS * (4 * CT * NC + 1)];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile = 4 * blockIdx.y + wave;
    const int64_t c0 = (int64_t)tile * CT;
    int64_t row = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = row < nrow;
    if (!valid) row = nrow - 1;
    const double u = uvw[3 * row], v = uvw[3 * row + 1], w = uvw[3 * row + 2];
    const double F0 = tilef[4 * tile], FD = tilef[1];  // one channel spacing for the whole band

    double acc[CT][NC][2];
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[j][c][0] = acc[j][c][1] = 0.0;

    const unsigned lane_off = (lane & (GROUP - 1)) * (unsigned)sizeof(double);
    const unsigned lg_off = ((lane + 4 * wave) & (GROUP - 1)) * (unsigned)sizeof(double);
    const double *__restrict__ rec = records + (int64_t)tile * nsrc_pad * (NG * GROUP);
    const int nb = nsrc_pad >> 2;
    double R[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) R[g] = rec[g * GROUP + (lane & (GROUP - 1))];
    double Lg = lgroups[(lane + 4 * wave) & (GROUP - 1)];
    asm volatile("" :: "v"(u), "v"(v), "v"(w), "s"(F0), "s"(FD));
#pragma unroll
    for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(R[g]));
    asm volatile("" : "+v"(Lg));

    // q and the channel-step phasor of this wave's source of a batch -> LDS
    auto produce = [&](int buf) {
        double q = 0.0;
        fmac_bcast<0>(q, Lg, u);
        fmac_bcast<1>(q, Lg, v);
        fmac_bcast<2>(q, Lg, w);
        double dr, di;
        sincos_quarter_turns<NTERM>(__dmul_rn(q, FD), dr, di);
        xch[buf][wave][0][lane] = q;
        xch[buf][wave][1][lane] = dr;
        xch[buf][wave][2][lane] = di;
    };
    produce(0);
    group_refresh<0>(Lg, lg_off, lgroups + (int64_t)(nb > 1 ? 1 : 0) * GROUP);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(Lg));
    __syncthreads();

#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
        // ---- produce batch b+1 (Lg was refreshed one batch ago: 4*NG record refreshes since) -------
        group_wait<4 * NG>(Lg);
        produce((b + 1) & 1);
        const int bn = (b + 2 < nb) ? b + 2 : nb - 1;
        group_refresh<0>(Lg, lg_off, lgroups + (int64_t)bn * GROUP);
        // ---- consume batch b on this wave's tile --------------------------------------------------
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int s = 4 * b + j;
            const int sn = (s + 1 < nsrc_pad) ? s + 1 : s;
            const double q = xch[b & 1][j][0][lane], dr = xch[b & 1][j][1][lane], di = xch[b & 1][j][2][lane];
            // the Lg refresh above sits between the previous batch's record refreshes and pass j = 0
            channel_pass<CT, NC, CPLX, NTERM, (j == 0 ? 1 : 0), true, NG, false>(acc, R, q, dr, di, F0, lane_off,
                                                                              rec + (int64_t)sn * (NG * GROUP), nullptr);
        });
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Line-aligned cooperative stores when the chunk holds every correlation and no column needs
    // the zero / NaN override (block-uniform conditions); per-lane stores otherwise.
    int tstate_any = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) tstate_any |= tilestate[(4 * blockIdx.y + k) * nchunk + chunk];
    if (NC == ncorr && tstate_any == 0) {
        store_quad_coop<CT, NC>(acc, stage, out, (int64_t)blockIdx.x * 64, nrow, nchan, ncorr,
                                (int64_t)4 * blockIdx.y * CT, wave, lane);
    } else {
        store_tile<CT, NC>(acc, out, row, valid, nchan, ncorr, c0, corr0, colstate,
                           tilestate[tile * nchunk + chunk]);
    }
}

// ---- exact kernel -----------------------------------------------------------------------
// Reference operation order, no contraction (kernels.py:57,61,65), full-accuracy sincos
// (sincos_radians: Cody-Waite + 7-term polynomials, ~1e-16) per (row, source, channel).  Channel-outermost inside the lane so only NC complex accumulators
// are live; real_phase is recomputed per channel (6 ops against a ~100-op sincos).
template <int CT, int NC, bool CPLX>
__global__ __launch_bounds__(ROWS_PER_BLOCK) void dft_exact_kernel(
    const double *__restrict__ uvw, const double *__restrict__ records, const double *__restrict__ freq_pad,
    const int *__restrict__ flags, const int *__restrict__ colstate, const int *__restrict__ tilestate,
    double *__restrict__ out, int64_t nrow, int nsrc, int nsrc_pad, int64_t nchan, int64_t ncorr, int64_t corr0,
    int chunk, int nchunk, int want_uniform, double constant)
{
    if (want_uniform >= 0 && flags[0] != want_uniform) return;
    constexpr int W = CPLX ? 2 : 1;
    constexpr int NG = record_groups(CT, NC, W);
    const int tile = blockIdx.y;
    const int64_t c0 = (int64_t)tile * CT;
    int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x;
    const bool valid = row < nrow;
    if (!valid) row = nrow - 1;
    const double u = uvw[3 * row], v = uvw[3 * row + 1], w = uvw[3 * row + 2];
    const double *__restrict__ rec = records + (int64_t)tile * nsrc_pad * (NG * GROUP);
    const int tstate = tilestate[tile * nchunk + chunk];

    for (int j = 0; j < CT; ++j) {
        if (c0 + j >= nchan) break;
        const double nu = freq_pad[c0 + j];
        double acc[NC][2];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c][0] = acc[c][1] = 0.0;
        for (int s = 0; s < nsrc; ++s) {
            const double *__restrict__ rs = rec + (int64_t)s * (NG * GROUP);
            const double l = rs[0], m = rs[1], n = rs[2];
            const double real_phase = __dmul_rn(
                constant, __dadd_rn(__dadd_rn(__dmul_rn(l, u), __dmul_rn(m, v)), __dmul_rn(n, w)));
            double yr, yi;
            sincos_radians(__dmul_rn(real_phase, nu), yr, yi);
            const double *__restrict__ g = rs + 4 + j * (NC * W);
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
// Tile width: the lane's accumulators (CT*NC*2 doubles) plus the record groups plus ~40
// working VGPRs must fit the 256 architectural VGPRs a VALU instruction can address; per
// (row, source, tile) the recurrence kernel spends ~50 fp64 ops of setup plus
// CT*(2 + 2*NC*(CPLX?2:1)) in the channel loop, so the widest tile that fits wins unless
// padding the last tile costs more.  Real images: CT in {13, 8}; complex: CT in {11, 8}.
int choose_ct(int64_t nchan, int nc_max, bool cplx)
{
    // candidates per correlation count: the accumulators of a tile are CT*NC complex numbers, so fewer correlations
    // afford wider tiles (the ~50-operation set-up per (row, source, tile) is then amortised over more channels)
    const int cands_c[2] = {11, 8};
    const int cands_r4[2] = {13, 8}, cands_r2[5] = {26, 22, 16, 13, 8}, cands_r1[4] = {52, 32, 13, 8};
    const int *cands = cplx ? cands_c : nc_max == 1 ? cands_r1 : nc_max == 2 ? cands_r2 : cands_r4;
    const int ncand = cplx ? 2 : nc_max == 2 ? 5 : nc_max == 1 ? 4 : 2;
    const int per_chan = 2 + 2 * nc_max * (cplx ? 2 : 1);
    int best = cands[0];
    int64_t best_cost = -1;
    for (int k = 0; k < ncand; ++k) {
        int ct = cands[k];
        int64_t cost = af_cdiv(nchan, ct) * (50 + (int64_t)ct * per_chan);
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
    bool mfma;  // the MFMA-accumulator kernels own the one-spacing band; only the fallbacks are launched here
    double constant;
    hipStream_t st;
};

template <int CT, int NC, bool CPLX>
int launch_chunk(const Args &a)
{
    const WsLayout &L = *a.L;
    char *ws = a.ws;
    const double *tilef = reinterpret_cast<const double *>(ws + L.tilef);
    const double *freq_pad = reinterpret_cast<const double *>(ws + L.freq);
    const int *flags = reinterpret_cast<const int *>(ws + L.flags);
    const int *colstate = reinterpret_cast<const int *>(ws + L.colstate);
    const int *tilestate = reinterpret_cast<const int *>(ws + L.tilestate);
    const double *records = reinterpret_cast<const double *>(ws + L.records) + L.chunk_off[a.chunk];
    dim3 grid((unsigned)af_cdiv(a.nrow, ROWS_PER_BLOCK), (unsigned)L.ntile), block(ROWS_PER_BLOCK);
    const int64_t corr0 = (int64_t)a.chunk * MAXNC;
    const double *lgroups = reinterpret_cast<const double *>(ws + L.lgroups);
    if (a.mode == AF_DFT_AUTO || a.mode == AF_DFT_RECURRENCE) {
        // Uniform channels (flags[0] == 1, set by dft_prep_freq or forced for AF_DFT_RECURRENCE).
        // Tiles in groups of four run the shared-phasor kernel when the whole band has one channel
        // spacing (flags[1] == 1) and the per-tile kernel otherwise; leftover tiles always run the
        // per-tile kernel.  Which of the launches does the work is decided on the device.
        const int quads = a.mfma ? 0 : (int)(L.ntile / 4), rest = (int)(L.ntile - 4 * (int64_t)quads);
        if (a.mfma) {
            // af_dft_mfma_run covers flags[1] == 1; the per-tile kernel covers per-tile spacings
            hipLaunchKernelGGL((dft_recurrence_dpp_kernel<CT, NC, CPLX, 7>), grid, block, 0, a.st, a.uvw, records,
                               tilef, flags, colstate, tilestate, a.out, a.nrow, (int)a.nsrc, (int)L.nsrc_pad,
                               a.nchan, a.ncorr, corr0, a.chunk, (int)L.nchunk, 0, 1, 0);
            AF_LAUNCH_CHECK();
        } else if (quads > 0) {
            dim3 g4((unsigned)af_cdiv(a.nrow, 64), (unsigned)quads);
            if (a.chunk == 0) af_prof_begin(a.st);  // measurement hook: the dominant kernel only
            hipLaunchKernelGGL((dft_recurrence_dpp4_kernel<CT, NC, CPLX, 7>), g4, block, 0, a.st, a.uvw, records,
                               lgroups, tilef, flags, colstate, tilestate, a.out, a.nrow, (int)L.nsrc_pad, a.nchan,
                               a.ncorr, corr0, a.chunk, (int)L.nchunk);
            if (a.chunk == 0) af_prof_end(a.st);
            AF_LAUNCH_CHECK();
            dim3 gq((unsigned)af_cdiv(a.nrow, ROWS_PER_BLOCK), (unsigned)(4 * quads));
            hipLaunchKernelGGL((dft_recurrence_dpp_kernel<CT, NC, CPLX, 7>), gq, block, 0, a.st, a.uvw, records,
                               tilef, flags, colstate, tilestate, a.out, a.nrow, (int)a.nsrc, (int)L.nsrc_pad,
                               a.nchan, a.ncorr, corr0, a.chunk, (int)L.nchunk, 0, 1, 0);
            AF_LAUNCH_CHECK();
        }
        if (!a.mfma && rest > 0) {
            dim3 gr((unsigned)af_cdiv(a.nrow, ROWS_PER_BLOCK), (unsigned)rest);
            if (a.chunk == 0 && quads == 0) af_prof_begin(a.st);
            hipLaunchKernelGGL((dft_recurrence_dpp_kernel<CT, NC, CPLX, 7>), gr, block, 0, a.st, a.uvw, records,
                               tilef, flags, colstate, tilestate, a.out, a.nrow, (int)a.nsrc, (int)L.nsrc_pad,
                               a.nchan, a.ncorr, corr0, a.chunk, (int)L.nchunk, 4 * quads, 1, -1);
            if (a.chunk == 0 && quads == 0) af_prof_end(a.st);
            AF_LAUNCH_CHECK();
        }
    }
    if (a.mode == AF_DFT_AUTO || a.mode == AF_DFT_EXACT) {
        // AUTO: runs iff flags[0] == 0 (non-uniform frequencies); EXACT: always (-1)
        if (a.chunk == 0 && a.mode == AF_DFT_EXACT) af_prof_begin(a.st);
        hipLaunchKernelGGL((dft_exact_kernel<CT, NC, CPLX>), grid, block, 0, a.st, a.uvw, records, freq_pad,
                           flags, colstate, tilestate, a.out, a.nrow, (int)a.nsrc, (int)L.nsrc_pad, a.nchan, a.ncorr,
                           corr0, a.chunk, (int)L.nchunk, a.mode == AF_DFT_EXACT ? -1 : 0, a.constant);
        if (a.chunk == 0 && a.mode == AF_DFT_EXACT) af_prof_end(a.st);
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

int launch_chunk_ct(bool cplx, int ct, int nc, const Args &a)
{
    if (cplx) return ct == 8 ? launch_chunk_nc<8, true>(nc, a) : launch_chunk_nc<11, true>(nc, a);
    switch (ct) {   // the wide tiles exist for the correlation counts choose_ct offers them to
    case 52: return launch_chunk<52, 1, false>(a);
    case 32: return launch_chunk<32, 1, false>(a);
    case 26: return launch_chunk<26, 2, false>(a);
    case 22: return launch_chunk<22, 2, false>(a);
    case 16: return launch_chunk<16, 2, false>(a);
    case 8: return launch_chunk_nc<8, false>(nc, a);
    default: return launch_chunk_nc<13, false>(nc, a);
    }
}

}  // namespace

AF_EXPORT size_t af_im_to_vis_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t ncorr, int image_is_complex)
{
    if (nsrc < 0 || nchan < 0 || ncorr < 0) return 0;
    // sized for the widest padding any tile width can produce
    size_t m = 0;
    const int cands[8] = {8, 11, 13, 16, 22, 26, 32, 52};
    for (int k = 0; k < 8; ++k) {
        WsLayout L;
        if (!ws_layout(L, nsrc, nchan, ncorr, image_is_complex, cands[k])) return 0;
        if (L.total > m) m = L.total;
    }
    if (af_dft_mfma_eligible(nchan, ncorr, image_is_complex != 0))
        m += af_dft_mfma_workspace_bytes(af_cdiv(nsrc > 0 ? nsrc : 1, 4) * 4, nchan, image_is_complex != 0);
    return m;
}

int af_chi2_launch(const double *model, const double *data, const double *weight, int64_t nrow, int64_t nchan, int64_t ncorr,
                   double *chi2_per_chan, const int *skip, hipStream_t st);   // af_chi2.hip

namespace {
int im_to_vis_impl(const double *image, int image_is_complex, const double *uvw, const double *lm,
                   const double *frequency, int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr, int convention, int mode,
                   double *out, void *workspace, size_t workspace_bytes, void *stream, const AfDftChi2 *chi);
}

AF_EXPORT int af_im_to_vis_f64(const double *image, int image_is_complex, const double *uvw, const double *lm,
                               const double *frequency, int64_t nsrc, int64_t nrow, int64_t nchan,
                               int64_t ncorr, int convention, int mode, double *out, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    return im_to_vis_impl(image, image_is_complex, uvw, lm, frequency, nsrc, nrow, nchan, ncorr, convention, mode, out,
                          workspace, workspace_bytes, stream, nullptr);
}

// The transform and the per-channel chi^2 of its result against `data` in one call:
//     out = im_to_vis(...);   chi2[nu] = sum_{row, corr} [weight] |data - out|^2       (af_chi2_c128's quantity)
// Where the MFMA kernels run (4 correlations, one channel spacing) chi^2 is summed in their epilogue, from the
// visibilities still in registers; everywhere else (and when the reference's zero-pixel / NaN-source semantics rewrite
// a column afterwards) the call falls back, on the device, to the separate pass -- same result either way.
AF_EXPORT int af_im_to_vis_chi2_f64(const double *image, int image_is_complex, const double *uvw, const double *lm,
                                    const double *frequency, int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr,
                                    int convention, int mode, double *out, const double *data, const double *weight,
                                    double *chi2_per_chan, void *workspace, size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(nchan == 0 || chi2_per_chan != nullptr, "af_im_to_vis_chi2_f64: chi2_per_chan is NULL");
    AF_REQUIRE(data != nullptr || nrow == 0 || nchan == 0 || ncorr == 0, "af_im_to_vis_chi2_f64: data is NULL");
    AfDftChi2 chi;
    chi.data = data; chi.weight = weight; chi.chi2 = chi2_per_chan;
    return im_to_vis_impl(image, image_is_complex, uvw, lm, frequency, nsrc, nrow, nchan, ncorr, convention, mode, out,
                          workspace, workspace_bytes, stream, &chi);
}

namespace {
int im_to_vis_impl(const double *image, int image_is_complex, const double *uvw, const double *lm,
                   const double *frequency, int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr, int convention, int mode,
                   double *out, void *workspace, size_t workspace_bytes, void *stream, const AfDftChi2 *chi)
{
    AF_REQUIRE(convention == AF_CONVENTION_FOURIER || convention == AF_CONVENTION_CASA,
               "convention not in ('fourier', 'casa')");
    const int clamp_n = (mode & AF_DFT_CLAMP_N) ? 1 : 0;
    const bool valu_only = (mode & AF_DFT_VALU_ONLY) != 0;
    mode &= ~(AF_DFT_CLAMP_N | AF_DFT_VALU_ONLY);
    AF_REQUIRE(mode == AF_DFT_AUTO || mode == AF_DFT_EXACT || mode == AF_DFT_RECURRENCE,
               "af_im_to_vis_f64: unknown mode %d", mode);
    AF_REQUIRE(nsrc >= 0 && nrow >= 0 && nchan >= 0 && ncorr >= 0, "af_im_to_vis_f64: negative extent");
    AF_REQUIRE(nsrc < (1LL << 31), "af_im_to_vis_f64: nsrc too large");
    AF_REQUIRE(ncorr <= 64 * MAXNC, "af_im_to_vis_f64: more than %d correlations", 64 * MAXNC);
    hipStream_t st = af_stream(stream);
    if (nrow == 0 || nchan == 0 || ncorr == 0) {
        if (chi && nchan > 0) AF_HIP(hipMemsetAsync(chi->chi2, 0, sizeof(double) * (size_t)nchan, st));
        return AF_OK;
    }
    AF_REQUIRE(out != nullptr && uvw != nullptr && frequency != nullptr, "af_im_to_vis_f64: NULL array");
    if (nsrc == 0) {  // np.zeros output (kernels.py:45)
        AF_HIP(hipMemsetAsync(out, 0, sizeof(double) * 2 * (size_t)(nrow * nchan * ncorr), st));
        return chi ? af_chi2_launch(out, chi->data, chi->weight, nrow, nchan, ncorr, chi->chi2, nullptr, st) : AF_OK;
    }
    AF_REQUIRE(image != nullptr && lm != nullptr, "af_im_to_vis_f64: NULL array");
    const bool cplx = image_is_complex != 0;
    const int ct = choose_ct(nchan, (int)(ncorr < MAXNC ? ncorr : MAXNC), cplx);
    WsLayout L;
    ws_layout(L, nsrc, nchan, ncorr, image_is_complex, ct);
    const bool mfma = !valu_only && mode != AF_DFT_EXACT && af_dft_mfma_eligible(nchan, ncorr, cplx);
    const size_t need = L.total + (mfma ? af_dft_mfma_workspace_bytes(L.nsrc_pad, nchan, cplx) : 0);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= need,
               "af_im_to_vis_f64: workspace too small (%zu < %zu)", workspace_bytes, need);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_im_to_vis_f64: workspace must be 256-byte aligned");
    AF_REQUIRE(L.ntile <= 65535, "af_im_to_vis_f64: too many channels");
    char *ws = static_cast<char *>(workspace);
    const int W = cplx ? 2 : 1;

    // flags[0] = 1: uniform until a tile says otherwise (1-byte memset of the low byte); tilestate = 0
    AF_HIP(hipMemsetAsync(ws + L.tilestate, 0, (size_t)L.ntile * L.nchunk * sizeof(int), st));
    AF_HIP(hipMemsetAsync(ws + L.flags, 0, 64 * sizeof(int), st));
    AF_HIP(hipMemsetAsync(ws + L.flags, 1, 1, st));
    AF_HIP(hipMemsetAsync(ws + L.flags + sizeof(int), 1, 1, st));  // flags[1] = 1: one channel spacing
    hipLaunchKernelGGL(dft_prep_src, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st, lm, nsrc, clamp_n,
                       reinterpret_cast<double *>(ws + L.lmn), reinterpret_cast<int *>(ws + L.srcbad));
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL(dft_prep_freq, dim3((unsigned)af_cdiv(L.ntile, 64)), dim3(64), 0, st, frequency, nchan,
                       L.ntile, ct, convention, mfma ? 64 : 0, reinterpret_cast<double *>(ws + L.tilef),
                       reinterpret_cast<double *>(ws + L.freq), reinterpret_cast<int *>(ws + L.flags));
    AF_LAUNCH_CHECK();
    if (mode == AF_DFT_RECURRENCE)  // caller asserts uniform spacing
        AF_HIP(hipMemsetAsync(ws + L.flags, 1, 1, st));
    for (int chunk = 0; chunk < (int)L.nchunk; ++chunk) {
        const int64_t total = L.ntile * L.nsrc_pad * (int64_t)L.chunk_groups[chunk] * GROUP;
        int64_t blocks = af_cdiv(total, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(dft_pack_records, dim3((unsigned)blocks), dim3(256), 0, st, image, W, nsrc, L.nsrc_pad,
                           nchan, ncorr, L.ntile, ct, chunk * MAXNC, L.chunk_nc[chunk], L.chunk_groups[chunk],
                           reinterpret_cast<const double *>(ws + L.lmn),
                           reinterpret_cast<const int *>(ws + L.srcbad),
                           reinterpret_cast<double *>(ws + L.records) + L.chunk_off[chunk]);
        AF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(dft_pack_lgroups, dim3((unsigned)af_cdiv(L.nsrc_pad * 4, 256)), dim3(256), 0, st,
                       reinterpret_cast<const double *>(ws + L.lmn), nsrc, L.nsrc_pad,
                       reinterpret_cast<double *>(ws + L.lgroups));
    AF_LAUNCH_CHECK();
    {
        int64_t ncol = L.ntile * ct * ncorr;
        hipLaunchKernelGGL(dft_colstate, dim3((unsigned)ncol), dim3(64), 0, st, image, W, nsrc, nchan, ncorr,
                           L.ntile, ct, reinterpret_cast<const int *>(ws + L.srcbad),
                           reinterpret_cast<int *>(ws + L.colstate), reinterpret_cast<int *>(ws + L.tilestate),
                           (int)L.nchunk, reinterpret_cast<int *>(ws + L.flags));
        AF_LAUNCH_CHECK();
    }
    Args a;
    a.L = &L; a.ws = ws; a.uvw = uvw; a.out = out;
    a.nrow = nrow; a.nsrc = nsrc; a.nchan = nchan; a.ncorr = ncorr;
    a.mode = mode; a.st = st; a.mfma = mfma;
    if (mfma) {
        if (chi) AF_HIP(hipMemsetAsync(chi->chi2, 0, sizeof(double) * (size_t)nchan, st));
        int rc = af_dft_mfma_run(image, (int)cplx, uvw, frequency, reinterpret_cast<const double *>(ws + L.lmn),
                                 reinterpret_cast<const int *>(ws + L.srcbad),
                                 reinterpret_cast<const double *>(ws + L.tilef),
                                 reinterpret_cast<const int *>(ws + L.flags),
                                 reinterpret_cast<const int *>(ws + L.colstate), convention, out, nrow, nsrc,
                                 L.nsrc_pad, nchan, ws + L.total, st, chi);
        if (rc != AF_OK) return rc;
    }
    a.constant = convention == AF_CONVENTION_FOURIER ? AF_MINUS_TWO_PI_OVER_C : AF_TWO_PI_OVER_C;
    for (int chunk = 0; chunk < (int)L.nchunk; ++chunk) {
        a.chunk = chunk;
        int rc = launch_chunk_ct(cplx, ct, L.chunk_nc[chunk], a);
        if (rc != AF_OK) return rc;
    }
    if (chi) {
        // the separate pass, unless the MFMA epilogues have done the sum (device flags (1, 1, 0): one channel spacing, no
        // column rewritten by the zero-pixel / NaN-source pass); without the MFMA path it always runs
        return af_chi2_launch(out, chi->data, chi->weight, nrow, nchan, ncorr, chi->chi2,
                              mfma ? reinterpret_cast<const int *>(ws + L.flags) : nullptr, st);
    }
    return AF_OK;
}
}  // namespace
