// Fused RIME predict with per-antenna beam-cube DDEs (BASELINE config 3):
//
//     V[r,nu] = sum_s  E_p(s,t,nu) . ( K(r,s,nu) B(s,nu) ) . E_q(s,t,nu)^H
//
// i.e. the reference's chain  phase_delay -> einsum("srf,sfij->srfij") -> beam_cube_dde ->
// predict_vis  (africanus/rime/examples/predict.py:107-134,404-472,525 with
// africanus/rime/phase.py:28-61, africanus/rime/fast_beam_cubes.py:57-240 and
// africanus/rime/predict.py:199-212) evaluated without materialising the (src,row,chan)
// coherencies (4.1 TB at C2) or the (src,time,ant,chan) Jones terms (130 GB at C3).
//
// One workgroup (512 lanes, one per CU: <= 250 VGPRs, up to 160 KB of LDS) owns one run of rows with
// equal time_index (<= 2048 rows: a whole 64-antenna timestep) and one channel, and walks the
// sources in batches:
//   stage 1  (beam) super-rounds of 512 Jones terms: every lane computes the voxel geometry of one (source, antenna)
//            term (32-bit byte offsets into the packed cube: 128-byte voxel records [corr][re, im, |.|, 0]), then four
//            sampling rounds in which the four lanes of a quad take the geometry of quad lane i by DPP broadcast and
//            sample one correlation each (8 x (16 + 8)-byte gathers, one cache line per voxel and quad) -> E, and
//            G = E.B(s,nu); both go to LDS as [src][component][antenna] so that stage 2's reads are conflict-free
//            (consecutive rows = consecutive antenna2) or broadcasts (antenna1).  The next super-round's source
//            coordinates are fetched one super-round ahead.  Bound by the L2 -> L1 line traffic of the gathers
//            (8 lines per Jones term: 2.1 TB at C3) and their address processing, not by arithmetic.
//   stage 2  (accumulate) lane = four rows: per source q = l u + m v + n w, the phasor from a 256-entry LDS table
//            times a short residual rotation, M = G_p . E_q^H (one 2x2 complex product instead of two because B
//            was folded into G per antenna), acc += K M.  66 fp64 + ~20 other instructions per (row, source):
//            fp64-VALU issue bound.
// Every per-antenna Jones is computed once per (timestep, channel, source) and reused by all
// baselines of the timestep from LDS; the output is written once.  Measured at C3 (tools/profile_fused.sh,
// AFHIP_FUSED_STAGE): 8-wave kernel 243 ms (stage 2 alone 159 ms, stage 1 adds ~89 ms back to back); wave-specialised
// kernel (WS: 8 accumulating + 4 sampling waves, double-buffered Jones; the default) 217 ms.
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "af_fused_device.h"

namespace {

constexpr int THREADS = 512;
constexpr int RPT = 4;  // rows per lane
constexpr int PLANE_PAD = 4;   // double2 elements of padding per (source, component) plane of the Jones arrays in LDS

// grid: (nitems, nchan); block 512.  Dynamic LDS: 2 * st * 4 * np double2 (E then G) + 6 * nant doubles + the feed
// rotations + the 256-entry phasor table.  NP > 0: the antenna stride np of the Jones arrays is the compile-time
// constant NP >= nant, so that the four components of a Jones term are one address plus immediate offsets; NP = 0:
// np = nant at run time.
// WS (wave-specialised): 12 waves -- the 8 waves of lanes 0..511 only accumulate (stage 2), the 4 waves of lanes
// 512..767 only sample the beam (stage 1), one batch ahead, into the other half of a double-buffered Jones region;
// one barrier per batch.  Each SIMD then holds two accumulating waves (fp64 issue bound) and one sampling wave
// (L1-fill bound), whose stalls the other two fill.
// ST > 0: the batch size `st` is the compile-time constant ST (and NP > 0): stage 2's source loop is unrolled and every
// Jones read is one per-row base address (set once per batch) plus an immediate offset.
// GR (grouped rows): items list GROUPS of up to four rows (p_i, q_j), i, j in {0, 1} -- 2 x 2 blocks of baselines that
// share their antennas (af_fused_plan_groups) -- instead of row ranges.  A lane owns one group: the Jones terms of p0, p1
// (G) and q0, q1 (E) serve four rows, 5 LDS reads per (row, source) instead of 8 (G p0, E q0 | E q1 | G p1 | E q0 again:
// with no more live registers than the row-by-row form), and the planes hold even and odd antennas in separate halves so
// that the lanes' q0 = 0, 2, 4, ... (and q1 = 1, 3, ...) are contiguous 16-byte elements: conflict-free reads.
template <bool FEED, bool GAUSS, int NP, bool WS, int ST, bool GR>
__global__ __launch_bounds__(WS ? THREADS + THREADS / 2 : THREADS) void fused_predict_kernel(
    const int32_t *__restrict__ items, const int32_t *__restrict__ ant1, const int32_t *__restrict__ ant2,
    const int32_t *__restrict__ groups,
    const double *__restrict__ uvw, const double *__restrict__ lmn, const double *__restrict__ f4,
    const double2 *__restrict__ brightness, const double *__restrict__ vrec,
    int64_t beam_lw, int64_t beam_mh, int64_t beam_nud, const double *__restrict__ lm_ext,
    const double *__restrict__ freq_data, const double *__restrict__ parangles,
    const double *__restrict__ point_errors, const double *__restrict__ antenna_scaling,
    const double2 *__restrict__ feed_rot, const double *__restrict__ gauss, const double *__restrict__ freq, int nsrc,
    int64_t nchan, int64_t ntime, int nant, int st_arg, double2 *__restrict__ out, int only_stage, int64_t f0)
{
    extern __shared__ double2 lds[];
    const int np = NP > 0 ? NP : nant;
    // double2 elements between the component planes of a source's Jones terms: the antenna stride plus 4 (64 bytes),
    // so that the four lanes of a quad, which write the four components of ONE term, hit four different bank groups
    // (at a plane stride of np * 16 bytes = a multiple of the 256-byte bank width they all hit the same one: measured,
    // a quarter of the kernel's LDS cycles were conflicts, 27 % of those from these writes)
    const int npp = np + PLANE_PAD;
    const int st = ST > 0 ? ST : st_arg;
    // element of antenna a inside a plane: GR splits the plane into an even-antenna and an odd-antenna half
    const int half = npp >> 1;
    auto slot_of = [&](int a) { return GR ? (a & 1) * half + (a >> 1) : a; };
    constexpr int NBUF = WS ? 2 : 1;              // Jones buffers: E then G, each [st][4][np]
    constexpr int NTHREADS = WS ? THREADS + THREADS / 2 : THREADS;
    constexpr int PLANES = WS ? THREADS / 2 : THREADS;   // lanes that sample the beam
    const int tid = threadIdx.x;
    const bool consumer = !WS || tid < THREADS;
    const int ptid = WS ? tid - THREADS : tid;    // lane number among the sampling lanes
    const int64_t f = f0 + blockIdx.y;            // planes hold channels f0 .. f0 + gridDim.y - 1
    const int t = items[4 * blockIdx.x + 0];
    const int64_t r0 = items[4 * blockIdx.x + 1];
    const int rc = items[4 * blockIdx.x + 2];

    // ---- stage-2 state: this lane's rows -----------------------------------------------------
    const double FT = f4[f] * (PH_TABLE / 4.0);   // 1/PH_TABLE turns per metre (exact scaling of f4)
    const double GK = freq[f] / FT;               // Gaussian shapes: (u nu) = us * GK
    double us[RPT], vs[RPT], ws[RPT];
    int a1[RPT], a2[RPT];
    C2 acc[RPT][4];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int rl = tid + k * THREADS;
        int64_t r;
        if constexpr (GR) {
            // item = (time, first group, group count): lane tid owns group r0 + tid; slot k = (i, j) = (k >> 1, k & 1)
            const bool have = consumer && tid < rc;
            const int32_t *g = groups + (r0 + (have ? tid : 0)) * 8;
            const int gr = g[4 + k];
            r = (have && gr >= 0) ? gr : 0;
            a1[k] = g[k >> 1];          // p_i
            a2[k] = g[2 + (k & 1)];     // q_j
        } else {
            r = r0 + ((consumer && rl < rc) ? rl : 0);
        }
        const double u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
        if constexpr (!GR) { a1[k] = ant1[r]; a2[k] = ant2[r]; }
        // this channel's table units per metre folded into the row's coordinates (one operation less per
        // (row, source); the Gaussian shape rescales them)
        us[k] = __dmul_rn(u, FT); vs[k] = __dmul_rn(v, FT); ws[k] = __dmul_rn(w, FT);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[k][c].re = acc[k][c].im = 0.0;
    }
    // element offsets (double2 units, from the start of the dynamic LDS) of this lane's Jones terms in the CURRENT
    // buffer: E planes of antenna2 / q, G planes of antenna1 / p.  Kept up to date in place from batch to batch (the
    // buffers alternate), so that the loop carries one register per operand and every read is offset + constant.
    int eoff[RPT], goff[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        eoff[k] = slot_of(a2[k]);
        goff[k] = st * 4 * npp + slot_of(a1[k]);
    }

    // ---- stage-1 state: per-antenna constants of this (timestep, channel) in LDS ------------------
    // ldsA[a] = (sin pa, cos pa, pe_l, pe_m, as_l, as_m); ldsR[a] = the antenna's 2x2 feed rotation (optional)
    double *ldsA = reinterpret_cast<double *>(lds + (size_t)NBUF * 2 * st * 4 * npp);
    double2 *ldsR = reinterpret_cast<double2 *>(ldsA + (size_t)6 * nant);
    double2 *ldsT = ldsR + (size_t)4 * nant;      // phasor table
    fine_table_init(ldsT, tid, NTHREADS);
    constexpr bool have_feed = FEED;
    if (have_feed)
        for (int i = tid; i < 4 * nant; i += NTHREADS) ldsR[i] = feed_rot[(int64_t)t * nant * 4 + i];
    for (int a = tid; a < nant; a += NTHREADS) {
        double sp, cp;
        sincos(parangles[(int64_t)t * nant + a], &sp, &cp);
        const double *pe = point_errors + (((int64_t)t * nant + a) * nchan + f) * 2;
        const double *as = antenna_scaling + ((int64_t)a * nchan + f) * 2;
        ldsA[6 * a + 0] = sp; ldsA[6 * a + 1] = cp;
        ldsA[6 * a + 2] = pe[0]; ldsA[6 * a + 3] = pe[1];
        ldsA[6 * a + 4] = as[0]; ldsA[6 * a + 5] = as[1];
    }
    FusedGrid grid;
    {
        const BeamGrid<double> g = beam_grid<double>(lm_ext, beam_lw, beam_mh, beam_nud);
        grid.lower_l = wave_uniform(g.lower_l); grid.lower_m = wave_uniform(g.lower_m); grid.lscale = wave_uniform(g.lscale); grid.mscale = wave_uniform(g.mscale);
        grid.lmaxf = g.lmaxf; grid.mmaxf = g.mmaxf; grid.lmaxi = (int)g.lmaxi; grid.mmaxi = (int)g.mmaxi;
        grid.stride_m = VREC * 8u;
        grid.stride_l = (unsigned)beam_mh * grid.stride_m;
    }
    const double fscale = freq_data[3 * f + 0];
    const int ntask = st * np;             // Jones slots per batch (antennas >= nant of a padded stride are skipped)
    const int e_corr = tid & 3;            // this lane's correlation in stage 1
    // this channel's plane (block-uniform base: the gathers below are scalar base + 32-bit lane offset), and this
    // lane's 32 bytes of a record
    const char *plane = reinterpret_cast<const char *>(vrec + (int64_t)blockIdx.y * beam_lw * beam_mh * VREC);
    const unsigned corr_off = e_corr * 32u;
    __syncthreads();

    auto stage1 = [&](int s0, double2 *ldsE, double2 *ldsG) {
        // ---- stage 1: E and G = E.B for the batch's (source, antenna) pairs -> LDS ------------------
        // Four lanes per Jones term, one per correlation: every lane gathers its own 32 bytes
        // (re, im, |.|) of each of the 8 voxel records, so a record's cache line is fetched once and no
        // cross-lane reduction is needed.  THREADS/4 terms per round.
        // Super-rounds of THREADS Jones terms: every lane first works out the voxel geometry of ONE term (its source
        // coordinates were fetched during the previous super-round); then four sampling rounds, in round i the four
        // lanes of a quad take the geometry of quad lane i by DPP broadcast and sample one correlation each -- the
        // geometry (a third of the stage's instructions) is computed once per term, not once per correlation.
        struct Own {
            int info;      // e_sl | e_ant << 11 | have_task << 30 | have << 31
            double2 lm;
        };
        auto fetch = [&](int task0) {
            Own T;
            const int task = task0 + ptid;
            int e_sl = task / np;          // NP > 0: a shift
            int e_ant = task - e_sl * np;
            const bool have_task = task < ntask && e_ant < nant;
            if (!have_task) e_sl = e_ant = 0;
            const bool have = have_task && s0 + e_sl < nsrc;
            T.info = e_sl | (e_ant << 11) | ((int)have_task << 30) | (int)((unsigned)have << 31);
            T.lm = *reinterpret_cast<const double2 *>(lmn + 4 * (have ? s0 + e_sl : 0));
            return T;
        };
        Own nxt = fetch(0);
        for (int task0 = 0; task0 < ntask && only_stage != 2; task0 += PLANES) {
            const Own own = nxt;
            if (task0 + PLANES < ntask) nxt = fetch(task0 + PLANES);
            FusedVoxels gx;
            {
                const int a = (own.info >> 11) & 1023;
                fused_voxels(grid, own.lm.x, own.lm.y, ldsA[6 * a + 0], ldsA[6 * a + 1], ldsA[6 * a + 2], ldsA[6 * a + 3],
                             ldsA[6 * a + 4], ldsA[6 * a + 5], fscale, gx);
            }
            // One sampling round in two halves: gathers issued, then consumed.
            struct Round {
                int info;
                double2 b0, b1, v[4];
                double ab[4], wt[4];
            };
            auto issue = [&](auto lane_c, Round &R) {
                constexpr int QL = decltype(lane_c)::value;
                const int info = quad_bcast<QL>(own.info);
                R.info = info;
                const int e_sl = info & 2047;
                const bool have = info < 0;
                // G[c] = E[2(c/2)] . B[c%2] + E[2(c/2)+1] . B[2 + c%2]: this lane needs column c%2 of B
                const double2 *bp = brightness + ((int64_t)(have ? s0 + e_sl : 0) * nchan + f) * 4;
                R.b0 = bp[e_corr & 1];
                R.b1 = bp[2 + (e_corr & 1)];
                unsigned off[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) off[k] = (unsigned)quad_bcast<QL>((int)gx.off[k]) + corr_off;
#pragma unroll
                for (int k = 0; k < 4; ++k) R.wt[k] = quad_bcast<QL>(gx.wt[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double *r = reinterpret_cast<const double *>(plane + (size_t)off[k]);
                    R.v[k] = *reinterpret_cast<const double2 *>(r);
                    R.ab[k] = r[2];
                }
            };
            auto finish = [&](const Round &R) {
                const int info = R.info;
                const int e_sl = info & 2047, e_ant = (info >> 11) & 1023;
                const bool have_task = (info >> 30) & 1, have = info < 0;
                const double2 b0 = R.b0, b1 = R.b1;
                double2 e = beam_reduce1(R.v, R.ab, R.wt);
                if (!have) e = make_double2(0.0, 0.0);
                // E[2i] and E[2i+1] of this lane's row i of the Jones matrix live in the even / odd lane of its pair
                C2 E0, E1;
                E0.re = pair_bcast<0>(e.x); E0.im = pair_bcast<0>(e.y);
                E1.re = pair_bcast<1>(e.x); E1.im = pair_bcast<1>(e.y);
                if constexpr (have_feed) {
                    // E <- E . R(t, antenna)  (einsum "stafij,tajk->stafik", rime/examples/predict.py:472):
                    // this lane's component (i, j = e_corr & 1) is E[i,0] R[0,j] + E[i,1] R[1,j]
                    const double2 r0 = ldsR[4 * e_ant + (e_corr & 1)], r1 = ldsR[4 * e_ant + 2 + (e_corr & 1)];
                    C2 R0, R1;
                    R0.re = r0.x; R0.im = r0.y; R1.re = r1.x; R1.im = r1.y;
                    C2 Er = cmul(E0, R0);
                    cmac(Er, E1, R1);
                    e = make_double2(Er.re, Er.im);
                    E0.re = pair_bcast<0>(Er.re); E0.im = pair_bcast<0>(Er.im);
                    E1.re = pair_bcast<1>(Er.re); E1.im = pair_bcast<1>(Er.im);
                }
                C2 B0, B1, G;
                B0.re = b0.x; B0.im = b0.y; B1.re = b1.x; B1.im = b1.y;
                G = cmul(E0, B0);
                cmac(G, E1, B1);
                if (have_task) {
                    ldsE[((size_t)e_sl * 4 + e_corr) * npp + slot_of(e_ant)] = e;
                    ldsG[((size_t)e_sl * 4 + e_corr) * npp + slot_of(e_ant)] = have ? make_double2(G.re, G.im)
                                                                          : make_double2(0.0, 0.0);
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>;
            // (two rounds in flight, measured again in round 2 now that a round's operands are 24 registers and fit: the
            // sampling waves ALONE finish 10 % sooner, 77.5 -> 69.4 ms, the whole kernel not at all, 174.3 vs 174.4 ms --
            // what the sampling waves cost the kernel is their issue slots, not their latency)
            Round R;
            issue(I0{}, R); finish(R);
            issue(I1{}, R); finish(R);
            issue(I2{}, R); finish(R);
            issue(I3{}, R); finish(R);
        }
    };
    auto stage2 = [&](int s0) {
        // ---- stage 2: every source of the batch, this lane's rows ----------------------------------
        const int nb = (nsrc - s0 < st) ? (nsrc - s0) : st;
        // the batch's source coordinates by scalar loads BEFORE the source loop: one SMEM round trip per batch where a
        // load at the head of every source's section waited one out per source (lgkmcnt is shared with the LDS reads of
        // the Jones terms): 177.6 -> 176.8 ms at BASELINE configs[2], same box
        double Lh[ST > 0 ? ST : 1], Mh[ST > 0 ? ST : 1], Nh[ST > 0 ? ST : 1];
        if constexpr (ST > 0) {
#pragma unroll
            for (int sl = 0; sl < ST; ++sl) {
                // ST > 0 always walks whole batches: a source beyond the last one has E = G = 0 in LDS (stage 1 writes
                // zeros for it) and adds exactly nothing; only its coordinates must come from a valid address
                const int sg = (s0 + sl >= nsrc) ? nsrc - 1 : s0 + sl;
                Lh[sl] = lmn[4 * sg]; Mh[sl] = lmn[4 * sg + 1]; Nh[sl] = lmn[4 * sg + 2];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        auto one_source = [&](int sl) {
            const int sg = (ST > 0 && s0 + sl >= nsrc) ? nsrc - 1 : s0 + sl;
            double l, m, n;
            if constexpr (ST > 0) { l = Lh[sl]; m = Mh[sl]; n = Nh[sl]; }
            else { l = lmn[4 * sg]; m = lmn[4 * sg + 1]; n = lmn[4 * sg + 2]; }
            // Gaussian shape factors exp(-(u1^2 + v1^2) (nu gs)^2) of this lane's rows (gaussian_shape.py:52-60);
            // computed ahead of the row loop so that exp's temporaries do not overlap the Jones algebra
            double shape[RPT];
            bool extended = false;
            if constexpr (GAUSS) {
                const double gel = gauss[4 * sg], gem = gauss[4 * sg + 1], ger = gauss[4 * sg + 2];
                extended = gauss[4 * sg + 3] != 0.0;  // block-uniform
                if (extended) {
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        // u nu = us * (nu / FT): the rows keep only their scaled coordinates (registers)
                        const double u1 = (us[k] * gem - vs[k] * gel) * ger * GK, v1 = (us[k] * gel + vs[k] * gem) * GK;
                        shape[k] = exp_neg(u1 * u1 + v1 * v1);
                    }
                }
            }
            const int so = sl * 4 * npp;
            // Jones operands of the row being worked on (re-used across the slots of a group)
            C2 Gp0, Gp1, Gp2, Gp3, Eq0, Eq1, Eq2, Eq3;
#define AF_LOAD_G(off) do { const double2 *b_ = lds + ((off) + so);                                                   \
        double2 x0 = b_[0], x1 = b_[npp], x2 = b_[2 * npp], x3 = b_[3 * npp];                                    \
        Gp0.re = x0.x; Gp0.im = x0.y; Gp1.re = x1.x; Gp1.im = x1.y; Gp2.re = x2.x; Gp2.im = x2.y;                 \
        Gp3.re = x3.x; Gp3.im = x3.y; } while (0)
#define AF_LOAD_E(off) do { const double2 *b_ = lds + ((off) + so);                                                   \
        double2 x0 = b_[0], x1 = b_[npp], x2 = b_[2 * npp], x3 = b_[3 * npp];                                    \
        Eq0.re = x0.x; Eq0.im = x0.y; Eq1.re = x1.x; Eq1.im = x1.y; Eq2.re = x2.x; Eq2.im = x2.y;                 \
        Eq3.re = x3.x; Eq3.im = x3.y; } while (0)
            // acc[k] += y M,  M = G_p . E_q^H :  M[i][j] = sum_k G[i][k] conj(E[j][k]).  The phasor first, then the
            // operand loads LOADS (none when the previous slot's operands serve), then the algebra: the operands are not
            // live while the phasor's temporaries are
#define AF_ROW(k, LOADS) do {                                                                                   \
        C2 y = table_phasor(ldsT, fma(n, ws[k], fma(m, vs[k], __dmul_rn(l, us[k]))));                            \
        if constexpr (GAUSS) { if (extended) { y.re *= shape[k]; y.im *= shape[k]; } }                           \
        LOADS;                                                                                                   \
        C2 M0 = cmulc(Gp0, Eq0); cmacc(M0, Gp1, Eq1);                                                            \
        C2 M1 = cmulc(Gp0, Eq2); cmacc(M1, Gp1, Eq3);                                                            \
        C2 M2 = cmulc(Gp2, Eq0); cmacc(M2, Gp3, Eq1);                                                            \
        C2 M3 = cmulc(Gp2, Eq2); cmacc(M3, Gp3, Eq3);                                                            \
        cmac(acc[k][0], y, M0); cmac(acc[k][1], y, M1); cmac(acc[k][2], y, M2); cmac(acc[k][3], y, M3);          \
    } while (0)
            if constexpr (GR) {
                // slots k = 2 i + j: (p0,q0) (p0,q1) (p1,q1) (p1,q0); every operand change is one 4-read load
                AF_ROW(0, AF_LOAD_G(goff[0]); AF_LOAD_E(eoff[0]));
                AF_ROW(1, AF_LOAD_E(eoff[1]));
                AF_ROW(3, AF_LOAD_G(goff[2]));
                AF_ROW(2, AF_LOAD_E(eoff[0]));
            } else {
                AF_ROW(0, AF_LOAD_G(goff[0]); AF_LOAD_E(eoff[0]));
                AF_ROW(1, AF_LOAD_G(goff[1]); AF_LOAD_E(eoff[1]));
                AF_ROW(2, AF_LOAD_G(goff[2]); AF_LOAD_E(eoff[2]));
                AF_ROW(3, AF_LOAD_G(goff[3]); AF_LOAD_E(eoff[3]));
            }
#undef AF_LOAD_G
#undef AF_LOAD_E
#undef AF_ROW
        };
        if (only_stage == 1) return;
        if constexpr (ST > 0) {
#pragma unroll
            for (int sl = 0; sl < ST; ++sl) {
                one_source(sl);
                // keep the sources apart: interleaved, their live Jones terms exceed the 168 registers of
                // three waves per SIMD
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 1
            for (int sl = 0; sl < nb; ++sl) one_source(sl);
        }
    };
    const size_t buf_elems = (size_t)2 * st * 4 * npp;   // one buffer: E then G
    if constexpr (!WS) {
        for (int s0 = 0; s0 < nsrc; s0 += st) {
            stage1(s0, lds, lds + (size_t)st * 4 * npp);
            __syncthreads();
            stage2(s0);
            __syncthreads();
        }
    } else {
        // producer: sample batch b, barrier b; consumer: barrier b, accumulate batch b.  When the producer passes
        // barrier b the consumers have finished batch b - 1, whose buffer batch b + 1 overwrites.  Two separate
        // loops: the accumulators must not be live in the sampling waves' code (168 registers at 3 waves per SIMD).
        if (!consumer) {
            // the lone sampling wave of a SIMD runs dependent chains behind memory latency: let it issue whenever it
            // is ready, ahead of the two accumulating waves (which always have independent work)
            __builtin_amdgcn_s_setprio(3);
            int b = 0;
            for (int s0 = 0; s0 < nsrc; s0 += st, ++b) {
                double2 *E = lds + (size_t)(b & 1) * buf_elems;
                stage1(s0, E, E + (size_t)st * 4 * npp);
                __syncthreads();
            }
            return;
        }
        int b = 0;
        for (int s0 = 0; s0 < nsrc; s0 += st, ++b) {
            __syncthreads();
            stage2(s0);
            // on to the other buffer, in place
            const int step = (b & 1) ? -(int)buf_elems : (int)buf_elems;
#pragma unroll
            for (int k = 0; k < RPT; ++k) { eoff[k] += step; goff[k] += step; }
        }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        // where slot k goes (not kept in registers through the main loop)
        int64_t orow = -1;
        if constexpr (GR) {
            if (tid < rc) orow = groups[(r0 + tid) * 8 + 4 + k];
        } else {
            if (tid + k * THREADS < rc) orow = r0 + tid + k * THREADS;
        }
        if (orow >= 0) {
            double2 *o = out + (orow * nchan + f) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = make_double2(acc[k][c].re, acc[k][c].im);
        }
    }
}

}  // namespace

// Host-side helper (host pointers): split rows (time_index non-decreasing... any order of equal
// runs) into items of consecutive rows with equal time_index, at most max_rows each;
// item = (time_index - min(time_index), row_start, row_count, 0).
AF_EXPORT int af_fused_plan_rows(const int64_t *time_index_host, int64_t nrow, int32_t *items_host,
                                 int64_t max_items, int64_t *nitems)
{
    AF_REQUIRE(nitems != nullptr, "af_fused_plan_rows: nitems is NULL");
    *nitems = 0;
    if (nrow == 0) return AF_OK;
    AF_REQUIRE(time_index_host != nullptr, "af_fused_plan_rows: time_index is NULL");
    int64_t tmin = time_index_host[0];
    for (int64_t r = 1; r < nrow; ++r) tmin = time_index_host[r] < tmin ? time_index_host[r] : tmin;
    const int64_t max_rows = (int64_t)THREADS * RPT;
    int64_t n = 0, start = 0;
    for (int64_t r = 1; r <= nrow; ++r) {
        if (r == nrow || time_index_host[r] != time_index_host[start] || r - start == max_rows) {
            if (items_host != nullptr) {
                AF_REQUIRE(n < max_items, "af_fused_plan_rows: more than %lld items", (long long)max_items);
                AF_REQUIRE(time_index_host[start] - tmin < (1LL << 31) && start < (1LL << 31),
                           "af_fused_plan_rows: index does not fit int32");
                items_host[4 * n + 0] = (int32_t)(time_index_host[start] - tmin);
                items_host[4 * n + 1] = (int32_t)start;
                items_host[4 * n + 2] = (int32_t)(r - start);
                items_host[4 * n + 3] = 0;
            }
            ++n;
            start = r;
        }
    }
    *nitems = n;
    return AF_OK;
}

// Host-side planner of the grouped form (host pointers): every run of consecutive rows with equal time_index is cut
// into GROUPS of up to four rows (p_i, q_j), i, j in {0, 1}: rows whose antennas fall into the same pair of antenna
// pairs (antenna1 >> 1, antenna2 >> 1) share a group (slot 2 (antenna1 & 1) + (antenna2 & 1)); groups left with a
// single row are then merged two by two (their rows become slots (0,0) and (1,1) of a group with unrelated p's and
// q's), so that a 64-antenna timestep (2016 baselines) is 512 groups = one workgroup.  Any row order, repeated
// baselines and autocorrelations are fine: a row that finds its slot taken opens another group.
//   groups (ngroups, 8) int32: p0, p1, q0, q1, row(0,0), row(0,1), row(1,0), row(1,1)  (row = -1: empty slot)
//   items  (nitems, 4)  int32: time_index - min(time_index), first group, group count (<= 512), 1
// Two-call protocol: with items_host == NULL or groups_host == NULL only the counts are returned.
AF_EXPORT int af_fused_plan_groups(const int64_t *time_index_host, const int32_t *antenna1_host,
                                   const int32_t *antenna2_host, int64_t nrow, int64_t nant, int32_t *items_host,
                                   int64_t max_items, int64_t *nitems, int32_t *groups_host, int64_t max_groups,
                                   int64_t *ngroups)
{
    AF_REQUIRE(nitems != nullptr && ngroups != nullptr, "af_fused_plan_groups: count pointer is NULL");
    *nitems = 0;
    *ngroups = 0;
    if (nrow == 0) return AF_OK;
    AF_REQUIRE(time_index_host && antenna1_host && antenna2_host, "af_fused_plan_groups: NULL array");
    AF_REQUIRE(nant >= 1 && nant <= 664 && nrow < (1LL << 31), "af_fused_plan_groups: bad extents");
    const bool write = items_host != nullptr && groups_host != nullptr;
    int64_t tmin = time_index_host[0];
    for (int64_t r = 1; r < nrow; ++r) tmin = time_index_host[r] < tmin ? time_index_host[r] : tmin;
    const int64_t hb = (nant + 1) / 2;
    std::vector<int32_t> stamp((size_t)(hb * hb), -1), last((size_t)(hb * hb), 0);
    struct Grp { int32_t p[2], q[2], row[4]; };
    std::vector<Grp> run;
    int64_t ni = 0, ng = 0, run_id = 0;
    const int32_t top = (int32_t)nant - 1;
    for (int64_t start = 0; start < nrow;) {
        int64_t end = start + 1;
        while (end < nrow && time_index_host[end] == time_index_host[start]) ++end;
        run.clear();
        for (int64_t r = start; r < end; ++r) {
            const int32_t a1 = antenna1_host[r], a2 = antenna2_host[r];
            AF_REQUIRE(a1 >= 0 && a1 < nant && a2 >= 0 && a2 < nant, "af_fused_plan_groups: antenna index out of range");
            const size_t key = (size_t)(a1 >> 1) * hb + (a2 >> 1);
            const int slot = 2 * (a1 & 1) + (a2 & 1);
            int g = -1;
            if (stamp[key] == (int32_t)run_id && run[(size_t)last[key]].row[slot] < 0) g = last[key];
            if (g < 0) {
                Grp G;
                G.p[0] = a1 & ~1; G.p[1] = (a1 | 1) > top ? top : (a1 | 1);
                G.q[0] = a2 & ~1; G.q[1] = (a2 | 1) > top ? top : (a2 | 1);
                G.row[0] = G.row[1] = G.row[2] = G.row[3] = -1;
                run.push_back(G);
                g = (int)run.size() - 1;
                stamp[key] = (int32_t)run_id;
                last[key] = g;
            }
            run[(size_t)g].row[slot] = (int32_t)r;
        }
        ++run_id;
        // merge the single-row groups two by two
        std::vector<Grp> out;
        out.reserve(run.size());
        int pending = -1;
        for (size_t g = 0; g < run.size(); ++g) {
            int cnt = 0, slot = 0;
            for (int k = 0; k < 4; ++k)
                if (run[g].row[k] >= 0) { ++cnt; slot = k; }
            if (cnt != 1) { out.push_back(run[g]); continue; }
            Grp S;     // the single row as slot (0,0) of a group of its own antennas
            S.p[0] = S.p[1] = run[g].p[slot >> 1]; S.q[0] = S.q[1] = run[g].q[slot & 1];
            S.row[0] = run[g].row[slot]; S.row[1] = S.row[2] = S.row[3] = -1;
            if (pending < 0) {
                out.push_back(S);
                pending = (int)out.size() - 1;
            } else {   // second single: slot (1,1) of the pending group
                out[(size_t)pending].p[1] = S.p[0];
                out[(size_t)pending].q[1] = S.q[0];
                out[(size_t)pending].row[3] = S.row[0];
                pending = -1;
            }
        }
        for (size_t g0 = 0; g0 < out.size(); g0 += THREADS) {
            const size_t cnt = out.size() - g0 < (size_t)THREADS ? out.size() - g0 : (size_t)THREADS;
            if (write) {
                AF_REQUIRE(ni < max_items && ng + (int64_t)cnt <= max_groups, "af_fused_plan_groups: output arrays too small");
                AF_REQUIRE(time_index_host[start] - tmin < (1LL << 31), "af_fused_plan_groups: index does not fit int32");
                items_host[4 * ni + 0] = (int32_t)(time_index_host[start] - tmin);
                items_host[4 * ni + 1] = (int32_t)ng;
                items_host[4 * ni + 2] = (int32_t)cnt;
                items_host[4 * ni + 3] = 1;
                for (size_t k = 0; k < cnt; ++k) {
                    const Grp &G = out[g0 + k];
                    int32_t *o = groups_host + 8 * (ng + (int64_t)k);
                    o[0] = G.p[0]; o[1] = G.p[1]; o[2] = G.q[0]; o[3] = G.q[1];
                    o[4] = G.row[0]; o[5] = G.row[1]; o[6] = G.row[2]; o[7] = G.row[3];
                }
            }
            ++ni;
            ng += (int64_t)cnt;
        }
        start = end;
    }
    *nitems = ni;
    *ngroups = ng;
    return AF_OK;
}

AF_EXPORT size_t af_fused_predict_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t beam_lw, int64_t beam_mh,
                                                  int64_t beam_nud)
{
    if (nsrc < 0 || nchan < 0 || beam_lw < 0 || beam_mh < 0 || beam_nud < 0) return 0;
    return fused_ws(nsrc, nchan, beam_lw, beam_mh, beam_nud).total;
}

AF_EXPORT int af_fused_predict_c128(const int32_t *items, int64_t nitems, const int32_t *antenna1,
                                    const int32_t *antenna2, const int32_t *groups, int64_t nrow, const double *lm,
                                    const double *uvw,
                                    const double *frequency, const double *brightness, int64_t nsrc, int64_t nchan,
                                    const double *beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud,
                                    const double *beam_lm_extents, const double *beam_freq_map,
                                    const double *parallactic_angles, int64_t ntime, int64_t nant,
                                    const double *point_errors, const double *antenna_scaling,
                                    const double *feed_rotation, const double *gauss_shape, int convention,
                                    double *out, void *workspace, size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(convention == AF_CONVENTION_FOURIER || convention == AF_CONVENTION_CASA,
               "convention not in ('fourier', 'casa')");
    AF_REQUIRE(beam_lw >= 2 && beam_mh >= 2 && beam_nud >= 2, "beam_lw, beam_mh and beam_nud must be >= 2");
    AF_REQUIRE(nitems >= 0 && nrow >= 0 && nsrc >= 0 && nchan >= 0 && ntime >= 0 && nant >= 0,
               "af_fused_predict_c128: negative extent");
    // one time step's Jones of a source batch + per-antenna constants live in LDS (160 KiB per workgroup)
    AF_REQUIRE(nant <= 664, "af_fused_predict_c128: more than 664 antennas");
    AF_REQUIRE(nsrc < (1LL << 31) && nchan <= 65535 && nitems < (1LL << 31), "af_fused_predict_c128: too large");
    hipStream_t st_ = af_stream(stream);
    if (nrow == 0 || nchan == 0) return AF_OK;
    AF_REQUIRE(out != nullptr, "af_fused_predict_c128: out is NULL");
    if (nsrc == 0 || nitems == 0) {
        AF_HIP(hipMemsetAsync(out, 0, sizeof(double) * 2 * 4 * (size_t)(nrow * nchan), st_));
        return AF_OK;
    }
    AF_REQUIRE(items && (groups || (antenna1 && antenna2)) && lm && uvw && frequency && brightness && beam && beam_lm_extents &&
                   beam_freq_map && parallactic_angles && point_errors && antenna_scaling,
               "af_fused_predict_c128: NULL array");
    const FusedWs W = fused_ws(nsrc, nchan, beam_lw, beam_mh, beam_nud);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= W.total, "af_fused_predict_c128: workspace too small (%zu < %zu)",
               workspace_bytes, W.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_fused_predict_c128: workspace must be 256-byte aligned");
    char *ws = static_cast<char *>(workspace);
    double *lmn = reinterpret_cast<double *>(ws + W.lmn), *f4 = reinterpret_cast<double *>(ws + W.f4);
    double *freq_data = reinterpret_cast<double *>(ws + W.freq_data), *planes = reinterpret_cast<double *>(ws + W.planes);

    hipLaunchKernelGGL(fused_prep_src, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st_, lm, nsrc, lmn);
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL(fused_prep_freq, dim3((unsigned)af_cdiv(nchan, 64)), dim3(64), 0, st_, frequency, nchan,
                       convention, f4);
    AF_LAUNCH_CHECK();
    int rc = af_freq_grid_interp_f64(frequency, nchan, beam_freq_map, beam_nud, freq_data, stream);
    if (rc != AF_OK) return rc;
    double *gp = reinterpret_cast<double *>(ws + W.gauss);
    {
        const double fwhm = 2.0 * sqrt(2.0 * log(2.0));  // gaussian_shape.py:23-25
        const double gs = (1.0 / fwhm) * sqrt(2.0) * 3.141592653589793 / AF_LIGHTSPEED;
        hipLaunchKernelGGL(fused_prep_gauss, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st_, gauss_shape, nsrc, gs,
                           gp);
        AF_LAUNCH_CHECK();
    }
    const int64_t ncell = beam_lw * beam_mh;
    AF_REQUIRE(ncell < (1LL << 25), "af_fused_predict_c128: beam cube too large (32-bit byte offsets into the 128-byte "
                                    "cell records of a channel plane: fewer than 2^25 cells per plane)");
    // sources per batch: as many as fit 128 KB of LDS (E and G: 128 bytes per (source, antenna))
    // (E and G: 128 bytes per (source, antenna); 112 bytes of constants per antenna; 160 KiB per workgroup)
    // antenna stride of the Jones arrays: a compile-time constant for the common array sizes
    const int NPv = (nant > 32 && nant <= 64) ? 64 : (nant > 64 && nant <= 128) ? 128 : 0;
    const int64_t np = NPv ? NPv : nant;
    const int64_t fixed = 112 * nant + PH_TABLE * 16;   // per-antenna constants, feed rotation, phasor table
    // wave-specialised variant (12 waves, double-buffered Jones): the default when two buffers of >= 2 sources fit
    // and no Gaussian shapes are folded in (that variant needs more than the 168 registers of 3 waves per SIMD);
    // AFHIP_FUSED_WS=0 selects the 8-wave kernel
    static const int ws_env = getenv("AFHIP_FUSED_WS") ? atoi(getenv("AFHIP_FUSED_WS")) : 1;
    const int64_t npp = np + PLANE_PAD;   // padded plane stride of the Jones arrays (see the kernel)
    const bool ws_mode = ws_env != 0 && (160 * 1024 - fixed) / (256 * npp) >= 2;
    const int nbuf = ws_mode ? 2 : 1;
    int st = (int)((160 * 1024 - fixed) / (128 * nbuf * npp));
    if (st > 1024 / np) st = (int)(1024 / np);
    {   // whole super-rounds of the sampling lanes: st * np a multiple of their number (a partly filled super-round
        // costs as much as a full one: 9 sources x 64 antennas on 256 lanes ran 3 super-rounds for 2.25 of work)
        const int64_t lanes = ws_mode ? THREADS / 2 : THREADS;
        int64_t m = 1;
        while ((m * np) % lanes != 0 && m < lanes) ++m;
        if (st >= m) st -= st % (int)m;
    }
    if (st > nsrc) st = (int)nsrc;
    if (st < 1) st = 1;
    const size_t lds_bytes = (size_t)nbuf * 2 * st * 4 * npp * sizeof(double2) + (size_t)nant * 6 * sizeof(double) +
                             (size_t)nant * 4 * sizeof(double2) + PH_TABLE * sizeof(double2);
    AF_REQUIRE(lds_bytes <= 160 * 1024, "af_fused_predict_c128: %zu bytes of LDS needed (nant = %lld)", lds_bytes,
               (long long)nant);
    const bool feed = feed_rotation != nullptr, gauss = gauss_shape != nullptr;
    // measurement hook (tools/profile_fused.sh): AFHIP_FUSED_STAGE=1 / 2 runs only the beam stage / only the
    // accumulation stage (results are then meaningless); unset = the real kernel
    // (profiling builds only: make HOOKS=1, -DAFHIP_STAGE_HOOKS; the shipped library does not read the environment here)
    static const int only_stage = AF_STAGE_ENV("AFHIP_FUSED_STAGE", 0);
    // channels in groups of PLANE_GROUP: interpolate the group's beam planes, then one launch for its channels (the
    // planes of a group are rewritten by the next group's pass on the same stream, after this group's kernel)
    auto launch = [&](auto kernel, int nthreads) -> int {
        AF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds_bytes));
        for (int64_t f0 = 0; f0 < nchan; f0 += PLANE_GROUP) {
            const int64_t nf = nchan - f0 < PLANE_GROUP ? nchan - f0 : PLANE_GROUP;
            int64_t blocks = af_cdiv(ncell * 4, 256);
            if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(beam_plane_kernel, dim3((unsigned)blocks, (unsigned)nf), dim3(256), 0, st_,
                               reinterpret_cast<const double2 *>(beam), ncell, beam_nud, freq_data, f0, planes);
            AF_LAUNCH_CHECK();
            if (f0 == 0) af_prof_begin(st_);
            hipLaunchKernelGGL(kernel, dim3((unsigned)nitems, (unsigned)nf), dim3(nthreads), lds_bytes, st_, items, antenna1,
                               antenna2, groups, uvw, lmn, f4, reinterpret_cast<const double2 *>(brightness), planes, beam_lw,
                               beam_mh, beam_nud, beam_lm_extents, freq_data, parallactic_angles, point_errors,
                               antenna_scaling, reinterpret_cast<const double2 *>(feed_rotation), gp, frequency, (int)nsrc,
                               nchan, ntime, (int)nant, st, reinterpret_cast<double2 *>(out), only_stage, f0);
            if (f0 == 0) af_prof_end(st_);
            AF_LAUNCH_CHECK();
        }
        return AF_OK;
    };
#define AF_FUSED_PICK(NPC, STC, GRC)                                                                                     \
    (ws_mode ? (feed ? (gauss ? launch(fused_predict_kernel<true, true, NPC, true, 0, GRC>, THREADS + THREADS / 2)        \
                              : launch(fused_predict_kernel<true, false, NPC, true, STC, GRC>, THREADS + THREADS / 2))      \
                     : (gauss ? launch(fused_predict_kernel<false, true, NPC, true, 0, GRC>, THREADS + THREADS / 2)       \
                              : launch(fused_predict_kernel<false, false, NPC, true, STC, GRC>, THREADS + THREADS / 2)))    \
             : (feed ? (gauss ? launch(fused_predict_kernel<true, true, NPC, false, 0, false>, THREADS)                     \
                              : launch(fused_predict_kernel<true, false, NPC, false, 0, false>, THREADS))                   \
                     : (gauss ? launch(fused_predict_kernel<false, true, NPC, false, 0, false>, THREADS)                    \
                              : launch(fused_predict_kernel<false, false, NPC, false, 0, false>, THREADS))))
    // the grouped form (items of af_fused_plan_groups) exists for the wave-specialised kernels; the 64-antenna-stride,
    // 8-sources-per-batch case (BASELINE configs[2]) has its source loop unrolled
    static const int unroll_env = getenv("AFHIP_FUSED_UNROLL") ? atoi(getenv("AFHIP_FUSED_UNROLL")) : 1;   // A/B hook
    AF_REQUIRE(groups == nullptr || ws_mode, "af_fused_predict_c128: grouped items need the wave-specialised kernel "
                                             "(AFHIP_FUSED_WS != 0, at most ~230 antennas): plan with af_fused_plan_rows");
    // (the unrolled source loop is for the variants without Gaussian shapes: eight inlined exp() bodies do not fit the
    // registers of three waves per SIMD)
    // (the unrolled source loop stays with the variants without Gaussian shapes: eight inlined envelope evaluations beside
    // the Jones algebra need ~3 KB of scratch per lane; the ROLLED grouped form fits with 7 spilled registers)
    const bool unroll = NPv == 64 && ws_mode && st == 8 && unroll_env && !gauss;
    if (groups != nullptr) {
        if (unroll) rc = AF_FUSED_PICK(64, 8, true);
        else rc = NPv == 64 ? AF_FUSED_PICK(64, 0, true) : NPv == 128 ? AF_FUSED_PICK(128, 0, true) : AF_FUSED_PICK(0, 0, true);
    } else if (unroll) rc = AF_FUSED_PICK(64, 8, false);
    else rc = NPv == 64 ? AF_FUSED_PICK(64, 0, false) : NPv == 128 ? AF_FUSED_PICK(128, 0, false) : AF_FUSED_PICK(0, 0, false);
#undef AF_FUSED_PICK
    return rc;
}
