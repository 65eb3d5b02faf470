// Fused RIME predict with per-antenna beam-cube DDEs for ANY uvw, SINGLE PRECISION: every input float32 / complex64,
// complex64 out -- the precision in which the reference runs its chain for such callers
// (africanus/util/type_inference.py:24-26: the promoted input type; africanus/rime/predict.py:542-544;
// africanus/rime/phase.py:28-61 and africanus/rime/fast_beam_cubes.py:57-240 with float32 arguments).
//
//     V[r,nu] = sum_s  E_p(s,t,nu) . ( K(r,s,nu) B(s,nu) ) . E_q(s,t,nu)^H
//
// The lane-per-row formulation of af_fused_predict.hip (rows that do not decompose by antenna: BASELINE configs[2] as it
// draws its uvw; Gaussian shapes; non-Hermitian brightness) re-designed for what single precision changes on this machine:
//   * the accumulating waves' Jones algebra -- M = G_p E_q^H, acc += y M: 48 FMAs per (row, source), two thirds of the fp64
//     kernel's issue slots -- is 24 packed instructions (v_pk_mul_f32 / v_pk_fma_f32 on (re, im) pairs: twice the fp64 rate);
//   * the phasor: the phase argument q = l u + m v + n w stays DOUBLE (|phase| reaches 1e4 rad: in float32 the result is lost,
//     as in the reference's float32 phase_delay), its reduction to a table step and a residual too (5 fp64 operations); the
//     table (1024 float2 entries, 8 KB) and the residual rotation (1 - t^2/2, t) are float32: 9 operations where the fp64
//     table phasor takes 12 fp64;
//   * Jones terms in LDS as float4 = two components of one antenna's term: a term is two ds_read_b128 instead of four, and a
//     batch of 16 sources x 64 antennas fits twice (the fp64 kernel: 8);
//   * the sampling waves read 64-byte float records (one 12-byte gather per correlation and corner) with the per-antenna part
//     of the coordinate map folded into six coefficients per (timestep, channel, antenna) (fused_voxels_folded: four FMAs per
//     term), as the single-precision GEMM form does.
// Workgroup: 12 waves -- lanes 0..511 accumulate (a lane owns four rows, or one GROUP of up to four rows that share their
// antennas: af_fused_plan_groups), lanes 512..767 sample the beam one batch ahead into the other half of a double-buffered
// Jones region; one barrier per batch.  Items / groups are the plans of af_fused_plan_rows / af_fused_plan_groups.
// The entry is CLOSER to the float64 chain on the same float32 inputs than the reference's own float32 chain
// (tests/test_gpu_fused_rows_c64.py: golden G17 and the fp64 kernel on the widened inputs).
#include <stdlib.h>

#include <type_traits>

#include "af_fused_device_f32.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int R_ACC = 512, R_SAMP = 256, R_THREADS = R_ACC + R_SAMP;
constexpr int RPT = 4;          // rows per accumulating lane
constexpr int PAIR_PAD = 2;     // float4 elements of padding per (source, component pair) plane

// a conj(b), acc += a conj(b), acc += a b on (re, im) pairs: two packed instructions each, the swaps and signs as operand
// modifiers (op_sel picks the half of a source that feeds the LOW result, op_sel_hi the HIGH one).  Written as asm: from
// vector expressions the compiler built the swapped / negated pairs with ~15 v_mov / v_xor per (row, source) beside the 24
// packed instructions.  A result whose HIGH half reads the LOW half of a source must not share that source's registers
// (early-clobber outputs); the accumulating forms read their accumulator straight (low -> low, high -> high).
__device__ __forceinline__ v2f cmulc2(v2f a, v2f b)
{
    v2f t, r;       // (ar br, ai br) then (+ ai bi, - ar bi)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=&v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=&v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// (the same in single steps, for callers that interleave several products)
__device__ __forceinline__ v2f cmulc2_a(v2f a, v2f b)
{
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=&v"(t) : "v"(a), "v"(b));
    return t;
}
__device__ __forceinline__ v2f cmulc2_b(v2f a, v2f b, v2f t)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=&v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
__device__ __forceinline__ void cmacc2_a(v2f &acc, v2f a, v2f b)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void cmacc2_b(v2f &acc, v2f a, v2f b)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void cmac2_a(v2f &acc, v2f a, v2f b)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void cmac2_b(v2f &acc, v2f a, v2f b)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void cmacc2(v2f &acc, v2f a, v2f b)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void cmac2(v2f &acc, v2f a, v2f b)
{
    // (+ ar br, + ar bi) then (- ai bi, + ai br)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));
}

// exp(2 pi i x / PH_TABLE) in float32 from a phase x kept in double (table steps): k = rint(x) by the 1.5 * 2^52 trick,
// theta = x - k in [-0.5, 0.5], table[k mod 1024] (float32) times (1 - t^2 / 2 + i t), t = 2 pi theta / 1024 <= 3.1e-3 (next
// terms 4.8e-9 and 3.5e-12: below float32 rounding).  Non-finite x -> NaN.  (af_im_to_vis_f32.hip's chain_phasor.)
__device__ __forceinline__ void table_init_f32(float2 *tab, int tid, int nthreads)
{
    for (int k = tid; k < PH_TABLE; k += nthreads) {
        double c, sn;
        sincos_quarter_turns<7>((double)k * (4.0 / PH_TABLE), c, sn);
        tab[k] = make_float2((float)c, (float)sn);
    }
}
__device__ __forceinline__ v2f phasor_f32(const float2 *__restrict__ tab, double x)
{
    const double MAGIC = 6755399441055744.0;
    const double a = __dadd_rn(x, MAGIC);
    const int k = __double2loint(a);
    const float th = (float)__dsub_rn(x, __dsub_rn(a, MAGIC));
    const float2 t = tab[k & (PH_TABLE - 1)];
    const float sn = __fmul_rn(th, 6.1359231515425649e-03f);                        // 2 pi / 1024
    const float cs = fmaf(__fmul_rn(th, th), -1.8824776459647568e-05f, 1.0f);       // (2 pi / 1024)^2 / 2
    // table entry times (cs + i sn): (tx cs, tx sn) then (- ty sn, + ty cs)
    const v2f tt = {t.x, t.y}, rs = {cs, sn};
    v2f p, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=&v"(p) : "v"(tt), "v"(rs));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=&v"(r) : "v"(tt), "v"(rs), "v"(p));
    return r;
}

// (el gs, em gs, er, is_extended) per source in double from the float32 shape parameters (gaussian_shape.py:45-50)
__global__ void prep_gauss_f32(const float *__restrict__ shape_params, int64_t nsrc, double gs, double *__restrict__ gp)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    double el = 0.0, em = 0.0, er = 0.0, ext = 0.0;
    if (shape_params != nullptr) {
        const double emaj = (double)shape_params[3 * s], emin = (double)shape_params[3 * s + 1], angle = (double)shape_params[3 * s + 2];
        el = emaj * sin(angle) * gs;
        em = emaj * cos(angle) * gs;
        er = emin / (emaj == 0.0 ? 1.0 : emaj);
        ext = (emaj != 0.0 || emin != 0.0) ? 1.0 : 0.0;
    }
    gp[4 * s + 0] = el; gp[4 * s + 1] = em; gp[4 * s + 2] = er; gp[4 * s + 3] = ext;
}

// grid: (nitems, channels of the plane group); block 768.  Dynamic LDS: six planes of per-antenna coefficients (double), the
// float2 phasor table, the feed rotations, then two Jones buffers, each E then G as [st][2][npp] float4 (components (0, 1) and
// (2, 3) of a term).  NP > 0: the antenna stride np is the compile-time constant NP >= nant; ST > 0: the batch size is the
// compile-time constant ST (the source loop is unrolled); GR: grouped rows (af_fused_plan_groups) -- even and odd antennas
// in separate halves of a plane, as in af_fused_predict.hip.
template <bool FEED, bool GAUSS, int NP, int ST, bool GR>
__global__ __launch_bounds__(R_THREADS) void fused_rows_c64_kernel(
    const int32_t *__restrict__ items, const int32_t *__restrict__ ant1, const int32_t *__restrict__ ant2,
    const int32_t *__restrict__ groups, const float *__restrict__ uvw, const double *__restrict__ lmn,
    const double *__restrict__ f4, const float2 *__restrict__ brightness, const float *__restrict__ vrec, int64_t beam_lw,
    int64_t beam_mh, int64_t beam_nud, const double *__restrict__ lm_ext, const double *__restrict__ freq_data,
    const float *__restrict__ parangles, const float *__restrict__ point_errors, const float *__restrict__ antenna_scaling,
    const float2 *__restrict__ feed_rot, const double *__restrict__ gauss, const double *__restrict__ freq_d, int nsrc,
    int64_t nchan, int64_t ntime, int nant, int st_arg, float2 *__restrict__ out, int64_t f0)
{
    extern __shared__ double lds_raw[];
    const int np = NP > 0 ? NP : nant;
    const int npa = (np + 1) & ~1;                 // coefficient planes: an even number of doubles (16-byte alignment behind)
    const int npp = np + PAIR_PAD;                 // float4 elements per (source, pair) plane
    const int st = ST > 0 ? ST : st_arg;
    const int half = npp >> 1;
    auto slot_of = [&](int a) { return GR ? (a & 1) * half + (a >> 1) : a; };
    double *ldsA = lds_raw;                                               // [6][npa]
    float2 *ldsT = reinterpret_cast<float2 *>(ldsA + 6 * npa);            // phasor table
    float2 *ldsR = ldsT + PH_TABLE;                                       // [npa][4] feed rotations
    v4f *ldsJ = reinterpret_cast<v4f *>(ldsR + 4 * npa);                  // Jones buffers
    const int tid = threadIdx.x;
    const bool consumer = tid < R_ACC;
    const int ptid = tid - R_ACC;
    const int64_t f = f0 + blockIdx.y;
    const int t = items[4 * blockIdx.x + 0];
    const int64_t r0 = items[4 * blockIdx.x + 1];
    const int rc = items[4 * blockIdx.x + 2];
    const int buf_elems = 2 * st * 2 * npp;                               // one buffer: E then G (float4 elements)

    // ---- per-antenna constants of this (timestep, channel) -------------------------------------------------------------
    table_init_f32(ldsT, tid, R_THREADS);
    const double FT = f4[f] * (PH_TABLE / 4.0);    // 1/PH_TABLE turns per metre
    const BeamGrid<double> bg = beam_grid<double>(lm_ext, beam_lw, beam_mh, beam_nud);
    const double fscale = freq_data[3 * f + 0];
    for (int a = tid; a < nant; a += R_THREADS) {
        double sp, cp;
        sincos((double)parangles[(int64_t)t * nant + a], &sp, &cp);
        const float *pe = point_errors + (((int64_t)t * nant + a) * nchan + f) * 2;
        const float *as = antenna_scaling + ((int64_t)a * nchan + f) * 2;
        const double pl = (double)pe[0], pm = (double)pe[1];
        const double kl = (double)as[0] * bg.lscale, km = (double)as[1] * bg.mscale;
        // the antenna's coordinate map, folded (fused_voxels_folded): one plane per coefficient
        ldsA[0 * npa + a] = fscale * cp * kl;
        ldsA[1 * npa + a] = -fscale * sp * kl;
        ldsA[2 * npa + a] = (pl * cp - pm * sp) * kl - bg.lower_l * bg.lscale;
        ldsA[3 * npa + a] = fscale * sp * km;
        ldsA[4 * npa + a] = fscale * cp * km;
        ldsA[5 * npa + a] = (pl * sp + pm * cp) * km - bg.lower_m * bg.mscale;
    }
    if constexpr (FEED)
        for (int i = tid; i < 4 * nant; i += R_THREADS) ldsR[i] = feed_rot[(int64_t)t * nant * 4 + i];
    __syncthreads();
    const int nbatch = (nsrc + st - 1) / st;

    if (!consumer) {
        // =================================== sampling waves (8-11) ======================================
        FusedGrid grid;
        grid.lower_l = grid.lower_m = grid.lscale = grid.mscale = 0.0;      // folded into the antennas' coefficients
        grid.lmaxf = wave_uniform(bg.lmaxf); grid.mmaxf = wave_uniform(bg.mmaxf);
        grid.lmaxi = __builtin_amdgcn_readfirstlane((int)bg.lmaxi); grid.mmaxi = __builtin_amdgcn_readfirstlane((int)bg.mmaxi);
        grid.stride_m = VREC32 * 4u;
        grid.stride_l = (unsigned)beam_mh * grid.stride_m;
        const int e_corr = ptid & 3, ei = e_corr >> 1, ej = e_corr & 1;
        const char *plane = reinterpret_cast<const char *>(vrec + (int64_t)blockIdx.y * beam_lw * beam_mh * VREC32);
        const unsigned corr_off = e_corr * 16u;
        const int ntask = st * np;
        // a lane's OWN term of a super-round of 256: in sampling round i the quads of a wave take the terms of quad lane i,
        // so lane (quad q, i) owns term 16 i + q of its wave's 64 -- the sixteen quads of one store then write sixteen
        // consecutive antennas (256 contiguous bytes per pair plane) instead of every fourth
        const int own_in_wave = (ptid & 3) * 16 + ((ptid & 63) >> 2);
        const int own_base = (ptid & ~63) + own_in_wave;
        __builtin_amdgcn_s_setprio(3);
        struct Own {
            unsigned base, dl, dm;
            float ld, md;
            int info;      // e_sl | e_ant << 11 | have_task << 30 | have << 31
        };
        struct Round {
            F3 v[4];
            float2 b0, b1;
        };
        for (int b = 0; b < nbatch; ++b) {
            const int s0 = b * st;
            float2 *E2 = reinterpret_cast<float2 *>(ldsJ + (size_t)(b & 1) * buf_elems);
            float2 *G2 = E2 + (size_t)st * 2 * npp * 2;
            auto fetch = [&](int task0, int &info) {
                const int task = task0 + own_base;
                int e_sl = task / np;          // NP > 0: a shift
                int e_ant = task - e_sl * np;
                const bool have_task = task < ntask && e_ant < nant;
                if (!have_task) e_sl = e_ant = 0;
                const bool have = have_task && s0 + e_sl < nsrc;
                info = e_sl | (e_ant << 11) | ((int)have_task << 30) | (int)((unsigned)have << 31);
                return *reinterpret_cast<const double2 *>(lmn + 4 * (have ? s0 + e_sl : 0));
            };
            int ninfo;
            double2 nlm = fetch(0, ninfo);
            for (int task0 = 0; task0 < ntask; task0 += R_SAMP) {
                Own S;
                S.info = ninfo;
                const double2 lm2 = nlm;
                if (task0 + R_SAMP < ntask) nlm = fetch(task0 + R_SAMP, ninfo);
                {
                    const int a = (S.info >> 11) & 1023;
                    FusedVoxelsC gx;
                    fused_voxels_folded(grid, lm2.x, lm2.y, ldsA[0 * npa + a], ldsA[1 * npa + a], ldsA[2 * npa + a],
                                        ldsA[3 * npa + a], ldsA[4 * npa + a], ldsA[5 * npa + a], gx);
                    S.base = gx.base; S.dl = gx.dl; S.dm = gx.dm;
                    S.ld = (float)gx.ld; S.md = (float)gx.md;
                }
                auto issue = [&](auto lane_c, Round &R) {
                    constexpr int QL = decltype(lane_c)::value;
                    const int info = quad_bcast<QL>(S.info);
                    const int e_sl = info & 2047;
                    const bool have = info < 0;
                    // G[c] = E[2(c/2)] . B[c%2] + E[2(c/2)+1] . B[2 + c%2]: this lane needs column c%2 of B
                    const float2 *bp = brightness + ((int64_t)(have ? s0 + e_sl : 0) * nchan + f) * 4;
                    R.b0 = bp[ej];
                    R.b1 = bp[2 + ej];
                    const unsigned base = (unsigned)quad_bcast<QL>((int)S.base) + corr_off;
                    const unsigned dl = (unsigned)quad_bcast<QL>((int)S.dl), dm = (unsigned)quad_bcast<QL>((int)S.dm);
                    const unsigned offs[4] = {base, base + dl, base + dm, base + dl + dm};
#pragma unroll
                    for (int k = 0; k < 4; ++k) R.v[k] = *reinterpret_cast<const F3 *>(plane + (size_t)offs[k]);
                };
                auto finish = [&](auto lane_c, const Round &R) {
                    constexpr int QL = decltype(lane_c)::value;
                    const int info = quad_bcast<QL>(S.info);
                    const int e_sl = info & 2047, e_ant = (info >> 11) & 1023;
                    const bool have_task = (info >> 30) & 1, have = info < 0;
                    const float ld = quad_bcastf<QL>(S.ld), md = quad_bcastf<QL>(S.md);
                    const float omld = __fsub_rn(1.0f, ld), ommd = __fsub_rn(1.0f, md);
                    const float wt[4] = {__fmul_rn(omld, ommd), __fmul_rn(ld, ommd), __fmul_rn(omld, md), __fmul_rn(ld, md)};
                    C2f e = beam_reduce1f(R.v, wt);
                    if (!have) { e.re = 0.0f; e.im = 0.0f; }
                    // E[2i] and E[2i+1] of this lane's row i of the Jones matrix live in the even / odd lane of its pair
                    C2f E0, E1;
                    E0.re = pair_bcastf<0>(e.re); E0.im = pair_bcastf<0>(e.im);
                    E1.re = pair_bcastf<1>(e.re); E1.im = pair_bcastf<1>(e.im);
                    if constexpr (FEED) {
                        // E <- E . R(t, antenna)  (einsum "stafij,tajk->stafik", rime/examples/predict.py:472)
                        const float2 q0 = ldsR[4 * e_ant + ej], q1 = ldsR[4 * e_ant + 2 + ej];
                        C2f Q0, Q1;
                        Q0.re = q0.x; Q0.im = q0.y; Q1.re = q1.x; Q1.im = q1.y;
                        e = cmulf(E0, Q0);
                        cmacf(e, E1, Q1);
                        E0.re = pair_bcastf<0>(e.re); E0.im = pair_bcastf<0>(e.im);
                        E1.re = pair_bcastf<1>(e.re); E1.im = pair_bcastf<1>(e.im);
                    }
                    C2f B0, B1;
                    B0.re = R.b0.x; B0.im = R.b0.y; B1.re = R.b1.x; B1.im = R.b1.y;
                    C2f Gv = cmulf(E0, B0);
                    cmacf(Gv, E1, B1);
                    if (have_task) {
                        const int at = ((e_sl * 2 + ei) * npp + slot_of(e_ant)) * 2 + ej;
                        E2[at] = make_float2(e.re, e.im);
                        G2[at] = have ? make_float2(Gv.re, Gv.im) : make_float2(0.0f, 0.0f);
                    }
                };
                using I0 = std::integral_constant<int, 0>;
                using I1 = std::integral_constant<int, 1>;
                using I2 = std::integral_constant<int, 2>;
                using I3 = std::integral_constant<int, 3>;
                Round R0, R1;
                issue(I0{}, R0); issue(I1{}, R1);
                finish(I0{}, R0); issue(I2{}, R0);
                finish(I1{}, R1); issue(I3{}, R1);
                finish(I2{}, R0); finish(I3{}, R1);
            }
            __syncthreads();
        }
        return;
    }

    // =================================== accumulating waves (0-7) ======================================
    const double GK = GAUSS ? freq_d[f] / FT : 0.0;     // Gaussian shapes: (u nu) = us * GK
    double us[RPT], vs[RPT], ws[RPT];
    int a1[RPT], a2[RPT];
    v2f acc[RPT][4];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int rl = tid + k * R_ACC;
        int64_t r;
        if constexpr (GR) {
            // item = (time, first group, group count): lane tid owns group r0 + tid; slot k = (i, j) = (k >> 1, k & 1)
            const bool have = tid < rc;
            const int32_t *g = groups + (r0 + (have ? tid : 0)) * 8;
            const int gr = g[4 + k];
            r = (have && gr >= 0) ? gr : 0;
            a1[k] = g[k >> 1];          // p_i
            a2[k] = g[2 + (k & 1)];     // q_j
        } else {
            r = r0 + (rl < rc ? rl : 0);
            a1[k] = ant1[r]; a2[k] = ant2[r];
        }
        const double u = (double)uvw[3 * r], v = (double)uvw[3 * r + 1], w = (double)uvw[3 * r + 2];
        us[k] = __dmul_rn(u, FT); vs[k] = __dmul_rn(v, FT); ws[k] = __dmul_rn(w, FT);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[k][c] = v2f{0.0f, 0.0f};
    }
    // float4-element offsets of this lane's Jones terms in the CURRENT buffer: E planes of antenna2 / q, G planes of antenna1 / p
    int eoff[RPT], goff[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        eoff[k] = slot_of(a2[k]);
        goff[k] = st * 2 * npp + slot_of(a1[k]);
    }
    for (int b = 0; b < nbatch; ++b) {
        const int s0 = b * st;
        __syncthreads();
        double Lh[ST > 0 ? ST : 1], Mh[ST > 0 ? ST : 1], Nh[ST > 0 ? ST : 1];
        if constexpr (ST > 0) {
#pragma unroll
            for (int sl = 0; sl < ST; ++sl) {
                // whole batches: a source beyond the last one has E = G = 0 in LDS and adds exactly nothing; only its
                // coordinates must come from a valid address
                const int sg = (s0 + sl >= nsrc) ? nsrc - 1 : s0 + sl;
                Lh[sl] = lmn[4 * sg]; Mh[sl] = lmn[4 * sg + 1]; Nh[sl] = lmn[4 * sg + 2];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        auto one_source = [&](int sl) {
            const int sg = (s0 + sl >= nsrc) ? nsrc - 1 : s0 + sl;
            double l, m, n;
            if constexpr (ST > 0) { l = Lh[sl]; m = Mh[sl]; n = Nh[sl]; }
            else { l = lmn[4 * sg]; m = lmn[4 * sg + 1]; n = lmn[4 * sg + 2]; }
            // Gaussian shape factors exp(-(u1^2 + v1^2) (nu gs)^2) of this lane's rows (gaussian_shape.py:52-60)
            float shape[RPT];
            bool extended = false;
            if constexpr (GAUSS) {
                const double gel = gauss[4 * sg], gem = gauss[4 * sg + 1], ger = gauss[4 * sg + 2];
                extended = gauss[4 * sg + 3] != 0.0;  // block-uniform
                if (extended) {
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        const double u1 = (us[k] * gem - vs[k] * gel) * ger * GK, v1 = (us[k] * gel + vs[k] * gem) * GK;
                        shape[k] = __expf(-(float)(u1 * u1 + v1 * v1));
                    }
                }
            }
            const int so = sl * 2 * npp;
            v2f Gp0, Gp1, Gp2, Gp3, Eq0, Eq1, Eq2, Eq3;
#define AF_LOAD_G(off) do { const v4f *b_ = ldsJ + ((off) + so); const v4f x0 = b_[0], x1 = b_[npp];                    \
        Gp0 = v2f{x0.x, x0.y}; Gp1 = v2f{x0.z, x0.w}; Gp2 = v2f{x1.x, x1.y}; Gp3 = v2f{x1.z, x1.w}; } while (0)
#define AF_LOAD_E(off) do { const v4f *b_ = ldsJ + ((off) + so); const v4f x0 = b_[0], x1 = b_[npp];                    \
        Eq0 = v2f{x0.x, x0.y}; Eq1 = v2f{x0.z, x0.w}; Eq2 = v2f{x1.x, x1.y}; Eq3 = v2f{x1.z, x1.w}; } while (0)
            // acc[k] += y M,  M = G_p . E_q^H :  M[i][j] = sum_k G[i][k] conj(E[j][k])
#define AF_ROW(k, LOADS) do {                                                                                   \
        v2f y = phasor_f32(ldsT, fma(n, ws[k], fma(m, vs[k], __dmul_rn(l, us[k]))));                              \
        if constexpr (GAUSS) { if (extended) y = y * v2f{shape[k], shape[k]}; }                                  \
        LOADS;                                                                                                   \
        /* the four elements of M side by side, step by step: no instruction reads the result of the one before it */ \
        v2f M0, M1, M2, M3;                                                                                      \
        { v2f t0 = cmulc2_a(Gp0, Eq0), t1 = cmulc2_a(Gp0, Eq2), t2 = cmulc2_a(Gp2, Eq0), t3 = cmulc2_a(Gp2, Eq2); \
          M0 = cmulc2_b(Gp0, Eq0, t0); M1 = cmulc2_b(Gp0, Eq2, t1); M2 = cmulc2_b(Gp2, Eq0, t2); M3 = cmulc2_b(Gp2, Eq2, t3); } \
        cmacc2_a(M0, Gp1, Eq1); cmacc2_a(M1, Gp1, Eq3); cmacc2_a(M2, Gp3, Eq1); cmacc2_a(M3, Gp3, Eq3);          \
        cmacc2_b(M0, Gp1, Eq1); cmacc2_b(M1, Gp1, Eq3); cmacc2_b(M2, Gp3, Eq1); cmacc2_b(M3, Gp3, Eq3);          \
        cmac2_a(acc[k][0], y, M0); cmac2_a(acc[k][1], y, M1); cmac2_a(acc[k][2], y, M2); cmac2_a(acc[k][3], y, M3); \
        cmac2_b(acc[k][0], y, M0); cmac2_b(acc[k][1], y, M1); cmac2_b(acc[k][2], y, M2); cmac2_b(acc[k][3], y, M3); \
    } while (0)
            if constexpr (GR) {
                // slots k = 2 i + j: (p0,q0) (p0,q1) (p1,q1) (p1,q0); every operand change is one two-read load
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
        if constexpr (ST > 0) {
#pragma unroll
            for (int sl = 0; sl < ST; ++sl) {
                one_source(sl);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const int nb = (nsrc - s0 < st) ? (nsrc - s0) : st;
#pragma unroll 1
            for (int sl = 0; sl < nb; ++sl) one_source(sl);
        }
        // on to the other buffer, in place
        const int step = (b & 1) ? -buf_elems : buf_elems;
#pragma unroll
        for (int k = 0; k < RPT; ++k) { eoff[k] += step; goff[k] += step; }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        int64_t orow = -1;
        if constexpr (GR) {
            if (tid < rc) orow = groups[(r0 + tid) * 8 + 4 + k];
        } else {
            if (tid + k * R_ACC < rc) orow = r0 + tid + k * R_ACC;
        }
        if (orow >= 0) {
            v4f *o = reinterpret_cast<v4f *>(out + (orow * nchan + f) * 4);
            o[0] = v4f{acc[k][0].x, acc[k][0].y, acc[k][1].x, acc[k][1].y};
            o[1] = v4f{acc[k][2].x, acc[k][2].y, acc[k][3].x, acc[k][3].y};
        }
    }
}

}  // namespace

// The single-precision form of af_fused_predict_c128 (include/afhip.h): items / groups as planned by af_fused_plan_rows /
// af_fused_plan_groups, every floating-point array float32 / complex64 (pairs of floats), out complex64.  Workspace:
// af_fused_predict_c64_workspace_bytes.
AF_EXPORT int af_fused_predict_c64(const int32_t *items, int64_t nitems, const int32_t *antenna1, const int32_t *antenna2,
                                   const int32_t *groups, int64_t nrow, const float *lm, const float *uvw,
                                   const float *frequency, const float *brightness, int64_t nsrc, int64_t nchan,
                                   const float *beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud,
                                   const float *beam_lm_extents, const float *beam_freq_map,
                                   const float *parallactic_angles, int64_t ntime, int64_t nant, const float *point_errors,
                                   const float *antenna_scaling, const float *feed_rotation, const float *gauss_shape,
                                   int convention, float *out, void *workspace, size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(convention == AF_CONVENTION_FOURIER || convention == AF_CONVENTION_CASA,
               "convention not in ('fourier', 'casa')");
    AF_REQUIRE(beam_lw >= 2 && beam_mh >= 2 && beam_nud >= 2, "beam_lw, beam_mh and beam_nud must be >= 2");
    AF_REQUIRE(nitems >= 0 && nrow >= 0 && nsrc >= 0 && nchan >= 0 && ntime >= 0 && nant >= 0,
               "af_fused_predict_c64: negative extent");
    AF_REQUIRE(nant <= 664, "af_fused_predict_c64: more than 664 antennas");
    AF_REQUIRE(nsrc < (1LL << 31) && nchan <= 65535 && nitems < (1LL << 31), "af_fused_predict_c64: too large");
    hipStream_t st_ = af_stream(stream);
    if (nrow == 0 || nchan == 0) return AF_OK;
    AF_REQUIRE(out != nullptr, "af_fused_predict_c64: out is NULL");
    if (nsrc == 0 || nitems == 0) {
        AF_HIP(hipMemsetAsync(out, 0, sizeof(float) * 2 * 4 * (size_t)(nrow * nchan), st_));
        return AF_OK;
    }
    AF_REQUIRE(items && (groups || (antenna1 && antenna2)) && lm && uvw && frequency && brightness && beam && beam_lm_extents &&
                   beam_freq_map && parallactic_angles && point_errors && antenna_scaling,
               "af_fused_predict_c64: NULL array");
    const WsS W = ws_s(nsrc, nchan, beam_lw, beam_mh, beam_nud);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= W.total, "af_fused_predict_c64: workspace too small (%zu < %zu)",
               workspace_bytes, W.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_fused_predict_c64: workspace must be 256-byte aligned");
    char *ws = static_cast<char *>(workspace);
    double *lmn = reinterpret_cast<double *>(ws + W.lmn), *f4 = reinterpret_cast<double *>(ws + W.f4);
    double *freq_d = reinterpret_cast<double *>(ws + W.freq_d), *fmap_d = reinterpret_cast<double *>(ws + W.fmap_d);
    double *ext_d = reinterpret_cast<double *>(ws + W.ext_d), *freq_data = reinterpret_cast<double *>(ws + W.freq_data);
    double *gp = reinterpret_cast<double *>(ws + W.gauss);
    float *planes_buf = reinterpret_cast<float *>(ws + W.planes);
    hipLaunchKernelGGL(prep_src_f32, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st_, lm, nsrc, lmn);
    AF_LAUNCH_CHECK();
    const int64_t nprep = nchan > beam_nud ? (nchan > 4 ? nchan : 4) : (beam_nud > 4 ? beam_nud : 4);
    hipLaunchKernelGGL(prep_freq_f32, dim3((unsigned)af_cdiv(nprep, 64)), dim3(64), 0, st_, frequency, nchan, convention,
                       beam_freq_map, beam_nud, beam_lm_extents, f4, freq_d, fmap_d, ext_d);
    AF_LAUNCH_CHECK();
    int rc = af_freq_grid_interp_f64(freq_d, nchan, fmap_d, beam_nud, freq_data, stream);
    if (rc != AF_OK) return rc;
    const bool feed = feed_rotation != nullptr, gauss = gauss_shape != nullptr;
    if (gauss) {
        const double fwhm = 2.0 * sqrt(2.0 * log(2.0));  // gaussian_shape.py:23-25
        const double gs = (1.0 / fwhm) * sqrt(2.0) * 3.141592653589793 / AF_LIGHTSPEED;
        hipLaunchKernelGGL(prep_gauss_f32, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st_, gauss_shape, nsrc, gs, gp);
        AF_LAUNCH_CHECK();
    }
    const int64_t ncell = beam_lw * beam_mh;
    AF_REQUIRE(ncell < (1LL << 25), "af_fused_predict_c64: beam cube too large (fewer than 2^25 cells per plane)");
    // antenna stride of the Jones arrays: a compile-time constant for the common array sizes
    const int NPv = (nant > 32 && nant <= 64) ? 64 : (nant > 64 && nant <= 128) ? 128 : 0;
    const int64_t np = NPv ? NPv : nant, npa = (np + 1) & ~1LL, npp = np + PAIR_PAD;
    const int64_t fixed = 48 * npa + PH_TABLE * 8 + 32 * npa;   // coefficient planes, phasor table, feed rotations
    // sources per batch: two buffers of E and G (64 bytes per (source, antenna)); whole super-rounds of the 256 sampling lanes
    int st = (int)((160 * 1024 - fixed) / (2 * 64 * npp));
    if (st > 2047) st = 2047;                                // (a term's source slot is 11 bits of its `info` word)
    if ((int64_t)st * np > 4096) st = (int)(4096 / np);     // (at most 16 super-rounds per batch)
    {
        int64_t m = 1;
        while ((m * np) % R_SAMP != 0 && m < R_SAMP) ++m;
        if (st >= m) st -= st % (int)m;
    }
    if (st > nsrc) st = (int)nsrc;
    if (st < 1) st = 1;
    // the unrolled form: 64-antenna stride, 8 sources per batch (no Gaussian shapes).  Measured on one box at BASELINE
    // configs[2]: 109.3 ms; the rolled loop on batches of 16 sources (half the barriers) 113.8 ms
    const bool unroll = NPv == 64 && !gauss && nsrc >= 8;
    if (unroll) st = 8;
    const size_t lds_bytes = (size_t)fixed + (size_t)2 * 2 * st * 2 * npp * 16;
    AF_REQUIRE(lds_bytes <= 160 * 1024, "af_fused_predict_c64: %zu bytes of LDS needed (nant = %lld)", lds_bytes, (long long)nant);
    auto launch = [&](auto kernel) -> int {
        AF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds_bytes));
        for (int64_t f0 = 0; f0 < nchan; f0 += PLANE_GROUP) {
            const int64_t nf = nchan - f0 < PLANE_GROUP ? nchan - f0 : PLANE_GROUP;
            int64_t blocks = af_cdiv(ncell * 4, 256);
            if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(beam_plane_kernel_f32, dim3((unsigned)blocks, (unsigned)nf), dim3(256), 0, st_,
                               reinterpret_cast<const float2 *>(beam), ncell, beam_nud, freq_data, f0, planes_buf);
            AF_LAUNCH_CHECK();
            if (f0 == 0) af_prof_begin(st_);
            hipLaunchKernelGGL(kernel, dim3((unsigned)nitems, (unsigned)nf), dim3(R_THREADS), lds_bytes, st_, items, antenna1,
                               antenna2, groups, uvw, lmn, f4, reinterpret_cast<const float2 *>(brightness), planes_buf, beam_lw,
                               beam_mh, beam_nud, ext_d, freq_data, parallactic_angles, point_errors, antenna_scaling,
                               reinterpret_cast<const float2 *>(feed_rotation), gp, freq_d, (int)nsrc, nchan, ntime, (int)nant,
                               st, reinterpret_cast<float2 *>(out), f0);
            if (f0 == 0) af_prof_end(st_);
            AF_LAUNCH_CHECK();
        }
        return AF_OK;
    };
#define AF_ROWS_PICK(NPC, STC, GRC)                                                                                      \
    (feed ? (gauss ? launch(fused_rows_c64_kernel<true, true, NPC, 0, GRC>)                                              \
                   : launch(fused_rows_c64_kernel<true, false, NPC, STC, GRC>))                                          \
          : (gauss ? launch(fused_rows_c64_kernel<false, true, NPC, 0, GRC>)                                             \
                   : launch(fused_rows_c64_kernel<false, false, NPC, STC, GRC>)))
    if (groups != nullptr) {
        if (unroll) rc = AF_ROWS_PICK(64, 8, true);
        else rc = NPv == 64 ? AF_ROWS_PICK(64, 0, true) : NPv == 128 ? AF_ROWS_PICK(128, 0, true) : AF_ROWS_PICK(0, 0, true);
    } else if (unroll) rc = AF_ROWS_PICK(64, 8, false);
    else rc = NPv == 64 ? AF_ROWS_PICK(64, 0, false) : NPv == 128 ? AF_ROWS_PICK(128, 0, false) : AF_ROWS_PICK(0, 0, false);
#undef AF_ROWS_PICK
    return rc;
}
