// im_to_vis, MFMA-accumulator path for gfx950: real images with 4 correlations on a uniformly spaced band.
//
// Same sum as af_im_to_vis.hip (africanus/dft/kernels.py:33-67), mapped so that the 4 correlations
// are the N dimension of v_mfma_f64_4x4x4_4b and the accumulators live in the MFMA C/D registers:
//   * a wave owns 16 rows x a tile of CT channels (CT = 64: 2 x 64 accumulators = all 256 AGPRs);
//     one MFMA step contracts 4 sources:  D[row, corr] += sum_k Y[row, src k] * I[src k, corr]
//     per channel, once for Re(Y) and once for Im(Y).  Lane layout of the instruction (measured,
//     tools/probe_mfma_f64.hip): A lane = 16 k + (row & 15), B lane = 16 k + 4 b + corr (the same
//     value for the four row blocks b), D lane = 16 (row & 3) + 4 (row >> 2) + corr.
//   * so every lane owns ONE (row, source) pair of the step and its VALU work is that pair's phasor
//     only: path difference, two quarter-turn sincos (tile start, channel step), then the
//     three-term recurrence over the tile, run as four independent chains (re/im x even/odd
//     channels, step 2 delta) one 8-channel group ahead of the MFMAs that consume it (error growth
//     (j/2)^2 eps <= 4e-13 at the end of a 64-channel tile: no re-anchoring, MFMA_ANCHOR = 64).
//     fp64 MFMA and fp64 VALU share one pipe on this chip (measured), so the
//     gain over the VALU kernels is not rate but registers: with the accumulators out of the
//     arch VGPRs a tile holds 64 channels instead of 13 and the per-tile setup is amortised 5x better.
//   * the image pixels of a 4-source step (CT x 4 x 4 doubles) and the NEXT step's (l,m,n) come as one
//     contiguous record, copied global -> LDS by global_load_lds_dwordx4 one step ahead (two LDS
//     stages, one barrier per step) and read as ds_read_b128 (two channels per read) in the shadow of
//     the MFMAs; the set-up of step it+1 (table phasors) is sliced over the channel groups of step it.
//   * measured (tools/microbench_issue.hip): a lone wave issues an fp64 VALU op every ~5.5 cycles and an
//     fp64 MFMA 4x4x4 every 16.4; per step 128 MFMAs + ~220 VALU ops = ~3300 cycles against 3500 measured.
#include <stdlib.h>

#include <type_traits>

#include "af_dft_mfma.h"
#include "af_mfma_phasor.h"

namespace {

constexpr int THREADS = 256;        // 4 waves x 16 rows

// doubles of one (tile, step) record: header + CT channels x 16 (source k, corr) x 1 (real pixel) or 3 (Re, Im, -Im)
// (Gaussian sources, af_gauss_mfma_run: a second header row with the NEXT step's shape parameters)
__host__ __device__ constexpr int stage_doubles(int ct, bool cplx, bool gauss = false)
{
    return (ct * (cplx ? 3 : 1) + 1 + (gauss ? 1 : 0)) * 16;
}

// record of (tile, step): [ (l,m,n,0) x 4 sources of step+1 | CT/2 channel pairs x 16 (source k, corr n) x 2 channels ];
// complex images: ... x 2 channels x (Re, Im, -Im) -- the -Im copy lets Re(Y I) = ReY ReI + ImY (-ImI) run as
// two plain MFMAs (the instruction has no operand negation)
__global__ void mfma_pack_records(const double *__restrict__ image, int cplx, const double *__restrict__ lmn,
                                  const int *__restrict__ srcbad, int64_t nsrc, int64_t nit, int64_t nchan, int64_t c0,
                                  int CT, double *__restrict__ rec, const double *__restrict__ gauss = nullptr)
{
    const int64_t per = stage_doubles(CT, cplx != 0, gauss != nullptr);
    const int64_t total = nit * per;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t it = i / per;
        int64_t idx = i - it * per;
        double v = 0.0;
        if (idx < 16) {  // (l,m,n) of the NEXT step: its set-up runs during this one
            const int64_t s = 4 * (it + 1) + idx / 4;
            if (s < nsrc && (idx & 3) < 3) v = lmn[4 * s + (idx & 3)];
            rec[i] = v;
            continue;
        }
        if (gauss != nullptr) {  // second header row: (el, em, er, 0) of the next step's sources
            idx -= 16;
            if (idx < 16) {
                const int64_t s = 4 * (it + 1) + idx / 4;
                if (s < nsrc && (idx & 3) < 3) v = gauss[4 * s + (idx & 3)];
                rec[i] = v;
                continue;
            }
        }
        if (!cplx) {
            const int64_t e = idx - 16, pair = e / 32, r = e - pair * 32;
            const int64_t kn = r >> 1, j = 2 * pair + (r & 1);
            const int64_t s = 4 * it + (kn >> 2), ch = c0 + j;
            if (s < nsrc && ch < nchan && !srcbad[s]) v = image[(s * nchan + ch) * 4 + (kn & 3)];
        } else {
            const int64_t e = idx - 16, pair = e / 96, r = e - pair * 96;
            const int64_t kn = r / 6, t = r - kn * 6, j = 2 * pair + (t >= 3), comp = t % 3;
            const int64_t s = 4 * it + (kn >> 2), ch = c0 + j;
            if (s < nsrc && ch < nchan && !srcbad[s]) {
                const double *px = image + ((s * nchan + ch) * 4 + (kn & 3)) * 2;
                v = comp == 0 ? px[0] : (comp == 1 ? px[1] : -px[1]);
            }
        }
        rec[i] = v;
    }
}

// quarter turns per metre at the first channel of every tile of the launch
__global__ void mfma_tile_f0(const double *__restrict__ freq, int64_t nchan, int64_t c0, int CT, int ntile, int sign,
                             double *__restrict__ f0)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntile) f0[t] = 4.0 * (double)sign * freq[c0 + (int64_t)t * CT] / AF_LIGHTSPEED;
}

// grid: (ceil(nrow/64), tiles of the launch); block: 4 waves, wave w on rows 64 bx + 16 w ...
// x + (x of the lane the DPP control names): 64-lane sums without LDS round trips
template <int CTRL> __device__ __forceinline__ double chi_dpp_add(double x)
{
    const double y = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true),
                                      __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true));
    return x + y;
}
template <int LANE> __device__ __forceinline__ double chi_readlane(double x)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), LANE), __builtin_amdgcn_readlane(__double2loint(x), LANE));
}

// CHI2: the epilogue also adds sum [w] |data - vis|^2 over the block's rows and the four correlations to chi2[chan]
// (the visibilities are in registers there: the separate chi^2 pass re-reads all of them)
// GAUSS (complex pixels = brightness matrices): the lane's phasor carries the source's Gaussian envelope
//     shape(row, s, nu) = exp(-a nu^2),   a = u1^2 + v1^2,  u1 = (u em - v el) er,  v1 = u el + v em
// (africanus/model/shape/gaussian_shape.py:45-60) -- on the tile's arithmetic progression nu_j = nu_0 + j d a
// second-order PRODUCT recurrence e_(j+1) = e_j r_j, r_(j+1) = r_j c with e_0 = exp(-a nu_0^2),
// r_0 = exp(-a (2 nu_0 d + d^2)), c = exp(-2 a d^2): three exponentials per (row, source, tile), spread over the
// set-up slices, and four products per channel (e, r, Re, Im).  The phasors themselves keep their three-term recurrence:
// Y (8 channels, advanced in place) is the recurrence state, Z = e Y the MFMA operand.  The shape parameters of a step's
// four sources travel in a second header row of the record, pre-scaled so that nu is in the kernel's 1/256-turn units.
// A point source (a = 0) has e = r = c = 1 exactly.
struct GaussEnvelope {
    double el, em, er;     // shape parameters of the lane's source
    double a;              // u1^2 + v1^2
    double e0, r0, c;      // recurrence start of the tile
};
// the envelope's slices of a step's set-up: 1 quadratic form, 2 e_0, 3 r_0, 6 c (hdr: the record's second header row)
__device__ __forceinline__ void gauss_setup_slice(GaussEnvelope &G, int slice, const double *hdr4, double u, double v,
                                                  double q0, double q1, double q2)
{
    switch (slice) {
    case 0: {
        const double2 h = *reinterpret_cast<const double2 *>(hdr4);
        G.el = h.x; G.em = h.y; G.er = hdr4[2];
        break;
    }
    case 1: {
        const double u1 = __dmul_rn(fma(u, G.em, -__dmul_rn(v, G.el)), G.er), v1 = fma(u, G.el, __dmul_rn(v, G.em));
        G.a = fma(u1, u1, __dmul_rn(v1, v1));
        break;
    }
    case 2: G.e0 = exp_neg(__dmul_rn(G.a, q0)); break;
    // (a falling band has r_0 > 1; beyond e^700 the envelope is 0 on the whole tile, and 0 x inf must not appear)
    case 3: G.r0 = exp_neg(fmax(__dmul_rn(G.a, q1), -700.0)); break;
    case 6: G.c = exp_neg(__dmul_rn(G.a, q2)); break;
    default: break;
    }
}

template <int CT, bool CPLX, bool CHI2 = false, bool GAUSS = false>
__global__ __launch_bounds__(THREADS) void dft_mfma_kernel(
    const double *__restrict__ uvw, const double *__restrict__ records, const double *__restrict__ tile_f0,
    const double *__restrict__ tilef, const int *__restrict__ flags, const double *__restrict__ lmn,
    double *__restrict__ out, int64_t nrow, int nsrc, int nit, int64_t nchan, int64_t c0_first,
    const double2 *__restrict__ chi_data = nullptr, const double *__restrict__ chi_weight = nullptr,
    double *__restrict__ chi2 = nullptr, const double *__restrict__ gauss = nullptr)
{
    static_assert(!GAUSS || CPLX, "the Gaussian variant takes brightness matrices");
    if (flags[0] != 1 || flags[1] != 1) return;  // one channel spacing for the whole band, decided on the device
    constexpr int STAGE = stage_doubles(CT, CPLX, GAUSS);
    constexpr int HDR = GAUSS ? 32 : 16;          // doubles of the record's header
    constexpr int UNITS = STAGE / 2;              // 16-byte units of a stage
    __shared__ double smem[2 * STAGE];
    __shared__ double2 ptab[PHASOR_TABLE];   // exp(2 pi i k / 256): af_sincos.h table phasor
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = lane >> 4;                      // the source of a step this lane computes the phasor of
    const int tile = blockIdx.y;
    const int64_t c0 = c0_first + (int64_t)tile * CT;
    const double *__restrict__ rec = records + (int64_t)tile * nit * STAGE;
    int64_t row = (int64_t)blockIdx.x * 64 + wave * 16 + (lane & 15);
    if (row >= nrow) row = nrow - 1;
    const double u = uvw[3 * row], v = uvw[3 * row + 1], w = uvw[3 * row + 2];
    const double F0 = 64.0 * tile_f0[tile], FD = 64.0 * tilef[1];   // quarter turns -> 1/256 turns per metre (exact)
    const int boff = (k * 4 + (lane & 3)) * (CPLX ? 6 : 2);  // B operand: pixel(s) of (source k, corr lane & 3)

    double are[CT], aim[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) are[j] = aim[j] = 0.0;

    mfma_stage_load<UNITS>(rec, smem, wave, lane);
    table_phasor_init(ptab, tid, THREADS);
    asm volatile("" :: "v"(u), "v"(v), "v"(w), "s"(F0), "s"(FD));  // hipcc's own waits land here, not in the loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    constexpr int GP = 4;                         // channel pairs per B register group (8 channels)
    constexpr int NGRP = CT / 2 / GP;

    double yr[2][8], yi[2][8];                    // GAUSS: yr[0] / yi[0] = Y (recurrence state), yr[1] / yi[1] = Z = e Y
    PhasorSetup cur_, nxt_;
    GaussEnvelope env_;                           // of the NEXT step while a step runs
    double ee = 1.0, er = 1.0, ec = 1.0;          // envelope recurrence of the current step
    // exponents per unit of a: nu_0^2, 2 nu_0 d + d^2, 2 d^2 in the kernel's frequency units (block-uniform)
    const double q0 = F0 * F0, q1 = fma(2.0 * F0, FD, FD * FD), q2 = 2.0 * FD * FD;
    {   // step 0 is set up in one piece from the global (l,m,n)
        cur_.a0 = cur_.a1 = cur_.a2 = 0.0;
        if (k < nsrc) { cur_.a0 = lmn[4 * k]; cur_.a1 = lmn[4 * k + 1]; cur_.a2 = lmn[4 * k + 2]; }
#pragma unroll
        for (int sl = 1; sl < 8; ++sl) phasor_setup_slice(cur_, sl, nullptr, u, v, w, F0, FD, ptab, yr[0], yi[0]);
        if constexpr (GAUSS) {
            env_.el = env_.em = env_.er = 0.0;
            if (k < nsrc) { env_.el = gauss[4 * k]; env_.em = gauss[4 * k + 1]; env_.er = gauss[4 * k + 2]; }
#pragma unroll
            for (int sl = 1; sl < 8; ++sl) gauss_setup_slice(env_, sl, nullptr, u, v, q0, q1, q2);
            ee = env_.e0; er = env_.r0; ec = env_.c;
        }
    }
    nxt_ = cur_;

#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
        const int cur = it & 1;
        if (it + 1 < nit) mfma_stage_load<UNITS>(rec + (int64_t)(it + 1) * STAGE, smem + (cur ^ 1) * STAGE, wave, lane);
        const double *S = smem + cur * STAGE;     // header: (l,m,n) of step it + 1
        const double2 *B = reinterpret_cast<const double2 *>(S + HDR + boff);
        constexpr int BPG = CPLX ? 12 : GP;       // 16-byte B reads per group of 8 channels
        constexpr int BSTRIDE = CPLX ? 48 : 16;   // double2 units between channel pairs
        double2 bg[2][BPG];
        double anr = cur_.y0r, ani = cur_.y0i;    // phasor at the current anchor (re-anchoring only if MFMA_ANCHOR < CT)
        auto load_b = [&](int g, int t) {         // t-th read of group g
            if constexpr (CPLX) bg[g & 1][t] = B[(g * GP + t / 3) * BSTRIDE + (t % 3)];
            else bg[g & 1][t] = B[(g * GP + t) * BSTRIDE];
        };
        // pixel components of channel jj of group g: real image -> b; complex -> (br, bi, -bi)
        auto pix = [&](int g, int jj, int comp) -> double {
            if constexpr (CPLX) {
                const int f = (jj >> 1) * 6 + (jj & 1) * 3 + comp;  // position among the pair's 6 doubles
                return (f & 1) ? bg[g & 1][f >> 1].y : bg[g & 1][f >> 1].x;
            } else {
                return (jj & 1) ? bg[g & 1][jj >> 1].y : bg[g & 1][jj >> 1].x;
            }
        };
#pragma unroll
        for (int t = 0; t < BPG; ++t) load_b(0, t);
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            if constexpr (GAUSS) {
                // Z of THIS group from Y (the previous group's MFMAs have been issued), then Y moves on in place
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    yr[1][t] = __dmul_rn(yr[0][t], ee);
                    yi[1][t] = __dmul_rn(yi[0][t], ee);
                    ee = __dmul_rn(ee, er);
                    er = __dmul_rn(er, ec);
                }
                if (g + 1 < NGRP) {
                    static_assert(!GAUSS || CT <= MFMA_ANCHOR, "no re-anchoring in the Gaussian variant");
#pragma unroll
                    for (int t = 0; t < 8; ++t) {      // in place: Y[t] <- k2 Y[t-2] - Y[t-4] over the group boundary
                        const double r2 = t >= 2 ? yr[0][t - 2] : yr[0][t + 6], r4 = t >= 4 ? yr[0][t - 4] : yr[0][t + 4];
                        const double i2 = t >= 2 ? yi[0][t - 2] : yi[0][t + 6], i4 = t >= 4 ? yi[0][t - 4] : yi[0][t + 4];
                        yr[0][t] = fma(cur_.k2, r2, -r4); yi[0][t] = fma(cur_.k2, i2, -i4);
                    }
                }
            } else
            if (g + 1 < NGRP) {  // phasors of the next 8 channels
                if (((g + 1) * 8) % MFMA_ANCHOR == 0) {
                    const double tr = fma(anr, cur_.ar, -__dmul_rn(ani, cur_.ai));
                    const double ti = fma(anr, cur_.ai, __dmul_rn(ani, cur_.ar));
                    anr = tr; ani = ti;
                    phasor_first_segment(cur_, anr, ani, yr[(g + 1) & 1], yi[(g + 1) & 1]);
                } else {
                    phasor_next_segment(cur_, yr[(g + 1) & 1], yi[(g + 1) & 1], yr[g & 1], yi[g & 1]);
                }
            }
            // slices of the next step's set-up that belong to this group (slice 7 writes yr[0], free since
            // the MFMAs of group NGRP-2 were issued)
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)
                if (sl * NGRP / 8 == g) {
                    phasor_setup_slice(nxt_, sl, S + 4 * k, u, v, w, F0, FD, ptab, yr[0], yi[0]);
                    if constexpr (GAUSS) gauss_setup_slice(env_, sl, S + 16 + 4 * k, u, v, q0, q1, q2);
                }
            __builtin_amdgcn_sched_barrier(0);
            const int zb = GAUSS ? 1 : (g & 1);   // the MFMAs' A operands: Z (Gaussian) or this group's phasors
            if constexpr (!CPLX) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = g * 8 + jj;
                    const double b = pix(g, jj, 0);
                    are[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(yr[g & 1][jj], b, are[j], 0, 0, 0);
                    aim[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(yi[g & 1][jj], b, aim[j], 0, 0, 0);
                    // pixels of the next 8 channels: LDS reads issue in the shadow of the 16-cycle MFMAs
                    if (g + 1 < NGRP && (jj & 1)) {
                        load_b(g + 1, jj >> 1);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                    }
                }
            } else {
                // vis += (yr + i yi)(br + i bi): four MFMAs per channel on two accumulators; the four passes
                // keep MFMAs on the same accumulator 16 instructions apart
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int j = g * 8 + jj;
                        if (pass == 0) are[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(yr[zb][jj], pix(g, jj, 0), are[j], 0, 0, 0);
                        if (pass == 1) aim[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(yr[zb][jj], pix(g, jj, 1), aim[j], 0, 0, 0);
                        if (pass == 2) are[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(yi[zb][jj], pix(g, jj, 2), are[j], 0, 0, 0);
                        if (pass == 3) aim[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(yi[zb][jj], pix(g, jj, 0), aim[j], 0, 0, 0);
                        const int t = pass * 8 + jj;
                        if (g + 1 < NGRP && t < BPG) {
                            load_b(g + 1, t);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur_ = nxt_;
        if constexpr (GAUSS) { ee = env_.e0; er = env_.r0; ec = env_.c; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next stage has landed in LDS
        __syncthreads();
    }

    // D lane = 16 i + 4 b + corr holds row 4 b + i: a wave store covers 16 rows x 64 contiguous bytes,
    // consecutive channels complete the 128-byte lines
    const int orow = 4 * ((lane >> 2) & 3) + (lane >> 4), ocorr = lane & 3;
    const int64_t r = (int64_t)blockIdx.x * 64 + wave * 16 + orow;
    const int nvalid = (int)((nchan - c0 < CT) ? (nchan - c0) : CT);  // block-uniform: the band's last tile may be short
    const bool live = r < nrow;
    if (!CHI2 && !live) return;
    double2 *__restrict__ o = reinterpret_cast<double2 *>(out) + ((live ? r : 0) * nchan + c0) * 4 + ocorr;
    if (live) {
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            if (j < nvalid) o[j * 4] = make_double2(are[j], aim[j]);
        }
    }
    if constexpr (CHI2) {
        // chi^2 of the rows just written.  The accumulators are DEAD here: the sums read the visibilities back (the lane's
        // own stores, served by L2) four channels at a time, so this part needs ~40 registers and the kernel keeps the main
        // loop's 128 + 128 (with the differences taken from the accumulators the 32-channel kernel needed 143-160 VGPRs: ONE
        // wave per SIMD instead of two, 26.9 instead of 20.4 ms).  The pointer goes through an empty asm so that the
        // compiler does not forward the stored registers into the loads.  64-lane sums: DPP adds inside the 16-lane rows
        // (v_mov_dpp quad_perm / row_half_mirror / row_mirror: no LDS round trips), four v_readlane across the rows.
        // Every lane stays to the end (a block barrier); rows beyond the last add nothing.  The stage buffers are free:
        // the loop's last barrier is behind every wave.
        const double2 *vis_back = o;
        asm volatile("" : "+v"(vis_back));
        const int64_t cell = ((live ? r : 0) * nchan + c0) * 4 + ocorr;
        const double2 *__restrict__ dat = chi_data + cell;
        const double *__restrict__ wgt = chi_weight ? chi_weight + cell : nullptr;
        double *part = smem;                                  // [wave][CT]
        // channels whose loads are in flight together (8: no change, 21.97 against 21.98 ms; 16: 168 VGPRs, one wave per SIMD)
        constexpr int CG = 4;
#pragma unroll
        for (int g = 0; g < CT / CG; ++g) {
            double2 d[CG], m[CG];
            double wv[CG];
#pragma unroll
            for (int k = 0; k < CG; ++k) {
                const int j = CG * g + k;
                const bool on = j < nvalid && live;
                m[k] = on ? vis_back[j * 4] : make_double2(0.0, 0.0);
                d[k] = on ? dat[j * 4] : make_double2(0.0, 0.0);
                wv[k] = (on && wgt) ? wgt[j * 4] : 1.0;
            }
#pragma unroll
            for (int k = 0; k < CG; ++k) {
                const int j = CG * g + k;
                const double dr = d[k].x - m[k].x, di = d[k].y - m[k].y;
                double a = fma(dr, dr, di * di) * wv[k];
                a = chi_dpp_add<0xB1>(a);       // quad_perm [1,0,3,2]
                a = chi_dpp_add<0x4E>(a);       // quad_perm [2,3,0,1]
                a = chi_dpp_add<0x141>(a);      // row_half_mirror: the other quad of the 8
                a = chi_dpp_add<0x140>(a);      // row_mirror: the other 8 of the row
                const double t01 = chi_readlane<0>(a) + chi_readlane<16>(a), t23 = chi_readlane<32>(a) + chi_readlane<48>(a);
                if (lane == 0) part[wave * CT + j] = t01 + t23;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (tid < nvalid) {
            const double total = (part[tid] + part[CT + tid]) + (part[2 * CT + tid] + part[3 * CT + tid]);
            atomicAdd(&chi2[c0 + tid], total);
        }
    }
}

// The reference skips zero pixels (`if image[s,nu,c]:`, kernels.py:64): an all-zero (chan, corr) column
// stays exactly 0 and a non-finite source poisons only the columns where its pixel is nonzero.  The
// main kernel accumulates zeros for both; this pass rewrites the (rare) special columns.  One thread
// per (row, column), grid-stride; returns at once when dft_colstate found no special column (flags[2] == 0).
__global__ void mfma_fix_columns(const int *__restrict__ flags, const int *__restrict__ colstate,
                                 double *__restrict__ out, int64_t nrow, int64_t nchan)
{
    if (flags[0] != 1 || flags[1] != 1 || flags[2] == 0) return;
    const int64_t ncol = nchan * 4, total = nrow * ncol;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int st = colstate[i % ncol];
        if (st == 0) continue;
        const double v = st == 1 ? 0.0 : __longlong_as_double(0x7ff8000000000000LL);
        reinterpret_cast<double2 *>(out)[i] = make_double2(v, v);
    }
}

struct Plan {
    int ct;              // width of the main tiles: 64 or 32 channels (main_ct)
    int64_t nfull;       // main tiles (a last partial tile wider than half a tile counts)
    int tail_ct;         // 0, 16 or 32: width of the last tile when narrower
    int64_t tail_c0;
    size_t f0_off, rec_off, tail_rec_off, total;
};

// Tile width of the band's main launch.  Real images: 32 channels -- 128 AGPRs + 128 VGPRs, so TWO waves share a SIMD
// and one wave's MFMAs run in the other's issue gaps (a lone wave issues an fp64 VALU op every ~5.5 cycles, not 4):
// 20.7 ms against 21.6 ms with 64-channel tiles at 1e6 x 64 x 1000, although the per-tile set-up is amortised over
// half as many channels.  Complex images need 340 registers at 32 channels (one wave per SIMD either way): 64.
// AFHIP_DFT_MFMA_CT=64 / 32 overrides (A/B measurements).
int main_ct(bool cplx)
{
    static const int env = getenv("AFHIP_DFT_MFMA_CT") ? atoi(getenv("AFHIP_DFT_MFMA_CT")) : 0;
    if (env == 64 || (env == 32 && !cplx)) return env;
    return cplx ? 64 : 32;
}

Plan make_plan(int64_t nsrc_pad, int64_t nchan, bool cplx, bool gauss = false)
{
    Plan p;
    const int ct = main_ct(cplx), half = ct / 2;
    const int64_t nit = nsrc_pad / 4, rem = nchan % ct;
    p.ct = ct;
    p.nfull = nchan / ct + (rem > half ? 1 : 0);
    p.tail_ct = (rem == 0 || rem > half) ? 0 : ((ct == 64 && rem > 16) ? 32 : 16);
    p.tail_c0 = (nchan / ct) * ct;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    p.f0_off = take((size_t)(p.nfull + 1) * sizeof(double));
    p.rec_off = take((size_t)p.nfull * nit * stage_doubles(ct, cplx, gauss) * sizeof(double));
    p.tail_rec_off = take((size_t)(p.tail_ct ? nit * stage_doubles(p.tail_ct, cplx, gauss) : 0) * sizeof(double));
    p.total = o;
    return p;
}

template <int CT, bool CPLX>
int run_tiles(const double *image, const double *uvw, const double *frequency, const double *lmn, const int *srcbad,
              const double *tilef, const int *flags, int sign, double *out, int64_t nrow,
              int64_t nsrc, int64_t nit, int64_t nchan, int64_t c0, int64_t ntile, double *f0, double *rec, bool prof,
              hipStream_t st, const AfDftChi2 *chi = nullptr, const double *gauss = nullptr)
{
    const int64_t per = stage_doubles(CT, CPLX, gauss != nullptr);
    for (int64_t t = 0; t < ntile; ++t) {
        int64_t blocks = af_cdiv(nit * per, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(mfma_pack_records, dim3((unsigned)blocks), dim3(256), 0, st, image, (int)CPLX, lmn, srcbad, nsrc,
                           nit, nchan, c0 + t * CT, CT, rec + t * nit * per, gauss);
        AF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(mfma_tile_f0, dim3((unsigned)af_cdiv(ntile, 64)), dim3(64), 0, st, frequency, nchan, c0, CT,
                       (int)ntile, sign, f0);
    AF_LAUNCH_CHECK();
    if (prof) af_prof_begin(st);  // measurement hook: the dominant kernel only
    const dim3 grid((unsigned)af_cdiv(nrow, 64), (unsigned)ntile);
    if constexpr (CPLX) {
        if (gauss != nullptr && chi != nullptr)
            hipLaunchKernelGGL((dft_mfma_kernel<CT, true, true, true>), grid, dim3(THREADS), 0, st, uvw, rec, f0, tilef, flags,
                               lmn, out, nrow, (int)nsrc, (int)nit, nchan, c0, reinterpret_cast<const double2 *>(chi->data),
                               chi->weight, chi->chi2, gauss);
        else if (gauss != nullptr)
            hipLaunchKernelGGL((dft_mfma_kernel<CT, true, false, true>), grid, dim3(THREADS), 0, st, uvw, rec, f0, tilef, flags,
                               lmn, out, nrow, (int)nsrc, (int)nit, nchan, c0, nullptr, nullptr, nullptr, gauss);
    }
    if (gauss != nullptr) {
        // (launched above)
    } else if (chi != nullptr)
        hipLaunchKernelGGL((dft_mfma_kernel<CT, CPLX, true>), grid, dim3(THREADS), 0,
                           st, uvw, rec, f0, tilef, flags, lmn, out, nrow, (int)nsrc, (int)nit, nchan, c0,
                           reinterpret_cast<const double2 *>(chi->data), chi->weight, chi->chi2);
    else
        hipLaunchKernelGGL((dft_mfma_kernel<CT, CPLX>), grid, dim3(THREADS), 0, st,
                           uvw, rec, f0, tilef, flags, lmn, out, nrow, (int)nsrc, (int)nit, nchan, c0);
    if (prof) af_prof_end(st);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

}  // namespace

bool af_dft_mfma_eligible(int64_t nchan, int64_t ncorr, bool image_is_complex)
{
    (void)image_is_complex;  // complex pixels: four MFMAs per channel instead of two
    return ncorr == 4 && nchan >= 14 && nchan / 32 + 1 <= 65535;
}

size_t af_dft_mfma_workspace_bytes(int64_t nsrc_pad, int64_t nchan, bool image_is_complex, bool gauss)
{
    return make_plan(nsrc_pad, nchan, image_is_complex, gauss).total;
}

namespace {
template <bool CPLX>
int run_all(const double *image, const double *uvw, const double *frequency, const double *lmn, const int *srcbad,
            const double *tilef, const int *flags, const int *colstate, int sign, double *out, int64_t nrow, int64_t nsrc,
            int64_t nsrc_pad, int64_t nchan, void *workspace, hipStream_t st, const AfDftChi2 *chi,
            const double *gauss = nullptr)
{
    const Plan p = make_plan(nsrc_pad, nchan, CPLX, gauss != nullptr);
    char *ws = static_cast<char *>(workspace);
    double *f0 = reinterpret_cast<double *>(ws + p.f0_off);
    const int64_t nit = nsrc_pad / 4;
    int rc = AF_OK;
    if (p.nfull > 0 && p.ct == 32)
        rc = run_tiles<32, CPLX>(image, uvw, frequency, lmn, srcbad, tilef, flags, sign, out, nrow, nsrc, nit, nchan,
                                 0, p.nfull, f0, reinterpret_cast<double *>(ws + p.rec_off), true, st, chi, gauss);
    else if (p.nfull > 0)
        rc = run_tiles<64, CPLX>(image, uvw, frequency, lmn, srcbad, tilef, flags, sign, out, nrow, nsrc, nit, nchan,
                                 0, p.nfull, f0, reinterpret_cast<double *>(ws + p.rec_off), true, st, chi, gauss);
    if (rc != AF_OK) return rc;
    double *trec = reinterpret_cast<double *>(ws + p.tail_rec_off);
    if (p.tail_ct == 32)
        rc = run_tiles<32, CPLX>(image, uvw, frequency, lmn, srcbad, tilef, flags, sign, out, nrow, nsrc, nit, nchan,
                                 p.tail_c0, 1, f0 + p.nfull, trec, p.nfull == 0, st, chi, gauss);
    else if (p.tail_ct == 16)
        rc = run_tiles<16, CPLX>(image, uvw, frequency, lmn, srcbad, tilef, flags, sign, out, nrow, nsrc, nit, nchan,
                                 p.tail_c0, 1, f0 + p.nfull, trec, p.nfull == 0, st, chi, gauss);
    if (rc != AF_OK) return rc;
    if (gauss != nullptr) return rc;    // brightness matrices: no zero-pixel / NaN-source column semantics
    // grid-stride sweep; its blocks return at once unless a special column exists
    int64_t fix_blocks = af_cdiv(nrow * nchan * 4, 256);
    if (fix_blocks > 2048) fix_blocks = 2048;
    hipLaunchKernelGGL(mfma_fix_columns, dim3((unsigned)fix_blocks), dim3(256), 0, st, flags, colstate, out, nrow, nchan);
    AF_LAUNCH_CHECK();
    return rc;
}
}  // namespace

int af_dft_mfma_run(const double *image, int image_is_complex, const double *uvw, const double *frequency,
                    const double *lmn, const int *srcbad, const double *tilef, const int *flags, const int *colstate,
                    int sign, double *out, int64_t nrow, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, void *workspace,
                    hipStream_t st, const AfDftChi2 *chi2)
{
    return image_is_complex ? run_all<true>(image, uvw, frequency, lmn, srcbad, tilef, flags, colstate, sign, out, nrow,
                                            nsrc, nsrc_pad, nchan, workspace, st, chi2)
                            : run_all<false>(image, uvw, frequency, lmn, srcbad, tilef, flags, colstate, sign, out, nrow,
                                             nsrc, nsrc_pad, nchan, workspace, st, chi2);
}

// Gaussian (and point) sources without direction-dependent terms on the same kernels (af_gauss_dft.hip): brightness
// (nsrc, nchan, 4) complex128 takes the place of a complex image, gauss (nsrc, 4) = (el, em, er, -) in the kernel's
// frequency units (af_gauss_predict_c128 scales them).  Runs iff flags[0] == flags[1] == 1.
int af_gauss_mfma_run(const double *brightness, const double *gauss, const double *uvw, const double *frequency,
                      const double *lmn, const int *srcbad, const double *tilef, const int *flags, int sign, double *out,
                      int64_t nrow, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, void *workspace, hipStream_t st,
                      const AfDftChi2 *chi2)
{
    return run_all<true>(brightness, uvw, frequency, lmn, srcbad, tilef, flags, nullptr, sign, out, nrow, nsrc, nsrc_pad, nchan,
                         workspace, st, chi2, gauss);
}
