// Per-lane phasor machinery of the MFMA-accumulator direct-transform kernels
// (af_im_to_vis_mfma.hip, af_vis_to_im_mfma.hip): a lane owns one (row, source) pair of a
// 4-deep MFMA step and produces that pair's phasor for every channel of the tile.
#pragma once
#include "af_common.h"
#include "af_sincos.h"

constexpr int MFMA_ANCHOR = 64;  // channels between re-anchored phasors (a power of two)
constexpr int MFMA_ANCHOR_LOG2 = MFMA_ANCHOR == 64 ? 6 : MFMA_ANCHOR == 32 ? 5 : 4;
static_assert((1 << MFMA_ANCHOR_LOG2) == MFMA_ANCHOR, "anchor spacing 16, 32 or 64");

// global -> LDS copy of `units` 16-byte units by a 256-lane block: wave `wave` copies units
// [e0 + 64 wave, +64) of every 256-unit trip; the LDS destination of a wave instruction is
// base + lane * 16.  Issued from asm so that hipcc does not drain it (vmcnt(0)) in front of the LDS
// reads of the stage being computed; the caller retires it with an explicit s_waitcnt vmcnt(0).
template <int UNITS>
__device__ __forceinline__ void mfma_stage_load(const double *src, double *lds_stage, int wave, int lane)
{
#pragma unroll
    for (int e0 = 0; e0 < UNITS; e0 += 256) {
        const int ebase = e0 + wave * 64;  // wave-uniform
        if (ebase + lane < UNITS) {
            const double *g = src + (ebase + lane) * 2;
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_stage + ebase * 2));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
        }
    }
}

// Set-up state of one pair: (a0,a1,a2) is the step-side coordinate triple read from the record
// header ((l,m,n) in im_to_vis, (u,v,w) in vis_to_im); the lane-side triple is passed in.
struct PhasorSetup {
    double a0, a1, a2;
    TablePhasorStage sd, s0;
    double dr, di, y0r, y0i;  // channel-step phasor d, phasor at the tile's first channel
    double ar, ai;            // d^MFMA_ANCHOR
    double kk, k2;            // 2 cos(delta), 2 cos(2 delta)
};

// first 8 phasors after an anchor (y0r, y0i): y1 = y0 d, y2, y3 by the three-term recurrence, then
// four independent chains (re/im x even/odd channels) with step 2 delta
__device__ __forceinline__ void phasor_first_segment(const PhasorSetup &P, double y0r, double y0i, double (&Yr)[8],
                                                     double (&Yi)[8])
{
    Yr[0] = y0r; Yi[0] = y0i;
    Yr[1] = fma(y0r, P.dr, -__dmul_rn(y0i, P.di)); Yi[1] = fma(y0r, P.di, __dmul_rn(y0i, P.dr));
    Yr[2] = fma(P.kk, Yr[1], -Yr[0]); Yi[2] = fma(P.kk, Yi[1], -Yi[0]);
    Yr[3] = fma(P.kk, Yr[2], -Yr[1]); Yi[3] = fma(P.kk, Yi[2], -Yi[1]);
#pragma unroll
    for (int t = 4; t < 8; ++t) { Yr[t] = fma(P.k2, Yr[t - 2], -Yr[t - 4]); Yi[t] = fma(P.k2, Yi[t - 2], -Yi[t - 4]); }
}

// the next 8 phasors from the previous 8
__device__ __forceinline__ void phasor_next_segment(const PhasorSetup &P, double (&Yr)[8], double (&Yi)[8],
                                                    const double (&Pr)[8], const double (&Pi)[8])
{
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const double r2 = t >= 2 ? Yr[t - 2] : Pr[t + 6], r4 = t >= 4 ? Yr[t - 4] : Pr[t + 4];
        const double i2 = t >= 2 ? Yi[t - 2] : Pi[t + 6], i4 = t >= 4 ? Yi[t - 4] : Pi[t + 4];
        Yr[t] = fma(P.k2, r2, -r4); Yi[t] = fma(P.k2, i2, -i4);
    }
}

// The set-up cut into 8 slices, so that the set-up of step it+1 can be spread over the channel groups of
// step it (its dependent chains then hide behind the MFMAs):
//   0 header triple from LDS   1 path difference, range reductions + table reads   2 residual sines   3 residual
//   cosines   4 table entry x residual rotation -> d, y0   5 d^MFMA_ANCHOR by squarings (dead when no tile exceeds the spacing), 2cos(delta), 2cos(2 delta)
//   7 first 8 phasors
// F0, FD: the tile's first-channel frequency and the channel step in 1/256 turns per metre (quarter turns x 64);
// `table`: the block's PHASOR_TABLE-entry phasor table in LDS (table_phasor_init).
__device__ __forceinline__ void phasor_setup_slice(PhasorSetup &P, int slice, const double *hdr4, double c0, double c1,
                                                   double c2, double F0, double FD, const double2 *table,
                                                   double (&Yr)[8], double (&Yi)[8])
{
    switch (slice) {
    case 0: {
        const double2 h = *reinterpret_cast<const double2 *>(hdr4);
        P.a0 = h.x; P.a1 = h.y; P.a2 = hdr4[2];
        break;
    }
    case 1: {
        const double q = fma(P.a2, c2, fma(P.a1, c1, __dmul_rn(P.a0, c0)));  // path difference in metres
        table_phasor_reduce(P.sd, table, __dmul_rn(q, FD));
        table_phasor_reduce(P.s0, table, __dmul_rn(q, F0));
        break;
    }
    case 2: table_phasor_sin(P.sd); table_phasor_sin(P.s0); break;
    case 3: table_phasor_cos(P.sd); table_phasor_cos(P.s0); break;
    case 4: table_phasor_finish(P.sd, P.dr, P.di); table_phasor_finish(P.s0, P.y0r, P.y0i); break;
    case 5: {
        double ar = P.dr, ai = P.di;
#pragma unroll
        for (int t = 0; t < MFMA_ANCHOR_LOG2; ++t) {   // d^MFMA_ANCHOR; dead code when no tile is longer than the spacing
            const double nr = fma(ar, ar, -__dmul_rn(ai, ai)), ni = __dmul_rn(__dadd_rn(ar, ar), ai);
            ar = nr; ai = ni;
        }
        P.ar = ar; P.ai = ai;
        P.kk = __dadd_rn(P.dr, P.dr);
        P.k2 = fma(P.kk, P.kk, -2.0);
        break;
    }
    case 7: phasor_first_segment(P, P.y0r, P.y0i, Yr, Yi); break;
    default: break;
    }
}
