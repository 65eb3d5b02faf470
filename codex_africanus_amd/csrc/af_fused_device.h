// Device code shared by the two fused-predict kernels: af_fused_predict.hip (lane = rows: any uvw) and
// af_fused_gemm.hip (antenna-decomposable uvw: one complex GEMM per (timestep, channel) on the matrix cores).
// Workspace layout, the per-source / per-channel preparation kernels, the per-channel beam planes and their
// bilinear sampler (africanus/rime/fast_beam_cubes.py:110-238, frequency first), DPP broadcasts, complex helpers
// and the table phasor.  Everything sits in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <stdlib.h>

#include "af_common.h"
#include "af_beam_device.h"
#include "af_sincos.h"

namespace {

struct FusedWs {
    size_t lmn, f4, freq_data, gauss, planes, total;
};

// channels whose pre-interpolated beam planes are resident at a time (one launch of the main kernel per group)
constexpr int64_t PLANE_GROUP = 64;

FusedWs fused_ws(int64_t nsrc, int64_t nchan, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud)
{
    FusedWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    w.lmn = take((size_t)nsrc * 4 * sizeof(double));
    w.f4 = take((size_t)nchan * sizeof(double));
    w.freq_data = take((size_t)nchan * 3 * sizeof(double));
    w.gauss = take((size_t)nsrc * 4 * sizeof(double));  // (el*gs, em*gs, er, is_extended) per source
    (void)beam_nud;
    const int64_t group = nchan < PLANE_GROUP ? nchan : PLANE_GROUP;
    w.planes = take((size_t)group * beam_lw * beam_mh * 16 * sizeof(double));  // 128-B cell records per channel
    w.total = o;
    return w;
}

// n = sqrt(max(0, 1 - l^2 - m^2)) - 1: phase_delay's clamped form (africanus/rime/phase.py:42-43)
__global__ void fused_prep_src(const double *__restrict__ lm, int64_t nsrc, double *__restrict__ lmn)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    double l = lm[2 * s], m = lm[2 * s + 1];
    double n = __dsub_rn(__dsub_rn(1.0, __dmul_rn(l, l)), __dmul_rn(m, m));
    n = __dsub_rn(__dsqrt_rn(n < 0.0 ? 0.0 : n), 1.0);
    lmn[4 * s + 0] = l;
    lmn[4 * s + 1] = m;
    lmn[4 * s + 2] = n;
    lmn[4 * s + 3] = 0.0;
}

// quarter turns per metre of path difference for every channel: 4*sign*nu/c
__global__ void fused_prep_freq(const double *__restrict__ freq, int64_t nchan, int sign, double *__restrict__ f4)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nchan) f4[c] = 4.0 * (double)sign * freq[c] / AF_LIGHTSPEED;
}

// Gaussian source shapes (africanus/model/shape/gaussian_shape.py:45-50): el = emaj sin(pa), em = emaj cos(pa),
// er = emin / (emaj or 1), with the frequency scale gs folded into el and em; a source with emaj == emin == 0
// (or no shape array at all) is a point source: its shape factor is exactly 1 and is skipped.
__global__ void fused_prep_gauss(const double *__restrict__ shape_params, int64_t nsrc, double gs,
                                 double *__restrict__ gp)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    double el = 0.0, em = 0.0, er = 0.0, ext = 0.0;
    if (shape_params != nullptr) {
        const double emaj = shape_params[3 * s], emin = shape_params[3 * s + 1], angle = shape_params[3 * s + 2];
        el = emaj * sin(angle) * gs;
        em = emaj * cos(angle) * gs;
        er = emin / (emaj == 0.0 ? 1.0 : emaj);
        ext = (emaj != 0.0 || emin != 0.0) ? 1.0 : 0.0;
    }
    gp[4 * s + 0] = el; gp[4 * s + 1] = em; gp[4 * s + 2] = er; gp[4 * s + 3] = ext;
}

// Per-channel beam planes.  The trilinear sample of the reference (rime/fast_beam_cubes.py:170-225) is a sum over
// 8 voxels with weights w_lm * nud (lower frequency plane) and w_lm * (1 - nud) (upper plane), where the plane pair
// and nud depend on the CHANNEL only (freq_grid_interp, :10-54).  Interpolating along frequency first, once per
// (cell, channel), leaves a bilinear sample of 4 cells per Jones term: half the gathers, half the L2 -> L1 line
// traffic and half the weights of the 8-voxel form, for one extra pass over 2 planes per channel (microseconds).
// One 128-byte record per (channel, l, m): for each of the 4 correlations (re, im, |.|, 0) with
//   re, im = nud * v_lower + (1 - nud) * v_upper,   |.| = nud * |v_lower| + (1 - nud) * |v_upper|
// (the reference sums |v| of every voxel, :170-225: np.abs is taken before the interpolation, as here).
constexpr int VREC = 16;  // doubles per cell record
__global__ void beam_plane_kernel(const double2 *__restrict__ beam, int64_t ncell, int64_t beam_nud,
                                  const double *__restrict__ freq_data, int64_t f0, double *__restrict__ planes)
{
    const int64_t f = f0 + blockIdx.y;
    const double nud = freq_data[3 * f + 1], inv = __dsub_rn(1.0, nud);
    const int64_t gc0 = (int64_t)freq_data[3 * f + 2];
    double *__restrict__ rec = planes + (int64_t)blockIdx.y * ncell * VREC;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (cell, corr)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < ncell * 4; i += stride) {
        const int64_t cell = i >> 2, corr = i & 3;
        const double2 lo = beam[(cell * beam_nud + gc0) * 4 + corr], hi = beam[(cell * beam_nud + gc0 + 1) * 4 + corr];
        double4 r;
        r.x = fma(nud, lo.x, __dmul_rn(inv, hi.x));
        r.y = fma(nud, lo.y, __dmul_rn(inv, hi.y));
        r.z = fma(nud, hypot(lo.x, lo.y), __dmul_rn(inv, hypot(hi.x, hi.y)));
        r.w = 0.0;
        *reinterpret_cast<double4 *>(rec + i * 4) = r;
    }
}

// One correlation of beam_sample_corr (af_beam_device.h) from the channel's plane records: bilinear sums over
// the 4 cells (FMA-contracted and frequency-first: the fused path is checked to 1e-9, not bit
// for bit), then the amplitude-preserving normalisation corr_sum * absc_sum / |corr_sum|.
__device__ __forceinline__ double2 beam_reduce1(const double2 (&v)[4], const double (&ab)[4], const double (&wt)[4])
{
    double cre = 0.0, cim = 0.0, absc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cre = fma(wt[k], v[k].x, cre);
        cim = fma(wt[k], v[k].y, cim);
        absc = fma(wt[k], ab[k], absc);
    }
    // corr_sum * absc_sum / |corr_sum|  (:227-235; absc_sum itself when corr_sum == 0): 1/|.| by v_rsq_f64 (a ~2^-26
    // seed) and ONE Newton step (error 1.5 e^2 ~ 3e-16; the fused chain is checked to 1e-9 against the oracle, not bit
    // for bit.  Until round 5 two steps: four dependent fp64 operations per sampled Jones entry for nothing)
    const double n2 = fma(cre, cre, __dmul_rn(cim, cim));
    double y = __builtin_amdgcn_rsq(n2);
    y = fma(__dmul_rn(0.5, y), fma(-__dmul_rn(n2, y), y, 1.0), y);
    const double sc = (n2 == 0.0) ? absc : __dmul_rn(absc, y);
    return make_double2(__dmul_rn(cre, sc), __dmul_rn(cim, sc));
}

// Geometry of one beam sample on a channel plane: the (l, m) part of beam_voxels (af_beam_device.h; reference
// rime/fast_beam_cubes.py:130-169, same operations in the same order) with 32-bit BYTE offsets of the four corner
// cells into the plane's records and the four bilinear weights.
struct FusedGrid {
    double lower_l, lower_m, lscale, mscale, lmaxf, mmaxf;
    int lmaxi, mmaxi;
    unsigned stride_l, stride_m;  // bytes between consecutive l / m voxels of the packed cube
};
// a wave-uniform double computed on the vector unit (the grid scales come out of fp64 divisions) moved into scalar
// registers: the sampling waves live on a 168-register budget and every value that is the same in all lanes but sits in a
// vector register is two of them (round 5: the two scales were what the GEMM kernel's sampler spilled -- and reloaded
// behind an s_waitcnt vmcnt(0) that drained its gathers)
__device__ __forceinline__ double wave_uniform(double x)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
struct FusedVoxels {
    unsigned off[4];
    double wt[4];
};
// the same in 9 registers instead of 12 (the GEMM kernel's sampler): corner offsets as base + dl + dm, weights as the two
// fractional coordinates (wt = {(1-ld)(1-md), ld (1-md), (1-ld) md, ld md} is rebuilt where a round is consumed)
struct FusedVoxelsC {
    unsigned base, dl, dm;
    double ld, md;
};
__device__ __forceinline__ void fused_voxels(const FusedGrid &g, double l, double m, double sin_pa, double cos_pa,
                                             double pe_l, double pe_m, double as_l, double as_m, double freq_scale,
                                             FusedVoxels &vx)
{
    const double sl = __dmul_rn(l, freq_scale), sm = __dmul_rn(m, freq_scale);
    const double tl = __dadd_rn(sl, pe_l), tm = __dadd_rn(sm, pe_m);
    double vl = __dsub_rn(__dmul_rn(tl, cos_pa), __dmul_rn(tm, sin_pa));
    double vm = __dadd_rn(__dmul_rn(tl, sin_pa), __dmul_rn(tm, cos_pa));
    vl = __dmul_rn(vl, as_l);
    vm = __dmul_rn(vm, as_m);
    vl = __dmul_rn(g.lscale, __dsub_rn(vl, g.lower_l));
    vm = __dmul_rn(g.mscale, __dsub_rn(vm, g.lower_m));
    {   // max(zero, min(v, maxf)) with Python's comparison semantics (:150-151)
        const double t1 = vl < g.lmaxf ? vl : g.lmaxf; vl = 0.0 > t1 ? 0.0 : t1;
        const double t2 = vm < g.mmaxf ? vm : g.mmaxf; vm = 0.0 > t2 ? 0.0 : t2;
    }
    const double fl = floor(vl), fm = floor(vm);
    const int gl0 = (int)fl, gm0 = (int)fm;
    const double ld = __dsub_rn(vl, fl), md = __dsub_rn(vm, fm);
    const double omld = __dsub_rn(1.0, ld), ommd = __dsub_rn(1.0, md);
    vx.wt[0] = __dmul_rn(omld, ommd); vx.wt[1] = __dmul_rn(ld, ommd);
    vx.wt[2] = __dmul_rn(omld, md); vx.wt[3] = __dmul_rn(ld, md);
    const unsigned base = (unsigned)gl0 * g.stride_l + (unsigned)gm0 * g.stride_m;
    const unsigned dl = gl0 < g.lmaxi ? g.stride_l : 0u, dm = gm0 < g.mmaxi ? g.stride_m : 0u;  // upper neighbour clamped
    vx.off[0] = base; vx.off[1] = base + dl; vx.off[2] = base + dm; vx.off[3] = base + dl + dm;
}

__device__ __forceinline__ void fused_voxels_compact(const FusedGrid &g, double l, double m, double sin_pa, double cos_pa,
                                                     double pe_l, double pe_m, double as_l, double as_m, double freq_scale,
                                                     FusedVoxelsC &vx)
{
    const double sl = __dmul_rn(l, freq_scale), sm = __dmul_rn(m, freq_scale);
    const double tl = __dadd_rn(sl, pe_l), tm = __dadd_rn(sm, pe_m);
    double vl = __dsub_rn(__dmul_rn(tl, cos_pa), __dmul_rn(tm, sin_pa));
    double vm = __dadd_rn(__dmul_rn(tl, sin_pa), __dmul_rn(tm, cos_pa));
    vl = __dmul_rn(vl, as_l);
    vm = __dmul_rn(vm, as_m);
    vl = __dmul_rn(g.lscale, __dsub_rn(vl, g.lower_l));
    vm = __dmul_rn(g.mscale, __dsub_rn(vm, g.lower_m));
    {
        const double t1 = vl < g.lmaxf ? vl : g.lmaxf; vl = 0.0 > t1 ? 0.0 : t1;
        const double t2 = vm < g.mmaxf ? vm : g.mmaxf; vm = 0.0 > t2 ? 0.0 : t2;
    }
    const double fl = floor(vl), fm = floor(vm);
    const int gl0 = (int)fl, gm0 = (int)fm;
    vx.ld = __dsub_rn(vl, fl);
    vx.md = __dsub_rn(vm, fm);
    vx.base = (unsigned)gl0 * g.stride_l + (unsigned)gm0 * g.stride_m;
    vx.dl = gl0 < g.lmaxi ? g.stride_l : 0u;
    vx.dm = gm0 < g.mmaxi ? g.stride_m : 0u;
}

// The same with the per-antenna part of the coordinate map folded into six coefficients per slot (the GEMM kernel's set-up
// computes them once per (timestep, channel, antenna)):
//     vl = (((l fs + pe_l) cos - (m fs + pe_m) sin) as_l - lower_l) lscale = l c1l + m c2l + c0l     (vm likewise)
// -- four FMAs where the reference-ordered form above spends sixteen operations per sampled term; the coordinates differ
// from it by ~1e-16 of the cube's extent (a voxel boundary crossed that late moves a weight of ~1e-14: the bilinear
// sample is continuous), far inside the 1e-9 the fused chain is checked to.
__device__ __forceinline__ void fused_voxels_folded(const FusedGrid &g, double l, double m, double c1l, double c2l, double c0l,
                                                    double c1m, double c2m, double c0m, FusedVoxelsC &vx)
{
    double vl = fma(l, c1l, fma(m, c2l, c0l));
    double vm = fma(l, c1m, fma(m, c2m, c0m));
    {
        const double t1 = vl < g.lmaxf ? vl : g.lmaxf; vl = 0.0 > t1 ? 0.0 : t1;
        const double t2 = vm < g.mmaxf ? vm : g.mmaxf; vm = 0.0 > t2 ? 0.0 : t2;
    }
    const double fl = floor(vl), fm = floor(vm);
    const int gl0 = (int)fl, gm0 = (int)fm;
    vx.ld = __dsub_rn(vl, fl);
    vx.md = __dsub_rn(vm, fm);
    vx.base = (unsigned)gl0 * g.stride_l + (unsigned)gm0 * g.stride_m;
    vx.dl = gl0 < g.lmaxi ? g.stride_l : 0u;
    vx.dm = gm0 < g.mmaxi ? g.stride_m : 0u;
}

// value of quad lane QL in all four lanes of the quad; the neighbour lane ^ 1 (DPP quad_perm, no LDS crossbar)
template <int QL> __device__ __forceinline__ int quad_bcast(int x)
{
    return __builtin_amdgcn_mov_dpp(x, QL * 0x55, 0xf, 0xf, true);
}
template <int QL> __device__ __forceinline__ double quad_bcast(double x)
{
    return __hiloint2double(quad_bcast<QL>(__double2hiint(x)), quad_bcast<QL>(__double2loint(x)));
}
// the value of the even (ODD = 0) / odd (ODD = 1) lane of this lane's PAIR: quad_perm [0,0,2,2] / [1,1,3,3]
template <int ODD> __device__ __forceinline__ double pair_bcast(double x)
{
    constexpr int PERM = ODD ? 0xF5 : 0xA0;
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), PERM, 0xf, 0xf, true),
                            __builtin_amdgcn_mov_dpp(__double2loint(x), PERM, 0xf, 0xf, true));
}

struct C2 {
    double re, im;
};
__device__ __forceinline__ C2 cmul(C2 a, C2 b)
{
    C2 z;
    z.re = fma(a.re, b.re, -__dmul_rn(a.im, b.im));
    z.im = fma(a.re, b.im, __dmul_rn(a.im, b.re));
    return z;
}
// a * conj(b)
__device__ __forceinline__ C2 cmulc(C2 a, C2 b)
{
    C2 z;
    z.re = fma(a.re, b.re, __dmul_rn(a.im, b.im));
    z.im = fma(a.im, b.re, -__dmul_rn(a.re, b.im));
    return z;
}
// acc += a * conj(b)
__device__ __forceinline__ void cmacc(C2 &acc, C2 a, C2 b)
{
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(a.im, b.im, acc.re);
    acc.im = fma(a.im, b.re, acc.im);
    acc.im = fma(-a.re, b.im, acc.im);
}
// acc += a * b
__device__ __forceinline__ void cmac(C2 &acc, C2 a, C2 b)
{
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(-a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(a.im, b.re, acc.im);
}

// (exp_neg: af_sincos.h)

// exp(2 pi i x / PH_TABLE) for x = q * f4 * PH_TABLE / 4 (f4 in quarter turns per metre): the table phasor of
// af_sincos.h with a four times finer table (16 KB of LDS): exp(2 pi i k / 1024) from the table times a residual
// rotation |theta| <= pi / 1024 = 3.1e-3 by sin = theta - theta^3 / 6 (next term 2.3e-15) and cos = 1 - theta^2 / 2 +
// theta^4 / 24 (next term 1.2e-18): 12 fp64 operations per phasor instead of 14 (stage 2 is fp64-issue bound, every
// operation per (row, source) counts), errors far below the rounding of the phase argument itself (~1e-11).
constexpr int PH_TABLE = 1024;
__device__ __forceinline__ void fine_table_init(double2 *table, int tid, int nthreads)
{
    for (int i = tid; i < PH_TABLE; i += nthreads) {
        double c, sn;
        sincos_quarter_turns<7>((double)i * (4.0 / PH_TABLE), c, sn);
        table[i] = make_double2(c, sn);
    }
}
__device__ __forceinline__ C2 table_phasor(const double2 *__restrict__ table, double x)
{
    constexpr double T = 6.283185307179586476925 / PH_TABLE;
    constexpr double S1 = T, S3 = -T * T * T / 6.0;
    constexpr double C2c = -T * T / 2.0, C4 = T * T * T * T / 24.0;
    const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
    const double a = __dadd_rn(x, MAGIC);
    const double2 tk = table[__double2loint(a) & (PH_TABLE - 1)];
    const double z = __dsub_rn(x, __dsub_rn(a, MAGIC));  // [-0.5, 0.5]
    const double z2 = __dmul_rn(z, z);
    const double sn = __dmul_rn(z, fma(z2, S3, S1));
    const double cs = fma(z2, fma(z2, C4, C2c), 1.0);
    C2 y;
    y.re = fma(tk.x, cs, -__dmul_rn(tk.y, sn));
    y.im = fma(tk.y, cs, __dmul_rn(tk.x, sn));
    return y;
}

}  // namespace
