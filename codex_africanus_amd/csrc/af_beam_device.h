// Device-side beam cube sampling shared by beam_cube_dde (af_beam_cube.hip) and the fused
// predict (af_fused_predict.hip).  Restates africanus/rime/fast_beam_cubes.py:110-238 with the
// reference's operation order and explicitly rounded operations (no contraction).
#pragma once
#include "af_common.h"

template <typename T> struct BeamOps;
template <> struct BeamOps<double> {
    static __device__ __forceinline__ double mul(double a, double b) { return __dmul_rn(a, b); }
    static __device__ __forceinline__ double add(double a, double b) { return __dadd_rn(a, b); }
    static __device__ __forceinline__ double sub(double a, double b) { return __dsub_rn(a, b); }
    static __device__ __forceinline__ double div(double a, double b) { return __ddiv_rn(a, b); }
    static __device__ __forceinline__ double floor_(double a) { return floor(a); }
    static __device__ __forceinline__ double hypot_(double a, double b) { return hypot(a, b); }
    static __device__ __forceinline__ void sincos_(double p, double *s, double *c) { sincos(p, s, c); }
    typedef double2 vec2;
};
template <> struct BeamOps<float> {
    static __device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
    static __device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
    static __device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }
    static __device__ __forceinline__ float div(float a, float b) { return __fdiv_rn(a, b); }
    static __device__ __forceinline__ float floor_(float a) { return floorf(a); }
    static __device__ __forceinline__ float hypot_(float a, float b) { return hypotf(a, b); }
    static __device__ __forceinline__ void sincos_(float p, float *s, float *c) { sincosf(p, s, c); }
    typedef float2 vec2;
};

// Geometry of one beam sample: the 8 voxel offsets (in complex elements, before the
// correlation index) and their trilinear weights, in the reference's order (:170-225).
template <typename T, typename I = int64_t> struct BeamVoxels {
    I off[8];
    T wt[8];
};

// Cube-wide constants derived from the extents (:77-93).
template <typename T> struct BeamGrid {
    T lower_l, lower_m, lscale, mscale, lmaxf, mmaxf;
    int64_t lmaxi, mmaxi, beam_mh, beam_nud;
};

template <typename T>
__device__ __forceinline__ BeamGrid<T> beam_grid(const T *__restrict__ lm_ext, int64_t beam_lw, int64_t beam_mh,
                                                 int64_t beam_nud)
{
    using O = BeamOps<T>;
    BeamGrid<T> g;
    const T upper_l = lm_ext[1], upper_m = lm_ext[3];
    g.lower_l = lm_ext[0];
    g.lower_m = lm_ext[2];
    g.lmaxf = (T)(beam_lw - 1);
    g.mmaxf = (T)(beam_mh - 1);
    g.lmaxi = beam_lw - 1;
    g.mmaxi = beam_mh - 1;
    g.lscale = O::div(g.lmaxf, O::sub(upper_l, g.lower_l));
    g.mscale = O::div(g.mmaxf, O::sub(upper_m, g.lower_m));
    g.beam_mh = beam_mh;
    g.beam_nud = beam_nud;
    return g;
}

// (l, m) of the source, (sin, cos) of the parallactic angle, pointing error, antenna scaling and
// the channel's freq_data triple -> voxel offsets and weights (:117-163).
// J = the integer type of the offset arithmetic (int when the whole cube indexes within 31 bits: a 64-bit multiply is
// four to six instructions on this machine).
template <typename T, typename I, typename J = int64_t>
__device__ __forceinline__ void beam_voxels(const BeamGrid<T> &g, T l, T m, T sin_pa, T cos_pa, T pe_l, T pe_m,
                                            T as_l, T as_m, T freq_scale, T nud, int gc0, int ncorr,
                                            BeamVoxels<T, I> &vx)
{
    using O = BeamOps<T>;
    const T one = (T)1.0, zero = (T)0.0;
    const T inv_nud = O::sub(one, nud);
    const int gc1 = gc0 + 1;
    const T sl = O::mul(l, freq_scale), sm = O::mul(m, freq_scale);
    const T tl = O::add(sl, pe_l), tm = O::add(sm, pe_m);
    T vl = O::sub(O::mul(tl, cos_pa), O::mul(tm, sin_pa));
    T vm = O::add(O::mul(tl, sin_pa), O::mul(tm, cos_pa));
    vl = O::mul(vl, as_l);
    vm = O::mul(vm, as_m);
    vl = O::mul(g.lscale, O::sub(vl, g.lower_l));
    vm = O::mul(g.mscale, O::sub(vm, g.lower_m));
    {   // max(zero, min(v, maxf)) with Python's comparison semantics (:150-151)
        T t1 = vl < g.lmaxf ? vl : g.lmaxf; vl = zero > t1 ? zero : t1;
        T t2 = vm < g.mmaxf ? vm : g.mmaxf; vm = zero > t2 ? zero : t2;
    }
    const int gl0 = (int)O::floor_(vl), gm0 = (int)O::floor_(vm);
    const J gl1 = (gl0 + 1 < (J)g.lmaxi) ? (J)(gl0 + 1) : (J)g.lmaxi;
    const J gm1 = (gm0 + 1 < (J)g.mmaxi) ? (J)(gm0 + 1) : (J)g.mmaxi;
    const T ld = O::sub(vl, (T)gl0), md = O::sub(vm, (T)gm0);
    const T omld = O::sub(one, ld), ommd = O::sub(one, md);
    const J GL[8] = {(J)gl0, gl1, (J)gl0, gl1, (J)gl0, gl1, (J)gl0, gl1};
    const J GM[8] = {(J)gm0, (J)gm0, gm1, gm1, (J)gm0, (J)gm0, gm1, gm1};
    const J GC[8] = {(J)gc0, (J)gc0, (J)gc0, (J)gc0, (J)gc1, (J)gc1, (J)gc1, (J)gc1};
    vx.wt[0] = O::mul(O::mul(omld, ommd), nud);
    vx.wt[1] = O::mul(O::mul(ld, ommd), nud);
    vx.wt[2] = O::mul(O::mul(omld, md), nud);
    vx.wt[3] = O::mul(O::mul(ld, md), nud);
    vx.wt[4] = O::mul(O::mul(omld, ommd), inv_nud);
    vx.wt[5] = O::mul(O::mul(ld, ommd), inv_nud);
    vx.wt[6] = O::mul(O::mul(omld, md), inv_nud);
    vx.wt[7] = O::mul(O::mul(ld, md), inv_nud);
#pragma unroll
    for (int k = 0; k < 8; ++k) vx.off[k] = (I)(((GL[k] * (J)g.beam_mh + GM[k]) * (J)g.beam_nud + GC[k]) * (J)ncorr);
}

// One correlation of the sample: weighted sums of the complex voxels and of their amplitudes,
// then the amplitude-preserving normalisation (:170-235).  HAVE_ABS: `babs` is a cube of
// precomputed |beam| values with the same indexing (hypot is the dominant cost otherwise).
// All 8 voxel gathers are issued before the first is consumed (one memory latency, not eight).
template <typename T, typename I, bool HAVE_ABS>
__device__ __forceinline__ typename BeamOps<T>::vec2 beam_sample_corr(
    const typename BeamOps<T>::vec2 *__restrict__ fbeam, const T *__restrict__ babs, const BeamVoxels<T, I> &vx,
    int c)
{
    using O = BeamOps<T>;
    using V2 = typename O::vec2;
    const T zero = (T)0.0;
    V2 b[8];
    T ab[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        b[k] = fbeam[vx.off[k] + c];
        if constexpr (HAVE_ABS) ab[k] = babs[vx.off[k] + c];
    }
    T cre = zero, cim = zero, absc = zero;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const T wgt = vx.wt[k];
        if constexpr (!HAVE_ABS) ab[k] = O::hypot_(b[k].x, b[k].y);
        absc = O::add(absc, O::mul(wgt, ab[k]));
        // (wgt + 0j) * b as a full complex multiply (numba widens the real weight)
        const T pre = O::sub(O::mul(wgt, b[k].x), O::mul(zero, b[k].y));
        const T pim = O::add(O::mul(wgt, b[k].y), O::mul(zero, b[k].x));
        cre = O::add(cre, pre);
        cim = O::add(cim, pim);
    }
    const T div = O::hypot_(cre, cim);
    const T sc = (div == zero) ? absc : O::div(absc, div);
    V2 r;  // corr_sum * (sc + 0j)
    r.x = O::sub(O::mul(cre, sc), O::mul(cim, zero));
    r.y = O::add(O::mul(cre, zero), O::mul(cim, sc));
    return r;
}
