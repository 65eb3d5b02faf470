// Calibration consumers of the predict path for gfx950:
//   corrupt_vis   africanus/calibration/utils/corrupt_vis.py:10-101    vis = sum_dir G_p M_dir G_q^H
//   residual_vis  africanus/calibration/utils/residual_vis.py:11-119   res = vis - sum_dir G_p M_dir G_q^H (unflagged)
//   correct_vis   africanus/calibration/utils/correct_vis.py:10-115    cor = G_p^-1 vis G_q^-H            (unflagged)
// in the three gain layouts of africanus/calibration/utils/utils.py:6-45 (DIAG_DIAG, DIAG, FULL).
// HBM-bound streaming kernels: one lane per (row, chan) cell, directions summed in registers in the
// reference's order with the reference's operation order (bit-identical results); the time bin of a row is
// found by a binary search of the (normalised) bin starts instead of the reference's bin-outer loop.
#include <stdlib.h>

#include "af_common.h"

namespace {

struct C2 {
    double re, im;
};
__device__ __forceinline__ C2 cmul(C2 a, C2 b)
{
    C2 z;
    z.re = __dsub_rn(__dmul_rn(a.re, b.re), __dmul_rn(a.im, b.im));
    z.im = __dadd_rn(__dmul_rn(a.re, b.im), __dmul_rn(a.im, b.re));
    return z;
}
__device__ __forceinline__ C2 cadd(C2 a, C2 b) { return C2{__dadd_rn(a.re, b.re), __dadd_rn(a.im, b.im)}; }
__device__ __forceinline__ C2 csub(C2 a, C2 b) { return C2{__dsub_rn(a.re, b.re), __dsub_rn(a.im, b.im)}; }
__device__ __forceinline__ C2 cconj(C2 a) { return C2{a.re, -a.im}; }
__device__ __forceinline__ C2 cneg(C2 a) { return C2{-a.re, -a.im}; }
// CPython's _Py_c_quot, the algorithm numba lowers complex division to:
//   |b.re| >= |b.im|:  rat = b.im / b.re, den = b.re + b.im rat,  ((a.re + a.im rat) / den, (a.im - a.re rat) / den)
//   otherwise:         rat = b.re / b.im, den = b.re rat + b.im,  ((a.re rat + a.im) / den, (a.im rat - a.re) / den)
// As ONE instruction stream (round 3): which branch a lane takes depends on its data, so a wave ran both -- six
// divisions per quotient, 48 per cell of correct_vis with FULL gains, half of that kernel's instructions.  With
// (x, y) = the larger / smaller part of b and (p, q) = (a.re, a.im) or swapped, both branches are rat = y / x,
// den = x + y rat, re = (p + q rat) / den (fp addition commutes bit for bit), and the imaginary numerators are the two
// orders of one subtraction -- computed both (they differ in the sign of an exact zero) and selected.  Same bits.
__device__ __forceinline__ C2 cdiv(C2 a, C2 b)
{
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const bool first = fabs(b.re) >= fabs(b.im);
    const double x = first ? b.re : b.im, y = first ? b.im : b.re;
    const double p = first ? a.re : a.im, q = first ? a.im : a.re;
    const double rat = __ddiv_rn(y, x), den = __dadd_rn(x, __dmul_rn(y, rat));
    const double re = __ddiv_rn(__dadd_rn(p, __dmul_rn(q, rat)), den);
    // first: a.im - a.re rat = q - p rat;  second: a.im rat - a.re = p rat - q
    const double pr = __dmul_rn(p, rat);
    const double im = __ddiv_rn(first ? __dsub_rn(q, pr) : __dsub_rn(pr, q), den);
    const bool zero = first && b.re == 0.0;
    return C2{zero ? nan : re, zero ? nan : im};
}

constexpr int jones_elems(int mode, int ncorr) { return mode == 0 ? ncorr : (mode == 1 ? 2 : 4); }
constexpr int vis_elems(int mode, int ncorr) { return mode == 0 ? ncorr : 4; }

// min of the bin starts (the reference normalises time_bin_indices in place, corrupt_vis.py:85)
__global__ void calib_min_kernel(const int64_t *__restrict__ tbin_idx, int64_t ntime, int64_t *__restrict__ mn)
{
    __shared__ long long smin;
    if (threadIdx.x == 0) smin = 0x7fffffffffffffffLL;
    __syncthreads();
    long long m = 0x7fffffffffffffffLL;
    for (int64_t i = threadIdx.x; i < ntime; i += blockDim.x) m = tbin_idx[i] < m ? tbin_idx[i] : m;
    atomicMin(&smin, m);
    __syncthreads();
    if (threadIdx.x == 0) *mn = ntime > 0 ? smin : 0;
}

// time bin of every row: the last t with start[t] <= row, if row < start[t] + count[t]; -1 otherwise
__global__ void calib_rowbin_kernel(const int64_t *__restrict__ tbin_idx, const int64_t *__restrict__ tbin_counts,
                                    const int64_t *__restrict__ mn, int64_t ntime, int64_t nrow, int *__restrict__ rowbin)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrow) return;
    const int64_t off = *mn;
    int64_t lo = 0, hi = ntime;  // first t with start[t] > row
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (tbin_idx[mid] - off <= row) lo = mid + 1; else hi = mid;
    }
    const int64_t t = lo - 1;
    rowbin[row] = (t >= 0 && row < tbin_idx[t] - off + tbin_counts[t]) ? (int)t : -1;
}

// acc (+/-)= g . m . h^H for ONE direction, the reference's jones_mul / subtract_model bodies
template <int MODE, int NCORR, int SIGN>
__device__ __forceinline__ void jones_term(const C2 *__restrict__ g, const C2 *__restrict__ m, const C2 *__restrict__ h,
                                           C2 (&acc)[vis_elems(MODE, NCORR)])
{
    if constexpr (MODE == 0) {
#pragma unroll
        for (int c = 0; c < NCORR; ++c) {
            const C2 t = cmul(cmul(g[c], m[c]), cconj(h[c]));
            acc[c] = SIGN > 0 ? cadd(acc[c], t) : csub(acc[c], t);
        }
    } else if constexpr (MODE == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const C2 t = cmul(cmul(g[c >> 1], m[c]), cconj(h[c & 1]));
            acc[c] = SIGN > 0 ? cadd(acc[c], t) : csub(acc[c], t);
        }
    } else {
        const C2 tmp00 = cconj(h[0]), tmp01 = cconj(h[2]), tmp10 = cconj(h[1]), tmp11 = cconj(h[3]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const C2 t1 = cmul(g[2 * i], m[0]), t2 = cmul(g[2 * i + 1], m[2]);
            const C2 t3 = cmul(g[2 * i], m[1]), t4 = cmul(g[2 * i + 1], m[3]);
            const C2 o0 = cadd(cadd(cadd(cmul(t1, tmp00), cmul(t2, tmp00)), cmul(t3, tmp10)), cmul(t4, tmp10));
            const C2 o1 = cadd(cadd(cadd(cmul(t1, tmp01), cmul(t2, tmp01)), cmul(t3, tmp11)), cmul(t4, tmp11));
            acc[2 * i] = SIGN > 0 ? cadd(acc[2 * i], o0) : csub(acc[2 * i], o0);
            acc[2 * i + 1] = SIGN > 0 ? cadd(acc[2 * i + 1], o1) : csub(acc[2 * i + 1], o1);
        }
    }
}

// The per-antenna gain of a lane's (row, chan) cell is a J*16-byte record that the lane's neighbours do not share a
// cache line with (record stride ndir*J*16 bytes along the channel axis, other antennas elsewhere): read lane by lane,
// every 16-byte load instruction touches 64 lines, and the CU's address / tag path -- not HBM -- bounds the kernel
// (tools/microbench_gather.hip: 2.8 TB/s against 4.6 TB/s).  Instead J lanes fetch one record together (a full
// 64 / 32-byte segment per record and instruction) and the wave transposes through a private LDS region: slot
// 64 k + lane on the way in, an XOR-swizzled slot on the way out so that both directions are free of bank conflicts.
// `rec` is the lane's record index (units of J*16 bytes); every lane of the wave must call.
// RJ = units of 16 bytes per `rec` (the record a `rec` index counts): J when a call fetches one direction's record,
// J / ndir when it fetches the ndir adjacent records of a cell in one go (J = ndir x the layout's elements: all
// directions of a (time, antenna, chan) cell are contiguous, so each load instruction then covers whole cache lines --
// with two directions of FULL gains 8 cells x 128 bytes = 1 KB contiguous instead of 16 half-used lines).
template <int J, int RJ = J>
__device__ __forceinline__ void gather_gain(const C2 *__restrict__ jones, int rec, C2 (&g)[J], double2 *lds_wave)
{
    if constexpr (J == 1) {
        g[0] = jones[rec];
    } else {
        const int lane = threadIdx.x & 63;
        constexpr int CPI = 64 / J;   // cells per load instruction
        const double2 *src = reinterpret_cast<const double2 *>(jones);
#pragma unroll
        for (int k = 0; k < J; ++k) {
            const int c = k * CPI + lane / J, h = lane % J;
            const int rec_c = __shfl(rec, c, 64);
            const int hs = J >= 4 ? (h ^ ((c >> 1) & (J - 1))) : h;
            lds_wave[k * 64 + lane] = src[(int64_t)rec_c * RJ + hs];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int slot = J >= 4 ? (j ^ ((lane >> 1) & (J - 1))) : j;
            const double2 v = lds_wave[lane * J + slot];
            g[j] = C2{v.x, v.y};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// The lane-per-cell arrays (visibilities, model, result: V x 16 bytes per cell, cells of a wave contiguous) have the same
// problem in a milder form: read or written lane by lane, a 16-byte instruction touches 32 (V = 4) or 64 (two directions
// of model) half-used lines.  They go through the same transpose: coalesced on the memory side (consecutive lanes =
// consecutive 16 bytes), record-per-lane on the register side.  Loads are gather_gain with rec = the cell number.
template <int V>
__device__ __forceinline__ void coop_store(C2 *__restrict__ out, int64_t wave_cell0, int64_t ncells, const C2 (&acc)[V],
                                           double2 *lds_wave)
{
    const int lane = threadIdx.x & 63;
    if constexpr (V == 1) {
        if (wave_cell0 + lane < ncells) out[wave_cell0 + lane] = acc[0];
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int slot = V >= 4 ? (j ^ ((lane >> 1) & (V - 1))) : j;
            lds_wave[lane * V + slot] = make_double2(acc[j].re, acc[j].im);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int CPI = 64 / V;
        double2 *dst = reinterpret_cast<double2 *>(out);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int c = k * CPI + lane / V, h = lane % V;
            const int hs = V >= 4 ? (h ^ ((c >> 1) & (V - 1))) : h;
            // slot hs of cell c's record holds element hs ^ swizzle(c) = h: consecutive lanes store consecutive 16 bytes
            const double2 v = lds_wave[c * V + hs];
            if (wave_cell0 + c < ncells) dst[(wave_cell0 + c) * V + h] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Inverse gains of correct_vis with FULL (2 x 2) gains, once per (time, antenna, chan) instead of once per baseline
// and cell (each antenna's matrix is inverted for every one of its 63 baselines otherwise -- eight complex divisions,
// half of that kernel's instructions): ainv = G^-1 (correct_vis.py:70-77) and binv = (G^H)^-1 (:79-89), exactly the
// reference's operations on exactly its operands, so the main kernel's products see the same bits.
__global__ __launch_bounds__(256) void calib_inverse_kernel(const C2 *__restrict__ jones, int64_t nrec, C2 *__restrict__ ainv,
                                                            C2 *__restrict__ binv)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    const C2 *g = jones + r * 4;
    const C2 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
    const C2 det1 = csub(cmul(g0, g3), cmul(g1, g2));
    C2 *a = ainv + r * 4, *b = binv + r * 4;
    a[0] = cdiv(g3, det1); a[1] = cdiv(cneg(g1), det1);
    a[2] = cdiv(cneg(g2), det1); a[3] = cdiv(g0, det1);
    const C2 c0 = cconj(g0), c1 = cconj(g1), c2 = cconj(g2), c3 = cconj(g3);
    const C2 det2 = csub(cmul(c0, c3), cmul(c1, c2));
    b[0] = cdiv(c3, det2); b[1] = cdiv(cneg(c2), det2);
    b[2] = cdiv(cneg(c1), det2); b[3] = cdiv(c0, det2);
}

// OP 0 corrupt, 1 residual, 2 correct, 3 compute-and-corrupt (model per time bin, phase computed here);
// grid: ceil(nrow*nchan / 256)
// NDIRT = 2: the call has exactly two directions and fetches both records of a gain in one cooperative gather
// COOP: visibilities / model / result through the cooperative transposes (cells x directions < 2^31);
// ainv / binv: the tables of calib_inverse_kernel (OP 2, FULL gains) or NULL
template <int OP, int MODE, int NCORR, int NDIRT = 0, bool COOP = false>
// (corrupt_vis with cooperative IO sat at 129 registers: one over four waves per SIMD.  Capped there: 4.42 -> 4.54 TB/s;
// the same cap makes residual_vis spill: 4.2 -> 3.4 TB/s)
__global__ __launch_bounds__(256, OP == 0 ? 4 : 1) void calib_kernel(const int *__restrict__ rowbin, const int64_t *__restrict__ ant1,
                                                    const int64_t *__restrict__ ant2, const C2 *__restrict__ jones,
                                                    const C2 *__restrict__ vis, const unsigned char *__restrict__ flag,
                                                    const C2 *__restrict__ model, int64_t nrow, int64_t nant,
                                                    int64_t nchan, int64_t ndir, C2 *__restrict__ out,
                                                    const double *__restrict__ uvw = nullptr,
                                                    const double *__restrict__ freq = nullptr,
                                                    const double *__restrict__ lm = nullptr,
                                                    const C2 *__restrict__ ainv = nullptr,
                                                    const C2 *__restrict__ binv = nullptr)
{
    constexpr int J = jones_elems(MODE, NCORR), V = vis_elems(MODE, NCORR);
    constexpr int JT = NDIRT > 0 ? NDIRT * J : J;       // 16-byte units per gather
    constexpr int LT = COOP ? (NDIRT > 0 ? (NDIRT * V > JT ? NDIRT * V : JT) : (V > JT ? V : JT)) : JT;
    __shared__ double2 lds_gain[4][LT > 1 ? 64 * LT : 1];
    double2 *lds_wave = lds_gain[threadIdx.x >> 6];
    const int64_t cell_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in_range = cell_raw < nrow * nchan;      // out-of-range lanes still take part in the wave's gathers
    const int64_t cell = in_range ? cell_raw : 0;
    const int64_t row = cell / nchan, nu = cell - row * nchan;
    const int t = in_range ? rowbin[row] : -1;
    C2 acc[V];
#pragma unroll
    for (int c = 0; c < V; ++c) acc[c] = C2{0.0, 0.0};
    bool active = t >= 0;
    if ((OP == 1 || OP == 2) && active) {
        if constexpr (V == 4) {
            active = *reinterpret_cast<const unsigned *>(flag + cell * 4) == 0u;      // the cell's four flags in one word
        } else {
#pragma unroll
            for (int c = 0; c < V; ++c) active = active && flag[cell * V + c] == 0;
        }
    }
    const int64_t wave_cell0 = cell_raw - (threadIdx.x & 63);
    // record indices of the two gains of direction 0 (inactive lanes: record 0, fetched and ignored)
    int rec1 = 0, rec2 = 0;
    if (active) {
        const int64_t p = ant1[row], q = ant2[row];
        rec1 = (int)((((int64_t)t * nant + p) * nchan + nu) * ndir);
        rec2 = (int)((((int64_t)t * nant + q) * nchan + nu) * ndir);
    }
    // inactive lanes borrow the record of the wave's first active lane (a valid address whatever the extents are);
    // a wave without any active lane skips the gathers
    const unsigned long long act_mask = __ballot(active);
    const bool any_active = act_mask != 0;   // wave-uniform
    {
        const int first = any_active ? __ffsll((unsigned long long)act_mask) - 1 : 0;
        const int b1 = __shfl(rec1, first, 64), b2 = __shfl(rec2, first, 64);
        if (!active) { rec1 = b1; rec2 = b2; }
    }
    C2 g1[J], g2[J];
    if constexpr (OP == 0 || OP == 1) {
        if constexpr (OP == 1) {
            if constexpr (COOP) {
                C2 vv[V];
                gather_gain<V>(vis, (int)cell, vv, lds_wave);
                if (active) {
#pragma unroll
                    for (int c = 0; c < V; ++c) acc[c] = vv[c];
                }
            } else if (active) {
#pragma unroll
                for (int c = 0; c < V; ++c) acc[c] = vis[cell * V + c];
            }
        }
        if constexpr (NDIRT > 0) {
            C2 ga[JT], gb[JT];
            C2 mm[COOP ? NDIRT * V : 1];
            if constexpr (COOP) gather_gain<NDIRT * V>(model, (int)cell, mm, lds_wave);   // both directions of the cell
            if (any_active) {
                gather_gain<JT, J>(jones, rec1, ga, lds_wave);
                gather_gain<JT, J>(jones, rec2, gb, lds_wave);
            }
#pragma unroll
            for (int s = 0; s < NDIRT; ++s) {
                C2 m[V];
                if (active) {
#pragma unroll
                    for (int c = 0; c < V; ++c) {
                        if constexpr (COOP) m[c] = mm[s * V + c];
                        else m[c] = model[(cell * NDIRT + s) * V + c];
                    }
#pragma unroll
                    for (int j = 0; j < J; ++j) { g1[j] = ga[s * J + j]; g2[j] = gb[s * J + j]; }
                    jones_term<MODE, NCORR, OP == 0 ? +1 : -1>(g1, m, g2, acc);
                }
            }
        } else {
            for (int64_t s = 0; s < ndir; ++s) {
                C2 m[V];
                if constexpr (COOP) {
                    gather_gain<V>(model, (int)(cell * ndir + s), m, lds_wave);
                } else if (active) {
#pragma unroll
                    for (int c = 0; c < V; ++c) m[c] = model[(cell * ndir + s) * V + c];
                }
                if (any_active) {
                    gather_gain<J>(jones, rec1 + (int)s, g1, lds_wave);
                    gather_gain<J>(jones, rec2 + (int)s, g2, lds_wave);
                }
                if (active) jones_term<MODE, NCORR, OP == 0 ? +1 : -1>(g1, m, g2, acc);
            }
        }
    } else if constexpr (OP == 3) {
        // compute_and_corrupt_vis.py:14-22: source_vis = model[t,nu,s] * exp(1j * real_phase) / n, n = sqrt(1 - l^2 - m^2)
        const double u = uvw[3 * row], v = uvw[3 * row + 1], w = uvw[3 * row + 2], nuf = freq[nu];
        const int tt = active ? t : 0;
        const C2 *mt = model + ((int64_t)tt * nchan + nu) * ndir * V;
        for (int64_t s = 0; s < ndir; ++s) {
            if (any_active) {
                gather_gain<J>(jones, rec1 + (int)s, g1, lds_wave);
                gather_gain<J>(jones, rec2 + (int)s, g2, lds_wave);
            }
            if (active) {
                const double l = lm[((int64_t)tt * ndir + s) * 2], m = lm[((int64_t)tt * ndir + s) * 2 + 1];
                const double n = __dsqrt_rn(__dsub_rn(__dsub_rn(1.0, __dmul_rn(l, l)), __dmul_rn(m, m)));
                const double real_phase = __dmul_rn(
                    __dmul_rn(AF_MINUS_TWO_PI_OVER_C, nuf),
                    __dadd_rn(__dadd_rn(__dmul_rn(u, l), __dmul_rn(v, m)), __dmul_rn(w, __dsub_rn(n, 1.0))));
                double sp, cp;
                sincos(real_phase, &sp, &cp);
                const C2 ph{cp, sp}, nn{n, 0.0};
                C2 sv[V];
#pragma unroll
                for (int c = 0; c < V; ++c) sv[c] = cdiv(cmul(mt[s * V + c], ph), nn);
                jones_term<MODE, NCORR, +1>(g1, sv, g2, acc);
            }
        }
    } else {
        const bool tables = MODE == 2 && ainv != nullptr;      // kernel-uniform
        if (any_active) {
            gather_gain<J>(tables ? ainv : jones, rec1, g1, lds_wave);
            gather_gain<J>(tables ? binv : jones, rec2, g2, lds_wave);
        }
        C2 vv[V];
        if constexpr (COOP) gather_gain<V>(vis, (int)cell, vv, lds_wave);
        if (active) {
            const C2 *a1j = g1, *a2j = g2;
            const C2 *b = COOP ? vv : vis + cell * V;
            if constexpr (MODE == 0) {
#pragma unroll
                for (int c = 0; c < NCORR; ++c) acc[c] = cdiv(b[c], cmul(a1j[c], cconj(a2j[c])));
            } else if constexpr (MODE == 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = cdiv(b[c], cmul(a1j[c >> 1], cconj(a2j[c & 1])));
            } else {
                C2 a00, a01, a10, a11, b00, b01, b10, b11;
                if (tables) {
                    a00 = a1j[0]; a01 = a1j[1]; a10 = a1j[2]; a11 = a1j[3];
                    b00 = a2j[0]; b01 = a2j[1]; b10 = a2j[2]; b11 = a2j[3];
                } else {
                    const C2 det1 = csub(cmul(a1j[0], a1j[3]), cmul(a1j[1], a1j[2]));
                    a00 = cdiv(a1j[3], det1); a01 = cdiv(cneg(a1j[1]), det1);
                    a10 = cdiv(cneg(a1j[2]), det1); a11 = cdiv(a1j[0], det1);
                    const C2 c0 = cconj(a2j[0]), c1 = cconj(a2j[1]), c2 = cconj(a2j[2]), c3 = cconj(a2j[3]);
                    const C2 det2 = csub(cmul(c0, c3), cmul(c1, c2));
                    b00 = cdiv(c3, det2); b01 = cdiv(cneg(c2), det2);
                    b10 = cdiv(cneg(c1), det2); b11 = cdiv(c0, det2);
                }
                C2 t1 = cmul(a00, b[0]), t2 = cmul(a01, b[2]), t3 = cmul(a00, b[1]), t4 = cmul(a01, b[3]);
                acc[0] = cadd(cadd(cadd(cmul(t1, b00), cmul(t2, b00)), cmul(t3, b10)), cmul(t4, b10));
                acc[1] = cadd(cadd(cadd(cmul(t1, b01), cmul(t2, b01)), cmul(t3, b11)), cmul(t4, b11));
                t1 = cmul(a10, b[0]); t2 = cmul(a11, b[2]); t3 = cmul(a10, b[1]); t4 = cmul(a11, b[3]);
                acc[2] = cadd(cadd(cadd(cmul(t1, b00), cmul(t2, b00)), cmul(t3, b10)), cmul(t4, b10));
                acc[3] = cadd(cadd(cadd(cmul(t1, b01), cmul(t2, b01)), cmul(t3, b11)), cmul(t4, b11));
            }
        }
    }
    if constexpr (COOP) {
        coop_store<V>(out, wave_cell0, nrow * nchan, acc, lds_wave);
    } else {
        if (!in_range) return;
#pragma unroll
        for (int c = 0; c < V; ++c) out[cell * V + c] = acc[c];
    }
}

template <int OP>
int run(const int64_t *tbin_idx, const int64_t *tbin_counts, int64_t ntime, const int64_t *ant1, const int64_t *ant2,
        const double *jones, const double *vis, const unsigned char *flag, const double *model, int64_t nrow,
        int64_t nant, int64_t nchan, int64_t ndir, int mode, int ncorr, double *out, void *workspace,
        size_t workspace_bytes, void *stream, const char *who, const double *uvw = nullptr, const double *freq = nullptr,
        const double *lm = nullptr)
{
    AF_REQUIRE(mode >= 0 && mode <= 2, "%s: mode must be 0 (DIAG_DIAG), 1 (DIAG) or 2 (FULL)", who);
    AF_REQUIRE(ncorr == 1 || ncorr == 2, "ncorr cant be larger than 2");
    AF_REQUIRE(mode == 0 || ncorr == 2, "%s: DIAG and FULL gains need 2x2 visibilities", who);
    AF_REQUIRE(nrow >= 0 && nant >= 0 && nchan >= 0 && ndir >= 0 && ntime >= 0, "%s: negative extent", who);
    AF_REQUIRE(OP != 2 || ndir <= 1, "Jones has n_dir > 1. Cannot correct for direction dependent gains");
    AF_REQUIRE(OP != 2 || ndir == 1, "%s: jones needs one direction", who);
    AF_REQUIRE((double)ntime * (double)nant * (double)nchan * (double)(ndir > 0 ? ndir : 1) < 2147483648.0,
               "%s: more than 2^31 gain records (time x ant x chan x dir)", who);
    if (nrow == 0 || nchan == 0) return AF_OK;
    AF_REQUIRE(out && ant1 && ant2 && (ntime == 0 || (tbin_idx && tbin_counts)), "%s: NULL array", who);
    AF_REQUIRE(jones || ntime == 0, "%s: NULL jones", who);
    AF_REQUIRE(OP == 0 || OP == 3 || (vis && flag), "%s: NULL vis / flag", who);
    AF_REQUIRE(OP != 3 || (uvw && freq && (lm || ndir == 0)), "%s: NULL uvw / freq / lm", who);
    AF_REQUIRE(OP == 2 || model || ndir == 0, "%s: NULL model", who);
    const size_t need = 256 + (size_t)nrow * sizeof(int);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= need, "%s: workspace too small (%zu < %zu)", who,
               workspace_bytes, need);
    // correct_vis with FULL gains: room for the two tables of inverse gains (af_correct_vis_workspace_bytes)?
    const size_t tab_off = af_align_up(need, 256);
    const size_t nrec = (size_t)ntime * (size_t)nant * (size_t)nchan;
    const bool tables = OP == 2 && mode == 2 && nrec > 0 && workspace_bytes >= tab_off + 2 * nrec * 4 * sizeof(C2) &&
                        !(getenv("AFHIP_CALIB_TABLES") && atoi(getenv("AFHIP_CALIB_TABLES")) == 0);
    hipStream_t st = af_stream(stream);
    int64_t *mn = static_cast<int64_t *>(workspace);
    int *rowbin = reinterpret_cast<int *>(static_cast<char *>(workspace) + 256);
    hipLaunchKernelGGL(calib_min_kernel, dim3(1), dim3(256), 0, st, tbin_idx, ntime, mn);
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL(calib_rowbin_kernel, dim3((unsigned)af_cdiv(nrow, 256)), dim3(256), 0, st, tbin_idx, tbin_counts, mn,
                       ntime, nrow, rowbin);
    AF_LAUNCH_CHECK();
    const int64_t cells = nrow * nchan;
    AF_REQUIRE(af_cdiv(cells, 256) < (1LL << 31), "%s: problem too large for one launch", who);
    const dim3 grid((unsigned)af_cdiv(cells, 256)), block(256);
    const C2 *jn = reinterpret_cast<const C2 *>(jones), *vs = reinterpret_cast<const C2 *>(vis);
    const C2 *md = reinterpret_cast<const C2 *>(model);
    C2 *o = reinterpret_cast<C2 *>(out);
    C2 *ainv = nullptr, *binv = nullptr;
    if (tables) {
        ainv = reinterpret_cast<C2 *>(static_cast<char *>(workspace) + tab_off);
        binv = ainv + nrec * 4;
        hipLaunchKernelGGL(calib_inverse_kernel, dim3((unsigned)af_cdiv((int64_t)nrec, 256)), dim3(256), 0, st, jn,
                           (int64_t)nrec, ainv, binv);
        AF_LAUNCH_CHECK();
    }
    // lane-per-cell arrays through the cooperative transposes: 2 x 2 visibilities, record numbers within 31 bits
    // (AFHIP_CALIB_COOP=0: the direct form, for A/B runs)
    const bool coop = OP != 3 && (mode != 0 || ncorr == 2) && (double)cells * (double)(ndir > 0 ? ndir : 1) < 2147483648.0 &&
                      !(getenv("AFHIP_CALIB_COOP") && atoi(getenv("AFHIP_CALIB_COOP")) == 0);
    // two directions (corrupt / residual): both records of a gain per gather (AFHIP_CALIB_PAIR=0: one per direction)
    const bool pair = (OP == 0 || OP == 1) && ndir == 2 && !(getenv("AFHIP_CALIB_PAIR") && atoi(getenv("AFHIP_CALIB_PAIR")) == 0);
#define AF_CALIB_GO(M, N, D, C)                                                                                        \
    hipLaunchKernelGGL((calib_kernel<OP, M, N, D, C>), grid, block, 0, st, rowbin, ant1, ant2, jn, vs, flag, md, nrow,     \
                       nant, nchan, ndir, o, uvw, freq, lm, ainv, binv)
#define AF_CALIB_LAUNCH(M, N)                                                                                          \
    do {                                                                                                               \
        if constexpr (OP == 0 || OP == 1) {                                                                            \
            if (pair) {                                                                                                \
                if (coop && (M != 0 || N == 2)) AF_CALIB_GO(M, N, 2, true);                                            \
                else AF_CALIB_GO(M, N, 2, false);                                                                      \
                break;                                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
        if constexpr (OP != 3 && (M != 0 || N == 2)) {                                                                 \
            if (coop) { AF_CALIB_GO(M, N, 0, true); break; }                                                           \
        }                                                                                                              \
        AF_CALIB_GO(M, N, 0, false);                                                                                   \
    } while (0)
    if (mode == 0 && ncorr == 1) AF_CALIB_LAUNCH(0, 1);
    else if (mode == 0) AF_CALIB_LAUNCH(0, 2);
    else if (mode == 1) AF_CALIB_LAUNCH(1, 2);
    else AF_CALIB_LAUNCH(2, 2);
#undef AF_CALIB_LAUNCH
#undef AF_CALIB_GO
    AF_LAUNCH_CHECK();
    return AF_OK;
}

}  // namespace

AF_EXPORT size_t af_calibration_workspace_bytes(int64_t nrow) { return nrow < 0 ? 0 : 256 + (size_t)nrow * sizeof(int); }

// correct_vis with FULL gains: the base workspace plus two tables of inverse gains, 2 x (time x ant x chan) x 64 bytes.
// With only af_calibration_workspace_bytes(nrow) the call still works (every cell then inverts its two gains itself).
AF_EXPORT size_t af_correct_vis_workspace_bytes(int64_t nrow, int64_t ntime, int64_t nant, int64_t nchan)
{
    if (nrow < 0 || ntime < 0 || nant < 0 || nchan < 0) return 0;
    return af_align_up(256 + (size_t)nrow * sizeof(int), 256) + 2 * (size_t)ntime * (size_t)nant * (size_t)nchan * 64;
}

AF_EXPORT int af_corrupt_vis_c128(const int64_t *time_bin_indices, const int64_t *time_bin_counts, int64_t ntime,
                                  const int64_t *antenna1, const int64_t *antenna2, const double *jones,
                                  const double *model, int64_t nrow, int64_t nant, int64_t nchan, int64_t ndir, int mode,
                                  int ncorr, double *out, void *workspace, size_t workspace_bytes, void *stream)
{
    return run<0>(time_bin_indices, time_bin_counts, ntime, antenna1, antenna2, jones, nullptr, nullptr, model, nrow, nant,
                  nchan, ndir, mode, ncorr, out, workspace, workspace_bytes, stream, "af_corrupt_vis_c128");
}

AF_EXPORT int af_residual_vis_c128(const int64_t *time_bin_indices, const int64_t *time_bin_counts, int64_t ntime,
                                   const int64_t *antenna1, const int64_t *antenna2, const double *jones,
                                   const double *vis, const unsigned char *flag, const double *model, int64_t nrow,
                                   int64_t nant, int64_t nchan, int64_t ndir, int mode, int ncorr, double *out,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    return run<1>(time_bin_indices, time_bin_counts, ntime, antenna1, antenna2, jones, vis, flag, model, nrow, nant, nchan,
                  ndir, mode, ncorr, out, workspace, workspace_bytes, stream, "af_residual_vis_c128");
}

AF_EXPORT int af_correct_vis_c128(const int64_t *time_bin_indices, const int64_t *time_bin_counts, int64_t ntime,
                                  const int64_t *antenna1, const int64_t *antenna2, const double *jones,
                                  const double *vis, const unsigned char *flag, int64_t nrow, int64_t nant, int64_t nchan,
                                  int64_t ndir, int mode, int ncorr, double *out, void *workspace, size_t workspace_bytes,
                                  void *stream)
{
    return run<2>(time_bin_indices, time_bin_counts, ntime, antenna1, antenna2, jones, vis, flag, nullptr, nrow, nant, nchan,
                  ndir, mode, ncorr, out, workspace, workspace_bytes, stream, "af_correct_vis_c128");
}

AF_EXPORT int af_compute_and_corrupt_vis_c128(const int64_t *time_bin_indices, const int64_t *time_bin_counts,
                                              int64_t ntime, const int64_t *antenna1, const int64_t *antenna2,
                                              const double *jones, const double *model, const double *uvw,
                                              const double *frequency, const double *lm, int64_t nrow, int64_t nant,
                                              int64_t nchan, int64_t ndir, int mode, int ncorr, double *out,
                                              void *workspace, size_t workspace_bytes, void *stream)
{
    return run<3>(time_bin_indices, time_bin_counts, ntime, antenna1, antenna2, jones, nullptr, nullptr, model, nrow, nant,
                  nchan, ndir, mode, ncorr, out, workspace, workspace_bytes, stream, "af_compute_and_corrupt_vis_c128", uvw,
                  frequency, lm);
}
