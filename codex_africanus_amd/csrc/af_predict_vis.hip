// predict_vis: Jones-chain reduction V = G_p (B + sum_s E_ps X_pqs E_qs^H) G_q^H on
// materialised inputs (the reference's API-compatible form).
//
// Replaces africanus/rime/predict.py:574-617 and the factories it calls:
//   sum_coherencies_factory :193-252, jones_mul_factory :56-190, add_coh_factory :329-339,
//   apply_dies_factory :342-373, tmin normalisation :597.
//
// HBM-bound streaming reduction: one lane owns one (row, chan) cell with all of its
// correlations (a full 2x2 = 64 B of complex128), walks the sources in ascending order
// and keeps the running sum in registers, so `out` is written exactly once (the numba
// loop is source-outermost and re-streams `out` per source).  source_coh is read once,
// coalesced along the channel axis; the per-antenna DDE terms are gathers that hit L2
// (each (s,t,a,chan) Jones is shared by every baseline of the antenna).
// All complex arithmetic is spelled with explicitly rounded multiplies/adds in the
// reference's operation order, so results are bit-identical to the numba path.
//
// Two kernels:
//   * predict_vis_tile_kernel (DDE terms present, round 3): a workgroup owns RB rows x CT channels and walks the
//     sources; per source the DDE terms of EVERY antenna of the block's timestep(s) for its CT channels are copied
//     global -> LDS once (global_load_lds_dwordx4, coalesced, two sources ahead, three stages, one barrier per source)
//     and every baseline of the block takes its two Jones from LDS.  The per-lane Jones gathers -- two thirds of the
//     kernel's L2 -> L1 line requests, each 64-byte record its own line request -- are gone; source_coh still streams
//     from HBM straight into registers, two sources ahead.  Blocks are numbered so that the chan tiles of a row block
//     and the row blocks of a timestep share an XCD (its L2 then fetches a (source, timestep) Jones slab once).
//   * predict_vis_kernel (everything else: no DDEs, odd layouts, antennas that do not fit LDS): lane per cell,
//     per-lane gathers.
// Index guard: a row whose time index (after the tmin shift) is not in [0, ntime) or whose
// antennas are not in [0, nant) reads clamped indices, produces NaN in all its cells and sets a bit of the status
// word at workspace + 8 (1: time index, 2: antenna); the host wrapper raises ValueError when it next synchronises.
#include <stdlib.h>

#include "af_common.h"
#include "af_coop_io.h"

namespace {

constexpr int THREADS = 256;

template <typename T> struct R;
template <> struct R<double> {
    static __device__ __forceinline__ double mul(double a, double b) { return __dmul_rn(a, b); }
    static __device__ __forceinline__ double add(double a, double b) { return __dadd_rn(a, b); }
    static __device__ __forceinline__ double sub(double a, double b) { return __dsub_rn(a, b); }
    typedef double2 vec2;
};
template <> struct R<float> {
    static __device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
    static __device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
    static __device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }
    typedef float2 vec2;
};

template <typename T> struct Cx {
    T re, im;
};
// (a+bi)(c+di) = (ac - bd) + (ad + bc)i : numba's complex multiply, no contraction
template <typename T> __device__ __forceinline__ Cx<T> cmul(Cx<T> a, Cx<T> b)
{
    using O = R<T>;
    Cx<T> z;
    z.re = O::sub(O::mul(a.re, b.re), O::mul(a.im, b.im));
    z.im = O::add(O::mul(a.re, b.im), O::mul(a.im, b.re));
    return z;
}
template <typename T> __device__ __forceinline__ Cx<T> cadd(Cx<T> a, Cx<T> b)
{
    using O = R<T>;
    Cx<T> z;
    z.re = O::add(a.re, b.re);
    z.im = O::add(a.im, b.im);
    return z;
}
template <typename T> __device__ __forceinline__ Cx<T> cconj(Cx<T> a)
{
    Cx<T> z;
    z.re = a.re;
    z.im = -a.im;
    return z;
}

template <typename T, int NC> __device__ __forceinline__ void load_jones(const T *p, Cx<T> (&j)[NC])
{
    const typename R<T>::vec2 *q = reinterpret_cast<const typename R<T>::vec2 *>(p);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        typename R<T>::vec2 v = q[c];
        j[c].re = v.x;
        j[c].im = v.y;
    }
}

// streamed once, never re-read: non-temporal loads keep source_coh from evicting the Jones slabs the XCD's L2 is
// meant to hold (tile kernel)
template <typename T, int NC> __device__ __forceinline__ void load_jones_nt(const T *p, Cx<T> (&j)[NC])
{
    typedef T native2 __attribute__((ext_vector_type(2)));
    const native2 *q = reinterpret_cast<const native2 *>(p);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const native2 v = __builtin_nontemporal_load(q + c);
        j[c].re = v.x;
        j[c].im = v.y;
    }
}

// a1 * bl * a2^H (predict.py:93-122); 2x2: A1 . (BL . A2^H)
template <typename T, int NC, bool J2X2>
__device__ __forceinline__ void jones_mul3(const Cx<T> (&a1)[NC], const Cx<T> (&bl)[NC], const Cx<T> (&a2)[NC],
                                           Cx<T> (&r)[NC])
{
    if constexpr (J2X2) {
        Cx<T> xxH = cconj(a2[0]), xyH = cconj(a2[1]), yxH = cconj(a2[2]), yyH = cconj(a2[3]);
        Cx<T> xx = cadd(cmul(bl[0], xxH), cmul(bl[1], xyH));
        Cx<T> xy = cadd(cmul(bl[0], yxH), cmul(bl[1], yyH));
        Cx<T> yx = cadd(cmul(bl[2], xxH), cmul(bl[3], xyH));
        Cx<T> yy = cadd(cmul(bl[2], yxH), cmul(bl[3], yyH));
        r[0] = cadd(cmul(a1[0], xx), cmul(a1[1], yx));
        r[1] = cadd(cmul(a1[0], xy), cmul(a1[1], yy));
        r[2] = cadd(cmul(a1[2], xx), cmul(a1[3], yx));
        r[3] = cadd(cmul(a1[2], xy), cmul(a1[3], yy));
    } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) r[c] = cmul(cmul(a1[c], bl[c]), cconj(a2[c]));
    }
}

// jones_mul3 in two steps with identical operations and order: x = bl . a2^H, then r = a1 . x
template <typename T, int NC, bool J2X2>
__device__ __forceinline__ void jones_right(const Cx<T> (&bl)[NC], const Cx<T> (&a2)[NC], Cx<T> (&x)[NC])
{
    if constexpr (J2X2) {
        Cx<T> xxH = cconj(a2[0]), xyH = cconj(a2[1]), yxH = cconj(a2[2]), yyH = cconj(a2[3]);
        x[0] = cadd(cmul(bl[0], xxH), cmul(bl[1], xyH));
        x[1] = cadd(cmul(bl[0], yxH), cmul(bl[1], yyH));
        x[2] = cadd(cmul(bl[2], xxH), cmul(bl[3], xyH));
        x[3] = cadd(cmul(bl[2], yxH), cmul(bl[3], yyH));
    }
}
template <typename T, int NC, bool J2X2>
__device__ __forceinline__ void jones_left(const Cx<T> (&a1)[NC], const Cx<T> (&x)[NC], Cx<T> (&r)[NC])
{
    if constexpr (J2X2) {
        r[0] = cadd(cmul(a1[0], x[0]), cmul(a1[1], x[2]));
        r[1] = cadd(cmul(a1[0], x[1]), cmul(a1[1], x[3]));
        r[2] = cadd(cmul(a1[2], x[0]), cmul(a1[3], x[2]));
        r[3] = cadd(cmul(a1[2], x[1]), cmul(a1[3], x[3]));
    }
}

// a1 * a2^H (predict.py:129-147)
template <typename T, int NC, bool J2X2>
__device__ __forceinline__ void jones_mul2(const Cx<T> (&a1)[NC], const Cx<T> (&a2)[NC], Cx<T> (&r)[NC])
{
    if constexpr (J2X2) {
        Cx<T> xxH = cconj(a2[0]), xyH = cconj(a2[1]), yxH = cconj(a2[2]), yyH = cconj(a2[3]);
        r[0] = cadd(cmul(a1[0], xxH), cmul(a1[1], xyH));
        r[1] = cadd(cmul(a1[0], yxH), cmul(a1[1], yyH));
        r[2] = cadd(cmul(a1[2], xxH), cmul(a1[3], xyH));
        r[3] = cadd(cmul(a1[2], yxH), cmul(a1[3], yyH));
    } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) r[c] = cmul(a1[c], cconj(a2[c]));
    }
}

// Index guard: clamps (ti, a1, a2) into their arrays, reports in *status; true = the row's result must be NaN.
__device__ __forceinline__ bool guard_indices(int64_t &ti, int64_t &a1, int64_t &a2, int64_t ntime, int64_t nant,
                                              int *status)
{
    int flags = 0;
    if (ti < 0 || ti >= ntime) { flags |= AF_STATUS_TIME_INDEX; ti = ti < 0 ? 0 : (ntime > 0 ? ntime - 1 : 0); }
    if (a1 < 0 || a1 >= nant) { flags |= AF_STATUS_ANTENNA; a1 = 0; }
    if (a2 < 0 || a2 >= nant) { flags |= AF_STATUS_ANTENNA; a2 = 0; }
    if (flags) atomicOr(status, flags);
    return flags != 0;
}

template <typename T, int NC> __device__ __forceinline__ void store_cell(T *p, const Cx<T> (&acc)[NC], bool bad)
{
    typename R<T>::vec2 *o = reinterpret_cast<typename R<T>::vec2 *>(p);
    const T nan = (T)__builtin_nan("");
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        typename R<T>::vec2 v;
        v.x = bad ? nan : acc[c].re;
        v.y = bad ? nan : acc[c].im;
        o[c] = v;
    }
}

// tmin = min(time_index) (predict.py:597) without a host round trip
template <typename I>
__global__ void time_min_kernel(const I *__restrict__ time_index, int64_t nrow, long long *__restrict__ tmin)
{
    long long m = 0x7fffffffffffffffLL;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += (int64_t)gridDim.x * blockDim.x) {
        long long t = (long long)time_index[i];
        m = t < m ? t : m;
    }
    for (int off = 32; off > 0; off >>= 1) {
        long long o = __shfl_down(m, off, 64);
        m = o < m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMin(tmin, m);
}

// grid: ceil(nrow*nchan / 256) blocks of 256 lanes; lane = one (row, chan) cell.
template <typename T, typename I, int NC, bool J2X2, bool HAVE_DDES, bool HAVE_COH>
__global__ __launch_bounds__(THREADS) void predict_vis_kernel(
    const I *__restrict__ time_index, const I *__restrict__ ant1, const I *__restrict__ ant2, int64_t nrow,
    const T *__restrict__ dde1, const T *__restrict__ coh, const T *__restrict__ dde2,
    const T *__restrict__ die1, const T *__restrict__ bvis, const T *__restrict__ die2, int64_t nsrc,
    int64_t ntime, int64_t nant, int64_t nchan, const long long *__restrict__ tmin_p, int *__restrict__ status,
    T *__restrict__ out)
{
    const int64_t cell = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int64_t ncell = nrow * nchan;
    if (cell >= ncell) return;
    const int64_t r = cell / nchan, f = cell - r * nchan;
    const bool have_dies = die1 != nullptr;
    int64_t ti = 0, a1 = 0, a2 = 0;
    bool bad = false;
    if (HAVE_DDES || have_dies) {
        ti = (int64_t)time_index[r] - (int64_t)(*tmin_p);
        a1 = (int64_t)ant1[r];
        a2 = (int64_t)ant2[r];
        bad = guard_indices(ti, a1, a2, ntime, nant, status);
    }
    constexpr int CS = NC * 2;  // reals per cell
    Cx<T> acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c].re = acc[c].im = (T)0;

    // sum over sources, ascending (predict.py:199-246)
    if (HAVE_DDES || HAVE_COH) {
        const int64_t sstride_dde = ntime * nant * nchan * CS;
        const int64_t sstride_coh = ncell * CS;
        const T *p1 = HAVE_DDES ? dde1 + ((ti * nant + a1) * nchan + f) * CS : nullptr;
        const T *p2 = HAVE_DDES ? dde2 + ((ti * nant + a2) * nchan + f) * CS : nullptr;
        const T *pb = HAVE_COH ? coh + cell * CS : nullptr;
#pragma unroll 2
        for (int64_t s = 0; s < nsrc; ++s) {
            Cx<T> j1[NC], jb[NC], j2[NC], rr[NC];
            if (HAVE_DDES) {
                load_jones<T, NC>(p1 + s * sstride_dde, j1);
                load_jones<T, NC>(p2 + s * sstride_dde, j2);
            }
            if (HAVE_COH) load_jones<T, NC>(pb + s * sstride_coh, jb);
            if (HAVE_DDES && HAVE_COH) {
                jones_mul3<T, NC, J2X2>(j1, jb, j2, rr);
            } else if (HAVE_DDES) {
                jones_mul2<T, NC, J2X2>(j1, j2, rr);
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) rr[c] = jb[c];
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = cadd(acc[c], rr[c]);
        }
    }
    // out += base_vis (predict.py:329-339)
    if (bvis != nullptr) {
        Cx<T> b[NC];
        load_jones<T, NC>(bvis + cell * CS, b);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = cadd(acc[c], b[c]);
    }
    // out = die1 . out . die2^H (predict.py:353-367)
    if (have_dies) {
        Cx<T> g1[NC], g2[NC], rr[NC];
        load_jones<T, NC>(die1 + ((ti * nant + a1) * nchan + f) * CS, g1);
        load_jones<T, NC>(die2 + ((ti * nant + a2) * nchan + f) * CS, g2);
        jones_mul3<T, NC, J2X2>(g1, acc, g2, rr);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = rr[c];
    }
    store_cell<T, NC>(out + cell * CS, acc, bad);
}

// ---- (row block, chan tile) kernel: per-antenna DDE terms staged in LDS once per source ---------------------
// block -> (row block, chan tile): the dispatcher deals consecutive blocks round-robin over the 8 XCDs, so block i
// lives on XCD i % 8; the j = i / 8-th block of an XCD takes chan tile j % nct of that XCD's (j / nct)-th row block,
// and an XCD's row blocks come in groups of G consecutive ones (~ one timestep) before the next XCD's group starts.
// Default since round 3 ("rows first", group < 0): within an XCD's turn the G row blocks of ONE chan tile come first,
// then the next chan tile -- the ~64 workgroups resident on an XCD then copy the same Jones segments (same timestep, same
// channels) and all but the first find them in the XCD's L2: 4.75 -> 5.15 TB/s with DDE terms, the rate of the kernel
// without any gathers (tools/bench_predict_tile.py; AFHIP_PREDICT_ROWS_FIRST=0 restores chan tiles first).
template <typename T, typename I, int NC, bool J2X2, bool HAVE_COH, int CT, int TB, int CPT>
__global__ __launch_bounds__(TB) void predict_vis_tile_kernel(
    const I *__restrict__ time_index, const I *__restrict__ ant1, const I *__restrict__ ant2, int64_t nrow,
    const T *__restrict__ dde1, const T *__restrict__ coh, const T *__restrict__ dde2,
    const T *__restrict__ die1, const T *__restrict__ bvis, const T *__restrict__ die2, int64_t nsrc,
    int64_t ntime, int64_t nant, int64_t nchan, const long long *__restrict__ tmin_p, int *__restrict__ status,
    T *__restrict__ out, int nct, int64_t nrb, int group, int ts_max, int stage_reals, int trips_max, int die_lds)
{
    constexpr int RS = TB / CT;                      // rows per cell slot: a thread owns rows rl + k RS, k < CPT
    constexpr int RB = RS * CPT;                     // rows per block
    constexpr int CS = NC * 2;                       // reals per cell
    constexpr int UNIT = 16 / (int)sizeof(T);        // reals per 16-byte unit
    constexpr int SEG_UNITS = CT * CS / UNIT;        // units per (timestep, antenna) segment of the tile
    static_assert(CT * CS % UNIT == 0, "a tile segment is whole 16-byte units");
    extern __shared__ double2 tile_lds[];
    __shared__ long long t_lo_hi[2];

    const int64_t i = blockIdx.x;
    const int xcd = (int)(i & 7);
    const int64_t j = i >> 3;
    int ct;
    int64_t rb;
    if (group < 0) {
        // rows first: the `group` row blocks of an XCD's turn (~ one timestep) take the same chan tile one after the
        // other, so that the workgroups resident together on an XCD copy the SAME Jones segments
        const int g = -group;
        const int64_t turn = j / ((int64_t)nct * g), within = j % ((int64_t)nct * g);
        ct = (int)(within / g);
        rb = (turn * 8 + xcd) * g + within % g;
    } else {
        ct = (int)(j % nct);
        const int64_t q = j / nct;
        rb = ((q / group) * 8 + xcd) * group + q % group;
    }
    if (rb >= nrb) return;

    const int tid = threadIdx.x;
    const int fl = tid % CT, rl = tid / CT;
    const int64_t f0 = (int64_t)ct * CT, f = f0 + fl;
    const int64_t fc = f < nchan ? f : nchan - 1;
    int64_t ti[CPT], a1[CPT], a2[CPT], cell[CPT];
    bool bad[CPT], live[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int64_t r = rb * RB + k * RS + rl;
        live[k] = r < nrow && f < nchan;
        const int64_t rc = r < nrow ? r : nrow - 1;  // rows past the end read the last row's indices
        ti[k] = (int64_t)time_index[rc] - (int64_t)(*tmin_p);
        a1[k] = (int64_t)ant1[rc];
        a2[k] = (int64_t)ant2[rc];
        bad[k] = guard_indices(ti[k], a1[k], a2[k], ntime, nant, status);
        cell[k] = rc * nchan + fc;
    }

    // the block's timestep range (bad indices arrive clamped)
    if (tid == 0) { t_lo_hi[0] = 0x7fffffffffffffffLL; t_lo_hi[1] = 0; }   // clamped indices are >= 0: unsigned atomics
    __syncthreads();
    if (fl == 0) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            atomicMin((unsigned long long *)&t_lo_hi[0], (unsigned long long)ti[k]);
            atomicMax((unsigned long long *)&t_lo_hi[1], (unsigned long long)ti[k]);
        }
    }
    __syncthreads();
    const int64_t tlo = t_lo_hi[0];
    const int tsb = (int)(t_lo_hi[1] - tlo) + 1;
    const bool use_lds = tsb <= ts_max;              // block-uniform; otherwise per-lane gathers
    const int trips = (tsb * (int)nant * SEG_UNITS + TB - 1) / TB;   // copy instructions per wave and source (<= trips_max)

    const int64_t ncell = nrow * nchan;
    Cx<T> acc[CPT][NC];
#pragma unroll
    for (int k = 0; k < CPT; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[k][c].re = acc[k][c].im = (T)0;

    const int64_t sstride_dde = ntime * nant * nchan * CS;
    const int64_t sstride_coh = ncell * CS;

    // staging of per-antenna terms (the block's timesteps x every antenna x the tile's channels) into an LDS stage
    T *lds = reinterpret_cast<T *>(tile_lds);
    const int units = tsb * (int)nant * SEG_UNITS;
    const int wave = tid >> 6, lane = tid & 63;
    const int64_t seg_stride = nchan * CS;       // reals between antennas of one (source, timestep)
    // a tile that sticks out of the band copies the band's LAST CT channels instead (never reads past the array;
    // nchan >= CT is a launch condition) and its lanes find channel f at position f - f0c of the segment
    const int64_t f0c = f0 + CT <= nchan ? f0 : nchan - CT;
    const int flc = f < nchan ? (int)(f - f0c) : 0;
    int o1[CPT], o2[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        o1[k] = (((int)(ti[k] - tlo) * (int)nant + (int)a1[k]) * CT + flc) * CS;
        o2[k] = (((int)(ti[k] - tlo) * (int)nant + (int)a2[k]) * CT + flc) * CS;
    }
    // (timestep tlo, antenna 0, channel f0c) of a (time, antenna, chan, corr) array -> stage st
    auto stage_copy = [&](const T *src0, int st) {
        T *dst0 = lds + st * stage_reals;
        for (int t = 0; t < trips; ++t) {
            const int ebase = t * TB + wave * 64;    // wave-uniform
            int u = ebase + lane;
            u = u < units ? u : units - 1;
            const int seg = u / SEG_UNITS, k = u - seg * SEG_UNITS;
            const T *g = src0 + (int64_t)seg * seg_stride + k * UNIT;
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dst0 + ebase * UNIT));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
        }
    };
    const int64_t slab0 = tlo * nant * seg_stride + f0c * CS;

    if (use_lds && nsrc > 0) {
        // Three LDS stages: in iteration s the coherencies of source s + 1 are requested into registers, THEN the
        // copy of source s + 2's Jones terms is issued, source s is computed from stage s % 3, and the wait at the end
        // leaves exactly that copy outstanding (every wave issues the same `trips` copy instructions per source --
        // lanes past the end of the block's segment list re-copy the last unit into the stage's padding -- so the
        // count is one immediate per trip count).  One barrier per source.
        const T *src_t = dde1 + slab0;
        auto stage_load = [&](int64_t s, int st) { stage_copy(src_t + s * sstride_dde, st); };
        // wait until only the youngest `trips` vector-memory operations of this wave (one source's copy) are outstanding
        auto wait_keep_one_copy = [&]() {
            switch (trips) {
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        };

        Cx<T> cur[CPT][NC], nxt[CPT][NC];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if (HAVE_COH) load_jones<T, NC>(coh + cell[k] * CS, cur[k]);
#pragma unroll
            for (int c = 0; c < NC; ++c) nxt[k][c].re = nxt[k][c].im = (T)0;
        }
        stage_load(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (nsrc > 1) stage_load(1, 1);
        __syncthreads();
        int st = 0;
        for (int64_t s = 0; s < nsrc; ++s) {
            if (HAVE_COH && s + 1 < nsrc) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) load_jones<T, NC>(coh + (s + 1) * sstride_coh + cell[k] * CS, nxt[k]);
            }
            const bool more = s + 2 < nsrc;
            if (more) stage_load(s + 2, st >= 1 ? st - 1 : 2);     // (st + 2) % 3
            const T *base = lds + st * stage_reals;
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                Cx<T> j1[NC], j2[NC], rr[NC];
                load_jones<T, NC>(base + o1[k], j1);
                load_jones<T, NC>(base + o2[k], j2);
                if (HAVE_COH) jones_mul3<T, NC, J2X2>(j1, cur[k], j2, rr);
                else jones_mul2<T, NC, J2X2>(j1, j2, rr);
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[k][c] = cadd(acc[k][c], rr[c]);
            }
            if (more) wait_keep_one_copy();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (HAVE_COH) {
#pragma unroll
                for (int k = 0; k < CPT; ++k)
#pragma unroll
                    for (int c = 0; c < NC; ++c) cur[k][c] = nxt[k][c];
            }
            st = st == 2 ? 0 : st + 1;
        }
    } else if (nsrc > 0) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const T *p1 = dde1 + ((ti[k] * nant + a1[k]) * nchan + fc) * CS;
            const T *p2 = dde2 + ((ti[k] * nant + a2[k]) * nchan + fc) * CS;
            for (int64_t s = 0; s < nsrc; ++s) {
                Cx<T> j1[NC], jb[NC], j2[NC], rr[NC];
                load_jones<T, NC>(p1 + s * sstride_dde, j1);
                load_jones<T, NC>(p2 + s * sstride_dde, j2);
                if (HAVE_COH) {
                    load_jones<T, NC>(coh + s * sstride_coh + cell[k] * CS, jb);
                    jones_mul3<T, NC, J2X2>(j1, jb, j2, rr);
                } else {
                    jones_mul2<T, NC, J2X2>(j1, j2, rr);
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[k][c] = cadd(acc[k][c], rr[c]);
            }
        }
    }
    // out += base_vis (predict.py:329-339), then out = die1 . out . die2^H (:353-367): the DIE terms of the block's
    // timesteps go through an LDS stage like the DDE terms (die1 == die2 as every caller passes them: `die_lds`);
    // every stage is free here (the source loop ended with a barrier)
    const bool die_staged = die_lds && use_lds;          // block-uniform
    if (die_staged) stage_copy(die1 + slab0, 0);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        if (bvis != nullptr && live[k]) {
            Cx<T> b[NC];
            load_jones<T, NC>(bvis + cell[k] * CS, b);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[k][c] = cadd(acc[k][c], b[c]);
        }
    }
    if (die_staged) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        if (!live[k]) continue;
        if (die1 != nullptr) {
            Cx<T> g1[NC], g2[NC], rr[NC];
            if (die_staged) {
                load_jones<T, NC>(lds + o1[k], g1);
                load_jones<T, NC>(lds + o2[k], g2);
            } else {
                load_jones<T, NC>(die1 + ((ti[k] * nant + a1[k]) * nchan + f) * CS, g1);
                load_jones<T, NC>(die2 + ((ti[k] * nant + a2[k]) * nchan + f) * CS, g2);
            }
            jones_mul3<T, NC, J2X2>(g1, acc[k], g2, rr);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[k][c] = rr[c];
        }
        store_cell<T, NC>(out + cell[k] * CS, acc[k], bad[k]);
    }
}

// ---- (row block, chan tile) kernel, streamed form ----------------------------------------------------------------
// Measured on the form above (profiles/r03_aux_bench_predict_tile_pmc_hbm.json): FETCH 24.4 GB for 17.2 GB of
// coherencies -- essentially every per-source Jones copy (0.5 byte per coherency byte at 128 rows per block) comes from
// beyond the XCD's L2, whatever the block order and the cache policy of the coherency loads -- and 24.4 GB in 3.9 ms IS
// the 6.3 TB/s the fabric delivers: the kernel's time is its fetch.  So the copy has to serve more rows: here a
// workgroup of 512 lanes owns K sub-blocks of 128 rows (K cells per lane) and ONE copy of a source's Jones terms serves
// all of them: 0.5 / K byte per coherency byte.  K = 2 for 64-byte cells (c128 2 x 2: four sub-blocks need more than the
// 256 registers of two waves per SIMD -- the compiler spills, and a spill is a vector-memory operation inside counted
// waits), K = 4 for narrower cells.  The registers that takes are free because
// the coherencies travel through LDS too: a ring of three 32 KB coherency tiles (one sub-block of one source each),
// copied by global_load_lds_dwordx4 two tiles ahead, beside two Jones slots of up to TWO timesteps each (a block of
// K x 128 rows straddles a timestep boundary one time in 16 / K at 2016 rows per timestep) = exactly the CU's 160 KB at
// 64 antennas, c128 2 x 2.  Every load of
// the loop is an asm-issued LDS copy, counted by one vmcnt immediate per tile step; one barrier per tile step.  Same
// arithmetic in the same order: bit-identical.
template <typename T, typename I, int NC, bool J2X2, int CT, int TB, int K>
__global__ __launch_bounds__(TB) void predict_vis_stream_kernel(
    const I *__restrict__ time_index, const I *__restrict__ ant1, const I *__restrict__ ant2, int64_t nrow,
    const T *__restrict__ dde1, const T *__restrict__ coh, const T *__restrict__ dde2,
    const T *__restrict__ die1, const T *__restrict__ bvis, const T *__restrict__ die2, int64_t nsrc,
    int64_t ntime, int64_t nant, int64_t nchan, const long long *__restrict__ tmin_p, int *__restrict__ status,
    T *__restrict__ out, int nct, int64_t nrb, int group, int ts_max, int jones_reals, int coh_reals)
{
    constexpr int RS = TB / CT;                      // rows of a sub-block
    constexpr int RB = RS * K;                       // rows of the block
    constexpr int CS = NC * 2;
    constexpr int UNIT = 16 / (int)sizeof(T);
    constexpr int SEG_UNITS = CT * CS / UNIT;        // units per (timestep, antenna) Jones segment = per row of a coherency tile
    static_assert(CT * CS % UNIT == 0, "a tile segment is whole 16-byte units");
    constexpr int COH_UNITS = RS * SEG_UNITS;
    constexpr int TRIPS_C = (COH_UNITS + TB - 1) / TB;
    extern __shared__ double2 tile_lds[];
    T *lds = reinterpret_cast<T *>(tile_lds);
    T *ldsJ = lds;                                   // 2 slots of jones_reals
    T *ldsC = lds + 2 * jones_reals;                 // 3 slots of coh_reals
    long long *t_lo_hi = reinterpret_cast<long long *>(ldsC + 2 * coh_reals);   // scratch in the third coherency slot, before the loop

    const int64_t i = blockIdx.x;
    const int xcd = (int)(i & 7);
    const int64_t jb_ = i >> 3;
    int ct;
    int64_t rb;
    if (group < 0) {    // rows first (see predict_vis_tile_kernel)
        const int g = -group;
        const int64_t turn = jb_ / ((int64_t)nct * g), within = jb_ % ((int64_t)nct * g);
        ct = (int)(within / g);
        rb = (turn * 8 + xcd) * g + within % g;
    } else {
        ct = (int)(jb_ % nct);
        const int64_t q = jb_ / nct;
        rb = ((q / group) * 8 + xcd) * group + q % group;
    }
    if (rb >= nrb) return;

    const int tid = threadIdx.x;
    const int fl = tid % CT, rl = tid / CT;
    const int64_t f0 = (int64_t)ct * CT, f = f0 + fl;
    const int64_t fc = f < nchan ? f : nchan - 1;
    // per-cell state is RE-DERIVED where it is needed (prologue, fallback, epilogue) instead of being kept: K cells x
    // (time, two antennas, cell index) in 64 bits would cost 32 registers the accumulators need
    auto cell_state = [&](int k, int64_t &ti, int64_t &a1, int64_t &a2, int64_t &cell, bool &live) -> bool {
        // the pointers pass through an empty asm so that the compiler re-reads the indices at every use instead of
        // keeping K x 4 64-bit values of the prologue alive across the source loop for the epilogue
        const I *tp = time_index, *p1 = ant1, *p2 = ant2;
        asm volatile("" : "+s"(tp), "+s"(p1), "+s"(p2));
        const int64_t r = rb * RB + k * RS + rl;
        live = r < nrow && f < nchan;
        const int64_t rc = r < nrow ? r : nrow - 1;
        ti = (int64_t)tp[rc] - (int64_t)(*tmin_p);
        a1 = (int64_t)p1[rc];
        a2 = (int64_t)p2[rc];
        cell = rc * nchan + fc;
        return guard_indices(ti, a1, a2, ntime, nant, status);
    };
    if (tid == 0) { t_lo_hi[0] = 0x7fffffffffffffffLL; t_lo_hi[1] = 0; }
    __syncthreads();
    if (fl == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            int64_t ti, a1, a2, cell; bool live;
            cell_state(k, ti, a1, a2, cell, live);
            atomicMin((unsigned long long *)&t_lo_hi[0], (unsigned long long)ti);
            atomicMax((unsigned long long *)&t_lo_hi[1], (unsigned long long)ti);
        }
    }
    __syncthreads();
    const int64_t tlo = t_lo_hi[0];
    const int tsb = (int)(t_lo_hi[1] - tlo) + 1;
    // block-uniform; the array's last, partial block takes the per-lane gathers too (no clamping in the copy loops)
    const bool use_lds = tsb <= ts_max && rb * RB + RB <= nrow;
    __syncthreads();                                 // the scratch is about to be reused as a coherency slot

    const int64_t ncell = nrow * nchan;
    Cx<T> acc[K][NC];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[k][c].re = acc[k][c].im = (T)0;
    const int64_t sstride_dde = ntime * nant * nchan * CS;
    const int64_t sstride_coh = ncell * CS;

    // base_vis, DIE terms, store (predict.py:329-367) for cell k
    struct Cell { Cx<T> c[NC]; };
    auto finish_cell = [&](int k, Cell cellv) {
        Cx<T> (&v)[NC] = cellv.c;
        int64_t ti, a1, a2, cell; bool live;
        const bool bad = cell_state(k, ti, a1, a2, cell, live);
        if (!live) return;
        if (bvis != nullptr) {
            Cx<T> b[NC];
            load_jones<T, NC>(bvis + cell * CS, b);
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = cadd(v[c], b[c]);
        }
        if (die1 != nullptr) {
            Cx<T> g1[NC], g2[NC], rr[NC];
            load_jones<T, NC>(die1 + ((ti * nant + a1) * nchan + f) * CS, g1);
            load_jones<T, NC>(die2 + ((ti * nant + a2) * nchan + f) * CS, g2);
            jones_mul3<T, NC, J2X2>(g1, v, g2, rr);
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = rr[c];
        }
        store_cell<T, NC>(out + cell * CS, v, bad);
    };

    if (use_lds) {
        const int units_j = tsb * (int)nant * SEG_UNITS;
        const int trips_j = (units_j + TB - 1) / TB;
        const int wave = tid >> 6;
        const int64_t seg_stride = nchan * CS;
        const int64_t f0c = f0 + CT <= nchan ? f0 : nchan - CT;     // a tile that sticks out copies the band's last CT channels
        const int flc = f < nchan ? (int)(f - f0c) : 0;
        int o1[K], o2[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            int64_t ti, a1, a2, cell; bool live;
            cell_state(k, ti, a1, a2, cell, live);
            o1[k] = (((int)(ti - tlo) * (int)nant + (int)a1) * CT + flc) * CS;
            o2[k] = (((int)(ti - tlo) * (int)nant + (int)a2) * CT + flc) * CS;
        }
        const int oc = (rl * CT + flc) * CS;
        const T *src_j = dde1 + tlo * nant * seg_stride + f0c * CS;
        const T *src_c = coh + f0c * CS;
        auto dma = [&](const T *g, T *dst) {
            const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)dst);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(d) : "memory");
        };
        // every wave issues the same number of copy instructions (lanes past the end re-copy the last unit into padding).
        // Per-lane source offsets are worked out ONCE (they do not depend on the source); a copy is then one 64-bit add.
        constexpr int MAXTJ = 8;
        int offj[MAXTJ];                              // reals from the source's first segment: < 2^31 (one block's timesteps)
#pragma unroll
        for (int t = 0; t < MAXTJ; ++t) {
            int u = t * TB + tid;
            u = u < units_j ? u : units_j - 1;
            const int seg = u / SEG_UNITS, kk = u - seg * SEG_UNITS;
            offj[t] = seg * (int)seg_stride + kk * UNIT;
        }
        // coherency tile of sub-block k: rows rb RB + k RS + row; offsets relative to the block's first row
        int offc[TRIPS_C];
#pragma unroll
        for (int t = 0; t < TRIPS_C; ++t) {
            int u = t * TB + tid;
            u = u < COH_UNITS ? u : COH_UNITS - 1;
            const int row = u / SEG_UNITS, kk = u - row * SEG_UNITS;
            offc[t] = row * (int)seg_stride + kk * UNIT;
        }
        const T *src_cb = src_c + rb * RB * seg_stride;          // block-uniform
        const int64_t sub_stride = (int64_t)RS * seg_stride;
        auto load_jones_stage = [&](int sslot, const T *sj) {
            T *dst0 = ldsJ + sslot * jones_reals;
#pragma unroll
            for (int t = 0; t < MAXTJ; ++t)
                if (t < trips_j) dma(sj + offj[t], dst0 + (t * TB + wave * 64) * UNIT);
        };
        auto load_coh_tile = [&](int cslot, int k, const T *sc) {    // sc: the source's slab at the block's first row
            T *dst0 = ldsC + cslot * coh_reals;
            const T *sk = sc + k * sub_stride;
#pragma unroll
            for (int t = 0; t < TRIPS_C; ++t) dma(sk + offc[t], dst0 + (t * TB + wave * 64) * UNIT);
        };
        auto wait_keep = [&](int n) {                // all but the youngest n copy instructions have landed
            switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        };
        // tile steps n = K s + k; coherency slot n % 3, Jones slot s % 2; the step after next is requested first
        const T *sj_next = src_j + sstride_dde;       // Jones of source s + 1
        int cs_req = 2 % 3, k_req = 2 % K;           // slot / sub-block of the NEXT tile to request (step n + 2)
        const T *sc_req = src_cb + (2 / K) * sstride_coh;
        load_jones_stage(0, src_j);
        load_coh_tile(0, 0, src_cb);
        if (nsrc * K > 1) { load_coh_tile(1, 1 % K, src_cb + (1 / K) * sstride_coh); wait_keep(TRIPS_C); }
        else wait_keep(0);
        __syncthreads();
        int cslot = 0;
        for (int64_t s = 0; s < nsrc; ++s) {
            const T *jbase = ldsJ + (int)(s & 1) * jones_reals;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                int outstanding = 0;                 // copies issued in this step: they may stay in flight over the barrier
                if (s * K + k + 2 < nsrc * K) {
                    load_coh_tile(cs_req, k_req, sc_req);
                    outstanding += TRIPS_C;
                    cs_req = cs_req == 2 ? 0 : cs_req + 1;
                    if (++k_req == K) { k_req = 0; sc_req += sstride_coh; }
                }
                if (k == 0 && s + 1 < nsrc) {
                    load_jones_stage((int)((s + 1) & 1), sj_next);
                    sj_next += sstride_dde;
                    outstanding += trips_j;
                }
                const T *cbase = ldsC + cslot * coh_reals;
                {
                    // operands loaded when they are needed (coherency and right-hand Jones first, the left-hand Jones
                    // after their product): the scheduler otherwise front-loads all twelve reads of every sub-block
                    Cx<T> jb[NC], j2[NC], x[NC], j1[NC], rr[NC];
                    load_jones<T, NC>(cbase + oc, jb);
                    if constexpr (J2X2) {            // A1 . (BL . A2^H)
                        load_jones<T, NC>(jbase + o2[k], j2);
                        jones_right<T, NC, J2X2>(jb, j2, x);
                        __builtin_amdgcn_sched_barrier(0);
                        load_jones<T, NC>(jbase + o1[k], j1);
                        jones_left<T, NC, J2X2>(j1, x, rr);
                    } else {                         // (a1 bl) conj(a2), element by element
                        load_jones<T, NC>(jbase + o1[k], j1);
#pragma unroll
                        for (int c = 0; c < NC; ++c) x[c] = cmul(j1[c], jb[c]);
                        __builtin_amdgcn_sched_barrier(0);
                        load_jones<T, NC>(jbase + o2[k], j2);
#pragma unroll
                        for (int c = 0; c < NC; ++c) rr[c] = cmul(x[c], cconj(j2[c]));
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[k][c] = cadd(acc[k][c], rr[c]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (outstanding == TRIPS_C) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(TRIPS_C) : "memory");
                else wait_keep(outstanding);
                __syncthreads();
                cslot = cslot == 2 ? 0 : cslot + 1;
            }
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            int64_t ti, a1, a2, cell; bool live;
            cell_state(k, ti, a1, a2, cell, live);
            const T *p1 = dde1 + ((ti * nant + a1) * nchan + fc) * CS;
            const T *p2 = dde2 + ((ti * nant + a2) * nchan + fc) * CS;
            Cell sumc;
            Cx<T> (&sum)[NC] = sumc.c;
#pragma unroll
            for (int c = 0; c < NC; ++c) sum[c].re = sum[c].im = (T)0;
            for (int64_t s = 0; s < nsrc; ++s) {
                Cx<T> j1[NC], jb[NC], j2[NC], rr[NC];
                load_jones<T, NC>(p1 + s * sstride_dde, j1);
                load_jones<T, NC>(p2 + s * sstride_dde, j2);
                load_jones<T, NC>(coh + s * sstride_coh + cell * CS, jb);
                jones_mul3<T, NC, J2X2>(j1, jb, j2, rr);
#pragma unroll
                for (int c = 0; c < NC; ++c) sum[c] = cadd(sum[c], rr[c]);
            }
            finish_cell(k, sumc);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        Cell v;
#pragma unroll
        for (int c = 0; c < NC; ++c) v.c[c] = acc[k][c];
        finish_cell(k, v);
    }
}

struct PArgs {
    const void *time_index, *ant1, *ant2;
    int index_bytes;
    int64_t nrow;
    const void *dde1, *coh, *dde2, *die1, *bvis, *die2;
    int64_t nsrc, ntime, nant, nchan;
    long long *tmin;
    int *status;
    void *out;
    hipStream_t st;
};

constexpr int TILE_LDS_BUDGET = 144 * 1024;   // of the CU's 160 KiB (the kernel keeps 16 bytes of static LDS)

inline int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// (row block, chan tile) kernel: returns AF_ENOTSUP when the shape does not suit it (caller falls back)
template <typename T, typename I, int NC, bool J2X2, bool HAVE_COH, int CT, int TB, int CPT>
int launch_tile(const PArgs &a)
{
    constexpr int CS = NC * 2, RB = TB / CT * CPT;
    const size_t seg_bytes = (size_t)CT * CS * sizeof(T);                 // one antenna of one timestep
    const size_t per_ts = (size_t)a.nant * seg_bytes;
    if (per_ts == 0 || a.nchan < CT) return AF_ENOTSUP;
    // a stage holds ts timesteps, padded to whole copy trips of TB x 16 bytes; three stages (two sources ahead).
    // ts = 2 lets a block straddle a timestep boundary; ts = 1 leaves room for more workgroups per CU and sends
    // the straddling blocks (1 in 16 at 2016 rows per timestep) through per-lane gathers.
    auto stage_bytes = [&](int ts) { return af_align_up((size_t)ts * per_ts, (size_t)TB * 16); };
    int ts_max = env_int("AFHIP_PREDICT_TILE_TS", 1);
    ts_max = ts_max < 1 ? 1 : ts_max > 2 ? 2 : ts_max;
    const size_t stages = a.nsrc > 0 ? 3 : 1;                             // DIE terms only: one stage, filled once
    while (ts_max > 1 && stages * stage_bytes(ts_max) > (size_t)TILE_LDS_BUDGET) --ts_max;
    if (stages * stage_bytes(ts_max) > (size_t)TILE_LDS_BUDGET) return AF_ENOTSUP;
    const size_t lds = stages * stage_bytes(ts_max);
    const int stage_reals = (int)(stage_bytes(ts_max) / sizeof(T));
    const int trips_max = (int)(stage_bytes(ts_max) / ((size_t)TB * 16));
    if (trips_max > 4 && a.nsrc > 0) return AF_ENOTSUP;                    // the source loop's counted waits cover 1..4 trips
    const int nct = (int)af_cdiv(a.nchan, CT);
    const int64_t nrb = af_cdiv(a.nrow, RB);
    // row blocks per XCD turn.  Rows first: the ~64 resident workgroups of an XCD are `group` row blocks x 64 / group chan
    // tiles -- more row blocks = more sharing of a Jones segment (fetch 1 + 0.5 / group bytes per coherency byte), more
    // chan tiles = longer contiguous pieces of a row in flight (DRAM pages).  Measured on two boxes (tools/
    // ab_predict_order.py): 16 / 8 / 4 row blocks 5.15 / - / - TB/s on one, 4.50 / 4.55 / 4.62 on the other: 8.
    int group = env_int("AFHIP_PREDICT_ROWS_FIRST", 1) ? (1024 / RB > 0 ? 1024 / RB : 1) : (2048 / RB > 0 ? 2048 / RB : 1);
    if (env_int("AFHIP_PREDICT_GROUP", 0) > 0) group = env_int("AFHIP_PREDICT_GROUP", 0);   // measurement hook
    const int64_t nrb_padded = af_cdiv(nrb, 8 * (int64_t)group) * 8 * group;
    const int64_t blocks = nrb_padded * nct;
    if (blocks >= (1LL << 31)) return AF_ENOTSUP;
    auto kernel = predict_vis_tile_kernel<T, I, NC, J2X2, HAVE_COH, CT, TB, CPT>;
    if (lds + 256 > 64 * 1024)   // dynamic + the kernel's static LDS beyond the default 64 KiB limit
        AF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(TB), lds, a.st, (const I *)a.time_index, (const I *)a.ant1,
                       (const I *)a.ant2, a.nrow, (const T *)a.dde1, (const T *)a.coh, (const T *)a.dde2,
                       (const T *)a.die1, (const T *)a.bvis, (const T *)a.die2, a.nsrc, a.ntime, a.nant, a.nchan,
                       a.tmin, a.status, (T *)a.out, nct, nrb, env_int("AFHIP_PREDICT_ROWS_FIRST", 1) ? -group : group, ts_max,
                       stage_reals, trips_max,
                       (a.die1 != nullptr && a.die1 == a.die2 && ((uintptr_t)a.die1 & 15) == 0 && env_int("AFHIP_PREDICT_DIE_LDS", 1)) ? 1 : 0);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

// streamed form: needs source_coh, and 2 Jones slots + 3 coherency tiles within the CU's 160 KB
template <typename T, typename I, int NC, bool J2X2, int CT, int TB, int K>
int launch_stream(const PArgs &a)
{
    constexpr int CS = NC * 2, RB = TB / CT * K;
    const size_t seg_bytes = (size_t)CT * CS * sizeof(T);
    if (a.nant == 0 || a.nchan < CT || a.coh == nullptr || ((uintptr_t)a.coh & 15) != 0) return AF_ENOTSUP;
    const size_t coh_bytes = af_align_up((size_t)(TB / CT) * seg_bytes, (size_t)TB * 16);
    auto jones_bytes = [&](int ts) { return af_align_up((size_t)ts * a.nant * seg_bytes, (size_t)TB * 16); };
    const size_t lds_cap = 160 * 1024;
    int ts_max = 2;
    while (ts_max > 0 && 2 * jones_bytes(ts_max) + 3 * coh_bytes > lds_cap) --ts_max;
    if (ts_max < 1) return AF_ENOTSUP;
    if (jones_bytes(ts_max) / ((size_t)TB * 16) > 8 || jones_bytes(ts_max) / ((size_t)TB * 16) + coh_bytes / ((size_t)TB * 16) > 12)
        return AF_ENOTSUP;   // the kernel's copy loops cover <= 8 Jones trips, its counted waits <= 12 instructions
    const size_t lds = 2 * jones_bytes(ts_max) + 3 * coh_bytes;
    const int nct = (int)af_cdiv(a.nchan, CT);
    const int64_t nrb = af_cdiv(a.nrow, RB);
    const int group = 2048 / RB > 0 ? 2048 / RB : 1;
    const int64_t nrb_padded = af_cdiv(nrb, 8 * (int64_t)group) * 8 * group;
    const int64_t blocks = nrb_padded * nct;
    if (blocks >= (1LL << 31)) return AF_ENOTSUP;
    auto kernel = predict_vis_stream_kernel<T, I, NC, J2X2, CT, TB, K>;
    AF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(TB), lds, a.st, (const I *)a.time_index, (const I *)a.ant1,
                       (const I *)a.ant2, a.nrow, (const T *)a.dde1, (const T *)a.coh, (const T *)a.dde2,
                       (const T *)a.die1, (const T *)a.bvis, (const T *)a.die2, a.nsrc, a.ntime, a.nant, a.nchan,
                       a.tmin, a.status, (T *)a.out, nct, nrb, env_int("AFHIP_PREDICT_ROWS_FIRST", 1) ? -group : group, ts_max,
                       (int)(jones_bytes(ts_max) / sizeof(T)),
                       (int)(coh_bytes / sizeof(T)));
    AF_LAUNCH_CHECK();
    return AF_OK;
}

// ---- calls without DDE terms: lane = one (row, chan) cell, cooperative IO ----------------------------------------
// out = die1[t, p, nu] . (base_vis + sum_s coh[s]) . die2[t, q, nu]^H  with any of the three parts absent:
// apply_gains (predict.py:623-647: DIE terms + base_vis) and the plain coherency stream (predict.py:229-246).  The
// cell, its coherencies and its two gains are 32- or 64-byte records: every array (and the result) travels through
// af_coop_io.h's wave transposes, whole cache lines per instruction, where the lane-per-record loads of
// predict_vis_kernel touch 32 quarter-used lines per instruction.  The arithmetic is predict_vis_kernel's, operation for
// operation (sources ascending from 0, + base_vis, then jones_mul3).  Measured: DESIGN 3.2.
template <typename T, typename I, int NC, bool J2X2, bool HAVE_COH>
__global__ __launch_bounds__(THREADS) void predict_cell_coop_kernel(
    const I *__restrict__ time_index, const I *__restrict__ ant1, const I *__restrict__ ant2, int64_t nrow,
    const T *__restrict__ coh, int64_t nsrc, const T *__restrict__ die1, const T *__restrict__ bvis,
    const T *__restrict__ die2, int64_t ntime, int64_t nant, int64_t nchan, const long long *__restrict__ tmin_p,
    int *__restrict__ status, T *__restrict__ out)
{
    constexpr int U = NC * 2 * (int)sizeof(T) / 16;    // 16-byte units per cell
    static_assert(U == 2 || U == 4, "32- or 64-byte cells");
    __shared__ double2 lds_all[THREADS / 64][64 * U];
    double2 *lds_wave = lds_all[threadIdx.x >> 6];
    const int64_t ncell = nrow * nchan;
    const int64_t cell_raw = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const bool in_range = cell_raw < ncell;             // out-of-range lanes still take part in the wave's transposes
    const int64_t cell = in_range ? cell_raw : ncell - 1;
    const bool have_dies = die1 != nullptr;             // kernel-uniform
    auto unpack = [](const double2 (&u)[U], Cx<T> (&j)[NC]) {
        if constexpr (sizeof(T) == 8) {
#pragma unroll
            for (int c = 0; c < NC; ++c) { j[c].re = (T)u[c].x; j[c].im = (T)u[c].y; }
        } else {
#pragma unroll
            for (int c = 0; c < NC; c += 2) {       // one unit = two complex64 values
                const unsigned long long lo = (unsigned long long)__double_as_longlong(u[c / 2].x);
                const unsigned long long hi = (unsigned long long)__double_as_longlong(u[c / 2].y);
                j[c].re = (T)__uint_as_float((unsigned)lo); j[c].im = (T)__uint_as_float((unsigned)(lo >> 32));
                j[c + 1].re = (T)__uint_as_float((unsigned)hi); j[c + 1].im = (T)__uint_as_float((unsigned)(hi >> 32));
            }
        }
    };
    Cx<T> acc[NC], rr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c].re = acc[c].im = (T)0;
    double2 ub[U];
    if constexpr (HAVE_COH) {
        // sum over sources, ascending (predict.py:229-246)
        const double2 *p = reinterpret_cast<const double2 *>(coh);
        for (int64_t s = 0; s < nsrc; ++s) {
            coop_gather_units<U>(p + s * ncell * U, (int)cell, ub, lds_wave);
            unpack(ub, rr);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = cadd(acc[c], rr[c]);
        }
    }
    if (bvis != nullptr) {                              // out += base_vis (predict.py:329-339)
        coop_gather_units<U>(reinterpret_cast<const double2 *>(bvis), (int)cell, ub, lds_wave);
        unpack(ub, rr);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = cadd(acc[c], rr[c]);
    }
    bool bad = false;
    if (have_dies) {                                    // out = die1 . out . die2^H (predict.py:353-367)
        const int64_t r = cell / nchan, f = cell - r * nchan;
        int64_t ti = (int64_t)time_index[r] - (int64_t)(*tmin_p), a1 = (int64_t)ant1[r], a2 = (int64_t)ant2[r];
        bad = guard_indices(ti, a1, a2, ntime, nant, status);
        double2 u1[U], u2[U];
        coop_gather_units<U>(reinterpret_cast<const double2 *>(die1), (int)((ti * nant + a1) * nchan + f), u1, lds_wave);
        coop_gather_units<U>(reinterpret_cast<const double2 *>(die2), (int)((ti * nant + a2) * nchan + f), u2, lds_wave);
        Cx<T> g1[NC], g2[NC];
        unpack(u1, g1); unpack(u2, g2);
        jones_mul3<T, NC, J2X2>(g1, acc, g2, rr);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = rr[c];
    }
    const T nan = (T)__builtin_nan("");
    double2 uo[U];
    if constexpr (sizeof(T) == 8) {
#pragma unroll
        for (int c = 0; c < NC; ++c) uo[c] = make_double2(bad ? nan : acc[c].re, bad ? nan : acc[c].im);
    } else {
#pragma unroll
        for (int c = 0; c < NC; c += 2) {
            auto pack = [&](Cx<T> v) {
                const unsigned re = __float_as_uint((float)(bad ? nan : v.re)), im = __float_as_uint((float)(bad ? nan : v.im));
                return __longlong_as_double((long long)(((unsigned long long)im << 32) | re));
            };
            uo[c / 2] = make_double2(pack(acc[c]), pack(acc[c + 1]));
        }
    }
    coop_store_units<U>(reinterpret_cast<double2 *>(out), cell_raw - (threadIdx.x & 63), ncell, uo, lds_wave);
}

// AF_ENOTSUP: cells that are not 32 or 64 bytes, more than 2^31 cells / gain records, small calls, unaligned arrays
template <typename T, typename I, int NC, bool J2X2>
int launch_cell_coop(const PArgs &a)
{
    constexpr int CB = NC * 2 * (int)sizeof(T);        // bytes per cell
    if constexpr (CB != 32 && CB != 64) {
        return AF_ENOTSUP;
    } else {
        const int64_t ncell = a.nrow * a.nchan, ngain = a.ntime * a.nant * a.nchan;
        if (ncell >= (1LL << 31) / (CB / 16) || ngain >= (1LL << 31) / (CB / 16) || ncell < (1LL << 16)) return AF_ENOTSUP;
        if (((uintptr_t)a.die1 | (uintptr_t)a.die2 | (uintptr_t)a.bvis | (uintptr_t)a.coh | (uintptr_t)a.out) & 15) return AF_ENOTSUP;
        const dim3 grid((unsigned)af_cdiv(ncell, THREADS));
        if (a.coh != nullptr)
            hipLaunchKernelGGL((predict_cell_coop_kernel<T, I, NC, J2X2, true>), grid, dim3(THREADS), 0, a.st,
                               (const I *)a.time_index, (const I *)a.ant1, (const I *)a.ant2, a.nrow, (const T *)a.coh, a.nsrc,
                               (const T *)a.die1, (const T *)a.bvis, (const T *)a.die2, a.ntime, a.nant, a.nchan, a.tmin,
                               a.status, (T *)a.out);
        else
            hipLaunchKernelGGL((predict_cell_coop_kernel<T, I, NC, J2X2, false>), grid, dim3(THREADS), 0, a.st,
                               (const I *)a.time_index, (const I *)a.ant1, (const I *)a.ant2, a.nrow, (const T *)a.coh, a.nsrc,
                               (const T *)a.die1, (const T *)a.bvis, (const T *)a.die2, a.ntime, a.nant, a.nchan, a.tmin,
                               a.status, (T *)a.out);
        AF_LAUNCH_CHECK();
        return AF_OK;
    }
}

// whether the tile kernel applies: DDE terms (dde1 == dde2 as every caller passes them -- the stage holds ONE array),
// 16-byte aligned antenna segments, enough rows to fill the chip
template <typename T, int NC>
bool tile_applies(const PArgs &a)
{
    constexpr int CS = NC * 2;
    if (env_int("AFHIP_PREDICT_TILE", 1) == 0) return false;
    if (!a.dde1 || a.dde1 != a.dde2 || a.nsrc < 1) return false;
    if ((a.nchan * CS * sizeof(T)) % 16 != 0 || ((uintptr_t)a.dde1 & 15)) return false;
    if (a.nchan < 4 || a.nrow * a.nchan < (int64_t)1 << 16) return false;
    return true;
}

template <typename T, typename I, int NC, bool J2X2, bool HAVE_DDES, bool HAVE_COH>
int launch(const PArgs &a)
{
    const int64_t blocks = af_cdiv(a.nrow * a.nchan, THREADS);
    AF_REQUIRE(blocks < (1LL << 31), "af_predict_vis: problem too large for one launch");
    hipLaunchKernelGGL((predict_vis_kernel<T, I, NC, J2X2, HAVE_DDES, HAVE_COH>), dim3((unsigned)blocks),
                       dim3(THREADS), 0, a.st, (const I *)a.time_index, (const I *)a.ant1, (const I *)a.ant2,
                       a.nrow, (const T *)a.dde1, (const T *)a.coh, (const T *)a.dde2, (const T *)a.die1,
                       (const T *)a.bvis, (const T *)a.die2, a.nsrc, a.ntime, a.nant, a.nchan, a.tmin, a.status,
                       (T *)a.out);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

template <typename T, typename I, int NC, bool J2X2>
int launch_presence(const PArgs &a)
{
    const bool ddes = a.dde1 != nullptr, coh = a.coh != nullptr;
    if (ddes && tile_applies<T, NC>(a)) {
        // 16 bytes per cell at least: CT = 4 cells of a 2x2 c128 (64 B) make 256-byte segments; narrower cells take
        // wider tiles so that a segment stays >= 128 bytes
        constexpr int CT = (NC * 2 * sizeof(T) >= 64) ? 4 : (NC * 2 * sizeof(T) >= 32) ? 8 : 16;
        int rc;
        // the streamed form is opt-in: on the same box it measures within 3 % of the form below (4.57-4.69 against
        // 4.75 TB/s; tools/bench_predict_tile.py) -- its smaller fetch is paid for with one workgroup per CU
        if (coh && env_int("AFHIP_PREDICT_STREAM", 0) != 0) {
            // sub-blocks per workgroup: 4 where the accumulators leave room (<= 32 bytes per cell), else 2
            constexpr int KSUB = (NC * 2 * sizeof(T) >= 64) ? 2 : 4;
            rc = launch_stream<T, I, NC, J2X2, CT, 512, KSUB>(a);
            if (rc != AF_ENOTSUP) return rc;
        }
        if constexpr (sizeof(T) == 8 && sizeof(I) == 4 && NC == 4 && J2X2) {
            // measurement hook (tools/bench_predict_tile.py): other tile shapes for the c128 2x2 case
            const int ct = env_int("AFHIP_PREDICT_TILE_CT", CT), tb = env_int("AFHIP_PREDICT_TILE_TB", 512);
            const int cpt = env_int("AFHIP_PREDICT_TILE_CPT", 1);
            if (coh && ct == 8 && tb == 1024 && cpt == 1) rc = launch_tile<T, I, NC, J2X2, true, 8, 1024, 1>(a);
            else if (coh && ct == 4 && tb == 512 && cpt == 2) rc = launch_tile<T, I, NC, J2X2, true, 4, 512, 2>(a);
            else rc = coh ? launch_tile<T, I, NC, J2X2, true, CT, 512, 1>(a) : launch_tile<T, I, NC, J2X2, false, CT, 512, 1>(a);
        } else {
            rc = coh ? launch_tile<T, I, NC, J2X2, true, CT, 512, 1>(a) : launch_tile<T, I, NC, J2X2, false, CT, 512, 1>(a);
        }
        if (rc != AF_ENOTSUP) return rc;
    }
    // no DDE terms -- apply_gains (predict.py:623-647: DIE terms + base_vis) and the coherency stream with or without
    // DIE terms / base_vis: lane per cell with cooperative IO (round 4; AFHIP_APPLY_COOP=0 / AFHIP_PREDICT_COOP=0 fall
    // through to round 3's forms below)
    if (!ddes && (coh ? env_int("AFHIP_PREDICT_COOP", 1) != 0 : (a.die1 != nullptr && a.bvis != nullptr && env_int("AFHIP_APPLY_COOP", 1) != 0))) {
        const int rc = launch_cell_coop<T, I, NC, J2X2>(a);
        if (rc != AF_ENOTSUP) return rc;
    }
    // ... and any other call with DIE terms but neither DDE terms nor coherencies: the tile kernel
    // with no sources -- the per-lane gain gathers were what held these calls at 3.3 TB/s
    if (!ddes && !coh && a.die1 != nullptr && a.die1 == a.die2 && env_int("AFHIP_PREDICT_TILE", 1) != 0 &&
        env_int("AFHIP_PREDICT_DIE_LDS", 1) != 0 && (a.nchan * NC * 2 * sizeof(T)) % 16 == 0 && ((uintptr_t)a.die1 & 15) == 0 &&
        a.nchan >= 4 && a.nrow * a.nchan >= ((int64_t)1 << 16)) {
        constexpr int CT = (NC * 2 * sizeof(T) >= 64) ? 4 : (NC * 2 * sizeof(T) >= 32) ? 8 : 16;
        // one LDS stage leaves room for wider tiles: 2 CT measured best (3.73 / 3.93 / 3.82 TB/s at CT / 2 CT / 4 CT for
        // 2x2 complex128, tools/bench_apply_gains.py; AFHIP_APPLY_CT is the measurement hook)
        const int want = env_int("AFHIP_APPLY_CT", 2 * CT);
        int rc = AF_ENOTSUP;
        if (want == 4 * CT && a.nchan >= 4 * CT) rc = launch_tile<T, I, NC, J2X2, false, 4 * CT, 512, 1>(a);
        else if (want == 2 * CT && a.nchan >= 2 * CT) rc = launch_tile<T, I, NC, J2X2, false, 2 * CT, 512, 1>(a);
        if (rc == AF_ENOTSUP) rc = launch_tile<T, I, NC, J2X2, false, CT, 512, 1>(a);
        if (rc != AF_ENOTSUP) return rc;
    }
    if (ddes && coh) return launch<T, I, NC, J2X2, true, true>(a);
    if (ddes) return launch<T, I, NC, J2X2, true, false>(a);
    if (coh) return launch<T, I, NC, J2X2, false, true>(a);
    return launch<T, I, NC, J2X2, false, false>(a);
}

template <typename T, typename I>
int launch_layout(const PArgs &a, int ncorr, int jones_kind)
{
    if (jones_kind == AF_JONES_2X2) return launch_presence<T, I, 4, true>(a);
    switch (ncorr) {
    case 1: return launch_presence<T, I, 1, false>(a);
    case 2: return launch_presence<T, I, 2, false>(a);
    default: return launch_presence<T, I, 4, false>(a);
    }
}

template <typename T>
int predict_vis(const void *time_index, const void *ant1, const void *ant2, int index_bytes, int64_t nrow,
                const T *dde1, const T *coh, const T *dde2, const T *die1, const T *bvis, const T *die2,
                int64_t nsrc, int64_t ntime, int64_t nant, int64_t nchan, int ncorr, int jones_kind, T *out,
                void *workspace, size_t workspace_bytes, void *stream)
{
    AF_REQUIRE((dde1 != nullptr) == (dde2 != nullptr), "Both dde1_jones and dde2_jones must be present or absent");
    AF_REQUIRE((die1 != nullptr) == (die2 != nullptr), "Both die1_jones and die2_jones must be present or absent");
    AF_REQUIRE(dde1 || coh || die1 || bvis, "No Jones Matrices were supplied");
    AF_REQUIRE(jones_kind == AF_JONES_DIAG || jones_kind == AF_JONES_2X2, "af_predict_vis: bad jones_kind %d",
               jones_kind);
    AF_REQUIRE(ncorr == 1 || ncorr == 2 || ncorr == 4, "af_predict_vis: ncorr %d not in (1, 2, 4)", ncorr);
    AF_REQUIRE(jones_kind != AF_JONES_2X2 || ncorr == 4, "af_predict_vis: 2x2 Jones need ncorr == 4");
    AF_REQUIRE(index_bytes == 4 || index_bytes == 8, "af_predict_vis: index_bytes %d not in (4, 8)", index_bytes);
    AF_REQUIRE(nrow >= 0 && nsrc >= 0 && ntime >= 0 && nant >= 0 && nchan >= 0, "af_predict_vis: negative extent");
    if (nrow == 0 || nchan == 0) return AF_OK;
    AF_REQUIRE(out != nullptr, "af_predict_vis: out is NULL");
    AF_REQUIRE(time_index && ant1 && ant2, "af_predict_vis: NULL index array");
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= 16, "af_predict_vis: workspace too small");
    hipStream_t st = af_stream(stream);
    long long *tmin = static_cast<long long *>(workspace);
    int *status = reinterpret_cast<int *>(tmin + 1);
    // tmin <- INT64_MAX (0x7f7f... is large enough and byte-settable), then device-side min; status <- 0
    AF_HIP(hipMemsetAsync(tmin, 0x7f, sizeof(long long), st));
    AF_HIP(hipMemsetAsync(status, 0, 8, st));
    if (dde1 != nullptr || die1 != nullptr) {
        int64_t blocks = af_cdiv(nrow, 256 * 8);
        if (blocks > 1024) blocks = 1024;
        if (index_bytes == 4)
            hipLaunchKernelGGL((time_min_kernel<int32_t>), dim3((unsigned)blocks), dim3(256), 0, st,
                               (const int32_t *)time_index, nrow, tmin);
        else
            hipLaunchKernelGGL((time_min_kernel<int64_t>), dim3((unsigned)blocks), dim3(256), 0, st,
                               (const int64_t *)time_index, nrow, tmin);
        AF_LAUNCH_CHECK();
    }
    PArgs a;
    a.time_index = time_index; a.ant1 = ant1; a.ant2 = ant2; a.index_bytes = index_bytes; a.nrow = nrow;
    a.dde1 = dde1; a.coh = coh; a.dde2 = dde2; a.die1 = die1; a.bvis = bvis; a.die2 = die2;
    a.nsrc = (dde1 || coh) ? nsrc : 0; a.ntime = ntime; a.nant = nant; a.nchan = nchan;
    a.tmin = tmin; a.status = status; a.out = out; a.st = st;
    return index_bytes == 4 ? launch_layout<T, int32_t>(a, ncorr, jones_kind)
                            : launch_layout<T, int64_t>(a, ncorr, jones_kind);
}

}  // namespace

AF_EXPORT size_t af_predict_vis_workspace_bytes(void) { return 256; }

AF_EXPORT int af_predict_vis_c128(const void *time_index, const void *antenna1, const void *antenna2,
                                  int index_bytes, int64_t nrow, const double *dde1_jones,
                                  const double *source_coh, const double *dde2_jones, const double *die1_jones,
                                  const double *base_vis, const double *die2_jones, int64_t nsrc, int64_t ntime,
                                  int64_t nant, int64_t nchan, int ncorr, int jones_kind, double *out,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    return predict_vis<double>(time_index, antenna1, antenna2, index_bytes, nrow, dde1_jones, source_coh,
                               dde2_jones, die1_jones, base_vis, die2_jones, nsrc, ntime, nant, nchan, ncorr,
                               jones_kind, out, workspace, workspace_bytes, stream);
}

AF_EXPORT int af_predict_vis_c64(const void *time_index, const void *antenna1, const void *antenna2,
                                 int index_bytes, int64_t nrow, const float *dde1_jones, const float *source_coh,
                                 const float *dde2_jones, const float *die1_jones, const float *base_vis,
                                 const float *die2_jones, int64_t nsrc, int64_t ntime, int64_t nant,
                                 int64_t nchan, int ncorr, int jones_kind, float *out, void *workspace,
                                 size_t workspace_bytes, void *stream)
{
    return predict_vis<float>(time_index, antenna1, antenna2, index_bytes, nrow, dde1_jones, source_coh,
                              dde2_jones, die1_jones, base_vis, die2_jones, nsrc, ntime, nant, nchan, ncorr,
                              jones_kind, out, workspace, workspace_bytes, stream);
}
