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
#include "af_common.h"

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
    int64_t ntime, int64_t nant, int64_t nchan, const long long *__restrict__ tmin_p, T *__restrict__ out)
{
    const int64_t cell = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int64_t ncell = nrow * nchan;
    if (cell >= ncell) return;
    const int64_t r = cell / nchan, f = cell - r * nchan;
    const bool have_dies = die1 != nullptr;
    int64_t ti = 0, a1 = 0, a2 = 0;
    if (HAVE_DDES || have_dies) {
        ti = (int64_t)time_index[r] - (int64_t)(*tmin_p);
        a1 = (int64_t)ant1[r];
        a2 = (int64_t)ant2[r];
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
    typename R<T>::vec2 *o = reinterpret_cast<typename R<T>::vec2 *>(out + cell * CS);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        typename R<T>::vec2 v;
        v.x = acc[c].re;
        v.y = acc[c].im;
        o[c] = v;
    }
}

struct PArgs {
    const void *time_index, *ant1, *ant2;
    int index_bytes;
    int64_t nrow;
    const void *dde1, *coh, *dde2, *die1, *bvis, *die2;
    int64_t nsrc, ntime, nant, nchan;
    long long *tmin;
    void *out;
    hipStream_t st;
};

template <typename T, typename I, int NC, bool J2X2, bool HAVE_DDES, bool HAVE_COH>
int launch(const PArgs &a)
{
    const int64_t blocks = af_cdiv(a.nrow * a.nchan, THREADS);
    AF_REQUIRE(blocks < (1LL << 31), "af_predict_vis: problem too large for one launch");
    hipLaunchKernelGGL((predict_vis_kernel<T, I, NC, J2X2, HAVE_DDES, HAVE_COH>), dim3((unsigned)blocks),
                       dim3(THREADS), 0, a.st, (const I *)a.time_index, (const I *)a.ant1, (const I *)a.ant2,
                       a.nrow, (const T *)a.dde1, (const T *)a.coh, (const T *)a.dde2, (const T *)a.die1,
                       (const T *)a.bvis, (const T *)a.die2, a.nsrc, a.ntime, a.nant, a.nchan, a.tmin,
                       (T *)a.out);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

template <typename T, typename I, int NC, bool J2X2>
int launch_presence(const PArgs &a)
{
    const bool ddes = a.dde1 != nullptr, coh = a.coh != nullptr;
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
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= sizeof(long long),
               "af_predict_vis: workspace too small");
    hipStream_t st = af_stream(stream);
    long long *tmin = static_cast<long long *>(workspace);
    // tmin <- INT64_MAX (0x7f7f... is large enough and byte-settable), then device-side min
    AF_HIP(hipMemsetAsync(tmin, 0x7f, sizeof(long long), st));
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
    a.tmin = tmin; a.out = out; a.st = st;
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
