// Beam-cube direction-dependent effects for gfx950.
//
// Replaces africanus/rime/fast_beam_cubes.py:
//   freq_grid_interp :10-54   (binary search of each channel in beam_freq_map)
//   beam_cube_dde    :57-240  (rotate/scale lm, trilinear 8-voxel interpolation of the complex
//                              cube AND of its amplitude, amplitude-preserving normalisation)
// One lane owns one correlation of one output Jones (source, time, antenna, channel), channel and
// correlation fastest, so stores are coalesced and the lanes of a Jones gather adjacent records; the
// cube and its amplitude (computed once per call: the reference takes |.| of every gathered voxel) are
// read-only and L2/Infinity-Cache resident.  Arithmetic keeps the reference's operation order with
// explicitly rounded operations (no contraction).
#include <stdlib.h>

#include "af_common.h"
#include "af_beam_device.h"

namespace {

template <typename T> using B = BeamOps<T>;

// fast_beam_cubes.py:10-54
template <typename T>
__global__ void freq_grid_interp_kernel(const T *__restrict__ frequency, int64_t nchan,
                                        const T *__restrict__ beam_freq_map, int64_t beam_nud,
                                        T *__restrict__ freq_data)
{
    using O = B<T>;
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nchan) return;
    const T freq = frequency[f];
    int64_t lower = 0, upper = beam_nud - 1;
    while (lower <= upper) {
        int64_t mid = lower + (upper - lower) / 2;
        T beam_freq = beam_freq_map[mid];
        if (beam_freq < freq) lower = mid + 1;
        else if (beam_freq > freq) upper = mid - 1;
        else { lower = mid; break; }
    }
    lower = lower < upper ? lower : upper;
    upper = lower + 1;
    T scale, weight, pos;
    if (lower == -1) {
        scale = O::div(freq, beam_freq_map[0]); weight = (T)1.0; pos = (T)0.0;
    } else if (upper == beam_nud) {
        scale = O::div(freq, beam_freq_map[beam_nud - 1]); weight = (T)0.0; pos = (T)(beam_nud - 2);
    } else {
        T freq_low = beam_freq_map[lower], freq_high = beam_freq_map[upper];
        scale = (T)1.0;
        weight = O::div(O::sub(freq_high, freq), O::sub(freq_high, freq_low));
        pos = (T)lower;
    }
    freq_data[3 * f + 0] = scale;
    freq_data[3 * f + 1] = weight;
    freq_data[3 * f + 2] = pos;
}

// One record (re, im, |.|, 0) per (voxel, correlation), built once per call: the reference takes np.abs of
// every gathered voxel (fast_beam_cubes.py:187-222: 8 x ncorr hypot per Jones), the cube has far fewer voxels than
// the call has samples, and a sample then needs ONE 4-element gather per voxel instead of two.
template <typename T>
__global__ void beam_pack_kernel(const typename BeamOps<T>::vec2 *__restrict__ beam, int64_t n, T *__restrict__ rec)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const typename BeamOps<T>::vec2 b = beam[i];
        rec[4 * i + 0] = b.x;
        rec[4 * i + 1] = b.y;
        rec[4 * i + 2] = BeamOps<T>::hypot_(b.x, b.y);
        rec[4 * i + 3] = (T)0.0;
    }
}

// beam_sample_corr (af_beam_device.h) reading the packed records: same operations in the same order
template <typename T, typename I>
__device__ __forceinline__ typename BeamOps<T>::vec2 beam_sample_rec(const T *__restrict__ rec,
                                                                     const BeamVoxels<T, I> &vx, int c)
{
    using O = BeamOps<T>;
    using V2 = typename O::vec2;
    struct alignas(4 * sizeof(T)) V4 { T x, y, z, w; };
    const T zero = (T)0.0;
    V4 b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = *reinterpret_cast<const V4 *>(rec + 4 * (vx.off[k] + c));
    T cre = zero, cim = zero, absc = zero;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const T wgt = vx.wt[k];
        absc = O::add(absc, O::mul(wgt, b[k].z));
        const T pre = O::sub(O::mul(wgt, b[k].x), O::mul(zero, b[k].y));
        const T pim = O::add(O::mul(wgt, b[k].y), O::mul(zero, b[k].x));
        cre = O::add(cre, pre);
        cim = O::add(cim, pim);
    }
    const T div = O::hypot_(cre, cim);
    const T sc = (div == zero) ? absc : O::div(absc, div);
    V2 r;
    r.x = O::sub(O::mul(cre, sc), O::mul(cim, zero));
    r.y = O::add(O::mul(cre, zero), O::mul(cim, sc));
    return r;
}

// (sin, cos) of every parallactic angle once per call (:118-119)
template <typename T>
__global__ void beam_parangle_kernel(const T *__restrict__ parangles, int64_t n, T *__restrict__ sc)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) BeamOps<T>::sincos_(parangles[i], &sc[2 * i], &sc[2 * i + 1]);
}

// fast_beam_cubes.py:110-238; one lane per (Jones, correlation): grid ceil(nsrc*ntime*nant*nchan*ncorr / 256).
// The gathers are request-rate bound: the ncorr lanes of a Jones read ncorr adjacent 32-byte records of a
// voxel (one cache line at ncorr = 4) and store adjacent outputs; the voxel set-up is recomputed per lane
// (cheaper than exchanging 16 values).
template <typename T>
__global__ __launch_bounds__(256) void beam_cube_dde_kernel(
    const T *__restrict__ vrec, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud, int ncorr,
    const T *__restrict__ lm_ext, const T *__restrict__ lm, int64_t nsrc, const T *__restrict__ pa_sc,
    int64_t ntime, int64_t nant, const T *__restrict__ point_errors, const T *__restrict__ antenna_scaling,
    const T *__restrict__ freq_data, int64_t nchan, T *__restrict__ out)
{
    using O = B<T>;
    using V2 = typename O::vec2;
    const int64_t lane_idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = nsrc * ntime * nant * nchan;
    const int64_t idx = lane_idx / ncorr;
    const int c = (int)(lane_idx - idx * ncorr);
    if (idx >= total) return;
    const int64_t f = idx % nchan;
    const int64_t a = (idx / nchan) % nant;
    const int64_t t = (idx / (nchan * nant)) % ntime;
    const int64_t s = idx / (nchan * nant * ntime);

    const BeamGrid<T> grid = beam_grid<T>(lm_ext, beam_lw, beam_mh, beam_nud);
    const T sin_pa = pa_sc[2 * (t * nant + a)], cos_pa = pa_sc[2 * (t * nant + a) + 1];
    const T *pe = point_errors + ((t * nant + a) * nchan + f) * 2;
    const T *as = antenna_scaling + (a * nchan + f) * 2;
    BeamVoxels<T> vx;
    beam_voxels<T, int64_t>(grid, lm[2 * s], lm[2 * s + 1], sin_pa, cos_pa, pe[0], pe[1], as[0], as[1], freq_data[3 * f + 0],
                   freq_data[3 * f + 1], (int)freq_data[3 * f + 2], ncorr, vx);
    reinterpret_cast<V2 *>(out)[lane_idx] = beam_sample_rec<T, int64_t>(vrec, vx, c);
}

// AFHIP_BEAM_BLOCK=0: the flat-index kernel above (measurement hook, read per call)
inline bool beam_block_enabled()
{
    const char *v = getenv("AFHIP_BEAM_BLOCK");
    return !(v && v[0] == '0');
}

// The cube-wide constants (two divisions) once per call instead of once per lane.
template <typename T>
__global__ void beam_grid_kernel(const T *__restrict__ lm_ext, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud,
                                 BeamGrid<T> *__restrict__ grid)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *grid = beam_grid<T>(lm_ext, beam_lw, beam_mh, beam_nud);
}

// The same sampling with the index arithmetic taken out of the lanes (round 3).  The kernel above spends ~650 of its
// ~850 vector instructions per lane on integers: four 64-bit divisions to split the flat index into (source, time,
// antenna, chan, corr), 64-bit voxel offsets, and the two divisions of the cube constants.  Here the grid is
// (antenna x chan chunks, time, source): the block's (s, t) -- and a, when an antenna fills a block -- are scalars, a
// lane's (chan, corr) a shift and a mask
// (NCL = log2(ncorr), or -1: one 32-bit division), the voxel offsets 32-bit (I) and the constants read from `grid`.
// Same floating-point operations in the same order: bit-identical to the kernel above.
template <typename T, typename I, int NCL>
__global__ __launch_bounds__(256) void beam_cube_dde_block_kernel(
    const T *__restrict__ vrec, const BeamGrid<T> *__restrict__ gridp, int ncorr, const T *__restrict__ lm,
    const T *__restrict__ pa_sc, int ntime, int nant, const T *__restrict__ point_errors,
    const T *__restrict__ antenna_scaling, const T *__restrict__ freq_data, int nchan, int chunks, T *__restrict__ out)
{
    using V2 = typename B<T>::vec2;
    const int lanes = nchan * ncorr;                       // per (source, time, antenna)
    int a, j;
    if (chunks > 0) {                                      // `chunks` blocks of 256 lanes per antenna
        a = (int)blockIdx.x;
        int chunk = 0;
        if (chunks > 1) { a = (int)(blockIdx.x / (unsigned)chunks); chunk = (int)blockIdx.x - a * chunks; }
        j = chunk * 256 + (int)threadIdx.x;
    } else {                                               // 2^-chunks lanes per antenna, 256 >> -chunks antennas per block
        const int sh = -chunks;
        a = (int)blockIdx.x * (256 >> sh) + ((int)threadIdx.x >> sh);
        j = (int)threadIdx.x & ((1 << sh) - 1);
    }
    const int t = (int)blockIdx.y, s = (int)blockIdx.z;
    if (j >= lanes || a >= nant) return;
    int f, c;
    if constexpr (NCL >= 0) { f = j >> NCL; c = j & ((1 << NCL) - 1); }
    else { f = (int)((unsigned)j / (unsigned)ncorr); c = j - f * ncorr; }
    const BeamGrid<T> grid = *gridp;
    const int64_t ta = (int64_t)t * nant + a;
    const T sin_pa = pa_sc[2 * ta], cos_pa = pa_sc[2 * ta + 1];
    const T *pe = point_errors + (ta * nchan + f) * 2;
    const T *as = antenna_scaling + ((int64_t)a * nchan + f) * 2;
    BeamVoxels<T, I> vx;
    beam_voxels<T, I, I>(grid, lm[2 * s], lm[2 * s + 1], sin_pa, cos_pa, pe[0], pe[1], as[0], as[1], freq_data[3 * f + 0],
                         freq_data[3 * f + 1], (int)freq_data[3 * f + 2], ncorr, vx);
    const int64_t base = (((int64_t)s * ntime + t) * nant + a) * lanes;
    reinterpret_cast<V2 *>(out)[base + j] = beam_sample_rec<T, I>(vrec, vx, c);
}

template <typename T>
int freq_grid_interp(const T *frequency, int64_t nchan, const T *beam_freq_map, int64_t beam_nud, T *freq_data,
                     void *stream)
{
    AF_REQUIRE(nchan >= 0 && beam_nud >= 1, "af_freq_grid_interp: bad extents");
    if (nchan == 0) return AF_OK;
    AF_REQUIRE(frequency && beam_freq_map && freq_data, "af_freq_grid_interp: NULL array");
    hipLaunchKernelGGL((freq_grid_interp_kernel<T>), dim3((unsigned)af_cdiv(nchan, 64)), dim3(64), 0,
                       af_stream(stream), frequency, nchan, beam_freq_map, beam_nud, freq_data);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

template <typename T>
struct BeamWs {
    size_t freq_data, pa_sc, babs, grid, total;
};

template <typename T>
BeamWs<T> beam_ws(int64_t beam_lw, int64_t beam_mh, int64_t beam_nud, int64_t ncorr, int64_t ntime, int64_t nant,
                  int64_t nchan)
{
    BeamWs<T> w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    w.freq_data = take((size_t)nchan * 3 * sizeof(T));
    w.pa_sc = take((size_t)ntime * nant * 2 * sizeof(T));
    w.babs = take((size_t)beam_lw * beam_mh * beam_nud * ncorr * 4 * sizeof(T));  // (re, im, |.|, 0) records
    w.grid = take(sizeof(BeamGrid<T>));
    w.total = o;
    return w;
}

template <typename T>
int beam_cube_dde(const T *beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud, int ncorr,
                  const T *beam_lm_extents, const T *beam_freq_map, const T *lm, int64_t nsrc,
                  const T *parallactic_angles, int64_t ntime, int64_t nant, const T *point_errors,
                  const T *antenna_scaling, const T *frequency, int64_t nchan, T *out, void *workspace,
                  size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(beam_lw >= 2 && beam_mh >= 2 && beam_nud >= 2, "beam_lw, beam_mh and beam_nud must be >= 2");
    AF_REQUIRE(ncorr >= 1, "af_beam_cube_dde: ncorr must be >= 1");
    AF_REQUIRE(nsrc >= 0 && ntime >= 0 && nant >= 0 && nchan >= 0, "af_beam_cube_dde: negative extent");
    const int64_t total = nsrc * ntime * nant * nchan;
    if (total == 0) return AF_OK;
    AF_REQUIRE(beam && beam_lm_extents && beam_freq_map && lm && parallactic_angles && point_errors &&
                   antenna_scaling && frequency && out,
               "af_beam_cube_dde: NULL array");
    const BeamWs<T> W = beam_ws<T>(beam_lw, beam_mh, beam_nud, ncorr, ntime, nant, nchan);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= W.total, "af_beam_cube_dde: workspace too small (%zu < %zu)",
               workspace_bytes, W.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_beam_cube_dde: workspace must be 256-byte aligned");
    char *ws = static_cast<char *>(workspace);
    T *freq_data = reinterpret_cast<T *>(ws + W.freq_data);
    T *pa_sc = reinterpret_cast<T *>(ws + W.pa_sc);
    T *babs = reinterpret_cast<T *>(ws + W.babs);
    hipStream_t st = af_stream(stream);
    int rc = freq_grid_interp<T>(frequency, nchan, beam_freq_map, beam_nud, freq_data, stream);
    if (rc != AF_OK) return rc;
    const int64_t nvox = beam_lw * beam_mh * beam_nud * ncorr;
    int64_t ablocks = af_cdiv(nvox, 256);
    if (ablocks > 8192) ablocks = 8192;
    hipLaunchKernelGGL((beam_pack_kernel<T>), dim3((unsigned)ablocks), dim3(256), 0, st,
                       reinterpret_cast<const typename BeamOps<T>::vec2 *>(beam), nvox, babs);
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL((beam_parangle_kernel<T>), dim3((unsigned)af_cdiv(ntime * nant, 256)), dim3(256), 0, st,
                       parallactic_angles, ntime * nant, pa_sc);
    AF_LAUNCH_CHECK();
    // block-indexed kernel: (source, time, antenna) from the grid, 32-bit offsets into the records
    const int64_t lanes = nchan * ncorr;
    if (beam_block_enabled() && ntime <= 65535 && nsrc <= 65535 && lanes < (1LL << 24) &&
        nvox * 4 < (1LL << 31) && nant < (1LL << 20)) {
        // >= 129 lanes per antenna: whole 256-lane blocks per antenna; fewer: a power-of-two slot per antenna, several
        // antennas per block
        int64_t chunks = af_cdiv(lanes, 256), xblocks = nant * chunks;
        if (lanes <= 128) {
            int sh = 0;
            while ((1 << sh) < lanes) ++sh;
            chunks = -sh;
            xblocks = af_cdiv(nant, 256 >> sh);
        }
        if (xblocks < (1LL << 31)) {
            BeamGrid<T> *gridp = reinterpret_cast<BeamGrid<T> *>(ws + W.grid);
            hipLaunchKernelGGL((beam_grid_kernel<T>), dim3(1), dim3(64), 0, st, beam_lm_extents, beam_lw, beam_mh, beam_nud, gridp);
            AF_LAUNCH_CHECK();
            const dim3 g((unsigned)xblocks, (unsigned)ntime, (unsigned)nsrc), b(256);
#define AF_BEAM_LAUNCH(NCL)                                                                                            \
    hipLaunchKernelGGL((beam_cube_dde_block_kernel<T, int, NCL>), g, b, 0, st, babs, gridp, ncorr, lm, pa_sc, (int)ntime,    \
                       (int)nant, point_errors, antenna_scaling, freq_data, (int)nchan, (int)chunks, out)
            if (ncorr == 4) AF_BEAM_LAUNCH(2);
            else if (ncorr == 2) AF_BEAM_LAUNCH(1);
            else if (ncorr == 1) AF_BEAM_LAUNCH(0);
            else AF_BEAM_LAUNCH(-1);
#undef AF_BEAM_LAUNCH
            AF_LAUNCH_CHECK();
            return AF_OK;
        }
    }
    const int64_t blocks = af_cdiv(total * ncorr, 256);
    AF_REQUIRE(blocks < (1LL << 31), "af_beam_cube_dde: problem too large for one launch");
    hipLaunchKernelGGL((beam_cube_dde_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, babs, beam_lw, beam_mh,
                       beam_nud, ncorr, beam_lm_extents, lm, nsrc, pa_sc, ntime, nant, point_errors, antenna_scaling,
                       freq_data, nchan, out);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

}  // namespace

AF_EXPORT int af_freq_grid_interp_f64(const double *frequency, int64_t nchan, const double *beam_freq_map,
                                      int64_t beam_nud, double *freq_data, void *stream)
{
    return freq_grid_interp<double>(frequency, nchan, beam_freq_map, beam_nud, freq_data, stream);
}

AF_EXPORT int af_freq_grid_interp_f32(const float *frequency, int64_t nchan, const float *beam_freq_map,
                                      int64_t beam_nud, float *freq_data, void *stream)
{
    return freq_grid_interp<float>(frequency, nchan, beam_freq_map, beam_nud, freq_data, stream);
}

AF_EXPORT int af_beam_cube_dde_c128(const double *beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud,
                                    int ncorr, const double *beam_lm_extents, const double *beam_freq_map,
                                    const double *lm, int64_t nsrc, const double *parallactic_angles,
                                    int64_t ntime, int64_t nant, const double *point_errors,
                                    const double *antenna_scaling, const double *frequency, int64_t nchan,
                                    double *out, void *workspace, size_t workspace_bytes, void *stream)
{
    return beam_cube_dde<double>(beam, beam_lw, beam_mh, beam_nud, ncorr, beam_lm_extents, beam_freq_map, lm, nsrc,
                                 parallactic_angles, ntime, nant, point_errors, antenna_scaling, frequency, nchan,
                                 out, workspace, workspace_bytes, stream);
}

AF_EXPORT size_t af_beam_cube_dde_workspace_bytes(int64_t beam_lw, int64_t beam_mh, int64_t beam_nud, int64_t ncorr,
                                                  int64_t ntime, int64_t nant, int64_t nchan, int is_f32)
{
    if (beam_lw < 0 || beam_mh < 0 || beam_nud < 0 || ncorr < 0 || ntime < 0 || nant < 0 || nchan < 0) return 0;
    return is_f32 ? beam_ws<float>(beam_lw, beam_mh, beam_nud, ncorr, ntime, nant, nchan).total
                  : beam_ws<double>(beam_lw, beam_mh, beam_nud, ncorr, ntime, nant, nchan).total;
}

AF_EXPORT int af_beam_cube_dde_c64(const float *beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud,
                                   int ncorr, const float *beam_lm_extents, const float *beam_freq_map,
                                   const float *lm, int64_t nsrc, const float *parallactic_angles, int64_t ntime,
                                   int64_t nant, const float *point_errors, const float *antenna_scaling,
                                   const float *frequency, int64_t nchan, float *out, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    return beam_cube_dde<float>(beam, beam_lw, beam_mh, beam_nud, ncorr, beam_lm_extents, beam_freq_map, lm, nsrc,
                                parallactic_angles, ntime, nant, point_errors, antenna_scaling, frequency, nchan,
                                out, workspace, workspace_bytes, stream);
}
