// Beam-cube direction-dependent effects for gfx950.
//
// Replaces africanus/rime/fast_beam_cubes.py:
//   freq_grid_interp :10-54   (binary search of each channel in beam_freq_map)
//   beam_cube_dde    :57-240  (rotate/scale lm, trilinear 8-voxel interpolation of the complex
//                              cube AND of its amplitude, amplitude-preserving normalisation)
// One lane owns one output Jones (source, time, antenna, channel) with the channel axis
// fastest, so stores are coalesced; the cube gathers hit L2/Infinity Cache (the whole cube
// is read-only and resident).  Arithmetic keeps the reference's operation order with
// explicitly rounded operations (no contraction).
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

// fast_beam_cubes.py:110-238; grid: ceil(nsrc*ntime*nant*nchan / 256)
template <typename T>
__global__ __launch_bounds__(256) void beam_cube_dde_kernel(
    const T *__restrict__ beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud, int ncorr,
    const T *__restrict__ lm_ext, const T *__restrict__ lm, int64_t nsrc, const T *__restrict__ parangles,
    int64_t ntime, int64_t nant, const T *__restrict__ point_errors, const T *__restrict__ antenna_scaling,
    const T *__restrict__ freq_data, int64_t nchan, T *__restrict__ out)
{
    using O = B<T>;
    using V2 = typename O::vec2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = nsrc * ntime * nant * nchan;
    if (idx >= total) return;
    const int64_t f = idx % nchan;
    const int64_t a = (idx / nchan) % nant;
    const int64_t t = (idx / (nchan * nant)) % ntime;
    const int64_t s = idx / (nchan * nant * ntime);

    const BeamGrid<T> grid = beam_grid<T>(lm_ext, beam_lw, beam_mh, beam_nud);
    T sin_pa, cos_pa;
    O::sincos_(parangles[t * nant + a], &sin_pa, &cos_pa);
    const T *pe = point_errors + ((t * nant + a) * nchan + f) * 2;
    const T *as = antenna_scaling + (a * nchan + f) * 2;
    BeamVoxels<T> vx;
    beam_voxels<T, int64_t>(grid, lm[2 * s], lm[2 * s + 1], sin_pa, cos_pa, pe[0], pe[1], as[0], as[1], freq_data[3 * f + 0],
                   freq_data[3 * f + 1], (int)freq_data[3 * f + 2], ncorr, vx);
    const V2 *fbeam = reinterpret_cast<const V2 *>(beam);
    V2 *o = reinterpret_cast<V2 *>(out) + idx * ncorr;
    for (int c = 0; c < ncorr; ++c) o[c] = beam_sample_corr<T, int64_t, false>(fbeam, nullptr, vx, c);
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
int beam_cube_dde(const T *beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud, int ncorr,
                  const T *beam_lm_extents, const T *beam_freq_map, const T *lm, int64_t nsrc,
                  const T *parallactic_angles, int64_t ntime, int64_t nant, const T *point_errors,
                  const T *antenna_scaling, const T *frequency, int64_t nchan, T *out, T *freq_data_ws,
                  void *stream)
{
    AF_REQUIRE(beam_lw >= 2 && beam_mh >= 2 && beam_nud >= 2, "beam_lw, beam_mh and beam_nud must be >= 2");
    AF_REQUIRE(ncorr >= 1, "af_beam_cube_dde: ncorr must be >= 1");
    AF_REQUIRE(nsrc >= 0 && ntime >= 0 && nant >= 0 && nchan >= 0, "af_beam_cube_dde: negative extent");
    const int64_t total = nsrc * ntime * nant * nchan;
    if (total == 0) return AF_OK;
    AF_REQUIRE(beam && beam_lm_extents && beam_freq_map && lm && parallactic_angles && point_errors &&
                   antenna_scaling && frequency && out && freq_data_ws,
               "af_beam_cube_dde: NULL array");
    int rc = freq_grid_interp<T>(frequency, nchan, beam_freq_map, beam_nud, freq_data_ws, stream);
    if (rc != AF_OK) return rc;
    const int64_t blocks = af_cdiv(total, 256);
    AF_REQUIRE(blocks < (1LL << 31), "af_beam_cube_dde: problem too large for one launch");
    hipLaunchKernelGGL((beam_cube_dde_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, af_stream(stream), beam,
                       beam_lw, beam_mh, beam_nud, ncorr, beam_lm_extents, lm, nsrc, parallactic_angles, ntime,
                       nant, point_errors, antenna_scaling, freq_data_ws, nchan, out);
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
                                    double *out, double *freq_data_ws, void *stream)
{
    return beam_cube_dde<double>(beam, beam_lw, beam_mh, beam_nud, ncorr, beam_lm_extents, beam_freq_map, lm, nsrc,
                                 parallactic_angles, ntime, nant, point_errors, antenna_scaling, frequency, nchan,
                                 out, freq_data_ws, stream);
}

AF_EXPORT int af_beam_cube_dde_c64(const float *beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud,
                                   int ncorr, const float *beam_lm_extents, const float *beam_freq_map,
                                   const float *lm, int64_t nsrc, const float *parallactic_angles, int64_t ntime,
                                   int64_t nant, const float *point_errors, const float *antenna_scaling,
                                   const float *frequency, int64_t nchan, float *out, float *freq_data_ws,
                                   void *stream)
{
    return beam_cube_dde<float>(beam, beam_lw, beam_mh, beam_nud, ncorr, beam_lm_extents, beam_freq_map, lm, nsrc,
                                parallactic_angles, ntime, nant, point_errors, antenna_scaling, frequency, nchan,
                                out, freq_data_ws, stream);
}
