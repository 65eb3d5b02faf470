// Term producers of the RIME chain for gfx950: small element-wise kernels whose outputs
// feed predict_vis / the fused predict.
//   feed_rotation   africanus/rime/feeds.py:14-73
//   gaussian shape  africanus/model/shape/gaussian_shape.py:11-62
#include "af_common.h"
#include "af_sincos.h"

namespace {

// linear: [[c, s], [-s, c]]; circular: diag(e^{-i pa}, e^{+i pa})  (feeds.py:21-45)
template <typename T, typename T2>
__global__ void feed_rotation_kernel(const T *__restrict__ pa, int64_t n, int feed_type, T2 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T s, c;
    if constexpr (sizeof(T) == 8) sincos(pa[i], &s, &c);
    else sincosf(pa[i], &s, &c);
    T2 *o = out + 4 * i;
    const T z = (T)0;
    if (feed_type == 0) {
        o[0].x = c; o[0].y = z; o[1].x = s; o[1].y = z;
        o[2].x = -s; o[2].y = z; o[3].x = c; o[3].y = z;
    } else {
        o[0].x = c; o[0].y = -s; o[1].x = z; o[1].y = z;
        o[2].x = z; o[2].y = z; o[3].x = c; o[3].y = s;
    }
}

// per source: el = emaj sin(angle), em = emaj cos(angle), er = emin / (emaj or 1)  (gaussian_shape.py:45-50)
__global__ void gauss_params_kernel(const double *__restrict__ shape_params, int64_t nsrc, double *__restrict__ p)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    const double emaj = shape_params[3 * s], emin = shape_params[3 * s + 1], angle = shape_params[3 * s + 2];
    p[4 * s + 0] = __dmul_rn(emaj, sin(angle));
    p[4 * s + 1] = __dmul_rn(emaj, cos(angle));
    p[4 * s + 2] = emin / (emaj == 0.0 ? 1.0 : emaj);
    p[4 * s + 3] = 0.0;
}

// shape[s,r,f] = exp(-(fu1^2 + fv1^2)) in the reference's operation order (:52-60); one lane per (s,r,f),
// channel fastest (coalesced 8-byte stores)
__global__ __launch_bounds__(256) void gaussian_shape_kernel(const double *__restrict__ uvw,
                                                             const double *__restrict__ frequency,
                                                             const double *__restrict__ p, int64_t nsrc, int64_t nrow,
                                                             int64_t nchan, double gauss_scale, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nsrc * nrow * nchan) return;
    const int64_t f = i % nchan, r = (i / nchan) % nrow, s = i / (nchan * nrow);
    const double el = p[4 * s], em = p[4 * s + 1], er = p[4 * s + 2];
    const double u = uvw[3 * r], v = uvw[3 * r + 1];
    const double u1 = __dmul_rn(__dsub_rn(__dmul_rn(u, em), __dmul_rn(v, el)), er);
    const double v1 = __dadd_rn(__dmul_rn(u, el), __dmul_rn(v, em));
    const double sf = __dmul_rn(frequency[f], gauss_scale);
    const double fu1 = __dmul_rn(u1, sf), fv1 = __dmul_rn(v1, sf);
    out[i] = exp(-__dadd_rn(__dmul_rn(fu1, fu1), __dmul_rn(fv1, fv1)));
}

// float ** int as numba lowers it: exponentiation by squaring
__device__ __forceinline__ double ipow_sq(double a, int e)
{
    double r = 1.0;
    while (e != 0) {
        if (e & 1) r = __dmul_rn(r, a);
        e >>= 1;
        a = __dmul_rn(a, a);
    }
    return r;
}

// spectral_model (spec_model.py:173-213): one lane per (source, chan, pol), pol fastest
__global__ void spectral_model_kernel(const double *__restrict__ stokes, const double *__restrict__ spi,
                                      const double *__restrict__ ref_freq, const double *__restrict__ frequency,
                                      const int *__restrict__ base, int64_t nsrc, int64_t nspi, int64_t npol,
                                      int64_t nchan, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsrc * nchan * npol) return;
    const int64_t p = i % npol, f = (i / npol) % nchan, s = i / (npol * nchan);
    const double ratio = frequency[f] / ref_freq[s];
    const int b = base[p];
    double v;
    if (b == 0) {
        v = stokes[s * npol + p];
        for (int64_t si = 0; si < nspi; ++si) v = __dmul_rn(v, pow(ratio, spi[(s * nspi + si) * npol + p]));
    } else {
        const double lr = b == 1 ? log(ratio) : log10(ratio);
        double acc = 0.0;
        for (int64_t si = 0; si < nspi; ++si)
            acc = __dadd_rn(acc, __dmul_rn(spi[(s * nspi + si) * npol + p], ipow_sq(lr, (int)si + 1)));
        v = __dmul_rn(stokes[s * npol + p], b == 1 ? exp(acc) : pow(10.0, acc));
    }
    out[i] = v;
}

}  // namespace

AF_EXPORT int af_spectral_model_f64(const double *stokes, const double *spi, const double *ref_freq,
                                    const double *frequency, const int *base, int64_t nsrc, int64_t nspi, int64_t npol,
                                    int64_t nchan, double *out, void *stream)
{
    AF_REQUIRE(nsrc >= 0 && nspi >= 0 && npol >= 1 && nchan >= 0, "af_spectral_model_f64: bad extents");
    if (nsrc == 0 || nchan == 0) return AF_OK;
    AF_REQUIRE(stokes && ref_freq && frequency && base && out && (spi || nspi == 0), "af_spectral_model_f64: NULL array");
    const int64_t total = nsrc * nchan * npol;
    AF_REQUIRE(af_cdiv(total, 256) < (1LL << 31), "af_spectral_model_f64: problem too large for one launch");
    hipLaunchKernelGGL(spectral_model_kernel, dim3((unsigned)af_cdiv(total, 256)), dim3(256), 0, af_stream(stream), stokes,
                       spi, ref_freq, frequency, base, nsrc, nspi, npol, nchan, out);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

AF_EXPORT int af_feed_rotation_f64(const double *parallactic_angles, int64_t n, int feed_type, double *out, void *stream)
{
    AF_REQUIRE(feed_type == AF_FEED_LINEAR || feed_type == AF_FEED_CIRCULAR, "Invalid feed_type");
    AF_REQUIRE(n >= 0, "af_feed_rotation_f64: negative extent");
    if (n == 0) return AF_OK;
    AF_REQUIRE(parallactic_angles && out, "af_feed_rotation_f64: NULL array");
    hipLaunchKernelGGL((feed_rotation_kernel<double, double2>), dim3((unsigned)af_cdiv(n, 256)), dim3(256), 0,
                       af_stream(stream), parallactic_angles, n, feed_type, reinterpret_cast<double2 *>(out));
    AF_LAUNCH_CHECK();
    return AF_OK;
}

AF_EXPORT int af_feed_rotation_f32(const float *parallactic_angles, int64_t n, int feed_type, float *out, void *stream)
{
    AF_REQUIRE(feed_type == AF_FEED_LINEAR || feed_type == AF_FEED_CIRCULAR, "Invalid feed_type");
    AF_REQUIRE(n >= 0, "af_feed_rotation_f32: negative extent");
    if (n == 0) return AF_OK;
    AF_REQUIRE(parallactic_angles && out, "af_feed_rotation_f32: NULL array");
    hipLaunchKernelGGL((feed_rotation_kernel<float, float2>), dim3((unsigned)af_cdiv(n, 256)), dim3(256), 0,
                       af_stream(stream), parallactic_angles, n, feed_type, reinterpret_cast<float2 *>(out));
    AF_LAUNCH_CHECK();
    return AF_OK;
}

AF_EXPORT int af_gaussian_shape_f64(const double *uvw, const double *frequency, const double *shape_params, int64_t nsrc,
                                    int64_t nrow, int64_t nchan, double *out, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    AF_REQUIRE(nsrc >= 0 && nrow >= 0 && nchan >= 0, "af_gaussian_shape_f64: negative extent");
    if (nsrc == 0 || nrow == 0 || nchan == 0) return AF_OK;
    AF_REQUIRE(uvw && frequency && shape_params && out, "af_gaussian_shape_f64: NULL array");
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= (size_t)nsrc * 4 * sizeof(double),
               "af_gaussian_shape_f64: workspace too small (nsrc * 32 bytes)");
    hipStream_t st = af_stream(stream);
    double *p = static_cast<double *>(workspace);
    hipLaunchKernelGGL(gauss_params_kernel, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st, shape_params, nsrc, p);
    AF_LAUNCH_CHECK();
    // gaussian_shape.py:23-25
    const double fwhm = 2.0 * sqrt(2.0 * log(2.0));
    const double gauss_scale = (1.0 / fwhm) * sqrt(2.0) * 3.141592653589793 / AF_LIGHTSPEED;
    const int64_t total = nsrc * nrow * nchan;
    AF_REQUIRE(af_cdiv(total, 256) < (1LL << 31), "af_gaussian_shape_f64: problem too large for one launch");
    hipLaunchKernelGGL(gaussian_shape_kernel, dim3((unsigned)af_cdiv(total, 256)), dim3(256), 0, st, uvw, frequency, p,
                       nsrc, nrow, nchan, gauss_scale, out);
    AF_LAUNCH_CHECK();
    return AF_OK;
}
