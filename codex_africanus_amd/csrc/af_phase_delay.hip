// phase_delay: K[s,r,c] = exp(i*C*(l u + m v + n w)*nu) materialised as (src,row,chan).
//
// Replaces africanus/rime/phase.py:28-61.  Pure write-bandwidth kernel (16 B per element
// for complex128): one lane owns one (row, chan) cell and walks a block of sources, so
// every store instruction writes 64 consecutive complex values (1 KiB per wave).
// Arithmetic follows the reference's operation order with no fp contraction:
//   n = sqrt(max(0, 1 - l^2 - m^2)) - 1 (:42-43), real_phase = C*(l*u + m*v + n*w) (:49),
//   p = real_phase*nu (:53), out = (cos p, sin p) (:58-59); constants in lm's dtype (:23-25).
// p is the reference's value bit for bit; cos/sin come from sincos_radians (af_sincos.h): Cody-Waite
// reduction + 7-term polynomials, ~1e-16 absolute, a third of the library routine's instructions.
#include "af_common.h"
#include "af_sincos.h"

namespace {

constexpr int SRC_PER_BLOCK = 16;
constexpr int THREADS = 256;

template <typename T> struct Ops;
template <> struct Ops<double> {
    static __device__ __forceinline__ double mul(double a, double b) { return __dmul_rn(a, b); }
    static __device__ __forceinline__ double add(double a, double b) { return __dadd_rn(a, b); }
    static __device__ __forceinline__ double sub(double a, double b) { return __dsub_rn(a, b); }
    static __device__ __forceinline__ double sqrt_(double a) { return __dsqrt_rn(a); }
    static __device__ __forceinline__ void sincos_(double p, double *s, double *c) { sincos_radians(p, *c, *s); }
    typedef double2 vec2;
    static __device__ __forceinline__ vec2 make2(double a, double b) { return make_double2(a, b); }
};
template <> struct Ops<float> {
    static __device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
    static __device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
    static __device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }
    // correctly rounded float sqrt via double (53 >= 2*24+2 bits: double rounding is innocuous)
    static __device__ __forceinline__ float sqrt_(float a) { return (float)__dsqrt_rn((double)a); }
    // float sin/cos evaluated in double and rounded once (within 0.5 ulp + of libm's cosf/sinf)
    static __device__ __forceinline__ void sincos_(float p, float *s, float *c)
    {
        double sd, cd;
        sincos_radians((double)p, cd, sd);
        *s = (float)sd;
        *c = (float)cd;
    }
    typedef float2 vec2;
    static __device__ __forceinline__ vec2 make2(float a, float b) { return make_float2(a, b); }
};

// grid: (ceil(nrow*nchan / 256), ceil(nsrc / SRC_PER_BLOCK))
template <typename T>
__global__ __launch_bounds__(THREADS) void phase_delay_kernel(const T *__restrict__ lm, int64_t nsrc,
                                                              const T *__restrict__ uvw, int64_t nrow,
                                                              const T *__restrict__ freq, int64_t nchan,
                                                              T constant, T *__restrict__ out)
{
    using O = Ops<T>;
    __shared__ T s_lmn[SRC_PER_BLOCK][3];
    const int64_t s0 = (int64_t)blockIdx.y * SRC_PER_BLOCK;
    const int ns = (int)((nsrc - s0 < SRC_PER_BLOCK) ? (nsrc - s0) : SRC_PER_BLOCK);
    if (threadIdx.x < ns) {
        const T one = (T)1.0, zero = (T)0.0;
        T l = lm[2 * (s0 + threadIdx.x)], m = lm[2 * (s0 + threadIdx.x) + 1];
        T n = O::sub(O::sub(one, O::mul(l, l)), O::mul(m, m));
        n = O::sub(O::sqrt_(n < zero ? zero : n), one);
        s_lmn[threadIdx.x][0] = l;
        s_lmn[threadIdx.x][1] = m;
        s_lmn[threadIdx.x][2] = n;
    }
    __syncthreads();
    const int64_t cell = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int64_t ncell = nrow * nchan;
    if (cell >= ncell) return;
    const int64_t r = cell / nchan, c = cell - r * nchan;
    const T u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
    const T nu = freq[c];
    typename O::vec2 *o = reinterpret_cast<typename O::vec2 *>(out) + s0 * ncell + cell;
    for (int k = 0; k < ns; ++k) {
        const T l = s_lmn[k][0], m = s_lmn[k][1], n = s_lmn[k][2];
        const T real_phase = O::mul(constant, O::add(O::add(O::mul(l, u), O::mul(m, v)), O::mul(n, w)));
        const T p = O::mul(real_phase, nu);
        T sp, cp;
        O::sincos_(p, &sp, &cp);
        o[(int64_t)k * ncell] = O::make2(cp, sp);
    }
}

template <typename T>
int phase_delay(const T *lm, int64_t nsrc, const T *uvw, int64_t nrow, const T *freq, int64_t nchan,
                int convention, T *out, void *stream)
{
    AF_REQUIRE(convention == AF_CONVENTION_FOURIER || convention == AF_CONVENTION_CASA,
               "convention not in ('fourier', 'casa')");
    AF_REQUIRE(nsrc >= 0 && nrow >= 0 && nchan >= 0, "af_phase_delay: negative extent");
    if (nsrc == 0 || nrow == 0 || nchan == 0) return AF_OK;
    AF_REQUIRE(lm && uvw && freq && out, "af_phase_delay: NULL array");
    // phase.py:23-25,29-32: neg_two_pi_over_c cast to lm.dtype; 'casa' negates the cast value
    const T neg = (T)AF_MINUS_TWO_PI_OVER_C;
    const T constant = convention == AF_CONVENTION_FOURIER ? neg : -neg;
    const int64_t gx = af_cdiv(nrow * nchan, THREADS), gy = af_cdiv(nsrc, SRC_PER_BLOCK);
    AF_REQUIRE(gx < (1LL << 31) && gy <= 65535, "af_phase_delay: problem too large for one launch");
    hipLaunchKernelGGL((phase_delay_kernel<T>), dim3((unsigned)gx, (unsigned)gy), dim3(THREADS), 0,
                       af_stream(stream), lm, nsrc, uvw, nrow, freq, nchan, constant, out);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

}  // namespace

AF_EXPORT int af_phase_delay_f64(const double *lm, int64_t nsrc, const double *uvw, int64_t nrow,
                                 const double *frequency, int64_t nchan, int convention, double *out,
                                 void *stream)
{
    return phase_delay<double>(lm, nsrc, uvw, nrow, frequency, nchan, convention, out, stream);
}

AF_EXPORT int af_phase_delay_f32(const float *lm, int64_t nsrc, const float *uvw, int64_t nrow,
                                 const float *frequency, int64_t nchan, int convention, float *out,
                                 void *stream)
{
    return phase_delay<float>(lm, nsrc, uvw, nrow, frequency, nchan, convention, out, stream);
}
