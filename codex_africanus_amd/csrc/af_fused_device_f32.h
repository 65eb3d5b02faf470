// Single-precision device code shared by the two complex64 fused-predict kernels: af_fused_gemm_c64.hip (antenna-decomposable
// uvw: one complex GEMM per (timestep, channel) on the fp32 matrix cores) and af_fused_predict_c64.hip (lane = rows: any uvw).
// Float workspace layout, the preparation kernels that widen the float32 inputs the set-up reads in double, the per-channel
// beam planes as 64-byte float records and their bilinear sampler (africanus/rime/fast_beam_cubes.py:110-238, frequency first),
// DPP broadcasts and packed complex helpers.  Anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "af_fused_device.h"

namespace {

constexpr int VREC32 = 16;          // floats per cell record: 4 correlations x (re, im, |.|, 0)

template <int QL> __device__ __forceinline__ float quad_bcastf(float x)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), QL * 0x55, 0xf, 0xf, true));
}
template <int ODD> __device__ __forceinline__ float pair_bcastf(float x)
{
    constexpr int PERM = ODD ? 0xF5 : 0xA0;
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), PERM, 0xf, 0xf, true));
}
struct C2f {
    float re, im;
};
typedef float v2f __attribute__((ext_vector_type(2)));
// complex arithmetic on (re, im) pairs: v_pk_mul_f32 / v_pk_fma_f32, two float32 operations per instruction (54.6 -> 53.5 ms),
// the swaps and signs of the products as operand modifiers (op_sel picks the half of a source that feeds the LOW result,
// op_sel_hi the HIGH one) -- written as asm: from vector expressions the compiler builds the swapped / negated pairs with
// v_mov / v_xor.  Early-clobber outputs: a result whose HIGH half reads the LOW half of a source must not share its registers.
__device__ __forceinline__ C2f cmulf(C2f a, C2f b)
{
    const v2f av = {a.re, a.im}, bv = {b.re, b.im};
    v2f t, z;       // (ar br, ar bi) then (- ai bi, + ai br)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=&v"(t) : "v"(av), "v"(bv));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=&v"(z) : "v"(av), "v"(bv), "v"(t));
    C2f r;
    r.re = z.x; r.im = z.y;
    return r;
}
__device__ __forceinline__ void cmacf(C2f &acc, C2f a, C2f b)
{
    const v2f av = {a.re, a.im}, bv = {b.re, b.im};
    v2f z = {acc.re, acc.im};
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(z) : "v"(av), "v"(bv));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(z) : "v"(av), "v"(bv));
    acc.re = z.x; acc.im = z.y;
}

struct alignas(4) F3 {
    float x, y, z;
};

// ---- preparation: float32 inputs -> the double arrays the set-up reads ----------------------------------------------
__global__ void prep_src_f32(const float *__restrict__ lm, int64_t nsrc, double *__restrict__ lmn)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    const double l = (double)lm[2 * s], m = (double)lm[2 * s + 1];
    double n = __dsub_rn(__dsub_rn(1.0, __dmul_rn(l, l)), __dmul_rn(m, m));
    n = __dsub_rn(__dsqrt_rn(n < 0.0 ? 0.0 : n), 1.0);                 // phase_delay's clamped n (africanus/rime/phase.py:42-43)
    lmn[4 * s + 0] = l; lmn[4 * s + 1] = m; lmn[4 * s + 2] = n; lmn[4 * s + 3] = 0.0;
}
__global__ void prep_freq_f32(const float *__restrict__ freq, int64_t nchan, int sign, const float *__restrict__ fmap,
                              int64_t nud, const float *__restrict__ ext, double *__restrict__ f4, double *__restrict__ freq_d,
                              double *__restrict__ fmap_d, double *__restrict__ ext_d)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nchan) {
        freq_d[c] = (double)freq[c];
        f4[c] = 4.0 * (double)sign * (double)freq[c] / AF_LIGHTSPEED;
    }
    if (c < nud) fmap_d[c] = (double)fmap[c];
    if (c < 4) ext_d[c] = (double)ext[c];
}
// per-channel planes of float records (frequency first, as beam_plane_kernel of af_fused_device.h; interpolated in double,
// rounded once)
__global__ void beam_plane_kernel_f32(const float2 *__restrict__ beam, int64_t ncell, int64_t beam_nud,
                                      const double *__restrict__ freq_data, int64_t f0, float *__restrict__ planes)
{
    const int64_t f = f0 + blockIdx.y;
    const double nud = freq_data[3 * f + 1], inv = 1.0 - nud;
    const int64_t gc0 = (int64_t)freq_data[3 * f + 2];
    float *__restrict__ rec = planes + (int64_t)blockIdx.y * ncell * VREC32;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // (cell, corr)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < ncell * 4; i += stride) {
        const int64_t cell = i >> 2, corr = i & 3;
        const float2 lo = beam[(cell * beam_nud + gc0) * 4 + corr], hi = beam[(cell * beam_nud + gc0 + 1) * 4 + corr];
        float4 r;
        r.x = (float)(nud * (double)lo.x + inv * (double)hi.x);
        r.y = (float)(nud * (double)lo.y + inv * (double)hi.y);
        r.z = (float)(nud * hypot((double)lo.x, (double)lo.y) + inv * hypot((double)hi.x, (double)hi.y));
        r.w = 0.0f;
        *reinterpret_cast<float4 *>(rec + i * 4) = r;
    }
}

// one correlation of the bilinear sample + the amplitude-preserving normalisation corr_sum * absc_sum / |corr_sum|
// (africanus/rime/fast_beam_cubes.py:170-235) in float32; 1 / |.| by v_rsq_f32 (1 ulp)
__device__ __forceinline__ C2f beam_reduce1f(const F3 (&v)[4], const float (&wt)[4])
{
    float cre = 0.0f, cim = 0.0f, absc = 0.0f;
    v2f c2 = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const v2f w2 = {wt[k], wt[k]}, x2 = {v[k].x, v[k].y};
        c2 = __builtin_elementwise_fma(w2, x2, c2);
        absc = fmaf(wt[k], v[k].z, absc);
    }
    cre = c2.x; cim = c2.y;

    const float n2 = fmaf(cre, cre, __fmul_rn(cim, cim));
    const float sc = (n2 == 0.0f) ? absc : __fmul_rn(absc, __builtin_amdgcn_rsqf(n2));
    C2f r;
    r.re = __fmul_rn(cre, sc);
    r.im = __fmul_rn(cim, sc);
    return r;
}

struct WsS {
    size_t lmn, f4, freq_d, fmap_d, ext_d, freq_data, gauss, planes, total;
};
WsS ws_s(int64_t nsrc, int64_t nchan, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud)
{
    WsS w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    w.lmn = take((size_t)nsrc * 4 * sizeof(double));
    w.f4 = take((size_t)nchan * sizeof(double));
    w.freq_d = take((size_t)nchan * sizeof(double));
    w.fmap_d = take((size_t)beam_nud * sizeof(double));
    w.ext_d = take(4 * sizeof(double));
    w.freq_data = take((size_t)nchan * 3 * sizeof(double));
    w.gauss = take((size_t)nsrc * 4 * sizeof(double));    // (el gs, em gs, er, is_extended) per source: the row kernel only
    const int64_t group = nchan < PLANE_GROUP ? nchan : PLANE_GROUP;
    w.planes = take((size_t)group * beam_lw * beam_mh * VREC32 * sizeof(float));
    w.total = o;
    return w;
}

}  // namespace
