// im_to_vis with float32 inputs and a complex64 result, computed natively in single precision (round 3).
//
// Replaces africanus/dft/kernels.py:33-67 for the case every input is float32: the reference then runs its whole
// loop in float32 (result dtype by promotion, africanus/dft/kernels.py:26-31, africanus/util/type_inference.py:24-26),
// including the phase C (l u + m v + n w) nu -- thousands of radians for kilometre baselines: 1.4e-4 of the peak
// visibility in error at 4 km (golden G13).  Here the PHASES are float64 and everything after them float32:
//   * fp64 VALU, per (row, source, channel tile): q = l u + m v + n w and two phases.
//       - calls whose phases can exceed ~100 rad (`chain_regime`, decided on the device from max |uvw|, max |lmn| and
//         max nu): the phase of the tile's MIDDLE channel and the channel step, reduced in fp64 (`chain_phasor`), give
//         float32 anchor and step phasors from a 1024-entry float32 table; the tile's other phasors follow by a
//         float32 rotation recurrence outwards from the anchor (packed v_pk_mul_f32 / v_pk_fma_f32, <= CT / 2 steps);
//       - metre baselines, where the reference's float32 loop is itself at the float32 floor: the fp64 table phasors
//         of af_sincos.h (~2e-16) and the fp64 three-term recurrence y[j+1] = 2 cos(d) y[j] - y[j-1], every phasor
//         rounded to float32 once (golden G3 asks for that);
//   * fp32 MATRIX pipe, per channel: v_mfma_f32_4x4x1_16b_f32 -- sixteen 4x4x1 outer products -- with
//       A = the source's pixels: ONE register holds 16 blocks x 4 floats = the 4 correlations of 16 channels, and the
//           instruction's CBSZ = 4 / ABID = k fields broadcast block k (channel k) to all sixteen blocks,
//       B = the lane's own phasor component (lane = row: block b, column j = lane 4 b + j),
//       D = 4 registers per lane = the 4 correlations of the lane's row (layout measured: tools/probe/probe_mfma_f32.hip),
//     i.e. acc[chan][0..3] += pixel[chan][0..3] * y in one instruction where the VALU needs four, with a wave-uniform
//     operand at no cost (no DPP, no scalar loads, no LDS) and the accumulators in AGPRs.  Two MFMAs per channel for a
//     real image (re, im), four for a complex one (records carry (re, im, -im): the MFMA cannot negate an operand).
// An 8-cycle MFMA holds the SIMD's vector issue for its whole duration, so nothing overlaps it: a source iteration is
// its VALU instructions (~4.7 cycles each from one wave) plus 8 cycles per MFMA; per channel the chain form issues two
// packed float32 instructions where the fp64 form issues 2 FMA + 2 conversions (DESIGN.md 3.10 has the budget and the
// measured times).  Accuracy: closer to the float64 transform of the float32 inputs than the reference's float32
// loop in every regime (tests/test_gpu_f32.py: golden G3 at metre baselines, G13 at 4 km, a sweep in between).
//
// Frequencies.  A float32 frequency axis is never an exact arithmetic progression (64-128 Hz of rounding at L band),
// and over a 4 km baseline that rounding is worth 1.2e-4 of the peak visibility -- as much as the reference's own
// error (G13).  The prep pass classifies the band on the device: 0 = every tile exactly uniform (to 2 ulp of float64):
// plain recurrence; 1 = uniform to within float32 rounding: recurrence on the tile's least-squares grid plus the
// first-order correction y_c (1 + i q kappa_c), kappa_c = 2 pi (nu_c - grid_c) / c (3 float32 operations per channel,
// second-order term < 1e-7); 2 = anything else: `dft_f32_exact_kernel`, one fp64 sincos per (row, source, channel).
// AF_DFT_RECURRENCE (the caller asserts a uniform band) runs class 1 as class 0; AF_DFT_EXACT forces class 2.
#include <stdlib.h>

#include <type_traits>

#include "af_common.h"
#include "af_sincos.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int ROWS = 256;
constexpr int G32 = 64;   // floats per record register = 16 MFMA blocks x 4

template <int I, int N, typename F>
__device__ __forceinline__ void static_for32(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for32<I + 1, N>(f);
    }
}

struct Ws32 {
    size_t flags, lmn, srcbad, tilef, kappa, freqc, colstate, tilestate, records, total;
    int64_t ntile;
    int ct, groups;
};

// MFMA blocks (of 4 correlations) a channel takes in a record: (re) or (re, im, -im)
constexpr int blocks_per_chan(bool cplx) { return cplx ? 3 : 1; }
// channels per record register (16 blocks)
constexpr int chans_per_reg(bool cplx) { return 16 / blocks_per_chan(cplx); }
constexpr int groups_of(int ct, bool cplx) { return (ct + chans_per_reg(cplx) - 1) / chans_per_reg(cplx); }

// channel-tile width: 2 x CT accumulator quads = 8 CT AGPRs; 16 channels (128 AGPRs) leave room for two waves per
// SIMD (AFHIP_F32_CT=22 / 32: wider tiles for A/B runs; complex images 15 / 20 / 30: whole 5-channel registers)
int tile_width(bool cplx)
{
    static const int want = getenv("AFHIP_F32_CT") ? atoi(getenv("AFHIP_F32_CT")) : 16;
    if (cplx) return want >= 32 ? 30 : want >= 22 ? 20 : 15;
    return want >= 32 ? 32 : want >= 22 ? 22 : 16;
}

Ws32 ws32_layout(int64_t nsrc, int64_t nchan, int64_t ncorr, int is_complex)
{
    Ws32 L;
    L.ct = tile_width(is_complex != 0);
    L.ntile = af_cdiv(nchan > 0 ? nchan : 1, L.ct);
    L.groups = groups_of(L.ct, is_complex != 0);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    L.flags = take(16 * sizeof(int));
    L.lmn = take((size_t)(nsrc > 0 ? nsrc : 1) * 4 * sizeof(double));
    L.srcbad = take((size_t)(nsrc > 0 ? nsrc : 1) * sizeof(int));
    L.tilef = take((size_t)L.ntile * 2 * sizeof(double));
    L.kappa = take((size_t)L.ntile * L.ct * sizeof(float));
    L.freqc = take((size_t)(nchan > 0 ? nchan : 1) * sizeof(double));
    L.colstate = take((size_t)L.ntile * L.ct * (ncorr > 0 ? ncorr : 1) * sizeof(int));
    L.tilestate = take((size_t)L.ntile * sizeof(int));
    L.records = take((size_t)L.ntile * (nsrc > 0 ? nsrc : 1) * L.groups * G32 * sizeof(float));
    L.total = o;
    return L;
}

// flags[0] = the band's class (f32_prep_freq); [2..4] = non-negative floats (bit patterns, atomicMax as unsigned):
// max |uvw|^2 over the rows, max (l^2 + m^2 + n^2) over the sources, max |nu| / c -- their product bounds the phase
constexpr int FLAG_ROW_REACH = 2, FLAG_SRC_REACH = 3, FLAG_FREQ_REACH = 4;

__global__ __launch_bounds__(256) void f32_rows_reach(const float *__restrict__ uvw, int64_t nrow, int *__restrict__ flags)
{
    float best = 0.0f;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (int64_t)gridDim.x * blockDim.x) {
        const float u = uvw[3 * r], v = uvw[3 * r + 1], w = uvw[3 * r + 2];
        const float d = u * u + v * v + w * w;
        if (isfinite(d)) best = fmaxf(best, d);
    }
    for (int o = 32; o; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned *>(flags) + FLAG_ROW_REACH, __float_as_uint(best));
}

// Largest phase of the call in radians (an upper bound).  The float32 recurrence below ("chain") carries ~1e-7 of a
// phasor's magnitude per step; the reference's float32 loop carries 6e-8 of the PHASE (radians): beyond ~100 rad the
// chain is the closer of the two by a wide margin, below it the fp64-phasor kernel keeps the contract (golden G3)
constexpr float CHAIN_REACH_RAD = 100.0f;
__device__ __forceinline__ bool chain_regime(const int *__restrict__ flags)
{
    const float reach = 6.2831855f * sqrtf(__int_as_float(flags[FLAG_ROW_REACH]) * __int_as_float(flags[FLAG_SRC_REACH])) *
                        __int_as_float(flags[FLAG_FREQ_REACH]);
    return reach >= CHAIN_REACH_RAD;
}

// exp(2 pi i x / 1024) in float32 from a phase x kept in float64: k = rint(x) by the 1.5 * 2^52 trick (the integer is
// the low word of the sum: no conversion), theta = x - k in [-0.5, 0.5] table steps, table[k mod 1024] (float32, 8 KB of
// LDS) times (1 - t^2 / 2 + i t), t = 2 pi theta / 1024 <= 3.1e-3 (next terms 4.8e-9 and 3.5e-12): 5 fp64 + 9 fp32
// operations where the fp64 table phasor takes ~14 fp64 and two conversions.  Non-finite x -> NaN.
constexpr int CHAIN_TABLE = 1024;
constexpr int CHAIN_ROWS = 256;   // (512 with a barrier per source, so that the two waves of a SIMD run their float32 phases
                                  // together, measured slower: 19.5 against 17.7 ms at C2's counts)

// Packed float32 forms of the recurrence.  ONE wave issues a vector instruction every ~5 cycles whether it is
// v_fma_f32 or v_pk_fma_f32 (tools/probe/probe_f32_issue.hip: 4.5-5.0 / 5.0; two waves together 2.3 / 4.4), and in this
// kernel a wave's float32 phase runs while its SIMD partner is held by its MFMAs: the packed forms halve the phase.
// y = (re, im) in a register pair, s = (cos, sin):  y s  = (re c - im s, im c + re s) as  t = (-im s, re s) then
// fma(y, (c, c), t) -- per component the same two roundings as the scalar form (product, then fma).  Outputs are
// early-clobber: a packed instruction whose HIGH result reads the LOW half of a source (op_sel_hi 0) must not have
// that source as its destination.
__device__ __forceinline__ v2f rot_up(v2f y, v2f s)
{
    v2f t, r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=&v"(t) : "v"(y), "v"(s));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=&v"(r) : "v"(y), "v"(s), "v"(t));
    return r;
}
// y conj(s) = (re c + im s, im c - re s)
__device__ __forceinline__ v2f rot_down(v2f y, v2f s)
{
    v2f t, r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=&v"(t) : "v"(y), "v"(s));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=&v"(r) : "v"(y), "v"(s), "v"(t));
    return r;
}
// a step of both chains, products first: up = a s, down = b conj(s)
__device__ __forceinline__ void rot_both(v2f a, v2f b, v2f s, v2f &up, v2f &down)
{
    v2f t, u;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=&v"(t) : "v"(a), "v"(s));
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=&v"(u) : "v"(b), "v"(s));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=&v"(up) : "v"(a), "v"(s), "v"(t));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=&v"(down) : "v"(b), "v"(s), "v"(u));
}
// (q kappa_j, q kappa_j+1) from q in the low half of q2
__device__ __forceinline__ v2f theta_pair(v2f q2, v2f kap2)
{
    v2f r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=&v"(r) : "v"(q2), "v"(kap2));
    return r;
}
// y (1 + i theta) = (re - theta im, im + theta re), theta = the low (HI = 0) or high (HI = 1) half of th
template <int HI> __device__ __forceinline__ v2f first_order(v2f y, v2f th)
{
    v2f r;
    if constexpr (HI == 0)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=&v"(r) : "v"(y), "v"(th));
    else
        asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=&v"(r) : "v"(y), "v"(th));
    return r;
}
__device__ __forceinline__ void chain_table_init(float2 *tab, int tid, int nthreads)
{
    for (int k = tid; k < CHAIN_TABLE; k += nthreads) {
        double sn, cs;
        sincospi(2.0 * (double)k / (double)CHAIN_TABLE, &sn, &cs);
        tab[k] = make_float2((float)cs, (float)sn);
    }
}
__device__ __forceinline__ void chain_phasor(const float2 *tab, double x, float &re, float &im)
{
    const double MAGIC = 6755399441055744.0;
    const double a = __dadd_rn(x, MAGIC);
    const int k = __double2loint(a);
    const float th = (float)__dsub_rn(x, __dsub_rn(a, MAGIC));
    const float2 t = tab[k & (CHAIN_TABLE - 1)];
    const float sn = __fmul_rn(th, 6.1359231515425649e-03f);                        // 2 pi / 1024
    const float cs = fmaf(__fmul_rn(th, th), -1.8824776459647568e-05f, 1.0f);       // (2 pi / 1024)^2 / 2
    re = fmaf(t.x, cs, -__fmul_rn(t.y, sn));
    im = fmaf(t.x, sn, __fmul_rn(t.y, cs));
}

// n = sqrt(1 - l^2 - m^2) - 1 (kernels.py:54, unclamped) in float64 from the float32 coordinates
// (clamp: phase_delay's n = sqrt(max(0, 1 - l^2 - m^2)) - 1, africanus/rime/phase.py:42-43 -- AF_DFT_CLAMP_N)
__global__ void f32_prep_src(const float *__restrict__ lm, int64_t nsrc, double *__restrict__ lmn, int *__restrict__ srcbad,
                             int *__restrict__ flags, int clamp)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    const double l = (double)lm[2 * s], m = (double)lm[2 * s + 1];
    double n = 1.0 - l * l - m * m;
    if (clamp && n < 0.0) n = 0.0;
    n = sqrt(n) - 1.0;
    const bool bad = !(isfinite(l) && isfinite(m) && isfinite(n));
    srcbad[s] = bad ? 1 : 0;
    if (!bad) atomicMax(reinterpret_cast<unsigned *>(flags) + FLAG_SRC_REACH, __float_as_uint((float)(l * l + m * m + n * n)));
    lmn[4 * s + 0] = bad ? 0.0 : l;
    lmn[4 * s + 1] = bad ? 0.0 : m;
    lmn[4 * s + 2] = bad ? 0.0 : n;
    lmn[4 * s + 3] = 0.0;
}

// per tile: the grid (first frequency, step) in turns per metre, the residuals kappa, the band's class
__global__ void f32_prep_freq(const float *__restrict__ freq, int64_t nchan, int64_t ntile, int CT, int sign,
                              double *__restrict__ tilef, float *__restrict__ kappa, double *__restrict__ freqc,
                              int *__restrict__ flags)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntile) return;
    const int64_t c0 = t * CT;
    const int64_t nc = nchan - c0 < CT ? nchan - c0 : CT;
    // least-squares line through the tile's channels (equal to the end-point line for an exact progression)
    double sx = 0.0, sy = 0.0, sxx = 0.0, sxy = 0.0;
    const double fref = (double)freq[c0];
    for (int64_t j = 0; j < nc; ++j) {
        const double y = (double)freq[c0 + j] - fref;
        sx += (double)j; sy += y; sxx += (double)j * j; sxy += (double)j * y;
    }
    double df = 0.0, f0 = fref;
    if (nc > 1) {
        const double den = (double)nc * sxx - sx * sx;
        df = ((double)nc * sxy - sx * sy) / den;
        f0 = fref + (sy - df * sx) / (double)nc;
    }
    // an exact progression must come out exactly: prefer the end-point line when it reproduces every channel
    const double dfe = nc > 1 ? ((double)freq[c0 + nc - 1] - fref) / (double)(nc - 1) : 0.0;
    bool exact = isfinite(fref) && isfinite(dfe);
    for (int64_t j = 0; j < nc && exact; ++j) {
        const double f = (double)freq[c0 + j], pred = fref + (double)j * dfe;
        if (!(fabs(f - pred) <= 2.0 * 2.220446049250313e-16 * fmax(fabs(f), fabs(pred)))) exact = false;
    }
    if (exact) { f0 = fref; df = dfe; }
    int cls = exact ? 0 : 1;
    const double s1 = (double)sign / AF_LIGHTSPEED;
    for (int64_t j = 0; j < CT; ++j) {
        double eps = 0.0;
        if (j < nc) {
            const double f = (double)freq[c0 + j];
            eps = f - (f0 + (double)j * df);
            // float32 rounding of a value near f: half an ulp = |f| 2^-24; allow four
            if (!(fabs(eps) <= 4.0 * 5.9604644775390625e-08 * fabs(f))) cls = 2;
            freqc[c0 + j] = 4.0 * f * s1;      // quarter turns per metre (dft_f32_exact_kernel)
            if (isfinite(f)) atomicMax(reinterpret_cast<unsigned *>(flags) + FLAG_FREQ_REACH, __float_as_uint((float)fabs(f * s1)));
        }
        kappa[t * CT + j] = (float)(6.283185307179586 * eps * s1);
    }
    if (!isfinite(f0) || !isfinite(df)) cls = 2;
    tilef[2 * t + 0] = f0 * s1;
    tilef[2 * t + 1] = df * s1;
    atomicMax(&flags[0], cls);
}

// records: [tile][source][groups * 64] floats; register g, block k of it = channel g * CPR + k / BPC of the tile,
// plane k % BPC of (re | re, im, -im), floats = correlations 0..3 (zero beyond ncorr).  Zero beyond nchan and for
// sources whose (l, m, n) is not finite (their effect goes through colstate).
template <typename P>
__global__ void f32_pack_records(const P *__restrict__ image, int cplx, int64_t nsrc, int64_t nchan, int64_t ncorr,
                                 int64_t ntile, int CT, int groups, const int *__restrict__ srcbad,
                                 float *__restrict__ rec)
{
    const int W = cplx ? 2 : 1, BPC = cplx ? 3 : 1, CPR = 16 / BPC;
    const int64_t per = (int64_t)groups * G32;
    const int64_t total = ntile * nsrc * per;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t slot = i % per, s = (i / per) % nsrc, tile = i / (per * nsrc);
        const int g = (int)(slot / G32), blk = (int)(slot % G32) / 4, c = (int)(slot % 4);
        const int j = g * CPR + blk / BPC, plane = blk % BPC;
        const int64_t ch = tile * CT + j;
        float v = 0.0f;
        if (blk < CPR * BPC && j < CT && ch < nchan && c < ncorr && !srcbad[s]) {
            const P *px = image + ((s * nchan + ch) * ncorr + c) * W;
            v = plane == 0 ? (float)px[0] : plane == 1 ? (float)px[W - 1] : -(float)px[W - 1];
        }
        rec[i] = v;
    }
}

// column state per (chan, corr): the reference's `if image[s,nu,c]:` (kernels.py:64) only matters when a phasor is
// not finite: an all-zero column stays exactly 0 (state 1), a non-finite source with a nonzero pixel poisons it (2)
__global__ __launch_bounds__(64) void f32_colstate(const float *__restrict__ image, int W, int64_t nsrc, int64_t nchan,
                                                   int64_t ncorr, int CT, const int *__restrict__ srcbad,
                                                   int *__restrict__ colstate, int *__restrict__ tilestate)
{
    const int64_t i = blockIdx.x;
    const int64_t c = i % ncorr, ch = i / ncorr;
    int state = 0;
    if (ch < nchan) {
        bool any_nz = false, poison = false;
        for (int64_t s = threadIdx.x; s < nsrc; s += 64) {
            const float *px = image + ((s * nchan + ch) * ncorr + c) * W;
            const bool nz = (px[0] != 0.0f) || (W == 2 && px[1] != 0.0f);
            any_nz |= nz;
            poison |= (nz && srcbad[s]);
        }
        const bool w_nz = __ballot(any_nz) != 0ULL, w_poison = __ballot(poison) != 0ULL;
        state = w_poison ? 2 : (w_nz ? 0 : 1);
    }
    if (threadIdx.x == 0) {
        colstate[i] = state;
        if (state) atomicOr(&tilestate[ch / CT], state);
    }
}

// rows of 64 lanes through LDS, out as fully coalesced 8-byte stores (per-lane stores at a row stride of
// nchan * ncorr * 8 bytes leave partially written lines behind)
template <int CT, int NC, int TB = ROWS>
__device__ __forceinline__ void store_tile32(const v4f (&are)[CT], const v4f (&aim)[CT], float2 *stage, float *out,
                                             int64_t row0, int64_t nrow, int64_t nchan, int64_t c0, int wave, int lane,
                                             const int *__restrict__ colstate, int tstate)
{
    constexpr int PER_ROW = CT * NC;
    constexpr int STRIDE = PER_ROW + 1;
    const int64_t seg_chans = nchan - c0 < CT ? nchan - c0 : CT;
    const int seg_len = (int)(seg_chans * NC);
    for (int pass = 0; pass < TB / 64; ++pass) {
        if (wave == pass) {
            float2 *dst = stage + lane * STRIDE;
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) dst[j * NC + c] = make_float2(are[j][c], aim[j][c]);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * PER_ROW; e += TB) {
            const int rl = e / PER_ROW, col = e - rl * PER_ROW;
            const int64_t row = row0 + pass * 64 + rl;
            if (row < nrow && col < seg_len) {
                float2 v = stage[rl * STRIDE + col];
                if (tstate) {   // block-uniform: rare zero-column / NaN-source semantics
                    const int st = colstate[c0 * NC + col];
                    if (st == 1) v = make_float2(0.0f, 0.0f);
                    if (st == 2) v = make_float2(__builtin_nanf(""), __builtin_nanf(""));
                }
                reinterpret_cast<float2 *>(out)[(row * nchan + c0) * NC + col] = v;
            }
        }
        __syncthreads();
    }
}

// grid: (ceil(nrow / 256), tiles); block 256 = 4 waves of 64 consecutive rows on tile blockIdx.y.
// CHAIN: the tile's phasors by a float32 rotation recurrence outwards from the tile's middle channel (anchor and step
// from `chain_phasor`: fp64 phase, float32 table) instead of the fp64 three-term recurrence: 4 float32 operations per
// channel (8 cycles) where the fp64 form takes 2 FMA + 2 conversions (16), and a third of the set-up.  `force_chain`:
// -1 = by the call's phase bound (`chain_regime`), 0 / 1 = AFHIP_F32_CHAIN.
template <int CT, int NC, bool CPLX, bool CORR, bool CHAIN>
__global__ __launch_bounds__((CHAIN ? CHAIN_ROWS : ROWS), (CT <= 16 ? 2 : 1)) void dft_f32_kernel(
    const float *__restrict__ uvw, const float *__restrict__ records, const double *__restrict__ lmn,
    const double *__restrict__ tilef, const float *__restrict__ kappa, const int *__restrict__ flags,
    const int *__restrict__ colstate, const int *__restrict__ tilestate, float *__restrict__ out, int64_t nrow,
    int nsrc, int64_t nchan, int want_class, int force_chain)
{
    if (flags[0] != want_class) return;   // decided on the device by f32_prep_freq
    if ((force_chain >= 0 ? force_chain != 0 : chain_regime(flags)) != CHAIN) return;
    constexpr int TB = CHAIN ? CHAIN_ROWS : ROWS;
    __shared__ float2 stage[64 * (CT * NC + 1)];
    __shared__ double2 ptab[CHAIN ? 1 : PHASOR_TABLE];
    __shared__ float2 ftab[CHAIN ? CHAIN_TABLE : 1];
    if constexpr (CHAIN) chain_table_init(ftab, threadIdx.x, TB);
    else table_phasor_init(ptab, threadIdx.x, TB);
    __syncthreads();
    constexpr int BPC = blocks_per_chan(CPLX), CPR = chans_per_reg(CPLX);
    constexpr int NG = groups_of(CT, CPLX);
    const int tile = blockIdx.y;
    const int64_t c0 = (int64_t)tile * CT;
    int64_t row = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (row >= nrow) row = nrow - 1;
    const double u = (double)uvw[3 * row], v = (double)uvw[3 * row + 1], w = (double)uvw[3 * row + 2];
    // turns per metre -> 1/256 turns per metre (table_phasor's unit: an exact scaling)
    const double F0 = 256.0 * tilef[2 * tile], FD = 256.0 * tilef[2 * tile + 1];
    // chain: the anchor channel's frequency and the step in 1/1024 turns per metre
    constexpr int JA = CT / 2;
    const double FA = (double)CHAIN_TABLE * fma((double)JA, tilef[2 * tile + 1], tilef[2 * tile]);
    const double FS = (double)CHAIN_TABLE * tilef[2 * tile + 1];
    float kap[CT + 1];
    if constexpr (CORR) {
#pragma unroll
        for (int j = 0; j < CT; ++j) kap[j] = kappa[tile * CT + j];   // wave-uniform: scalar registers
        kap[CT] = 0.0f;
    }
    v4f are[CT], aim[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) are[j] = aim[j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    const int lane = threadIdx.x & 63;
    const float *__restrict__ rec = records + (int64_t)tile * nsrc * (NG * G32) + lane;
    float R[NG], Rn[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) R[g] = rec[g * G32];

#pragma unroll 1
    for (int s = 0; s < nsrc; ++s) {
        const int sn = s + 1 < nsrc ? s + 1 : s;
#pragma unroll
        for (int g = 0; g < NG; ++g) Rn[g] = rec[(int64_t)sn * (NG * G32) + g * G32];
        const double l = lmn[4 * s], m = lmn[4 * s + 1], n = lmn[4 * s + 2];
        const double q = fma(n, w, fma(m, v, __dmul_rn(l, u)));
        const float qf = (float)q;
        float yr[CT], yi[CT];
        double y0r, y0i, y1r, y1i, k2;
        if constexpr (CHAIN) {
            v2f y[CT], st;
            {
                float ar, ai, sr, si;
                chain_phasor(ftab, __dmul_rn(q, FA), ar, ai);
                chain_phasor(ftab, __dmul_rn(q, FS), sr, si);
                y[JA] = v2f{ar, ai};
                st = v2f{sr, si};
            }
            // two independent chains, a step of each in turn (the asm statements are volatile: they keep this order,
            // and the s_nop below stays between the last of them and the first MFMA)
            static_for32<1, JA + 1>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (JA + k < CT) rot_both(y[JA + k - 1], y[JA - k + 1], st, y[JA + k], y[JA - k]);
                else y[JA - k] = rot_down(y[JA - k + 1], st);     // downwards: times the conjugate step
            });
            if constexpr (CORR) {                       // y (1 + i theta_j), theta_j = q kappa_j
                const v2f q2 = {qf, qf};
                static_for32<0, (CT + 1) / 2>([&](auto jc) {
                    constexpr int j = 2 * decltype(jc)::value;
                    const v2f th = theta_pair(q2, v2f{kap[j], kap[j + 1]});
                    y[j] = first_order<0>(y[j], th);
                    if constexpr (j + 1 < CT) y[j + 1] = first_order<1>(y[j + 1], th);
                });
            }
#pragma unroll
            for (int j = 0; j < CT; ++j) { yr[j] = y[j].x; yi[j] = y[j].y; }
            // an MFMA must not read a register within two wait states of the VALU instruction that wrote it; the
            // compiler keeps that distance for instructions it knows, not for the asm statements above
            asm volatile("s_nop 1" ::: "memory");
        } else {
            double dr, di;
            table_phasor(ptab, __dmul_rn(q, F0), y0r, y0i);
            table_phasor(ptab, __dmul_rn(q, FD), dr, di);
            k2 = __dadd_rn(dr, dr);
            y1r = fma(y0r, dr, -__dmul_rn(y0i, di));
            y1i = fma(y0r, di, __dmul_rn(y0i, dr));
        }
        // phase 1, fp64 VALU only: the tile's phasors, rounded to float32 once.  phase 2, matrix pipe only: the MFMAs
        // back to back.  (Interleaved instruction by instruction a lone wave pays ~10 cycles at every MFMA <-> VALU
        // switch -- measured, tools/probe/probe_mfma_f32.hip: 16 MFMAs 152 cycles, 16 fp64 FMAs 108, alternating 340 --
        // whereas whole phases of different waves of a SIMD overlap.)
        static_for32<0, CT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (!CHAIN) {
                double yrd, yid;
                if constexpr (j == 0) { yrd = y0r; yid = y0i; }
                else if constexpr (j == 1) { yrd = y1r; yid = y1i; }
                else {
                    yrd = fma(k2, y1r, -y0r);
                    yid = fma(k2, y1i, -y0i);
                    y0r = y1r; y0i = y1i; y1r = yrd; y1i = yid;
                }
                yr[j] = (float)yrd; yi[j] = (float)yid;
            }
            if constexpr (CORR && !CHAIN) {   // y (1 + i theta), theta = q kappa_j
                const float th = qf * kap[j];
                const float cr = fmaf(-th, yi[j], yr[j]), ci = fmaf(th, yr[j], yi[j]);
                yr[j] = cr; yi[j] = ci;
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for32<0, CT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int g = j / CPR, b0 = (j % CPR) * BPC;
            if constexpr (CPLX) {
                are[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(R[g], yr[j], are[j], 4, b0, 0);       // + re * yr
                aim[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(R[g], yr[j], aim[j], 4, b0 + 1, 0);   // + im * yr
            } else {
                are[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(R[g], yr[j], are[j], 4, b0, 0);
                aim[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(R[g], yi[j], aim[j], 4, b0, 0);
            }
        });
        if constexpr (CPLX) {
            static_for32<0, CT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int g = j / CPR, b0 = (j % CPR) * BPC;
                are[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(R[g], yi[j], are[j], 4, b0 + 2, 0);   // - im * yi
                aim[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(R[g], yi[j], aim[j], 4, b0, 0);       // + re * yi
            });
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) R[g] = Rn[g];
    }
    store_tile32<CT, NC, TB>(are, aim, stage, out, (int64_t)blockIdx.x * TB, nrow, nchan, c0,
                             __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63, colstate, tilestate[tile]);
}

// any band: one phasor per (row, source, channel), phase in fp64 turns.  grid: (ceil(nrow / 256), nchan).
template <int NC, bool CPLX>
__global__ __launch_bounds__(ROWS) void dft_f32_exact_kernel(
    const float *__restrict__ uvw, const float *__restrict__ image, const double *__restrict__ lmn,
    const int *__restrict__ srcbad, const double *__restrict__ freqc, const int *__restrict__ flags,
    const int *__restrict__ colstate, float *__restrict__ out, int64_t nrow, int nsrc, int64_t nchan, int want_class)
{
    if (want_class >= 0 && flags[0] != want_class) return;
    constexpr int W = CPLX ? 2 : 1;
    const int64_t ch = blockIdx.y;
    const int64_t row = (int64_t)blockIdx.x * ROWS + threadIdx.x;
    if (row >= nrow) return;
    const double u = (double)uvw[3 * row], v = (double)uvw[3 * row + 1], w = (double)uvw[3 * row + 2];
    const double fc = freqc[ch];
    float acc[NC][2];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c][0] = acc[c][1] = 0.0f;
    for (int s = 0; s < nsrc; ++s) {
        if (srcbad[s]) continue;   // wave-uniform; such sources act through colstate
        const double l = lmn[4 * s], m = lmn[4 * s + 1], n = lmn[4 * s + 2];
        const double q = fma(n, w, fma(m, v, __dmul_rn(l, u)));
        double yrd, yid;
        sincos_quarter_turns<7>(__dmul_rn(q, fc), yrd, yid);       // fc in quarter turns per metre
        const float yr = (float)yrd, yi = (float)yid;
        const float *px = image + ((int64_t)s * nchan + ch) * NC * W;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if constexpr (CPLX) {
                const float pr = px[2 * c], pi = px[2 * c + 1];
                acc[c][0] = fmaf(pr, yr, fmaf(-pi, yi, acc[c][0]));
                acc[c][1] = fmaf(pi, yr, fmaf(pr, yi, acc[c][1]));
            } else {
                const float p = px[c];
                acc[c][0] = fmaf(p, yr, acc[c][0]);
                acc[c][1] = fmaf(p, yi, acc[c][1]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float2 o = make_float2(acc[c][0], acc[c][1]);
        const int st = colstate[ch * NC + c];
        if (st == 1) o = make_float2(0.0f, 0.0f);
        if (st == 2) o = make_float2(__builtin_nanf(""), __builtin_nanf(""));
        reinterpret_cast<float2 *>(out)[(row * nchan + ch) * NC + c] = o;
    }
}

struct Args32 {
    const float *uvw, *image;
    char *ws;
    const Ws32 *L;
    float *out;
    int64_t nrow, nchan;
    int nsrc, mode;
    hipStream_t st;
};

template <int CT, int NC, bool CPLX>
int launch32(const Args32 &a)
{
    const Ws32 &L = *a.L;
    const dim3 grid((unsigned)af_cdiv(a.nrow, ROWS), (unsigned)L.ntile), block(ROWS);
    const float *rec = reinterpret_cast<const float *>(a.ws + L.records);
    const double *lmn = reinterpret_cast<const double *>(a.ws + L.lmn), *tilef = reinterpret_cast<const double *>(a.ws + L.tilef);
    const float *kappa = reinterpret_cast<const float *>(a.ws + L.kappa);
    const int *flags = reinterpret_cast<const int *>(a.ws + L.flags), *colstate = reinterpret_cast<const int *>(a.ws + L.colstate);
    const int *tilestate = reinterpret_cast<const int *>(a.ws + L.tilestate);
    if (a.mode != AF_DFT_EXACT) {
        af_prof_begin(a.st);
        // (band class) x (phasor form): decided on the device, three of the four launches return at once
        static const int force = getenv("AFHIP_F32_CHAIN") ? atoi(getenv("AFHIP_F32_CHAIN")) : -1;
        const dim3 gridc((unsigned)af_cdiv(a.nrow, CHAIN_ROWS), (unsigned)L.ntile), blockc(CHAIN_ROWS);
        hipLaunchKernelGGL((dft_f32_kernel<CT, NC, CPLX, false, true>), gridc, blockc, 0, a.st, a.uvw, rec, lmn, tilef, kappa, flags,
                           colstate, tilestate, a.out, a.nrow, a.nsrc, a.nchan, 0, force);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL((dft_f32_kernel<CT, NC, CPLX, true, true>), gridc, blockc, 0, a.st, a.uvw, rec, lmn, tilef, kappa, flags,
                           colstate, tilestate, a.out, a.nrow, a.nsrc, a.nchan, 1, force);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL((dft_f32_kernel<CT, NC, CPLX, false, false>), grid, block, 0, a.st, a.uvw, rec, lmn, tilef, kappa, flags,
                           colstate, tilestate, a.out, a.nrow, a.nsrc, a.nchan, 0, force);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL((dft_f32_kernel<CT, NC, CPLX, true, false>), grid, block, 0, a.st, a.uvw, rec, lmn, tilef, kappa, flags,
                           colstate, tilestate, a.out, a.nrow, a.nsrc, a.nchan, 1, force);
        af_prof_end(a.st);
        AF_LAUNCH_CHECK();
    }
    const dim3 gridx((unsigned)af_cdiv(a.nrow, ROWS), (unsigned)a.nchan);
    hipLaunchKernelGGL((dft_f32_exact_kernel<NC, CPLX>), gridx, block, 0, a.st, a.uvw, a.image, lmn,
                       reinterpret_cast<const int *>(a.ws + L.srcbad), reinterpret_cast<const double *>(a.ws + L.freqc), flags,
                       colstate, a.out, a.nrow, a.nsrc, a.nchan, a.mode == AF_DFT_EXACT ? -1 : 2);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

template <bool CPLX>
int launch32_nc(const Args32 &a, int64_t ncorr)
{
    constexpr int CT0 = CPLX ? 15 : 16, CT1 = CPLX ? 20 : 22, CT2 = CPLX ? 30 : 32;
    const int ct = a.L->ct;
    if (ncorr == 4) return ct == CT2 ? launch32<CT2, 4, CPLX>(a) : ct == CT1 ? launch32<CT1, 4, CPLX>(a) : launch32<CT0, 4, CPLX>(a);
    if (ct != CT0) {
        af_set_error("af_im_to_vis_f32: AFHIP_F32_CT applies to 4 correlations only");
        return AF_EINVAL;
    }
    if (ncorr == 2) return launch32<CT0, 2, CPLX>(a);
    return launch32<CT0, 1, CPLX>(a);
}


// =====================================================================================================================
// vis_to_im in single precision (africanus/dft/kernels.py:104-146 with complex64 visibilities and float32 coordinates):
//     im[s,nu,c] = sum_r [no correlation of (r,nu) flagged] ( cos p Re V - sin p Im V ),  p = C (l u + m v + n w) nu
// the machinery above with rows and sources swapped: a lane owns one SOURCE and a tile of CT channels x 4 real
// accumulators (one MFMA quad per channel), the wave walks the rows of its row partition; a row's record holds the
// planes (Re V, -Im V) of the tile's channels (2 blocks per channel, 8 channels per register), flag-masked by the
// pack pass; phasors in fp64 (table + recurrence), rounded to float32 once, then
//     acc[chan] += (Re V) yr ; acc[chan] += (-Im V) yi          (two MFMAs per channel).
// Partial images per row partition are added in partition order (fp64 sums) by v2i32_reduce.
struct WsV32 {
    size_t flags, lmn, srcbad, tilef, kappa, freqc, uvw, chan_any, records, partial, total;
    int64_t ntile, npart, rows_per_part;
    int ct, groups;
};

constexpr int V32_CT = 32;

WsV32 wsv32_layout(int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr)
{
    WsV32 L;
    L.ct = V32_CT;
    L.ntile = af_cdiv(nchan > 0 ? nchan : 1, L.ct);
    L.groups = L.ct / 8;
    const int64_t nsg = af_cdiv(nsrc > 0 ? nsrc : 1, ROWS);
    int64_t npart = af_cdiv(2048, nsg * L.ntile);
    const int64_t most = af_cdiv(nrow > 0 ? nrow : 1, 256);           // at least 256 rows per partition
    if (npart > most) npart = most;
    if (npart < 1) npart = 1;
    L.rows_per_part = af_cdiv(nrow > 0 ? nrow : 1, npart);
    L.npart = af_cdiv(nrow > 0 ? nrow : 1, L.rows_per_part);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    L.flags = take(16 * sizeof(int));
    L.lmn = take((size_t)(nsrc > 0 ? nsrc : 1) * 4 * sizeof(double));
    L.srcbad = take((size_t)(nsrc > 0 ? nsrc : 1) * sizeof(int));
    L.tilef = take((size_t)L.ntile * 2 * sizeof(double));
    L.kappa = take((size_t)L.ntile * L.ct * sizeof(float));
    L.freqc = take((size_t)(nchan > 0 ? nchan : 1) * sizeof(double));
    L.uvw = take((size_t)(nrow > 0 ? nrow : 1) * 4 * sizeof(double));
    L.chan_any = take((size_t)L.ntile * L.ct * sizeof(int));
    L.records = take((size_t)L.ntile * (nrow > 0 ? nrow : 1) * L.groups * G32 * sizeof(float));
    L.partial = take((size_t)L.npart * (nsrc > 0 ? nsrc : 1) * (nchan > 0 ? nchan : 1) * 4 * sizeof(float));
    L.total = o;
    return L;
}

// (u, v, w, 0) of every row in fp64, zeros for a non-finite row (its unflagged cells become NaN in the records)
__global__ void v2i32_prep_rows(const float *__restrict__ uvw, int64_t nrow, double *__restrict__ uvw64)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    const double a = (double)uvw[3 * r], b = (double)uvw[3 * r + 1], c = (double)uvw[3 * r + 2];
    const bool ok = isfinite(a) && isfinite(b) && isfinite(c);
    uvw64[4 * r + 0] = ok ? a : 0.0; uvw64[4 * r + 1] = ok ? b : 0.0; uvw64[4 * r + 2] = ok ? c : 0.0;
    uvw64[4 * r + 3] = ok ? 0.0 : 1.0;
}

// records: [tile][row][groups * 64] floats; register g, block k = channel 8 g + k / 2 of the tile, plane k % 2 of
// (Re V, -Im V), floats = correlations 0..3.  Zero for flagged (row, chan) cells (ANY correlation flagged,
// kernels.py:139-140), channels beyond the band and correlations beyond ncorr; NaN for unflagged cells of a
// non-finite row.  chan_any[chan] = some row of the channel is unflagged.
// One lane per (tile, row, channel of the tile): its 8 floats -- (Re V) and (-Im V) of the 4 correlations -- are 32
// contiguous bytes of the record (channel j = 8 g + q sits at floats 8 j ..), its inputs one flag word and ncorr
// adjacent visibilities.  grid: (ceil(nrow * CT / 256), tiles).  (The first form had one lane per FLOAT with three
// 64-bit divisions each: 3.0 ms for C2's 4 GB at 1.5 TB/s.)
template <int CT>
__global__ __launch_bounds__(256) void v2i32_pack(const float2 *__restrict__ vis, const unsigned char *__restrict__ vflags,
                                                  const double *__restrict__ uvw64, int64_t nrow, int64_t nchan, int ncorr,
                                                  int groups, float *__restrict__ rec, int *__restrict__ chan_any)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = e / CT;
    const int j = (int)(e - r * CT), tile = blockIdx.y;
    if (r >= nrow) return;
    const int64_t ch = (int64_t)tile * CT + j;
    v4f re = {0.0f, 0.0f, 0.0f, 0.0f}, im = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ch < nchan) {
        const int64_t cell = (r * nchan + ch) * ncorr;
        bool flagged = false;
        for (int k = 0; k < ncorr; ++k) flagged |= vflags[cell + k] != 0;
        if (!flagged) {
            chan_any[ch] = 1;                                  // benign race: all writers store 1
            const bool bad_row = uvw64[4 * r + 3] != 0.0;
            for (int c = 0; c < ncorr; ++c) {
                const float2 x = vis[cell + c];
                re[c] = bad_row ? __builtin_nanf("") : x.x;
                im[c] = bad_row ? __builtin_nanf("") : -x.y;
            }
        }
    }
    v4f *o = reinterpret_cast<v4f *>(rec + ((int64_t)tile * nrow + r) * ((int64_t)groups * G32) + 8 * j);
    o[0] = re;
    o[1] = im;
}

// grid: (ceil(nsrc / 256), tiles, row partitions); block 256 = 4 waves of 64 consecutive sources
// CHAIN: phasors as in dft_f32_kernel<..., CHAIN> (float32 rotation recurrence from the tile's middle channel).
template <int CT, bool CORR, bool CHAIN>
__global__ __launch_bounds__(ROWS, 2) void v2i_f32_kernel(
    const double *__restrict__ lmn, const double *__restrict__ uvw64, const float *__restrict__ records,
    const double *__restrict__ tilef, const float *__restrict__ kappa, const int *__restrict__ flags,
    float *__restrict__ partial, int64_t nsrc, int64_t nrow, int64_t rows_per_part, int64_t nchan, int want_class,
    int force_chain)
{
    if (flags[0] != want_class) return;
    if ((force_chain >= 0 ? force_chain != 0 : chain_regime(flags)) != CHAIN) return;
    __shared__ double2 ptab[CHAIN ? 1 : PHASOR_TABLE];
    __shared__ float2 ftab[CHAIN ? CHAIN_TABLE : 1];
    if constexpr (CHAIN) chain_table_init(ftab, threadIdx.x, ROWS);
    else table_phasor_init(ptab, threadIdx.x, ROWS);
    __syncthreads();
    constexpr int NG = CT / 8;
    const int tile = blockIdx.y;
    const int64_t c0 = (int64_t)tile * CT;
    int64_t src = (int64_t)blockIdx.x * ROWS + threadIdx.x;
    const bool valid = src < nsrc;
    if (!valid) src = nsrc - 1;
    const double l = lmn[4 * src], m = lmn[4 * src + 1], n = lmn[4 * src + 2];
    const double F0 = 256.0 * tilef[2 * tile], FD = 256.0 * tilef[2 * tile + 1];
    // chain: the anchor channel's frequency and the step in 1/1024 turns per metre
    constexpr int JA = CT / 2;
    const double FA = (double)CHAIN_TABLE * fma((double)JA, tilef[2 * tile + 1], tilef[2 * tile]);
    const double FS = (double)CHAIN_TABLE * tilef[2 * tile + 1];
    float kap[CT + 1];
    if constexpr (CORR) {
#pragma unroll
        for (int j = 0; j < CT; ++j) kap[j] = kappa[tile * CT + j];
        kap[CT] = 0.0f;
    }
    v4f acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    const int64_t r0 = (int64_t)blockIdx.z * rows_per_part;
    const int64_t r1 = r0 + rows_per_part < nrow ? r0 + rows_per_part : nrow;
    const int lane = threadIdx.x & 63;
    const float *__restrict__ rec = records + (int64_t)tile * nrow * (NG * G32) + lane;
    float R[NG], Rn[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) R[g] = rec[r0 * (NG * G32) + g * G32];

#pragma unroll 1
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t rn = r + 1 < r1 ? r + 1 : r;
#pragma unroll
        for (int g = 0; g < NG; ++g) Rn[g] = rec[rn * (NG * G32) + g * G32];
        const double u = uvw64[4 * r], v = uvw64[4 * r + 1], w = uvw64[4 * r + 2];   // wave-uniform: scalar loads
        const double q = fma(n, w, fma(m, v, __dmul_rn(l, u)));
        const float qf = (float)q;
        float yr[CT], yi[CT];
        if constexpr (CHAIN) {
            v2f y[CT], st;
            {
                float ar, ai, sr, si;
                chain_phasor(ftab, __dmul_rn(q, FA), ar, ai);
                chain_phasor(ftab, __dmul_rn(q, FS), sr, si);
                y[JA] = v2f{ar, ai};
                st = v2f{sr, si};
            }
            static_for32<1, JA + 1>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (JA + k < CT) rot_both(y[JA + k - 1], y[JA - k + 1], st, y[JA + k], y[JA - k]);
                else y[JA - k] = rot_down(y[JA - k + 1], st);
            });
            if constexpr (CORR) {
                const v2f q2 = {qf, qf};
                static_for32<0, (CT + 1) / 2>([&](auto jc) {
                    constexpr int j = 2 * decltype(jc)::value;
                    const v2f th = theta_pair(q2, v2f{kap[j], kap[j + 1]});
                    y[j] = first_order<0>(y[j], th);
                    if constexpr (j + 1 < CT) y[j + 1] = first_order<1>(y[j + 1], th);
                });
            }
#pragma unroll
            for (int j = 0; j < CT; ++j) { yr[j] = y[j].x; yi[j] = y[j].y; }
            asm volatile("s_nop 1" ::: "memory");   // VALU write -> MFMA read: two wait states (see dft_f32_kernel)
        } else {
            double y0r, y0i, dr, di;
            table_phasor(ptab, __dmul_rn(q, F0), y0r, y0i);
            table_phasor(ptab, __dmul_rn(q, FD), dr, di);
            const double k2 = __dadd_rn(dr, dr);
            double y1r = fma(y0r, dr, -__dmul_rn(y0i, di));
            double y1i = fma(y0r, di, __dmul_rn(y0i, dr));
            static_for32<0, CT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                double yrd, yid;
                if constexpr (j == 0) { yrd = y0r; yid = y0i; }
                else if constexpr (j == 1) { yrd = y1r; yid = y1i; }
                else {
                    yrd = fma(k2, y1r, -y0r);
                    yid = fma(k2, y1i, -y0i);
                    y0r = y1r; y0i = y1i; y1r = yrd; y1i = yid;
                }
                yr[j] = (float)yrd; yi[j] = (float)yid;
                if constexpr (CORR) {
                    const float th = qf * kap[j];
                    const float cr = fmaf(-th, yi[j], yr[j]), ci = fmaf(th, yr[j], yi[j]);
                    yr[j] = cr; yi[j] = ci;
                }
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for32<0, CT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(R[j / 8], yr[j], acc[j], 4, (j % 8) * 2, 0);        // + Re V cos
        });
        static_for32<0, CT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(R[j / 8], yi[j], acc[j], 4, (j % 8) * 2 + 1, 0);    // - Im V sin
        });
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) R[g] = Rn[g];
    }
    if (!valid) return;
    float *__restrict__ o = partial + (((int64_t)blockIdx.z * nsrc + src) * nchan + c0) * 4;
#pragma unroll
    for (int j = 0; j < CT; ++j)
        if (c0 + j < nchan) *reinterpret_cast<float4 *>(o + j * 4) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
}

// any band: one fp64 sincos per (source, row, channel).  grid: (ceil(nsrc / 256), nchan, row partitions)
__global__ __launch_bounds__(ROWS) void v2i_f32_exact_kernel(
    const double *__restrict__ lmn, const double *__restrict__ uvw64, const float2 *__restrict__ vis,
    const unsigned char *__restrict__ vflags, const double *__restrict__ freqc, const int *__restrict__ flags,
    float *__restrict__ partial, int64_t nsrc, int64_t nrow, int64_t rows_per_part, int64_t nchan, int ncorr,
    int want_class)
{
    if (want_class >= 0 && flags[0] != want_class) return;
    const int64_t ch = blockIdx.y;
    int64_t src = (int64_t)blockIdx.x * ROWS + threadIdx.x;
    const bool valid = src < nsrc;
    if (!valid) src = nsrc - 1;
    const double l = lmn[4 * src], m = lmn[4 * src + 1], n = lmn[4 * src + 2];
    const double fc = freqc[ch];
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int64_t r0 = (int64_t)blockIdx.z * rows_per_part;
    const int64_t r1 = r0 + rows_per_part < nrow ? r0 + rows_per_part : nrow;
    for (int64_t r = r0; r < r1; ++r) {
        const unsigned char *f = vflags + (r * nchan + ch) * ncorr;
        bool flagged = false;
        for (int k = 0; k < ncorr; ++k) flagged |= f[k] != 0;
        if (flagged) continue;   // wave-uniform
        const double u = uvw64[4 * r], v = uvw64[4 * r + 1], w = uvw64[4 * r + 2];
        const bool badrow = uvw64[4 * r + 3] != 0.0;
        const double q = fma(n, w, fma(m, v, __dmul_rn(l, u)));
        double yrd, yid;
        sincos_quarter_turns<7>(__dmul_rn(q, fc), yrd, yid);
        const float yr = badrow ? __builtin_nanf("") : (float)yrd, yi = badrow ? __builtin_nanf("") : (float)yid;
        for (int c = 0; c < ncorr; ++c) {
            const float2 x = vis[(r * nchan + ch) * ncorr + c];
            acc[c] = fmaf(x.x, yr, fmaf(-x.y, yi, acc[c]));
        }
    }
    if (!valid) return;
    float *__restrict__ o = partial + (((int64_t)blockIdx.z * nsrc + src) * nchan + ch) * 4;
    *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// out[s, chan, c] = sum over the row partitions, in partition order (fp64 sums); exactly 0 for a channel without an
// unflagged row; NaN for a source outside the unit disc (p = NaN in the reference) where the channel has data
__global__ void v2i32_reduce(const float *__restrict__ partial, const int *__restrict__ chan_any,
                             const int *__restrict__ srcbad, int64_t npart, int64_t nsrc, int64_t nchan, int ncorr,
                             float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (source, chan, corr)
    if (i >= nsrc * nchan * ncorr) return;
    const int c = (int)(i % ncorr);
    const int64_t ch = (i / ncorr) % nchan, s = i / (ncorr * nchan);
    double sum = 0.0;
    for (int64_t p = 0; p < npart; ++p) sum += (double)partial[((p * nsrc + s) * nchan + ch) * 4 + c];
    float v = (float)sum;
    if (!chan_any[ch]) v = 0.0f;
    else if (srcbad[s]) v = __builtin_nanf("");
    out[i] = v;
}

}  // namespace

AF_EXPORT size_t af_im_to_vis_f32_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t ncorr, int image_is_complex)
{
    if (nsrc < 0 || nchan < 0 || ncorr < 0) return 0;
    return ws32_layout(nsrc, nchan, ncorr, image_is_complex).total;
}

AF_EXPORT int af_im_to_vis_f32(const float *image, int image_is_complex, const float *uvw, const float *lm,
                               const float *frequency, int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr,
                               int convention, int mode, float *out, void *workspace, size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(convention == AF_CONVENTION_FOURIER || convention == AF_CONVENTION_CASA,
               "convention not in ('fourier', 'casa')");
    const int clamp_n = (mode & AF_DFT_CLAMP_N) ? 1 : 0;
    mode &= ~AF_DFT_CLAMP_N;
    AF_REQUIRE(mode == AF_DFT_AUTO || mode == AF_DFT_EXACT || mode == AF_DFT_RECURRENCE, "af_im_to_vis_f32: unknown mode %d", mode);
    AF_REQUIRE(nsrc >= 0 && nrow >= 0 && nchan >= 0 && ncorr >= 0, "af_im_to_vis_f32: negative extent");
    AF_REQUIRE(nsrc < (1LL << 31), "af_im_to_vis_f32: nsrc too large");
    if (!(ncorr == 0 || ncorr == 1 || ncorr == 2 || ncorr == 4)) {
        af_set_error("af_im_to_vis_f32: %lld correlations (1, 2 or 4 have a float32 kernel; promote to float64 for others)",
                     (long long)ncorr);
        return AF_ENOTSUP;
    }
    hipStream_t st = af_stream(stream);
    if (nrow == 0 || nchan == 0 || ncorr == 0) return AF_OK;
    AF_REQUIRE(out != nullptr && uvw != nullptr && frequency != nullptr, "af_im_to_vis_f32: NULL array");
    if (nsrc == 0) {   // np.zeros output (kernels.py:45)
        AF_HIP(hipMemsetAsync(out, 0, sizeof(float) * 2 * (size_t)(nrow * nchan * ncorr), st));
        return AF_OK;
    }
    AF_REQUIRE(image != nullptr && lm != nullptr, "af_im_to_vis_f32: NULL array");
    const Ws32 L = ws32_layout(nsrc, nchan, ncorr, image_is_complex);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= L.total, "af_im_to_vis_f32: workspace too small (%zu < %zu)",
               workspace_bytes, L.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_im_to_vis_f32: workspace must be 256-byte aligned");
    AF_REQUIRE(L.ntile <= 65535 && nchan <= 65535, "af_im_to_vis_f32: too many channels");
    char *ws = static_cast<char *>(workspace);
    const int W = image_is_complex ? 2 : 1;
    AF_HIP(hipMemsetAsync(ws + L.flags, 0, 16 * sizeof(int), st));
    AF_HIP(hipMemsetAsync(ws + L.tilestate, 0, (size_t)L.ntile * sizeof(int), st));
    hipLaunchKernelGGL(f32_prep_src, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st, lm, nsrc,
                       reinterpret_cast<double *>(ws + L.lmn), reinterpret_cast<int *>(ws + L.srcbad),
                       reinterpret_cast<int *>(ws + L.flags), clamp_n);
    AF_LAUNCH_CHECK();
    {
        int64_t blocks = af_cdiv(nrow, 256 * 16);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(f32_rows_reach, dim3((unsigned)blocks), dim3(256), 0, st, uvw, nrow, reinterpret_cast<int *>(ws + L.flags));
        AF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(f32_prep_freq, dim3((unsigned)af_cdiv(L.ntile, 64)), dim3(64), 0, st, frequency, nchan, L.ntile, L.ct,
                       convention, reinterpret_cast<double *>(ws + L.tilef), reinterpret_cast<float *>(ws + L.kappa),
                       reinterpret_cast<double *>(ws + L.freqc), reinterpret_cast<int *>(ws + L.flags));
    AF_LAUNCH_CHECK();
    if (mode == AF_DFT_RECURRENCE)   // the caller asserts a uniform band: the grid of every tile, no correction
        AF_HIP(hipMemsetAsync(ws + L.flags, 0, sizeof(int), st));
    {
        const int64_t total = L.ntile * nsrc * (int64_t)L.groups * G32;
        int64_t blocks = af_cdiv(total, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL((f32_pack_records<float>), dim3((unsigned)blocks), dim3(256), 0, st, image, image_is_complex ? 1 : 0, nsrc, nchan, ncorr,
                           L.ntile, L.ct, L.groups, reinterpret_cast<const int *>(ws + L.srcbad),
                           reinterpret_cast<float *>(ws + L.records));
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL(f32_colstate, dim3((unsigned)(L.ntile * L.ct * ncorr)), dim3(64), 0, st, image, W, nsrc, nchan,
                           ncorr, L.ct, reinterpret_cast<const int *>(ws + L.srcbad), reinterpret_cast<int *>(ws + L.colstate),
                           reinterpret_cast<int *>(ws + L.tilestate));
        AF_LAUNCH_CHECK();
    }
    Args32 a;
    a.uvw = uvw; a.image = image; a.ws = ws; a.L = &L; a.out = out; a.nrow = nrow; a.nchan = nchan; a.nsrc = (int)nsrc;
    a.mode = mode; a.st = st;
    return image_is_complex ? launch32_nc<true>(a, ncorr) : launch32_nc<false>(a, ncorr);
}

AF_EXPORT size_t af_vis_to_im_f32_workspace_bytes(int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr)
{
    if (nsrc < 0 || nrow < 0 || nchan < 0 || ncorr < 0) return 0;
    return wsv32_layout(nsrc, nrow, nchan, ncorr).total;
}

AF_EXPORT int af_vis_to_im_f32(const float *vis, const float *uvw, const float *lm, const float *frequency,
                               const unsigned char *flags, int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr,
                               int convention, int mode, float *out, void *workspace, size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(convention == AF_CONVENTION_FOURIER || convention == AF_CONVENTION_CASA,
               "convention not in ('fourier', 'casa')");
    AF_REQUIRE(mode == AF_DFT_AUTO || mode == AF_DFT_EXACT || mode == AF_DFT_RECURRENCE, "af_vis_to_im_f32: unknown mode %d", mode);
    AF_REQUIRE(nsrc >= 0 && nrow >= 0 && nchan >= 0 && ncorr >= 0, "af_vis_to_im_f32: negative extent");
    if (!(ncorr == 0 || ncorr == 1 || ncorr == 2 || ncorr == 4)) {
        af_set_error("af_vis_to_im_f32: %lld correlations (1, 2 or 4 have a float32 kernel; promote to float64 for others)",
                     (long long)ncorr);
        return AF_ENOTSUP;
    }
    hipStream_t st = af_stream(stream);
    if (nsrc == 0 || nchan == 0 || ncorr == 0) return AF_OK;
    AF_REQUIRE(out != nullptr, "af_vis_to_im_f32: out is NULL");
    if (nrow == 0) {   // np.zeros output (kernels.py:96)
        AF_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)(nsrc * nchan * ncorr), st));
        return AF_OK;
    }
    AF_REQUIRE(vis && uvw && lm && frequency && flags, "af_vis_to_im_f32: NULL array");
    const WsV32 L = wsv32_layout(nsrc, nrow, nchan, ncorr);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= L.total, "af_vis_to_im_f32: workspace too small (%zu < %zu)",
               workspace_bytes, L.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_vis_to_im_f32: workspace must be 256-byte aligned");
    AF_REQUIRE(L.ntile <= 65535 && nchan <= 65535 && L.npart <= 65535, "af_vis_to_im_f32: too many channels / partitions");
    char *ws = static_cast<char *>(workspace);
    int *wflags = reinterpret_cast<int *>(ws + L.flags), *chan_any = reinterpret_cast<int *>(ws + L.chan_any);
    double *lmn = reinterpret_cast<double *>(ws + L.lmn), *uvw64 = reinterpret_cast<double *>(ws + L.uvw);
    int *srcbad = reinterpret_cast<int *>(ws + L.srcbad);
    float *rec = reinterpret_cast<float *>(ws + L.records), *partial = reinterpret_cast<float *>(ws + L.partial);
    AF_HIP(hipMemsetAsync(wflags, 0, 16 * sizeof(int), st));
    AF_HIP(hipMemsetAsync(chan_any, 0, (size_t)L.ntile * L.ct * sizeof(int), st));
    hipLaunchKernelGGL(f32_prep_src, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st, lm, nsrc, lmn, srcbad, wflags, 0);
    AF_LAUNCH_CHECK();
    // vis_to_im's 'fourier' is exp(+2 pi i ...): the opposite sign of im_to_vis (kernels.py:113-118)
    hipLaunchKernelGGL(f32_prep_freq, dim3((unsigned)af_cdiv(L.ntile, 64)), dim3(64), 0, st, frequency, nchan, L.ntile, L.ct,
                       -convention, reinterpret_cast<double *>(ws + L.tilef), reinterpret_cast<float *>(ws + L.kappa),
                       reinterpret_cast<double *>(ws + L.freqc), wflags);
    AF_LAUNCH_CHECK();
    if (mode == AF_DFT_RECURRENCE) AF_HIP(hipMemsetAsync(wflags, 0, sizeof(int), st));
    hipLaunchKernelGGL(v2i32_prep_rows, dim3((unsigned)af_cdiv(nrow, 256)), dim3(256), 0, st, uvw, nrow, uvw64);
    AF_LAUNCH_CHECK();
    {
        int64_t blocks = af_cdiv(nrow, 256 * 16);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(f32_rows_reach, dim3((unsigned)blocks), dim3(256), 0, st, uvw, nrow, wflags);
        AF_LAUNCH_CHECK();
    }
    {
        static_assert(V32_CT * 8 == (V32_CT / 8) * G32, "a channel's 8 floats tile the record");
        AF_REQUIRE(af_cdiv(nrow * V32_CT, 256) < (1LL << 31), "af_vis_to_im_f32: too many rows");
        hipLaunchKernelGGL((v2i32_pack<V32_CT>), dim3((unsigned)af_cdiv(nrow * V32_CT, 256), (unsigned)L.ntile), dim3(256), 0, st,
                           reinterpret_cast<const float2 *>(vis), flags, uvw64, nrow, nchan, (int)ncorr, L.groups, rec, chan_any);
        AF_LAUNCH_CHECK();
    }
    const double *tilef = reinterpret_cast<const double *>(ws + L.tilef);
    const float *kappa = reinterpret_cast<const float *>(ws + L.kappa);
    const dim3 block(ROWS);
    const unsigned nsg = (unsigned)af_cdiv(nsrc, ROWS);
    if (mode != AF_DFT_EXACT) {
        const dim3 grid(nsg, (unsigned)L.ntile, (unsigned)L.npart);
        af_prof_begin(st);
        static const int force = getenv("AFHIP_F32_CHAIN") ? atoi(getenv("AFHIP_F32_CHAIN")) : -1;
        hipLaunchKernelGGL((v2i_f32_kernel<V32_CT, false, true>), grid, block, 0, st, lmn, uvw64, rec, tilef, kappa, wflags, partial,
                           nsrc, nrow, L.rows_per_part, nchan, 0, force);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL((v2i_f32_kernel<V32_CT, true, true>), grid, block, 0, st, lmn, uvw64, rec, tilef, kappa, wflags, partial,
                           nsrc, nrow, L.rows_per_part, nchan, 1, force);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL((v2i_f32_kernel<V32_CT, false, false>), grid, block, 0, st, lmn, uvw64, rec, tilef, kappa, wflags, partial,
                           nsrc, nrow, L.rows_per_part, nchan, 0, force);
        AF_LAUNCH_CHECK();
        hipLaunchKernelGGL((v2i_f32_kernel<V32_CT, true, false>), grid, block, 0, st, lmn, uvw64, rec, tilef, kappa, wflags, partial,
                           nsrc, nrow, L.rows_per_part, nchan, 1, force);
        af_prof_end(st);
        AF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(v2i_f32_exact_kernel, dim3(nsg, (unsigned)nchan, (unsigned)L.npart), block, 0, st, lmn, uvw64,
                       reinterpret_cast<const float2 *>(vis), flags, reinterpret_cast<const double *>(ws + L.freqc), wflags, partial,
                       nsrc, nrow, L.rows_per_part, nchan, (int)ncorr, mode == AF_DFT_EXACT ? -1 : 2);
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL(v2i32_reduce, dim3((unsigned)af_cdiv(nsrc * nchan * ncorr, 256)), dim3(256), 0, st, partial, chan_any, srcbad,
                       L.npart, nsrc, nchan, (int)ncorr, out);
    AF_LAUNCH_CHECK();
    return AF_OK;
}
