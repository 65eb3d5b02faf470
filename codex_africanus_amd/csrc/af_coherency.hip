// Stokes <-> correlation conversion for gfx950 (the fourth term producer):
//   africanus/model/coherency/conversion.py:18-48 (the eight Stokes->corr and eight corr->Stokes products),
//   :207-216 (one product per output element, written into the output's schema position).
// The schema resolution (which two inputs feed which output) is host logic; the kernel applies a table of at most
// 12 products.  HBM-bound: one lane per (element, output) so that a wave's stores are contiguous; the two operands
// of a lane lie in the same <= 192-byte input element as its neighbours' and come out of L1/L2.
#include "af_common.h"

namespace {

struct ConvTable {
    unsigned long long src1, src2;  // 4 bits per output: input position + 1, 0 = the implicit Stokes default (0)
    unsigned long long op;          // 4 bits per output: AF_CONV_*
};

template <typename T> struct Cx { T r, i; };

// numpy's complex product and quotient loops, as the reference's lambdas reach them
template <typename T> __device__ __forceinline__ Cx<T> times_j(Cx<T> b)   // b * 1j
{
    return {b.r * (T)0 - b.i * (T)1, b.r * (T)1 + b.i * (T)0};
}
template <typename T> __device__ __forceinline__ Cx<T> over_two(Cx<T> a)   // a / 2
{
    const T rat = (T)0 / (T)2, scl = (T)1 / ((T)2 + (T)0 * rat);
    return {(a.r + a.i * rat) * scl, (a.i - a.r * rat) * scl};
}
template <typename T> __device__ __forceinline__ Cx<T> over_two_j(Cx<T> a)   // a / 2j
{
    const T rat = (T)0 / (T)2, scl = (T)1 / ((T)2 + (T)0 * rat);
    return {(a.r * rat + a.i) * scl, (a.i * rat - a.r) * scl};
}

template <typename T, bool CIN, bool COUT>
__global__ __launch_bounds__(256) void coherency_convert_kernel(const T *__restrict__ in, int64_t total, int nin,
                                                                int nout, ConvTable tab, T *__restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int64_t e = idx / nout;
    const int o = (int)(idx - e * nout);
    const int s1 = (int)((tab.src1 >> (4 * o)) & 15), s2 = (int)((tab.src2 >> (4 * o)) & 15);
    const int op = (int)((tab.op >> (4 * o)) & 15);
    constexpr int W = CIN ? 2 : 1;
    const T *x = in + e * nin * W;
    Cx<T> a = {(T)0, (T)0}, b = {(T)0, (T)0};
    if (s1) { a.r = x[(s1 - 1) * W]; if constexpr (CIN) a.i = x[(s1 - 1) * W + 1]; }
    if (s2) { b.r = x[(s2 - 1) * W]; if constexpr (CIN) b.i = x[(s2 - 1) * W + 1]; }
    Cx<T> y;
    switch (op) {
    case AF_CONV_ADD: y = {(a.r + b.r) + (T)0, (a.i + b.i) + (T)0}; break;
    case AF_CONV_SUB: y = {(a.r - b.r) + (T)0, (a.i - b.i) + (T)0}; break;
    case AF_CONV_ADDJ: { const Cx<T> t = times_j(b); y = {a.r + t.r, a.i + t.i}; } break;
    case AF_CONV_SUBJ: { const Cx<T> t = times_j(b); y = {a.r - t.r, a.i - t.i}; } break;
    case AF_CONV_HALF_ADD:
        if constexpr (CIN) y = over_two(Cx<T>{a.r + b.r, a.i + b.i});
        else y = {(a.r + b.r) / (T)2, (T)0};
        break;
    case AF_CONV_HALF_SUB:
        if constexpr (CIN) y = over_two(Cx<T>{a.r - b.r, a.i - b.i});
        else y = {(a.r - b.r) / (T)2, (T)0};
        break;
    default: y = over_two_j(Cx<T>{a.r - b.r, a.i - b.i}); break;  // AF_CONV_HALF_SUB_OVER_J
    }
    if constexpr (COUT) {
        using T2 = typename std::conditional<sizeof(T) == 8, double2, float2>::type;
        T2 v; v.x = y.r; v.y = y.i;
        reinterpret_cast<T2 *>(out)[idx] = v;
    } else {
        out[idx] = y.r;
    }
}

template <typename T>
int launch(const void *in, bool cin, bool cout, int64_t nelem, int nin, int nout, ConvTable tab, void *out,
           hipStream_t st)
{
    const int64_t total = nelem * nout;
    if (total == 0) return AF_OK;
    const dim3 grid((unsigned)af_cdiv(total, 256)), block(256);
    if (cin)
        hipLaunchKernelGGL((coherency_convert_kernel<T, true, true>), grid, block, 0, st, (const T *)in, total, nin, nout,
                           tab, (T *)out);
    else if (cout)
        hipLaunchKernelGGL((coherency_convert_kernel<T, false, true>), grid, block, 0, st, (const T *)in, total, nin,
                           nout, tab, (T *)out);
    else
        hipLaunchKernelGGL((coherency_convert_kernel<T, false, false>), grid, block, 0, st, (const T *)in, total, nin,
                           nout, tab, (T *)out);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

}  // namespace

AF_EXPORT int af_coherency_convert(const void *input, int in_kind, int64_t nelem, int nin, int nout,
                                   const int *src1_host, const int *src2_host, const int *op_host, void *out,
                                   int out_kind, void *stream)
{
    AF_REQUIRE(in_kind >= AF_KIND_F32 && in_kind <= AF_KIND_C128 && out_kind >= AF_KIND_F32 && out_kind <= AF_KIND_C128,
               "af_coherency_convert: unknown dtype kind (%d -> %d)", in_kind, out_kind);
    AF_REQUIRE((in_kind & 1) == (out_kind & 1), "af_coherency_convert: input and output precisions differ (%d -> %d)",
               in_kind, out_kind);
    const bool cin = in_kind >= AF_KIND_C64, cout = out_kind >= AF_KIND_C64;
    AF_REQUIRE(!cin || cout, "af_coherency_convert: complex input needs a complex output");
    AF_REQUIRE(nelem >= 0 && nin >= 1 && nin <= 12 && nout >= 1 && nout <= 12,
               "af_coherency_convert: bad extents (nelem=%lld, nin=%d, nout=%d; schemas hold at most 12 names)",
               (long long)nelem, nin, nout);
    AF_REQUIRE(src1_host && src2_host && op_host && (nelem == 0 || (input && out)), "af_coherency_convert: NULL argument");
    ConvTable tab = {0, 0, 0};
    for (int o = 0; o < nout; ++o) {
        const int a = src1_host[o], b = src2_host[o], op = op_host[o];
        AF_REQUIRE(a >= -1 && a < nin && b >= -1 && b < nin, "af_coherency_convert: output %d reads input %d / %d of %d",
                   o, a, b, nin);
        AF_REQUIRE(op >= AF_CONV_ADD && op <= AF_CONV_HALF_SUB_OVER_J, "af_coherency_convert: output %d: unknown product %d",
                   o, op);
        AF_REQUIRE(cout || op == AF_CONV_HALF_ADD || op == AF_CONV_HALF_SUB,
                   "af_coherency_convert: output %d: product %d is complex, the output is real", o, op);
        tab.src1 |= (unsigned long long)(a + 1) << (4 * o);
        tab.src2 |= (unsigned long long)(b + 1) << (4 * o);
        tab.op |= (unsigned long long)op << (4 * o);
    }
    hipStream_t st = (hipStream_t)stream;
    if (in_kind & 1) return launch<double>(input, cin, cout, nelem, nin, nout, tab, out, st);
    return launch<float>(input, cin, cout, nelem, nin, nout, tab, out, st);
}
