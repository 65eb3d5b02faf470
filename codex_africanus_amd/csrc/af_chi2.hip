// chi-squared of model visibilities against data: chi2[nu] = sum_{r,c} w[r,nu,c] |data - model|^2.
//
// The reference has no chi^2 function ("parity unpinned"; its natural home
// is africanus/calibration/utils/residual_vis.py:63); it is defined here because the
// row-sharded multi-GPU predict (BASELINE config 4) reduces exactly this quantity across
// GPUs with one RCCL all-reduce of an (nchan,) vector.  HBM-bound: reads model and data
// once (2 x 16 B per cell), fully coalesced; per-lane partial sums in registers over the
// rows of the block, one double atomicAdd per (block, column).
#include "af_common.h"

namespace {

// grid: blocks over row ranges; block: 256 lanes striding the (chan*corr) columns of a row
__global__ __launch_bounds__(256) void chi2_kernel(const double2 *__restrict__ model,
                                                   const double2 *__restrict__ data,
                                                   const double *__restrict__ weight, int64_t nrow,
                                                   int64_t nchan, int64_t ncorr, int64_t rows_per_block,
                                                   double *__restrict__ chi2)
{
    const int64_t ncol = nchan * ncorr;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < nrow) ? r0 + rows_per_block : nrow;
    for (int64_t col = threadIdx.x; col < ncol; col += blockDim.x) {
        double acc = 0.0;
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t i = r * ncol + col;
            const double2 m = model[i], d = data[i];
            const double dr = d.x - m.x, di = d.y - m.y;
            const double a = fma(dr, dr, di * di);
            acc = weight ? fma(weight[i], a, acc) : acc + a;
        }
        atomicAdd(&chi2[col / ncorr], acc);
    }
}

}  // namespace

AF_EXPORT int af_chi2_c128(const double *model, const double *data, const double *weight, int64_t nrow,
                           int64_t nchan, int64_t ncorr, double *chi2_per_chan, void *stream)
{
    AF_REQUIRE(nrow >= 0 && nchan >= 0 && ncorr >= 0, "af_chi2_c128: negative extent");
    if (nchan == 0) return AF_OK;
    AF_REQUIRE(chi2_per_chan != nullptr, "af_chi2_c128: chi2_per_chan is NULL");
    hipStream_t st = af_stream(stream);
    AF_HIP(hipMemsetAsync(chi2_per_chan, 0, sizeof(double) * (size_t)nchan, st));
    if (nrow == 0 || ncorr == 0) return AF_OK;
    AF_REQUIRE(model && data, "af_chi2_c128: NULL array");
    // ~8 blocks per CU on a 256-CU part, at least 8 rows per block
    int64_t rows_per_block = af_cdiv(nrow, 2048);
    if (rows_per_block < 8) rows_per_block = 8;
    const int64_t blocks = af_cdiv(nrow, rows_per_block);
    hipLaunchKernelGGL(chi2_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const double2 *>(model), reinterpret_cast<const double2 *>(data), weight,
                       nrow, nchan, ncorr, rows_per_block, chi2_per_chan);
    AF_LAUNCH_CHECK();
    return AF_OK;
}
