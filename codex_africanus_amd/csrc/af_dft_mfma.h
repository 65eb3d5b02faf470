// Internal interface of the MFMA-accumulator im_to_vis path (af_im_to_vis_mfma.hip), used by
// af_im_to_vis.hip's dispatch.  Not part of the C ABI.
#pragma once
#include "af_common.h"

// exactly 4 correlations (real or complex pixels), enough channels to fill a 16-channel tile
bool af_dft_mfma_eligible(int64_t nchan, int64_t ncorr, bool image_is_complex);

// bytes of workspace the path needs behind the common prep arrays (256-byte aligned start)
size_t af_dft_mfma_workspace_bytes(int64_t nsrc_pad, int64_t nchan, bool image_is_complex, bool gauss = false);

// Packs the records and launches the kernels on `st`.  The kernels do the work iff the prep pass
// found one channel spacing for the whole band (flags[0] == 1 and flags[1] == 1); otherwise they
// return at once and the caller's fallback launches cover the band.
//   lmn     (nsrc,4)  cleaned (l, m, n, 0) of dft_prep_src         srcbad (nsrc) non-finite sources
//   tilef   [1] = channel step in quarter turns per metre           flags  device flags, [2] = any special column
//   colstate (chan, 4) zero-column / NaN-source overrides of the reference's `if image[s,nu,c]:`
// chi2 (optional): the kernels' epilogues add  sum_{row, corr} [w] |data - vis|^2  of their rows to chi2[chan]
// (af_im_to_vis_chi2_f64: the visibilities are still in registers there; a separate pass would read them back).
struct AfDftChi2 {
    const double *data;      // (nrow, nchan, 4) complex128
    const double *weight;    // (nrow, nchan, 4) real, or NULL
    double *chi2;            // (nchan), zeroed by the caller
};
int af_dft_mfma_run(const double *image, int image_is_complex, const double *uvw, const double *frequency, const double *lmn,
                    const int *srcbad, const double *tilef, const int *flags, const int *colstate, int sign,
                    double *out, int64_t nrow, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, void *workspace,
                    hipStream_t st, const AfDftChi2 *chi2 = nullptr);

// The same kernels with a Gaussian envelope folded into every lane's phasor (Gaussian / point sources without DDEs,
// af_gauss_predict_c128): brightness (nsrc, nchan, 4) complex128 for the image, gauss (nsrc, 4) = (el, em, er, -) scaled to
// the kernels' 1/256-turn frequency units; workspace: af_dft_mfma_workspace_bytes(nsrc_pad, nchan, true, true).
int af_gauss_mfma_run(const double *brightness, const double *gauss, const double *uvw, const double *frequency,
                      const double *lmn, const int *srcbad, const double *tilef, const int *flags, int sign, double *out,
                      int64_t nrow, int64_t nsrc, int64_t nsrc_pad, int64_t nchan, void *workspace, hipStream_t st,
                      const AfDftChi2 *chi2 = nullptr);
