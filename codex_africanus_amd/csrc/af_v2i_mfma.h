// Internal interface of the MFMA-accumulator vis_to_im path (af_vis_to_im_mfma.hip), used by
// af_vis_to_im.hip's dispatch.  Not part of the C ABI.
#pragma once
#include "af_common.h"

// exactly 4 correlations, enough channels to fill a 16-channel tile
bool af_v2i_mfma_eligible(int64_t nchan, int64_t ncorr);

// bytes of the record region (shared with the VALU kernels' records: only one of the two is built)
size_t af_v2i_mfma_workspace_bytes(int64_t nrow, int64_t nchan);

// Builds tile constants, packs the records and launches the kernels on `st`.  flags[3] (preset to 1 by the
// caller) is cleared on the device when some 64-channel tile is not uniformly spaced; pack and kernels then
// return at once and the caller's VALU launches (which test flags[3] == 0) do the work.
//   lmn (nsrc,4) from v2i_prep_src; chan_any (chan) set where a channel has an unflagged row;
//   partial [part][src][chan][4] partial images, rows_per_part a multiple of 4.
int af_v2i_mfma_run(const double *vis, const unsigned char *vflags, const double *uvw, const double *frequency,
                    const double *lmn, int *flags, int *chan_any, int sign, double *partial, int64_t nsrc,
                    int64_t nrow, int64_t nchan, int64_t npart, int64_t rows_per_part, int force_uniform,
                    void *workspace, hipStream_t st);
