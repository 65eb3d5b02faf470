/*
 * afhip.h -- C ABI of libafhip.so: the MI355X (gfx950) implementation of
 * codex-africanus' RIME visibility-predict hot path.
 *
 * Every entry point replaces one function of the reference (file:line given
 * per declaration, paths relative to the reference tree).  The ABI is plain C:
 * raw pointers, explicit extents, an int status.  Unless a declaration says
 * otherwise, data pointers are DEVICE pointers (HBM) and `stream` is a
 * hipStream_t passed as void* (NULL = the default stream); calls are
 * asynchronous with respect to the host and never synchronise the device.
 * Complex arrays are interleaved (re, im) pairs of the stated real type;
 * all arrays are C-contiguous with the reference's axis order.
 *
 * Status: 0 = AF_OK, AF_EINVAL = bad argument, AF_ENOMEM, AF_ENOTSUP,
 * >= AF_EHIP_BASE = AF_EHIP_BASE + hipError_t.  af_last_error() returns a
 * thread-local message for the last failing call on this thread.
 *
 * Thread safety: all entry points are re-entrant; there is no global mutable
 * state besides per-thread error text (reference kernels are nogil and called
 * concurrently from dask worker threads, africanus/util/numba.py:9-12).
 */
#ifndef AFHIP_H
#define AFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AF_OK 0
#define AF_EINVAL 1
#define AF_ENOMEM 2
#define AF_ENOTSUP 3
#define AF_EHIP_BASE 1000

/* sign conventions: africanus/rime/phase.py:29-34, africanus/dft/kernels.py:34-39 */
#define AF_CONVENTION_FOURIER (-1) /* exp(-2 pi i ...) : minus_two_pi_over_c */
#define AF_CONVENTION_CASA (+1)    /* exp(+2 pi i ...) : two_pi_over_c       */

/* im_to_vis / fused-predict phasor evaluation modes */
#define AF_DFT_AUTO 0       /* channel recurrence when `frequency` is uniformly spaced
                               (decided on the device, no host sync), else AF_DFT_EXACT */
#define AF_DFT_EXACT 1      /* reference operation order + full-accuracy sincos per (row,src,chan) */
#define AF_DFT_RECURRENCE 2 /* force the recurrence (caller asserts uniform spacing) */
/* OR-able into `mode`: compute n = sqrt(max(0, 1-l^2-m^2)) - 1 as phase_delay does
 * (africanus/rime/phase.py:42-43) instead of im_to_vis' unclamped form: the
 * phase_delay -> einsum -> predict_vis chain without materialising the coherencies */
#define AF_DFT_CLAMP_N 0x100
/* OR-able into `mode`: keep real 4-correlation images on the VALU recurrence kernels instead of
 * the MFMA-accumulator kernels (same results to rounding; for comparison and tests) */
#define AF_DFT_VALU_ONLY 0x200

/* Jones layouts: africanus/rime/predict.py:10-12 */
#define AF_JONES_DIAG 1 /* JONES_1_OR_2: corr shape (1,) or (2,), element-wise products */
#define AF_JONES_2X2 2  /* JONES_2X2:    corr shape (2,2), 2x2 matrix products */

/* ---- runtime -------------------------------------------------------------- */
int af_version(void);
/* thread-local text of the last error on the calling thread ("" if none) */
const char *af_last_error(void);
int af_device_count(int *count);
int af_set_device(int device);
int af_get_device(int *device);
/* device name / arch (e.g. "gfx950") / CU count of `device` */
int af_device_info(int device, char *name, size_t name_len, char *arch, size_t arch_len,
                   int *compute_units, size_t *total_mem);
int af_malloc(void **dptr, size_t bytes);
int af_free(void *dptr);
int af_malloc_host(void **hptr, size_t bytes); /* pinned host memory */
int af_free_host(void *hptr);
/* Per-device scratch pool (SURVEY 8(b) "Ownership": the reference allocates with np.zeros / np.empty per call,
 * africanus/rime/predict.py:271, africanus/dft/kernels.py:45; on the device the equivalent hipMalloc / hipFree
 * pair per array per call synchronises the device).  af_pool_malloc returns a block of >= `bytes` on the calling
 * thread's current device, re-using a cached block when one fits; af_pool_free hands it back to the cache (work
 * that uses the block must be complete, or every later user must run on the same stream).  _host: page-locked
 * host memory.  Cached bytes are capped per device (env AFHIP_POOL_LIMIT, default 32 GiB; AFHIP_PINNED_LIMIT,
 * default 8 GiB, for the host list), least recently freed blocks go first.  af_pool_trim(device, keep) releases
 * cached blocks of `device` (-1 = host list) down to `keep` bytes; af_pool_stats reports cached / handed-out
 * bytes and cache hits / misses. */
int af_pool_malloc(void **dptr, size_t bytes);
int af_pool_free(void *dptr);
int af_pool_malloc_host(void **hptr, size_t bytes);
int af_pool_free_host(void *hptr);
int af_pool_trim(int device, size_t keep_bytes);
int af_pool_stats(int device, size_t *cached_bytes, size_t *in_use_bytes, int64_t *hits, int64_t *misses);
/* The calling thread's own (non-blocking) stream on its current device, created at first use: host threads
 * (dask workers; the reference kernels are nogil, africanus/util/numba.py:9-12) drive the device concurrently
 * without meeting on the NULL stream.  Owned by the library; do not destroy. */
int af_thread_stream(void **stream);
/* Releases every cached pool block and destroys the per-thread streams; no other call may be in flight.
 * Idempotent, and the library stays usable afterwards. */
int af_shutdown(void);
int af_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int af_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int af_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
int af_memset(void *dst_dev, int value, size_t bytes, void *stream);
int af_stream_create(void **stream);
int af_stream_destroy(void *stream);
int af_stream_synchronize(void *stream);
int af_device_synchronize(void);
/* hipEvent pair timing on `stream` (measurement support) */
int af_event_create(void **event);
int af_event_destroy(void *event);
int af_event_record(void *event, void *stream);
int af_stream_wait_event(void *stream, void *event);   /* work enqueued on `stream` after this call waits for `event` */
int af_event_synchronize(void *event);
int af_event_elapsed_ms(void *start, void *stop, float *ms);
/* Measurement hook (bench.py): while set (non-NULL) on the calling thread, entry
 * points record `start` right before and `stop` right after their dominant kernel launch(es),
 * on the stream those are launched on.  Pass NULLs to clear. */
int af_profile_events(void *start, void *stop);
/* element-wise precision conversion of n reals (dtype promotion on the device) */
int af_convert_f32_to_f64(const float *src, double *dst, int64_t n, void *stream);
int af_convert_f64_to_f32(const double *src, float *dst, int64_t n, void *stream);

/* ---- phase_delay ------------------------------------------------------------
 * Replaces africanus.rime.phase_delay (africanus/rime/phase.py:11-63).
 *   lm (nsrc,2), uvw (nrow,3), frequency (nchan) -> out (nsrc,nrow,nchan) complex
 *   n = sqrt(max(0, 1-l^2-m^2)) - 1 (clamped); p = C*(l*u+m*v+n*w)*nu in the
 *   reference's operation order; out = (cos p, sin p).
 * _f64: float64 in, complex128 out.  _f32: float32 in, complex64 out (all
 * arithmetic in float32, as the reference does when every input is float32). */
int af_phase_delay_f64(const double *lm, int64_t nsrc, const double *uvw, int64_t nrow,
                       const double *frequency, int64_t nchan, int convention,
                       double *out, void *stream);
int af_phase_delay_f32(const float *lm, int64_t nsrc, const float *uvw, int64_t nrow,
                       const float *frequency, int64_t nchan, int convention,
                       float *out, void *stream);

/* ---- im_to_vis --------------------------------------------------------------
 * Replaces africanus.dft.im_to_vis (africanus/dft/kernels.py:14-69).
 *   image (nsrc,nchan,ncorr) real (image_is_complex=0) or complex (=1);
 *   uvw (nrow,3); lm (nsrc,2); frequency (nchan) -> out (nrow,nchan,ncorr) complex128
 *   vis[r,nu,c] = sum_s exp(i*C*(l u+m v+n w)*nu) * image[s,nu,c], n unclamped
 *   (NaN outside the unit disc), zero pixels skipped (kernels.py:54,64).
 * `workspace`: device scratch of at least af_im_to_vis_workspace_bytes(...) bytes,
 * 256-byte aligned; it must stay untouched until the call's work on `stream` is done.
 * `mode`: AF_DFT_AUTO / AF_DFT_EXACT / AF_DFT_RECURRENCE. */
size_t af_im_to_vis_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t ncorr,
                                    int image_is_complex);
int af_im_to_vis_f64(const double *image, int image_is_complex, const double *uvw,
                     const double *lm, const double *frequency, int64_t nsrc, int64_t nrow,
                     int64_t nchan, int64_t ncorr, int convention, int mode, double *out,
                     void *workspace, size_t workspace_bytes, void *stream);
/* The transform and the per-channel chi^2 of its result in one call (the step of the row-sharded predict, SURVEY 8(e):
 * predict, then chi2[nu] = sum_{row, corr} [weight] |data - vis|^2, then ONE all-reduce of that vector):
 * out as af_im_to_vis_f64; data (nrow,nchan,ncorr) complex128; weight (nrow,nchan,ncorr) float64 or NULL;
 * chi2_per_chan (nchan) float64 = what af_chi2_c128(out, data, weight) gives (to the order of its atomic sums).  Where
 * the MFMA kernels run the sum is formed in their epilogue from the visibilities still in registers (no second pass over
 * them); every other case falls back to the separate pass on the device.  Same workspace as af_im_to_vis_f64. */
int af_im_to_vis_chi2_f64(const double *image, int image_is_complex, const double *uvw, const double *lm,
                          const double *frequency, int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr,
                          int convention, int mode, double *out, const double *data, const double *weight,
                          double *chi2_per_chan, void *workspace, size_t workspace_bytes, void *stream);

/* im_to_vis for single-precision callers: float32 image (complex64 when image_is_complex), uvw, lm and frequency ->
 * complex64 out.  The reference runs this case entirely in float32 (result dtype by promotion,
 * africanus/dft/kernels.py:26-31, africanus/util/type_inference.py:24-26); here phases are formed in float64 from the
 * promoted inputs, phasors / channel recurrence / sums are float32 (csrc/af_im_to_vis_f32.hip).  Contract: at least as
 * close to the float64 transform of the same inputs as the reference's float32 loop.  ncorr in {1, 2, 4}
 * (AF_ENOTSUP otherwise: promote and call af_im_to_vis_f64).  Same zero-pixel / NaN-source semantics, same `mode`
 * values (AF_DFT_RECURRENCE additionally asserts that the float32 frequency axis is meant to be uniform: its
 * rounding is then not followed channel by channel); `mode | AF_DFT_CLAMP_N`: n = sqrt(max(0, 1 - l^2 - m^2)) - 1, phase_delay's
 * form (africanus/rime/phase.py:42-43) -- what the fused predict without DDEs asks for on single-precision inputs.
 * The _f32 transforms are a CONVENIENCE for single-precision callers (the dtype contract, half the bytes), not a fast
 * path: at BASELINE configs[1]'s counts af_im_to_vis_f32 runs 1.08 x the rate of af_im_to_vis_f64 (0.46 of the fp32
 * vector peak: the 8-cycle v_mfma_f32_4x4x1 blocks leave no issue slots for the float32 phasor recurrence beside them,
 * DESIGN.md 3.10); the fp64 entry is the one the roofline of this library is stated on. */
size_t af_im_to_vis_f32_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t ncorr, int image_is_complex);
int af_im_to_vis_f32(const float *image, int image_is_complex, const float *uvw, const float *lm,
                     const float *frequency, int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr,
                     int convention, int mode, float *out, void *workspace, size_t workspace_bytes,
                     void *stream);

/* vis_to_im for single-precision callers: complex64 vis, float32 uvw / lm / frequency -> float32 image (the reference
 * computes this case in float32, africanus/dft/kernels.py:84-93); phasors in float64, products and sums in float32
 * (csrc/af_im_to_vis_f32.hip), partial images of the row partitions added in float64.  Same flag / non-finite
 * semantics as af_vis_to_im_f64; ncorr in {1, 2, 4} (AF_ENOTSUP otherwise). */
size_t af_vis_to_im_f32_workspace_bytes(int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr);
int af_vis_to_im_f32(const float *vis, const float *uvw, const float *lm, const float *frequency,
                     const unsigned char *flags, int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr,
                     int convention, int mode, float *out, void *workspace, size_t workspace_bytes, void *stream);

/* ---- vis_to_im --------------------------------------------------------------
 * Replaces africanus.dft.vis_to_im (africanus/dft/kernels.py:72-148), the adjoint of im_to_vis.
 *   vis (nrow,nchan,ncorr) complex128; uvw (nrow,3); lm (nsrc,2); frequency (nchan);
 *   flags (nrow,nchan,ncorr) bytes (nonzero = flagged) -> out (nsrc,nchan,ncorr) float64
 *   im[s,nu,c] = sum_r cos(p) vis.re - sin(p) vis.im over the (r,nu) none of whose correlations
 *   is flagged (:139-140); p = C (l u + m v + n w) nu with C = +2pi/c for AF_CONVENTION_FOURIER
 *   (:113-118, the opposite of im_to_vis); n unclamped (:125).
 * `mode` as for af_im_to_vis_f64.  The row sum is evaluated in row partitions that are added in
 * a fixed order (deterministic).  `workspace`: >= af_vis_to_im_workspace_bytes(...) device bytes
 * (it holds a flag-masked, tile-major repack of `vis`: about 1.1x its size). */
size_t af_vis_to_im_workspace_bytes(int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncorr);
int af_vis_to_im_f64(const double *vis, const double *uvw, const double *lm, const double *frequency,
                     const unsigned char *flags, int64_t nsrc, int64_t nrow, int64_t nchan,
                     int64_t ncorr, int convention, int mode, double *out, void *workspace,
                     size_t workspace_bytes, void *stream);

/* ---- predict_vis ------------------------------------------------------------
 * Replaces africanus.rime.predict_vis (africanus/rime/predict.py:466-619) and
 * apply_gains (:622-649: dies + base_vis only).
 *   V = G_p ( B + sum_s E_ps X_pqs E_qs^H ) G_q^H
 *   time_index/antenna1/antenna2 (nrow) integer, index_bytes = 4 (int32) or 8 (int64);
 *   time_index is normalised by its own minimum inside the call (predict.py:597)
 *   dde1/dde2 (nsrc,ntime,nant,nchan,ncorr)  -- both or neither (predict.py:403-404)
 *   source_coh (nsrc,nrow,nchan,ncorr)
 *   die1/die2 (ntime,nant,nchan,ncorr)       -- both or neither (predict.py:406-407)
 *   base_vis (nrow,nchan,ncorr)
 *   out (nrow,nchan,ncorr)
 * Absent terms are NULL.  ncorr in {1,2,4}; jones_kind AF_JONES_2X2 requires
 * ncorr == 4.  Sums run over sources in ascending order with the reference's
 * operation order and no fp contraction, so results are bit-identical to the
 * numba path.  `workspace`: >= af_predict_vis_workspace_bytes() device bytes.
 * Index guard (the reference fails under numba's boundscheck, predict.py:597-607): a row whose normalised time
 * index is outside [0, ntime) or whose antennas are outside [0, nant) reads clamped indices, gets NaN in all its
 * cells, and sets AF_STATUS_TIME_INDEX / AF_STATUS_ANTENNA in the int32 status word at workspace + 8 (zeroed by
 * the call, valid once the stream has passed the call); the entry itself still returns AF_OK (it does not
 * synchronise).  Only calls with DDE or DIE terms read the index arrays. */
#define AF_STATUS_TIME_INDEX 1
#define AF_STATUS_ANTENNA 2
#define AF_PREDICT_VIS_STATUS_OFFSET 8
size_t af_predict_vis_workspace_bytes(void);
int af_predict_vis_c128(const void *time_index, const void *antenna1, const void *antenna2,
                        int index_bytes, int64_t nrow, const double *dde1_jones,
                        const double *source_coh, const double *dde2_jones,
                        const double *die1_jones, const double *base_vis,
                        const double *die2_jones, int64_t nsrc, int64_t ntime, int64_t nant,
                        int64_t nchan, int ncorr, int jones_kind, double *out,
                        void *workspace, size_t workspace_bytes, void *stream);
int af_predict_vis_c64(const void *time_index, const void *antenna1, const void *antenna2,
                       int index_bytes, int64_t nrow, const float *dde1_jones,
                       const float *source_coh, const float *dde2_jones,
                       const float *die1_jones, const float *base_vis,
                       const float *die2_jones, int64_t nsrc, int64_t ntime, int64_t nant,
                       int64_t nchan, int ncorr, int jones_kind, float *out,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ---- beam cubes -------------------------------------------------------------
 * Replaces africanus.rime.fast_beam_cubes.freq_grid_interp
 * (africanus/rime/fast_beam_cubes.py:10-54): frequency (nchan), beam_freq_map
 * (beam_nud) -> freq_data (nchan,3) = (scale, lower weight, lower grid index). */
int af_freq_grid_interp_f64(const double *frequency, int64_t nchan, const double *beam_freq_map,
                            int64_t beam_nud, double *freq_data, void *stream);
int af_freq_grid_interp_f32(const float *frequency, int64_t nchan, const float *beam_freq_map,
                            int64_t beam_nud, float *freq_data, void *stream);
/* Replaces africanus.rime.beam_cube_dde (africanus/rime/fast_beam_cubes.py:57-240).
 *   beam (beam_lw,beam_mh,beam_nud,ncorr) complex; beam_lm_extents (2,2);
 *   beam_freq_map (beam_nud); lm (nsrc,2); parallactic_angles (ntime,nant);
 *   point_errors (ntime,nant,nchan,2); antenna_scaling (nant,nchan,2);
 *   frequency (nchan) -> out (nsrc,ntime,nant,nchan,ncorr) complex.
 * `workspace`: 256-byte aligned device scratch of af_beam_cube_dde_workspace_bytes(...) bytes
 * (frequency grid, (sin, cos) of the parallactic angles and |beam|, all built once per call). */
size_t af_beam_cube_dde_workspace_bytes(int64_t beam_lw, int64_t beam_mh, int64_t beam_nud, int64_t ncorr,
                                        int64_t ntime, int64_t nant, int64_t nchan, int is_f32);
int af_beam_cube_dde_c128(const double *beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud,
                          int ncorr, const double *beam_lm_extents, const double *beam_freq_map,
                          const double *lm, int64_t nsrc, const double *parallactic_angles,
                          int64_t ntime, int64_t nant, const double *point_errors,
                          const double *antenna_scaling, const double *frequency, int64_t nchan,
                          double *out, void *workspace, size_t workspace_bytes, void *stream);
int af_beam_cube_dde_c64(const float *beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud,
                         int ncorr, const float *beam_lm_extents, const float *beam_freq_map,
                         const float *lm, int64_t nsrc, const float *parallactic_angles,
                         int64_t ntime, int64_t nant, const float *point_errors,
                         const float *antenna_scaling, const float *frequency, int64_t nchan,
                         float *out, void *workspace, size_t workspace_bytes,
                         void *stream);

/* ---- fused predict with beam-cube DDEs -------------------------------------------
 * The reference's chain phase_delay -> einsum("srf,sfij->srfij", phase, brightness) ->
 * beam_cube_dde -> predict_vis(time_index, a1, a2, dde, coh, dde)
 * (africanus/rime/examples/predict.py:107-134,404-472,525; africanus/rime/phase.py:28-61;
 * africanus/rime/fast_beam_cubes.py:57-240; africanus/rime/predict.py:199-212) in one kernel:
 *   out[r,nu] = sum_s E_p(s,t,nu) . (K(r,s,nu) B(s,nu)) . E_q(s,t,nu)^H     (2x2 complex128)
 * without materialising the (src,row,chan) coherencies or the (src,time,ant,chan) Jones terms.
 *   items (nitems,4) int32 DEVICE: (time index into parallactic_angles/point_errors, row_start,
 *       row_count <= 2048, 0): runs of consecutive rows with equal time_index, built by
 *       af_fused_plan_rows (HOST pointers) from the row time_index array
 *   antenna1/antenna2 (nrow) int32; lm (nsrc,2); uvw (nrow,3); frequency (nchan);
 *   brightness (nsrc,nchan,2,2) complex128; beam (lw,mh,nud,2,2) complex128 and the other
 *   beam_cube_dde arguments as in af_beam_cube_dde_c128; out (nrow,nchan,2,2) complex128.
 *   feed_rotation (ntime,nant,2,2) complex128 or NULL: E <- E . R(t,a), the einsum
 *       "stafij,tajk->stafik" of africanus/rime/examples/predict.py:472 (af_feed_rotation_f64 output);
 *   gauss_shape (nsrc,3) = (major, minor, orientation) [rad] or NULL: K <- K . shape(r,s,nu), the Gaussian
 *       shape function of africanus/model/shape/gaussian_shape.py:11-62; (0,0,.) rows are point sources.
 * DIE terms / base_vis are applied afterwards with af_predict_vis_c128 (source_coh = out).
 * Limits (AF_EINVAL beyond them): nant <= 664 (one time step's Jones of a source in LDS), beam cube < 2^25 voxels
 * (32-bit byte offsets into its 128-byte voxel records), nchan <= 65535. */
/* dtype note: af_fused_predict_* and af_wsclean_predict_f64 exist in float64 / complex128 only.  float32 callers of the
 * fused predict or of wsclean_predict (the reference computes in the input precision: africanus/util/type_inference.py:24-26,
 * africanus/rime/wsclean_predict.py:11-84) are served by widening the inputs and rounding the complex128 result ONCE to
 * complex64 in the binding: the dtype contract holds, the values are at least as accurate as the reference's float32
 * loop.  Native float32 entries exist for phase_delay, predict_vis, beam_cube_dde, feed_rotation, convert, im_to_vis and
 * vis_to_im. */
int af_fused_plan_rows(const int64_t *time_index_host, int64_t nrow, int32_t *items_host,
                       int64_t max_items, int64_t *nitems);
size_t af_fused_predict_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t beam_lw, int64_t beam_mh,
                                        int64_t beam_nud);
/* Grouped form of the plan (HOST pointers): runs of equal time_index cut into groups of up to four rows that share
 * their antennas, rows (p_i, q_j), i, j in {0, 1}: groups (ngroups,8) int32 = p0, p1, q0, q1, row(0,0), row(0,1),
 * row(1,0), row(1,1) (-1 = empty); items (nitems,4) = (time index, first group, group count <= 512, 1).  With NULL
 * output arrays only the counts are returned.  A lane of the kernel then owns one group and reads the Jones terms of
 * its two + two antennas once for four baselines.  Pass the DEVICE copies as `items` and `groups` of
 * af_fused_predict_c128 (antenna1 / antenna2 may then be NULL); `groups` NULL = the row-range items of
 * af_fused_plan_rows.  Grouped items need the wave-specialised kernel (no gauss_shape). */
int af_fused_plan_groups(const int64_t *time_index_host, const int32_t *antenna1_host, const int32_t *antenna2_host,
                         int64_t nrow, int64_t nant, int32_t *items_host, int64_t max_items, int64_t *nitems,
                         int32_t *groups_host, int64_t max_groups, int64_t *ngroups);
int af_fused_predict_c128(const int32_t *items, int64_t nitems, const int32_t *antenna1,
                          const int32_t *antenna2, const int32_t *groups, int64_t nrow, const double *lm, const double *uvw,
                          const double *frequency, const double *brightness, int64_t nsrc,
                          int64_t nchan, const double *beam, int64_t beam_lw, int64_t beam_mh,
                          int64_t beam_nud, const double *beam_lm_extents, const double *beam_freq_map,
                          const double *parallactic_angles, int64_t ntime, int64_t nant,
                          const double *point_errors, const double *antenna_scaling,
                          const double *feed_rotation, const double *gauss_shape, int convention,
                          double *out, void *workspace, size_t workspace_bytes, void *stream);

/* ---- the same predict for ANTENNA-DECOMPOSABLE uvw: one complex GEMM per (timestep, channel) ----------------
 * When uvw_pq = uvw_p - uvw_q per timestep (every real Measurement Set) the phasor of phase_delay
 * (africanus/rime/phase.py:45-61) factorises into per-antenna phasors and the chain above becomes
 *     V_pq(t,nu) = sum_s G_ps A_qs^H,   A_as = k_as E_as,  G_as = A_as X_s,  k_as = exp(i C nu (l,m,n)_s . uvw_a)
 * i.e. M = G H^H with M (2 nant x 2 nant), K = 2 nsrc: evaluated with v_mfma_f64_16x16x4 on the upper block triangle
 * (csrc/af_fused_gemm.hip).  Replaces the same reference functions as af_fused_predict_c128; results agree with it
 * and with the reference chain to the rounding of the phase argument plus 2 pi nu / c |lmn| x the plan's residual.
 * af_fused_plan_antennas (HOST pointers; O(row)): per timestep a spanning-tree integration of uvw over the baseline
 *   graph + two Gauss-Seidel sweeps towards the least-squares antenna coordinates, component means removed;
 *   *decomposable = 1 iff max_rows |x_p - x_q - uvw_pq|_inf <= tol [m] (reported in *max_residual) and no
 *   (step, antenna1, antenna2) occurs twice.  nsteps = max(time_index) - min(time_index) + 1;
 *   ant_uvw_host (nsteps, nant, 3) double; rowmap_host (nsteps, nap, nap) int32, nap = 8 ceil(nant / 8): the row of
 *   baseline (p, q) of the step or -1.  nant <= 512.
 * af_fused_predict_antennas_c128 (DEVICE pointers): ant_uvw / rowmap = device copies of the plan; the other arguments
 *   as af_fused_predict_c128 (same workspace size); writes out[row] for every row the map names.  No gauss_shape (it
 *   depends on the baseline: use af_fused_predict_c128).  nant <= 64: one workgroup per (timestep, channel) holds the
 *   whole upper block triangle; 65 .. 512 antennas: the blocks of 8 antennas are cut into super-blocks of 8 blocks,
 *   one workgroup per (timestep, channel, super-block or half of a pair of super-blocks).
 *   BRIGHTNESS MUST BE HERMITIAN (X[1][0] == conj X[0][1], real diagonal: what africanus.model.coherency.convert makes of
 *   real Stokes parameters): the entry evaluates the upper block triangle of M and serves a baseline stored the other way
 *   round with the conjugate transpose of the computed element, A_q X^H A_p^H.  The reference's chain accepts any complex
 *   matrices; a binding must send others to af_fused_predict_c128 (codex_africanus_amd.rime.fused._hermitian does).  The
 *   same holds for af_fused_predict_antennas_c64.
 * af_fused_gemm_slots(nant): baseline slots (8 x 8-antenna tiles x 64) the GEMM form evaluates per (timestep, channel),
 *   what its cost is proportional to (rows per step / slots = the fill factor callers dispatch by); 0 beyond 512. */
int af_fused_plan_antennas(const int64_t *time_index_host, const int32_t *antenna1_host, const int32_t *antenna2_host,
                           const double *uvw_host, int64_t nrow, int64_t nant, double tol, int64_t nsteps,
                           double *ant_uvw_host, int32_t *rowmap_host, double *max_residual, int *decomposable);
int64_t af_fused_gemm_slots(int64_t nant);
int af_fused_predict_antennas_c128(const double *ant_uvw, const int32_t *rowmap, int64_t nsteps, int64_t nrow,
                                   const double *lm, const double *frequency, const double *brightness, int64_t nsrc,
                                   int64_t nchan, const double *beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud,
                                   const double *beam_lm_extents, const double *beam_freq_map,
                                   const double *parallactic_angles, int64_t ntime, int64_t nant,
                                   const double *point_errors, const double *antenna_scaling,
                                   const double *feed_rotation, int convention, double *out, void *workspace,
                                   size_t workspace_bytes, void *stream);

/* The SINGLE-PRECISION form of af_fused_predict_antennas_c128 (round 6): every array float32 / complex64 (pairs of
 * floats) except the plan's (ant_uvw stays double: the planner solves in double), complex64 out -- the precision in which
 * the reference runs this chain when every input is single precision (africanus/util/type_inference.py:24-26: the promoted
 * input type; africanus/rime/predict.py:542-544, phase.py:28-61 and fast_beam_cubes.py:57-240 with float32 arguments).
 * v_mfma_f32_16x16x4_f32 on float operand panels, float32 beam planes and sampler arithmetic; the antenna phasor and the
 * voxel coordinates in double (csrc/af_fused_gemm_c64.hip): CLOSER to the float64 chain on the same float32 inputs than
 * the reference's own float32 chain (golden G17).  Workspace: af_fused_predict_c64_workspace_bytes.  nant <= 512. */
size_t af_fused_predict_c64_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud);
int af_fused_predict_antennas_c64(const double *ant_uvw, const int32_t *rowmap, int64_t nsteps, int64_t nrow,
                                  const float *lm, const float *frequency, const float *brightness, int64_t nsrc,
                                  int64_t nchan, const float *beam, int64_t beam_lw, int64_t beam_mh, int64_t beam_nud,
                                  const float *beam_lm_extents, const float *beam_freq_map,
                                  const float *parallactic_angles, int64_t ntime, int64_t nant,
                                  const float *point_errors, const float *antenna_scaling, const float *feed_rotation,
                                  int convention, float *out, void *workspace, size_t workspace_bytes, void *stream);

/* The SINGLE-PRECISION form of af_fused_predict_c128 (round 6): ANY uvw (rows that do not decompose by antenna -- BASELINE
 * configs[2] as it draws its uvw --, Gaussian shapes, non-Hermitian brightness), every floating-point array float32 /
 * complex64, complex64 out; items / groups / antenna1 / antenna2 as planned by af_fused_plan_rows / af_fused_plan_groups.
 * Replaces the same reference chain as af_fused_predict_c128 for single-precision callers (the precision rule and file:line
 * as for af_fused_predict_antennas_c64 above).  Packed float32 Jones algebra, float32 beam planes and sampler; the phase
 * argument l u + m v + n w and its reduction in double (csrc/af_fused_predict_c64.hip).  Workspace:
 * af_fused_predict_c64_workspace_bytes.  nant <= 664. */
int af_fused_predict_c64(const int32_t *items, int64_t nitems, const int32_t *antenna1, const int32_t *antenna2,
                         const int32_t *groups, int64_t nrow, const float *lm, const float *uvw, const float *frequency,
                         const float *brightness, int64_t nsrc, int64_t nchan, const float *beam, int64_t beam_lw,
                         int64_t beam_mh, int64_t beam_nud, const float *beam_lm_extents, const float *beam_freq_map,
                         const float *parallactic_angles, int64_t ntime, int64_t nant, const float *point_errors,
                         const float *antenna_scaling, const float *feed_rotation, const float *gauss_shape, int convention,
                         float *out, void *workspace, size_t workspace_bytes, void *stream);

/* Plan guard (DEVICE pointers except none; O(row) on the device, no host round trip).  The predict entries above read
 * the PLAN's arrays (items / groups / antenna1 / antenna2, ant_uvw / rowmap), not the call's own index arrays and uvw: a
 * plan re-used with other rows or other uvw would silently compute with the old ones (the reference has no plan:
 * africanus/rime/predict.py:199-212 reads time_index / antenna1 / antenna2 of the call).  af_fused_plan_check, enqueued
 * AFTER the predict on the same stream, verifies for every row r
 *     time_index[r] - time_index[0] == plan_step[r] - plan_step[0],  antenna1[r] == plan_antenna1[r],  antenna2 likewise,
 *     and, with ant_uvw (nsteps, nant, 3) != NULL:  |x_a1 - x_a2 - uvw[r]|_inf <= tol  [m]
 * (plan_step = time_index - min(time_index) at planning time, int32; index_bytes = 4 / 8: int32 / int64 index arrays).
 * On a mismatch AF_STATUS_PLAN_INDEX / AF_STATUS_PLAN_UVW is set in *status (int32, DEVICE; zeroed by the call) and `out`
 * (out_doubles doubles; may be NULL) is filled with NaN. */
#define AF_STATUS_PLAN_INDEX 4
#define AF_STATUS_PLAN_UVW 8
int af_fused_plan_check(const void *time_index, const void *antenna1, const void *antenna2, int index_bytes,
                        const double *uvw, int64_t nrow, const int32_t *plan_step, const int32_t *plan_antenna1,
                        const int32_t *plan_antenna2, const double *ant_uvw, int64_t nant, double tol, double *out,
                        int64_t out_doubles, int32_t *status, void *stream);

/* ---- Gaussian (and point) sources without direction-dependent terms ----------------------------------------
 * out[r,nu] = sum_s shape(r,s,nu) K(r,s,nu) X_s(nu): the reference's phase_delay (africanus/rime/phase.py:11-63) x
 * gaussian shape (africanus/model/shape/gaussian_shape.py:11-62) x brightness, summed over the sources
 * (africanus/rime/examples/predict.py:107-134; africanus/rime/predict.py:229-246), as one direct transform whose phasor
 * carries the envelope (csrc/af_gauss_dft.hip).  lm (nsrc,2), uvw (nrow,3), frequency (nchan), brightness
 * (nsrc,nchan,2,2) complex128, gauss_shape (nsrc,3) = (major, minor, orientation) [rad] or NULL (point sources only);
 * out (nrow,nchan,2,2) complex128.  Any channel spacing: bands of >= 14 channels with one spacing run on the
 * MFMA-accumulator transform kernels (csrc/af_im_to_vis_mfma.hip, the envelope as a product recurrence in the lane's
 * phasor), uniform 8-channel tiles of other bands take the same recurrences lane = row, the rest one sincos and one
 * exponential per channel; which, is decided on the device. */
size_t af_gauss_predict_workspace_bytes(int64_t nsrc, int64_t nchan);
int af_gauss_predict_c128(const double *lm, const double *uvw, const double *frequency, const double *brightness,
                          const double *gauss_shape, int64_t nsrc, int64_t nrow, int64_t nchan, int convention,
                          double *out, void *workspace, size_t workspace_bytes, void *stream);
/* The same predict and chi2[nu] = sum_{row, corr} [weight] |data - out|^2 (af_chi2_c128's quantity; data (nrow,nchan,2,2)
 * complex128, weight the same shape real or NULL) in ONE call -- the step of the row-sharded predict, SURVEY 8(e): on bands
 * the MFMA-accumulator kernels own, chi^2 is summed in their epilogue (as af_im_to_vis_chi2_f64 does); elsewhere the call
 * falls back, on the device, to the separate pass.  No reference counterpart (parity unpinned: checked against numpy). */
int af_gauss_predict_chi2_c128(const double *lm, const double *uvw, const double *frequency, const double *brightness,
                               const double *gauss_shape, int64_t nsrc, int64_t nrow, int64_t nchan, int convention,
                               double *out, const double *data, const double *weight, double *chi2_per_chan,
                               void *workspace, size_t workspace_bytes, void *stream);

/* ---- chi-squared ---------------------------------------------------------------
 * chi2_per_chan[nu] = sum_{r,c} weight[r,nu,c] * |data[r,nu,c] - model[r,nu,c]|^2
 * (weight NULL = 1).  model/data (nrow,nchan,ncorr) complex128, weight real float64,
 * chi2_per_chan (nchan) float64, zeroed by the call.  No reference counterpart (the
 * reference has no chi^2; africanus/calibration/utils/residual_vis.py:63 forms the
 * residual): this is the quantity the row-sharded multi-GPU predict all-reduces.
 * ORDER DEPENDENCE: the per-channel sums are accumulated with floating-point atomic adds (here and in the _chi2
 * epilogues of the transforms): two calls on the same inputs agree to the order of the adds, ~1e-13 relative -- a
 * statistic, not a bit pattern.  Tests compare chi^2 with rtol 1e-12, visibilities bit for bit. */
int af_chi2_c128(const double *model, const double *data, const double *weight, int64_t nrow,
                 int64_t nchan, int64_t ncorr, double *chi2_per_chan, void *stream);

/* ---- term producers ---------------------------------------------------------------
 * Replaces africanus.rime.feed_rotation (africanus/rime/feeds.py:14-73): parallactic_angles (n) real ->
 * out (n,2,2) complex of the same precision; AF_FEED_LINEAR [[c, s], [-s, c]], AF_FEED_CIRCULAR
 * diag(e^{-i pa}, e^{+i pa}).  Any other feed_type is AF_EINVAL ("Invalid feed_type"). */
#define AF_FEED_LINEAR 0
#define AF_FEED_CIRCULAR 1
int af_feed_rotation_f64(const double *parallactic_angles, int64_t n, int feed_type, double *out, void *stream);
int af_feed_rotation_f32(const float *parallactic_angles, int64_t n, int feed_type, float *out, void *stream);
/* Replaces africanus.model.shape.gaussian (africanus/model/shape/gaussian_shape.py:11-62): uvw (nrow,3),
 * frequency (nchan), shape_params (nsrc,3) = (major, minor, orientation) [rad] -> out (nsrc,nrow,nchan)
 * float64.  `workspace`: nsrc*32 bytes of device scratch. */
int af_gaussian_shape_f64(const double *uvw, const double *frequency, const double *shape_params, int64_t nsrc,
                          int64_t nrow, int64_t nchan, double *out, void *workspace, size_t workspace_bytes,
                          void *stream);

/* ---- calibration consumers of the predict ------------------------------------------------
 * Replace africanus.calibration.utils.corrupt_vis (calibration/utils/corrupt_vis.py:58-101), residual_vis
 * (residual_vis.py:63-119) and correct_vis (correct_vis.py:63-115).  mode: 0 DIAG_DIAG (jones and vis carry
 * ncorr = 1 or 2 values), 1 DIAG (jones (2), vis (2,2)), 2 FULL (jones (2,2), vis (2,2)) -- check_type of
 * calibration/utils/utils.py:11-45.  J = ncorr | 2 | 4 gain values, V = ncorr | 4 | 4 visibility values.
 *   time_bin_indices / time_bin_counts (ntime) int64 (bin starts are normalised by their minimum on the device,
 *   as the reference does in place); antenna1/antenna2 (nrow) int64; jones (ntime,nant,nchan,ndir,J);
 *   model (nrow,nchan,ndir,V); vis (nrow,nchan,V); flag (nrow,nchan,V) bytes; out (nrow,nchan,V); complex128.
 *   Rows outside every bin and (row, chan) cells with any flagged correlation are 0 in the output of
 *   residual / correct.  correct_vis needs ndir == 1.  workspace: af_calibration_workspace_bytes(nrow). */
size_t af_calibration_workspace_bytes(int64_t nrow);
/* af_correct_vis_c128 with FULL gains: with this (larger) workspace the call first inverts every (time, antenna, chan)
 * gain once -- G^-1 and (G^H)^-1 of africanus/calibration/utils/correct_vis.py:70-89, the reference's operations on
 * the reference's operands, so results do not change by a bit -- instead of once per baseline and cell. */
size_t af_correct_vis_workspace_bytes(int64_t nrow, int64_t ntime, int64_t nant, int64_t nchan);
int af_corrupt_vis_c128(const int64_t *time_bin_indices, const int64_t *time_bin_counts, int64_t ntime,
                        const int64_t *antenna1, const int64_t *antenna2, const double *jones,
                        const double *model, int64_t nrow, int64_t nant, int64_t nchan, int64_t ndir, int mode,
                        int ncorr, double *out, void *workspace, size_t workspace_bytes, void *stream);
int af_residual_vis_c128(const int64_t *time_bin_indices, const int64_t *time_bin_counts, int64_t ntime,
                         const int64_t *antenna1, const int64_t *antenna2, const double *jones,
                         const double *vis, const unsigned char *flag, const double *model, int64_t nrow,
                         int64_t nant, int64_t nchan, int64_t ndir, int mode, int ncorr, double *out,
                         void *workspace, size_t workspace_bytes, void *stream);
int af_correct_vis_c128(const int64_t *time_bin_indices, const int64_t *time_bin_counts, int64_t ntime,
                        const int64_t *antenna1, const int64_t *antenna2, const double *jones,
                        const double *vis, const unsigned char *flag, int64_t nrow, int64_t nant, int64_t nchan,
                        int64_t ndir, int mode, int ncorr, double *out, void *workspace, size_t workspace_bytes,
                        void *stream);

/* ---- convolutional degridding ---------------------------------------------------------------
 * Replaces africanus.gridding.perleypolyhedron.degridder.degridder (gridding/perleypolyhedron/degridder.py:79-175)
 * with the gather convolution policies (policies/convolution_policies.py:188-323; packed != 0:
 * "conv_1d_axisymmetric_packed_gather", else "..._unpacked_gather"), baseline_transform_policy "None", the facet
 * phase rotation (phase_rotate != 0: "phase_rotate", policies/phase_transform_policies.py:9-35) and the
 * Stokes -> correlation policy given as ncorr (2 or 4) complex per-correlation factors
 * (policies/stokes_conversion_policies.py:8-137), corr_factors (ncorr) complex128 DEVICE.
 *   uvw (nrow,3); gridstack (nband,npix,npix) complex128; wavelengths (nchan); chanmap (nchan) int64 band of
 *   every channel; cell [arcsec]; image_centre / phase_centre (ra, dec) [rad] HOST pointers;
 *   convolution_kernel (oversampling * (width + 2)) float64 as made by kernels.pack_kernel / the unpacked form;
 *   out (nrow,nchan,ncorr) complex128.  workspace (optional, af_degridder_workspace_bytes(nrow), 256-byte
 *   aligned): lets the call process the rows in uv-tile order (a counting sort of their mid-band position),
 *   which keeps the gathers of concurrent waves inside one cache-sized neighbourhood; results are unchanged. */
size_t af_degridder_workspace_bytes(int64_t nrow);
int af_degridder_c128(const double *uvw, const double *gridstack, const double *wavelengths,
                      const int64_t *chanmap, double cell, const double *image_centre_host,
                      const double *phase_centre_host, const double *convolution_kernel, int64_t kernel_width,
                      int64_t kernel_oversampling, int phase_rotate, const double *corr_factors, int ncorr,
                      int packed, int64_t nrow, int64_t nchan, int64_t npix, double *out, void *workspace,
                      size_t workspace_bytes, void *stream);

/* Replaces africanus.model.coherency.convert (africanus/model/coherency/conversion.py:207-216; products :18-48):
 * Stokes <-> correlation conversion.  input (nelem, nin) of dtype `in_kind`, out (nelem, nout) of dtype `out_kind`
 * (AF_KIND_*: same precision; complex input -> complex output; a real output admits only the two real products).
 * Output o is the product op[o] of input columns src1[o], src2[o] (-1 = the implicit Stokes default 0,
 * `implicit_stokes=True`), the three tables being HOST arrays of nout ints (the schema resolution is host logic):
 *   AF_CONV_ADD  a + b + 0j   (RR, XX)      AF_CONV_SUB   a - b + 0j  (LL, YY)
 *   AF_CONV_ADDJ a + b*1j     (RL, XY)      AF_CONV_SUBJ  a - b*1j    (LR, YX)
 *   AF_CONV_HALF_ADD (a+b)/2  (I; Q of RL,LR)   AF_CONV_HALF_SUB (a-b)/2  (Q of XX,YY; V of RR,LL)
 *   AF_CONV_HALF_SUB_OVER_J (a-b)/2j  (U of RL,LR; V of XY,YX)
 * nin, nout <= 12 (a schema names each of I,Q,U,V,RR,RL,LR,LL,XX,XY,YX,YY at most once). */
#define AF_KIND_F32 0
#define AF_KIND_F64 1
#define AF_KIND_C64 2
#define AF_KIND_C128 3
#define AF_CONV_ADD 0
#define AF_CONV_SUB 1
#define AF_CONV_ADDJ 2
#define AF_CONV_SUBJ 3
#define AF_CONV_HALF_ADD 4
#define AF_CONV_HALF_SUB 5
#define AF_CONV_HALF_SUB_OVER_J 6
int af_coherency_convert(const void *input, int in_kind, int64_t nelem, int nin, int nout, const int *src1_host,
                         const int *src2_host, const int *op_host, void *out, int out_kind, void *stream);

/* Replaces africanus.model.spectral.spectral_model (africanus/model/spectral/spec_model.py:102-236):
 * stokes (nsrc,npol), spi (nsrc,nspi,npol), ref_freq (nsrc), frequency (nchan), base (npol) int32 DEVICE array of
 * 0 "std"  I prod_i (nu/nu0)^spi_i, 1 "log"  I exp(sum_i spi_i ln(nu/nu0)^(i+1)), 2 "log10" (same in base 10)
 * -> out (nsrc,nchan,npol) float64. */
int af_spectral_model_f64(const double *stokes, const double *spi, const double *ref_freq, const double *frequency,
                          const int *base, int64_t nsrc, int64_t nspi, int64_t npol, int64_t nchan, double *out,
                          void *stream);

/* Predict from the sky model: the reference's predict script forms the brightness with
 *   stokes = spectral_model(stokes, spi, ref_freq, frequency, base); brightness = convert(stokes, [I,Q,U,V], corr_schema)
 * (africanus/rime/examples/predict.py:494-498) in front of the predict.  These entry points take (stokes, spi,
 * ref_freq) -- stokes (nsrc,npol), spi (nsrc,nspi,npol), ref_freq (nsrc), base (npol) int32 DEVICE as in
 * af_spectral_model_f64; src1/src2/op (ncorr) HOST tables as in af_coherency_convert (the resolved corr_schema) -- and
 * run both steps on the device into the call's workspace: no (source, chan, corr) array on the caller's side.
 * image_is_complex = 0 when every correlation is a real product of the (real) Stokes spectra (e.g. XX, YY from I, Q):
 * the transform then runs its real-image kernels.  The rest as af_im_to_vis_f64 / af_fused_predict_c128 (whose
 * brightness is always the four complex correlations, ncorr = 4). */
size_t af_im_to_vis_model_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t npol, int64_t ncorr, int image_is_complex);
int af_im_to_vis_model_f64(const double *stokes, const double *spi, const double *ref_freq, const int *base,
                           int64_t nspi, int64_t npol, const int *src1_host, const int *src2_host, const int *op_host,
                           int64_t ncorr, int image_is_complex, const double *uvw, const double *lm,
                           const double *frequency, int64_t nsrc, int64_t nrow, int64_t nchan, int convention, int mode,
                           double *out, void *workspace, size_t workspace_bytes, void *stream);
size_t af_fused_predict_model_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t npol, int64_t beam_lw,
                                              int64_t beam_mh, int64_t beam_nud);
int af_fused_predict_model_c128(const double *stokes, const double *spi, const double *ref_freq, const int *base,
                                int64_t nspi, int64_t npol, const int *src1_host, const int *src2_host,
                                const int *op_host, const int32_t *items, int64_t nitems, const int32_t *antenna1,
                                const int32_t *antenna2, const int32_t *groups, int64_t nrow, const double *lm,
                                const double *uvw,
                                const double *frequency, int64_t nsrc, int64_t nchan, const double *beam, int64_t beam_lw,
                                int64_t beam_mh, int64_t beam_nud, const double *beam_lm_extents,
                                const double *beam_freq_map, const double *parallactic_angles, int64_t ntime,
                                int64_t nant, const double *point_errors, const double *antenna_scaling,
                                const double *feed_rotation, const double *gauss_shape, int convention, double *out,
                                void *workspace, size_t workspace_bytes, void *stream);
/* the same for antenna-decomposable uvw (af_fused_predict_antennas_c128 after the model steps) */
int af_fused_predict_antennas_model_c128(const double *stokes, const double *spi, const double *ref_freq, const int *base,
                                         int64_t nspi, int64_t npol, const int *src1_host, const int *src2_host,
                                         const int *op_host, const double *ant_uvw, const int32_t *rowmap, int64_t nsteps,
                                         int64_t nrow, const double *lm, const double *frequency, int64_t nsrc,
                                         int64_t nchan, const double *beam, int64_t beam_lw, int64_t beam_mh,
                                         int64_t beam_nud, const double *beam_lm_extents, const double *beam_freq_map,
                                         const double *parallactic_angles, int64_t ntime, int64_t nant,
                                         const double *point_errors, const double *antenna_scaling,
                                         const double *feed_rotation, int convention, double *out, void *workspace,
                                         size_t workspace_bytes, void *stream);

/* wgridder-style degridding, image -> visibilities at a requested accuracy (BASELINE configs[4]):
 * the entry under africanus.gridding.wgridder.model (africanus/gridding/wgridder/im2vis.py:14-61), whose arithmetic is the
 * un-vendored ducc0.wgridder.dirty2ms -- parity is pinned only by the reference's accuracy contract
 * (africanus/gridding/wgridder/tests/test_wgridder.py:18-113): relative l2 error against the direct transform
 *   vis[r,nu] = sum_xy image[x,y] / n * exp(-2 pi i nu/c (u x + v y - w (n - 1)))   <= epsilon.
 * One imaging band per call: freq (nchan_band) are columns chan0.. of the (nrow, nchan_total) arrays vis (complex128,
 * the band's columns are overwritten), wgt (float64 or NULL) and mask (bytes or NULL: 0 = skip, result 0).
 * image (nx, ny) float64; corr_u (nx) / corr_v (ny): 1 / Fourier transform of the kernel along the padded axes
 * (af_wgrid_padded: the smallest even 2-3-5-7-smooth size >= 2 n); quad_t / quad_w (48): Gauss-Legendre nodes / weights on (0, 1); kernel_width W and beta: the
 * exponential-of-semicircle kernel exp(beta (sqrt(1 - (2t/W)^2) - 1)); [wl_min, wl_max]: range of w nu / c over the
 * band, signs as they are in uvw (HOST scalars, they size the w-plane loop; any superset of the true range is valid);
 * max_abs_nm1: largest |n - 1| of the image.  The image is real, so V(-u,-v,-w) = conj V(u,v,w): visibilities with w < 0
 * are evaluated at the mirrored point and conjugated (the adjoint grids their conjugates there), and the planes cover
 * [min |w nu/c|, max |w nu/c|] only -- af_wgrid_planes() counts them.  All device work is
 * enqueued on `stream` (the visibility sort of large image -> vis calls on a library-owned side stream that starts
 * behind `stream`'s work at the call and is joined before the visibilities are written); the plane transforms of image rows / columns of 512, 1024, 2048, 4096 or 8192 cells are own kernels
 * (csrc/af_wgridder.hip, wg_fill_fft_rows), every other size uses hipFFT (plans cached per device and size, released by
 * af_shutdown). */
int64_t af_wgrid_padded(int64_t n);
/* Workspace of both directions.  planes: w-plane grids resident at a time (>= 1; af_wgrid_planes() of them = a single
 * pass over the visibilities); nchan_max / nplanes_total: the most channels / w-planes of a band the workspace will serve;
 * kernel_width: W (they size the tables of the on-device visibility sort).  Even image sizes only (as ducc0). */
size_t af_wgrid_workspace_bytes(int64_t nx, int64_t ny, int64_t planes, int64_t nrow, int64_t nchan_max,
                                int64_t nplanes_total, int kernel_width);
int64_t af_wgrid_planes(double wl_min, double wl_max, double max_abs_nm1, int kernel_width, int do_wstacking);
/* Precision of the w-planes of the calling THREAD's following af_wgrid_im2vis_f64 calls; returns the previous mode.
 * AF_WGRID_PLANES_F32: float32 planes (float32 FFTs, fp64 sums) when kernel_width <= 7, i.e. a requested accuracy of
 * 1e-5 or coarser -- what the reference's single-precision calls get from ducc0 (float32 image:
 * africanus/gridding/wgridder/im2vis.py:41-47; its tests then ask l2 <= max(epsilon, 3e-7) and adjointness to 1e-4,
 * gridding/wgridder/tests/test_wgridder.py:55-108,125-188).  Default AF_WGRID_PLANES_F64 (adjointness to 1e-12). */
#define AF_WGRID_PLANES_F64 0
#define AF_WGRID_PLANES_F32 1
int af_wgrid_plane_precision(int mode);
int af_wgrid_im2vis_f64(const double *uvw, const double *freq, int64_t nrow, int64_t nchan_band, int64_t chan0,
                        int64_t nchan_total, const double *image, int64_t nx, int64_t ny, double cellx, double celly,
                        const double *corr_u, const double *corr_v, const double *quad_t, const double *quad_w,
                        int kernel_width, double beta, double wl_min, double wl_max, double max_abs_nm1,
                        int do_wstacking, const double *wgt, const unsigned char *mask, double *vis, void *workspace,
                        size_t workspace_bytes, void *stream);

/* The adjoint (replaces africanus.gridding.wgridder.dirty, africanus/gridding/wgridder/vis2im.py:15-72, i.e.
 * ducc0.wgridder.ms2dirty): `image` (nx, ny) float64 is OVERWRITTEN with
 *     (1 / n) sum_{r, c} Re( wgt vis exp(+2 pi i nu/c (u x + v y - w (n - 1))) )
 * over the visibilities with mask != 0 of columns chan0 .. chan0 + nchan_band of vis (complex128) / wgt / mask.  Same
 * geometry arguments and workspace as af_wgrid_im2vis_f64, whose exact transpose this is. */
int af_wgrid_vis2im_f64(const double *uvw, const double *freq, int64_t nrow, int64_t nchan_band, int64_t chan0,
                        int64_t nchan_total, const double *vis, int64_t nx, int64_t ny, double cellx, double celly,
                        const double *corr_u, const double *corr_v, const double *quad_t, const double *quad_w,
                        int kernel_width, double beta, double wl_min, double wl_max, double max_abs_nm1,
                        int do_wstacking, const double *wgt, const unsigned char *mask, double *image, void *workspace,
                        size_t workspace_bytes, void *stream);

/* Replaces africanus.calibration.utils.compute_and_corrupt_vis (calibration/utils/compute_and_corrupt_vis.py:73-152):
 * corrupt_vis with the model coherencies formed on the fly from a time-variable point-source model,
 *   source_vis = model[t,nu,dir] * exp(-2 pi i nu/c (u l + v m + w (n - 1))) / n,  n = sqrt(1 - l^2 - m^2),
 * model (ntime,nchan,ndir,V), lm (ntime,ndir,2), uvw (nrow,3), frequency (nchan); the rest as af_corrupt_vis_c128. */
int af_compute_and_corrupt_vis_c128(const int64_t *time_bin_indices, const int64_t *time_bin_counts,
                                    int64_t ntime, const int64_t *antenna1, const int64_t *antenna2,
                                    const double *jones, const double *model, const double *uvw,
                                    const double *frequency, const double *lm, int64_t nrow, int64_t nant,
                                    int64_t nchan, int64_t ndir, int mode, int ncorr, double *out,
                                    void *workspace, size_t workspace_bytes, void *stream);

/* Replaces africanus.gridding.perleypolyhedron.gridder.gridder (gridding/perleypolyhedron/gridder.py:12-117), the
 * adjoint of the degridder: conv_policy 0 "conv_1d_axisymmetric_unpacked_scatter", 1 "..._packed_scatter",
 * 2 "conv_nn_scatter" (policies/convolution_policies.py:6-185); corr_factors (ncorr) complex128 DEVICE = the
 * corr2stokes policy (policies/stokes_conversion_policies.py:143-180) as per-correlation factors; phase_rotate != 0
 * applies the facet phase rotation with sign +1 first; do_normalize divides every band by its summed tap weights
 * + 1e-8.  vis (nrow,nchan,ncorr) complex128 (not modified); gridstack (nband,npix,npix) complex128, zeroed by the
 * call; the adds are hardware fp64 atomics, so results are reproducible to rounding only.
 * workspace: af_gridder_workspace_bytes(nrow, nchan, nband, npix). */
size_t af_gridder_workspace_bytes(int64_t nrow, int64_t nchan, int64_t nband, int64_t npix);
int af_gridder_c128(const double *uvw, const double *vis, const double *wavelengths, const int64_t *chanmap,
                    int64_t npix, double cell, const double *image_centre_host, const double *phase_centre_host,
                    const double *convolution_kernel, int64_t kernel_width, int64_t kernel_oversampling,
                    int phase_rotate, const double *corr_factors, int ncorr, int conv_policy, int do_normalize,
                    int64_t nrow, int64_t nchan, int64_t nband, double *gridstack, void *workspace,
                    size_t workspace_bytes, void *stream);

/* ---- WSClean component-list predict ------------------------------------------------
 * Replaces africanus.model.wsclean.spectra (africanus/model/wsclean/spec_model.py:70-126) and
 * africanus.rime.wsclean_predict (africanus/rime/wsclean_predict.py:11-120):
 *   spectrum[s,f] = flux[s] * exp(sum_k coeffs[s,k] * log(nu_f/ref_freq[s])^(k+1))   log_poly[s] != 0
 *                 = flux[s] + sum_k coeffs[s,k] * (nu_f/ref_freq[s] - 1)^(k+1)       otherwise
 *   vis[r,f]      = sum_s spectrum[s,f] * shape_s(r,f) * exp(+2 pi i (u l + v m + w (n-1)) nu_f / c)
 * shape = 1 for point components (is_gaussian[s] == 0), the elliptical Gaussian envelope of
 * gauss_shape[s] = (major, minor, position angle) [rad] otherwise (wsclean_predict.py:49-77).
 * uvw (nrow,3), lm (nsrc,2), is_gaussian / log_poly (nsrc) bytes, flux / ref_freq (nsrc), coeffs
 * (nsrc,ncoeffs), gauss_shape (nsrc,3), frequency (nchan), all float64; out (nrow,nchan) complex128
 * (the reference's trailing correlation axis of length 1).  mode as for af_im_to_vis_f64 (without
 * AF_DFT_CLAMP_N): the recurrence path also advances the Gaussian envelope by a product recurrence. */
int af_wsclean_spectra_f64(const double *flux, const double *coeffs, const unsigned char *log_poly,
                           const double *ref_freq, const double *frequency, int64_t nsrc, int64_t ncoeffs,
                           int64_t nchan, double *out, void *stream);
size_t af_wsclean_predict_workspace_bytes(int64_t nsrc, int64_t nchan);
int af_wsclean_predict_f64(const double *uvw, const double *lm, const unsigned char *is_gaussian,
                           const double *flux, const double *coeffs, const unsigned char *log_poly,
                           const double *ref_freq, const double *gauss_shape, const double *frequency,
                           int64_t nsrc, int64_t nrow, int64_t nchan, int64_t ncoeffs, int mode, double *out,
                           void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AFHIP_H */
