// Predict from the SKY MODEL: Stokes parameters + spectral indices in, visibilities out.
//
// The reference's predict script forms the per-source brightness on the host side of its graph,
//     stokes     = spectral_model(stokes, spi, ref_freq, frequency, base)      africanus/model/spectral/spec_model.py:102-236
//     brightness = convert(stokes, ["I","Q","U","V"], corr_schema)             africanus/model/coherency/conversion.py:207-216
// (africanus/rime/examples/predict.py:494-498) and hands the (source, chan, corr) array to the predict.  These entry
// points take (stokes, spi, ref_freq) and evaluate both steps on the device, into the call's own workspace, in front
// of the transform: the caller never holds (or uploads) a (source, chan, corr) array.  The two steps are the library's
// own stand-alone kernels (af_spectral_model_f64, af_coherency_convert: bit-identical to the reference's functions),
// so the result equals the stand-alone chain bit for bit.
#include "af_common.h"

namespace {
struct ModelWs { size_t spec, image, rest, total; };
ModelWs model_ws(int64_t nsrc, int64_t nchan, int64_t npol, int64_t ncorr, size_t rest_bytes)
{
    ModelWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    w.spec = take((size_t)(nsrc * nchan * npol) * sizeof(double));
    w.image = take((size_t)(nsrc * nchan * ncorr) * 2 * sizeof(double));
    w.rest = take(rest_bytes);
    w.total = o;
    return w;
}

// stokes -> spectrum -> correlations, into `image` (complex128, or float64 when image_is_complex == 0)
int model_image(const double *stokes, const double *spi, const double *ref_freq, const double *frequency, const int *base,
                int64_t nsrc, int64_t nspi, int64_t npol, int64_t nchan, const int *src1_host, const int *src2_host,
                const int *op_host, int ncorr, int image_is_complex, double *spec, double *image, void *stream)
{
    int rc = af_spectral_model_f64(stokes, spi, ref_freq, frequency, base, nsrc, nspi, npol, nchan, spec, stream);
    if (rc != AF_OK) return rc;
    return af_coherency_convert(spec, AF_KIND_F64, nsrc * nchan, (int)npol, ncorr, src1_host, src2_host, op_host, image,
                                image_is_complex ? AF_KIND_C128 : AF_KIND_F64, stream);
}
}  // namespace

AF_EXPORT size_t af_im_to_vis_model_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t npol, int64_t ncorr,
                                                    int image_is_complex)
{
    if (nsrc < 0 || nchan < 0 || npol < 0 || ncorr < 0) return 0;
    return model_ws(nsrc, nchan, npol, ncorr, af_im_to_vis_workspace_bytes(nsrc, nchan, ncorr, image_is_complex)).total;
}

AF_EXPORT int af_im_to_vis_model_f64(const double *stokes, const double *spi, const double *ref_freq, const int *base,
                                     int64_t nspi, int64_t npol, const int *src1_host, const int *src2_host,
                                     const int *op_host, int64_t ncorr, int image_is_complex, const double *uvw,
                                     const double *lm, const double *frequency, int64_t nsrc, int64_t nrow, int64_t nchan,
                                     int convention, int mode, double *out, void *workspace, size_t workspace_bytes,
                                     void *stream)
{
    AF_REQUIRE(nsrc >= 0 && nrow >= 0 && nchan >= 0 && nspi >= 0, "af_im_to_vis_model_f64: negative extent");
    AF_REQUIRE(npol >= 1 && npol <= 12 && ncorr >= 1 && ncorr <= 12, "af_im_to_vis_model_f64: 1..12 polarisations / correlations");
    const size_t rest = af_im_to_vis_workspace_bytes(nsrc, nchan, ncorr, image_is_complex);
    const ModelWs W = model_ws(nsrc, nchan, npol, ncorr, rest);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= W.total, "af_im_to_vis_model_f64: workspace too small (%zu < %zu)",
               workspace_bytes, W.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_im_to_vis_model_f64: workspace must be 256-byte aligned");
    char *ws = static_cast<char *>(workspace);
    double *image = reinterpret_cast<double *>(ws + W.image);
    if (nsrc > 0 && nchan > 0) {
        AF_REQUIRE(stokes && spi && ref_freq && base && frequency, "af_im_to_vis_model_f64: NULL array");
        int rc = model_image(stokes, spi, ref_freq, frequency, base, nsrc, nspi, npol, nchan, src1_host, src2_host, op_host,
                             (int)ncorr, image_is_complex, reinterpret_cast<double *>(ws + W.spec), image, stream);
        if (rc != AF_OK) return rc;
    }
    return af_im_to_vis_f64(image, image_is_complex, uvw, lm, frequency, nsrc, nrow, nchan, ncorr, convention, mode, out,
                            ws + W.rest, rest, stream);
}

AF_EXPORT size_t af_fused_predict_model_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t npol, int64_t beam_lw,
                                                        int64_t beam_mh, int64_t beam_nud)
{
    if (nsrc < 0 || nchan < 0 || npol < 0) return 0;
    return model_ws(nsrc, nchan, npol, 4, af_fused_predict_workspace_bytes(nsrc, nchan, beam_lw, beam_mh, beam_nud)).total;
}

AF_EXPORT int af_fused_predict_model_c128(const double *stokes, const double *spi, const double *ref_freq, const int *base,
                                          int64_t nspi, int64_t npol, const int *src1_host, const int *src2_host,
                                          const int *op_host, const int32_t *items, int64_t nitems, const int32_t *antenna1,
                                          const int32_t *antenna2, const int32_t *groups, int64_t nrow, const double *lm,
                                          const double *uvw,
                                          const double *frequency, int64_t nsrc, int64_t nchan, const double *beam,
                                          int64_t beam_lw, int64_t beam_mh, int64_t beam_nud, const double *beam_lm_extents,
                                          const double *beam_freq_map, const double *parallactic_angles, int64_t ntime,
                                          int64_t nant, const double *point_errors, const double *antenna_scaling,
                                          const double *feed_rotation, const double *gauss_shape, int convention, double *out,
                                          void *workspace, size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(nsrc >= 0 && nchan >= 0 && nspi >= 0 && npol >= 1 && npol <= 12, "af_fused_predict_model_c128: bad extents");
    const size_t rest = af_fused_predict_workspace_bytes(nsrc, nchan, beam_lw, beam_mh, beam_nud);
    const ModelWs W = model_ws(nsrc, nchan, npol, 4, rest);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= W.total,
               "af_fused_predict_model_c128: workspace too small (%zu < %zu)", workspace_bytes, W.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_fused_predict_model_c128: workspace must be 256-byte aligned");
    char *ws = static_cast<char *>(workspace);
    double *brightness = reinterpret_cast<double *>(ws + W.image);
    if (nsrc > 0 && nchan > 0) {
        AF_REQUIRE(stokes && spi && ref_freq && base && frequency, "af_fused_predict_model_c128: NULL array");
        int rc = model_image(stokes, spi, ref_freq, frequency, base, nsrc, nspi, npol, nchan, src1_host, src2_host, op_host, 4,
                             1, reinterpret_cast<double *>(ws + W.spec), brightness, stream);
        if (rc != AF_OK) return rc;
    }
    return af_fused_predict_c128(items, nitems, antenna1, antenna2, groups, nrow, lm, uvw, frequency, brightness, nsrc, nchan, beam,
                                 beam_lw, beam_mh, beam_nud, beam_lm_extents, beam_freq_map, parallactic_angles, ntime, nant,
                                 point_errors, antenna_scaling, feed_rotation, gauss_shape, convention, out, ws + W.rest, rest,
                                 stream);
}

// The sky-model form of af_fused_predict_antennas_c128 (antenna-decomposable uvw: csrc/af_fused_gemm.hip): brightness =
// convert(spectral_model(...)) on the device, then the GEMM-form predict.  Workspace: af_fused_predict_model_workspace_bytes.
AF_EXPORT int af_fused_predict_antennas_model_c128(const double *stokes, const double *spi, const double *ref_freq,
                                                   const int *base, int64_t nspi, int64_t npol, const int *src1_host,
                                                   const int *src2_host, const int *op_host, const double *ant_uvw,
                                                   const int32_t *rowmap, int64_t nsteps, int64_t nrow, const double *lm,
                                                   const double *frequency, int64_t nsrc, int64_t nchan, const double *beam,
                                                   int64_t beam_lw, int64_t beam_mh, int64_t beam_nud,
                                                   const double *beam_lm_extents, const double *beam_freq_map,
                                                   const double *parallactic_angles, int64_t ntime, int64_t nant,
                                                   const double *point_errors, const double *antenna_scaling,
                                                   const double *feed_rotation, int convention, double *out, void *workspace,
                                                   size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(nsrc >= 0 && nchan >= 0 && nspi >= 0 && npol >= 1 && npol <= 12,
               "af_fused_predict_antennas_model_c128: bad extents");
    const size_t rest = af_fused_predict_workspace_bytes(nsrc, nchan, beam_lw, beam_mh, beam_nud);
    const ModelWs W = model_ws(nsrc, nchan, npol, 4, rest);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= W.total,
               "af_fused_predict_antennas_model_c128: workspace too small (%zu < %zu)", workspace_bytes, W.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_fused_predict_antennas_model_c128: workspace must be 256-byte aligned");
    char *ws = static_cast<char *>(workspace);
    double *brightness = reinterpret_cast<double *>(ws + W.image);
    if (nsrc > 0 && nchan > 0) {
        AF_REQUIRE(stokes && spi && ref_freq && base && frequency, "af_fused_predict_antennas_model_c128: NULL array");
        int rc = model_image(stokes, spi, ref_freq, frequency, base, nsrc, nspi, npol, nchan, src1_host, src2_host, op_host, 4,
                             1, reinterpret_cast<double *>(ws + W.spec), brightness, stream);
        if (rc != AF_OK) return rc;
    }
    return af_fused_predict_antennas_c128(ant_uvw, rowmap, nsteps, nrow, lm, frequency, brightness, nsrc, nchan, beam, beam_lw,
                                          beam_mh, beam_nud, beam_lm_extents, beam_freq_map, parallactic_angles, ntime, nant,
                                          point_errors, antenna_scaling, feed_rotation, convention, out, ws + W.rest, rest,
                                          stream);
}
