#!/opt/conda/bin/python3.9
"""
Generate golden input/output vectors for the RIME predict hot path by running
the REAL reference (codex-africanus v0.4.4, numba CPU path) in the build
container.  The vectors (data only) are committed under tests/golden/*.npz and
are what pins oracle/ and the HIP kernels; the reference itself never travels.

Run (build container only; /root/reference does not exist on the GPU box):

    NUMBA_CACHE_DIR=/tmp/numba_cache PYTHONPATH=/root/reference:tests/golden \
        /opt/conda/bin/python3.9 tests/golden/make_golden.py

Each .npz holds the inputs and the reference's outputs for one group:
  g1_phase_delay.npz   phase_delay      (africanus/rime/phase.py:11)
  g2_predict_vis.npz   predict_vis      (africanus/rime/predict.py:466), 27 presence/corr combos
  g3_im_to_vis.npz     im_to_vis        (africanus/dft/kernels.py:14)
  g4_beam.npz          freq_grid_interp / beam_cube_dde (africanus/rime/fast_beam_cubes.py:10,57)
  g6_vis_to_im.npz     vis_to_im        (africanus/dft/kernels.py:72), flags / real vis / shapes
  g7_wsclean.npz       wsclean_predict  (africanus/rime/wsclean_predict.py:86), point + Gaussian components
  g5_chain_c1.npz      BASELINE config C1 (10k rows, 16 chan, 100 src, 4 corr): sampled rows + checksums
"""
import os
import sys

import ref_shim  # noqa: F401  (must come first)
import numpy as np

from africanus.rime.phase import phase_delay
from africanus.rime.predict import predict_vis
from africanus.dft.kernels import im_to_vis, vis_to_im
from africanus.rime.fast_beam_cubes import beam_cube_dde, freq_grid_interp

HERE = os.path.dirname(os.path.abspath(__file__))


def rc(rng, shape):
    return rng.random(shape) + 1j * rng.random(shape)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


# ----------------------------------------------------------------------------
def g1_phase_delay():
    rng = np.random.default_rng(101)
    lm = (rng.random((7, 2)) - 0.5) * 0.2
    lm[3] = [0.8, 0.9]            # l^2 + m^2 > 1 -> clamp branch (phase.py:43)
    lm[5] = [0.0, 0.0]            # phase centre
    uvw = (rng.random((33, 3)) - 0.5) * 2e4
    freq = np.linspace(0.856e9, 1.712e9, 5)
    out = dict(lm=lm, uvw=uvw, frequency=freq)
    for conv in ("fourier", "casa"):
        out["f64_" + conv] = phase_delay(lm, uvw, freq, convention=conv)
    # float32: small baselines so the float32 phase is meaningful
    lm32 = lm.astype(np.float32)
    uvw32 = ((rng.random((33, 3)) - 0.5) * 20).astype(np.float32)
    freq32 = freq.astype(np.float32)
    out.update(lm32=lm32, uvw32=uvw32, frequency32=freq32)
    for conv in ("fourier", "casa"):
        out["f32_" + conv] = phase_delay(lm32, uvw32, freq32, convention=conv)
    # the reference's own known-answer case (africanus/rime/tests/test_rime.py:19-47)
    rng2 = np.random.default_rng(7)
    uvw_k = rng2.random((100, 3))
    lm_k = rng2.random((10, 2))
    freq_k = np.linspace(0.856e9, 0.856e9 * 2, 64, endpoint=True)
    uvw_k[2] = [1, 2, 3]
    lm_k[3] = [0.1, 0.2]
    freq_k[5] = 0.856e9
    out.update(kat_lm=lm_k, kat_uvw=uvw_k, kat_frequency=freq_k,
               kat_fourier=phase_delay(lm_k, uvw_k, freq_k, convention="fourier")[3, 2],
               kat_casa=phase_delay(lm_k, uvw_k, freq_k, convention="casa")[3, 2])
    save("g1_phase_delay.npz", **out)


# ----------------------------------------------------------------------------
CORR_SHAPES = {"c1": (1,), "c2": (2,), "c22": (2, 2)}
DDE_PRESENCE = {"ddecoh": (True, True, True), "dde": (True, False, True), "coh": (False, True, False)}
DIE_PRESENCE = {"diebv": (True, True, True), "die": (True, False, True), "bv": (False, True, False)}


def g2_predict_vis():
    # shapes of africanus/rime/tests/test_predict.py:23-30,85-87
    s, t, a, c, r = 21, 4, 4, 5, 10
    time_idx = np.asarray([0, 0, 1, 1, 2, 2, 2, 2, 3, 3])
    ant1 = np.asarray([0, 0, 0, 0, 1, 1, 1, 2, 2, 3])
    ant2 = np.asarray([0, 1, 2, 3, 1, 2, 3, 2, 3, 3])
    out = dict(time_idx=time_idx, ant1=ant1, ant2=ant2)
    rng = np.random.default_rng(202)
    for ck, corr_shape in CORR_SHAPES.items():
        arrs = dict(
            a1=rc(rng, (s, t, a, c) + corr_shape), bl=rc(rng, (s, r, c) + corr_shape),
            a2=rc(rng, (s, t, a, c) + corr_shape), g1=rc(rng, (t, a, c) + corr_shape),
            bv=rc(rng, (r, c) + corr_shape), g2=rc(rng, (t, a, c) + corr_shape))
        for k, v in arrs.items():
            out["%s_%s" % (ck, k)] = v
        for dk, (a1j, blj, a2j) in DDE_PRESENCE.items():
            for gk, (g1j, bvis, g2j) in DIE_PRESENCE.items():
                vis = predict_vis(
                    time_idx, ant1, ant2,
                    arrs["a1"] if a1j else None, arrs["bl"] if blj else None,
                    arrs["a2"] if a2j else None, arrs["g1"] if g1j else None,
                    arrs["bv"] if bvis else None, arrs["g2"] if g2j else None)
                out["%s_%s_%s_vis" % (ck, dk, gk)] = vis
        # time index offset + int32 indices (africanus/rime/cuda/tests/test_cuda_predict.py:43-45)
        vis = predict_vis((time_idx + 10).astype(np.int32), ant1.astype(np.int32),
                          ant2.astype(np.int32), arrs["a1"], arrs["bl"], arrs["a2"],
                          arrs["g1"], arrs["bv"], arrs["g2"])
        out["%s_offset_vis" % ck] = vis
        # complex64 inputs -> complex64 output
        a64 = {k: v.astype(np.complex64) for k, v in arrs.items()}
        out["%s_c64_vis" % ck] = predict_vis(time_idx, ant1, ant2, a64["a1"], a64["bl"], a64["a2"],
                                             a64["g1"], a64["bv"], a64["g2"])
    save("g2_predict_vis.npz", **out)


# ----------------------------------------------------------------------------
def g3_im_to_vis():
    rng = np.random.default_rng(303)
    nsrc, nrow, nchan = 13, 50, 6
    lm = (rng.random((nsrc, 2)) - 0.5) * 0.1
    uvw = (rng.random((nrow, 3)) - 0.5) * 8e3
    uvw[:, 2] *= 0.1
    freq = np.linspace(0.856e9, 1.712e9, nchan)
    freq_nonuniform = freq * (1.0 + 0.01 * rng.random(nchan))
    out = dict(lm=lm, uvw=uvw, frequency=freq, frequency_nonuniform=freq_nonuniform)
    for ncorr in (1, 2, 4):
        img_r = rng.standard_normal((nsrc, nchan, ncorr))
        img_r[rng.random(img_r.shape) < 0.2] = 0.0          # zero pixels are skipped (kernels.py:64)
        img_c = img_r + 1j * rng.standard_normal(img_r.shape)
        img_c[rng.random(img_c.shape) < 0.2] = 0.0
        out["img_r%d" % ncorr] = img_r
        out["img_c%d" % ncorr] = img_c
        for conv in ("fourier", "casa"):
            out["vis_r%d_%s" % (ncorr, conv)] = im_to_vis(img_r, uvw, lm, freq, convention=conv)
            out["vis_c%d_%s" % (ncorr, conv)] = im_to_vis(img_c, uvw, lm, freq, convention=conv)
        out["vis_r%d_nonuniform" % ncorr] = im_to_vis(img_r, uvw, lm, freq_nonuniform)
        out["vis_c%d_nonuniform" % ncorr] = im_to_vis(img_c, uvw, lm, freq_nonuniform)
        out["vis_r%d_c64" % ncorr] = im_to_vis(img_r, uvw, lm, freq, dtype=np.complex64)
    # 5 correlations (generic ncorr path) and 70 channels (several channel tiles + remainder)
    img5 = rng.standard_normal((nsrc, nchan, 5))
    out["img_r5"] = img5
    out["vis_r5_fourier"] = im_to_vis(img5, uvw, lm, freq)
    freq70 = np.linspace(0.9e9, 1.6e9, 70)
    img70 = rng.standard_normal((nsrc, 70, 4))
    out.update(frequency70=freq70, img_r70=img70, vis_r70_fourier=im_to_vis(img70, uvw, lm, freq70))
    # a source outside the unit disc -> NaN phase; only its non-zero pixels poison (kernels.py:54,64)
    lm_nan = lm.copy()
    lm_nan[4] = [0.9, 0.8]
    img_nan = out["img_r4"].copy()
    img_nan[4, :, :] = 0.0
    img_nan[4, 2, 1] = 1.5
    out.update(lm_nan=lm_nan, img_nan=img_nan, vis_nan=im_to_vis(img_nan, uvw, lm_nan, freq))
    # float32 inputs -> complex64 output (small baselines)
    uvw32 = (uvw * 1e-3).astype(np.float32)
    out.update(uvw32=uvw32,
               vis_f32=im_to_vis(out["img_r4"].astype(np.float32), uvw32,
                                 lm.astype(np.float32), freq.astype(np.float32)))
    save("g3_im_to_vis.npz", **out)


# ----------------------------------------------------------------------------
def g4_beam():
    rng = np.random.default_rng(404)
    # fixtures of africanus/rime/tests/test_fast_beams.py:18-40
    beam_freq_map = np.array([0.5, 0.56, 0.7, 0.91, 1.0])
    freqs = np.array([0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1])
    out = dict(beam_freq_map=beam_freq_map, freqs=freqs,
               freq_data=freq_grid_interp(freqs, beam_freq_map))
    src, time, ants, chans = 10, 5, 4, freqs.shape[0]
    lm = (rng.random((src, 2)) - 0.5) * 1.6          # spans most of the cube, some clamp at edges
    lm[0] = [1.3, -1.2]                               # outside the extents -> clamped (:150-151)
    parangles = rng.random((time, ants)) * np.pi / 12
    point_errors = (rng.random((time, ants, chans, 2)) - 0.5) * 0.05
    antenna_scaling = 1.0 + (rng.random((ants, chans, 2)) - 0.5) * 0.1
    extents = np.asarray([[-1.0, 1.0], [-1.0, 1.0]])
    beam = rc(rng, (10, 10, beam_freq_map.shape[0], 2, 2))
    beam[3, 4, 2] = 0.0                               # exercise the div == 0 branch neighbourhood
    out.update(lm=lm, parangles=parangles, point_errors=point_errors,
               antenna_scaling=antenna_scaling, extents=extents, beam=beam)
    out["ddes"] = beam_cube_dde(beam, extents, beam_freq_map, lm, parangles, point_errors,
                                antenna_scaling, freqs)
    # an all-zero cube: the `div == 0` branch (:229-231)
    out["ddes_zero"] = beam_cube_dde(np.zeros_like(beam), extents, beam_freq_map, lm[:2], parangles,
                                     point_errors, antenna_scaling, freqs)
    # float32 / complex64 and 1 correlation
    f32 = np.float32
    out["ddes_f32"] = beam_cube_dde(beam.astype(np.complex64), extents.astype(f32),
                                    beam_freq_map.astype(f32), lm.astype(f32), parangles.astype(f32),
                                    point_errors.astype(f32), antenna_scaling.astype(f32),
                                    freqs.astype(f32))
    beam1 = np.ascontiguousarray(beam[..., 0, :1])
    out["ddes_1corr"] = beam_cube_dde(beam1, extents, beam_freq_map, lm, parangles, point_errors,
                                      antenna_scaling, freqs)
    # the reference's small known-answer test (test_fast_beams.py:43-127): seed 42 -> 0.470255+0.4786j
    np.random.seed(42)
    kb = np.random.random((2, 2, 2, 1)) + 1j * np.random.random((2, 2, 2, 1))
    out["kat_beam"] = kb
    out["kat_ddes"] = beam_cube_dde(kb, np.asarray([[-1.0, 1.0], [-1.0, 1.0]]), np.asarray([0.0, 1.0]),
                                    np.asarray([[0.1, 0.1]]), np.zeros((1, 1)), np.zeros((1, 1, 1, 2)),
                                    np.ones((1, 1, 2)), np.asarray([0.3]))
    save("g4_beam.npz", **out)


# ----------------------------------------------------------------------------
def c1_inputs(seed=0, nrow=10000, nchan=16, nsrc=100, nant=7):
    """Synthetic inputs of SURVEY.md 8(d), shape C1.  Re-implemented identically
    (same rng call order) in tests/_synth.py so only seeds + samples are stored."""
    rng = np.random.default_rng(seed)
    rad = 0.05 * np.sqrt(rng.random(nsrc))
    ang = 2 * np.pi * rng.random(nsrc)
    lm = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1)
    uvw = np.empty((nrow, 3))
    uvw[:, 0] = rng.uniform(-4000, 4000, nrow)
    uvw[:, 1] = rng.uniform(-4000, 4000, nrow)
    uvw[:, 2] = rng.uniform(-400, 400, nrow)
    freq = np.linspace(0.856e9, 1.712e9, nchan)
    stokes_i = rng.lognormal(0.0, 1.0, nsrc)
    q, u, v = (0.1 * rng.standard_normal(nsrc) for _ in range(3))
    # linear feeds: [I+Q, U+iV, U-iV, I-Q]
    bright = np.stack([stokes_i + q, u + 1j * v, u - 1j * v, stokes_i - q], axis=1)  # (src, 4) complex
    nbl = nant * (nant - 1) // 2
    a1, a2 = np.triu_indices(nant, 1)
    ntime = -(-nrow // nbl)
    ant1 = np.tile(a1, ntime)[:nrow].astype(np.int32)
    ant2 = np.tile(a2, ntime)[:nrow].astype(np.int32)
    time_index = np.repeat(np.arange(ntime, dtype=np.int32), nbl)[:nrow]
    return dict(lm=lm, uvw=uvw, frequency=freq, brightness=bright, ant1=ant1, ant2=ant2,
                time_index=time_index, ntime=ntime, nant=nant, rng=rng)


def g5_chain_c1():
    d = c1_inputs()
    lm, uvw, freq, bright = d["lm"], d["uvw"], d["frequency"], d["brightness"]
    nsrc, nchan = lm.shape[0], freq.shape[0]
    rows = np.linspace(0, uvw.shape[0] - 1, 64).astype(np.int64)
    out = dict(seed=np.int64(0), sample_rows=rows)
    # (a) im_to_vis with the real image pattern [I+Q, U, U, I-Q]
    image_r = np.broadcast_to(bright.real[:, None, :], (nsrc, nchan, 4)).copy()
    vis = im_to_vis(image_r, uvw, lm, freq)
    out.update(dft_rows=vis[rows], dft_sum=vis.sum(), dft_abssum=np.abs(vis).sum())
    # (b) chain phase_delay -> einsum -> predict_vis (africanus/rime/examples/predict.py:107-134,525)
    phase = phase_delay(lm, uvw, freq)
    coh = np.einsum("srf,si->srfi", phase, bright).reshape(nsrc, uvw.shape[0], nchan, 2, 2)
    vis2 = predict_vis(d["time_index"], d["ant1"], d["ant2"], None, coh, None, None, None, None)
    out.update(chain_rows=vis2[rows], chain_sum=vis2.sum(), chain_abssum=np.abs(vis2).sum())
    # (c) the same with per-antenna DIE gains and a base_vis
    rng = d["rng"]
    die = (1.0 + 0.1 * rng.standard_normal((d["ntime"], d["nant"], nchan, 2, 2))
           + 0.1j * rng.standard_normal((d["ntime"], d["nant"], nchan, 2, 2)))
    bvis = 0.01 * (rng.standard_normal(vis2.shape) + 1j * rng.standard_normal(vis2.shape))
    vis3 = predict_vis(d["time_index"], d["ant1"], d["ant2"], None, coh, None, die, bvis, die)
    out.update(die_rows=vis3[rows], die_sum=vis3.sum(), die_abssum=np.abs(vis3).sum())
    save("g5_chain_c1.npz", **out)


# ----------------------------------------------------------------------------
def g6_vis_to_im():
    """vis_to_im (africanus/dft/kernels.py:72): flags, real/complex vis, conventions, shapes."""
    rng = np.random.default_rng(606)
    nsrc, nrow, nchan = 11, 57, 6
    lm = (rng.random((nsrc, 2)) - 0.5) * 0.1
    uvw = (rng.random((nrow, 3)) - 0.5) * 8e3
    uvw[:, 2] *= 0.1
    freq = np.linspace(0.856e9, 1.712e9, nchan)
    freq_nonuniform = freq * (1.0 + 0.01 * rng.random(nchan))
    out = dict(lm=lm, uvw=uvw, frequency=freq, frequency_nonuniform=freq_nonuniform)
    for ncorr in (1, 2, 4):
        vis = rng.standard_normal((nrow, nchan, ncorr)) + 1j * rng.standard_normal((nrow, nchan, ncorr))
        flags = rng.random((nrow, nchan, ncorr)) < 0.15
        out["vis%d" % ncorr] = vis
        out["flags%d" % ncorr] = flags
        for conv in ("fourier", "casa"):
            out["im%d_%s" % (ncorr, conv)] = vis_to_im(vis, uvw, lm, freq, flags, convention=conv)
        out["im%d_nonuniform" % ncorr] = vis_to_im(vis, uvw, lm, freq_nonuniform, flags)
    out["im4_realvis"] = vis_to_im(out["vis4"].real.copy(), uvw, lm, freq, out["flags4"])
    out["im4_noflags"] = vis_to_im(out["vis4"], uvw, lm, freq, np.zeros_like(out["flags4"]))
    out["im4_f32"] = vis_to_im(out["vis4"], uvw, lm, freq, out["flags4"], dtype=np.float32)
    # 70 channels (several tiles), 300 rows (several row partitions)
    freq70 = np.linspace(0.9e9, 1.6e9, 70)
    uvw300 = (rng.random((300, 3)) - 0.5) * 8e3
    vis70 = rng.standard_normal((300, 70, 4)) + 1j * rng.standard_normal((300, 70, 4))
    flags70 = rng.random((300, 70, 4)) < 0.1
    out.update(frequency70=freq70, uvw300=uvw300, vis70=vis70, flags70=flags70,
               im70=vis_to_im(vis70, uvw300, lm, freq70, flags70))
    # a source outside the unit disc: NaN image row (kernels.py:125)
    lm_nan = lm.copy()
    lm_nan[2] = [0.9, 0.8]
    out.update(lm_nan=lm_nan, im4_nan=vis_to_im(out["vis4"], uvw, lm_nan, freq, out["flags4"]))
    save("g6_vis_to_im.npz", **out)


# ----------------------------------------------------------------------------
def g7_wsclean():
    """wsclean_predict (africanus/rime/wsclean_predict.py:86) and spectra
    (africanus/model/wsclean/spec_model.py:70), recipe of rime/tests/test_wsclean_predict.py:26-60
    with realistic baselines / source sizes so that the Gaussian envelope is not ~1 or ~0."""
    from africanus.rime.wsclean_predict import wsclean_predict
    from africanus.model.wsclean.spec_model import spectra
    rs = np.random.RandomState(42)
    out = {}
    for tag, (row, src, chan) in dict(small=(10, 21, 5), big=(130, 37, 70)).items():
        source_sel = rs.randint(0, 2, src).astype(np.bool_)
        source_type = np.where(source_sel, "POINT", "GAUSSIAN")
        gauss_shape = np.stack([np.abs(rs.normal(size=src)) * 2e-4, np.abs(rs.normal(size=src)) * 1e-4,
                                rs.uniform(0, np.pi, src)], axis=1)
        uvw = rs.normal(size=(row, 3)) * 1500.0
        uvw[:, 2] *= 0.1
        lm = rs.normal(size=(src, 2)) * 1e-2
        flux = rs.normal(size=src)
        ncoeff = 2 if tag == "small" else 4
        coeffs = rs.normal(size=(src, ncoeff)) * np.array([1.0, 0.5, 0.2, 0.1])[:ncoeff]
        log_poly = rs.randint(0, 2, src).astype(np.bool_)
        flux[log_poly] = np.abs(flux[log_poly])
        coeffs[log_poly] = np.abs(coeffs[log_poly])
        freq = np.linspace(0.856e9, 2 * 0.856e9, chan)
        ref_freq = np.full(src, freq[freq.shape[0] // 2])
        vis = wsclean_predict(uvw, lm, source_type, flux, coeffs, log_poly, ref_freq, gauss_shape, freq)
        out.update({tag + "_uvw": uvw, tag + "_lm": lm, tag + "_is_gauss": ~source_sel, tag + "_flux": flux,
                    tag + "_coeffs": coeffs, tag + "_log_poly": log_poly, tag + "_ref_freq": ref_freq,
                    tag + "_gauss_shape": gauss_shape, tag + "_freq": freq, tag + "_vis": vis,
                    tag + "_spectrum": spectra(flux, coeffs, log_poly, ref_freq, freq)})
        if tag == "small":
            fnu = freq * (1 + 0.01 * rs.random_sample(chan))
            out["small_freq_nonuniform"] = fnu
            out["small_vis_nonuniform"] = wsclean_predict(uvw, lm, source_type, flux, coeffs, log_poly, ref_freq,
                                                          gauss_shape, fnu)
    save("g7_wsclean.npz", **out)


# ----------------------------------------------------------------------------
def g8_producers():
    """feed_rotation (africanus/rime/feeds.py:50) and the Gaussian shape function
    (africanus/model/shape/gaussian_shape.py:11), recipes of rime/tests/test_rime.py:50-77 and
    model/shape/tests/test_gaussian_shape.py:11-24 with realistic source sizes / baselines."""
    from africanus.rime import feed_rotation
    from africanus.model.shape import gaussian
    rs = np.random.RandomState(8)
    pa = rs.uniform(-np.pi, np.pi, (5, 7))
    uvw = rs.normal(size=(40, 3)) * 1500.0
    freq = np.linspace(0.856e9, 2 * 0.856e9, 16)
    shape_params = np.stack([np.abs(rs.normal(size=9)) * 2e-4, np.abs(rs.normal(size=9)) * 1e-4,
                             rs.uniform(0, np.pi, 9)], axis=1)
    shape_params[3, 0] = 0.0      # emaj == 0: er = emin / 1
    from africanus.model.spectral import spectral_model
    stokes = rs.normal(size=(11, 4))
    stokes[:, 0] = np.abs(stokes[:, 0]) + 0.5
    spi = rs.normal(size=(11, 3, 4)) * np.array([0.7, 0.2, 0.05])[None, :, None]
    ref_freq = rs.uniform(1.0e9, 1.4e9, 11)
    spec = {"spec_std": spectral_model(stokes, spi, ref_freq, freq, base=0),
            "spec_log": spectral_model(stokes, spi, ref_freq, freq, base=1),
            "spec_log10": spectral_model(stokes, spi, ref_freq, freq, base=2),
            "spec_list": spectral_model(stokes, spi, ref_freq, freq, base=[0, 1, 2]),
            "spec_nopol": spectral_model(stokes[:, 0].copy(), spi[:, :, 0].copy(), ref_freq, freq, base="log")}
    save("g8_producers.npz", stokes=stokes, spi=spi, spec_ref_freq=ref_freq, **spec, pa=pa, feed_linear=feed_rotation(pa, "linear"), feed_circular=feed_rotation(pa, "circular"),
         feed_linear_f32=feed_rotation(pa.astype(np.float32), "linear"),
         uvw=uvw, freq=freq, shape_params=shape_params, gauss=gaussian(uvw, freq, shape_params))


# ----------------------------------------------------------------------------
def g9_calibration():
    """corrupt_vis / residual_vis / correct_vis (africanus/calibration/utils/{corrupt,residual,correct}_vis.py)
    for the four (corr, jones) layouts of calibration/utils/tests/test_utils.py:10-18, seeded; time bins as
    produced by chunkify_rows (utils.py:48-61)."""
    from africanus.calibration.utils import corrupt_vis, residual_vis, correct_vis, chunkify_rows
    from africanus.calibration.utils import compute_and_corrupt_vis
    rs = np.random.RandomState(9)
    ntime, nant, nchan, ndir = 5, 4, 6, 3
    a1, a2 = np.triu_indices(nant, 1)
    nbl = a1.shape[0]
    time = np.repeat(np.arange(ntime, dtype=np.float64) * 10.0, nbl)
    ant1, ant2 = np.tile(a1, ntime).astype(np.int32), np.tile(a2, ntime).astype(np.int32)
    nrow = time.shape[0]
    _, tbi, tbc = chunkify_rows(time, ntime)
    rc = lambda *sh: rs.normal(size=sh) + 1j * rs.normal(size=sh)
    cc_uvw = rs.normal(size=(nrow, 3)) * 800.0
    cc_freq = np.linspace(1.0e9, 1.2e9, nchan)
    cc_lm = rs.normal(size=(ntime, ndir, 2)) * 0.02
    out = dict(time=time, ant1=ant1, ant2=ant2, tbin_idx=tbi, tbin_counts=tbc, cc_uvw=cc_uvw, cc_freq=cc_freq, cc_lm=cc_lm)
    for tag, corr, jcorr in (("dd1", (1,), (1,)), ("dd2", (2,), (2,)), ("diag", (2, 2), (2,)), ("full", (2, 2), (2, 2))):
        jones = rc(ntime, nant, nchan, ndir, *jcorr) + 1.0
        model = rc(nrow, nchan, ndir, *corr)
        vis = corrupt_vis(tbi.copy(), tbc, ant1, ant2, jones, model)
        data = vis + 0.1 * rc(*vis.shape)
        flag = rs.random_sample(vis.shape) < 0.1
        res = residual_vis(tbi.copy(), tbc, ant1, ant2, jones, data, flag, model)
        j1 = np.ascontiguousarray(jones[:, :, :, :1])
        cor = correct_vis(tbi.copy(), tbc, ant1, ant2, j1, data, flag)
        tmodel = rc(ntime, nchan, ndir, *corr)
        out[tag + "_tmodel"] = tmodel
        out[tag + "_ccvis"] = compute_and_corrupt_vis(tbi.copy(), tbc, ant1, ant2, jones, tmodel, cc_uvw, cc_freq, cc_lm)
        out.update({tag + "_jones": jones, tag + "_model": model, tag + "_vis": vis, tag + "_data": data,
                    tag + "_flag": flag, tag + "_residual": res, tag + "_corrected": cor})
    save("g9_calibration.npz", **out)


# ----------------------------------------------------------------------------
def g10_degridder():
    """Perley-polyhedron convolutional degridder (africanus/gridding/perleypolyhedron/degridder.py:178
    degridder_serial) with the kernels of kernels.py (kbsinc, pack_kernel), recipe of
    gridding/perleypolyhedron/tests/test_ppgridder.py:180-376: W = 7 taps, oversampling 9."""
    from africanus.gridding.perleypolyhedron import kernels
    from africanus.gridding.perleypolyhedron.degridder import degridder_serial
    rs = np.random.RandomState(10)
    npix, nrow, nchan, W, OS = 64, 60, 5, 7, 9
    cell = 4.0                                      # arcsec
    wavelengths = 299792458.0 / np.linspace(1.0e9, 1.4e9, nchan)
    umax = 0.45 * 1.0 / np.deg2rad(cell / 3600.0) * wavelengths.min()
    uvw = rs.uniform(-1, 1, (nrow, 3)) * umax
    uvw[:, 2] *= 0.05
    uvw[0] = [0.0, 0.0, 0.0]
    uvw[1, :2] = [umax * 1.3, -umax * 1.3]         # taps that fall off the grid
    grid = rs.normal(size=(2, npix, npix)) + 1j * rs.normal(size=(2, npix, npix))
    chanmap = np.array([0, 0, 1, 1, 1])
    kern = kernels.kbsinc(W, oversample=OS)
    pkern = kernels.pack_kernel(kern, W, oversample=OS)
    pc = np.array([0.3, -0.5])
    ic = np.array([0.3 + 0.01, -0.5 + 0.008])
    out = dict(uvw=uvw, grid=grid, wavelengths=wavelengths, chanmap=chanmap, cell=cell, kern=kern, pkern=pkern,
               phase_centre=pc, image_centre=ic, W=W, OS=OS)
    cases = [("packed_I4", "None", "None", "XXXYYXYY_FROM_I", "conv_1d_axisymmetric_packed_gather", pkern, pc),
             ("unpacked_I2", "None", "None", "XXYY_FROM_I", "conv_1d_axisymmetric_unpacked_gather", kern, pc),
             ("packed_V4_rot", "None", "phase_rotate", "XXXYYXYY_FROM_V", "conv_1d_axisymmetric_packed_gather", pkern, ic),
             ("packed_Q2_rot", "None", "phase_rotate", "XXYY_FROM_Q", "conv_1d_axisymmetric_packed_gather", pkern, ic),
             ("unpacked_U4", "None", "None", "RRRLLRLL_FROM_U", "conv_1d_axisymmetric_unpacked_gather", kern, pc)]
    for tag, bpol, ppol, spol, cpol, k, centre in cases:
        out[tag] = degridder_serial(uvw.copy(), grid, wavelengths, chanmap, cell, (centre[0], centre[1]),
                                    (pc[0], pc[1]), k, W, OS, bpol, ppol, spol, cpol)
    # the adjoint: gridder.py:12-117 with the scatter policies
    from africanus.gridding.perleypolyhedron.gridder import gridder
    vis2 = rs.normal(size=(nrow, nchan, 2)) + 1j * rs.normal(size=(nrow, nchan, 2))
    vis4 = rs.normal(size=(nrow, nchan, 4)) + 1j * rs.normal(size=(nrow, nchan, 4))
    out.update(gvis2=vis2, gvis4=vis4)
    gcases = [("grid_unpacked_I2", vis2, "None", "I_FROM_XXYY", "conv_1d_axisymmetric_unpacked_scatter", kern, pc, False),
              ("grid_packed_V4_rot_norm", vis4, "phase_rotate", "V_FROM_XXXYYXYY", "conv_1d_axisymmetric_packed_scatter",
               pkern, ic, True),
              ("grid_packed_Q2_rot", vis2, "phase_rotate", "Q_FROM_XXYY", "conv_1d_axisymmetric_packed_scatter", pkern, ic,
               False),
              ("grid_nn_U4", vis4, "None", "U_FROM_RRRLLRLL", "conv_nn_scatter", kern, pc, True)]
    inside = np.abs(uvw[:, :2]).max(axis=1) < umax          # conv_nn_scatter has no bounds check in the reference
    out["grid_nn_rows"] = inside
    for tag, v, ppol, spol, cpol, k, centre, norm in gcases:
        rows = inside if cpol == "conv_nn_scatter" else np.ones(nrow, bool)
        out[tag] = gridder(uvw[rows].copy(), v[rows].copy(), wavelengths, chanmap, npix, cell, (centre[0], centre[1]),
                           (pc[0], pc[1]), k, W, OS, "None", ppol, spol, cpol, do_normalize=norm)
    save("g10_degridder.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10"]
    fns = dict(g1=g1_phase_delay, g2=g2_predict_vis, g3=g3_im_to_vis, g4=g4_beam, g5=g5_chain_c1,
               g6=g6_vis_to_im, g7=g7_wsclean, g8=g8_producers, g9=g9_calibration, g10=g10_degridder)
    for w in which:
        fns[w]()
