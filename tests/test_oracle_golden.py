"""
Pins the CPU oracle (oracle/rime_oracle.c) to the real reference: every oracle
function is compared with golden vectors captured from codex-africanus' numba
path (tests/golden/make_golden.py) and with the reference tests' own
known-answer values.  CPU only.
"""
import numpy as np
import pytest
from numpy.testing import assert_array_almost_equal, assert_array_equal

import oracle
from codex_africanus_amd.testing import synthetic_inputs

CORR = {"c1": (1,), "c2": (2,), "c22": (2, 2)}
DDE = {"ddecoh": (True, True, True), "dde": (True, False, True), "coh": (False, True, False)}
DIE = {"diebv": (True, True, True), "die": (True, False, True), "bv": (False, True, False)}


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) if np.size(a) else 0.0


# --------------------------------------------------------------------------- phase_delay
@pytest.mark.parametrize("conv", ["fourier", "casa"])
def test_phase_delay_f64_bit_exact(g1, conv):
    out = oracle.phase_delay(g1["lm"], g1["uvw"], g1["frequency"], convention=conv)
    ref = g1["f64_" + conv]
    assert out.dtype == ref.dtype == np.complex128 and out.shape == ref.shape
    # same operation order + same libm: identical bits
    assert_array_equal(out, ref)


@pytest.mark.parametrize("conv", ["fourier", "casa"])
def test_phase_delay_f32(g1, conv):
    out = oracle.phase_delay(g1["lm32"], g1["uvw32"], g1["frequency32"], convention=conv)
    ref = g1["f32_" + conv]
    assert out.dtype == ref.dtype == np.complex64
    assert maxabs(out, ref) < 2e-6


@pytest.mark.parametrize("conv, sign", [("fourier", 1), ("casa", -1)])
def test_phase_delay_reference_kat(g1, conv, sign):
    """africanus/rime/tests/test_rime.py:19-47 (exact equality there)."""
    out = oracle.phase_delay(g1["kat_lm"], g1["kat_uvw"], g1["kat_frequency"], convention=conv)
    minus_two_pi_over_c = -2 * np.pi / 2.99792458e8
    u, v, w, l, m, freq = 1, 2, 3, 0.1, 0.2, 0.856e9
    n = np.sqrt(1.0 - l**2 - m**2) - 1.0
    phase = sign * minus_two_pi_over_c * (u * l + v * m + w * n) * freq
    assert np.exp(1j * phase) == out[3, 2, 5]
    assert out[3, 2, 5] == g1["kat_" + conv][5]


def test_phase_delay_bad_convention(g1):
    with pytest.raises(ValueError):
        oracle.phase_delay(g1["lm"], g1["uvw"], g1["frequency"], convention="bob")


# --------------------------------------------------------------------------- predict_vis
@pytest.mark.parametrize("ck", list(CORR))
@pytest.mark.parametrize("dk", list(DDE))
@pytest.mark.parametrize("gk", list(DIE))
def test_predict_vis_27_combos_bit_exact(g2, ck, dk, gk):
    a1j, blj, a2j = DDE[dk]
    g1j, bvis, g2j = DIE[gk]
    get = lambda k: g2["%s_%s" % (ck, k)]
    out = oracle.predict_vis(
        g2["time_idx"], g2["ant1"], g2["ant2"],
        get("a1") if a1j else None, get("bl") if blj else None, get("a2") if a2j else None,
        get("g1") if g1j else None, get("bv") if bvis else None, get("g2") if g2j else None)
    ref = g2["%s_%s_%s_vis" % (ck, dk, gk)]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert_array_equal(out, ref)


@pytest.mark.parametrize("ck", list(CORR))
def test_predict_vis_einsum_identity(g2, ck):
    """africanus/rime/tests/test_predict.py:110-126 (6 decimals there)."""
    sig1, sig2 = (("srcij,srcjk,srclk->rcil", "rcij,rcjk,rclk->rcil") if ck == "c22"
                  else ("srci,srci,srci->rci", "rci,rci,rci->rci"))
    get = lambda k: g2["%s_%s" % (ck, k)]
    ti, a1, a2 = g2["time_idx"], g2["ant1"], g2["ant2"]
    out = oracle.predict_vis(ti, a1, a2, get("a1"), get("bl"), get("a2"), get("g1"), get("bv"), get("g2"))
    v = np.einsum(sig1, get("a1")[:, ti, a1], get("bl"), get("a2")[:, ti, a2].conj()) + get("bv")
    v = np.einsum(sig2, get("g1")[ti, a1], v, get("g2")[ti, a2].conj())
    assert_array_almost_equal(v, out)


@pytest.mark.parametrize("ck", list(CORR))
def test_predict_vis_time_offset_and_int32(g2, ck):
    get = lambda k: g2["%s_%s" % (ck, k)]
    out = oracle.predict_vis((g2["time_idx"] + 10).astype(np.int32), g2["ant1"].astype(np.int32),
                             g2["ant2"].astype(np.int32), get("a1"), get("bl"), get("a2"),
                             get("g1"), get("bv"), get("g2"))
    assert_array_equal(out, g2["%s_offset_vis" % ck])


@pytest.mark.parametrize("ck", list(CORR))
def test_predict_vis_c64(g2, ck):
    get = lambda k: g2["%s_%s" % (ck, k)].astype(np.complex64)
    out = oracle.predict_vis(g2["time_idx"], g2["ant1"], g2["ant2"], get("a1"), get("bl"), get("a2"),
                             get("g1"), get("bv"), get("g2"))
    ref = g2["%s_c64_vis" % ck]
    assert out.dtype == ref.dtype == np.complex64
    assert_array_equal(out, ref)


# --------------------------------------------------------------------------- im_to_vis
@pytest.mark.parametrize("ncorr", [1, 2, 4])
@pytest.mark.parametrize("kind", ["r", "c"])
@pytest.mark.parametrize("conv", ["fourier", "casa"])
def test_im_to_vis_bit_exact(g3, ncorr, kind, conv):
    img = g3["img_%s%d" % (kind, ncorr)]
    out = oracle.im_to_vis(img, g3["uvw"], g3["lm"], g3["frequency"], convention=conv)
    ref = g3["vis_%s%d_%s" % (kind, ncorr, conv)]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert_array_equal(out, ref)


@pytest.mark.parametrize("key, img, freq", [
    ("vis_r4_nonuniform", "img_r4", "frequency_nonuniform"),
    ("vis_c2_nonuniform", "img_c2", "frequency_nonuniform"),
    ("vis_r5_fourier", "img_r5", "frequency"),
    ("vis_r70_fourier", "img_r70", "frequency70"),
])
def test_im_to_vis_other_shapes(g3, key, img, freq):
    out = oracle.im_to_vis(g3[img], g3["uvw"], g3["lm"], g3[freq])
    assert_array_equal(out, g3[key])


def test_im_to_vis_nan_source(g3):
    out = oracle.im_to_vis(g3["img_nan"], g3["uvw"], g3["lm_nan"], g3["frequency"])
    ref = g3["vis_nan"]
    assert np.isnan(ref[:, 2, 1]).all() and np.isnan(ref).sum() == ref.shape[0]
    assert_array_equal(np.isnan(out), np.isnan(ref))
    assert_array_equal(out[~np.isnan(ref)], ref[~np.isnan(ref)])


def test_im_to_vis_c64_output(g3):
    out = oracle.im_to_vis(g3["img_r4"], g3["uvw"], g3["lm"], g3["frequency"], dtype=np.complex64)
    ref = g3["vis_r4_c64"]
    assert out.dtype == ref.dtype == np.complex64
    assert_array_equal(out, ref)


def test_im_to_vis_f32_inputs(g3):
    out = oracle.im_to_vis(g3["img_r4"].astype(np.float32), g3["uvw32"],
                           g3["lm"].astype(np.float32), g3["frequency"].astype(np.float32))
    ref = g3["vis_f32"]
    assert out.dtype == ref.dtype == np.complex64
    # inputs are promoted to float64 in the oracle; numba mixes float32 products in
    assert maxabs(out, ref) < 5e-4 * np.abs(ref).max()


def test_im_to_vis_omp_matches_serial(g3):
    a = oracle.im_to_vis(g3["img_c4"], g3["uvw"], g3["lm"], g3["frequency"], omp=False)
    b = oracle.im_to_vis(g3["img_c4"], g3["uvw"], g3["lm"], g3["frequency"], omp=True)
    assert_array_equal(a, b)


def test_im_to_vis_fft_kat():
    """africanus/dft/tests/test_dft.py:86-133: DFT on a regular grid == FFT."""
    rng = np.random.default_rng(123)
    Fs, iFs = np.fft.fftshift, np.fft.ifftshift
    npix, nsource = 29, 25
    image = np.zeros((npix, npix, 1))
    image[rng.integers(5, npix - 5, nsource), rng.integers(5, npix - 5, nsource), 0] = \
        rng.standard_normal(nsource)
    fft_image = Fs(np.fft.fft2(iFs(image[:, :, 0])))[:, :, None]
    deltal = 0.001
    l_coord = np.arange(-(npix // 2), npix // 2 + 1) * deltal
    ll, mm = np.meshgrid(l_coord, l_coord)
    lm = np.vstack((ll.flatten(), mm.flatten())).T
    u = Fs(np.fft.fftfreq(npix, d=deltal))
    uu, vv = np.meshgrid(u, u)
    uvw = np.zeros((npix**2, 3))
    uvw[:, 0], uvw[:, 1] = uu.flatten(), vv.flatten()
    frequency = np.ones(1) * 2.99792458e8
    for conv in ("fourier", "casa"):
        vis = oracle.im_to_vis(image.reshape(npix**2, 1, 1), uvw, lm, frequency, convention=conv)
        ref = fft_image.reshape(npix**2, 1, 1)
        ref = np.conj(ref) if conv == "casa" else ref
        assert_array_almost_equal(vis, ref, decimal=13)


# --------------------------------------------------------------------------- beams
def test_freq_grid_interp_kat(g4):
    """africanus/rime/tests/test_fast_beams.py:130-150."""
    fd = oracle.freq_grid_interp(g4["freqs"], g4["beam_freq_map"])
    assert_array_equal(fd, g4["freq_data"])
    assert_array_almost_equal(fd[:, 0], [0.8, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.1])
    assert_array_equal(np.int32(fd[:, 2]), [0, 0, 1, 2, 2, 2, 3, 3])
    assert_array_almost_equal(fd[:, 1], [1.0, 1.0, 0.71428571, 1.0, 0.52380952, 0.04761905, 0.0, 0.0])


def _beam_args(g4, lm=None, beam=None, dtype=None):
    args = [g4["beam"] if beam is None else beam, g4["extents"], g4["beam_freq_map"],
            g4["lm"] if lm is None else lm, g4["parangles"], g4["point_errors"],
            g4["antenna_scaling"], g4["freqs"]]
    if dtype is not None:
        args = [a.astype(np.complex64 if np.iscomplexobj(a) else dtype) for a in args]
    return args


def test_beam_cube_dde_golden(g4):
    out = oracle.beam_cube_dde(*_beam_args(g4))
    ref = g4["ddes"]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert maxabs(out, ref) < 1e-15
    out0 = oracle.beam_cube_dde(*_beam_args(g4, lm=g4["lm"][:2], beam=np.zeros_like(g4["beam"])))
    assert_array_equal(out0, g4["ddes_zero"])
    beam1 = np.ascontiguousarray(g4["beam"][..., 0, :1])
    out1 = oracle.beam_cube_dde(*_beam_args(g4, beam=beam1))
    assert maxabs(out1, g4["ddes_1corr"]) < 1e-15


def test_beam_cube_dde_f32(g4):
    out = oracle.beam_cube_dde(*_beam_args(g4, dtype=np.float32))
    ref = g4["ddes_f32"]
    assert out.dtype == ref.dtype == np.complex64
    assert maxabs(out, ref) < 1e-5


def test_beam_cube_dde_reference_kat(g4):
    """africanus/rime/tests/test_fast_beams.py:43-127: seed 42 -> 0.470255+0.4786j."""
    ddes = oracle.beam_cube_dde(g4["kat_beam"], np.asarray([[-1.0, 1.0], [-1.0, 1.0]]),
                                np.asarray([0.0, 1.0]), np.asarray([[0.1, 0.1]]), np.zeros((1, 1)),
                                np.zeros((1, 1, 1, 2)), np.ones((1, 1, 2)), np.asarray([0.3]))
    assert_array_almost_equal([[[[[0.470255 + 0.4786j]]]]], ddes)
    assert maxabs(ddes, g4["kat_ddes"]) < 1e-16
    with pytest.raises(ValueError):
        oracle.beam_cube_dde(g4["kat_beam"][:1], np.asarray([[-1.0, 1.0], [-1.0, 1.0]]),
                             np.asarray([0.0, 1.0]), np.asarray([[0.1, 0.1]]), np.zeros((1, 1)),
                             np.zeros((1, 1, 1, 2)), np.ones((1, 1, 2)), np.asarray([0.3]))


# --------------------------------------------------------------------------- C1 chain
def test_chain_c1_samples_and_checksums(g5):
    """BASELINE config C1 (10k rows, 16 chan, 100 src, 4 corr): oracle vs the
    reference's sampled rows and whole-array checksums."""
    d = synthetic_inputs(seed=int(g5["seed"]), nrow=10000, nchan=16, nsrc=100, nant=7)
    rows = g5["sample_rows"]
    image_r = np.broadcast_to(d["brightness"].real[:, None, :], (100, 16, 4)).copy()
    vis = oracle.im_to_vis(image_r, d["uvw"], d["lm"], d["frequency"], omp=True)
    assert_array_equal(vis[rows], g5["dft_rows"])
    assert abs(vis.sum() - g5["dft_sum"]) <= 1e-9 * abs(g5["dft_sum"])
    assert abs(np.abs(vis).sum() - g5["dft_abssum"]) <= 1e-12 * g5["dft_abssum"]

    phase = oracle.phase_delay(d["lm"], d["uvw"], d["frequency"])
    coh = np.einsum("srf,si->srfi", phase, d["brightness"]).reshape(100, 10000, 16, 2, 2)
    vis2 = oracle.predict_vis(d["time_index"], d["ant1"], d["ant2"], None, coh, None, None, None, None)
    assert_array_equal(vis2[rows], g5["chain_rows"])
    assert abs(np.abs(vis2).sum() - g5["chain_abssum"]) <= 1e-12 * g5["chain_abssum"]

    rng = d["rng"]
    shp = (d["ntime"], d["nant"], 16, 2, 2)
    die = 1.0 + 0.1 * rng.standard_normal(shp) + 0.1j * rng.standard_normal(shp)
    bvis = 0.01 * (rng.standard_normal(vis2.shape) + 1j * rng.standard_normal(vis2.shape))
    vis3 = oracle.predict_vis(d["time_index"], d["ant1"], d["ant2"], None, coh, None, die, bvis, die)
    assert_array_equal(vis3[rows], g5["die_rows"])
    assert abs(np.abs(vis3).sum() - g5["die_abssum"]) <= 1e-12 * g5["die_abssum"]


# --------------------------------------------------------------------------- vis_to_im
@pytest.fixture(scope="module")
def g6():
    from conftest import load_golden
    return load_golden("g6_vis_to_im.npz")


@pytest.mark.parametrize("ncorr", [1, 2, 4])
@pytest.mark.parametrize("conv", ["fourier", "casa"])
def test_vis_to_im_bit_exact(g6, ncorr, conv):
    out = oracle.vis_to_im(g6["vis%d" % ncorr], g6["uvw"], g6["lm"], g6["frequency"], g6["flags%d" % ncorr],
                           convention=conv)
    ref = g6["im%d_%s" % (ncorr, conv)]
    assert out.shape == ref.shape and out.dtype == ref.dtype == np.float64
    assert_array_equal(out, ref)


def test_vis_to_im_other_cases(g6):
    f = lambda *a, **k: oracle.vis_to_im(*a, **k)
    assert_array_equal(f(g6["vis4"], g6["uvw"], g6["lm"], g6["frequency_nonuniform"], g6["flags4"]),
                       g6["im4_nonuniform"])
    assert_array_equal(f(g6["vis4"].real.copy(), g6["uvw"], g6["lm"], g6["frequency"], g6["flags4"]),
                       g6["im4_realvis"])
    assert_array_equal(f(g6["vis4"], g6["uvw"], g6["lm"], g6["frequency"], np.zeros_like(g6["flags4"])),
                       g6["im4_noflags"])
    assert_array_equal(f(g6["vis70"], g6["uvw300"], g6["lm"], g6["frequency70"], g6["flags70"]), g6["im70"])
    out32 = f(g6["vis4"], g6["uvw"], g6["lm"], g6["frequency"], g6["flags4"], dtype=np.float32)
    assert out32.dtype == np.float32
    assert np.abs(out32 - g6["im4_f32"]).max() < 2e-5 * np.abs(g6["im4_f32"]).max()
    nan = f(g6["vis4"], g6["uvw"], g6["lm_nan"], g6["frequency"], g6["flags4"])
    assert_array_equal(np.isnan(nan), np.isnan(g6["im4_nan"]))
    assert np.isnan(nan[2]).all() and not np.isnan(np.delete(nan, 2, axis=0)).any()


def test_vis_to_im_adjointness():
    """africanus/dft/tests/test_dft.py:136-177: <y, R x> == <R^H y, x>."""
    rng = np.random.default_rng(123)
    nsource, nrow, nchan, ncorr = 21, 31, 3, 4
    uvw = 100 * rng.random((nrow, 3))
    lm = 0.01 * rng.standard_normal((nsource, 2))
    frequency = np.arange(1, nchan + 1) * 2.99792458e8
    gamma_im = rng.standard_normal((nsource, nchan, ncorr))
    gamma_vis = rng.standard_normal((nrow, nchan, ncorr))
    flag = np.zeros((nrow, nchan, ncorr), dtype=bool)
    lhs = np.vdot(gamma_vis, oracle.im_to_vis(gamma_im, uvw, lm, frequency)).real
    rhs = np.vdot(oracle.vis_to_im(gamma_vis, uvw, lm, frequency, flag), gamma_im)
    assert abs(lhs - rhs) < 1e-13 * max(1.0, abs(lhs))


# ---- wsclean_predict / spectra (fused term producers) -------------------------------------
def _wsc_args(g7, tag, freq_key=None):
    st = np.where(g7[tag + "_is_gauss"], "GAUSSIAN", "POINT")
    return (g7[tag + "_uvw"], g7[tag + "_lm"], st, g7[tag + "_flux"], g7[tag + "_coeffs"], g7[tag + "_log_poly"],
            g7[tag + "_ref_freq"], g7[tag + "_gauss_shape"], g7[freq_key or tag + "_freq"])


@pytest.mark.parametrize("tag", ["small", "big"])
def test_wsclean_spectra_and_predict_bit_exact(g7, tag):
    """oracle vs africanus.model.wsclean.spectra / africanus.rime.wsclean_predict run here
    (tests/golden/make_golden.py g7): bit for bit, 2 and 4 spectral coefficients."""
    a = _wsc_args(g7, tag)
    assert_array_equal(oracle.spectra(a[3], a[4], a[5], a[6], a[8]), g7[tag + "_spectrum"])
    out = oracle.wsclean_predict(*a)
    assert out.shape == g7[tag + "_vis"].shape and out.dtype == np.complex128
    assert_array_equal(out, g7[tag + "_vis"])


def test_wsclean_predict_nonuniform_and_errors(g7):
    assert_array_equal(oracle.wsclean_predict(*_wsc_args(g7, "small", "small_freq_nonuniform")),
                       g7["small_vis_nonuniform"])
    a = list(_wsc_args(g7, "small"))
    a[2] = np.where(np.arange(a[2].shape[0]) == 3, "DISK", a[2])
    with pytest.raises(ValueError, match="POINT or GAUSSIAN"):
        oracle.wsclean_predict(*a)
    with pytest.raises(ValueError, match="don't match"):
        oracle.spectra(a[3][:-1], a[4], a[5], a[6], a[8])


def test_wsclean_point_sources_equal_casa_dft(g7):
    """a point-only component list is im_to_vis(spectrum[..., None], convention='casa') term by term
    (africanus/rime/wsclean_predict.py:42-47 against africanus/dft/kernels.py:57-67)."""
    a = list(_wsc_args(g7, "small"))
    a[2] = np.full(a[2].shape, "POINT")
    spec = oracle.spectra(a[3], a[4], a[5], a[6], a[8])
    dft = oracle.im_to_vis(spec[:, :, None].copy(), a[0], a[1], a[8], convention="casa")
    assert np.abs(oracle.wsclean_predict(*a) - dft).max() < 1e-13


# ---- term producers: feed_rotation, Gaussian shape ---------------------------------------------------------
def test_feed_rotation_and_gaussian_shape_bit_exact(g8):
    """oracle vs africanus.rime.feed_rotation / africanus.model.shape.gaussian run here (make_golden.py g8)"""
    assert_array_equal(oracle.feed_rotation(g8["pa"], "linear"), g8["feed_linear"])
    assert_array_equal(oracle.feed_rotation(g8["pa"], "circular"), g8["feed_circular"])
    f32 = oracle.feed_rotation(g8["pa"].astype(np.float32), "linear")
    assert f32.dtype == np.complex64 and np.abs(f32 - g8["feed_linear_f32"]).max() < 2e-7
    assert_array_equal(oracle.gaussian_shape(g8["uvw"], g8["freq"], g8["shape_params"]), g8["gauss"])
    with pytest.raises(ValueError, match="Invalid feed_type"):
        oracle.feed_rotation(g8["pa"], "elliptical")


def test_feed_rotation_reference_kat():
    """africanus/rime/tests/test_rime.py:50-77: the matrices spelled out"""
    pa = np.random.default_rng(0).random((10, 5))
    fr = oracle.feed_rotation(pa, "linear")
    assert_array_equal(fr, np.array([[np.cos(pa), np.sin(pa)], [-np.sin(pa), np.cos(pa)]]).transpose(2, 3, 0, 1))
    fc = oracle.feed_rotation(pa, "circular")
    ref = np.zeros(pa.shape + (2, 2), np.complex128)
    ref[..., 0, 0] = np.cos(pa) - 1j * np.sin(pa)
    ref[..., 1, 1] = np.cos(pa) + 1j * np.sin(pa)
    assert_array_equal(fc, ref)


# ---- calibration consumers: corrupt_vis, residual_vis, correct_vis -------------------------------------------
CALIB_TAGS = ["dd1", "dd2", "diag", "full"]


@pytest.mark.parametrize("tag", CALIB_TAGS)
def test_calibration_utils_bit_exact(g9, tag):
    """oracle vs africanus.calibration.utils.{corrupt,residual,correct}_vis run here (make_golden.py g9), the
    four layouts of calibration/utils/tests/test_utils.py:10-18"""
    a = (g9["tbin_idx"], g9["tbin_counts"], g9["ant1"], g9["ant2"])
    assert_array_equal(oracle.corrupt_vis(*a, g9[tag + "_jones"], g9[tag + "_model"]), g9[tag + "_vis"])
    assert_array_equal(oracle.residual_vis(*a, g9[tag + "_jones"], g9[tag + "_data"], g9[tag + "_flag"],
                                           g9[tag + "_model"]), g9[tag + "_residual"])
    j1 = np.ascontiguousarray(g9[tag + "_jones"][:, :, :, :1])
    assert_array_equal(oracle.correct_vis(*a, j1, g9[tag + "_data"], g9[tag + "_flag"]), g9[tag + "_corrected"])


@pytest.mark.parametrize("tag", CALIB_TAGS)
def test_corrupt_vis_equals_predict_vis(g9, tag):
    """africanus/calibration/utils/tests/test_utils.py:21-78: corrupt_vis == predict_vis on transposed arrays"""
    jones, model = g9[tag + "_jones"], g9[tag + "_model"]
    if tag == "diag":
        full = np.zeros(jones.shape[:4] + (2, 2), np.complex128)
        full[..., 0, 0], full[..., 1, 1] = jones[..., 0], jones[..., 1]
        jones = full
    nd = jones.ndim
    jt = np.ascontiguousarray(np.transpose(jones, [3, 0, 1, 2] + list(range(4, nd))))
    mt = np.ascontiguousarray(np.transpose(model, [2, 0, 1] + list(range(3, model.ndim))))
    time_index = np.unique(g9["time"], return_inverse=True)[1]
    ref = oracle.predict_vis(time_index, g9["ant1"], g9["ant2"], jt, mt, jt, None, None, None)
    assert_array_almost_equal(ref, g9[tag + "_vis"], decimal=10)


# ---- convolutional degridder (Perley polyhedron, BASELINE configs[4]) -----------------------------------------
DEGRID_CASES = [
    ("packed_I4", "None", "XXXYYXYY_FROM_I", "conv_1d_axisymmetric_packed_gather", "pkern", "phase_centre"),
    ("unpacked_I2", "None", "XXYY_FROM_I", "conv_1d_axisymmetric_unpacked_gather", "kern", "phase_centre"),
    ("packed_V4_rot", "phase_rotate", "XXXYYXYY_FROM_V", "conv_1d_axisymmetric_packed_gather", "pkern", "image_centre"),
    ("packed_Q2_rot", "phase_rotate", "XXYY_FROM_Q", "conv_1d_axisymmetric_packed_gather", "pkern", "image_centre"),
    ("unpacked_U4", "None", "RRRLLRLL_FROM_U", "conv_1d_axisymmetric_unpacked_gather", "kern", "phase_centre"),
]


@pytest.mark.parametrize("tag, ppol, spol, cpol, kern, centre", DEGRID_CASES)
def test_degridder_oracle_vs_reference(g10, tag, ppol, spol, cpol, kern, centre):
    """oracle vs africanus.gridding.perleypolyhedron.degridder.degridder_serial run here (make_golden.py g10).
    The reference is compiled with fastmath=True, so agreement is to rounding: 1e-14 without the facet phase
    rotation, 1e-10 with it (a phase of ~1e4 rad re-associated)."""
    out = oracle.degridder(g10["uvw"], g10["grid"], g10["wavelengths"], g10["chanmap"], float(g10["cell"]), g10[centre],
                           g10["phase_centre"], g10[kern], int(g10["W"]), int(g10["OS"]), "None", ppol, spol, cpol)
    ref = g10[tag]
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= (1e-10 if ppol == "phase_rotate" else 1e-14) * np.abs(ref).max()


def test_degridder_kernels_match_reference(g10):
    """numpy kernel generators against the reference's kbsinc / pack_kernel outputs stored with the golden"""
    from codex_africanus_amd.gridding.perleypolyhedron import kernels
    W, OS = int(g10["W"]), int(g10["OS"])
    k = kernels.kbsinc(W, oversample=OS)
    np.testing.assert_allclose(k, g10["kern"], rtol=1e-13, atol=1e-16)
    assert_array_equal(kernels.pack_kernel(g10["kern"], W, OS), g10["pkern"])
    assert_array_equal(kernels.unpack_kernel(g10["pkern"], W, OS), g10["kern"])
    assert kernels.uspace(W, OS).shape == (OS * (W + 2),)
    assert abs(kernels.sinc(W, OS).sum() - 1) < 1e-14 and abs(kernels.hanningsinc(W, oversample=OS).sum() - 1) < 1e-14


def test_spectral_model_bit_exact(g8):
    """oracle vs africanus.model.spectral.spectral_model run here (make_golden.py g8): the three bases, a
    per-polarisation list, and the no-polarisation form"""
    a = (g8["stokes"], g8["spi"], g8["spec_ref_freq"], g8["freq"])
    for key, base in (("spec_std", 0), ("spec_log", 1), ("spec_log10", 2), ("spec_list", [0, 1, 2])):
        assert_array_equal(oracle.spectral_model(*a, base=base), g8[key])
    assert_array_equal(oracle.spectral_model(g8["stokes"][:, 0].copy(), g8["spi"][:, :, 0].copy(), g8["spec_ref_freq"],
                                             g8["freq"], base="log"), g8["spec_nopol"])


@pytest.mark.parametrize("tag", CALIB_TAGS)
def test_compute_and_corrupt_vis_bit_exact(g9, tag):
    """oracle vs africanus.calibration.utils.compute_and_corrupt_vis run here (make_golden.py g9)"""
    a = (g9["tbin_idx"], g9["tbin_counts"], g9["ant1"], g9["ant2"])
    out = oracle.compute_and_corrupt_vis(*a, g9[tag + "_jones"], g9[tag + "_tmodel"], g9["cc_uvw"], g9["cc_freq"], g9["cc_lm"])
    assert_array_equal(out, g9[tag + "_ccvis"])


GRID_CASES = [
    ("grid_unpacked_I2", "gvis2", "None", "I_FROM_XXYY", "conv_1d_axisymmetric_unpacked_scatter", "kern", "phase_centre", False),
    ("grid_packed_V4_rot_norm", "gvis4", "phase_rotate", "V_FROM_XXXYYXYY", "conv_1d_axisymmetric_packed_scatter", "pkern",
     "image_centre", True),
    ("grid_packed_Q2_rot", "gvis2", "phase_rotate", "Q_FROM_XXYY", "conv_1d_axisymmetric_packed_scatter", "pkern",
     "image_centre", False),
    ("grid_nn_U4", "gvis4", "None", "U_FROM_RRRLLRLL", "conv_nn_scatter", "kern", "phase_centre", True),
]


@pytest.mark.parametrize("tag, vkey, ppol, spol, cpol, kern, centre, norm", GRID_CASES)
def test_gridder_oracle_vs_reference(g10, tag, vkey, ppol, spol, cpol, kern, centre, norm):
    """oracle vs africanus.gridding.perleypolyhedron.gridder.gridder run here (make_golden.py g10); fastmath in the
    reference: agreement to rounding"""
    rows = g10["grid_nn_rows"] if cpol == "conv_nn_scatter" else np.ones(g10["uvw"].shape[0], bool)
    out = oracle.gridder(g10["uvw"][rows], g10[vkey][rows], g10["wavelengths"], g10["chanmap"], 64, float(g10["cell"]),
                         g10[centre], g10["phase_centre"], g10[kern], int(g10["W"]), int(g10["OS"]), "None", ppol, spol,
                         cpol, do_normalize=norm)
    ref = g10[tag]
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= (1e-10 if ppol == "phase_rotate" else 1e-14) * np.abs(ref).max()


def test_convert_oracle_matches_reference(g11):
    """Stokes <-> correlation convert (africanus/model/coherency/conversion.py:207): every schema pair of the
    reference's test_convert.py:13-28 plus circular / implicit-Stokes / two-candidate cases, four input dtypes;
    values, dtype and shape identical."""
    import json
    cases = json.loads(str(g11["cases"]))
    assert len(cases) >= 22
    for i, (isch, osch, implicit) in enumerate(cases):
        for kind in ("f64", "c128", "f32", "c64"):
            ref = g11["out_%d_%s" % (i, kind)]
            got = oracle.convert(g11["in_%d_%s" % (i, kind)], isch, osch, implicit)
            assert got.dtype == ref.dtype and got.shape == ref.shape, (i, kind)
            assert np.array_equal(got, ref), (i, kind)


def test_convert_oracle_non_finite(g11):
    import json
    with np.errstate(all="ignore"):
        for j, (isch, osch) in enumerate(json.loads(str(g11["nf_cases"]))):
            for kind in ("real", "cplx"):
                got, ref = oracle.convert(g11["nf_in_" + kind], isch, osch), g11["nf_out_%d_%s" % (j, kind)]
                for part in (np.real, np.imag):
                    assert np.array_equal(part(got), part(ref), equal_nan=True), (isch, osch, kind)


def test_convert_reference_kat():
    """The known answers of model/coherency/tests/test_convert.py:66-134."""
    I, Q, U, V = 1.0 + 1j, 2.0 + 2j, 3.0 + 3j, 4.0 + 4j
    x = np.asarray([[I, Q, U, V]])
    lin = oracle.convert(x, ["I", "Q", "U", "V"], ["XX", "XY", "YX", "YY"])
    assert np.all(lin == [[I + Q, U + V * 1j, U - V * 1j, I - Q]])
    circ = oracle.convert(x, [1, 2, 3, 4], [5, 6, 7, 8])
    assert np.all(circ == [[I + V, Q + U * 1j, Q - U * 1j, I - V]])
    assert np.all(oracle.convert(lin, ["XX", "XY", "YX", "YY"], ["I", "Q", "U", "V"]) == x)
    assert np.all(oracle.convert(circ, ["RR", "RL", "LR", "LL"], ["I", "Q", "U", "V"]) == x)
    v = oracle.convert(np.asarray([I]), ["I"], ["XX", "XY", "YX", "YY"], implicit_stokes=True)
    assert v[0] == I and v[-1] == I


@pytest.mark.parametrize("band", ["rising40", "falling33", "rising80"])
@pytest.mark.parametrize("conv", ["fourier", "casa"])
def test_gaussian_predict_chain_against_the_reference(band, conv):
    """G15 (tests/golden/make_golden_gauss.py): the reference's phase_delay x gaussian shape x brightness summed by
    predict_vis, africanus/rime/examples/predict.py:107-134 -- the oracle chain the Gaussian kernels are checked against
    reproduces it to rounding (the einsum's summation order is numpy's)."""
    from conftest import load_golden
    g = load_golden("g15_gauss.npz")
    freq, X = g["frequency_" + band], g["brightness_" + band]
    phase = oracle.phase_delay(g["lm"], g["uvw"], freq, conv)
    shape = oracle.gaussian_shape(g["uvw"], freq, g["shape_params"])
    coh = np.einsum("srf,srf,sfij->srfij", phase, shape, X)
    vis = oracle.predict_vis(g["time_index"], g["antenna1"], g["antenna2"], None, coh, None, None, None, None)
    ref = g["vis_%s_%s" % (band, conv)]
    assert vis.shape == ref.shape
    assert np.abs(vis - ref).max() <= 1e-13 * float(g["scale_" + band])


@pytest.mark.parametrize("case", ["beam", "beam_die", "beam_feed"])
def test_fused_chain_on_measurement_set_uvw_against_the_reference(case):
    """G16 (tests/golden/make_golden_gemm.py): the oracle chain phase_delay -> einsum -> beam_cube_dde [-> feed rotation] ->
    predict_vis reproduces the reference's on antenna-decomposable uvw (what the GEMM-form tests compare with)."""
    from conftest import load_golden
    from fused_cases import linear_feed_rotation, oracle_fused_predict_vis
    g, h = load_golden("g14_fused_dask.npz"), load_golden("g16_fused_gemm.npz")
    kw = {}
    if case == "beam_die":
        kw = dict(die1_jones=g["die"], base_vis=g["base_vis"], die2_jones=g["die"])
    if case == "beam_feed":
        kw = dict(feed_rotation=linear_feed_rotation(g["parallactic_angles"]))
    vis = oracle_fused_predict_vis(g["time_index"], g["antenna1"], g["antenna2"], g["lm"], h["uvw"], g["frequency"], g["brightness"],
                                   g["beam"], g["beam_lm_extents"], g["beam_freq_map"], g["parallactic_angles"],
                                   g["point_errors"], g["antenna_scaling"], **kw)
    ref = h["vis_" + case]
    assert vis.shape == ref.shape and np.abs(vis - ref).max() <= 1e-12 * max(float(h["scale"]), float(np.abs(ref).max()))
