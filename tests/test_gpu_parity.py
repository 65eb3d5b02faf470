"""
GPU parity tests: the HIP path (through the C ABI) against the golden vectors captured
from the real reference and against the CPU oracle on seeded inputs.  Tolerances:
  predict_vis   bit-exact (integer-style equality of every float)
  phase_delay   |err| <= 4e-16 * max(1, |phase|)-free bound: 1e-15 absolute (f64 sincos, <=1 ulp each side)
  im_to_vis     exact mode 1e-12 relative to sum_s |image|; recurrence mode 1e-8 absolute (north star)
  beam_cube_dde 1e-14 absolute (f64), 1e-5 (f32)
"""
import numpy as np
import pytest
from numpy.testing import assert_array_equal

import oracle
from codex_africanus_amd import rime, dft
from codex_africanus_amd.dft import kernels as dft_kernels
from codex_africanus_amd.testing import synthetic_inputs, real_image

pytestmark = pytest.mark.gpu

CORR = {"c1": (1,), "c2": (2,), "c22": (2, 2)}
DDE = {"ddecoh": (True, True, True), "dde": (True, False, True), "coh": (False, True, False)}
DIE = {"diebv": (True, True, True), "die": (True, False, True), "bv": (False, True, False)}


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) if np.size(a) else 0.0


@pytest.fixture
def dft_mode():
    old = dft_kernels.get_mode()
    yield dft_kernels.set_mode
    dft_kernels.set_mode(old)


# ---------------------------------------------------------------------------- phase_delay
@pytest.mark.parametrize("conv", ["fourier", "casa"])
def test_phase_delay_f64_golden(g1, conv):
    out = rime.phase_delay(g1["lm"], g1["uvw"], g1["frequency"], convention=conv)
    ref = g1["f64_" + conv]
    assert out.dtype == ref.dtype and out.shape == ref.shape
    assert maxabs(out, ref) < 1e-15


@pytest.mark.parametrize("conv", ["fourier", "casa"])
def test_phase_delay_f32_golden(g1, conv):
    out = rime.phase_delay(g1["lm32"], g1["uvw32"], g1["frequency32"], convention=conv)
    ref = g1["f32_" + conv]
    assert out.dtype == ref.dtype == np.complex64
    assert maxabs(out, ref) < 2e-6


@pytest.mark.parametrize("conv, sign", [("fourier", 1), ("casa", -1)])
def test_phase_delay_reference_kat(g1, conv, sign):
    """africanus/rime/tests/test_rime.py:19-47."""
    out = rime.phase_delay(g1["kat_lm"], g1["kat_uvw"], g1["kat_frequency"], convention=conv)
    minus_two_pi_over_c = -2 * np.pi / 2.99792458e8
    n = np.sqrt(1.0 - 0.1**2 - 0.2**2) - 1.0
    phase = sign * minus_two_pi_over_c * (1 * 0.1 + 2 * 0.2 + 3 * n) * 0.856e9
    assert abs(np.exp(1j * phase) - out[3, 2, 5]) < 5e-16
    assert maxabs(out[3, 2], g1["kat_" + conv]) < 1e-15


def test_phase_delay_vs_oracle_seeded():
    d = synthetic_inputs(seed=11, nrow=3000, nchan=16, nsrc=37)
    out = rime.phase_delay(d["lm"], d["uvw"], d["frequency"])
    ref = oracle.phase_delay(d["lm"], d["uvw"], d["frequency"])
    assert maxabs(out, ref) < 1e-15
    assert np.abs(np.abs(out) - 1.0).max() < 1e-15       # unit modulus


def test_phase_delay_torch_device_resident():
    import torch
    d = synthetic_inputs(seed=12, nrow=500, nchan=8, nsrc=5)
    t = lambda a: torch.from_numpy(a).cuda()
    out = rime.phase_delay(t(d["lm"]), t(d["uvw"]), t(d["frequency"]))
    assert isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.complex128
    assert_array_equal(out.cpu().numpy(), rime.phase_delay(d["lm"], d["uvw"], d["frequency"]))


def test_phase_delay_empty():
    out = rime.phase_delay(np.zeros((0, 2)), np.zeros((4, 3)), np.ones(3))
    assert out.shape == (0, 4, 3)


# ---------------------------------------------------------------------------- predict_vis
@pytest.mark.parametrize("ck", list(CORR))
@pytest.mark.parametrize("dk", list(DDE))
@pytest.mark.parametrize("gk", list(DIE))
def test_predict_vis_27_combos_bit_exact(g2, ck, dk, gk):
    a1j, blj, a2j = DDE[dk]
    g1j, bvis, g2j = DIE[gk]
    get = lambda k: g2["%s_%s" % (ck, k)]
    out = rime.predict_vis(
        g2["time_idx"], g2["ant1"], g2["ant2"],
        get("a1") if a1j else None, get("bl") if blj else None, get("a2") if a2j else None,
        get("g1") if g1j else None, get("bv") if bvis else None, get("g2") if g2j else None)
    ref = g2["%s_%s_%s_vis" % (ck, dk, gk)]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert_array_equal(out, ref)


@pytest.mark.parametrize("ck", list(CORR))
def test_predict_vis_offset_int32_and_c64(g2, ck):
    get = lambda k: g2["%s_%s" % (ck, k)]
    out = rime.predict_vis((g2["time_idx"] + 10).astype(np.int32), g2["ant1"].astype(np.int32),
                           g2["ant2"].astype(np.int32), get("a1"), get("bl"), get("a2"),
                           get("g1"), get("bv"), get("g2"))
    assert_array_equal(out, g2["%s_offset_vis" % ck])
    g64 = lambda k: get(k).astype(np.complex64)
    out64 = rime.predict_vis(g2["time_idx"], g2["ant1"], g2["ant2"], g64("a1"), g64("bl"), g64("a2"),
                             g64("g1"), g64("bv"), g64("g2"))
    assert out64.dtype == np.complex64
    assert_array_equal(out64, g2["%s_c64_vis" % ck])


def test_apply_gains(g2):
    get = lambda k: g2["c22_%s" % k]
    out = rime.apply_gains(g2["time_idx"], g2["ant1"], g2["ant2"], get("g1"), get("bv"), get("g2"))
    ref = oracle.apply_gains(g2["time_idx"], g2["ant1"], g2["ant2"], get("g1"), get("bv"), get("g2"))
    assert_array_equal(out, ref)


def test_predict_vis_chain_c1_golden(g5):
    """BASELINE config C1 chain: phase_delay -> einsum -> predict_vis (+ DIEs, base_vis)."""
    d = synthetic_inputs(seed=int(g5["seed"]), nrow=10000, nchan=16, nsrc=100, nant=7)
    rows = g5["sample_rows"]
    phase = oracle.phase_delay(d["lm"], d["uvw"], d["frequency"])   # caller-side input
    coh = np.einsum("srf,si->srfi", phase, d["brightness"]).reshape(100, 10000, 16, 2, 2)
    vis = rime.predict_vis(d["time_index"], d["ant1"], d["ant2"], None, coh, None, None, None, None)
    assert_array_equal(vis[rows], g5["chain_rows"])
    assert abs(np.abs(vis).sum() - g5["chain_abssum"]) <= 1e-12 * g5["chain_abssum"]
    rng = d["rng"]
    shp = (d["ntime"], d["nant"], 16, 2, 2)
    die = 1.0 + 0.1 * rng.standard_normal(shp) + 0.1j * rng.standard_normal(shp)
    bvis = 0.01 * (rng.standard_normal(vis.shape) + 1j * rng.standard_normal(vis.shape))
    vis3 = rime.predict_vis(d["time_index"], d["ant1"], d["ant2"], None, coh, None, die, bvis, die)
    assert_array_equal(vis3[rows], g5["die_rows"])


def test_predict_vis_dde_vs_oracle_seeded():
    rng = np.random.default_rng(5)
    s, t, a, c, r = 9, 6, 5, 7, 333
    rc = lambda *sh: rng.standard_normal(sh) + 1j * rng.standard_normal(sh)
    ti = np.sort(rng.integers(0, t, r)) + 3
    a1, a2 = rng.integers(0, a, r), rng.integers(0, a, r)
    dde, coh, die, bv = rc(s, t, a, c, 2, 2), rc(s, r, c, 2, 2), rc(t, a, c, 2, 2), rc(r, c, 2, 2)
    out = rime.predict_vis(ti, a1, a2, dde, coh, dde, die, bv, die)
    ref = oracle.predict_vis(ti, a1, a2, dde, coh, dde, die, bv, die)
    assert_array_equal(out, ref)


def test_predict_vis_empty_rows():
    out = rime.predict_vis(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32),
                           source_coh=np.zeros((3, 0, 4, 2, 2), np.complex128))
    assert out.shape == (0, 4, 2, 2)


# ---------------------------------------------------------------------------- im_to_vis
def _scale(img):
    return float(np.abs(img).sum(axis=0).max())


@pytest.mark.parametrize("mode, rtol", [("exact", 1e-14), ("auto", 1e-11), ("valu", 1e-11)])
@pytest.mark.parametrize("ncorr", [1, 2, 4])
@pytest.mark.parametrize("kind", ["r", "c"])
@pytest.mark.parametrize("conv", ["fourier", "casa"])
def test_im_to_vis_golden(g3, dft_mode, mode, rtol, ncorr, kind, conv):
    dft_mode(mode)
    img = g3["img_%s%d" % (kind, ncorr)]
    out = dft.im_to_vis(img, g3["uvw"], g3["lm"], g3["frequency"], convention=conv)
    ref = g3["vis_%s%d_%s" % (kind, ncorr, conv)]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert maxabs(out, ref) <= rtol * _scale(img)


@pytest.mark.parametrize("mode, rtol", [("exact", 1e-14), ("auto", 1e-11), ("valu", 1e-11)])
@pytest.mark.parametrize("key, img, freq", [
    ("vis_r4_nonuniform", "img_r4", "frequency_nonuniform"),
    ("vis_c2_nonuniform", "img_c2", "frequency_nonuniform"),
    ("vis_r5_fourier", "img_r5", "frequency"),
    ("vis_r70_fourier", "img_r70", "frequency70"),
])
def test_im_to_vis_other_shapes(g3, dft_mode, mode, rtol, key, img, freq):
    dft_mode(mode)
    out = dft.im_to_vis(g3[img], g3["uvw"], g3["lm"], g3[freq])
    assert maxabs(out, g3[key]) <= rtol * _scale(g3[img])


def test_im_to_vis_nonuniform_auto_takes_exact_path(g3, dft_mode):
    dft_mode("auto")
    a = dft.im_to_vis(g3["img_r4"], g3["uvw"], g3["lm"], g3["frequency_nonuniform"])
    dft_mode("exact")
    b = dft.im_to_vis(g3["img_r4"], g3["uvw"], g3["lm"], g3["frequency_nonuniform"])
    assert_array_equal(a, b)


@pytest.mark.parametrize("mode", ["exact", "auto", "valu"])
def test_im_to_vis_nan_source_semantics(g3, dft_mode, mode):
    """A source outside the unit disc poisons only its non-zero pixels' columns (kernels.py:54,64)."""
    dft_mode(mode)
    out = dft.im_to_vis(g3["img_nan"], g3["uvw"], g3["lm_nan"], g3["frequency"])
    ref = g3["vis_nan"]
    assert_array_equal(np.isnan(out), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert maxabs(out[ok], ref[ok]) <= 1e-11 * _scale(g3["img_nan"])


def test_im_to_vis_all_zero_column_stays_zero(g3, dft_mode):
    dft_mode("auto")
    img = g3["img_r4"].copy()
    img[:, 3, 2] = 0.0
    uvw = g3["uvw"].copy()
    uvw[7] = np.nan                      # NaN row: only non-zero columns become NaN
    out = dft.im_to_vis(img, uvw, g3["lm"], g3["frequency"])
    ref = oracle.im_to_vis(img, uvw, g3["lm"], g3["frequency"])
    assert_array_equal(np.isnan(out), np.isnan(ref))
    assert (out[:, 3, 2] == 0).all()


def test_im_to_vis_dtype_rules(g3, dft_mode):
    dft_mode("exact")
    out = dft.im_to_vis(g3["img_r4"], g3["uvw"], g3["lm"], g3["frequency"], dtype=np.complex64)
    assert out.dtype == np.complex64
    assert maxabs(out, g3["vis_r4_c64"]) < 2e-6 * _scale(g3["img_r4"])
    out32 = dft.im_to_vis(g3["img_r4"].astype(np.float32), g3["uvw32"], g3["lm"].astype(np.float32),
                          g3["frequency"].astype(np.float32))
    assert out32.dtype == np.complex64
    assert maxabs(out32, g3["vis_f32"]) < 5e-4 * np.abs(g3["vis_f32"]).max()


def test_im_to_vis_fft_kat(dft_mode):
    """africanus/dft/tests/test_dft.py:86-133: DFT on a regular grid == FFT (decimal 13)."""
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
    for mode in ("exact", "auto"):
        dft_mode(mode)
        for conv in ("fourier", "casa"):
            vis = dft.im_to_vis(image.reshape(npix**2, 1, 1), uvw, lm, frequency, convention=conv)
            ref = fft_image.reshape(npix**2, 1, 1)
            ref = np.conj(ref) if conv == "casa" else ref
            np.testing.assert_array_almost_equal(vis, ref, decimal=13)


def test_im_to_vis_phase_centre(dft_mode):
    """africanus/dft/tests/test_dft.py:12-42: a single source at the phase centre."""
    dft_mode("auto")
    rng = np.random.default_rng(0)
    nrow, npix, nchan, ncorr = 100, 35, 11, 2
    uvw = rng.random((nrow, 3))
    x = np.linspace(-0.1, 0.1, npix)
    ll, mm = np.meshgrid(x, x)
    lm = np.vstack((ll.flatten(), mm.flatten())).T
    frequency = np.linspace(1.0, 2.0, nchan)
    image = np.zeros((npix, npix, nchan, ncorr))
    Inu = (frequency / frequency[nchan // 2]) ** (-0.7)
    image[npix // 2, npix // 2] = Inu[:, None]
    vis = dft.im_to_vis(image.reshape(npix**2, nchan, ncorr), uvw, lm, frequency)
    assert np.abs(vis - Inu[None, :, None]).max() < 1e-13


@pytest.mark.parametrize("mode", ["exact", "auto"])
def test_im_to_vis_analytic_sum(dft_mode, mode):
    """africanus/dft/tests/test_dft.py:45-84: several channels and sources, one correlation, against the phasor
    sum written out with numpy's complex exponential (decimal=14 there)."""
    from codex_africanus_amd.constants import minus_two_pi_over_c
    dft_mode(mode)
    rng = np.random.default_rng(123)
    nrow, nchan, nsource = 100, 3, 5
    uvw = rng.random((nrow, 3))
    frequency = np.linspace(1.0e9, 2.0e9, nchan)
    image = (rng.standard_normal(nsource)[:, None] * (frequency / frequency[nchan // 2]) ** (-0.7))[:, :, None]
    lm = 0.001 + 0.1 * rng.random((nsource, 2))
    vis = dft.im_to_vis(image, uvw, lm, frequency)[..., 0]
    n = np.sqrt(1.0 - lm[:, 0] ** 2 - lm[:, 1] ** 2)
    path = uvw[:, 0, None] * lm[None, :, 0] + uvw[:, 1, None] * lm[None, :, 1] + uvw[:, 2, None] * (n[None, :] - 1)
    expect = np.einsum("rsf,sf->rf", np.exp(1j * minus_two_pi_over_c * path[:, :, None] * frequency[None, None, :]),
                       image[:, :, 0])
    np.testing.assert_array_almost_equal(vis, expect, decimal=13 if mode == "auto" else 14)


def test_symmetric_covariance(dft_mode):
    """africanus/dft/tests/test_dft.py:297-331: the image-plane precision matrix R^H R sampled at the source
    positions (im_to_vis of a unit source followed by vis_to_im) is symmetric."""
    dft_mode("auto")
    rng = np.random.default_rng(123)
    nsource, nrows = 25, 1000
    lm = -0.05 + 0.1 * rng.random((nsource, 2))
    uvw = rng.standard_normal((nrows, 3)) * 1000
    uvw[:, 2] = 0.0
    freq = np.array([1.0e9])
    flags = np.zeros((nrows, 1, 1), dtype=np.bool_)
    unit = np.ones((1, 1, 1))
    psf = np.zeros((nsource, nsource))
    for j in range(nsource):
        k = dft.im_to_vis(unit, uvw, lm[j:j + 1], freq)
        psf[:, j] = dft.vis_to_im(k, uvw, lm, freq, flags)[:, 0, 0]
    assert np.abs(psf - psf.T).max() < 1e-13 * nrows      # entries are sums of `nrows` unit phasors
    assert np.abs(np.diag(psf) - nrows).max() < 1e-10


@pytest.mark.parametrize("mode, tol", [("exact", 2e-12), ("auto", 1e-8)])
def test_im_to_vis_c1_golden(g5, dft_mode, mode, tol):
    """BASELINE config C1 (10k rows, 16 chan, 100 src, 4 corr) against the reference's rows."""
    dft_mode(mode)
    d = synthetic_inputs(seed=int(g5["seed"]), nrow=10000, nchan=16, nsrc=100, nant=7)
    vis = dft.im_to_vis(real_image(d), d["uvw"], d["lm"], d["frequency"])
    assert maxabs(vis[g5["sample_rows"]], g5["dft_rows"]) < tol
    assert abs(np.abs(vis).sum() - g5["dft_abssum"]) <= 1e-9 * g5["dft_abssum"]


def test_im_to_vis_linearity_and_row_independence(dft_mode):
    """Size-independent properties: linear in the image; a row's result does not depend on
    which other rows are in the call (row sharding is exact)."""
    dft_mode("auto")
    d = synthetic_inputs(seed=21, nrow=5000, nchan=64, nsrc=50)
    img = real_image(d)
    a = dft.im_to_vis(img, d["uvw"], d["lm"], d["frequency"])
    b = dft.im_to_vis(2.0 * img, d["uvw"], d["lm"], d["frequency"])
    assert_array_equal(b, 2.0 * a)       # power-of-two scaling commutes with rounding
    part = dft.im_to_vis(img, d["uvw"][1234:2345], d["lm"], d["frequency"])
    assert_array_equal(part, a[1234:2345])
    ref = oracle.im_to_vis(img, d["uvw"][:300], d["lm"], d["frequency"])
    assert maxabs(a[:300], ref) < 1e-8


@pytest.mark.parametrize("nchan", [14, 16, 17, 32, 33, 48, 64, 65, 96, 100, 130])
def test_im_to_vis_mfma_channel_tilings(dft_mode, nchan):
    """real 4-correlation images on a uniform band take the MFMA-accumulator kernels: every tile
    combination (64-wide tiles, 16- and 32-wide last tiles, ragged ends), row counts that do not fill
    a 64-row block and source counts that do not fill a 4-source step; against the oracle and
    against the VALU kernels."""
    rng = np.random.default_rng(nchan)
    nrow, nsrc = 203, 37
    uvw = rng.standard_normal((nrow, 3)) * 3000.0
    lm = rng.standard_normal((nsrc, 2)) * 0.03
    freq = np.linspace(0.9e9, 1.7e9, nchan)
    img = rng.standard_normal((nsrc, nchan, 4))
    ref = oracle.im_to_vis(img, uvw, lm, freq)
    dft_mode("auto")
    out = dft.im_to_vis(img, uvw, lm, freq)
    dft_mode("valu")
    valu = dft.im_to_vis(img, uvw, lm, freq)
    assert maxabs(out, ref) <= 1e-11 * _scale(img)
    assert maxabs(valu, ref) <= 1e-11 * _scale(img)
    dft_mode("auto")
    assert_array_equal(dft.im_to_vis(img, uvw[50:117], lm, freq), out[50:117])   # row independence
    casa = dft.im_to_vis(img, uvw, lm, freq, convention="casa")
    assert_array_equal(casa, np.conj(out))
    # complex pixels: four MFMAs per channel
    cimg = img + 1j * rng.standard_normal((nsrc, nchan, 4))
    cref = oracle.im_to_vis(cimg, uvw, lm, freq)
    cout = dft.im_to_vis(cimg, uvw, lm, freq)
    dft_mode("valu")
    cvalu = dft.im_to_vis(cimg, uvw, lm, freq)
    assert maxabs(cout, cref) <= 1e-11 * _scale(cimg)
    assert maxabs(cvalu, cref) <= 1e-11 * _scale(cimg)


def test_im_to_vis_mfma_special_columns(dft_mode):
    """zero-pixel and NaN-source semantics (kernels.py:54,64) on the MFMA path, 70 channels"""
    rng = np.random.default_rng(5)
    nrow, nsrc, nchan = 130, 11, 70
    uvw = rng.standard_normal((nrow, 3)) * 1000.0
    lm = rng.standard_normal((nsrc, 2)) * 0.05
    lm[4] = [0.9, 0.8]                         # outside the unit disc: n is NaN
    freq = np.linspace(1.0e9, 1.5e9, nchan)
    img = rng.standard_normal((nsrc, nchan, 4))
    img[:, 5, 1] = 0.0                         # an all-zero column stays exactly zero
    img[4, 40:, :] = 0.0                       # the NaN source only poisons channels < 40 ...
    img[4, :, 3] = 0.0                         # ... and not correlation 3
    uvw[17] = np.nan                           # a NaN row poisons its non-zero columns only
    for image in (img, img + 1j * np.where(img != 0, rng.standard_normal(img.shape), 0.0)):
        ref = oracle.im_to_vis(image, uvw, lm, freq)
        for mode in ("auto", "valu"):
            dft_mode(mode)
            out = dft.im_to_vis(image, uvw, lm, freq)
            assert_array_equal(np.isnan(out), np.isnan(ref))
            ok = ~np.isnan(ref)
            assert maxabs(out[ok], ref[ok]) <= 1e-11 * _scale(np.nan_to_num(image))
            assert (out[:, 5, 1] == 0).all()


@pytest.mark.parametrize("nchan", [14, 16, 33, 64, 70, 100, 130])
def test_vis_to_im_mfma_channel_tilings(dft_mode, nchan):
    """4-correlation vis_to_im on a uniform band takes the MFMA-accumulator kernels: 64/32/16-wide
    tiles, rows that do not fill a 4-row step or a partition, sources that do not fill a wave; flags,
    a NaN row (flagged in some channels only) and a source outside the unit disc; against the oracle."""
    rng = np.random.default_rng(100 + nchan)
    nrow, nsrc = 1003, 37
    uvw = rng.standard_normal((nrow, 3)) * 3000.0
    lm = rng.standard_normal((nsrc, 2)) * 0.03
    freq = np.linspace(0.9e9, 1.7e9, nchan)
    vis = rng.standard_normal((nrow, nchan, 4)) + 1j * rng.standard_normal((nrow, nchan, 4))
    flags = rng.random((nrow, nchan, 4)) < 0.05
    flags[:, 3, :] = True                       # a channel with no unflagged row stays exactly 0
    ref = oracle.vis_to_im(vis, uvw, lm, freq, flags)
    scale = np.abs(vis).sum(axis=0).max()
    for conv in ("fourier", "casa"):
        dft_mode("auto")
        out = dft.vis_to_im(vis, uvw, lm, freq, flags, convention=conv)
        r = ref if conv == "fourier" else oracle.vis_to_im(vis, uvw, lm, freq, flags, convention="casa")
        assert out.shape == r.shape and maxabs(out, r) <= 1e-11 * scale
        assert (out[:, 3, :] == 0).all()
    # NaN semantics: a non-finite row poisons exactly the channels where it is unflagged; a source
    # outside the unit disc is NaN wherever a channel has an unflagged row
    uvw2, lm2, flags2 = uvw.copy(), lm.copy(), flags.copy()
    uvw2[11] = np.nan
    flags2[11, : nchan // 2, :] = True
    flags2[11, nchan // 2:, :] = False
    lm2[5] = [0.9, 0.8]
    ref2 = oracle.vis_to_im(vis, uvw2, lm2, freq, flags2)
    out2 = dft.vis_to_im(vis, uvw2, lm2, freq, flags2)
    assert_array_equal(np.isnan(out2), np.isnan(ref2))
    ok = ~np.isnan(ref2)
    assert maxabs(out2[ok], ref2[ok]) <= 1e-11 * scale


@pytest.mark.parametrize("nrow, nsrc", [(1, 1), (3, 2), (5, 3), (63, 4), (65, 5), (129, 9)])
def test_mfma_paths_tiny_extents(dft_mode, nrow, nsrc):
    """fewer rows than a wave's 16, fewer sources than one 4-source step: im_to_vis and vis_to_im"""
    rng = np.random.default_rng(1000 * nrow + nsrc)
    nchan = 64
    uvw = rng.standard_normal((nrow, 3)) * 2000.0
    lm = rng.standard_normal((nsrc, 2)) * 0.02
    freq = np.linspace(1.0e9, 1.4e9, nchan)
    img = rng.standard_normal((nsrc, nchan, 4))
    dft_mode("auto")
    out = dft.im_to_vis(img, uvw, lm, freq)
    assert maxabs(out, oracle.im_to_vis(img, uvw, lm, freq)) <= 1e-11 * max(_scale(img), 1.0)
    cimg = img + 1j * rng.standard_normal(img.shape)
    assert maxabs(dft.im_to_vis(cimg, uvw, lm, freq), oracle.im_to_vis(cimg, uvw, lm, freq)) <= 1e-11 * _scale(cimg)
    vis = rng.standard_normal((nrow, nchan, 4)) + 1j * rng.standard_normal((nrow, nchan, 4))
    flags = rng.random((nrow, nchan, 4)) < 0.1
    got = dft.vis_to_im(vis, uvw, lm, freq, flags)
    ref = oracle.vis_to_im(vis, uvw, lm, freq, flags)
    assert maxabs(got, ref) <= 1e-11 * max(np.abs(vis).sum(axis=0).max(), 1.0)


# ---------------------------------------------------------------------------- beams
def _beam_args(g4, lm=None, beam=None, dtype=None):
    args = [g4["beam"] if beam is None else beam, g4["extents"], g4["beam_freq_map"],
            g4["lm"] if lm is None else lm, g4["parangles"], g4["point_errors"],
            g4["antenna_scaling"], g4["freqs"]]
    if dtype is not None:
        args = [a.astype(np.complex64 if np.iscomplexobj(a) else dtype) for a in args]
    return args


def test_freq_grid_interp(g4):
    fd = rime.freq_grid_interp(g4["freqs"], g4["beam_freq_map"])
    assert_array_equal(fd, g4["freq_data"])


def test_beam_cube_dde_golden(g4):
    out = rime.beam_cube_dde(*_beam_args(g4))
    ref = g4["ddes"]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert maxabs(out, ref) < 1e-14
    out0 = rime.beam_cube_dde(*_beam_args(g4, lm=g4["lm"][:2], beam=np.zeros_like(g4["beam"])))
    assert_array_equal(out0, g4["ddes_zero"])
    beam1 = np.ascontiguousarray(g4["beam"][..., 0, :1])
    assert maxabs(rime.beam_cube_dde(*_beam_args(g4, beam=beam1)), g4["ddes_1corr"]) < 1e-14


def test_beam_cube_dde_f32(g4):
    out = rime.beam_cube_dde(*_beam_args(g4, dtype=np.float32))
    assert out.dtype == np.complex64
    assert maxabs(out, g4["ddes_f32"]) < 1e-5


def test_beam_cube_dde_reference_kat(g4):
    """africanus/rime/tests/test_fast_beams.py:43-127: 0.470255+0.4786j."""
    ddes = rime.beam_cube_dde(g4["kat_beam"], np.asarray([[-1.0, 1.0], [-1.0, 1.0]]),
                              np.asarray([0.0, 1.0]), np.asarray([[0.1, 0.1]]), np.zeros((1, 1)),
                              np.zeros((1, 1, 1, 2)), np.ones((1, 1, 2)), np.asarray([0.3]))
    np.testing.assert_array_almost_equal([[[[[0.470255 + 0.4786j]]]]], ddes)
    assert maxabs(ddes, g4["kat_ddes"]) < 1e-15


# ---------------------------------------------------------------------------- chi^2
@pytest.mark.parametrize("shape", [(1237, 13, 4), (4099, 64, 4), (3001, 16, 2), (5003, 8, 1), (2050, 64, 2),
                                   (70000, 64, 4), (3, 5, 4), (1, 1, 1), (777, 3, 2),
                                   (9000, 64, 1), (2100, 256, 1), (1100, 512, 2), (40000, 3, 1), (8200, 64, 3)])
def test_chi2_against_numpy(shape):
    """af_chi2_c128 has no reference counterpart (parity unpinned): checked against numpy -- the row-block and the
    flat-sweep kernels, 1 / 2 / 4 correlations (the lanes of a channel are added up before the atomics)."""
    import torch
    from codex_africanus_amd import sharding
    rng = np.random.default_rng(9)
    m = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    d = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    w = rng.random(shape)
    t = lambda a: torch.from_numpy(a).cuda()
    c = sharding.chi2(t(m), t(d)).cpu().numpy()
    np.testing.assert_allclose(c, (np.abs(d - m) ** 2).sum(axis=(0, 2)), rtol=1e-12)
    cw = sharding.chi2(t(m), t(d), t(w)).cpu().numpy()
    np.testing.assert_allclose(cw, (w * np.abs(d - m) ** 2).sum(axis=(0, 2)), rtol=1e-12)


# ---------------------------------------------------------------------------- full size (C2)
def test_im_to_vis_full_size_c2_properties():
    """BASELINE configs[1] at full size (1e6 rows x 64 chan x 1000 src x 4 corr), device resident:
    sampled rows against the CPU oracle (north-star tolerance 1e-8 absolute), exact linearity under a
    power-of-two scaling, and row-shard invariance (a shard's rows equal the same rows of the full
    call bit for bit: the multi-GPU sharding changes nothing)."""
    import torch
    d = synthetic_inputs(seed=0, nrow=16, nchan=64, nsrc=1000, nant=64)
    rng = np.random.default_rng(1000)
    nrow = 1000000
    uvw = np.empty((nrow, 3))
    uvw[:, 0] = rng.uniform(-4000, 4000, nrow)
    uvw[:, 1] = rng.uniform(-4000, 4000, nrow)
    uvw[:, 2] = rng.uniform(-400, 400, nrow)
    image = real_image(d)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_img, d_uvw, d_lm, d_fr = t(image), t(uvw), t(d["lm"]), t(d["frequency"])
    vis = dft.im_to_vis(d_img, d_uvw, d_lm, d_fr)
    assert tuple(vis.shape) == (nrow, 64, 4) and vis.dtype == torch.complex128
    rows = np.linspace(0, nrow - 1, 48).astype(np.int64)
    ref = oracle.im_to_vis(image, uvw[rows], d["lm"], d["frequency"], omp=True)
    got = vis[torch.from_numpy(rows).cuda()].cpu().numpy()
    assert np.abs(got - ref).max() < 1e-8
    # linearity: x4 commutes with every rounding
    vis4 = dft.im_to_vis(d_img * 4.0, d_uvw, d_lm, d_fr)
    assert torch.equal(vis4, vis * 4.0)
    del vis4
    # row-shard invariance on shard boundaries that are not multiples of the block size
    a, b = 123457, 654321
    part = dft.im_to_vis(d_img, d_uvw[a:b], d_lm, d_fr)
    assert torch.equal(part, vis[a:b])
    # checksum of checksums: per-channel power is finite and positive
    power = (vis.real ** 2 + vis.imag ** 2).sum(dim=(0, 2))
    assert bool(torch.isfinite(power).all()) and bool((power > 0).all())


def test_phase_delay_large_phases_and_fallback():
    """|p| spans the Cody-Waite range (< 2^19 pi/2 ~ 8.2e5 rad) and the library fallback beyond it."""
    rng = np.random.default_rng(17)
    lm = (rng.random((9, 2)) - 0.5) * 0.6
    freq = np.linspace(0.9e9, 1.7e9, 7)
    for span in (3e4, 3e6, 3e8):         # |p| up to ~1e4, ~1e6 (both branches), ~1e8 (fallback)
        uvw = (rng.random((257, 3)) - 0.5) * span
        out = rime.phase_delay(lm, uvw, freq)
        ref = oracle.phase_delay(lm, uvw, freq)
        assert maxabs(out, ref) < 1e-15
    out = rime.phase_delay(lm, np.full((3, 3), np.inf), freq)
    assert np.isnan(out).all()


# ---------------------------------------------------------------------------- vis_to_im
@pytest.fixture(scope="module")
def g6():
    from conftest import load_golden
    return load_golden("g6_vis_to_im.npz")


def _vscale(vis):
    return float(np.abs(vis).sum(axis=0).max())


@pytest.mark.parametrize("mode, rtol", [("exact", 1e-14), ("auto", 1e-11)])
@pytest.mark.parametrize("ncorr", [1, 2, 4])
@pytest.mark.parametrize("conv", ["fourier", "casa"])
def test_vis_to_im_golden(g6, dft_mode, mode, rtol, ncorr, conv):
    dft_mode(mode)
    vis = g6["vis%d" % ncorr]
    out = dft.vis_to_im(vis, g6["uvw"], g6["lm"], g6["frequency"], g6["flags%d" % ncorr], convention=conv)
    ref = g6["im%d_%s" % (ncorr, conv)]
    assert out.shape == ref.shape and out.dtype == ref.dtype == np.float64
    assert maxabs(out, ref) <= rtol * _vscale(vis)


@pytest.mark.parametrize("mode, rtol", [("exact", 1e-14), ("auto", 1e-11)])
def test_vis_to_im_other_cases(g6, dft_mode, mode, rtol):
    dft_mode(mode)
    sc = _vscale(g6["vis4"])
    f = dft.vis_to_im
    assert maxabs(f(g6["vis4"], g6["uvw"], g6["lm"], g6["frequency_nonuniform"], g6["flags4"]),
                  g6["im4_nonuniform"]) <= rtol * sc
    assert maxabs(f(g6["vis4"].real.copy(), g6["uvw"], g6["lm"], g6["frequency"], g6["flags4"]),
                  g6["im4_realvis"]) <= rtol * sc
    assert maxabs(f(g6["vis4"], g6["uvw"], g6["lm"], g6["frequency"], np.zeros_like(g6["flags4"])),
                  g6["im4_noflags"]) <= rtol * sc
    # 70 channels x 300 rows: several tiles and row partitions
    assert maxabs(f(g6["vis70"], g6["uvw300"], g6["lm"], g6["frequency70"], g6["flags70"]),
                  g6["im70"]) <= rtol * _vscale(g6["vis70"])
    out32 = f(g6["vis4"], g6["uvw"], g6["lm"], g6["frequency"], g6["flags4"], dtype=np.float32)
    assert out32.dtype == np.float32 and maxabs(out32, g6["im4_f32"]) < 2e-5 * np.abs(g6["im4_f32"]).max()
    nan = f(g6["vis4"], g6["uvw"], g6["lm_nan"], g6["frequency"], g6["flags4"])
    assert_array_equal(np.isnan(nan), np.isnan(g6["im4_nan"]))


def test_vis_to_im_flagged_kat(dft_mode):
    """africanus/dft/tests/test_dft.py:180-215: everything flagged but row 0 at the origin."""
    dft_mode("auto")
    rng = np.random.default_rng(123)
    nsource, nrow, nchan, ncorr = 21, 31, 3, 4
    uvw = 100 * rng.random((nrow, 3))
    uvw[0] = 0.0
    lm = 0.01 * rng.standard_normal((nsource, 2))
    vis = rng.standard_normal((nrow, nchan, ncorr)) + 1j * rng.standard_normal((nrow, nchan, ncorr))
    vis[0] = 1.0
    flags = np.ones((nrow, nchan, ncorr), dtype=bool)
    flags[0] = False
    frequency = np.ones(nchan) * 2.99792458e8
    im = dft.vis_to_im(vis, uvw, lm, frequency, flags)
    np.testing.assert_array_almost_equal(im, np.ones((nsource, nchan, ncorr)), decimal=13)
    # a fully flagged channel stays exactly zero, even for a NaN source
    flags[0, 1] = True
    lm[3] = [0.9, 0.9]
    im = dft.vis_to_im(vis, uvw, lm, frequency, flags)
    ref = oracle.vis_to_im(vis, uvw, lm, frequency, flags)
    assert (im[:, 1] == 0).all() and np.isnan(im[3, 0]).all()
    assert_array_equal(np.isnan(im), np.isnan(ref))


def test_vis_to_im_adjointness_and_shard_sum(dft_mode):
    """<y, R x> = <R^H y, x> (test_dft.py:136-177) at a larger shape, and the image of all rows is
    the sum of the images of row shards (what the multi-GPU all-reduce relies on)."""
    dft_mode("auto")
    d = synthetic_inputs(seed=31, nrow=6000, nchan=64, nsrc=130)
    rng = d["rng"]
    x = rng.standard_normal((130, 64, 4))
    y = rng.standard_normal((6000, 64, 4)) + 1j * rng.standard_normal((6000, 64, 4))
    flag = np.zeros(y.shape, dtype=bool)
    Rx = dft.im_to_vis(x, d["uvw"], d["lm"], d["frequency"])
    RHy = dft.vis_to_im(y, d["uvw"], d["lm"], d["frequency"], flag, convention="fourier")
    lhs, rhs = np.vdot(y, Rx).real, np.vdot(RHy, x)
    assert abs(lhs - rhs) < 1e-10 * abs(lhs)
    parts = sum(dft.vis_to_im(y[a:b], d["uvw"][a:b], d["lm"], d["frequency"], flag[a:b])
                for a, b in ((0, 2500), (2500, 6000)))
    assert maxabs(parts, RHy) < 1e-11 * np.abs(RHy).max()
    ref = oracle.vis_to_im(y, d["uvw"], d["lm"][:8], d["frequency"], flag, omp=True)
    assert maxabs(RHy[:8], ref) < 1e-8
