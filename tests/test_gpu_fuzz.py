"""Seeded random-shape sweeps of the HIP entry points against the oracle: extents that straddle every tile /
step / partition boundary, random flags and zero pixels, both conventions, every phasor mode."""
import numpy as np
import pytest

import oracle
from codex_africanus_amd import dft, rime

pytestmark = pytest.mark.gpu

MODES = ("auto", "valu", "exact")


@pytest.fixture(autouse=True)
def _restore_mode():
    m = dft.get_mode()
    yield
    dft.set_mode(m)


def _freq(rng, nchan, uniform):
    f = np.linspace(0.9e9, 0.9e9 + 6.5e6 * max(nchan - 1, 1), nchan)
    if not uniform and nchan > 2:
        f = f * (1 + 1e-3 * rng.random(nchan))
    return f


def _recurrence_tol(uvw, lm, freq):
    """Relative tolerance of the recurrence modes: 1e-11, or the rounding of the phase argument itself when the
    longest path difference spans thousands of turns (the kernels form q * (nu / c) in turns, the reference
    (C * q) * nu in radians: each product rounds at ~1e-16 of a phase of 2 pi * turns; seed 6469 of
    tools/stress_random.py: 25 km baselines, 6650 turns, 1.2e-11)."""
    n = np.sqrt(np.maximum(0.0, 1.0 - (lm ** 2).sum(axis=1))) - 1.0
    lmn = np.abs(np.c_[lm, n])
    turns = float((np.abs(uvw) @ lmn.T).max() * np.max(freq) / 299792458.0)
    return max(1e-11, 1e-15 * 2 * np.pi * turns)


@pytest.mark.parametrize("seed", range(24))
def test_im_to_vis_random_shapes(seed):
    rng = np.random.default_rng(seed)
    nrow, nsrc = int(rng.integers(1, 400)), int(rng.integers(1, 40))
    nchan = int(rng.choice([1, 2, 7, 13, 14, 15, 16, 17, 31, 32, 33, 47, 63, 64, 65, 80, 97, 128, 129]))
    ncorr = int(rng.choice([1, 2, 4, 4, 4]))
    cplx = bool(rng.integers(0, 2))
    uniform = bool(rng.integers(0, 4))
    mode = MODES[seed % 3]
    conv = ("fourier", "casa")[seed % 2]
    uvw = rng.standard_normal((nrow, 3)) * rng.choice([10.0, 1000.0, 8000.0])
    lm = rng.standard_normal((nsrc, 2)) * 0.03
    img = rng.standard_normal((nsrc, nchan, ncorr))
    if cplx:
        img = img + 1j * rng.standard_normal(img.shape)
    img[rng.random(img.shape) < 0.1] = 0.0
    freq = _freq(rng, nchan, uniform)
    dft.set_mode(mode)
    out = dft.im_to_vis(img, uvw, lm, freq, convention=conv)
    ref = oracle.im_to_vis(img, uvw, lm, freq, convention=conv)
    scale = max(float(np.abs(img).sum(axis=0).max()), 1e-300)
    tol = 1e-14 if (mode == "exact" or (not uniform and nchan > 2)) else _recurrence_tol(uvw, lm, freq)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= tol * scale, (nrow, nsrc, nchan, ncorr, cplx, uniform, mode)


@pytest.mark.parametrize("ncorr, nchan", [(1, 13), (1, 31), (1, 32), (1, 33), (1, 52), (1, 53), (1, 64), (1, 104),
                                          (1, 130), (2, 15), (2, 16), (2, 17), (2, 26), (2, 27), (2, 64), (2, 78)])
@pytest.mark.parametrize("mode", ["auto", "exact"])
def test_im_to_vis_wide_tiles_for_few_correlations(ncorr, nchan, mode):
    """1 and 2 correlations run channel tiles of up to 52 / 26 channels (choose_ct): every candidate width, tiles that
    end exactly at / one channel past the band, zero pixels, both phasor modes and a non-uniform band."""
    rng = np.random.default_rng(1000 * ncorr + nchan)
    nrow, nsrc = 257, 19
    uvw = rng.standard_normal((nrow, 3)) * 3000.0
    lm = rng.standard_normal((nsrc, 2)) * 0.03
    img = rng.standard_normal((nsrc, nchan, ncorr))
    img[rng.random(img.shape) < 0.1] = 0.0
    dft.set_mode(mode)
    try:
        for uniform in (True, False):
            freq = _freq(rng, nchan, uniform)
            out = dft.im_to_vis(img, uvw, lm, freq)
            ref = oracle.im_to_vis(img, uvw, lm, freq)
            scale = float(np.abs(img).sum(axis=0).max())
            tol = 1e-14 if (mode == "exact" or not uniform) else 1e-11
            assert np.abs(out - ref).max() <= tol * scale, (uniform,)
    finally:
        dft.set_mode("auto")


@pytest.mark.parametrize("seed", range(24))
def test_vis_to_im_random_shapes(seed):
    rng = np.random.default_rng(100 + seed)
    nrow, nsrc = int(rng.integers(1, 700)), int(rng.integers(1, 90))
    nchan = int(rng.choice([1, 3, 13, 14, 16, 17, 32, 33, 48, 64, 65, 96, 130]))
    ncorr = int(rng.choice([1, 2, 4, 4, 4]))
    uniform = bool(rng.integers(0, 4))
    mode = ("auto", "exact", "auto")[seed % 3]
    conv = ("fourier", "casa")[seed % 2]
    uvw = rng.standard_normal((nrow, 3)) * rng.choice([10.0, 1000.0, 8000.0])
    lm = rng.standard_normal((nsrc, 2)) * 0.03
    vis = rng.standard_normal((nrow, nchan, ncorr)) + 1j * rng.standard_normal((nrow, nchan, ncorr))
    flags = rng.random((nrow, nchan, ncorr)) < rng.choice([0.0, 0.05, 0.6])
    freq = _freq(rng, nchan, uniform)
    dft.set_mode(mode)
    out = dft.vis_to_im(vis, uvw, lm, freq, flags, convention=conv)
    ref = oracle.vis_to_im(vis, uvw, lm, freq, flags, convention=conv)
    scale = max(float(np.abs(vis).sum(axis=0).max()), 1.0)
    tol = 1e-14 if (mode == "exact" or (not uniform and nchan > 2)) else _recurrence_tol(uvw, lm, freq)
    assert np.abs(out - ref).max() <= tol * scale, (nrow, nsrc, nchan, ncorr, uniform, mode)


@pytest.mark.parametrize("ncorr, nchan", [(1, 16), (1, 31), (1, 32), (1, 33), (1, 63), (1, 64), (1, 65), (1, 130),
                                          (2, 15), (2, 16), (2, 31), (2, 32), (2, 33), (2, 64), (2, 70)])
@pytest.mark.parametrize("mode", ["auto", "exact"])
def test_vis_to_im_wide_tiles_for_few_correlations(ncorr, nchan, mode):
    """1 and 2 correlations run channel tiles of up to 64 / 32 channels (choose_ct_v): every candidate width, tiles
    ending at / one past the band, flags, both phasor modes and a non-uniform band."""
    rng = np.random.default_rng(2000 * ncorr + nchan)
    nrow, nsrc = 333, 21
    uvw = rng.standard_normal((nrow, 3)) * 3000.0
    lm = rng.standard_normal((nsrc, 2)) * 0.03
    vis = rng.standard_normal((nrow, nchan, ncorr)) + 1j * rng.standard_normal((nrow, nchan, ncorr))
    flags = rng.random((nrow, nchan, ncorr)) < 0.1
    dft.set_mode(mode)
    try:
        for uniform in (True, False):
            freq = _freq(rng, nchan, uniform)
            out = dft.vis_to_im(vis, uvw, lm, freq, flags)
            ref = oracle.vis_to_im(vis, uvw, lm, freq, flags)
            scale = float(np.abs(vis).sum(axis=0).max())
            tol = 1e-14 if (mode == "exact" or not uniform) else 1e-11
            assert np.abs(out - ref).max() <= tol * scale, (uniform,)
    finally:
        dft.set_mode("auto")


@pytest.mark.parametrize("seed", range(16))
def test_wsclean_predict_random_shapes(seed):
    rng = np.random.default_rng(200 + seed)
    nrow, nsrc = int(rng.integers(1, 300)), int(rng.integers(1, 60))
    nchan = int(rng.choice([1, 2, 8, 9, 16, 24, 25, 40, 41, 64, 81]))
    ncoeff = int(rng.integers(1, 5))
    uniform = bool(rng.integers(0, 4))
    mode = ("auto", "exact")[seed % 2]
    isg = rng.random(nsrc) < rng.choice([0.0, 0.5, 1.0])
    st = np.where(isg, "GAUSSIAN", "POINT")
    uvw = rng.standard_normal((nrow, 3)) * 1500.0
    lm = rng.standard_normal((nsrc, 2)) * 0.02
    flux = rng.uniform(0.1, 2.0, nsrc)
    coeffs = rng.standard_normal((nsrc, ncoeff)) * (0.5 ** np.arange(1, ncoeff + 1))
    log_poly = rng.random(nsrc) < 0.5
    gshape = np.stack([rng.uniform(0, 3e-4, nsrc), rng.uniform(0, 2e-4, nsrc), rng.uniform(0, np.pi, nsrc)], axis=1)
    freq = _freq(rng, nchan, uniform)
    ref_freq = np.full(nsrc, float(freq[nchan // 2]))
    args = (uvw, lm, st, flux, coeffs, log_poly, ref_freq, gshape, freq)
    dft.set_mode(mode)
    out = rime.wsclean_predict(*args)
    ref = oracle.wsclean_predict(*args)
    scale = max(float(np.abs(oracle.spectra(flux, coeffs, log_poly, ref_freq, freq)).sum(axis=0).max()), 1.0)
    tol = 1e-13 if (mode == "exact" or (not uniform and nchan > 2)) else 1e-11
    assert np.abs(out - ref).max() <= tol * scale, (nrow, nsrc, nchan, ncoeff, uniform, mode)


@pytest.mark.parametrize("seed", range(12))
def test_predict_vis_random_shapes_bit_exact(seed):
    rng = np.random.default_rng(300 + seed)
    nsrc, ntime, nant, nchan = int(rng.integers(1, 6)), int(rng.integers(1, 5)), int(rng.integers(2, 9)), int(rng.integers(1, 20))
    corr = [(1,), (2,), (2, 2)][seed % 3]
    a1, a2 = np.triu_indices(nant, 1)
    nbl = a1.shape[0]
    time_index = np.repeat(np.arange(ntime), nbl) + int(rng.integers(0, 50))
    ant1, ant2 = np.tile(a1, ntime), np.tile(a2, ntime)
    nrow = time_index.shape[0]
    rc = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    have = rng.integers(0, 2, 4).astype(bool)
    dde = rc(nsrc, ntime, nant, nchan, *corr) if have[0] else None
    coh = rc(nsrc, nrow, nchan, *corr) if (have[1] or not have[0]) else None
    die = rc(ntime, nant, nchan, *corr) if have[2] else None
    bv = rc(nrow, nchan, *corr) if have[3] else None
    idx_t = (np.int32, np.int64)[seed % 2]
    args = (time_index.astype(idx_t), ant1.astype(idx_t), ant2.astype(idx_t), dde, coh, dde, die, bv, die)
    out = rime.predict_vis(*args)
    ref = oracle.predict_vis(*args)
    np.testing.assert_array_equal(out, ref)


def test_piecewise_uniform_bands_take_the_per_tile_kernels():
    """two sub-bands with different channel spacings: every VALU tile (13 / 16 channels) is uniform but
    the MFMA tiles are not -- the device-side flags must route the call to the per-tile recurrence kernels"""
    rng = np.random.default_rng(7)
    nrow, nsrc = 150, 21
    uvw = rng.standard_normal((nrow, 3)) * 3000.0
    lm = rng.standard_normal((nsrc, 2)) * 0.03
    # im_to_vis tiles are 13 channels wide: 2 x 13 channels
    f13 = np.concatenate([1.0e9 + 1e6 * np.arange(13), 1.2e9 + 3e6 * np.arange(13)])
    img = rng.standard_normal((nsrc, 26, 4))
    dft.set_mode("auto")
    out = dft.im_to_vis(img, uvw, lm, f13)
    assert np.abs(out - oracle.im_to_vis(img, uvw, lm, f13)).max() <= 1e-11 * np.abs(img).sum(axis=0).max()
    # vis_to_im VALU tiles are 16 wide: 2 x 16 channels
    f16 = np.concatenate([1.0e9 + 1e6 * np.arange(16), 1.2e9 + 3e6 * np.arange(16)])
    vis = rng.standard_normal((nrow, 32, 4)) + 1j * rng.standard_normal((nrow, 32, 4))
    flags = rng.random((nrow, 32, 4)) < 0.05
    got = dft.vis_to_im(vis, uvw, lm, f16, flags)
    ref = oracle.vis_to_im(vis, uvw, lm, f16, flags)
    assert np.abs(got - ref).max() <= 1e-11 * np.abs(vis).sum(axis=0).max()


@pytest.mark.parametrize("sub,nsub", [(8, 2), (16, 2), (13, 2), (26, 2), (11, 2), (8, 3), (13, 5), (64, 2), (32, 3)])
@pytest.mark.parametrize("cplx", [False, True])
def test_equal_width_subbands_with_a_gap(sub, nsub, cplx):
    """concatenated spectral windows: SAME channel width, a gap between the windows.  Every VALU tile is
    uniform and all tiles share one spacing, but an MFMA tile (64 channels, evaluated as nu[c0] + j dnu) that
    straddles a gap is not an arithmetic progression: the device-side flags must keep such bands off
    dft_mfma_kernel (af_im_to_vis.hip dft_prep_freq).  Gaps on the VALU tile boundaries 8 / 11 / 13 /
    16, inside a tile, and on a 64-channel MFMA tile boundary (harmless: every MFMA tile has its own start)."""
    rng = np.random.default_rng(1000 * sub + nsub)
    nrow, nsrc = 130, 19
    uvw = rng.standard_normal((nrow, 3)) * 3000.0
    lm = rng.standard_normal((nsrc, 2)) * 0.03
    freq = np.concatenate([1.0e9 + 0.2e9 * k + 1e6 * np.arange(sub) for k in range(nsub)])
    img = rng.standard_normal((nsrc, freq.size, 4))
    if cplx:
        img = img + 1j * rng.standard_normal(img.shape)
    dft.set_mode("auto")
    out = dft.im_to_vis(img, uvw, lm, freq)
    ref = oracle.im_to_vis(img, uvw, lm, freq)
    assert np.abs(out - ref).max() <= 1e-11 * np.abs(img).sum(axis=0).max()
    # the adjoint keeps its own per-MFMA-tile test (flags[3]); same bands
    got = dft.vis_to_im(ref, uvw, lm, freq, np.zeros(ref.shape, dtype=bool))
    want = oracle.vis_to_im(ref, uvw, lm, freq, np.zeros(ref.shape, dtype=bool))
    assert np.abs(got - want).max() <= 1e-11 * np.abs(ref).sum(axis=0).max()


@pytest.mark.parametrize("seed", range(24))
def test_degridder_gridder_random_shapes(seed):
    """kernel widths 1..11, oversampling 1..63, odd grid sizes, rows below and above the uv-tile-order threshold, band
    maps, every Stokes policy in turn, taps on and off the grid; both directions against the oracle"""
    from codex_africanus_amd.gridding.perleypolyhedron import kernels
    from codex_africanus_amd.gridding.perleypolyhedron.degridder import degridder, STOKES_TO_CORR
    from codex_africanus_amd.gridding.perleypolyhedron.gridder import gridder, CORR_TO_STOKES
    rng = np.random.default_rng(400 + seed)
    W, OS = int(rng.choice([1, 3, 5, 7, 7, 7, 9, 11])), int(rng.choice([1, 3, 9, 63]))
    npix, nrow, nchan = int(rng.choice([16, 33, 64, 100])), int(rng.choice([1, 7, 300, 5000])), int(rng.integers(1, 6))
    nband = int(rng.integers(1, 3))
    chanmap = rng.integers(0, nband, nchan)
    chanmap[0] = nband - 1
    wl = 299792458.0 / np.linspace(1.0e9, 1.3e9, nchan)
    cell = 5.0
    uvw = rng.uniform(-1, 1, (nrow, 3)) * rng.choice([0.3, 0.6]) / np.deg2rad(cell / 3600.0) * wl.min()
    k = kernels.hanningsinc(W, oversample=OS) if W > 1 else np.full(OS * 3, 1.0 / (OS * 3))
    ppol = ("None", "phase_rotate")[seed % 2]
    packed = seed % 3 != 0
    kk = kernels.pack_kernel(k, W, OS) if packed else k
    grid = rng.standard_normal((nband, npix, npix)) + 1j * rng.standard_normal((nband, npix, npix))
    a = (uvw, grid, wl, chanmap, cell, (0.1, 0.2), (0.11, 0.19), kk, W, OS, "None", ppol, sorted(STOKES_TO_CORR)[seed % 16],
         "conv_1d_axisymmetric_packed_gather" if packed else "conv_1d_axisymmetric_unpacked_gather")
    out, ref = degridder(*a), oracle.degridder(*a)
    assert np.abs(out - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1e-300), (W, OS, npix, nrow)
    spol = sorted(CORR_TO_STOKES)[seed % 15]
    ncorr = len(CORR_TO_STOKES[spol])
    vis = rng.standard_normal((nrow, nchan, ncorr)) + 1j * rng.standard_normal((nrow, nchan, ncorr))
    b = (uvw, vis, wl, chanmap, npix, cell, (0.1, 0.2), (0.11, 0.19), kk, W, OS, "None", ppol, spol,
         "conv_1d_axisymmetric_packed_scatter" if packed else "conv_1d_axisymmetric_unpacked_scatter")
    out, ref = gridder(*b, do_normalize=bool(seed % 2)), oracle.gridder(*b, do_normalize=bool(seed % 2))
    assert np.abs(out - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1e-300), (W, OS, npix, nrow)
