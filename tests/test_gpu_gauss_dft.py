"""
Gaussian (and point) sources without direction-dependent terms: af_gauss_predict_c128 (csrc/af_gauss_dft.hip) behind
``rime.fused_predict_vis(..., gauss_shape=...)`` with no beam.  Reference chain: phase_delay x gaussian shape x
brightness summed over the sources -- einsum("srf,srf,sfij->srfij") + predict_vis(source_coh)
(africanus/rime/examples/predict.py:107-134, africanus/model/shape/gaussian_shape.py:21-62), restated by the CPU oracle.
Tolerance 1e-9 of the per-visibility sum of |brightness| (polynomial phasors, recurrences along the channel tile).
"""
import ctypes

import numpy as np
import pytest

import oracle
from codex_africanus_amd import rime, _lib
from test_gpu_fused import _problem, _scale

pytestmark = pytest.mark.gpu


def _shapes(nsrc, seed=6):
    rng = np.random.default_rng(seed)
    sp = np.stack([rng.uniform(0, 3e-4, nsrc), rng.uniform(0, 2e-4, nsrc), rng.uniform(0, np.pi, nsrc)], axis=1)
    sp[::4] = 0.0                     # point sources in between
    return sp


def _chain(d, sp, convention="fourier"):
    phase = oracle.phase_delay(d["lm"], d["uvw"], d["frequency"], convention)
    shape = oracle.gaussian_shape(d["uvw"], d["frequency"], sp)
    coh = np.einsum("srf,srf,sfij->srfij", phase, shape, d["X"])
    return oracle.predict_vis(d["time_index"], d["ant1"], d["ant2"], None, coh, None, None, None, None)


# channel counts on and off the 8-channel tile, row counts on and off the 256-row block
@pytest.mark.parametrize("nrow, nchan", [(1, 1), (255, 7), (256, 8), (700, 9), (1000, 16), (513, 29), (300, 64)])
def test_uniform_bands(nrow, nchan):
    d = _problem(31, nrow, nchan, 19, 7, with_beam=False)
    sp = _shapes(19)
    out = rime.fused_predict_vis(d["time_index"], d["ant1"], d["ant2"], d["lm"], d["uvw"], d["frequency"], d["X"],
                                 gauss_shape=sp)
    ref = _chain(d, sp)
    assert out.shape == ref.shape and np.abs(out - ref).max() <= 1e-9 * _scale(d)


# bands the MFMA-accumulator form owns (>= 14 channels, one spacing): every tile plan of af_im_to_vis_mfma.hip -- one short
# 64-channel tile, 16- and 32-channel tails, several full tiles -- and source counts on and off the 4-source step
@pytest.mark.parametrize("nrow, nchan, nsrc", [(130, 14, 1), (64, 33, 4), (257, 48, 5), (100, 80, 19), (65, 100, 7),
                                               (70, 130, 8), (1, 96, 3)])
def test_mfma_form_tile_plans(nrow, nchan, nsrc):
    d = _problem(35, nrow, nchan, nsrc, 7, with_beam=False)
    sp = _shapes(nsrc, 8)
    for conv in ("fourier", "casa"):
        out = rime.fused_predict_vis(d["time_index"], d["ant1"], d["ant2"], d["lm"], d["uvw"], d["frequency"], d["X"],
                                     gauss_shape=sp, convention=conv)
        ref = _chain(d, sp, conv)
        assert out.shape == ref.shape and np.abs(out - ref).max() <= 1e-9 * _scale(d)


def test_mfma_form_falling_band_with_an_envelope_that_underflows():
    """On a falling band the envelope's channel ratio r_0 exceeds 1; where exp(-a nu_0^2) is already 0 it may overflow:
    0 x inf must not appear."""
    d = _problem(36, 200, 32, 6, 6, with_beam=False)
    sp = _shapes(6, 3)
    sp[1] = [5e-2, 4e-2, 0.7]
    sp[2] = [2e-3, 1e-3, 2.0]
    freq = d["frequency"][::-1].copy()
    dd = dict(d, frequency=freq)
    out = rime.fused_predict_vis(dd["time_index"], dd["ant1"], dd["ant2"], dd["lm"], dd["uvw"], freq, dd["X"], gauss_shape=sp)
    assert np.isfinite(out).all()
    assert np.abs(out - _chain(dd, sp)).max() <= 1e-9 * _scale(dd)
    # the same band in 8-channel pieces runs the lane = row kernels (fewer than 14 channels): same visibilities
    for c0 in range(0, 32, 8):
        part = rime.fused_predict_vis(dd["time_index"], dd["ant1"], dd["ant2"], dd["lm"], dd["uvw"], freq[c0:c0 + 8],
                                      dd["X"][:, c0:c0 + 8], gauss_shape=sp)
        assert np.isfinite(part).all()
        assert np.abs(part - out[:, c0:c0 + 8]).max() <= 1e-9 * _scale(dd)


@pytest.mark.parametrize("seed", range(10))
def test_random_shapes(seed):
    """random rows / channels / sources, rising and falling bands, both conventions, all-point and all-extended lists"""
    rng = np.random.default_rng(4000 + seed)
    nrow, nchan, nsrc = int(rng.integers(1, 400)), int(rng.integers(1, 150)), int(rng.integers(1, 40))
    d = _problem(40 + seed, nrow, nchan, nsrc, int(rng.integers(3, 9)), with_beam=False)
    sp = _shapes(nsrc, seed)
    if seed % 5 == 3:
        sp[:] = 0.0
    if seed % 5 == 4:
        sp[::4] = [1e-4, 5e-5, 1.0]
    freq = d["frequency"][::-1].copy() if seed % 2 else d["frequency"]
    dd = dict(d, frequency=freq)
    conv = ("fourier", "casa")[(seed // 2) % 2]
    out = rime.fused_predict_vis(dd["time_index"], dd["ant1"], dd["ant2"], dd["lm"], dd["uvw"], freq, dd["X"], gauss_shape=sp,
                                 convention=conv)
    assert np.abs(out - _chain(dd, sp, conv)).max() <= 1e-9 * _scale(dd), (nrow, nchan, nsrc)


def test_non_uniform_band_descending_band_and_casa():
    d = _problem(32, 400, 21, 15, 6, with_beam=False)
    sp = _shapes(15, 2)
    rng = np.random.default_rng(3)
    for freq in (np.sort(rng.uniform(0.9e9, 1.7e9, 21)), d["frequency"][::-1].copy()):
        dd = dict(d, frequency=freq)
        for conv in ("fourier", "casa"):
            out = rime.fused_predict_vis(dd["time_index"], dd["ant1"], dd["ant2"], dd["lm"], dd["uvw"], freq, dd["X"],
                                         gauss_shape=sp, convention=conv)
            assert np.abs(out - _chain(dd, sp, conv)).max() <= 1e-9 * _scale(dd)


def test_envelope_that_underflows_and_a_source_outside_the_disc():
    d = _problem(33, 300, 16, 9, 6, with_beam=False)
    sp = _shapes(9, 4)
    sp[1] = [5e-2, 4e-2, 0.7]          # 3 degrees across: the envelope is 0 on every baseline beyond a few metres
    d["lm"][2] = [0.9, 0.8]            # l^2 + m^2 > 1: phase_delay clamps n
    out = rime.fused_predict_vis(d["time_index"], d["ant1"], d["ant2"], d["lm"], d["uvw"], d["frequency"], d["X"],
                                 gauss_shape=sp)
    assert np.isfinite(out).all()
    assert np.abs(out - _chain(d, sp)).max() <= 1e-9 * _scale(d)


def test_sky_model_inputs_and_flat_spectrum():
    d = _problem(34, 500, 12, 11, 6, with_beam=False)
    sp = _shapes(11, 5)
    rng = np.random.default_rng(9)
    stokes = np.stack([rng.lognormal(0, 1, 11)] + [0.1 * rng.standard_normal(11) for _ in range(3)], axis=1)
    spi = rng.uniform(-1.0, 0.2, (11, 2, 4))
    rf = rng.uniform(0.9e9, 1.5e9, 11)
    out = rime.fused_predict_vis(d["time_index"], d["ant1"], d["ant2"], d["lm"], d["uvw"], d["frequency"], None,
                                 gauss_shape=sp, stokes=stokes, spi=spi, ref_freq=rf)
    st = oracle.spectral_model(stokes, spi, rf, d["frequency"], base=0)
    I, Q, U, V = (st[..., k] for k in range(4))           # noqa: E741
    d2 = dict(d, X=np.stack([I + Q, U + 1j * V, U - 1j * V, I - Q], axis=-1).reshape(11, 12, 2, 2))
    assert np.abs(out - _chain(d2, sp)).max() <= 1e-9 * _scale(d2)
    flat = rime.fused_predict_vis(d["time_index"], d["ant1"], d["ant2"], d["lm"], d["uvw"], d["frequency"], d["X"][:, 0],
                                  gauss_shape=sp)
    d3 = dict(d, X=np.broadcast_to(d["X"][:, :1], d["X"].shape))
    assert np.abs(flat - _chain(d3, sp)).max() <= 1e-9 * _scale(d3)


def test_through_the_c_abi_and_against_the_beam_route():
    """device pointers straight into af_gauss_predict_c128; the same call through the fused beam kernel with an
    identity cube (the route such calls took until round 4) agrees and is slower"""
    import torch
    dev = torch.device("cuda:0")
    d = _problem(35, 40000, 64, 200, 64, with_beam=False)
    sp = _shapes(200, 7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    lm, uvw, fr, X, gs = t(d["lm"]), t(d["uvw"]), t(d["frequency"]), t(d["X"]), t(sp)
    out = torch.empty((40000, 64, 2, 2), dtype=torch.complex128, device=dev)
    lib = _lib.load()
    nb = int(lib.af_gauss_predict_workspace_bytes(200, 64))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    call = lambda: _lib.call("af_gauss_predict_c128", P(lm), P(uvw), P(fr), P(X), P(gs), 200, 40000, 64, -1, P(out), P(ws), nb,
                             stream)
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms_direct = e0.elapsed_time(e1) / 3
    rows = np.linspace(0, 39999, 64).astype(np.int64)
    sub = dict(d, uvw=d["uvw"][rows], time_index=d["time_index"][rows], ant1=d["ant1"][rows], ant2=d["ant2"][rows])
    assert np.abs(out.cpu().numpy()[rows] - _chain(sub, sp)).max() <= 1e-9 * _scale(d)
    # the identity-cube route through the beam kernel
    ident = np.zeros((2, 2, 2, 2, 2), dtype=np.complex128)
    ident[..., 0, 0] = ident[..., 1, 1] = 1.0
    ntime = d["ntime"]
    args = (t(d["time_index"]), t(d["ant1"]), t(d["ant2"]), lm, uvw, fr, X, t(ident), t(np.array([[-2.0, 2.0], [-2.0, 2.0]])),
            t(np.array([0.4e9, 3.5e9])), t(np.zeros((ntime, 64))), t(np.zeros((ntime, 64, 64, 2))), t(np.ones((64, 64, 2))))
    plan = rime.fused_plan(d["time_index"], d["ant1"], d["ant2"], 64)
    other = rime.fused_predict_vis(*args, gauss_shape=gs, plan=plan)
    torch.cuda.synchronize()
    e0.record()
    other = rime.fused_predict_vis(*args, gauss_shape=gs, plan=plan)
    e1.record()
    torch.cuda.synchronize()
    ms_beam = e0.elapsed_time(e1)
    assert float((other - out).abs().max()) <= 1e-9 * _scale(d)
    print("gauss direct %.3f ms, identity-beam route %.3f ms" % (ms_direct, ms_beam))
    assert ms_direct < ms_beam


@pytest.mark.parametrize("nrow, nchan, uniform, weighted", [(300, 64, True, False), (1000, 80, True, True), (257, 9, True, True),
                                                            (400, 21, False, False), (64, 33, True, False)])
def test_predict_and_chi2_in_one_call(nrow, nchan, uniform, weighted):
    """af_gauss_predict_chi2_c128: visibilities bit-equal to af_gauss_predict_c128, chi^2 = numpy's sum -- in the MFMA
    kernels' epilogue (uniform bands of >= 14 channels, every tile plan) and through the separate pass (short and
    non-uniform bands: the fallback is taken on the device)."""
    import torch
    dev = torch.device("cuda:0")
    nsrc = 13
    d = _problem(50 + nrow, nrow, nchan, nsrc, 7, with_beam=False)
    sp = _shapes(nsrc, 11)
    freq = d["frequency"] if uniform else np.sort(np.random.default_rng(1).uniform(0.9e9, 1.7e9, nchan))
    rng = np.random.default_rng(2)
    data = rng.standard_normal((nrow, nchan, 2, 2)) + 1j * rng.standard_normal((nrow, nchan, 2, 2))
    wgt = rng.random((nrow, nchan, 2, 2)) if weighted else None
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    lm, uvw, fr, X, gs, dd, ww = t(d["lm"]), t(d["uvw"]), t(freq), t(d["X"]), t(sp), t(data), t(wgt)
    lib = _lib.load()
    nb = int(lib.af_gauss_predict_workspace_bytes(nsrc, nchan))
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    P = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    plain = torch.empty((nrow, nchan, 2, 2), dtype=torch.complex128, device=dev)
    both = torch.empty_like(plain)
    chi2 = torch.full((nchan,), -1.0, dtype=torch.float64, device=dev)
    _lib.call("af_gauss_predict_c128", P(lm), P(uvw), P(fr), P(X), P(gs), nsrc, nrow, nchan, -1, P(plain), P(ws), nb, stream)
    _lib.call("af_gauss_predict_chi2_c128", P(lm), P(uvw), P(fr), P(X), P(gs), nsrc, nrow, nchan, -1, P(both), P(dd), P(ww),
              P(chi2), P(ws), nb, stream)
    torch.cuda.synchronize()
    assert torch.equal(plain, both)
    v = both.cpu().numpy()
    want = ((np.abs(data - v) ** 2) * (1.0 if wgt is None else wgt)).sum(axis=(0, 2, 3))
    np.testing.assert_allclose(chi2.cpu().numpy(), want, rtol=1e-12)
    dd2 = dict(d, frequency=freq)
    assert np.abs(v - _chain(dd2, sp)).max() <= 1e-9 * _scale(dd2)


@pytest.mark.parametrize("band", ["rising40", "falling33", "rising80"])
@pytest.mark.parametrize("conv", ["fourier", "casa"])
def test_mfma_form_against_the_reference_itself(band, conv):
    """G15: the REFERENCE's own phase_delay x gaussian x brightness -> predict_vis on bands of 40 / 33 (falling) / 80 uniformly
    spaced channels, i.e. on the MFMA-accumulator form (one short tile; a falling band; a full tile and a 16-channel tail),
    generated by tests/golden/make_golden_gauss.py under the real numba.  1e-9 of the summed |brightness|."""
    from conftest import load_golden
    g = load_golden("g15_gauss.npz")
    freq, X = g["frequency_" + band], g["brightness_" + band]
    out = rime.fused_predict_vis(g["time_index"], g["antenna1"], g["antenna2"], g["lm"], g["uvw"], freq, X,
                                 gauss_shape=g["shape_params"], convention=conv)
    ref = g["vis_%s_%s" % (band, conv)]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert np.abs(out - ref).max() <= 1e-9 * float(g["scale_" + band])
