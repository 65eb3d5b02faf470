"""
af_im_to_vis_f32: float32 inputs -> complex64, phases in float64, sums in float32; phasors by the fp64 recurrence
rounded once (calls whose phases stay below ~100 rad: golden G3) or by the float32 rotation recurrence from an
fp64-phase anchor (everything larger: G13, the sweep below)
(csrc/af_im_to_vis_f32.hip; the reference case is africanus/dft/kernels.py:26-31 with every input float32, where its
whole loop runs in float32).  Contract: CLOSER to the float64 transform of the same float32 inputs
than the reference's own float32 result.  The reference's float32 results are (a) the recorded golden vector
g3["vis_f32"] (small baselines) and (b) at realistic baselines the oracle's float32 restatement of the same loop
(oracle.im_to_vis on float32 arrays runs the reference's operation order in float32; it is pinned bit for bit to the
recorded vector in tests/test_oracle_golden.py where the recording exists).
"""
import numpy as np
import pytest

import oracle
from codex_africanus_amd import dft

pytestmark = pytest.mark.gpu


def _truth(img, uvw, lm, freq, conv="fourier"):
    """float64 transform of the float32 inputs (promoted exactly)"""
    cplx = np.iscomplexobj(img)
    return oracle.im_to_vis(img.astype(np.complex128 if cplx else np.float64), uvw.astype(np.float64),
                            lm.astype(np.float64), freq.astype(np.float64), convention=conv)


def _inputs(rng, nsrc, nrow, nchan, ncorr, cplx, scale=4e3, uniform_f32=False):
    lm = ((rng.random((nsrc, 2)) - 0.5) * 0.1).astype(np.float32)
    uvw = ((rng.random((nrow, 3)) - 0.5) * 2 * scale).astype(np.float32)
    uvw[:, 2] *= 0.1
    if uniform_f32:
        freq = (0.856e9 + np.arange(nchan) * 13586432.0).astype(np.float32)     # multiples of 128 Hz: exact in float32
        assert np.all(freq.astype(np.float64) == 0.856e9 + np.arange(nchan) * 13586432.0)
    else:
        freq = np.linspace(0.856e9, 1.712e9, nchan).astype(np.float32)
    img = rng.standard_normal((nsrc, nchan, ncorr)).astype(np.float32)
    img[rng.random(img.shape) < 0.15] = 0.0
    if cplx:
        img = (img + 1j * rng.standard_normal(img.shape)).astype(np.complex64)
    return img, uvw, lm, freq


def test_golden_f32_closer_than_the_reference():
    g3 = np.load(__file__.replace("test_gpu_f32.py", "golden/g3_im_to_vis.npz"))
    img, uvw, lm, fr = (g3["img_r4"].astype(np.float32), g3["uvw32"], g3["lm"].astype(np.float32),
                        g3["frequency"].astype(np.float32))
    got = dft.im_to_vis(img, uvw, lm, fr)
    assert got.dtype == np.complex64 and got.shape == g3["vis_f32"].shape
    truth = _truth(img, uvw, lm, fr)
    e_ours, e_ref = np.abs(got - truth).max(), np.abs(g3["vis_f32"] - truth).max()
    assert e_ours <= e_ref, (e_ours, e_ref)
    assert e_ours < 2e-5 * np.abs(truth).max()


G13 = np.load(__file__.replace("test_gpu_f32.py", "golden/g13_f32.npz"))


@pytest.mark.parametrize("case", [str(c) for c in G13["cases"]])
def test_closer_than_the_reference_float32_loop_at_real_baselines(case):
    """G13 (tests/golden/make_golden_f32.py): the REAL reference's float32 results at 4 km baselines, 40 rows x 40
    sources, 6 ... 70 channels, 1 / 2 / 4 correlations, real and complex pixels, both classes of float32 band (a
    linspace cast to float32: rounded, the corrected recurrence; an exactly representable grid: the plain one).  The
    reference's float32 phases carry ~1e-4 of the peak visibility there; this entry must be several times closer to the
    float64 transform of the same inputs."""
    key, conv = case.split("|")
    img, uvw, lm, fr, ref32 = (G13[key + s] for s in ("_img", "_uvw", "_lm", "_freq", "_vis"))
    got = dft.im_to_vis(img, uvw, lm, fr, convention=conv)
    assert got.dtype == np.complex64 and got.shape == ref32.shape
    truth = _truth(img, uvw, lm, fr, conv)
    scale = np.abs(truth).max()
    e_ours, e_ref = np.abs(got - truth).max() / scale, np.abs(ref32 - truth).max() / scale
    assert e_ours < 0.25 * e_ref, (case, e_ours, e_ref)
    assert e_ours < 3e-5, (case, e_ours)


@pytest.mark.parametrize("ncorr", [1, 2, 4])
@pytest.mark.parametrize("cplx", [False, True])
def test_against_the_float64_transform_more_rows_and_tiles(cplx, ncorr):
    """700 rows (partial blocks), 70 channels (several tiles + a remainder) against the float64 transform"""
    rng = np.random.default_rng(17 + ncorr)
    for uniform in (False, True):
        img, uvw, lm, fr = _inputs(rng, 40, 700, 70, ncorr, cplx, uniform_f32=uniform)
        got = dft.im_to_vis(img, uvw, lm, fr)
        truth = _truth(img, uvw, lm, fr)
        assert np.abs(got - truth).max() < 3e-5 * np.abs(truth).max()


@pytest.mark.parametrize("scale", [0.5, 8.0, 60.0, 500.0, 4e3, 3e4])
def test_both_phasor_forms_across_baseline_lengths(scale):
    """The library picks the phasor form on the device from a bound of the call's largest phase (rows' max |uvw| x
    sources' max |lmn| x max frequency, ~100 rad): metre baselines take the fp64 recurrence, kilometre baselines the
    float32 rotation recurrence.  Either side of the switch, both band classes, real and complex pixels: within 3e-6
    of the peak visibility of the float64 transform (the float32 sums alone carry ~1e-6; the reference's float32 loop
    is at 1e-4 from ~4 km on, golden G13)."""
    rng = np.random.default_rng(int(scale * 10) + 3)
    for uniform in (False, True):
        for cplx in (False, True):
            img, uvw, lm, fr = _inputs(rng, 60, 300, 40, 4, cplx, scale=scale, uniform_f32=uniform)
            got = dft.im_to_vis(img, uvw, lm, fr)
            truth = _truth(img, uvw, lm, fr)
            peak = np.abs(truth).max()
            e_ours = np.abs(got - truth).max() / peak
            assert e_ours < 3e-6, (scale, uniform, cplx, e_ours)


def test_modes_classes_and_device_resident():
    """non-uniform float32 band -> the per-channel kernel (class 2); 'exact' forces it; 'recurrence' treats the rounded
    linspace as the uniform grid it was meant to be (error of the order of the float32 rounding of the axis);
    torch tensors in -> tensor out"""
    import torch
    rng = np.random.default_rng(5)
    img, uvw, lm, fr = _inputs(rng, 30, 500, 48, 4, False)
    truth = _truth(img, uvw, lm, fr)
    scale = np.abs(truth).max()
    auto = dft.im_to_vis(img, uvw, lm, fr)
    with dft.mode("exact"):
        exact = dft.im_to_vis(img, uvw, lm, fr)
    with dft.mode("recurrence"):
        rec = dft.im_to_vis(img, uvw, lm, fr)
    assert np.abs(auto - truth).max() / scale < 5e-5
    assert np.abs(exact - truth).max() / scale < 2e-5
    assert np.abs(rec - truth).max() / scale < 2e-3
    fr2 = (fr.astype(np.float64) * (1 + 0.01 * rng.random(48))).astype(np.float32)      # really non-uniform
    got = dft.im_to_vis(img, uvw, lm, fr2)
    t2 = _truth(img, uvw, lm, fr2)
    assert np.abs(got - t2).max() / np.abs(t2).max() < 2e-5
    dev = torch.device("cuda:0")
    t = dft.im_to_vis(*(torch.from_numpy(a).to(dev) for a in (img, uvw, lm, fr)))
    assert t.dtype == torch.complex64 and np.array_equal(t.cpu().numpy(), auto)


def test_zero_columns_nan_sources_and_empty():
    """kernels.py:54,64 in single precision: a source outside the unit disc poisons only the columns where it has a
    non-zero pixel; all-zero columns stay exactly zero; no sources -> zeros"""
    rng = np.random.default_rng(9)
    img, uvw, lm, fr = _inputs(rng, 12, 300, 30, 4, False)
    lm[4] = [0.9, 0.8]
    img[4] = 0.0
    img[4, 2, 1] = 1.5
    img[:, 7, 3] = 0.0
    got = dft.im_to_vis(img, uvw, lm, fr)
    ref = oracle.im_to_vis(img.astype(np.float64), uvw.astype(np.float64), lm.astype(np.float64), fr.astype(np.float64))
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.isnan(got[:, 2, 1]).all() and not np.isnan(got[:, 2, 0]).any()
    assert np.all(got[:, 7, 3] == 0)
    ok = ~np.isnan(ref)
    assert np.abs(got[ok] - ref[ok]).max() < 5e-5 * np.abs(ref[ok]).max()
    empty = dft.im_to_vis(img[:0], uvw, lm[:0], fr)
    assert empty.shape == (300, 30, 4) and not empty.any()


def test_full_size_c2_float32_linearity_and_sample():
    """BASELINE configs[1] in single precision, device resident: x2 image -> x2 visibilities exactly; a row sample
    against the float64 transform"""
    import torch
    from codex_africanus_amd.testing import synthetic_inputs, real_image
    dev = torch.device("cuda:0")
    d = synthetic_inputs(seed=0, nrow=16, nchan=64, nsrc=1000, nant=64)
    rng = np.random.default_rng(1000)
    nrow = 1000000
    uvw = np.empty((nrow, 3), np.float32)
    uvw[:, 0] = rng.uniform(-4000, 4000, nrow); uvw[:, 1] = rng.uniform(-4000, 4000, nrow); uvw[:, 2] = rng.uniform(-400, 400, nrow)
    img, lm, fr = real_image(d).astype(np.float32), d["lm"].astype(np.float32), d["frequency"].astype(np.float32)
    t = [torch.from_numpy(a).to(dev) for a in (img, uvw, lm, fr)]
    v1 = dft.im_to_vis(*t)
    v2 = dft.im_to_vis(t[0] * 2, *t[1:])
    assert torch.equal(v2, v1 * 2)
    rows = np.linspace(0, nrow - 1, 64).astype(np.int64)
    truth = _truth(img, uvw[rows], lm, fr)
    got = v1[torch.from_numpy(rows).to(dev)].cpu().numpy()
    assert np.abs(got - truth).max() < 1e-4 * np.abs(truth).max()


# ------------------------------------------------------------------------------------------- vis_to_im, single precision
def _truth_v2i(vis, uvw, lm, fr, flags, conv="fourier"):
    return oracle.vis_to_im(vis.astype(np.complex128), uvw.astype(np.float64), lm.astype(np.float64),
                            fr.astype(np.float64), flags, convention=conv)


@pytest.mark.parametrize("kind", ["linspace", "exact"])
def test_vis_to_im_f32_closer_than_the_reference(kind):
    """G13: the REAL reference's float32 vis_to_im (africanus/dft/kernels.py:72-148 with complex64 visibilities and
    float32 coordinates) at 4 km baselines, 300 rows x 24 chan x 4 corr -> 12 sources, 5 % flags"""
    vis, uvw, lm, flags = G13["v2i_vis"], G13["v2i_uvw"], G13["v2i_lm"], G13["v2i_flags"]
    fr, ref32 = G13["v2i_%s_freq" % kind], G13["v2i_%s_im" % kind]
    got = dft.vis_to_im(vis, uvw, lm, fr, flags)
    assert got.dtype == np.float32 and got.shape == ref32.shape
    truth = _truth_v2i(vis, uvw, lm, fr, flags)
    scale = np.abs(truth).max()
    e_ours, e_ref = np.abs(got - truth).max() / scale, np.abs(ref32 - truth).max() / scale
    assert e_ours < 0.25 * e_ref, (kind, e_ours, e_ref)
    assert e_ours < 3e-5


@pytest.mark.parametrize("ncorr", [1, 2, 4])
def test_vis_to_im_f32_shapes_flags_nonfinite_and_adjointness(ncorr):
    """several row partitions and channel tiles (5000 rows, 70 channels), a fully flagged channel (stays exactly 0), a
    non-finite uvw row (its unflagged cells poison every source, flagged ones nothing), a source outside the unit
    disc (NaN where the channel has data), 'casa', the per-channel kernel, and <y, R x> = <R^H y, x> against
    af_im_to_vis_f32"""
    rng = np.random.default_rng(40 + ncorr)
    nrow, nchan, nsrc = 5000, 70, 37
    lm = ((rng.random((nsrc, 2)) - 0.5) * 0.1).astype(np.float32)
    uvw = ((rng.random((nrow, 3)) - 0.5) * 8e3).astype(np.float32)
    uvw[:, 2] *= np.float32(0.1)
    fr = np.linspace(0.9e9, 1.6e9, nchan).astype(np.float32)
    vis = (rng.standard_normal((nrow, nchan, ncorr)) + 1j * rng.standard_normal((nrow, nchan, ncorr))).astype(np.complex64)
    flags = rng.random((nrow, nchan, ncorr)) < 0.03
    flags[:, 11, :] = True
    for conv in ("fourier", "casa"):
        got = dft.vis_to_im(vis, uvw, lm, fr, flags, convention=conv)
        truth = _truth_v2i(vis, uvw, lm, fr, flags, conv)
        assert got.dtype == np.float32 and np.all(got[:, 11, :] == 0)
        assert np.abs(got - truth).max() < 3e-5 * np.abs(truth).max()
    with dft.mode("exact"):
        ex = dft.vis_to_im(vis, uvw, lm, fr, flags)
    assert np.abs(ex - truth_f(vis, uvw, lm, fr, flags)).max() < 3e-5 * np.abs(truth).max()
    # adjointness with the single-precision forward transform (unflagged cells only)
    keep = ~flags.any(axis=2)
    x = rng.standard_normal((nsrc, nchan, ncorr)).astype(np.float32)
    rx = dft.im_to_vis(x, uvw, lm, fr)                       # im_to_vis 'fourier' = exp(-i ...); vis_to_im 'fourier' = exp(+i ...)
    lhs = np.sum((vis.conj() * rx).real[keep])               # <y, R x> over unflagged cells, real part
    rhs = np.sum(dft.vis_to_im(vis.conj(), uvw, lm, fr, flags, convention="casa").astype(np.float64) * x)
    assert abs(lhs - rhs) < 2e-4 * (abs(lhs) + np.abs(vis).sum() * 1e-3)
    # non-finite row and source
    uvw2 = uvw.copy()
    uvw2[123] = np.nan
    flags2 = flags.copy()
    flags2[123, 5, :] = True
    g2 = dft.vis_to_im(vis, uvw2, lm, fr, flags2)
    t2 = _truth_v2i(vis, uvw2, lm, fr, flags2)
    assert np.array_equal(np.isnan(g2), np.isnan(t2))
    assert not np.isnan(g2[:, 5, :]).any() and np.all(g2[:, 11, :] == 0)
    lm2 = lm.copy()
    lm2[3] = [0.9, 0.8]
    g3 = dft.vis_to_im(vis, uvw, lm2, fr, flags)
    assert np.isnan(g3[3, 0]).all() and np.all(g3[3, 11] == 0) and not np.isnan(np.delete(g3, 3, axis=0)).any()


def truth_f(vis, uvw, lm, fr, flags):
    return _truth_v2i(vis, uvw, lm, fr, flags)
