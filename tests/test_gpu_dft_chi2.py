"""
af_im_to_vis_chi2_f64 / dft.im_to_vis_chi2: the transform (africanus/dft/kernels.py:14-69) and the per-channel chi^2 of
its result in one call -- in the MFMA kernels' epilogue where they run, by the separate pass everywhere else.  The
visibilities must equal im_to_vis's bit for bit; chi^2 must equal numpy's sum over the SAME visibilities to rounding
(the device sums with atomics: order varies) whichever path computed it.
"""
import numpy as np
import pytest

from codex_africanus_amd import dft
from codex_africanus_amd.testing import synthetic_inputs, real_image

pytestmark = pytest.mark.gpu


def _check(image, d, freq, weight=False, convention="fourier", seed=0):
    rng = np.random.default_rng(seed)
    vis = dft.im_to_vis(image, d["uvw"], d["lm"], freq, convention=convention)
    data = vis + 0.1 * (rng.standard_normal(vis.shape) + 1j * rng.standard_normal(vis.shape))
    w = rng.random(vis.shape) if weight else None
    got_vis, chi2 = dft.im_to_vis_chi2(image, d["uvw"], d["lm"], freq, data, w, convention=convention)
    assert got_vis.dtype == np.complex128 and chi2.dtype == np.float64 and chi2.shape == (freq.shape[0],)
    np.testing.assert_array_equal(got_vis, vis)                      # NaN == NaN here
    a = np.abs(data - vis) ** 2
    want = ((a * w) if weight else a).sum(axis=(0, 2))
    ok = np.isfinite(want)
    assert np.allclose(chi2[ok], want[ok], rtol=1e-12, atol=0)
    assert np.isnan(chi2[~ok]).all()
    return chi2


# the MFMA path: 4 correlations, uniform band; tiles of 32 + tails of 16 / 32, short last tiles, rows off the 64-row block
@pytest.mark.parametrize("nrow, nchan", [(64, 14), (1000, 16), (333, 32), (700, 45), (1, 64), (4097, 64), (500, 70), (250, 100)])
@pytest.mark.parametrize("cplx", [False, True])
def test_epilogue_chi2_on_the_mfma_path(nrow, nchan, cplx):
    d = synthetic_inputs(seed=5, nrow=nrow, nchan=nchan, nsrc=37, nant=7)
    image = real_image(d)
    if cplx:
        image = image * (1.0 + 0.3j)
    _check(image, d, d["frequency"], weight=(nrow % 2 == 0), seed=nrow)


def test_fallbacks_give_the_same_chi2():
    d = synthetic_inputs(seed=6, nrow=900, nchan=40, nsrc=23, nant=7)
    image = real_image(d)
    rng = np.random.default_rng(1)
    _check(image, d, np.sort(rng.uniform(0.9e9, 1.7e9, 40)))                  # non-uniform band: kernels decide on the device
    _check(image[:, :, :2], d, d["frequency"], weight=True)                    # 2 correlations: no MFMA path
    _check(image[:, :9], d, d["frequency"][:9])                                # too few channels for it
    _check(image, d, d["frequency"], convention="casa")
    z = image.copy()
    z[:, 3, 1] = 0.0                                                           # an all-zero (chan, corr) column: rewritten after
    _check(z, d, d["frequency"])                                               # the MFMA kernels -> chi^2 recomputed
    bad = dict(d, lm=d["lm"].copy())
    bad["lm"][2] = [0.9, 0.8]                                                  # NaN source (n is not clamped): NaN columns
    _check(image, bad, d["frequency"])
    with dft.mode("exact"):
        _check(image, d, d["frequency"])


def test_device_resident_and_empty_cases():
    import torch
    dev = torch.device("cuda:0")
    d = synthetic_inputs(seed=7, nrow=2000, nchan=64, nsrc=50, nant=7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    image = real_image(d)
    vis = dft.im_to_vis(t(image), t(d["uvw"]), t(d["lm"]), t(d["frequency"]))
    data = vis + 0.01
    v2, chi2 = dft.im_to_vis_chi2(t(image), t(d["uvw"]), t(d["lm"]), t(d["frequency"]), data)
    assert v2.is_cuda and chi2.is_cuda and torch.equal(v2, vis)
    want = ((data - vis).abs() ** 2).sum(dim=(0, 2))
    assert torch.allclose(chi2, want, rtol=1e-12, atol=0)
    # no sources: zero visibilities, chi^2 of the data alone; no rows: zeros
    v0, c0 = dft.im_to_vis_chi2(image[:0], d["uvw"], d["lm"][:0], d["frequency"], data.cpu().numpy())
    assert not v0.any() and np.allclose(c0, (np.abs(data.cpu().numpy()) ** 2).sum(axis=(0, 2)), rtol=1e-12)
    v0, c0 = dft.im_to_vis_chi2(image, d["uvw"][:0], d["lm"], d["frequency"], np.zeros((0, 64, 4), np.complex128))
    assert v0.shape == (0, 64, 4) and not c0.any()
    with pytest.raises(ValueError, match="data must have the shape"):
        dft.im_to_vis_chi2(image, d["uvw"], d["lm"], d["frequency"], np.zeros((5, 64, 4), np.complex128))
