"""GPU parity of rime.wsclean_predict / spectra (fused term producers) against the golden
vectors of the reference (tests/golden/g7_wsclean.npz, africanus/rime/wsclean_predict.py:86 and
africanus/model/wsclean/spec_model.py:70 run by tests/golden/make_golden.py) and against the oracle."""
import numpy as np
import pytest

import oracle
from codex_africanus_amd import dft, rime
from codex_africanus_amd.rime.wsclean_predict import spectra

pytestmark = pytest.mark.gpu


def _args(g7, tag, freq_key=None):
    st = np.where(g7[tag + "_is_gauss"], "GAUSSIAN", "POINT")
    return [g7[tag + "_uvw"], g7[tag + "_lm"], st, g7[tag + "_flux"], g7[tag + "_coeffs"], g7[tag + "_log_poly"],
            g7[tag + "_ref_freq"], g7[tag + "_gauss_shape"], g7[freq_key or tag + "_freq"]]


@pytest.fixture(autouse=True)
def _restore_mode():
    m = dft.get_mode()
    yield
    dft.set_mode(m)


@pytest.mark.parametrize("tag", ["small", "big"])
def test_spectra_golden(g7, tag):
    a = _args(g7, tag)
    out = spectra(a[3], a[4], a[5], a[6], a[8])
    ref = g7[tag + "_spectrum"]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    # device log/exp are within 1 ulp of libm's: 4 ulp on the polynomial
    assert np.abs(out - ref).max() <= 1e-15 * np.abs(ref).max() * 8
    # scalar log_poly broadcasts (spec_model.py:57-63)
    np.testing.assert_allclose(spectra(a[3], a[4], False, a[6], a[8]),
                               oracle.spectra(a[3], a[4], np.zeros(a[3].shape[0], bool), a[6], a[8]), rtol=1e-14)


@pytest.mark.parametrize("tag", ["small", "big"])
@pytest.mark.parametrize("mode", ["auto", "exact", "recurrence"])
def test_wsclean_predict_golden(g7, tag, mode):
    dft.set_mode(mode)
    out = rime.wsclean_predict(*_args(g7, tag))
    ref = g7[tag + "_vis"]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    scale = np.abs(g7[tag + "_spectrum"]).sum(axis=0).max()   # sum_s |spectrum|: the natural error scale
    tol = 1e-14 if mode == "exact" else 1e-11                  # north_star: < 1e-8 absolute
    assert np.abs(out - ref).max() <= tol * scale


def test_wsclean_predict_nonuniform_takes_exact_path(g7):
    out = rime.wsclean_predict(*_args(g7, "small", "small_freq_nonuniform"))
    scale = np.abs(g7["small_spectrum"]).sum(axis=0).max()
    assert np.abs(out - g7["small_vis_nonuniform"]).max() <= 1e-14 * scale


def test_wsclean_predict_source_orderings_and_types(g7):
    """all points, all Gaussians, and boolean source_type; points equal the CASA direct transform."""
    a = _args(g7, "big")
    nsrc = a[2].shape[0]
    for st in (np.full(nsrc, "POINT"), np.full(nsrc, "GAUSSIAN")):
        b = list(a)
        b[2] = st
        ref = oracle.wsclean_predict(*b)
        out = rime.wsclean_predict(*b)
        assert np.abs(out - ref).max() <= 1e-11 * np.abs(g7["big_spectrum"]).sum(axis=0).max()
    b = list(a)
    b[2] = g7["big_is_gauss"]
    np.testing.assert_array_equal(rime.wsclean_predict(*b), rime.wsclean_predict(*a))
    b[2] = np.full(nsrc, "POINT")
    spec = spectra(a[3], a[4], a[5], a[6], a[8])
    vis = dft.im_to_vis(spec[:, :, None].copy(), a[0], a[1], a[8], convention="casa")
    assert np.abs(rime.wsclean_predict(*b) - vis).max() <= 1e-11 * np.abs(spec).sum(axis=0).max()


def test_wsclean_predict_dtype_shapes_and_errors(g7):
    a = _args(g7, "small")
    f32 = [x.astype(np.float32) if x.dtype == np.float64 else x for x in a]
    out = rime.wsclean_predict(*f32)
    assert out.dtype == np.complex64 and out.shape == g7["small_vis"].shape
    ref = oracle.wsclean_predict(*[x.astype(np.float64) if x.dtype == np.float32 else x for x in f32])
    assert np.abs(out - ref).max() <= 1e-6 * np.abs(ref).max()
    b = list(a)
    b[2] = np.where(np.arange(b[2].shape[0]) == 0, "DISK", b[2])
    with pytest.raises(ValueError, match="POINT or GAUSSIAN"):
        rime.wsclean_predict(*b)
    with pytest.raises(ValueError, match="don't match"):
        rime.wsclean_predict(a[0], a[1], a[2], a[3], a[4][:-1], a[5], a[6], a[7], a[8])
    # empty component list -> zeros (wsclean_predict.py:27); empty rows -> empty
    z = rime.wsclean_predict(a[0], a[1][:0], a[2][:0], a[3][:0], a[4][:0], a[5][:0], a[6][:0], a[7][:0], a[8])
    assert z.shape == (a[0].shape[0], a[8].shape[0], 1) and not z.any()
    assert rime.wsclean_predict(a[0][:0], *a[1:]).shape == (0, a[8].shape[0], 1)


@pytest.mark.parametrize("nchan", [1, 7, 8, 23, 41, 64, 100])
def test_wsclean_predict_channel_tilings(nchan):
    """every tile width (8..40) and ragged last tiles; 300 components so both passes loop"""
    rs = np.random.RandomState(nchan)
    nsrc, nrow = 300, 333
    isg = rs.randint(0, 2, nsrc).astype(bool)
    st = np.where(isg, "GAUSSIAN", "POINT")
    uvw = rs.normal(size=(nrow, 3)) * 2000.0
    lm = rs.normal(size=(nsrc, 2)) * 2e-2
    flux, coeffs = rs.uniform(0.1, 2.0, nsrc), rs.normal(size=(nsrc, 3)) * [0.7, 0.2, 0.05]
    log_poly = rs.randint(0, 2, nsrc).astype(bool)
    gshape = np.stack([rs.uniform(0, 3e-4, nsrc), rs.uniform(0, 2e-4, nsrc), rs.uniform(0, np.pi, nsrc)], axis=1)
    gshape[isg.nonzero()[0][:2], 0] = 0.0          # emaj == 0: er = emin / 1 (wsclean_predict.py:54)
    freq = np.linspace(0.9e9, 1.7e9, nchan) if nchan > 1 else np.array([1.1e9])
    ref_freq = np.full(nsrc, 1.3e9)
    args = (uvw, lm, st, flux, coeffs, log_poly, ref_freq, gshape, freq)
    ref = oracle.wsclean_predict(*args)
    out = rime.wsclean_predict(*args)
    scale = np.abs(oracle.spectra(flux, coeffs, log_poly, ref_freq, freq)).sum(axis=0).max()
    assert np.abs(out - ref).max() <= 1e-11 * scale


def test_wsclean_predict_torch_zero_copy(g7):
    import torch
    a = _args(g7, "big")
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) if x.dtype.kind in "fb" else x for x in a]
    t[2] = torch.from_numpy(g7["big_is_gauss"]).to(dev)
    out = rime.wsclean_predict(*t)
    assert isinstance(out, torch.Tensor) and out.device.type == "cuda" and out.dtype == torch.complex128
    np.testing.assert_array_equal(out.cpu().numpy(), rime.wsclean_predict(*a))


def test_chunked_and_dask_wsclean_predict(g7):
    """source / row / chan chunking as africanus/rime/tests/test_wsclean_predict.py:63-110"""
    from codex_africanus_amd import chunked
    a = _args(g7, "big")
    ref = g7["big_vis"]
    scale = np.abs(g7["big_spectrum"]).sum(axis=0).max()
    out = chunked.wsclean_predict(*a, chunks={"source": 10, "row": 50, "chan": (40, 30)})
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 1e-11 * scale
    # the dask front-end of the same chunking: tests/dask_cases.py (run by tests/test_gpu_dask_conda.py)
