"""
Predict from the sky model (africanus/rime/examples/predict.py:494-498): the model-level entry
points take (stokes, spi, ref_freq) and evaluate spectral_model -> convert on the device inside the call.  They must
equal the chain of the three stand-alone functions bit for bit (same kernels, same operands), and that chain is pinned
to the reference by the goldens G8 (spectral_model), G11 (convert), G3 (im_to_vis); here also against the oracle chain.
"""
import numpy as np
import pytest
from numpy.testing import assert_array_equal

import oracle
from codex_africanus_amd import dft, rime
from codex_africanus_amd.model.spectral import spectral_model
from codex_africanus_amd.model.coherency import convert
from codex_africanus_amd.testing import synthetic_inputs

pytestmark = pytest.mark.gpu

LINEAR = [["XX", "XY"], ["YX", "YY"]]
CIRCULAR = [["RR", "RL"], ["LR", "LL"]]


def _sky(seed, nsrc, nspi, npol=4):
    rng = np.random.default_rng(seed)
    stokes = np.concatenate([rng.lognormal(0, 1, (nsrc, 1)), 0.1 * rng.standard_normal((nsrc, npol - 1))], axis=1)
    spi = rng.uniform(-1.0, 0.3, (nsrc, nspi, npol))
    ref_freq = rng.uniform(0.9e9, 1.5e9, nsrc)
    return stokes[:, :npol], spi, ref_freq


@pytest.mark.parametrize("schema, npol, base", [(LINEAR, 4, 0), (CIRCULAR, 4, "log"), (["XX", "YY"], 2, 0),
                                                (["RR", "LL"], 4, [0, 1, 2, 0]), (["XX", "XY", "YX", "YY"], 4, "log10")])
@pytest.mark.parametrize("nchan", [5, 64])
def test_im_to_vis_from_model_equals_the_chain(schema, npol, base, nchan):
    d = synthetic_inputs(seed=5, nrow=300, nchan=nchan, nsrc=37, nant=7)
    stokes, spi, ref_freq = _sky(11, 37, 2, npol)
    out = dft.im_to_vis_from_model(stokes, spi, ref_freq, d["uvw"], d["lm"], d["frequency"], corr_schema=schema, base=base)
    spec = spectral_model(stokes, spi, ref_freq, d["frequency"], base=base)
    image = convert(spec, ["I", "Q", "U", "V"][:npol], schema)
    chain = dft.im_to_vis(image.reshape(37, nchan, -1), d["uvw"], d["lm"], d["frequency"])
    assert out.shape == (300, nchan) + np.shape(schema) and out.dtype == np.complex128
    assert_array_equal(out.reshape(chain.shape), chain)
    # and the oracle's chain (C restatements of the reference's three functions)
    ospec = oracle.spectral_model(stokes, spi, ref_freq, d["frequency"], base=base)
    oimage = oracle.convert(ospec, ["I", "Q", "U", "V"][:npol], schema).reshape(37, nchan, -1)
    ref = oracle.im_to_vis(oimage, d["uvw"], d["lm"], d["frequency"])
    assert np.abs(out.reshape(ref.shape) - ref).max() <= 1e-11 * np.abs(oimage).sum(axis=0).max()


def test_im_to_vis_from_model_device_resident_and_errors():
    import torch
    d = synthetic_inputs(seed=6, nrow=130, nchan=16, nsrc=9, nant=5)
    stokes, spi, ref_freq = _sky(12, 9, 1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = dft.im_to_vis_from_model(t(stokes), t(spi), t(ref_freq), t(d["uvw"]), t(d["lm"]), t(d["frequency"]))
    host = dft.im_to_vis_from_model(stokes, spi, ref_freq, d["uvw"], d["lm"], d["frequency"])
    assert_array_equal(out.cpu().numpy(), host)
    with pytest.raises(ValueError, match="Correlations on stokes and spi"):
        dft.im_to_vis_from_model(stokes, spi[:, :, :2], ref_freq, d["uvw"], d["lm"], d["frequency"])
    with pytest.raises(ValueError, match="Invalid base"):
        dft.im_to_vis_from_model(stokes, spi, ref_freq, d["uvw"], d["lm"], d["frequency"], base="cubic")
    with pytest.raises(ValueError, match="Unknown output"):
        dft.im_to_vis_from_model(stokes, spi, ref_freq, d["uvw"], d["lm"], d["frequency"], corr_schema=["XX", "ZZ"])
    with pytest.raises(ValueError, match="number of sources"):
        dft.im_to_vis_from_model(stokes, spi, ref_freq[:4], d["uvw"], d["lm"], d["frequency"])


@pytest.mark.parametrize("schema", [LINEAR, CIRCULAR])
def test_fused_predict_from_model_equals_brightness_call(schema):
    from test_gpu_fused import _problem
    d = _problem(21, 900, 8, 23, 9)
    stokes, spi, ref_freq = _sky(13, 23, 2)
    args = (d["time_index"], d["ant1"], d["ant2"], d["lm"], d["uvw"], d["frequency"])
    beam = (d["beam"], d["extents"], d["beam_freq_map"], d["pa"], d["pe"], d["as"])
    X = convert(spectral_model(stokes, spi, ref_freq, d["frequency"]), ["I", "Q", "U", "V"], schema)
    with_b = rime.fused_predict_vis(*args, X, *beam)
    with_m = rime.fused_predict_vis(*args, None, *beam, stokes=stokes, spi=spi, ref_freq=ref_freq, corr_schema=schema)
    assert_array_equal(with_m, with_b)
    # without DDEs: the model-level direct transform (phase_delay's clamped n)
    nb = rime.fused_predict_vis(*args, X)
    nm = rime.fused_predict_vis(*args, stokes=stokes, spi=spi, ref_freq=ref_freq, corr_schema=schema)
    assert_array_equal(nm, nb)
    with pytest.raises(ValueError, match="either brightness or all of"):
        rime.fused_predict_vis(*args, X, stokes=stokes, spi=spi, ref_freq=ref_freq)
    with pytest.raises(ValueError, match="2 x 2 schema"):
        rime.fused_predict_vis(*args, stokes=stokes, spi=spi, ref_freq=ref_freq, corr_schema=["XX", "YY"])
