#!/opt/conda/bin/python3.9
"""
G15 -- Gaussian and point sources without direction-dependent terms, by the REFERENCE's own functions, as
africanus/rime/examples/predict.py:107-134,525 composes them (conda python 3.9: numba 0.54 + tests/golden/ref_shim.py):

    NUMBA_CACHE_DIR=/tmp/numba_cache PYTHONPATH=/root/reference:tests/golden:. \
        /opt/conda/bin/python3.9 tests/golden/make_golden_gauss.py

    phase = africanus.rime.phase_delay(lm, uvw, frequency, convention)
    shape = africanus.model.shape.gaussian(uvw, frequency, shape_params)
    coh   = np.einsum("srf,srf,sfij->srfij", phase, shape, brightness)
    vis   = africanus.rime.predict_vis(time_index, antenna1, antenna2, None, coh, None, None, None, None)

on bands of >= 14 uniformly spaced channels -- rising (40 channels: one short 64-channel tile of the MFMA-accumulator
form), falling (33), and 80 channels (a full tile and a 16-channel tail) -- which is where the build's
``af_gauss_predict_c128`` takes its MFMA-accumulator kernels (tests/test_gpu_gauss_dft.py).  Stores inputs and results in
g15_gauss.npz.
"""
import os

import ref_shim  # noqa: F401  (must come first)
import numpy as np

from africanus.rime import phase_delay, predict_vis
from africanus.model.shape import gaussian as gaussian_shape

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(15)
    nsrc, nant, ntime = 11, 5, 4
    a1, a2 = np.triu_indices(nant, 1)
    nbl = a1.size
    nrow = nbl * ntime
    out = dict(antenna1=np.tile(a1, ntime).astype(np.int32), antenna2=np.tile(a2, ntime).astype(np.int32),
               time_index=np.repeat(np.arange(ntime), nbl).astype(np.int32))
    lm = rng.uniform(-0.04, 0.04, (nsrc, 2))
    uvw = np.stack([rng.uniform(-3000, 3000, nrow), rng.uniform(-3000, 3000, nrow), rng.uniform(-300, 300, nrow)], axis=1)
    shape_params = np.stack([rng.uniform(0, 3e-4, nsrc), rng.uniform(0, 2e-4, nsrc), rng.uniform(0, np.pi, nsrc)], axis=1)
    shape_params[::3] = 0.0              # point sources in between
    out.update(lm=lm, uvw=uvw, shape_params=shape_params)
    bands = {"rising40": np.linspace(0.856e9, 1.712e9, 40), "falling33": np.linspace(1.5e9, 0.9e9, 33),
             "rising80": np.linspace(1.0e9, 1.4e9, 80)}
    for name, freq in bands.items():
        nchan = freq.size
        st = np.stack([rng.lognormal(0, 1, nsrc)] + [0.1 * rng.standard_normal(nsrc) for _ in range(3)], axis=1)
        slope = rng.uniform(-0.8, 0.2, nsrc)
        spec = (freq[None, :] / freq[0]) ** slope[:, None]                     # (source, chan)
        I, Q, U, V = (st[:, k, None] * spec for k in range(4))               # noqa: E741
        X = np.stack([I + Q, U + 1j * V, U - 1j * V, I - Q], axis=-1).reshape(nsrc, nchan, 2, 2)
        out["frequency_" + name] = freq
        out["brightness_" + name] = X
        for conv in ("fourier", "casa"):
            phase = phase_delay(lm, uvw, freq, convention=conv)
            shape = gaussian_shape(uvw, freq, shape_params)
            coh = np.einsum("srf,srf,sfij->srfij", phase, shape, X)
            vis = predict_vis(out["time_index"], out["antenna1"], out["antenna2"], None, coh, None, None, None, None)
            out["vis_%s_%s" % (name, conv)] = vis
            out["scale_%s" % name] = np.abs(X).sum(axis=(0, 2, 3)).max()
    np.savez_compressed(os.path.join(HERE, "g15_gauss.npz"), **out)
    print("wrote g15_gauss.npz:", {k: v.shape for k, v in out.items() if k.startswith("vis_")})


if __name__ == "__main__":
    main()
