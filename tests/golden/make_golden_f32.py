#!/opt/conda/bin/python3.9
"""
G13: the REAL reference's float32 results of im_to_vis (and vis_to_im) at realistic baselines -- every input float32,
so the reference runs its whole loop in float32 (africanus/dft/kernels.py:26-31, africanus/util/type_inference.py:24-26).
These vectors are what "closer to the float64 transform than the reference's own float32 output" (af_im_to_vis_f32,
VERDICT r2 item 6) is measured against; tests/test_gpu_f32.py computes the float64 transform of the same float32
inputs with the oracle.

Run (build container only):

    NUMBA_CACHE_DIR=/tmp/numba_cache PYTHONPATH=/root/reference:tests/golden \
        /opt/conda/bin/python3.9 tests/golden/make_golden_f32.py
"""
import os

import ref_shim  # noqa: F401  (must come first)
import numpy as np

from africanus.dft.kernels import im_to_vis, vis_to_im

HERE = os.path.dirname(os.path.abspath(__file__))
NSRC, NROW = 40, 40


def band(kind, nchan):
    if kind == "exact":      # multiples of 128 Hz below 2^31 Hz: exactly representable, an exact progression in float32
        return (0.856e9 + np.arange(nchan) * 13586432.0).astype(np.float32)
    return np.linspace(0.856e9, 1.712e9, nchan).astype(np.float32)     # rounded: 64-128 Hz off the progression


def main():
    out = {}
    cases = []
    for nchan, ncorr, cplx in ((64, 4, False), (64, 4, True), (22, 2, False), (70, 1, True), (6, 4, False), (48, 2, True)):
        for kind in ("linspace", "exact"):
            rng = np.random.default_rng(1300 + nchan + 10 * ncorr + int(cplx))
            lm = ((rng.random((NSRC, 2)) - 0.5) * 0.1).astype(np.float32)
            uvw = ((rng.random((NROW, 3)) - 0.5) * 8e3).astype(np.float32)
            uvw[:, 2] *= np.float32(0.1)
            img = rng.standard_normal((NSRC, nchan, ncorr)).astype(np.float32)
            img[rng.random(img.shape) < 0.15] = 0.0
            if cplx:
                img = (img + 1j * rng.standard_normal(img.shape)).astype(np.complex64)
            fr = band(kind, nchan)
            conv = "casa" if (nchan == 22) else "fourier"
            key = "c%d_%d_%s_%s" % (nchan, ncorr, "c" if cplx else "r", kind)
            vis = im_to_vis(img, uvw, lm, fr, convention=conv)
            assert vis.dtype == np.complex64
            out.update({key + "_img": img, key + "_uvw": uvw, key + "_lm": lm, key + "_freq": fr, key + "_vis": vis})
            cases.append("%s|%s" % (key, conv))
    # the adjoint: complex64 visibilities, float32 coordinates -> float32 image
    rng = np.random.default_rng(1399)
    nrow, nchan, ncorr, nsrc = 300, 24, 4, 12
    lm = ((rng.random((nsrc, 2)) - 0.5) * 0.1).astype(np.float32)
    uvw = ((rng.random((nrow, 3)) - 0.5) * 8e3).astype(np.float32)
    uvw[:, 2] *= np.float32(0.1)
    vis = (rng.standard_normal((nrow, nchan, ncorr)) + 1j * rng.standard_normal((nrow, nchan, ncorr))).astype(np.complex64)
    flags = rng.random((nrow, nchan, ncorr)) < 0.05
    for kind in ("linspace", "exact"):
        fr = band(kind, nchan)
        im = vis_to_im(vis, uvw, lm, fr, flags)
        assert im.dtype == np.float32
        out.update({"v2i_%s_freq" % kind: fr, "v2i_%s_im" % kind: im})
    out.update(v2i_vis=vis, v2i_uvw=uvw, v2i_lm=lm, v2i_flags=flags)
    out["cases"] = np.array(cases)
    path = os.path.join(HERE, "g13_f32.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB), %d im_to_vis cases" % (path, os.path.getsize(path) / 1024.0, len(cases)))


if __name__ == "__main__":
    main()
