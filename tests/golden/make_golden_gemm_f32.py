#!/opt/conda/bin/python3.9
"""
G17 -- the predict with beam-cube DDEs on a Measurement Set's uvw in SINGLE PRECISION, by the REFERENCE's own functions
(conda python 3.9: numba 0.54 + tests/golden/ref_shim.py; run AFTER make_golden_fused_dask.py and make_golden_gemm.py,
whose g14 / g16 files it reads):

    NUMBA_CACHE_DIR=/tmp/numba_cache PYTHONPATH=/root/reference:tests/golden:. \
        /opt/conda/bin/python3.9 tests/golden/make_golden_gemm_f32.py

Every input float32 / complex64, so the reference runs its whole chain in single precision
(africanus/util/type_inference.py:24-26):

    phase = africanus.rime.phase_delay(lm, uvw, frequency)                       float32 phases -> complex64
    coh   = np.einsum("srf,sfij->srfij", phase, brightness)
    dde   = africanus.rime.beam_cube_dde(beam, extents, freq_map, lm, parangles, point_errors, scaling, frequency)
    [dde  = np.einsum("stafij,tajk->stafik", dde, africanus.rime.feed_rotation(parangles, "linear"))]
    vis   = africanus.rime.predict_vis(time_index, antenna1, antenna2, dde, coh, dde)          complex64

and, as the yardstick, the SAME chain on the same values promoted to float64 (``*_vis64``).  Two cases: "a" = G14's sky,
beam and per-antenna terms with G16's antenna coordinates (5 antennas), "b" = 12 antennas, 3 timesteps, 40 sources, 8
channels, a 17 x 17 x 5 cube, 3 km baselines (where the reference's float32 phases are off by ~1e-3 rad).  uvw are
differences of float32 antenna coordinates, rounded to float32 as a Measurement Set's would be.  What
tests/test_gpu_fused_gemm_c64.py requires of af_fused_predict_antennas_c64: closer to ``*_vis64`` than ``*_vis32`` is.
"""
import os

import ref_shim  # noqa: F401  (must come first)
import numpy as np

from africanus.rime import phase_delay, predict_vis, beam_cube_dde, feed_rotation

HERE = os.path.dirname(os.path.abspath(__file__))
F, C = np.float32, np.complex64


def chain(d, real, cplx, feed):
    lm, uvw, fr = d["lm"].astype(real), d["uvw"].astype(real), d["frequency"].astype(real)
    phase = phase_delay(lm, uvw, fr)
    coh = np.einsum("srf,sfij->srfij", phase, d["brightness"].astype(cplx))
    dde = beam_cube_dde(d["beam"].astype(cplx), d["beam_lm_extents"].astype(real), d["beam_freq_map"].astype(real), lm,
                        d["parallactic_angles"].astype(real), d["point_errors"].astype(real),
                        d["antenna_scaling"].astype(real), fr)
    if feed:
        dde = np.einsum("stafij,tajk->stafik", dde, feed_rotation(d["parallactic_angles"].astype(real), "linear"))
    vis = predict_vis(d["time_index"], d["antenna1"], d["antenna2"], dde, coh, dde, None, None, None)
    assert vis.dtype == cplx, vis.dtype
    return vis


def case_a():
    g, g16 = np.load(os.path.join(HERE, "g14_fused_dask.npz")), np.load(os.path.join(HERE, "g16_fused_gemm.npz"))
    d = {k: g[k] for k in ("time_index", "antenna1", "antenna2", "lm", "frequency", "brightness", "beam", "beam_lm_extents",
                           "beam_freq_map", "parallactic_angles", "point_errors", "antenna_scaling")}
    xyz = g16["ant_xyz"].astype(F)
    d["ant_xyz"] = xyz
    d["uvw"] = xyz[d["time_index"], d["antenna1"]] - xyz[d["time_index"], d["antenna2"]]      # float32 arithmetic
    return d


def case_b():
    rng = np.random.default_rng(17)
    nant, ntime, nsrc, nchan, lw, mh, nud = 12, 3, 40, 8, 17, 17, 5
    a1, a2 = np.triu_indices(nant, 1)
    nbl = a1.shape[0]
    d = dict(time_index=np.repeat(np.arange(ntime), nbl).astype(np.int32), antenna1=np.tile(a1, ntime).astype(np.int32),
             antenna2=np.tile(a2, ntime).astype(np.int32))
    xyz = (rng.uniform(-1, 1, (ntime, nant, 3)) * np.array([3000.0, 3000.0, 300.0])).astype(F)
    d["ant_xyz"] = xyz
    d["uvw"] = xyz[d["time_index"], d["antenna1"]] - xyz[d["time_index"], d["antenna2"]]
    d["lm"] = ((rng.random((nsrc, 2)) - 0.5) * 0.08).astype(F)
    d["frequency"] = np.linspace(0.9e9, 1.6e9, nchan).astype(F)
    stokes = rng.random((nsrc, 4)) * np.array([1.0, 0.1, 0.1, 0.05])
    I, Q, U, V = stokes.T
    X = np.stack([I + Q, U + 1j * V, U - 1j * V, I - Q], -1).reshape(nsrc, 1, 2, 2)
    spec = (d["frequency"][None, :].astype(np.float64) / 1.2e9) ** (-0.7 * rng.random((nsrc, 1)))
    d["brightness"] = (X * spec[:, :, None, None]).astype(C)
    gl = np.linspace(-1, 1, lw)
    ll, mm = np.meshgrid(gl, gl, indexing="ij")
    pattern = np.exp(-(ll ** 2 + mm ** 2) / 0.6) * np.exp(1j * (0.4 * ll - 0.3 * mm))
    gains = (1 + 0.05 * np.arange(nud))[:, None] * np.array([1.0, 0.06j, -0.05j, 0.93])[None, :]
    d["beam"] = (pattern[:, :, None, None] * gains[None, None]).reshape(lw, mh, nud, 2, 2).astype(C)
    d["beam_lm_extents"] = np.array([[-0.05, 0.05], [-0.05, 0.05]], dtype=F)
    d["beam_freq_map"] = np.linspace(0.85e9, 1.65e9, nud).astype(F)
    d["parallactic_angles"] = rng.uniform(0, np.pi / 5, (ntime, nant)).astype(F)
    d["point_errors"] = (2e-3 * rng.standard_normal((ntime, nant, nchan, 2))).astype(F)
    d["antenna_scaling"] = (1 + 1e-2 * rng.standard_normal((nant, nchan, 2))).astype(F)
    return d


def main():
    out = {}
    for name, d in (("a", case_a()), ("b", case_b())):
        for k, v in d.items():
            if k in ("time_index", "antenna1", "antenna2"):
                out["%s_%s" % (name, k)] = v
            else:
                out["%s_%s" % (name, k)] = v.astype(C if np.iscomplexobj(v) else F)
        d32 = {k: out["%s_%s" % (name, k)] for k in d}
        for feed in (False, True):
            tag = "_feed" if feed else ""
            out["%s_vis32%s" % (name, tag)] = chain(d32, F, C, feed)
            out["%s_vis64%s" % (name, tag)] = chain(d32, np.float64, np.complex128, feed)
        e = np.abs(out[name + "_vis32"] - out[name + "_vis64"]).max() / np.abs(out[name + "_vis64"]).max()
        print("case %s: %s, reference float32 chain vs its float64 chain: %.2e of the peak" % (name, out[name + "_vis32"].shape, e))
    np.savez_compressed(os.path.join(HERE, "g17_fused_gemm_f32.npz"), **out)


if __name__ == "__main__":
    main()
