#!/opt/conda/bin/python3.9
"""
G16 -- the predict with beam-cube DDEs on a Measurement Set's uvw (uvw_pq = uvw_p - uvw_q per timestep), by the
REFERENCE's own functions (africanus/rime/examples/predict.py:404-525; conda python 3.9: numba 0.54 +
tests/golden/ref_shim.py):

    NUMBA_CACHE_DIR=/tmp/numba_cache PYTHONPATH=/root/reference:tests/golden:. \
        /opt/conda/bin/python3.9 tests/golden/make_golden_gemm.py

    phase = africanus.rime.phase_delay(lm, uvw, frequency)
    coh   = np.einsum("srf,sfij->srfij", phase, brightness)
    dde   = africanus.rime.beam_cube_dde(beam, extents, freq_map, lm, parangles, point_errors, scaling, frequency)
    [dde  = np.einsum("stafij,tajk->stafik", dde, africanus.rime.feed_rotation(parangles, "linear"))]
    vis   = africanus.rime.predict_vis(time_index, antenna1, antenna2, dde, coh, dde, [die, base_vis, die])

on G14's sky, beam and per-antenna terms (tests/golden/g14_fused_dask.npz) with the rows' uvw replaced by differences of
per-(timestep, antenna) coordinates: the rows on which the build's ``rime.fused_predict_vis`` takes its GEMM form
(csrc/af_fused_gemm.hip; tests/test_gpu_fused_gemm.py).  Stores uvw, the antenna coordinates and the results in
g16_fused_gemm.npz (the other inputs are G14's).
"""
import os

import ref_shim  # noqa: F401  (must come first)
import numpy as np

from africanus.rime import phase_delay, predict_vis, beam_cube_dde, feed_rotation

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = np.load(os.path.join(HERE, "g14_fused_dask.npz"))
    ti, a1, a2 = g["time_index"], g["antenna1"], g["antenna2"]
    ntime, nant = g["parallactic_angles"].shape
    rng = np.random.default_rng(16)
    xyz = rng.uniform(-1, 1, (ntime, nant, 3)) * np.array([3000.0, 3000.0, 300.0])
    uvw = xyz[ti, a1] - xyz[ti, a2]
    out = dict(uvw=uvw, ant_xyz=xyz)
    phase = phase_delay(g["lm"], uvw, g["frequency"])
    coh = np.einsum("srf,sfij->srfij", phase, g["brightness"])
    dde = beam_cube_dde(g["beam"], g["beam_lm_extents"], g["beam_freq_map"], g["lm"], g["parallactic_angles"],
                        g["point_errors"], g["antenna_scaling"], g["frequency"])
    out["vis_beam"] = predict_vis(ti, a1, a2, dde, coh, dde, None, None, None)
    out["vis_beam_die"] = predict_vis(ti, a1, a2, dde, coh, dde, g["die"], g["base_vis"], g["die"])
    fr = feed_rotation(g["parallactic_angles"], "linear")
    ddef = np.einsum("stafij,tajk->stafik", dde, fr)
    out["vis_beam_feed"] = predict_vis(ti, a1, a2, ddef, coh, ddef, None, None, None)
    # per-visibility sum of |term| magnitudes (the scale the 1e-9 tolerance refers to)
    out["scale"] = np.abs(g["brightness"]).sum(axis=(0, 2, 3)).max() * float(np.abs(g["beam"]).max()) ** 2
    np.savez_compressed(os.path.join(HERE, "g16_fused_gemm.npz"), **out)
    print("wrote g16_fused_gemm.npz:", {k: v.shape for k, v in out.items() if k.startswith("vis_")})


if __name__ == "__main__":
    main()
