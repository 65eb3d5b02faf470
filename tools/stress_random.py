#!/usr/bin/env python
"""Extra seeds for the random-shape parity sweeps of tests/test_gpu_fuzz.py (the committed tests run 24 seeds each):
    python tools/stress_random.py [first_seed] [n_seeds]
Every case is checked against the CPU oracle exactly as in the test; stops at the first failure."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F   # noqa: E402

import numpy as np   # noqa: E402
import oracle        # noqa: E402
import test_gpu_fused as FU   # noqa: E402
from codex_africanus_amd import rime   # noqa: E402
from codex_africanus_amd.calibration.utils import corrupt_vis, residual_vis, correct_vis   # noqa: E402


def calibration_sweep(seed):
    """tests/test_gpu_calibration.py's random case with wider extents (whole and partial waves, up to 70 channels)"""
    rng = np.random.default_rng(seed)
    ntime, nant, nchan, ndir = int(rng.integers(1, 5)), int(rng.integers(2, 12)), int(rng.integers(1, 71)), int(rng.integers(1, 5))
    corr, jcorr = [((1,), (1,)), ((2,), (2,)), ((2, 2), (2,)), ((2, 2), (2, 2))][seed % 4]
    a1, a2 = np.triu_indices(nant, 1)
    nbl = a1.shape[0]
    extra = int(rng.integers(0, 3))
    nrow = ntime * nbl + extra
    idx_t = (np.int32, np.int64)[seed % 2]
    tbi = (np.arange(ntime) * nbl + 1000 * (seed % 3)).astype(idx_t)
    tbc = np.full(ntime, nbl).astype(idx_t)
    ant1 = np.concatenate([np.tile(a1, ntime), np.zeros(extra, int)]).astype(idx_t)
    ant2 = np.concatenate([np.tile(a2, ntime), np.ones(extra, int)]).astype(idx_t)
    rc = lambda *sh: rng.standard_normal(sh) + 1j * rng.standard_normal(sh)
    jones = rc(ntime, nant, nchan, ndir, *jcorr) + 1.0
    model = rc(nrow, nchan, ndir, *corr)
    data = rc(nrow, nchan, *corr)
    flag = rng.random(data.shape) < rng.choice([0.0, 0.2, 0.9])
    assert np.array_equal(corrupt_vis(tbi, tbc, ant1, ant2, jones, model), oracle.corrupt_vis(tbi, tbc, ant1, ant2, jones, model))
    assert np.array_equal(residual_vis(tbi, tbc, ant1, ant2, jones, data, flag, model),
                          oracle.residual_vis(tbi, tbc, ant1, ant2, jones, data, flag, model))
    j1 = np.ascontiguousarray(jones[:, :, :, :1])
    assert np.array_equal(correct_vis(tbi, tbc, ant1, ant2, j1, data, flag), oracle.correct_vis(tbi, tbc, ant1, ant2, j1, data, flag))


def fused_sweep(seed):
    """fused predict with beams against the oracle chain: random antenna / source / row / channel counts"""
    rng = np.random.default_rng(seed)
    nant = int(rng.choice([3, 7, 20, 33, 40, 64, 65, 100, 129]))
    nbl = nant * (nant - 1) // 2
    nrow = int(rng.integers(1, 3 * nbl + 2))
    nchan, nsrc = int(rng.integers(1, 6)), int(rng.integers(1, 40))
    d = FU._problem(seed, nrow, nchan, nsrc, nant)
    out = rime.fused_predict_vis(d["time_index"], d["ant1"], d["ant2"], d["lm"], d["uvw"], d["frequency"],
                                 d["X"], d["beam"], d["extents"], d["beam_freq_map"], d["pa"], d["pe"], d["as"])
    ref = FU._oracle_chain(d, True)
    assert np.abs(out - ref).max() < 1e-9 * FU._scale(d), (nant, nrow, nchan, nsrc)


first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sweeps = [F.test_im_to_vis_random_shapes, F.test_vis_to_im_random_shapes, F.test_wsclean_predict_random_shapes,
          F.test_predict_vis_random_shapes_bit_exact, F.test_degridder_gridder_random_shapes, calibration_sweep,
          fused_sweep]
t0 = time.time()
for seed in range(first, first + count):
    for fn in sweeps:
        try:
            fn(seed)
        except Exception:
            print("FAILED: %s(seed=%d)" % (fn.__name__, seed))
            raise
print("ok: %d seeds x %d sweeps in %.1f s" % (count, len(sweeps), time.time() - t0))
