#!/usr/bin/env python
"""Extra seeds for the random-shape parity sweeps of tests/test_gpu_fuzz.py (the committed tests run 24 seeds each), plus
calibration, fused-predict, beam / phase_delay and convert / chi^2 sweeps:
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


def fused_gemm_sweep(seed):
    """the GEMM form (Measurement-Set uvw) at random array sizes 2 .. 256 antennas -- every DIAG size, the RECT super-tiles
    with 1 .. 4 column blocks, the flat term stream's tail -- and random row layouts, against the oracle chain"""
    os.environ["AFHIP_GEMM_MIN_FILL"] = "0"
    try:
        import test_gpu_fused_gemm as FG
        rng = np.random.default_rng(seed)
        nant = int(rng.choice([2, 3, 5, 8, 9, 16, 17, 31, 33, 48, 57, 64, 65, 66, 72, 73, 80, 95, 96, 97, 104, 127, 128, 129, 160,
                               192, 197, 200, 224, 255, 256, 257, 300, 384]))
        nbl = nant * (nant - 1) // 2
        nrow = int(rng.integers(1, min(2 * nbl, nbl + 3000) + 2))
        nchan, nsrc = int(rng.integers(1, 5)), int(rng.integers(1, 30))
        d = FG._decomposable(FU._problem(seed, nrow, nchan, nsrc, nant), nant, seed=seed, keep=float(rng.choice([1.0, 0.8])),
                             swap=float(rng.choice([0.0, 0.3])), shuffle=bool(rng.integers(0, 2)), autos=bool(rng.integers(0, 2)))
        if rng.random() < 0.3:          # brightness matrices without any symmetry: the dispatcher must leave the GEMM form
            d["X"] = d["X"] + 0.3 * (rng.standard_normal(d["X"].shape) + 1j * rng.standard_normal(d["X"].shape))
        out = FG._call(d)
        ref = FU._oracle_chain(d, True)
        assert out.shape == ref.shape and (out.size == 0 or np.abs(out - ref).max() < 1e-9 * FU._scale(d)), (nant, nrow, nchan, nsrc)
    finally:
        os.environ.pop("AFHIP_GEMM_MIN_FILL", None)


def fused_gemm_c64_sweep(seed):
    """the single-precision GEMM form (round 6) at random array sizes -- DIAG (3M), RECT 8 x 4 (four products), RECT 8 x 8 (3M on
    block rows) -- and random row layouts, against the oracle's float64 chain on the promoted values"""
    os.environ["AFHIP_GEMM_MIN_FILL"] = "0"
    try:
        import test_gpu_fused_gemm as FG
        import test_gpu_fused_gemm_c64 as FC
        rng = np.random.default_rng(seed)
        nant = int(rng.choice([2, 3, 5, 8, 9, 16, 17, 24, 31, 33, 40, 48, 57, 64, 65, 66, 72, 80, 96, 97, 104, 127, 128, 129, 160, 197, 256]))
        nbl = nant * (nant - 1) // 2
        nrow = int(rng.integers(1, min(2 * nbl, nbl + 3000) + 2))
        nchan, nsrc = int(rng.integers(1, 5)), int(rng.integers(1, 30))
        d = FG._decomposable(FU._problem(seed, nrow, nchan, nsrc, nant), nant, seed=seed, keep=float(rng.choice([1.0, 0.8])),
                             swap=float(rng.choice([0.0, 0.3])), shuffle=bool(rng.integers(0, 2)), autos=bool(rng.integers(0, 2)))
        s = FC._single(d)
        out = FC._call_s(s)
        assert out.dtype == np.complex64, out.dtype
        ref = FC._chain64(s)
        assert out.shape == ref.shape and (out.size == 0 or np.abs(out - ref).max() < FC._tol(s, d) * FU._scale(d)), (
            nant, nrow, nchan, nsrc, np.abs(out - ref).max() / FU._scale(d), FC._tol(s, d))
    finally:
        os.environ.pop("AFHIP_GEMM_MIN_FILL", None)


def fused_rows_c64_sweep(seed):
    """the single-precision lane-per-row form (round 6: af_fused_predict_c64) at random array sizes and batch shapes -- uvw drawn
    per row, rows or groups, optional feed rotation / Gaussian shapes -- against the oracle's float64 chain on the promoted values
    (Gaussian shapes: against the double-precision kernel on the promoted values)"""
    import test_gpu_fused_gemm_c64 as FC
    import test_gpu_fused_rows_c64 as FR
    from codex_africanus_amd.rime import fused
    rng = np.random.default_rng(seed)
    nant = int(rng.choice([2, 3, 5, 7, 8, 16, 17, 31, 33, 40, 63, 64, 65, 70, 100, 128, 129, 150, 200]))
    nbl = nant * (nant - 1) // 2
    nrow = int(rng.integers(1, min(2 * nbl, nbl + 3000) + 2))
    nchan, nsrc = int(rng.integers(1, 5)), int(rng.integers(1, 40))
    d = FU._problem(seed, nrow, nchan, nsrc, nant)
    s = FR._rows_single(d)
    kw = {}
    if rng.integers(0, 2):
        kw["feed_rotation"] = rime.feed_rotation(s["pa"], "linear")
    gauss = bool(rng.integers(0, 3) == 0)
    if gauss:
        gs = np.zeros((nsrc, 3), np.float32)
        k = (nsrc + 1) // 2
        gs[::2] = np.stack([rng.uniform(1e-4, 4e-4, k), rng.uniform(5e-5, 1e-4, k), rng.uniform(0, np.pi, k)], axis=1).astype(np.float32)
        kw["gauss_shape"] = gs
    plan = fused.fused_plan(s["time_index"], s["ant1"], s["ant2"], nant, grouped=bool(rng.integers(0, 2)), uvw=s["uvw"], single=True)
    out = FC._call_s(s, plan=plan, **kw)
    assert out.dtype == np.complex64, out.dtype
    if kw:
        p = lambda a: a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)
        ref = rime.fused_predict_vis(s["time_index"], s["ant1"], s["ant2"], p(s["lm"]), p(s["uvw"]), p(s["frequency"]), p(s["X"]),
                                     p(s["beam"]), p(s["extents"]), p(s["beam_freq_map"]), p(s["pa"]), p(s["pe"]), p(s["as"]),
                                     **{k_: p(v) for k_, v in kw.items()})
    else:
        ref = FC._chain64(s)
    assert out.shape == ref.shape and (out.size == 0 or np.abs(out - ref).max() < FR._tol(s) * FU._scale(d)), (
        nant, nrow, nchan, nsrc, sorted(kw), np.abs(out - ref).max() / FU._scale(d), FR._tol(s))


def beam_and_phase_sweep(seed):
    """beam_cube_dde (arbitrary correlation dims, out-of-band channels, off-cube sources) and phase_delay"""
    rng = np.random.default_rng(seed)
    lw, mh, nud = int(rng.integers(2, 12)), int(rng.integers(2, 12)), int(rng.integers(2, 6))
    corr = [(), (1,), (2,), (4,), (2, 2), (3,)][seed % 6]
    nsrc, ntime, nant, nchan = int(rng.integers(1, 9)), int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.integers(1, 70))
    beam = rng.standard_normal((lw, mh, nud) + corr) + 1j * rng.standard_normal((lw, mh, nud) + corr)
    ext = np.array([[-0.05, 0.06], [-0.04, 0.05]])
    fmap = np.sort(rng.uniform(1.0e9, 1.5e9, nud))
    lm = rng.uniform(-0.08, 0.08, (nsrc, 2))
    pa = rng.uniform(-np.pi, np.pi, (ntime, nant))
    pe = 1e-2 * rng.standard_normal((ntime, nant, nchan, 2))
    asc = 1.0 + 0.1 * rng.standard_normal((nant, nchan, 2))
    freq = np.sort(rng.uniform(0.9e9, 1.6e9, nchan))
    got = rime.beam_cube_dde(beam, ext, fmap, lm, pa, pe, asc, freq)
    ref = oracle.beam_cube_dde(beam, ext, fmap, lm, pa, pe, asc, freq)
    # a random (non-smooth) cube makes corr_sum cancel now and then, and the amplitude normalisation absc / |corr_sum|
    # amplifies the last-bit differences of device and host hypot / division by that cancellation factor
    err, scale = np.abs(got - ref), max(np.abs(ref).max(), 1.0)
    assert got.shape == ref.shape and err.max() <= 1e-10 * scale and (err > 1e-13 * scale).sum() <= max(2, 1e-3 * err.size)
    uvw = rng.standard_normal((int(rng.integers(1, 200)), 3)) * 2000.0
    ph = rime.phase_delay(lm, uvw, freq, convention=("fourier", "casa")[seed % 2])
    assert np.abs(ph - oracle.phase_delay(lm, uvw, freq, convention=("fourier", "casa")[seed % 2])).max() < 1e-14


def convert_and_chi2_sweep(seed):
    import json
    import torch
    from codex_africanus_amd.model.coherency import convert
    from codex_africanus_amd import sharding
    rng = np.random.default_rng(seed)
    cases = json.loads(str(np.load(os.path.join(ROOT, "tests", "golden", "g11_convert.npz"))["cases"]))
    isch, osch, implicit = cases[seed % len(cases)]
    lead = tuple(int(x) for x in rng.integers(0, 9, size=int(rng.integers(0, 4))))
    x = rng.standard_normal(lead + np.asarray(isch).shape)
    if seed % 2:
        x = x + 1j * rng.standard_normal(x.shape)
    if seed % 3 == 0:
        x = x.astype(np.complex64 if seed % 2 else np.float32)
    got, ref = convert(x, isch, osch, implicit_stokes=implicit), oracle.convert(x, isch, osch, implicit)
    assert got.dtype == ref.dtype and got.shape == ref.shape and np.array_equal(got, ref)
    shape = (int(rng.integers(1, 3000)), int(rng.integers(1, 70)), int(rng.choice([1, 2, 4])))
    m = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    d = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    w = rng.random(shape)
    t = lambda a: torch.from_numpy(a).cuda()
    c = sharding.chi2(t(m), t(d), t(w) if seed % 2 else None).cpu().numpy()
    np.testing.assert_allclose(c, ((w if seed % 2 else 1.0) * np.abs(d - m) ** 2).sum(axis=(0, 2)), rtol=1e-12)


def wgridder_sweep(seed):
    """wgridder-shaped model against the direct transform at random small shapes: accuracy contract l2 <= epsilon, all three
    w sign mixes (the w fold), both visibility kernels (gather / tiles through LDS), and the mirror property bit for bit"""
    import test_gpu_wgridder as WG
    from codex_africanus_amd.gridding.wgridder import model
    rng = np.random.default_rng(seed)
    nx, ny = (2 * int(rng.integers(4, 21)) for _ in range(2))
    nchan = int(rng.integers(1, 6))
    nrow = int(rng.choice([int(rng.integers(20, 1500)), 70000 // nchan + int(rng.integers(1, 500))]))
    eps = float(rng.choice([1e-3, 1e-5, 1e-7]))
    cell, freq, uvw, fbi, fbc, image = WG._case(nx, ny, float(rng.uniform(2, 30)) * nx / max(nx, ny), nrow, nchan, 1, seed=seed)
    mode = seed % 3
    if mode:
        uvw[:, 2] = (1.0 if mode == 1 else -1.0) * (np.abs(uvw[:, 2]) + rng.uniform(0, 0.3) * np.abs(uvw[:, 2]).max())
    vis = model(uvw, freq, image, fbi, fbc, cell, epsilon=eps)
    sample = rng.choice(nrow, min(nrow, 300), replace=False)
    ref = WG._explicit_degridder(uvw[sample], freq, image[0], cell, cell)
    err = WG._l2error(vis[sample], ref)
    assert err <= eps, (err, eps, nx, ny, nrow, nchan, mode)
    assert np.array_equal(model(-uvw, freq, image, fbi, fbc, cell, epsilon=eps), np.conj(vis))


first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
only = os.environ.get("AFHIP_STRESS_ONLY")          # e.g. "fused_gemm_sweep,fused_gemm_c64_sweep": only the sweeps named
sweeps = [F.test_im_to_vis_random_shapes, F.test_vis_to_im_random_shapes, F.test_wsclean_predict_random_shapes,
          F.test_predict_vis_random_shapes_bit_exact, F.test_degridder_gridder_random_shapes, calibration_sweep,
          fused_sweep, fused_gemm_sweep, fused_gemm_c64_sweep, fused_rows_c64_sweep, beam_and_phase_sweep, convert_and_chi2_sweep, wgridder_sweep]
if only:
    sweeps = [fn for fn in sweeps if fn.__name__ in only.split(",")]
t0 = time.time()
for seed in range(first, first + count):
    for fn in sweeps:
        try:
            fn(seed)
        except Exception:
            print("FAILED: %s(seed=%d)" % (fn.__name__, seed))
            raise
print("ok: %d seeds x %d sweeps in %.1f s" % (count, len(sweeps), time.time() - t0))
