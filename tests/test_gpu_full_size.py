"""
BASELINE configs at FULL size on one MI355X, through the C ABI, on exactly the inputs `bench.py --workload ...`
builds (the tests instantiate bench.py's workload objects): rows sampled against the CPU oracle plus the
size-independent properties the domain offers -- linearity, row-shard invariance (a shard's rows equal the same
rows of the full call bit for bit: what the multi-GPU sharding relies on), finite checksums.

  configs[1]  im_to_vis 1e6 x 64 x 1000 x 4: tests/test_gpu_parity.py::test_im_to_vis_full_size_c2_properties
  configs[1'] the same with complex brightness (= fused predict without DDEs)           -- here
  configs[2]  fused predict with beam-cube DDEs, 64 antennas, 257 x 257 x 33 cube       -- here
  configs[4]  degridding of a 4096^2 grid, 1e6 rows x 64 chan, 7 x 7 taps               -- here
(configs[3] is configs[1] row-sharded over 8 GPUs: world-2 gloo tests in tests/test_chunked_and_sharding.py.)
"""
import argparse
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _workload(name, **over):
    import torch
    import bench
    from codex_africanus_amd import _lib
    args = argparse.Namespace(gpus=1, steps=1, warmup=0, rows=1000000, chans=64, sources=1000, seed=0, mode="auto",
                              workload=name, pa="random", npix=4096, backend="nccl", no_cpu_baseline=True,
                              cpu_seconds=1.0, check_rows=0)
    for k, v in over.items():
        setattr(args, k, v)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    wl = bench.WORKLOADS[name](args, 0, dev, _lib.load(), _lib, t)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    vis = torch.empty((args.rows, args.chans, wl.ncorr), dtype=torch.complex128, device=dev)
    wl.predict(vis, stream, P)
    torch.cuda.synchronize()
    return wl, vis, args


def _sample(vis, rows):
    import torch
    return vis[torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(vis.device)].cpu().numpy()


def test_fused_dde_full_size_c3():
    """BASELINE configs[2]: 1e6 rows x 64 chan x 1000 src, 64 antennas, 257 x 257 x 33 x 2 x 2 cube.  Rows of the
    first, a middle and the last (short) timestep against the reference chain phase_delay -> einsum ->
    beam_cube_dde -> predict_vis (oracle), north-star tolerance 1e-8 absolute; a timestep-aligned and an
    unaligned row shard equal the full call bit for bit; x2 brightness is exact."""
    import torch
    from codex_africanus_amd import rime
    wl, vis, args = _workload("fused_dde")
    h = wl.h
    nrow, nbl = args.rows, wl.nbl
    assert tuple(vis.shape) == (nrow, 64, 4)
    rows = np.concatenate([np.arange(0, nbl, 211), 250 * nbl + np.arange(5, nbl, 199),
                           np.arange((wl.ntime - 1) * nbl, nrow, 7)[:12]])
    ref, rows = wl.reference_rows(rows)           # the oracle chain on (up to) 32 of them
    got = _sample(vis, rows)
    err = np.abs(got - ref.reshape(got.shape)).max()
    assert err < 1e-8, err
    assert err < 1e-9 * np.abs(h["X"]).sum(axis=0).max()      # and the relative bound of tests/test_gpu_fused.py
    # row shards through the reference-shaped entry point (device tensors): 37 timesteps from timestep 100, and
    # a shard that starts and ends inside timesteps
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dv = wl.dv
    for a, b in ((100 * nbl, 137 * nbl), (123457, 234567)):
        t0, t1 = int(h["time_index"][a]), int(h["time_index"][b - 1]) + 1
        part = rime.fused_predict_vis(T(h["time_index"][a:b]), dv["a1"][a:b], dv["a2"][a:b], dv["lm"], dv["uvw"][a:b],
                                      dv["freq"], dv["X"], dv["beam"], dv["ext"], dv["fmap"], dv["pa"][t0:t1],
                                      dv["pe"][t0:t1], dv["asc"])
        assert torch.equal(part.reshape(b - a, 64, 4), vis[a:b])
    del part
    # linearity in the brightness: a power of two commutes with every rounding
    a, b = 400 * nbl, 420 * nbl
    t0, t1 = 400, 420
    twice = rime.fused_predict_vis(T(h["time_index"][a:b]), dv["a1"][a:b], dv["a2"][a:b], dv["lm"], dv["uvw"][a:b],
                                   dv["freq"], dv["X"] * 2.0, dv["beam"], dv["ext"], dv["fmap"], dv["pa"][t0:t1],
                                   dv["pe"][t0:t1], dv["asc"])
    assert torch.equal(twice.reshape(b - a, 64, 4), vis[a:b] * 2.0)
    power = (vis.real ** 2 + vis.imag ** 2).sum(dim=(0, 2))
    assert bool(torch.isfinite(power).all()) and bool((power > 0).all())


def test_fused_dde_full_size_c3_antenna_decomposable():
    """BASELINE configs[2] with a Measurement Set's uvw (differences of per-antenna coordinates): the GEMM form on the
    matrix cores (csrc/af_fused_gemm.hip).  Same checks: rows of three timesteps against the oracle chain < 1e-8 absolute
    and < 1e-9 relative; a timestep-aligned row shard through the reference-shaped entry point (which plans, decomposes
    and dispatches for itself) equals the full call bit for bit; x2 brightness exact; and the general kernel on the same
    rows agrees to 1e-10 relative."""
    import os
    import torch
    from codex_africanus_amd import rime
    from codex_africanus_amd.rime import fused
    wl, vis, args = _workload("fused_dde_ant")
    h = wl.h
    nrow, nbl = args.rows, wl.nbl
    assert wl.antennas and wl.plan_residual < 1e-10
    rows = np.concatenate([np.arange(0, nbl, 211), 250 * nbl + np.arange(5, nbl, 199),
                           np.arange((wl.ntime - 1) * nbl, nrow, 7)[:12]])
    ref, rows = wl.reference_rows(rows)
    got = _sample(vis, rows)
    err = np.abs(got - ref.reshape(got.shape)).max()
    scale = np.abs(h["X"]).sum(axis=0).max()
    assert err < 1e-8 and err < 1e-9 * scale, err
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dv = wl.dv
    a, b = 100 * nbl, 137 * nbl
    call = lambda X, **kw: rime.fused_predict_vis(T(h["time_index"][a:b]), dv["a1"][a:b], dv["a2"][a:b], dv["lm"],
                                                  dv["uvw"][a:b], dv["freq"], X, dv["beam"], dv["ext"], dv["fmap"],
                                                  dv["pa"][100:137], dv["pe"][100:137], dv["asc"], **kw)
    plan = fused.fused_plan(h["time_index"][a:b], h["ant1"][a:b], h["ant2"][a:b], 64, uvw=h["uvw"][a:b])
    assert plan.decomposable
    part = call(dv["X"], plan=plan)
    # (the shard's antenna coordinates are solved from the shard's rows: the same per-timestep solution as the full plan's)
    assert torch.equal(part.reshape(b - a, 64, 4), vis[a:b])
    assert torch.equal(call(dv["X"] * 2.0, plan=plan).reshape(b - a, 64, 4), vis[a:b] * 2.0)
    general = fused.fused_plan(h["time_index"][a:b], h["ant1"][a:b], h["ant2"][a:b], 64)       # no uvw: lane-per-row kernel
    assert not general.decomposable
    other = call(dv["X"], plan=general).reshape(b - a, 64, 4)
    assert float((other - vis[a:b]).abs().max()) < 1e-10 * scale


def test_degrid_full_size_c5():
    """BASELINE configs[4]: 4096^2 grid, 1e6 rows x 64 chan, 7 x 7 taps.  Sampled rows against the oracle degridder
    (to rounding: the sums run column-first), x2 grid exact, a row shard equals the full call bit for bit (the
    uv-tile ordering of the rows is transparent), XX == YY for Stokes I."""
    import torch
    from codex_africanus_amd.gridding.perleypolyhedron.degridder import degridder
    wl, vis, args = _workload("degrid")
    nrow = args.rows
    assert tuple(vis.shape) == (nrow, 64, 2)
    rows = np.linspace(0, nrow - 1, 300).astype(np.int64)
    ref, _ = wl.reference_rows(rows)
    got = _sample(vis, rows)
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    assert torch.equal(vis[:, :, 0], vis[:, :, 1])
    dv = wl.dv
    pol = ("None", "None", "XXYY_FROM_I", "conv_1d_axisymmetric_packed_gather")
    a, b = 123457, 654321
    part = degridder(dv["uvw"][a:b], wl.d_grid, dv["wl"], dv["cm"], wl.CELL, (0.0, 0.0), (0.0, 0.0), dv["k"], wl.W,
                     wl.OS, *pol)
    assert torch.equal(part, vis[a:b])
    twice = degridder(dv["uvw"][a:b], wl.d_grid * 2.0, dv["wl"], dv["cm"], wl.CELL, (0.0, 0.0), (0.0, 0.0), dv["k"],
                      wl.W, wl.OS, *pol)
    # the normalisation divides by (sum of weights + 1e-8): a factor 2 of the grid still commutes with it exactly
    assert torch.equal(twice, part * 2.0)
    assert bool(torch.isfinite(vis.real).all()) and bool(torch.isfinite(vis.imag).all())


def test_im_to_vis_complex_full_size():
    """configs[1] with complex brightness matrices (what the fused predict without DDEs runs): sampled rows vs the
    oracle < 1e-8, shard invariance, exact linearity."""
    import torch
    from codex_africanus_amd import dft
    wl, vis, args = _workload("dft_complex")
    nrow = args.rows
    rows = np.linspace(0, nrow - 1, 48).astype(np.int64)
    ref, _ = wl.reference_rows(rows)
    assert np.abs(_sample(vis, rows) - ref).max() < 1e-8
    a, b = 123457, 654321
    part = dft.im_to_vis(wl.d_image, wl.d_uvw[a:b], wl.d_lm, wl.d_freq)
    assert torch.equal(part, vis[a:b])
    assert torch.equal(dft.im_to_vis(wl.d_image * 4.0, wl.d_uvw[a:b], wl.d_lm, wl.d_freq), part * 4.0)


def test_gaussian_sources_full_size():
    """configs[1]'s counts with Gaussian + point sources and no DDEs (bench workload `gauss`: the MFMA-accumulator form with
    the envelope in the phasor): sampled rows vs the oracle chain phase_delay x gaussian_shape x brightness < 1e-8, a row
    shard equals the same rows of the full call bit for bit, x4 brightness is exact, no envelope -> the complex-image
    transform of the same sources to rounding."""
    import torch
    from codex_africanus_amd import rime
    wl, vis, args = _workload("gauss")
    nrow = args.rows
    rows = np.linspace(0, nrow - 1, 48).astype(np.int64)
    ref, _ = wl.reference_rows(rows)
    assert np.abs(_sample(vis, rows) - ref).max() < 1e-8
    lm, uvw, freq, X, shapes = wl.dv
    a, b = 123457, 654321
    idx = torch.zeros(b - a, dtype=torch.int32, device=uvw.device)
    call = lambda Xd, sh: rime.fused_predict_vis(idx, idx, idx, lm, uvw[a:b], freq, Xd.reshape(args.sources, 64, 2, 2),
                                                 gauss_shape=sh).reshape(b - a, 64, 4)
    part = call(X, shapes)
    assert torch.equal(part, vis[a:b])
    assert torch.equal(call(X * 4.0, shapes), part * 4.0)
    points = call(X, torch.zeros_like(shapes))
    assert bool(torch.isfinite(points.real).all())
    wl.shapes = np.zeros_like(wl.shapes)      # the oracle chain with point sources only, on a few rows
    assert np.abs(points[:16].cpu().numpy() - wl._chain(wl.uvw[a:a + 16])).max() < 1e-8


def test_wgridder_full_size_c5():
    """configs[4] through the wgridder-shaped entry: 4096^2 image, 1e6 rows x 64 channels, epsilon 1e-5 with
    w-stacking.  Sampled rows against the direct transform of the image's non-zero pixels (CPU oracle) within epsilon;
    exact linearity in the image; every visibility finite."""
    import torch
    import oracle
    from codex_africanus_amd.gridding.wgridder import model
    dev = torch.device("cuda", 0)
    npix, nrow, nchan, eps = 4096, 1000000, 64, 1e-5
    cell = np.deg2rad(2.0 / 3600.0)
    freq = np.linspace(0.856e9, 1.712e9, nchan)
    rng = np.random.default_rng(0)
    umax = 0.45 / cell * (299792458.0 / freq.max())
    uvw = np.zeros((nrow, 3))
    uvw[:, :2] = rng.uniform(-1, 1, (nrow, 2)) * umax
    uvw[:, 2] = rng.uniform(-400, 400, nrow)
    image = np.zeros((1, npix, npix))
    nz = rng.integers(0, npix, (3000, 2))
    image[0, nz[:, 0], nz[:, 1]] = rng.lognormal(0, 1, 3000)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_uvw, d_freq, d_img = t(uvw), t(freq), t(image)
    vis = model(d_uvw, d_freq, d_img, np.array([0]), np.array([nchan]), cell, epsilon=eps)
    assert tuple(vis.shape) == (nrow, nchan) and vis.dtype == torch.complex128
    assert bool(torch.isfinite(vis.real).all()) and bool(torch.isfinite(vis.imag).all())
    rows = np.linspace(0, nrow - 1, 96).astype(np.int64)
    ix, iy = np.nonzero(image[0])
    x, y = (ix - npix / 2) * cell, (iy - npix / 2) * cell
    n = np.sqrt(1 - x * x - y * y)
    src = np.broadcast_to((image[0, ix, iy] / n)[:, None, None], (ix.size, nchan, 1)).copy()
    ref = oracle.im_to_vis(src, uvw[rows] * np.array([1, 1, -1.0]), np.stack([x, y], 1), freq, omp=True)[:, :, 0]
    got = _sample(vis, rows)
    assert np.sqrt(np.sum(np.abs(got - ref) ** 2) / np.sum(np.abs(ref) ** 2)) <= eps
    twice = model(d_uvw, d_freq, d_img * 2.0, np.array([0]), np.array([nchan]), cell, epsilon=eps)
    assert torch.equal(twice, vis * 2.0)
    # the own row transforms (4096-cell rows: four radix-8 passes) against the hipFFT route on the same call
    import os
    os.environ["AFHIP_WGRID_FFT1"] = os.environ["AFHIP_WGRID_FFT2"] = "0"
    try:
        lib = model(d_uvw, d_freq, d_img, np.array([0]), np.array([nchan]), cell, epsilon=eps)
    finally:
        del os.environ["AFHIP_WGRID_FFT1"], os.environ["AFHIP_WGRID_FFT2"]
    assert not torch.equal(lib, vis)
    assert float((lib - vis).abs().max()) <= 1e-12 * float(vis.abs().max())


def test_wgridder_dirty_full_size_c5():
    """The adjoint at configs[4]'s counts: 1e6 rows x 64 channels -> 4096^2 dirty image, epsilon 1e-5.  A few pixels
    against the direct sum over ALL visibilities (CPU oracle), and <R x, v> == <x, R^H v> with the full-size model of a
    sparse image -- to rounding, because the two directions share planes and taps."""
    import torch
    import oracle
    from codex_africanus_amd.gridding.wgridder import dirty, model
    dev = torch.device("cuda", 0)
    npix, nrow, nchan, eps = 4096, 1000000, 64, 1e-5
    cell = np.deg2rad(2.0 / 3600.0)
    freq = np.linspace(0.856e9, 1.712e9, nchan)
    rng = np.random.default_rng(1)
    umax = 0.45 / cell * (299792458.0 / freq.max())
    uvw = np.zeros((nrow, 3))
    uvw[:, :2] = rng.uniform(-1, 1, (nrow, 2)) * umax
    uvw[:, 2] = rng.uniform(-400, 400, nrow)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_uvw, d_freq = t(uvw), t(freq)
    gen = torch.Generator(device=dev).manual_seed(7)
    d_vis = torch.randn(nrow, nchan, dtype=torch.complex128, device=dev, generator=gen)
    d_wgt = torch.rand(nrow, nchan, dtype=torch.float64, device=dev, generator=gen)
    fbi, fbc = np.array([0]), np.array([nchan])
    img = dirty(d_uvw, d_freq, d_vis, fbi, fbc, npix, npix, cell, weights=d_wgt, epsilon=eps)
    assert tuple(img.shape) == (1, npix, npix) and img.dtype == torch.float64
    assert bool(torch.isfinite(img).all())
    pix = rng.integers(0, npix, (5, 2))
    x, y = (pix[:, 0] - npix / 2) * cell, (pix[:, 1] - npix / 2) * cell
    n = np.sqrt(1 - x * x - y * y)
    wv = (d_vis * d_wgt).cpu().numpy()
    ref = oracle.vis_to_im(wv[:, :, None], uvw * np.array([1, 1, -1.0]), np.stack([x, y], 1), freq,
                           np.zeros(wv.shape + (1,), np.uint8), omp=True)[:, :, 0].sum(axis=1) / n
    got = img[0].cpu().numpy()[pix[:, 0], pix[:, 1]]
    rms = float(torch.sqrt(torch.mean(img[0] ** 2)))
    assert np.abs(got - ref).max() <= eps * rms * 10          # per pixel; the contract is an l2 one over the image
    image = np.zeros((1, npix, npix))
    nz = rng.integers(0, npix, (3000, 2))
    image[0, nz[:, 0], nz[:, 1]] = rng.lognormal(0, 1, 3000)
    d_img = t(image)
    vis = model(d_uvw, d_freq, d_img, fbi, fbc, cell, weights=d_wgt, epsilon=eps)
    lhs = float(torch.sum(d_vis.real * vis.real + d_vis.imag * vis.imag))
    rhs = float(torch.sum(d_img * img))
    assert abs(lhs - rhs) <= 1e-10 * float(torch.sum(d_img) * img.abs().max())


def test_bench_wgrid_workload_meets_the_contract():
    """`bench.py --workload wgrid` (configs[4] as named) calls af_wgrid_im2vis_f64 directly with its own set-up: at a
    reduced size its output meets the accuracy contract against the direct transform, the checker the bench itself uses."""
    wl, vis, args = _workload("wgrid", rows=30000, chans=16, npix=512)
    rows = np.linspace(0, args.rows - 1, 200).astype(np.int64)
    ref, _ = wl.reference_rows(rows)
    got = _sample(vis, rows)
    assert np.sqrt(np.sum(np.abs(got - ref) ** 2) / np.sum(np.abs(ref) ** 2)) <= wl.EPS
    r = wl.roofline(1e-3)
    assert r["kernel"] == "wg_degrid_tiles<7>" and r["alg_bytes"] > 0
