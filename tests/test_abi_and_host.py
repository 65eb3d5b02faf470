"""
CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/afhip.h declares, and the host wrappers reproduce the reference's
argument checking / dtype rules (africanus/rime/predict.py:380-463,542-563).
No compute entry point is called here.
"""
import os
import re

import numpy as np
import pytest

from codex_africanus_amd import _lib
from codex_africanus_amd.rime import predict as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    _lib.build()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "afhip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(af_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.af_version() >= 100
    assert lib.af_last_error() == b""


def test_workspace_queries_are_pure():
    lib = _lib.load()
    assert lib.af_predict_vis_workspace_bytes() >= 8
    small = lib.af_im_to_vis_workspace_bytes(10, 16, 4, 0)
    big = lib.af_im_to_vis_workspace_bytes(1000, 64, 4, 0)
    assert 0 < small < big
    # packed image dominates: nsrc * padded chans * ncorr * 8 bytes
    assert big >= 1000 * 64 * 4 * 8
    # complex pixels double the records (2 correlations: no MFMA-path records in either)
    assert lib.af_im_to_vis_workspace_bytes(1000, 64, 2, 1) > lib.af_im_to_vis_workspace_bytes(1000, 64, 2, 0)


def test_no_oracle_import_in_product():
    """The product must never route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "codex_africanus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f


# ---- predict_checks: same conditions / messages as the reference ---------------------
def z(*shape):
    return np.zeros(shape, dtype=np.complex128)


IDX = (np.zeros(10, np.int32),) * 3


def test_predict_checks_presence_tuple():
    tup = P.predict_checks(*IDX, z(3, 2, 4, 5, 2, 2), z(3, 10, 5, 2, 2), z(3, 2, 4, 5, 2, 2), None, None, None)
    assert tup == (True, True, True, False, False, False)


@pytest.mark.parametrize("kwargs, msg", [
    (dict(dde1_jones=z(3, 2, 4, 5, 2)), "Both dde1_jones and dde2_jones must be present or absent"),
    (dict(die2_jones=z(2, 4, 5, 2)), "Both die1_jones and die2_jones must be present or absent"),
    (dict(dde1_jones=z(3, 2, 4, 5), dde2_jones=z(3, 2, 4, 5)), r"dde1_jones.ndim 4 not in \(5, 6\)"),
    (dict(dde1_jones=z(3, 2, 4, 5, 2), dde2_jones=z(3, 2, 4, 5, 2, 2)), "dde1_jones.ndim != dde2_jones.ndim"),
    (dict(source_coh=z(3, 10, 5)), r"source_coh.ndim 3 not in \(4, 5\)"),
    (dict(base_vis=z(10, 5)), r"base_vis.ndim 2 not in \(3, 4\)"),
    (dict(die1_jones=z(2, 4, 5), die2_jones=z(2, 4, 5)), r"die1_jones.ndim 3 not in \(4, 5\)"),
    (dict(die1_jones=z(2, 4, 5, 2), die2_jones=z(2, 4, 5, 2, 2)), "die1_jones.ndim != die2_jones.ndim"),
    (dict(source_coh=z(3, 10, 5, 2), base_vis=z(10, 5, 2, 2)), "One of the following pre-conditions is broken"),
    (dict(dde1_jones=z(3, 2, 4, 5, 2, 2), dde2_jones=z(3, 2, 4, 5, 2, 2), source_coh=z(3, 10, 5, 2)),
     "One of the following pre-conditions is broken"),
])
def test_predict_vis_value_errors(kwargs, msg):
    with pytest.raises(ValueError, match=msg):
        P.predict_vis(*IDX, **kwargs)


def test_predict_vis_no_inputs():
    with pytest.raises(ValueError, match="No Jones Matrices were supplied"):
        P.predict_vis(*IDX)


def test_predict_vis_shape_mismatch_is_refused():
    with pytest.raises(ValueError, match="source_coh has shape"):
        P.predict_vis(*IDX, source_coh=z(3, 9, 5, 2, 2))
    with pytest.raises(ValueError, match="correlation shape"):
        P.predict_vis(*IDX, source_coh=z(3, 10, 5, 3))


def test_convention_errors_raise_before_any_device_work():
    from codex_africanus_amd.rime import phase_delay
    from codex_africanus_amd.dft import im_to_vis
    lm, uvw, fr = np.zeros((3, 2)), np.zeros((5, 3)), np.ones(4)
    with pytest.raises(ValueError, match=r"convention not in \('fourier', 'casa'\)"):
        phase_delay(lm, uvw, fr, convention="bob")
    with pytest.raises(ValueError, match=r"convention not in \('fourier', 'casa'\)"):
        im_to_vis(np.zeros((3, 4, 2)), uvw, lm, fr, convention="bob")
    from codex_africanus_amd.rime import beam_cube_dde
    with pytest.raises(ValueError, match="beam_lw, beam_mh and beam_nud must be >= 2"):
        beam_cube_dde(z(1, 2, 2, 1), np.zeros((2, 2)), np.zeros(2), lm, np.zeros((1, 1)),
                      np.zeros((1, 1, 4, 2)), np.ones((1, 4, 2)), fr)


def test_dft_mode_switch():
    from codex_africanus_amd.dft import kernels
    assert kernels.get_mode() in ("auto", "exact", "recurrence")
    old = kernels.get_mode()
    kernels.set_mode("exact")
    assert kernels.get_mode() == "exact"
    with pytest.raises(ValueError):
        kernels.set_mode("fast")
    kernels.set_mode(old)


def test_constants_match_reference_bits():
    from codex_africanus_amd import constants
    import math
    assert constants.c == 2.99792458e8
    assert constants.two_pi_over_c == 2 * math.pi / 2.99792458e8
    assert constants.minus_two_pi_over_c == -constants.two_pi_over_c


def test_convert_schema_errors_match_reference(g11):
    """convert's schema resolution is host logic: the same exception class and message as the reference recorded
    for each malformed call (tests/golden/make_golden_convert.py), raised before any device work."""
    import json
    from codex_africanus_amd.model.coherency import conversion as C
    bad = json.loads(str(g11["bad_cases"]))
    assert len(bad) == len(g11["errors"]) >= 8
    for (shape, isch, osch, implicit), rec in zip(bad, g11["errors"]):
        cls, msg = str(rec).split("|", 1)
        with pytest.raises(Exception) as ei:
            C.convert_setup(np.zeros(shape), isch, osch, implicit)
        assert type(ei.value).__name__ == cls, (isch, osch)
        assert str(ei.value) == msg, (isch, osch)


def test_convert_setup_resolution():
    from codex_africanus_amd.model.coherency import conversion as C
    x = np.zeros((3, 2, 2), np.float32)
    mapping, ishape, oshape, dtype = C.convert_setup(x, [["XX", "XY"], ["YX", "YY"]], ["I", "Q", "U", "V"], False)
    assert ishape == (2, 2) and oshape == (4,) and dtype == np.complex64
    assert mapping == [(0, 3, C.HALF_ADD, 0), (0, 3, C.HALF_SUB, 1), (1, 2, C.HALF_ADD, 2), (1, 2, C.HALF_SUB_OVER_J, 3)]
    # real products of real input stay real; integer input computes in float64
    assert C.convert_setup(x[..., 0], ["XX", "YY"], ["I", "Q"], False)[3] == np.float32
    assert C.convert_setup(np.zeros((3, 2), np.int32), ["XX", "YY"], ["I", "Q"], False)[3] == np.float64
    assert C.convert_setup(np.zeros((3, 2), np.int32), ["I", "Q"], ["XX", "YY"], False)[3] == np.complex128
    # the first candidate pair wins when both are present; defaults only under implicit_stokes
    m = C.convert_setup(np.zeros((3, 4)), ["RL", "LR", "XX", "YY"], ["Q"], False)[0]
    assert m == [(2, 3, C.HALF_SUB, 0)]
    m = C.convert_setup(np.zeros((3, 1)), ["I"], ["XX", "XY"], True)[0]
    assert m == [(0, -1, C.ADD, 0), (-1, -1, C.ADDJ, 1)]


@pytest.mark.parametrize("nant, nrow", [(64, 40000), (7, 300), (5, 37), (130, 9000), (2, 9), (1, 4)])
def test_fused_group_plan_covers_every_row_once(nant, nrow):
    """af_fused_plan_groups (host side, no GPU work): every row lands in exactly one slot of one group, slot (i, j)'s row
    has antenna1 = p_i and antenna2 = q_j, items hold at most 512 groups; a 64-antenna timestep (2016 baselines) is
    exactly 512 groups = one workgroup of the fused kernel."""
    import ctypes
    from codex_africanus_amd import _lib
    from codex_africanus_amd.testing import synthetic_inputs
    d = synthetic_inputs(seed=1, nrow=nrow, nchan=2, nsrc=2, nant=max(nant, 2))
    a1, a2 = d["ant1"] % nant, d["ant2"] % nant                 # nant = 1: autocorrelations only
    rng = np.random.default_rng(nant)
    if nant == 7:                                                # repeated baselines and a shuffled order inside timesteps
        a1, a2 = np.concatenate([a1, a1[:50]]), np.concatenate([a2, a2[:50]])
        d["time_index"] = np.concatenate([d["time_index"], np.full(50, d["time_index"][-1])])
        nrow += 50
    ti = np.ascontiguousarray(d["time_index"], np.int64) + 3    # plans are relative to the smallest time index
    a1, a2 = np.ascontiguousarray(a1, np.int32), np.ascontiguousarray(a2, np.int32)
    P = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    ni, ng = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.call("af_fused_plan_groups", P(ti), P(a1), P(a2), nrow, nant, None, 0, ctypes.byref(ni), None, 0, ctypes.byref(ng))
    items, groups = np.zeros((ni.value, 4), np.int32), np.zeros((ng.value, 8), np.int32)
    _lib.call("af_fused_plan_groups", P(ti), P(a1), P(a2), nrow, nant, P(items), ni.value, ctypes.byref(ni), P(groups),
              ng.value, ctypes.byref(ng))
    rows = groups[:, 4:].ravel()
    assert np.array_equal(np.sort(rows[rows >= 0]), np.arange(nrow))
    for k, (i, j) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        m = groups[:, 4 + k] >= 0
        assert np.array_equal(a1[groups[m, 4 + k]], groups[m, i]) and np.array_equal(a2[groups[m, 4 + k]], groups[m, 2 + j])
    assert groups[:, :4].min() >= 0 and groups[:, :4].max() < nant
    assert items[:, 2].max() <= 512 and items[:, 2].sum() == len(groups) and np.all(items[:, 3] == 1)
    assert np.array_equal(items[:, 1], np.concatenate([[0], np.cumsum(items[:, 2])[:-1]]))
    # every group's rows share one time index, which is the item's (relative to the minimum)
    for t, g0, cnt, _ in items:
        r = groups[g0:g0 + cnt, 4:].ravel()
        assert np.all(ti[r[r >= 0]] - ti.min() == t)
    if nant == 64:
        assert items[0, 2] == 512
    with pytest.raises(ValueError, match="antenna index out of range"):
        _lib.call("af_fused_plan_groups", P(ti), P(a1 + nant), P(a2), nrow, nant, None, 0, ctypes.byref(ni), None, 0,
                  ctypes.byref(ng))


def test_gemm_plan_host_side_up_to_512_antennas():
    """Round 5 (host side only, no GPU work): the antenna planner and the plan object for arrays beyond 64 antennas --
    decomposable, residual, row map, the slots the GEMM form pays for (super-tiles: super-blocks of 8 blocks, DIAG + 8 x 4
    RECT tiles) and the fill factor the dispatcher uses."""
    from codex_africanus_amd import _lib
    from codex_africanus_amd.rime import fused
    lib = _lib.load()
    tri = lambda n: n * (n + 1) // 2
    # tiles: one DIAG for <= 8 blocks; beyond, DIAGs of the 8-block super-blocks plus every pair's rectangle
    assert lib.af_fused_gemm_slots(64) == 36 * 64 and lib.af_fused_gemm_slots(5) == 64 and lib.af_fused_gemm_slots(0) == 0
    assert lib.af_fused_gemm_slots(128) == (2 * tri(8) + 64) * 64
    assert lib.af_fused_gemm_slots(197) == (3 * tri(8) + tri(1) + 3 * 64 + 3 * 8) * 64      # 25 blocks: 8 + 8 + 8 + 1
    assert lib.af_fused_gemm_slots(256) == (4 * tri(8) + 6 * 64) * 64
    assert lib.af_fused_gemm_slots(512) == (8 * tri(8) + 28 * 64) * 64 and lib.af_fused_gemm_slots(513) == 0       # round 6: 512 stations
    rng = np.random.default_rng(5)
    for nant in (70, 130, 256, 300):
        a1, a2 = np.triu_indices(nant, 1)
        nbl, ntime = a1.shape[0], 2
        ti = np.repeat(np.arange(ntime), nbl) + 3                      # an offset: the plan normalises it
        a1, a2 = np.tile(a1, ntime).astype(np.int32), np.tile(a2, ntime).astype(np.int32)
        x = rng.uniform(-1, 1, (ntime, nant, 3)) * [3000.0, 3000.0, 300.0]
        uvw = x[ti - 3, a1] - x[ti - 3, a2]
        plan = fused.fused_plan(ti, a1, a2, nant, uvw=uvw)
        assert plan.decomposable and plan.residual <= 1e-10 and plan.nsteps == ntime
        nap = 8 * ((nant + 7) // 8)
        assert plan.rowmap.shape == (ntime, nap, nap) and (plan.rowmap >= 0).sum() == ti.shape[0]
        assert np.array_equal(plan.rowmap[ti - 3, a1, a2], np.arange(ti.shape[0]))
        assert np.abs(plan.ant_uvw[ti - 3, a1] - plan.ant_uvw[ti - 3, a2] - uvw).max() <= 1e-10
        assert plan.fill == pytest.approx(nbl / lib.af_fused_gemm_slots(nant)) and 0.75 < plan.fill < 1.0
        assert np.array_equal(plan.step, ti - 3) and plan.step.dtype == np.int32
        # per-row uvw does not decompose: the plan says so and the caller stays on the lane-per-row kernel
        assert not fused.fused_plan(ti, a1, a2, nant, uvw=rng.standard_normal(uvw.shape)).decomposable
    assert fused.fused_plan(ti[:10], a1[:10] % 3, a2[:10] % 3 + 3, 520, uvw=uvw[:10]).decomposable is False   # > 512 antennas


def test_plans_of_single_precision_rows_say_at_which_tolerance_they_decompose():
    """Round 6 (host side only).  float32 rows of a Measurement Set are differences of antenna coordinates only to their own
    rounding: they decompose at 2^-22 max |uvw| when the call is all single precision (``single=True``) -- and such a plan is
    marked (``single_tol``), because the double-precision GEMM route must not take it (rime.fused._fused_predict_vis) --
    and not at the double tolerance.  An explicit ``decompose_tol`` is the caller's decision and is not marked."""
    from codex_africanus_amd.rime import fused
    rng = np.random.default_rng(11)
    nant, ntime = 20, 3
    a1, a2 = np.triu_indices(nant, 1)
    nbl = a1.shape[0]
    ti = np.repeat(np.arange(ntime), nbl)
    a1, a2 = np.tile(a1, ntime).astype(np.int32), np.tile(a2, ntime).astype(np.int32)
    x = (rng.uniform(-1, 1, (ntime, nant, 3)) * [3000.0, 3000.0, 300.0]).astype(np.float32)
    uvw32 = x[ti, a1] - x[ti, a2]
    assert uvw32.dtype == np.float32
    loose = fused.fused_plan(ti, a1, a2, nant, uvw=uvw32, single=True)
    assert loose.decomposable and loose.single_tol and 1e-10 < loose.residual <= loose.tol < 2e-3
    strict = fused.fused_plan(ti, a1, a2, nant, uvw=uvw32)
    assert not strict.decomposable and not strict.single_tol and strict.tol == fused.DECOMPOSE_TOL
    assert strict.items is not None and strict.groups is not None          # the lane-per-row kernels' plan is always there
    chosen = fused.fused_plan(ti, a1, a2, nant, uvw=uvw32, decompose_tol=1e-3)
    assert chosen.decomposable and not chosen.single_tol
    # double rows: `single` changes nothing
    uvw64 = x.astype(np.float64)[ti, a1] - x.astype(np.float64)[ti, a2]
    exact = fused.fused_plan(ti, a1, a2, nant, uvw=uvw64, single=True)
    assert exact.decomposable and not exact.single_tol and exact.residual <= 1e-10
    # the cache keeps the two kinds apart
    assert fused.cached_plan(ti, a1, a2, nant, uvw=uvw32, single=True).decomposable
    assert not fused.cached_plan(ti, a1, a2, nant, uvw=uvw32, single=False).decomposable
    assert fused._all_single(uvw32, None, x) and not fused._all_single(uvw32, uvw64)


def test_plan_cache_by_identity_does_not_outlive_the_plan_cache(monkeypatch):
    """the identity index of cached_plan (tensor objects -> plan) held strong references to plans keyed by
    tensors that no longer exist -- the row-chunk front-ends pass fresh slices every call -- so plans (host arrays + device
    copies) outlived their eviction from the plan cache.  Host side only: CPU tensors take the same code path."""
    import gc
    import weakref
    import torch
    from codex_africanus_amd.rime import fused
    monkeypatch.setenv("AFHIP_PLAN_CACHE", "2")
    fused._plan_cache.clear()
    fused._plan_ident.clear()
    nant = 5
    a1n, a2n = np.triu_indices(nant, 1)
    nbl = a1n.shape[0]

    def arrays(ntime):
        ti = torch.from_numpy(np.repeat(np.arange(ntime), nbl))
        return ti, torch.from_numpy(np.tile(a1n, ntime).astype(np.int32)), torch.from_numpy(np.tile(a2n, ntime).astype(np.int32))

    ti, a1, a2 = arrays(3)
    p = fused.cached_plan(ti, a1, a2, nant)
    assert fused.cached_plan(ti, a1, a2, nant) is p                     # same objects, same version: identity hit
    assert fused.cached_plan(ti[:], a1[:], a2[:], nant) is p            # fresh views (same address: same key): digest hit
    assert len(fused._plan_ident) == 1
    gc.collect()
    assert fused.cached_plan(ti, a1, a2, nant) is p                     # the views are gone: their entry cannot hit, digest
    ti.add_(0)                                                          # in-place write: the version counter moves on
    assert fused.cached_plan(ti, a1, a2, nant) is p                     # digest hit again (same contents)
    # the views of the call above died with it: their entry is purged at the next insert
    assert all(all(r is None or r() is not None for r in refs) for refs, _, _ in fused._plan_ident.values())
    # 20 other layouts through a cache of 2: the first plan is evicted and must be FREED, identity entries or not
    ref = weakref.ref(p)
    del p
    keep = []
    for k in range(20):
        t3 = arrays(4 + k)
        keep.append(t3)                                                 # the tensors stay alive: only the cache limit frees plans
        fused.cached_plan(*t3, nant)
    gc.collect()
    assert ref() is None
    assert len(fused._plan_cache) == 2 and len(fused._plan_ident) <= 8
    live = [e[1]() for e in fused._plan_ident.values()]
    assert sum(x is not None for x in live) <= 2
    # an identity entry whose plan was evicted falls through to the digest and rebuilds
    q = fused.cached_plan(ti, a1, a2, nant)
    assert q.nrow == 3 * nbl and fused.cached_plan(ti, a1, a2, nant) is q
    fused._plan_cache.clear()
    fused._plan_ident.clear()


def test_only_hermitian_brightness_takes_the_gemm_form():
    """Round 6: the GEMM form serves a baseline stored the other way round with the conjugate transpose of the computed
    element, which is right only for Hermitian brightness matrices; the dispatcher's test (host side)"""
    import torch
    from codex_africanus_amd.rime import fused
    rng = np.random.default_rng(3)
    I, Q, U, V = rng.random(7) + 1, 0.1 * rng.standard_normal(7), 0.1 * rng.standard_normal(7), 0.1 * rng.standard_normal(7)
    X = np.stack([I + Q, U + 1j * V, U - 1j * V, I - Q], -1).reshape(7, 2, 2)            # linear feeds, real Stokes
    Xc = np.broadcast_to(X[:, None], (7, 5, 2, 2)) * np.linspace(1, 2, 5)[None, :, None, None]
    for x in (X, Xc, X.astype(np.complex64), X.real.copy(), torch.from_numpy(np.ascontiguousarray(Xc))):
        assert fused._hermitian(x)
    bad = np.array(Xc)
    bad[3, 2, 1, 0] += 1e-12                                    # exactly, not approximately
    bad2 = np.array(X)
    bad2[0, 0, 0] += 1e-9j                                      # a diagonal entry with an imaginary part
    bad3 = rng.standard_normal((7, 2, 2)) + 1j * rng.standard_normal((7, 2, 2))
    for x in (bad, bad2, bad3, torch.from_numpy(bad), bad3.real.copy()):
        assert not fused._hermitian(x)
    assert fused._hermitian(None)
    # a device tensor's verdict is remembered by identity and in-place version
    t = torch.from_numpy(np.ascontiguousarray(Xc))
    assert fused._hermitian(t) and fused._hermitian(t)
    t[0, 0, 0, 1] += 1.0
    assert not fused._hermitian(t)


def test_wgridder_plane_count_folds_the_w_range():
    """af_wgrid_planes (host arithmetic only): the planes cover [min |w|, max |w|] -- visibilities with w < 0 are evaluated
    at their mirror points (real image) -- with plane 0 W/2 - 1 spacings below the smallest |w| and no spare plane behind
    the largest: ceil(span - 1) + W planes of spacing 1 / (4 max|n - 1|)."""
    from codex_africanus_amd import _lib
    lib = _lib.load()
    nm1, W = 1e-3, 7
    dw = 1.0 / (4.0 * nm1)
    planes = lambda lo, hi: int(lib.af_wgrid_planes(float(lo), float(hi), nm1, W, 1))
    assert planes(0.0, 0.0) == W and planes(5.0, 5.0) == W and planes(-3.0, 3.0) == W      # one w (or a range below one spacing)
    assert planes(0.0, 3.6 * dw) == 3 + W and planes(0.0, 4.0 * dw) == 4 + W                # ceil(span - 1 + eta) + W
    assert planes(-3.6 * dw, 3.6 * dw) == planes(0.0, 3.6 * dw)                              # symmetric range: its positive half
    assert planes(-9.3 * dw, -2.0 * dw) == planes(2.0 * dw, 9.3 * dw) == 7 + W               # all-negative: mirrored (span 7.3)
    assert planes(-2.0 * dw, 9.3 * dw) == planes(0.0, 9.3 * dw) == 9 + W                     # mixed: [0, max |w|]
    assert int(lib.af_wgrid_planes(-1e9, 1e9, nm1, W, 0)) == 1                               # no w-stacking: one plane
    assert planes(1.0, -1.0) == -1 and int(lib.af_wgrid_planes(0.0, float("nan"), nm1, W, 1)) == -1
