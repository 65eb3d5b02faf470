"""
The GEMM form of the fused predict (csrc/af_fused_gemm.hip): for antenna-decomposable uvw (uvw_pq = uvw_p - uvw_q per
timestep -- every real Measurement Set) the chain phase_delay -> einsum -> beam_cube_dde -> predict_vis
(africanus/rime/examples/predict.py:107-134,404-472,525) is V(t, nu) = G H^H, evaluated with fp64 MFMA.  Checked against
the CPU oracle chain to 1e-9 of the per-visibility sum of |term| magnitudes (north star: 1e-8), against the general
kernel on the same rows, and the dispatcher: rows that are not decomposable, repeated baselines, Gaussian shapes and
more than 512 antennas stay on the general kernel.
"""
import numpy as np
import pytest

import oracle
from codex_africanus_amd import rime
from codex_africanus_amd.rime import fused
from test_gpu_fused import _problem, _oracle_chain, _scale

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _gemm_at_any_fill(monkeypatch):
    """These tests are about the GEMM form's arithmetic and row layouts at small, sparse shapes: the fill-factor rule that
    would send most of them to the lane-per-row kernel (fused.GEMM_MIN_FILL) is off except where it is the subject."""
    monkeypatch.setenv("AFHIP_GEMM_MIN_FILL", "0")


def _decomposable(d, nant, seed=0, keep=1.0, swap=0.0, shuffle=False, autos=False):
    """Replace the problem's uvw by differences of per-(time, antenna) coordinates; optionally drop baselines, swap
    antenna1 / antenna2 of some rows, shuffle the rows inside every timestep, turn some rows into autocorrelations."""
    rng = np.random.default_rng(1000 + seed)
    ti, a1, a2 = d["time_index"].copy(), d["ant1"].copy(), d["ant2"].copy()
    nrow = ti.shape[0]
    if autos:
        sel = rng.random(nrow) < 0.05
        # an autocorrelation per (time, antenna) at most once: keep the first row of each (time, antenna1)
        key = ti.astype(np.int64) * nant + a1
        first = np.zeros(nrow, bool)
        first[np.unique(key, return_index=True)[1]] = True
        sel &= first
        a2[sel] = a1[sel]
    if swap:
        sw = rng.random(nrow) < swap
        a1[sw], a2[sw] = a2[sw].copy(), a1[sw].copy()
    order = np.arange(nrow)
    if keep < 1.0:
        order = order[rng.random(nrow) < keep]
    if shuffle:
        order = order[np.lexsort((rng.random(order.shape[0]), ti[order]))]
    x = rng.uniform(-1, 1, (d["ntime"], nant, 3)) * np.array([3000.0, 3000.0, 300.0])
    ti, a1, a2 = ti[order], a1[order], a2[order]
    d = dict(d)
    d["time_index"], d["ant1"], d["ant2"] = ti, a1, a2
    d["uvw"] = x[ti, a1] - x[ti, a2]
    d["ant_xyz"] = x
    return d


def _call(d, **kw):
    return rime.fused_predict_vis(d["time_index"], d["ant1"], d["ant2"], d["lm"], d["uvw"], d["frequency"], d["X"],
                                  d["beam"], d["extents"], d["beam_freq_map"], d["pa"], d["pe"], d["as"], **kw)


@pytest.mark.parametrize("nant, nrow", [(5, 37), (7, 300), (8, 500), (12, 1000), (17, 1500), (24, 2000), (33, 2500),
                                        (40, 1700), (47, 3000), (50, 3000), (57, 4000), (64, 4100),
                                        # beyond 64 antennas (round 5): super-tiles -- 65: 5 + 4 blocks; 80: 5 + 5; 96: 6 + 6;
                                        # 100: 5 + 4 + 4; 128: 6 + 5 + 5; 197: 5 x 5; 256: 6 + 6 + 5 + 5 + 5 + 5
                                        (65, 4300), (80, 3500), (96, 9300), (100, 5200), (128, 9000), (197, 20000),
                                        (256, 33000),
                                        # round 6: 8 x 8 RECT super-tiles (four-product form) wherever a column super-block has
                                        # more than 4 blocks (96: 8 + 4 stays 8 x 4; 104: 8 + 5; 128; 197; 256), and arrays up
                                        # to 512 antennas -- 300: 8 + 8 + 8 + 8 + 6 blocks; 512: eight full super-blocks
                                        (104, 5400), (300, 45000), (512, 131000)])
def test_gemm_form_matches_the_reference_chain(nant, nrow):
    d = _decomposable(_problem(3, nrow, 6, 23, nant), nant)
    plan = fused.fused_plan(d["time_index"], d["ant1"], d["ant2"], nant, uvw=d["uvw"])
    assert plan.decomposable and plan.residual <= 1e-10
    out = _call(d, plan=plan)
    ref = _oracle_chain(d, True)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() < 1e-9 * _scale(d)
    # without an explicit plan the call plans for itself, decomposes and takes the same route: same bits
    assert np.array_equal(_call(d), out)


def test_gemm_form_equals_the_general_kernel(monkeypatch):
    d = _decomposable(_problem(11, 4032, 5, 37, 64), 64)
    a = _call(d)
    monkeypatch.setenv("AFHIP_FUSED_GEMM", "0")
    b = _call(d)
    assert not np.array_equal(a, b)                       # two different kernels ...
    assert np.abs(a - b).max() < 1e-10 * _scale(d)        # ... one answer


@pytest.mark.parametrize("case", ["missing", "swapped", "shuffled", "autos", "all"])
def test_row_layouts(case):
    """baselines missing from some timesteps (flagged antennas), rows stored with antenna1 > antenna2 (served by the
    conjugate transpose of the computed tile), any row order inside a timestep, autocorrelations (diagonal blocks)"""
    kw = dict(missing=dict(keep=0.7), swapped=dict(swap=0.4), shuffled=dict(shuffle=True), autos=dict(autos=True),
              all=dict(keep=0.8, swap=0.3, shuffle=True, autos=True))[case]
    nant = 19
    d = _decomposable(_problem(5, 1500, 4, 17, nant), nant, seed=2, **kw)
    plan = fused.fused_plan(d["time_index"], d["ant1"], d["ant2"], nant, uvw=d["uvw"])
    assert plan.decomposable
    out = _call(d, plan=plan)
    ref = _oracle_chain(d, True)
    assert np.abs(out - ref).max() < 1e-9 * _scale(d)


def test_feed_rotation_model_dies_and_convention():
    nant = 13
    d = _decomposable(_problem(8, 900, 5, 11, nant), nant, seed=4)
    rng = d["rng"]
    fr = oracle.feed_rotation(d["pa"], "linear")
    out = _call(d, feed_rotation=fr, convention="casa")
    phase = oracle.phase_delay(d["lm"], d["uvw"], d["frequency"], "casa")
    coh = np.einsum("srf,sfij->srfij", phase, d["X"])
    dde = np.einsum("stafij,tajk->stafik", oracle.beam_cube_dde(d["beam"], d["extents"], d["beam_freq_map"], d["lm"],
                                                                 d["pa"], d["pe"], d["as"], d["frequency"]), fr)
    ref = oracle.predict_vis(d["time_index"], d["ant1"], d["ant2"], dde, coh, dde, None, None, None)
    assert np.abs(out - ref).max() < 1e-9 * _scale(d)
    # DIEs and base_vis ride on top (applied by predict_vis, as in the general route)
    shp = (d["ntime"], nant, 5, 2, 2)
    die = 1.0 + 0.1 * rng.standard_normal(shp) + 0.1j * rng.standard_normal(shp)
    bvis = 0.1 * (rng.standard_normal((900, 5, 2, 2)) + 1j * rng.standard_normal((900, 5, 2, 2)))
    out = _call(d, die1_jones=die, base_vis=bvis, die2_jones=die)
    assert np.abs(out - _oracle_chain(d, True, die, bvis)).max() < 2e-9 * _scale(d)
    # the sky model instead of brightness
    nsrc = d["lm"].shape[0]
    stokes = np.stack([rng.lognormal(0, 1, nsrc)] + [0.1 * rng.standard_normal(nsrc) for _ in range(3)], axis=1)
    spi = rng.uniform(-1.0, 0.2, (nsrc, 2, 4))
    rf = rng.uniform(0.9e9, 1.5e9, nsrc)
    out = rime.fused_predict_vis(d["time_index"], d["ant1"], d["ant2"], d["lm"], d["uvw"], d["frequency"], None,
                                 d["beam"], d["extents"], d["beam_freq_map"], d["pa"], d["pe"], d["as"],
                                 stokes=stokes, spi=spi, ref_freq=rf)
    st = oracle.spectral_model(stokes, spi, rf, d["frequency"], base=0)
    I, Q, U, V = (st[..., k] for k in range(4))           # noqa: E741
    d2 = dict(d, X=np.stack([I + Q, U + 1j * V, U - 1j * V, I - Q], axis=-1).reshape(nsrc, 5, 2, 2))
    assert np.abs(out - _oracle_chain(d2, True)).max() < 1e-9 * _scale(d2)


def test_device_resident_call_and_time_offset():
    import torch
    nant = 10
    d = _decomposable(_problem(9, 700, 4, 9, nant), nant, seed=5)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    plan = fused.fused_plan(d["time_index"] + 11, d["ant1"], d["ant2"], nant, uvw=d["uvw"])
    out = rime.fused_predict_vis(t(d["time_index"] + 11), t(d["ant1"]), t(d["ant2"]), t(d["lm"]), t(d["uvw"]),
                                 t(d["frequency"]), t(d["X"]), t(d["beam"]), t(d["extents"]), t(d["beam_freq_map"]),
                                 t(d["pa"]), t(d["pe"]), t(d["as"]), plan=plan)
    assert plan.decomposable and out.is_cuda
    assert np.abs(out.cpu().numpy() - _oracle_chain(d, True)).max() < 1e-9 * _scale(d)


def test_dispatcher_falls_back():
    """what must NOT take the GEMM route: uvw drawn per row (BASELINE's recipe), uvw decomposable only to 1e-6 m, the same
    baseline twice in a timestep, Gaussian shapes, more than 512 antennas -- all give the reference's answer"""
    nant = 9
    base = _problem(12, 600, 4, 13, nant)
    plan = fused.fused_plan(base["time_index"], base["ant1"], base["ant2"], nant, uvw=base["uvw"])
    assert not plan.decomposable and plan.residual > 1.0
    assert np.abs(_call(base, plan=plan) - _oracle_chain(base, True)).max() < 1e-9 * _scale(base)
    d = _decomposable(base, nant, seed=6)
    noisy = dict(d, uvw=d["uvw"] + 1e-6 * np.random.default_rng(1).standard_normal(d["uvw"].shape))
    plan = fused.fused_plan(noisy["time_index"], noisy["ant1"], noisy["ant2"], nant, uvw=noisy["uvw"])
    assert not plan.decomposable and 1e-8 < plan.residual < 1e-4
    assert fused.fused_plan(noisy["time_index"], noisy["ant1"], noisy["ant2"], nant, uvw=noisy["uvw"],
                            decompose_tol=1e-4).decomposable
    assert np.abs(_call(noisy) - _oracle_chain(noisy, True)).max() < 1e-9 * _scale(noisy)
    dup = dict(d)
    for k in ("time_index", "ant1", "ant2", "uvw"):
        dup[k] = np.concatenate([d[k], d[k][:5]])
    plan = fused.fused_plan(dup["time_index"], dup["ant1"], dup["ant2"], nant, uvw=dup["uvw"])
    assert not plan.decomposable
    order = np.argsort(dup["time_index"], kind="stable")
    for k in ("time_index", "ant1", "ant2", "uvw"):
        dup[k] = dup[k][order]
    assert np.abs(_call(dup) - _oracle_chain(dup, True)).max() < 1e-9 * _scale(dup)
    gs = np.zeros((13, 3))
    gs[::2] = [1e-4, 5e-5, 0.3]
    out = _call(d, gauss_shape=gs)
    phase = oracle.phase_delay(d["lm"], d["uvw"], d["frequency"]) * oracle.gaussian_shape(d["uvw"], d["frequency"], gs)
    dde = oracle.beam_cube_dde(d["beam"], d["extents"], d["beam_freq_map"], d["lm"], d["pa"], d["pe"], d["as"], d["frequency"])
    ref = oracle.predict_vis(d["time_index"], d["ant1"], d["ant2"], dde, np.einsum("srf,sfij->srfij", phase, d["X"]), dde,
                             None, None, None)
    assert np.abs(out - ref).max() < 1e-9 * _scale(d)
    big = _decomposable(_problem(13, 2500, 3, 5, 520), 520, seed=7)
    plan = fused.fused_plan(big["time_index"], big["ant1"], big["ant2"], 520, uvw=big["uvw"])
    assert not plan.decomposable
    assert np.abs(_call(big) - _oracle_chain(big, True)).max() < 1e-9 * _scale(big)


def test_plan_through_the_c_abi():
    """af_fused_plan_antennas called directly: residual, antenna coordinates up to a per-timestep shift, row map"""
    import ctypes
    from codex_africanus_amd import _lib
    nant = 6
    d = _decomposable(_problem(14, 45, 2, 3, nant), nant, seed=8, keep=0.8, swap=0.3)
    ti = d["time_index"].astype(np.int64)
    a1, a2 = d["ant1"].astype(np.int32), d["ant2"].astype(np.int32)
    nsteps = int(ti.max() - ti.min()) + 1
    au, rm = np.zeros((nsteps, nant, 3)), np.zeros((nsteps, 8, 8), np.int32)
    res, ok = ctypes.c_double(), ctypes.c_int()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _lib.call("af_fused_plan_antennas", P(ti), P(a1), P(a2), P(d["uvw"]), ti.shape[0], nant, 1e-10, nsteps, P(au), P(rm),
              ctypes.byref(res), ctypes.byref(ok))
    assert ok.value == 1 and res.value < 1e-11
    assert np.abs(au[ti, a1] - au[ti, a2] - d["uvw"]).max() == res.value
    for r in range(ti.shape[0]):
        assert rm[ti[r], a1[r], a2[r]] == r
    assert (rm >= 0).sum() == ti.shape[0]


@pytest.mark.parametrize("case", ["beam", "beam_die", "beam_feed"])
def test_gemm_form_against_the_reference_itself(case):
    """G16 (tests/golden/make_golden_gemm.py): the REFERENCE's phase_delay -> einsum -> beam_cube_dde [-> feed rotation] ->
    predict_vis [+ DIE terms, base_vis] under the real numba, on G14's sky / beam / per-antenna terms with a Measurement
    Set's uvw (differences of antenna coordinates): the rows the dispatcher sends to the GEMM form."""
    from conftest import load_golden
    from fused_cases import linear_feed_rotation
    g, h = load_golden("g14_fused_dask.npz"), load_golden("g16_fused_gemm.npz")
    nant = g["parallactic_angles"].shape[1]
    plan = fused.fused_plan(g["time_index"], g["antenna1"], g["antenna2"], nant, uvw=h["uvw"])
    assert plan.decomposable
    kw = {}
    if case == "beam_die":
        kw = dict(die1_jones=g["die"], base_vis=g["base_vis"], die2_jones=g["die"])
    if case == "beam_feed":
        kw = dict(feed_rotation=linear_feed_rotation(g["parallactic_angles"]))
    out = rime.fused_predict_vis(g["time_index"], g["antenna1"], g["antenna2"], g["lm"], h["uvw"], g["frequency"], g["brightness"],
                                 g["beam"], g["beam_lm_extents"], g["beam_freq_map"], g["parallactic_angles"], g["point_errors"],
                                 g["antenna_scaling"], plan=plan, **kw)
    ref = h["vis_" + case]
    scale = max(float(h["scale"]), float(np.abs(ref).max()))
    assert out.shape == ref.shape and np.abs(out - ref).max() <= 1e-9 * scale
    # the general kernel on the same rows (planner told not to decompose) agrees as well
    gen = rime.fused_predict_vis(g["time_index"], g["antenna1"], g["antenna2"], g["lm"], h["uvw"], g["frequency"], g["brightness"],
                                 g["beam"], g["beam_lm_extents"], g["beam_freq_map"], g["parallactic_angles"], g["point_errors"],
                                 g["antenna_scaling"], plan=fused.fused_plan(g["time_index"], g["antenna1"], g["antenna2"], nant), **kw)
    assert np.abs(gen - ref).max() <= 1e-9 * scale


@pytest.mark.parametrize("single", [False, True])
def test_brightness_matrices_that_are_not_hermitian(single, monkeypatch):
    """The reference's chain takes ANY complex (source, chan, 2, 2) brightness (einsum("srf,sfij->srfij")); coherency matrices
    of real Stokes parameters are Hermitian, and the GEMM form relies on it (a baseline stored with its antennas in the other
    block order gets the conjugate transpose of the computed element: A_q X^H A_p^H, not A_q X A_p^H).  Round 6: such a call
    takes the lane-per-row kernel -- and gives the reference's answer; until then a quarter of these rows were wrong
    (tools/r6_check_hermitian.py: 280 of 1200, by up to 0.25 of the summed brightness)."""
    nant = 19
    d = _decomposable(_problem(31, 1200, 4, 11, nant), nant, seed=3, swap=0.3)
    rng = np.random.default_rng(7)
    d["X"] = d["X"] + 0.3 * (rng.standard_normal(d["X"].shape) + 1j * rng.standard_normal(d["X"].shape))     # no symmetry left
    assert not fused._hermitian(d["X"])
    if single:
        import test_gpu_fused_gemm_c64 as FC
        s = FC._single(d)
        out = FC._call_s(s)
        assert out.dtype == np.complex64
        ref = FC._chain64(s)
        # (af_fused_predict_c64: the lane-per-row kernel in single precision, tests/test_gpu_fused_rows_c64.py)
        import test_gpu_fused_rows_c64 as FR
        assert np.abs(out - ref).max() <= FR._tol(s) * _scale(d)
        return
    out = _call(d)
    ref = _oracle_chain(d, True)
    assert np.abs(out - ref).max() < 1e-9 * _scale(d)
    monkeypatch.setenv("AFHIP_FUSED_GEMM", "0")
    assert np.array_equal(out, _call(d))                  # it WAS the lane-per-row kernel
    monkeypatch.delenv("AFHIP_FUSED_GEMM")
    # device tensors: the same route (one small reduction + one host read per brightness tensor)
    import torch
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dev = rime.fused_predict_vis(t(d["time_index"]), t(d["ant1"]), t(d["ant2"]), t(d["lm"]), t(d["uvw"]), t(d["frequency"]), t(d["X"]),
                                 t(d["beam"]), t(d["extents"]), t(d["beam_freq_map"]), t(d["pa"]), t(d["pe"]), t(d["as"]))
    assert np.array_equal(dev.cpu().numpy(), out)


def test_fill_factor_dispatch(monkeypatch):
    """the GEMM form pays for every baseline slot of the block triangle, the row kernel for every row: a
    20-antenna sub-array on a 64-antenna axis (190 rows per step against 2304 slots) must take the row kernel"""
    nant = 64
    d = _problem(21, 4032, 3, 11, nant)
    sub = (d["ant1"] < 20) & (d["ant2"] < 20)
    for k in ("time_index", "ant1", "ant2", "uvw"):
        d[k] = d[k][sub]
    d = _decomposable(d, nant, seed=9)
    plan = fused.fused_plan(d["time_index"], d["ant1"], d["ant2"], nant, uvw=d["uvw"])
    assert plan.decomposable and 0.05 < plan.fill < 0.12
    monkeypatch.delenv("AFHIP_GEMM_MIN_FILL")
    out = _call(d, plan=plan)                                   # the default rule: the row kernel
    monkeypatch.setenv("AFHIP_FUSED_GEMM", "0")
    assert np.array_equal(out, _call(d))
    monkeypatch.delenv("AFHIP_FUSED_GEMM")
    monkeypatch.setenv("AFHIP_GEMM_MIN_FILL", "0")
    forced = _call(d, plan=plan)                                # the GEMM form on the same plan: other bits, same answer
    assert not np.array_equal(out, forced)
    assert np.abs(out - forced).max() < 1e-10 * _scale(d)
    full = _decomposable(_problem(22, 4032, 2, 5, nant), nant, seed=10)
    assert fused.fused_plan(full["time_index"], full["ant1"], full["ant2"], nant, uvw=full["uvw"]).fill == 2016 / 2304.0


@pytest.mark.parametrize("route", ["gemm", "rows"])
def test_stale_plan_is_refused(route, monkeypatch):
    """VERDICT r4 item 7 / ADVICE r4: a plan is bound to the (time_index, antenna1, antenna2, uvw) it was made from; the
    call verifies a caller-supplied plan on the device and refuses a stale one instead of computing with the old arrays"""
    import torch
    import codex_africanus_amd as af
    if route == "rows":
        monkeypatch.setenv("AFHIP_FUSED_GEMM", "0")
    nant = 12
    d = _decomposable(_problem(23, 660, 3, 7, nant), nant, seed=11)
    plan = fused.fused_plan(d["time_index"], d["ant1"], d["ant2"], nant, uvw=d["uvw"])
    assert plan.decomposable == (route == "gemm")
    good = _call(d, plan=plan)
    assert np.abs(good - _oracle_chain(d, True)).max() < 1e-9 * _scale(d)
    # the same plan with a time offset on the call's side is the same layout
    assert np.array_equal(_call(dict(d, time_index=d["time_index"] + 5), plan=plan), good)
    # other antennas on the same rows
    other = dict(d, ant1=d["ant2"].copy(), ant2=d["ant1"].copy())
    with pytest.raises(ValueError, match="stale plan.*antenna"):
        _call(other, plan=plan)
    # rows of two timesteps exchanged (same antennas): the steps differ
    ti2 = d["time_index"].copy()
    ti2[ti2 == 1] = 7
    with pytest.raises(ValueError, match="stale plan"):
        _call(dict(d, time_index=ti2), plan=plan)
    if route == "gemm":
        # re-phased uvw (another field): decomposable, but not by THIS plan's antenna coordinates
        moved = _decomposable(d, nant, seed=12)
        assert not np.allclose(moved["uvw"], d["uvw"])
        with pytest.raises(ValueError, match="stale plan.*uvw"):
            _call(moved, plan=plan)
        # uvw equal to rounding (1e-12 m): the same plan serves
        near = dict(d, uvw=d["uvw"] + 1e-12)
        assert np.abs(_call(near, plan=plan) - good).max() < 1e-10 * _scale(d)
    # device tensors: nothing synchronises in the call; the result is NaN and the error surfaces at check_status()
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = [t(other[k]) for k in ("time_index", "ant1", "ant2", "lm", "uvw", "frequency", "X", "beam", "extents",
                                  "beam_freq_map", "pa", "pe", "as")]
    out = rime.fused_predict_vis(*args, plan=plan)
    with pytest.raises(ValueError, match="stale plan"):
        af.check_status()
    assert bool(torch.isnan(out.real).all())
    args = [t(d[k]) for k in ("time_index", "ant1", "ant2", "lm", "uvw", "frequency", "X", "beam", "extents",
                              "beam_freq_map", "pa", "pe", "as")]
    out = rime.fused_predict_vis(*args, plan=plan)
    af.check_status()
    assert np.array_equal(out.cpu().numpy(), good)


def test_cached_plan_by_tensor_identity():
    """device-resident index arrays are looked up by identity + torch's version counter before anything is
    copied to the host; an in-place change is a miss"""
    import torch
    nant = 9
    d = _decomposable(_problem(24, 300, 2, 3, nant), nant, seed=13)
    dev = torch.device("cuda:0")
    ti, a1, a2, uvw = (torch.from_numpy(np.ascontiguousarray(d[k])).to(dev) for k in ("time_index", "ant1", "ant2", "uvw"))
    p1 = fused.cached_plan(ti, a1, a2, nant, uvw=uvw)
    assert fused.cached_plan(ti, a1, a2, nant, uvw=uvw) is p1
    assert fused.cached_plan(ti.clone(), a1, a2, nant, uvw=uvw) is p1          # other object, same contents: the digest
    uvw.mul_(2.0)                                                             # in place: version counter moves
    p2 = fused.cached_plan(ti, a1, a2, nant, uvw=uvw)
    assert p2 is not p1 and p2.decomposable
    assert np.allclose(p2.ant_uvw, 2.0 * p1.ant_uvw)


@pytest.mark.parametrize("nant", [72, 130])
def test_large_arrays_row_layouts_feed_rotation_and_general_kernel(nant, monkeypatch):
    """super-tiles with missing baselines, swapped antennas, shuffled rows and autocorrelations; feed rotation (the one
    instantiation family that differs); GEMM form against the lane-per-row kernel on the same rows"""
    d = _decomposable(_problem(31, 2 * nant * (nant - 1) // 2 + 100, 3, 9, nant), nant, seed=14, keep=0.85, swap=0.3,
                      shuffle=True, autos=True)
    plan = fused.fused_plan(d["time_index"], d["ant1"], d["ant2"], nant, uvw=d["uvw"])
    assert plan.decomposable
    out = _call(d, plan=plan)
    ref = _oracle_chain(d, True)
    assert np.abs(out - ref).max() < 1e-9 * _scale(d)
    fr = oracle.feed_rotation(d["pa"], "linear")
    out_fr = _call(d, feed_rotation=fr)
    monkeypatch.setenv("AFHIP_FUSED_GEMM", "0")
    assert np.abs(_call(d) - out).max() < 1e-10 * _scale(d)
    assert np.abs(_call(d, feed_rotation=fr) - out_fr).max() < 1e-10 * _scale(d)


def test_numpy_call_downloads_in_timestep_aligned_chunks(monkeypatch):
    """the numpy-in / numpy-out GEMM route produces its result in timestep-aligned row chunks whose downloads overlap the
    next chunk's kernels (Call.result_rows with the plan's step boundaries as cut points): same bits as the one-piece call,
    also with steps that have no rows and a time offset"""
    nant = 24
    d = _decomposable(_problem(41, 6 * 276 + 100, 5, 13, nant), nant, seed=15)
    keep = (d["time_index"] != 2)                       # a step without rows in the middle
    for k in ("time_index", "ant1", "ant2", "uvw"):
        d[k] = d[k][keep]
    d["time_index"] = d["time_index"] + 4               # the per-time arrays are indexed from min(time_index), as in predict_vis
    monkeypatch.setenv("AFHIP_D2H_PIPELINE", "0")
    whole = _call(d)
    monkeypatch.delenv("AFHIP_D2H_PIPELINE")
    monkeypatch.setenv("AFHIP_D2H_CHUNK_MB", "0.1")      # 320 B per row: ~330 rows per chunk, cut at step boundaries (276 rows)
    piped = _call(d)
    assert np.array_equal(piped, whole)
    ref_d = dict(d, time_index=d["time_index"] - 4)
    assert np.abs(whole - _oracle_chain(ref_d, True)).max() < 1e-9 * _scale(d)


@pytest.mark.parametrize("seed", range(3000, 3020))
def test_gemm_form_random_arrays_and_layouts(seed):
    """seeded sweep over array sizes 2 .. 256 antennas (every DIAG size, RECT super-tiles with 1 .. 4 column blocks), source
    counts that leave partial batches, missing / swapped / shuffled / autocorrelation rows (tools/stress_random.py runs the
    same sweep on any number of further seeds)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("stress_random_fg", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools", "stress_random.py"))
    src = open(spec.origin).read().split("first = int(sys.argv[1])")[0]       # the sweep functions, not the driver loop
    ns = {"__file__": spec.origin, "__name__": "stress_random_fg"}
    exec(compile(src, spec.origin, "exec"), ns)
    ns["fused_gemm_sweep"](seed)
