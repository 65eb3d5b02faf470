"""
The single-precision GEMM form of the fused predict (csrc/af_fused_gemm_c64.hip, af_fused_predict_antennas_c64): every
input float32 / complex64 -> complex64, the precision in which the reference runs the chain for such callers
(africanus/util/type_inference.py:24-26).  Contract (as af_im_to_vis_f32's, G13): CLOSER to the float64 chain on the same
float32 inputs than the reference's own float32 chain is.  Golden G17 (tests/golden/make_golden_gemm_f32.py) holds both
of the REFERENCE's results -- float32 chain and float64 chain of the promoted values -- for two cases; larger arrays (every
super-tile kind) are checked against the oracle's float64 chain with the error the reference's float32 arithmetic shows on
G17 as the yardstick.
"""
import os

import numpy as np
import pytest

import oracle
from codex_africanus_amd import rime
from codex_africanus_amd.rime import fused
from test_gpu_fused import _problem, _scale
from test_gpu_fused_gemm import _decomposable

pytestmark = pytest.mark.gpu
G17 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g17_fused_gemm_f32.npz"))
NAMES = ("lm", "uvw", "frequency", "brightness", "beam", "beam_lm_extents", "beam_freq_map", "parallactic_angles", "point_errors",
         "antenna_scaling")


@pytest.fixture(autouse=True)
def _gemm_at_any_fill(monkeypatch):
    monkeypatch.setenv("AFHIP_GEMM_MIN_FILL", "0")


def _case(name):
    d = {k: G17["%s_%s" % (name, k)] for k in NAMES + ("time_index", "antenna1", "antenna2", "ant_xyz")}
    for k in NAMES:
        assert d[k].dtype in (np.float32, np.complex64), k
    return d


def _call(d, **kw):
    return rime.fused_predict_vis(d["time_index"], d["antenna1"], d["antenna2"], d["lm"], d["uvw"], d["frequency"], d["brightness"],
                                  d["beam"], d["beam_lm_extents"], d["beam_freq_map"], d["parallactic_angles"], d["point_errors"],
                                  d["antenna_scaling"], **kw)


@pytest.mark.parametrize("name", ["a", "b"])
@pytest.mark.parametrize("feed", [False, True])
def test_closer_to_the_float64_chain_than_the_reference_float32_chain(name, feed):
    d = _case(name)
    nant = d["parallactic_angles"].shape[1]
    plan = fused.fused_plan(d["time_index"], d["antenna1"], d["antenna2"], nant, uvw=d["uvw"], single=True)
    # float32 rows decompose at their own precision: the residual is a few float32 roundings of a 6 km difference
    assert plan.decomposable and plan.residual <= plan.tol and plan.tol < 2e-3
    kw = {}
    if feed:
        kw["feed_rotation"] = rime.feed_rotation(d["parallactic_angles"], "linear")
        assert kw["feed_rotation"].dtype == np.complex64
    got = _call(d, plan=plan, **kw)
    tag = "_feed" if feed else ""
    ref32, truth = G17["%s_vis32%s" % (name, tag)], G17["%s_vis64%s" % (name, tag)]
    assert got.dtype == np.complex64 and got.shape == ref32.shape
    peak = np.abs(truth).max()
    e_ours, e_ref = np.abs(got - truth).max() / peak, np.abs(ref32 - truth).max() / peak
    assert e_ours < 0.5 * e_ref, (e_ours, e_ref)
    assert e_ours < 1e-4, e_ours
    # without an explicit plan: the call plans for itself and takes the same route
    assert np.array_equal(_call(d, **kw), got)


def _single(d):
    """A test_gpu_fused problem in single precision: float32 antenna coordinates, rows = their float32 differences."""
    s = dict(d)
    f, c = np.float32, np.complex64
    xyz = d["ant_xyz"].astype(f)
    s["uvw"] = xyz[d["time_index"], d["ant1"]] - xyz[d["time_index"], d["ant2"]]
    for k in ("lm", "frequency", "extents", "beam_freq_map", "pa", "pe", "as"):
        s[k] = d[k].astype(f)
    s["X"], s["beam"] = d["X"].astype(c), d["beam"].astype(c)
    return s


def _chain64(s, rows=None):
    """the oracle's float64 chain on the promoted single-precision values"""
    r = slice(None) if rows is None else rows
    p = lambda a: a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    dde = oracle.beam_cube_dde(p(s["beam"]), p(s["extents"]), p(s["beam_freq_map"]), p(s["lm"]), p(s["pa"]), p(s["pe"]), p(s["as"]),
                               p(s["frequency"]))
    phase = oracle.phase_delay(p(s["lm"]), p(s["uvw"])[r], p(s["frequency"]))
    coh = np.einsum("srf,sfij->srfij", phase, p(s["X"]))
    return oracle.predict_vis(s["time_index"][r], s["ant1"][r], s["ant2"][r], dde, coh, dde, None, None, None)


def _tol(s, d):
    """What the single-precision GEMM form can differ from the float64 chain by, relative to _scale(d) = sum_s |X_s|: float32
    rows are not EXACTLY differences of antenna coordinates (each was rounded when it was made), so the form's baseline
    x_p - x_q differs from the row's uvw by the plan's residual (<= 2^-24 of 6 km = 3.6e-4 m): a phase of
    2 pi nu / c (|l| + |m| + |n|) residual per term, the same sign for neighbouring sources -- plus float32 arithmetic (a few
    1e-6 of the sum).  For scale: the reference's own float32 chain computes l u + m v + n w (hundreds of metres) in float32,
    ~1e-4 m of rounding at best."""
    nant = s["pa"].shape[1]
    plan = fused.fused_plan(s["time_index"], s["ant1"], s["ant2"], nant, uvw=s["uvw"], single=True)
    assert plan.decomposable and plan.residual <= plan.tol
    lm = s["lm"].astype(np.float64)
    n = np.sqrt(1.0 - (lm ** 2).sum(1)) - 1.0
    reach = float((np.abs(lm).sum(1) + np.abs(n)).max())
    # |V_ij| <= (sum_a |E_ia|) (sum_b |E_jb|) max |X_ab|: the beam's gain on top of the brightness _scale() measures
    gain = float(np.abs(s["beam"]).sum(axis=-1).max()) ** 2
    return 2.0 * np.pi * float(s["frequency"].max()) / 299792458.0 * reach * plan.residual * max(gain, 1.0) + 2e-5


def _call_s(s, **kw):
    return rime.fused_predict_vis(s["time_index"], s["ant1"], s["ant2"], s["lm"], s["uvw"], s["frequency"], s["X"], s["beam"],
                                  s["extents"], s["beam_freq_map"], s["pa"], s["pe"], s["as"], **kw)


# every instantiation kind: DIAG of 1 .. 8 blocks (padded super-rounds below 8), RECT 8 x 4 (65, 100: a column super-block of
# <= 4 blocks), RECT 8 x 8 with a short (104: 5 blocks) and full column super-block (128, 197, 300)
@pytest.mark.parametrize("nant, nrow", [(5, 37), (12, 600), (17, 1500), (24, 2000), (33, 2500), (40, 1700), (47, 3000), (57, 4000),
                                        (64, 4100), (65, 4300), (100, 5200), (104, 5400), (128, 9000), (197, 20000), (300, 45000)])
def test_against_the_float64_chain_at_every_super_tile_kind(nant, nrow):
    d = _decomposable(_problem(3, nrow, 6, 23, nant), nant)
    s = _single(d)
    got = _call_s(s)
    assert got.dtype == np.complex64
    truth = _chain64(s)
    # G17: the reference's own float32 chain is off by 2-3e-4 of the peak at these baselines (float32 phases of ~5000 rad).
    # What bounds this entry is the rows themselves (_tol): a few 1e-5 of the sum typically, 4e-4 at worst.
    assert np.abs(got - truth).max() < _tol(s, d) * _scale(d), (np.abs(got - truth).max() / _scale(d), _tol(s, d))


def test_more_channels_than_one_group_of_beam_planes():
    """70 channels: the per-channel beam planes are built 64 channels at a time (PLANE_GROUP), the kernel is launched per group"""
    nant = 9
    d = _decomposable(_problem(4, 300, 70, 5, nant), nant)
    s = _single(d)
    got = _call_s(s)
    assert got.dtype == np.complex64 and got.shape == (300, 70, 2, 2)
    assert np.abs(got - _chain64(s)).max() < _tol(s, d) * _scale(d)


def test_numpy_call_downloads_in_timestep_aligned_chunks(monkeypatch):
    """as on the double-precision GEMM route: the numpy-in / numpy-out call produces its complex64 result in timestep-aligned
    row chunks whose downloads overlap the next chunk's kernels -- same bits as the one-piece call, also with a step that has
    no rows"""
    nant = 24
    d = _decomposable(_problem(41, 6 * 276 + 100, 5, 13, nant), nant, seed=15)
    keep = (d["time_index"] != 2)
    for k in ("time_index", "ant1", "ant2", "uvw"):
        d[k] = d[k][keep]
    s = _single(d)
    monkeypatch.setenv("AFHIP_D2H_PIPELINE", "0")
    whole = _call_s(s)
    monkeypatch.delenv("AFHIP_D2H_PIPELINE")
    monkeypatch.setenv("AFHIP_D2H_CHUNK_MB", "0.05")     # 160 B per row: ~330 rows per chunk, cut at step boundaries (276 rows)
    piped = _call_s(s)
    assert piped.dtype == np.complex64 and np.array_equal(piped, whole)
    assert np.abs(whole - _chain64(s)).max() < _tol(s, d) * _scale(d)


def test_row_layouts_dies_and_the_plan_guard():
    nant = 19
    d = _decomposable(_problem(5, 1500, 4, 17, nant), nant, seed=2, keep=0.8, swap=0.3, shuffle=True, autos=True)
    s = _single(d)
    got = _call_s(s)
    truth = _chain64(s)
    assert np.abs(got - truth).max() < _tol(s, d) * _scale(d)
    # DIEs and base visibilities on top: predict_vis's complex64 kernel
    rng = np.random.default_rng(5)
    shp = (d["ntime"], nant, 4, 2, 2)
    die = (1.0 + 0.1 * rng.standard_normal(shp) + 0.1j * rng.standard_normal(shp)).astype(np.complex64)
    bvis = (0.1 * (rng.standard_normal(got.shape) + 1j * rng.standard_normal(got.shape))).astype(np.complex64)
    full = _call_s(s, die1_jones=die, base_vis=bvis, die2_jones=die)
    assert full.dtype == np.complex64
    ref = oracle.predict_vis(s["time_index"], s["ant1"], s["ant2"], None, got[None], None, die, bvis, die)
    assert np.array_equal(full, ref)
    # a plan of other rows is refused: NaN result and ValueError (af_fused_plan_check on the complex64 result)
    plan = fused.fused_plan(s["time_index"], s["ant1"], s["ant2"], nant, uvw=s["uvw"], single=True)
    other = dict(s)
    other["uvw"] = (s["uvw"] * np.float32(1.01)).astype(np.float32)
    with pytest.raises(ValueError, match="stale plan"):
        _call_s(other, plan=plan)


def test_mixed_precision_inputs_take_the_float64_route():
    """one float64 array among the inputs: the promoted type is double (africanus/util/type_inference.py:24-26), and the
    float32 rows -- decomposable only at their own precision -- are then NOT handed to the GEMM form: the double result
    follows the rows exactly (lane-per-row kernel), checked against the float64 chain to 1e-9"""
    nant = 12
    d = _decomposable(_problem(7, 600, 4, 9, nant), nant)
    s = _single(d)
    s["lm"] = s["lm"].astype(np.float64)
    out = _call_s(s)
    assert out.dtype == np.complex128
    assert not fused.fused_plan(s["time_index"], s["ant1"], s["ant2"], nant, uvw=s["uvw"]).decomposable
    assert np.abs(out - _chain64(s)).max() < 1e-9 * _scale(d)


def test_single_precision_rows_that_do_not_decompose_take_the_single_precision_row_kernel():
    """uvw drawn per row (BASELINE's recipe) in float32: no antenna coordinates reproduce them -- the lane-per-row kernel's
    single-precision form (af_fused_predict_c64, tests/test_gpu_fused_rows_c64.py) computes the call in the type the
    reference's rule gives it, complex64 (africanus/util/type_inference.py:24-26).  Without a beam: the direct transform in
    double, rounded once, same rule."""
    nant = 12
    d = _problem(7, 600, 4, 9, nant)
    s = _single(dict(d, ant_xyz=np.zeros((d["ntime"], nant, 3))))
    s["uvw"] = d["uvw"].astype(np.float32)
    out = _call_s(s)
    assert out.dtype == np.complex64
    truth = _chain64(s)
    assert np.abs(out - truth).max() <= 2e-5 * _scale(d)
    nobeam = rime.fused_predict_vis(s["time_index"], s["ant1"], s["ant2"], s["lm"], s["uvw"], s["frequency"], s["X"])
    assert nobeam.dtype == np.complex64


def test_device_resident_tensors():
    import torch
    nant = 24
    s = _single(_decomposable(_problem(9, 2000, 5, 15, nant), nant))
    host = _call_s(s)
    t = {k: (torch.from_numpy(np.ascontiguousarray(v)).cuda() if isinstance(v, np.ndarray) else v) for k, v in s.items()}
    dev = _call_s(t)
    assert dev.dtype == torch.complex64 and dev.is_cuda
    assert np.array_equal(dev.cpu().numpy(), host)


def test_the_entry_through_the_c_abi():
    """af_fused_predict_antennas_c64 called directly (device pointers, explicit workspace): the same bits as the front-end,
    its status codes and messages, the empty cases"""
    import ctypes
    import torch
    from codex_africanus_amd import _lib
    lib = _lib.load()
    nant = 12
    d = _decomposable(_problem(7, 600, 4, 9, nant), nant)
    s = _single(d)
    want = _call_s(s)
    plan = fused.fused_plan(s["time_index"], s["ant1"], s["ant2"], nant, uvw=s["uvw"], single=True)
    assert plan.decomposable
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    v = {k: t(s[k]) for k in ("lm", "frequency", "X", "beam", "extents", "beam_freq_map", "pa", "pe", "as")}
    au, rm = t(plan.ant_uvw), t(plan.rowmap)
    nsrc, nchan, nrow = 9, 4, 600
    lw, mh, nud = s["beam"].shape[:3]
    ntime = s["pa"].shape[0]
    ws_bytes = int(lib.af_fused_predict_c64_workspace_bytes(nsrc, nchan, lw, mh, nud))
    assert ws_bytes > 0 and lib.af_fused_predict_c64_workspace_bytes(-1, nchan, lw, mh, nud) == 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = torch.full((nrow, nchan, 2, 2), complex(float("nan"), 0.0), dtype=torch.complex64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def args(**kw):
        return [P(au), P(rm), plan.nsteps, nrow, kw.get("lm", P(v["lm"])), P(v["frequency"]), P(v["X"]), kw.get("nsrc", nsrc), nchan,
                P(v["beam"]), kw.get("lw", lw), mh, nud, P(v["extents"]), P(v["beam_freq_map"]), P(v["pa"]), kw.get("ntime", ntime),
                kw.get("nant", nant), P(v["pe"]), P(v["as"]), None, kw.get("conv", _lib.CONVENTION["fourier"]), P(out),
                kw.get("ws", P(ws)), kw.get("wsb", ws_bytes), stream]

    assert lib.af_fused_predict_antennas_c64(*args()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    assert lib.af_fused_predict_antennas_c64(*args(conv=0)) == 1 and b"convention not in ('fourier', 'casa')" in lib.af_last_error()
    assert lib.af_fused_predict_antennas_c64(*args(wsb=64)) == 1 and b"workspace too small" in lib.af_last_error()
    assert lib.af_fused_predict_antennas_c64(*args(ws=ctypes.c_void_p(ws.data_ptr() + 8))) == 1 and b"aligned" in lib.af_last_error()
    assert lib.af_fused_predict_antennas_c64(*args(lm=None)) == 1 and b"NULL" in lib.af_last_error()
    assert lib.af_fused_predict_antennas_c64(*args(nant=513)) == 1 and b"512 antennas" in lib.af_last_error()
    assert lib.af_fused_predict_antennas_c64(*args(ntime=plan.nsteps - 1)) == 1 and b"timesteps" in lib.af_last_error()
    assert lib.af_fused_predict_antennas_c64(*args(lw=1)) == 1 and b"must be >= 2" in lib.af_last_error()
    # no sources: zeros; a good call after failures still works
    out.fill_(complex(float("nan"), 0.0))
    assert lib.af_fused_predict_antennas_c64(*args(nsrc=0)) == 0
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0
    assert lib.af_fused_predict_antennas_c64(*args()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
