"""
The single-precision lane-per-row form of the fused predict (csrc/af_fused_predict_c64.hip, af_fused_predict_c64): every
input float32 / complex64 -> complex64 on ANY uvw -- rows that do not decompose by antenna (BASELINE configs[2] as it draws
them), Gaussian shapes, non-Hermitian brightness.  Contract as the GEMM form's (tests/test_gpu_fused_gemm_c64.py): CLOSER
to the float64 chain on the same float32 inputs than the reference's own float32 chain is (golden G17, recorded from the
reference: here its rows are sent through this kernel by switching the GEMM form off); larger arrays against the oracle's
float64 chain of the promoted values, every kernel variant (antenna strides 64 / 128 / run time, the unrolled batch of 8,
rows and groups, feed rotation, Gaussian shapes).
"""
import ctypes

import numpy as np
import pytest

import oracle
from codex_africanus_amd import rime
from codex_africanus_amd.rime import fused
from test_gpu_fused import _problem, _scale
from test_gpu_fused_gemm import _decomposable
from test_gpu_fused_gemm_c64 import G17, _case, _call, _single, _chain64, _call_s

pytestmark = pytest.mark.gpu


def _rows_single(d):
    """a test_gpu_fused problem (uvw drawn per row: no antenna coordinates reproduce them) in single precision"""
    nant = d["pa"].shape[1]
    s = _single(dict(d, ant_xyz=np.zeros((d["ntime"], nant, 3))))
    s["uvw"] = d["uvw"].astype(np.float32)
    return s


def _tol(s):
    """float32 arithmetic on sums of nsrc terms: a few 1e-6 of sum_s |X_s| times the beam's gain (|V_ij| <= (sum_a |E_ia|)
    (sum_b |E_jb|) max |X_ab|); the phases are double and exact to the float32 inputs"""
    gain = float(np.abs(s["beam"]).sum(axis=-1).max()) ** 2
    return 1e-5 * max(gain, 1.0)


@pytest.fixture
def spy(monkeypatch):
    """the C-ABI entries a front-end call went through"""
    from codex_africanus_amd import _lib
    seen = []
    real = _lib.call

    def call(name, *a):
        seen.append(name)
        return real(name, *a)
    monkeypatch.setattr(_lib, "call", call)
    return seen


@pytest.mark.parametrize("name", ["a", "b"])
@pytest.mark.parametrize("feed", [False, True])
def test_closer_to_the_float64_chain_than_the_reference_float32_chain(name, feed, monkeypatch, spy):
    monkeypatch.setenv("AFHIP_FUSED_GEMM", "0")
    d = _case(name)
    kw = {}
    if feed:
        kw["feed_rotation"] = rime.feed_rotation(d["parallactic_angles"], "linear")
    got = _call(d, **kw)
    assert "af_fused_predict_c64" in spy and "af_fused_predict_antennas_c64" not in spy
    tag = "_feed" if feed else ""
    ref32, truth = G17["%s_vis32%s" % (name, tag)], G17["%s_vis64%s" % (name, tag)]
    assert got.dtype == np.complex64 and got.shape == ref32.shape
    peak = np.abs(truth).max()
    e_ours, e_ref = np.abs(got - truth).max() / peak, np.abs(ref32 - truth).max() / peak
    assert e_ours < 0.5 * e_ref, (e_ours, e_ref)
    assert e_ours < 2e-5, e_ours


# antenna counts on every Jones-stride variant: run-time stride (<= 32 and > 128 antennas, odd counts), the compile-time
# strides 64 (23 sources: the unrolled batch of 8; 5 sources: the rolled one) and 128; rows spanning several timesteps
@pytest.mark.parametrize("grouped", [True, False])
@pytest.mark.parametrize("nant, nrow, nsrc", [(7, 300, 23), (5, 37, 3), (12, 1000, 40), (40, 1700, 23), (64, 2100, 23), (64, 2100, 5),
                                             (70, 2600, 11), (128, 9000, 9), (150, 12000, 7)])
def test_against_the_float64_chain_on_rows_that_do_not_decompose(nant, nrow, nsrc, grouped, spy):
    d = _problem(3, nrow, 6, nsrc, nant)
    s = _rows_single(d)
    plan = fused.fused_plan(s["time_index"], s["ant1"], s["ant2"], nant, grouped=grouped, uvw=s["uvw"], single=True)
    assert not plan.decomposable and (plan.groups is not None) == grouped
    got = _call_s(s, plan=plan)
    assert spy.count("af_fused_predict_c64") == 1
    assert got.dtype == np.complex64
    truth = _chain64(s)
    assert np.abs(got - truth).max() < _tol(s) * _scale(d), (np.abs(got - truth).max() / _scale(d), _tol(s))
    if grouped:
        # without an explicit plan: the call plans for itself and takes the same route
        assert np.array_equal(_call_s(s), got)


def test_feed_rotation_gaussian_shapes_and_casa_convention(spy):
    nant = 19
    d = _problem(11, 1500, 5, 17, nant)
    s = _rows_single(d)
    rng = np.random.default_rng(3)
    fr = rime.feed_rotation(s["pa"], "circular")
    assert fr.dtype == np.complex64
    gs = np.zeros((17, 3), np.float32)
    gs[::2] = np.stack([rng.uniform(1e-4, 4e-4, 9), rng.uniform(5e-5, 1e-4, 9), rng.uniform(0, np.pi, 9)], axis=1).astype(np.float32)
    p = lambda a: a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    for kw in ({"feed_rotation": fr}, {"gauss_shape": gs}, {"feed_rotation": fr, "gauss_shape": gs, "convention": "casa"}):
        got = _call_s(s, **kw)
        assert got.dtype == np.complex64
        # the double-precision kernel on the promoted values (itself checked against the oracle in tests/test_gpu_fused.py)
        kw64 = {k: (p(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
        want = rime.fused_predict_vis(s["time_index"], s["ant1"], s["ant2"], p(s["lm"]), p(s["uvw"]), p(s["frequency"]), p(s["X"]),
                                      p(s["beam"]), p(s["extents"]), p(s["beam_freq_map"]), p(s["pa"]), p(s["pe"]), p(s["as"]), **kw64)
        assert want.dtype == np.complex128
        assert np.abs(got - want).max() < _tol(s) * _scale(d), (kw.keys(), np.abs(got - want).max() / _scale(d))
    assert spy.count("af_fused_predict_c64") == 3 and spy.count("af_fused_predict_c128") == 3


def test_non_hermitian_brightness_on_decomposable_rows(spy):
    """the GEMM form needs Hermitian brightness matrices (rime.fused._hermitian): anything else takes this kernel"""
    nant = 24
    d = _decomposable(_problem(5, 2000, 4, 13, nant), nant)
    s = _single(d)
    rng = np.random.default_rng(8)
    s["X"] = (s["X"] + 0.2 * (rng.standard_normal(s["X"].shape) + 1j * rng.standard_normal(s["X"].shape))).astype(np.complex64)
    got = _call_s(s)
    assert "af_fused_predict_c64" in spy and "af_fused_predict_antennas_c64" not in spy
    d2 = dict(d, X=s["X"].astype(np.complex128))
    assert np.abs(got - _chain64(s)).max() < _tol(s) * _scale(d2)


def test_more_channels_than_one_group_of_beam_planes_and_dies():
    nant = 9
    d = _problem(4, 300, 70, 5, nant)
    s = _rows_single(d)
    got = _call_s(s)
    assert got.dtype == np.complex64 and got.shape == (300, 70, 2, 2)
    assert np.abs(got - _chain64(s)).max() < _tol(s) * _scale(d)
    rng = np.random.default_rng(5)
    shp = (d["ntime"], nant, 70, 2, 2)
    die = (1.0 + 0.1 * rng.standard_normal(shp) + 0.1j * rng.standard_normal(shp)).astype(np.complex64)
    bvis = (0.1 * (rng.standard_normal(got.shape) + 1j * rng.standard_normal(got.shape))).astype(np.complex64)
    full = _call_s(s, die1_jones=die, base_vis=bvis, die2_jones=die)
    assert full.dtype == np.complex64
    assert np.array_equal(full, oracle.predict_vis(s["time_index"], s["ant1"], s["ant2"], None, got[None], None, die, bvis, die))


def test_empty_calls():
    """no rows -> (0, chan, 2, 2); no sources -> zeros: both complex64"""
    nant = 12
    s = _rows_single(_problem(7, 600, 4, 9, nant))
    e = dict(s)
    for k in ("time_index", "ant1", "ant2", "uvw"):
        e[k] = s[k][:0]
    out = _call_s(e)
    assert out.dtype == np.complex64 and out.shape == (0, 4, 2, 2)
    z = dict(s, lm=s["lm"][:0], X=s["X"][:0])
    out = _call_s(z)
    assert out.dtype == np.complex64 and out.shape == (600, 4, 2, 2) and not out.any()


def test_stale_plan_is_refused():
    nant = 12
    s = _rows_single(_problem(7, 600, 4, 9, nant))
    plan = fused.fused_plan(s["time_index"], s["ant1"], s["ant2"], nant, uvw=s["uvw"], single=True)
    other = dict(s)
    other["ant1"], other["ant2"] = s["ant2"].copy(), s["ant1"].copy()
    with pytest.raises(ValueError, match="stale plan"):
        _call_s(other, plan=plan)


def test_device_resident_tensors():
    import torch
    nant = 24
    s = _rows_single(_problem(9, 2000, 5, 15, nant))
    host = _call_s(s)
    t = {k: (torch.from_numpy(np.ascontiguousarray(v)).cuda() if isinstance(v, np.ndarray) else v) for k, v in s.items()}
    dev = _call_s(t)
    assert dev.dtype == torch.complex64 and dev.is_cuda
    assert np.array_equal(dev.cpu().numpy(), host)


def test_the_entry_through_the_c_abi():
    """af_fused_predict_c64 called directly (device pointers, explicit workspace): the same bits as the front-end, its status
    codes and messages, the empty cases"""
    import torch
    from codex_africanus_amd import _lib
    lib = _lib.load()
    nant = 12
    s = _rows_single(_problem(7, 600, 4, 9, nant))
    want = _call_s(s)
    plan = fused.fused_plan(s["time_index"], s["ant1"], s["ant2"], nant, uvw=s["uvw"], single=True)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    v = {k: t(s[k]) for k in ("lm", "uvw", "frequency", "X", "beam", "extents", "beam_freq_map", "pa", "pe", "as")}
    items, groups, a1, a2 = t(plan.items), t(plan.groups), t(plan.antenna1), t(plan.antenna2)
    nsrc, nchan, nrow = 9, 4, 600
    lw, mh, nud = s["beam"].shape[:3]
    ntime = s["pa"].shape[0]
    ws_bytes = int(lib.af_fused_predict_c64_workspace_bytes(nsrc, nchan, lw, mh, nud))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = torch.full((nrow, nchan, 2, 2), complex(float("nan"), 0.0), dtype=torch.complex64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def args(**kw):
        return [P(items), kw.get("nitems", plan.n_items), P(a1), P(a2), P(groups), nrow, kw.get("lm", P(v["lm"])), P(v["uvw"]),
                P(v["frequency"]), P(v["X"]), kw.get("nsrc", nsrc), nchan, P(v["beam"]), kw.get("lw", lw), mh, nud, P(v["extents"]),
                P(v["beam_freq_map"]), P(v["pa"]), ntime, kw.get("nant", nant), P(v["pe"]), P(v["as"]), None, None,
                kw.get("conv", _lib.CONVENTION["fourier"]), P(out), kw.get("ws", P(ws)), kw.get("wsb", ws_bytes), stream]

    assert lib.af_fused_predict_c64(*args()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    assert lib.af_fused_predict_c64(*args(conv=0)) == 1 and b"convention not in ('fourier', 'casa')" in lib.af_last_error()
    assert lib.af_fused_predict_c64(*args(wsb=64)) == 1 and b"workspace too small" in lib.af_last_error()
    assert lib.af_fused_predict_c64(*args(ws=ctypes.c_void_p(ws.data_ptr() + 8))) == 1 and b"aligned" in lib.af_last_error()
    assert lib.af_fused_predict_c64(*args(lm=None)) == 1 and b"NULL" in lib.af_last_error()
    assert lib.af_fused_predict_c64(*args(nant=665)) == 1 and b"664 antennas" in lib.af_last_error()
    assert lib.af_fused_predict_c64(*args(lw=1)) == 1 and b"must be >= 2" in lib.af_last_error()
    for kw in ({"nsrc": 0}, {"nitems": 0}):         # nothing to add: zeros
        out.fill_(complex(float("nan"), 0.0))
        assert lib.af_fused_predict_c64(*args(**kw)) == 0
        torch.cuda.synchronize()
        assert float(out.abs().max()) == 0.0
    assert lib.af_fused_predict_c64(*args()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)


def test_sky_model_in_single_precision_takes_the_single_precision_kernels(spy, monkeypatch):
    """stokes / spi / ref_freq with every input single precision: the brightness array is made first (complex64, the
    reference's own order of work, rime/examples/predict.py:494-498) and the call runs on the single-precision kernels --
    the GEMM form on Measurement-Set rows, this kernel on any others"""
    monkeypatch.setenv("AFHIP_GEMM_MIN_FILL", "0")
    nant, nsrc = 24, 13
    rng = np.random.default_rng(12)
    stokes = np.stack([rng.lognormal(0, 1, nsrc)] + [0.1 * rng.standard_normal(nsrc) for _ in range(3)], axis=1).astype(np.float32)
    spi = rng.uniform(-1.0, 0.2, (nsrc, 2, 4)).astype(np.float32)
    rf = rng.uniform(0.9e9, 1.5e9, nsrc).astype(np.float32)
    p = lambda a: a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    for decomposable, entry in ((True, "af_fused_predict_antennas_c64"), (False, "af_fused_predict_c64")):
        d = _problem(5, 2000, 4, nsrc, nant)
        s = _single(_decomposable(d, nant)) if decomposable else _rows_single(d)
        del spy[:]
        got = rime.fused_predict_vis(s["time_index"], s["ant1"], s["ant2"], s["lm"], s["uvw"], s["frequency"], None, s["beam"],
                                     s["extents"], s["beam_freq_map"], s["pa"], s["pe"], s["as"], stokes=stokes, spi=spi, ref_freq=rf)
        assert got.dtype == np.complex64 and entry in spy, spy
        st = oracle.spectral_model(p(stokes), p(spi), p(rf), p(s["frequency"]), base=0)
        I, Q, U, V = (st[..., k] for k in range(4))           # noqa: E741
        X = np.stack([I + Q, U + 1j * V, U - 1j * V, I - Q], axis=-1).reshape(nsrc, 4, 2, 2)
        s2 = dict(s, X=X.astype(np.complex64))
        d2 = dict(d, X=X)
        tol = _tol(s2) if not decomposable else max(_tol(s2), __import__("test_gpu_fused_gemm_c64")._tol(s2, d2))
        assert np.abs(got - _chain64(s2)).max() < tol * _scale(d2) + 1e-6 * _scale(d2)


def test_a_plan_at_single_precision_tolerance_is_not_used_in_double(spy, monkeypatch):
    """float32 rows of a Measurement Set decompose only at their own precision (fused_plan(..., single=True)); a call that
    computes in double -- here: the single-precision kernels switched off -- must not take the GEMM form with such a plan"""
    monkeypatch.setenv("AFHIP_GEMM_MIN_FILL", "0")
    monkeypatch.setenv("AFHIP_FUSED_C64", "0")
    nant = 12
    d = _decomposable(_problem(7, 600, 4, 9, nant), nant)
    s = _single(d)
    plan = fused.fused_plan(s["time_index"], s["ant1"], s["ant2"], nant, uvw=s["uvw"], single=True)
    assert plan.decomposable and plan.single_tol
    out = _call_s(s, plan=plan)
    assert out.dtype == np.complex64 and "af_fused_predict_c128" in spy and "af_fused_predict_antennas_c128" not in spy
    truth = _chain64(s)
    assert np.abs(out - truth).max() <= 6.1e-8 * np.abs(truth).max() + 1e-9 * _scale(d)      # double, rounded once


def test_without_a_beam_the_single_precision_direct_transform(spy):
    """no DDEs, every input single precision: sum_s K X_s by af_im_to_vis_f32 (complex image, phase_delay's clamped n -- one
    source here sits outside the unit disc --, phases in double); one double array among the inputs: the double transform"""
    nant = 12
    d = _problem(7, 600, 4, 9, nant)
    s = _rows_single(d)
    s["lm"] = s["lm"].copy()
    s["lm"][3] = (0.8, 0.7)                                   # l^2 + m^2 > 1: n = -1 (clamped), not NaN
    out = rime.fused_predict_vis(s["time_index"], s["ant1"], s["ant2"], s["lm"], s["uvw"], s["frequency"], s["X"])
    assert out.dtype == np.complex64 and "af_im_to_vis_f32" in spy and "af_im_to_vis_f64" not in spy
    p = lambda a: a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    phase = oracle.phase_delay(p(s["lm"]), p(s["uvw"]), p(s["frequency"]))
    truth = np.einsum("srf,sfij->rfij", phase, p(s["X"]))
    assert np.isfinite(truth).all() and np.abs(out - truth).max() < 1e-5 * _scale(d), np.abs(out - truth).max() / _scale(d)
    del spy[:]
    out64 = rime.fused_predict_vis(s["time_index"], s["ant1"], s["ant2"], p(s["lm"]), s["uvw"], s["frequency"], s["X"])
    assert out64.dtype == np.complex128 and "af_im_to_vis_f64" in spy
    assert np.abs(out64 - truth).max() < 1e-9 * _scale(d)
