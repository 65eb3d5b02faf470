"""
No result may depend on what the allocator hands out.  Every call draws its scratch and result buffers from a pool (host
mode) or torch's caching allocator (device mode): memory that holds whatever ran before, including other processes' data.
With ``_device.POISON`` every such block starts as 0xFF bytes (NaN): a kernel that reads a cell before writing it, or
leaves a result cell unwritten, changes the result.  Each entry point below must return the same bits with and without
it, in both modes (numpy in -> numpy out, tensors in -> tensor out).
"""
import numpy as np
import pytest

from codex_africanus_amd import _device, dft, rime
from codex_africanus_amd.gridding.wgridder import model
from codex_africanus_amd.rime import fused
from test_gpu_fused import _problem
from test_gpu_fused_gemm import _call, _decomposable

pytestmark = pytest.mark.gpu


def _both(monkeypatch, f):
    """f() without poison, and with every block starting as 0xFF bytes (NaN / -1), 0x01 bytes (small positive integers,
    denormal doubles) and 0x40 bytes (doubles ~ 32, floats ~ 3): what reads as "no entry" in one pattern is data in another"""
    monkeypatch.setattr(_device, "POISON", False)
    clean = f()
    dirty = []
    for byte in (0xFF, 0x01, 0x40):
        monkeypatch.setattr(_device, "POISON", True)
        monkeypatch.setattr(_device, "POISON_BYTE", byte)
        dirty.append(f())
    return clean, dirty


def _same(a, bs):
    a = a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
    assert not np.isnan(a.view(np.float64) if a.dtype.kind == "c" else a).any()
    for b in bs:
        b = b.cpu().numpy() if hasattr(b, "cpu") else np.asarray(b)
        assert np.array_equal(a, b)


def _tensors(d, keys):
    import torch
    d = dict(d)
    for k in keys:
        d[k] = torch.from_numpy(np.ascontiguousarray(d[k])).cuda()
    return d


KEYS = ("time_index", "ant1", "ant2", "lm", "uvw", "frequency", "X", "beam", "extents", "beam_freq_map", "pa", "pe", "as")


@pytest.mark.parametrize("nant, nrow", [(7, 300), (64, 4100), (100, 5200)])
@pytest.mark.parametrize("device", (False, True))
def test_fused_predict_gemm_form(monkeypatch, nant, nrow, device):
    monkeypatch.setenv("AFHIP_GEMM_MIN_FILL", "0")
    d = _decomposable(_problem(3, nrow, 6, 23, nant), nant, keep=0.9)
    plan = fused.fused_plan(d["time_index"], d["ant1"], d["ant2"], nant, uvw=d["uvw"])
    assert plan.decomposable
    if device:
        d = _tensors(d, KEYS)
    _same(*_both(monkeypatch, lambda: _call(d, plan=plan)))
    _same(*_both(monkeypatch, lambda: _call(d)))


@pytest.mark.parametrize("device", (False, True))
def test_fused_predict_lane_per_row(monkeypatch, device):
    d = _problem(3, 1500, 6, 23, 17)
    if device:
        d = _tensors(d, KEYS)
    _same(*_both(monkeypatch, lambda: _call(d)))


@pytest.mark.parametrize("device", (False, True))
def test_direct_transforms_and_the_chain(monkeypatch, device):
    rng = np.random.default_rng(3)
    nrow, nchan, nsrc = 3000, 5, 37
    uvw = rng.standard_normal((nrow, 3)) * 100
    lm = rng.standard_normal((nsrc, 2)) * 0.01
    freq = np.linspace(1e9, 2e9, nchan)
    image = rng.standard_normal((nsrc, nchan, 4))
    vis = rng.standard_normal((nrow, nchan, 4)) + 1j * rng.standard_normal((nrow, nchan, 4))
    arrs = dict(uvw=uvw, lm=lm, freq=freq, image=image, vis=vis)
    if device:
        arrs = _tensors(arrs, arrs.keys())
    _same(*_both(monkeypatch, lambda: dft.im_to_vis(arrs["image"], arrs["uvw"], arrs["lm"], arrs["freq"])))
    _same(*_both(monkeypatch, lambda: dft.vis_to_im(arrs["vis"], arrs["uvw"], arrs["lm"], arrs["freq"],
                                                    np.zeros((nrow, nchan, 4), bool) if not device else
                                                    __import__("torch").zeros((nrow, nchan, 4), dtype=__import__("torch").bool, device="cuda"))))
    _same(*_both(monkeypatch, lambda: rime.phase_delay(arrs["lm"], arrs["uvw"], arrs["freq"])))


@pytest.mark.parametrize("nrow", (900, 40000))
def test_wgridder_model(monkeypatch, nrow):
    from test_gpu_wgridder import _case
    cell, freq, uvw, fbi, fbc, image = _case(24, 20, 20.0, nrow, 3, 1, seed=31)
    _same(*_both(monkeypatch, lambda: model(uvw, freq, image, fbi, fbc, cell, epsilon=1e-6)))


@pytest.mark.parametrize("corrs", [(2, 2), (1,)])
@pytest.mark.parametrize("device", (False, True))
def test_predict_vis_apply_gains_and_beam_cube(monkeypatch, corrs, device):
    from test_gpu_predict_tile import _case as tile_case
    rng = np.random.default_rng(11)
    d = tile_case(rng, 2500, 37, 4, 7, 21, corrs)
    if device:
        d = _tensors(d, d.keys())
    _same(*_both(monkeypatch, lambda: rime.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], d["coh"], d["dde"], d["die"], d["bvis"], d["die"])))
    _same(*_both(monkeypatch, lambda: rime.predict_vis(d["ti"], d["a1"], d["a2"], None, d["coh"], None, None, None, None)))
    _same(*_both(monkeypatch, lambda: rime.apply_gains(d["ti"], d["a1"], d["a2"], d["die"], d["bvis"], d["die"])))
    if corrs == (2, 2):
        p = _problem(3, 300, 6, 23, 7)
        if device:
            p = _tensors(p, ("beam", "extents", "beam_freq_map", "lm", "pa", "pe", "as", "frequency"))
        _same(*_both(monkeypatch, lambda: rime.beam_cube_dde(p["beam"], p["extents"], p["beam_freq_map"], p["lm"], p["pa"], p["pe"],
                                                             p["as"], p["frequency"])))


@pytest.mark.parametrize("nrow", (900, 40000))
def test_wgridder_dirty(monkeypatch, nrow):
    """(gridding with atomics: the sums' order varies from run to run, so to rounding, not bit for bit)"""
    from codex_africanus_amd.gridding.wgridder import dirty
    from test_gpu_wgridder import _case
    cell, freq, uvw, fbi, fbc, _ = _case(24, 20, 20.0, nrow, 3, 1, seed=31)
    rng = np.random.default_rng(5)
    vis = rng.standard_normal((nrow, 3)) + 1j * rng.standard_normal((nrow, 3))
    clean, dirties = _both(monkeypatch, lambda: dirty(uvw, freq, vis, fbi, fbc, 24, 20, cell, epsilon=1e-6))
    for d in dirties:
        assert np.abs(d - clean).max() <= 1e-11 * np.abs(clean).max()
