"""
predict_vis with DDE terms on the (row block, chan tile) kernel (csrc/af_predict_vis.hip, round 3): the per-antenna
Jones terms of a block's timestep(s) are staged in LDS once per source and shared by every baseline of the block
(africanus/rime/predict.py:199-212 is the loop it replaces).  Bit-exactness is the contract, as for the lane-per-cell
kernel: every case is compared with the CPU oracle (itself bit-identical to the reference, tests/test_oracle_golden.py)
with assert_array_equal.  Shapes are chosen so that the tile kernel runs (>= 65536 cells, dde1 is dde2) and so that
its special paths are hit: blocks inside one timestep, blocks straddling two, blocks spanning more timesteps than the
stage holds (per-lane gathers inside the tile kernel), unsorted time, chan tiles sticking out of the band, rows past
the end of the last block.  Also the index guard.
"""
import numpy as np
import pytest
from numpy.testing import assert_array_equal

import oracle
from codex_africanus_amd import rime

pytestmark = pytest.mark.gpu


def _case(rng, nrow, nchan, nsrc, nant, rows_per_time, corrs, dtype=np.complex128, idx=np.int32, sort_time=True,
          offset=0):
    ntime = -(-nrow // rows_per_time)
    ti = (np.arange(nrow) // rows_per_time)
    if not sort_time:
        ti = rng.permutation(ti)
    ti = (ti + offset).astype(idx)
    a1 = rng.integers(0, nant, nrow).astype(idx)
    a2 = rng.integers(0, nant, nrow).astype(idx)

    def c(*shape):
        return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)

    return dict(ti=ti, a1=a1, a2=a2, dde=c(nsrc, ntime, nant, nchan, *corrs), coh=c(nsrc, nrow, nchan, *corrs),
                die=c(ntime, nant, nchan, *corrs), bvis=c(nrow, nchan, *corrs))


SHAPES = [
    # nrow, nchan, nsrc, nant, rows per timestep
    (4100, 64, 3, 64, 2016),      # the measured shape in small: blocks inside a timestep and straddling two
    (2500, 37, 4, 7, 21),         # 7 antennas: a 128-row block spans 6-7 timesteps -> gathers inside the tile kernel
    (1030, 70, 2, 27, 351),       # 27 antennas; 70 channels: the last tile sticks out of the band
    (1500, 48, 5, 64, 100000),    # one timestep
]


@pytest.mark.parametrize("corrs", [(2, 2), (2,), (1,)])
@pytest.mark.parametrize("shape", SHAPES)
def test_tile_kernel_bit_exact_c128(shape, corrs):
    nrow, nchan, nsrc, nant, rpt = shape
    d = _case(np.random.default_rng(nrow + len(corrs)), nrow, nchan, nsrc, nant, rpt, corrs)
    for coh, die, bvis in ((d["coh"], None, None), (d["coh"], d["die"], d["bvis"]), (None, None, None), (None, d["die"], None)):
        got = rime.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], coh, d["dde"], die, bvis, die)
        ref = oracle.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], coh, d["dde"], die, bvis, die)
        assert_array_equal(got, ref)


@pytest.mark.parametrize("corrs", [(2, 2), (2,), (1,)])
def test_tile_kernel_bit_exact_c64_int64_offset_unsorted(corrs):
    rng = np.random.default_rng(11)
    for nchan, sort_time in ((64, True), (40, False)):
        d = _case(rng, 3000, nchan, 3, 16, 120, corrs, dtype=np.complex64, idx=np.int64, sort_time=sort_time, offset=10)
        got = rime.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], d["coh"], d["dde"], d["die"], d["bvis"], d["die"])
        ref = oracle.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], d["coh"], d["dde"], d["die"], d["bvis"], d["die"])
        assert got.dtype == np.complex64
        assert_array_equal(got, ref)


@pytest.mark.parametrize("corrs", [(2, 2), (2,), (1,)])
@pytest.mark.parametrize("shape", SHAPES)
def test_apply_gains_and_die_only_calls_on_the_tile_kernel(shape, corrs, monkeypatch):
    """apply_gains (africanus/rime/predict.py:623-647) = predict_vis with DIE terms and base_vis only: the same kernel
    with no sources, the gains of the block's timesteps staged in LDS once (one stage, 2 CT channels wide).  Bit-equal
    to the oracle; with distinct die1 / die2 arrays (lane kernel) and with AFHIP_PREDICT_DIE_LDS=0 (gathers) too;
    complex64 and int64 indices."""
    nrow, nchan, _, nant, rpt = shape
    rng = np.random.default_rng(nrow + 7 * len(corrs))
    d = _case(rng, nrow, nchan, 1, nant, rpt, corrs)
    ref = oracle.predict_vis(d["ti"], d["a1"], d["a2"], None, None, None, d["die"], d["bvis"], d["die"])
    assert_array_equal(rime.apply_gains(d["ti"], d["a1"], d["a2"], d["die"], d["bvis"], d["die"]), ref)
    assert_array_equal(rime.predict_vis(d["ti"], d["a1"], d["a2"], None, None, None, d["die"], d["bvis"], d["die"]), ref)
    other = d["die"][:, ::-1].copy()
    assert_array_equal(rime.apply_gains(d["ti"], d["a1"], d["a2"], d["die"], d["bvis"], other),
                       oracle.predict_vis(d["ti"], d["a1"], d["a2"], None, None, None, d["die"], d["bvis"], other))
    d32 = _case(rng, nrow, nchan, 1, nant, rpt, corrs, dtype=np.complex64, idx=np.int64, offset=3)
    assert_array_equal(rime.apply_gains(d32["ti"], d32["a1"], d32["a2"], d32["die"], d32["bvis"], d32["die"]),
                       oracle.predict_vis(d32["ti"], d32["a1"], d32["a2"], None, None, None, d32["die"], d32["bvis"], d32["die"]))
    # round 4: such calls (DIE terms + base_vis, 32- or 64-byte cells) run lane = cell with cooperative IO
    # (apply_dies_coop_kernel); AFHIP_APPLY_COOP=0 sends them to the tile kernel as in round 3, and with
    # AFHIP_PREDICT_DIE_LDS=0 on top to the per-lane gathers: same bits everywhere
    monkeypatch.setenv("AFHIP_APPLY_COOP", "0")
    assert_array_equal(rime.apply_gains(d["ti"], d["a1"], d["a2"], d["die"], d["bvis"], d["die"]), ref)
    assert_array_equal(rime.apply_gains(d["ti"], d["a1"], d["a2"], d["die"], d["bvis"], other),
                       oracle.predict_vis(d["ti"], d["a1"], d["a2"], None, None, None, d["die"], d["bvis"], other))
    monkeypatch.setenv("AFHIP_PREDICT_DIE_LDS", "0")
    assert_array_equal(rime.apply_gains(d["ti"], d["a1"], d["a2"], d["die"], d["bvis"], d["die"]), ref)


@pytest.mark.parametrize("corrs", [(2, 2), (2,), (1,)])
@pytest.mark.parametrize("dtype, idx", [(np.complex128, np.int32), (np.complex64, np.int64)])
def test_coherency_stream_without_dde_terms_lane_per_cell_cooperative(corrs, dtype, idx, monkeypatch):
    """Calls without DDE terms (round 4, predict_cell_coop_kernel): the coherency stream summed over sources, with and
    without base_vis / DIE terms (same and distinct arrays), 32- and 64-byte cells through the wave transposes, 16- and
    8-byte cells on the lane kernel as before; a row count that leaves the last wave partly empty; bit-equal to the oracle
    and to round 3's kernels (AFHIP_PREDICT_COOP=0)."""
    rng = np.random.default_rng(5 + len(corrs))
    nrow, nchan, nsrc, nant = 1111, 61, 5, 9
    d = _case(rng, nrow, nchan, nsrc, nant, 36, corrs, dtype=dtype, idx=idx, offset=4)
    other = d["die"][:, ::-1].copy()
    combos = ((None, None, None), (None, d["bvis"], None), (d["die"], d["bvis"], d["die"]), (d["die"], None, d["die"]),
              (d["die"], d["bvis"], other))
    got = []
    for die1, bvis, die2 in combos:
        out = rime.predict_vis(d["ti"], d["a1"], d["a2"], None, d["coh"], None, die1, bvis, die2)
        assert out.dtype == dtype
        assert_array_equal(out, oracle.predict_vis(d["ti"], d["a1"], d["a2"], None, d["coh"], None, die1, bvis, die2))
        got.append(out)
    monkeypatch.setenv("AFHIP_PREDICT_COOP", "0")
    for (die1, bvis, die2), out in zip(combos, got):
        assert_array_equal(rime.predict_vis(d["ti"], d["a1"], d["a2"], None, d["coh"], None, die1, bvis, die2), out)


@pytest.mark.parametrize("seed", range(10))
def test_calls_without_dde_terms_random_shapes_bit_exact(seed):
    """Random extents above the 65536-cell threshold of the cooperative lane-per-cell kernel: any subset of
    {coherencies, DIE terms (one array or two), base_vis}, every correlation layout, both precisions and index types,
    time indices in any order and offset, last wave partly empty -- bit-equal to the oracle."""
    rng = np.random.default_rng(900 + seed)
    nchan = int(rng.integers(17, 90))
    nrow = int(rng.integers(65536 // nchan + 1, 65536 // nchan + 900))
    nant, nsrc = int(rng.integers(2, 40)), int(rng.integers(1, 4))
    corrs = [(2, 2), (2,), (1,)][seed % 3]
    dtype, idx = [(np.complex128, np.int32), (np.complex64, np.int64)][(seed // 3) % 2]
    d = _case(rng, nrow, nchan, nsrc, nant, int(rng.integers(20, 500)), corrs, dtype=dtype, idx=idx, sort_time=bool(seed % 2),
              offset=int(rng.integers(0, 9)))
    have_coh, have_die, have_bv = bool(seed % 4 != 1), bool(seed % 5 != 0), bool(seed % 2 == 0 or seed % 4 == 1)
    die2 = d["die"][:, ::-1].copy() if seed % 3 == 1 else d["die"]
    args = (d["ti"], d["a1"], d["a2"], None, d["coh"] if have_coh else None, None, d["die"] if have_die else None,
            d["bvis"] if have_bv else None, die2 if have_die else None)
    if not (have_coh or have_die or have_bv):
        args = args[:7] + (d["bvis"],) + args[8:]
    out = rime.predict_vis(*args)
    assert out.dtype == dtype
    assert_array_equal(out, oracle.predict_vis(*args))


def test_tile_and_lane_kernels_agree_and_distinct_dde_arrays_fall_back(monkeypatch):
    """dde1 is not dde2 (legal, rare): the stage holds ONE array, so the call takes the lane-per-cell kernel; both
    kernels give the oracle's bits.  AFHIP_PREDICT_TILE=0 forces the lane kernel for the A/B."""
    d = _case(np.random.default_rng(5), 4100, 64, 3, 64, 2016, (2, 2))
    other = d["dde"][:, :, ::-1].copy()
    got = rime.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], d["coh"], other, None, None, None)
    assert_array_equal(got, oracle.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], d["coh"], other, None, None, None))
    tile = rime.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], d["coh"], d["dde"], None, None, None)
    monkeypatch.setenv("AFHIP_PREDICT_TILE", "0")
    lane = rime.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], d["coh"], d["dde"], None, None, None)
    assert_array_equal(tile, lane)


def test_device_resident_and_full_chip_shape():
    """torch tensors in, tensor out, on a shape that fills the chip (131072 rows x 64 chan x 4 sources, 64 antennas)"""
    import torch
    dev = torch.device("cuda:0")
    d = _case(np.random.default_rng(3), 131072, 64, 4, 64, 2016, (2, 2))
    t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
    got = rime.predict_vis(t["ti"], t["a1"], t["a2"], t["dde"], t["coh"], t["dde"], t["die"], t["bvis"], t["die"])
    rows = np.r_[0:300, 2000:2100, 65000:65300, 131072 - 200:131072]
    ref = oracle.predict_vis(d["ti"][rows], d["a1"][rows], d["a2"][rows], d["dde"], d["coh"][:, rows], d["dde"],
                             d["die"], d["bvis"][rows], d["die"])
    # the oracle normalises time_index by the minimum of the rows it is given: rows 0.. are among them
    assert_array_equal(got[torch.from_numpy(rows).to(dev)].cpu().numpy(), ref)


@pytest.mark.parametrize("big", [False, True])
def test_index_guard_device_mode(big):
    """Device-resident index tensors cannot be checked on the host without a synchronisation; the kernels clamp the
    read, write NaN into the rows concerned and flag the call: the NEXT call into the package (or check_status)
    raises ValueError.  Good rows are unaffected.  Both kernels (lane per cell: small call; tile: big call)."""
    import torch
    import codex_africanus_amd as pkg
    dev = torch.device("cuda:0")
    nrow = 3000 if big else 200
    d = _case(np.random.default_rng(8), nrow, 64, 2, 16, 120, (2, 2))
    good = oracle.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], d["coh"], d["dde"], None, None, None)
    a1 = d["a1"].copy()
    a1[17] = 16                       # one antenna past the end
    ti = d["ti"].copy()
    ti[nrow - 5] += 4000              # far beyond ntime
    t = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
    pkg.check_status()
    out = rime.predict_vis(torch.from_numpy(ti).to(dev), torch.from_numpy(a1).to(dev), t["a2"], t["dde"], t["coh"],
                           t["dde"], None, None, None).cpu().numpy()
    assert np.isnan(out[17]).all() and np.isnan(out[nrow - 5]).all()
    keep = np.ones(nrow, bool)
    keep[[17, nrow - 5]] = False
    assert_array_equal(out[keep], good[keep])
    with pytest.raises(ValueError, match="antenna1 / antenna2"):
        pkg.check_status()
    pkg.check_status()                                        # reported once
    # ... and without an explicit check the next call into the package raises
    rime.predict_vis(t["ti"], torch.from_numpy(a1).to(dev), t["a2"], t["dde"], t["coh"], t["dde"], None, None, None)
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="outside the 16 antennas"):
        rime.predict_vis(t["ti"], t["a1"], t["a2"], t["dde"], t["coh"], t["dde"], None, None, None)
    pkg.check_status()


def test_index_errors_survive_a_full_slot_table():
    """with every status slot in flight, watch() drained the table to make room and threw the drained calls'
    reports away.  More flagged calls than slots, no poll in between: every one of them is reported."""
    import torch
    from codex_africanus_amd import _device
    dev = torch.device("cuda:0")
    d = _device._DeferredStatus()
    d.SLOTS = 4
    ws = torch.zeros(64, dtype=torch.uint8, device=dev)
    ws[8:12].view(torch.int32).fill_(3)
    for k in range(11):
        d.watch(ws, 8, lambda flags, k=k: "call %d flags %d" % (k, flags))
    with pytest.raises(ValueError) as e:
        d.poll(wait=True)
    for k in range(11):
        assert "call %d flags 3" % k in str(e.value)
    d.poll(wait=True)                                         # reported once
    assert sorted(d._free) == [0, 1, 2, 3] and not d._pending and not d._carry


def test_index_guard_through_the_c_abi():
    """Host wrapper bypassed: the status word at workspace + 8 and the NaN rows, dies-only call."""
    import ctypes
    import torch
    from codex_africanus_amd import _lib
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2)
    nrow, nchan, ntime, nant = 50, 8, 3, 5
    ti = torch.from_numpy((np.arange(nrow) % ntime).astype(np.int32)).to(dev)
    a1 = torch.from_numpy(rng.integers(0, nant, nrow).astype(np.int32)).to(dev)
    a2n = rng.integers(0, nant, nrow).astype(np.int32)
    a2n[7] = -1
    a2 = torch.from_numpy(a2n).to(dev)
    die = torch.randn(ntime, nant, nchan, 2, 2, dtype=torch.complex128, device=dev)
    bv = torch.randn(nrow, nchan, 2, 2, dtype=torch.complex128, device=dev)
    out = torch.empty_like(bv)
    ws = torch.zeros(256, dtype=torch.uint8, device=dev)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    _lib.load()
    _lib.call("af_predict_vis_c128", P(ti), P(a1), P(a2), 4, nrow, None, None, None, P(die), P(bv), P(die), 0, ntime, nant,
              nchan, 4, _lib.AF_JONES_2X2, P(out), P(ws), 256, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    status = int(ws[8:12].view(torch.int32).item())
    assert status == _lib.AF_STATUS_ANTENNA
    o = out.cpu().numpy()
    assert np.isnan(o[7]).all() and np.isfinite(np.delete(o, 7, axis=0)).all()


@pytest.mark.parametrize("corrs", [(2, 2), (2,), (1,)])
@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_streamed_form_bit_exact(monkeypatch, corrs, dtype):
    """AFHIP_PREDICT_STREAM=1: coherencies through LDS as well, K sub-blocks per workgroup share one Jones copy (opt-in,
    measured alternative of the tile kernel): same bits as the oracle on blocks inside a timestep, straddling two,
    spanning more (per-lane gathers), band remainders and the array's partial last block"""
    monkeypatch.setenv("AFHIP_PREDICT_STREAM", "1")
    for shape in SHAPES:
        nrow, nchan, nsrc, nant, rpt = shape
        d = _case(np.random.default_rng(nrow), nrow, nchan, nsrc, nant, rpt, corrs, dtype=dtype)
        for die, bvis in ((None, None), (d["die"], d["bvis"])):
            got = rime.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], d["coh"], d["dde"], die, bvis, die)
            ref = oracle.predict_vis(d["ti"], d["a1"], d["a2"], d["dde"], d["coh"], d["dde"], die, bvis, die)
            assert_array_equal(got, ref)
