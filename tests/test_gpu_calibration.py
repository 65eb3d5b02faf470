"""GPU parity of the calibration consumers against the reference's golden vectors
(tests/golden/g9_calibration.npz) and the oracle: bit for bit, all four gain layouts."""
import numpy as np
import pytest
from numpy.testing import assert_array_equal

import oracle
from codex_africanus_amd.calibration.utils import corrupt_vis, residual_vis, correct_vis, chunkify_rows, check_type

pytestmark = pytest.mark.gpu
TAGS = ["dd1", "dd2", "diag", "full"]


@pytest.mark.parametrize("tag", TAGS)
def test_calibration_utils_golden_bit_exact(g9, tag):
    a = (g9["tbin_idx"], g9["tbin_counts"], g9["ant1"], g9["ant2"])
    tbi_before = g9["tbin_idx"].copy()
    vis = corrupt_vis(*a, g9[tag + "_jones"], g9[tag + "_model"])
    assert vis.shape == g9[tag + "_vis"].shape and vis.dtype == np.complex128
    assert_array_equal(vis, g9[tag + "_vis"])
    res = residual_vis(*a, g9[tag + "_jones"], g9[tag + "_data"], g9[tag + "_flag"], g9[tag + "_model"])
    assert_array_equal(res, g9[tag + "_residual"])
    j1 = np.ascontiguousarray(g9[tag + "_jones"][:, :, :, :1])
    cor = correct_vis(*a, j1, g9[tag + "_data"], g9[tag + "_flag"])
    assert_array_equal(cor, g9[tag + "_corrected"])
    assert_array_equal(g9["tbin_idx"], tbi_before)          # inputs untouched


@pytest.mark.parametrize("seed", range(8))
def test_calibration_utils_random_against_oracle(seed):
    """offset (dask-chunk style) bin starts, rows outside every bin, int32 / int64 indices, random flags"""
    rng = np.random.default_rng(seed)
    ntime, nant, nchan, ndir = int(rng.integers(1, 6)), int(rng.integers(2, 7)), int(rng.integers(1, 9)), int(rng.integers(1, 4))
    corr, jcorr = [((1,), (1,)), ((2,), (2,)), ((2, 2), (2,)), ((2, 2), (2, 2))][seed % 4]
    a1, a2 = np.triu_indices(nant, 1)
    nbl = a1.shape[0]
    counts = np.full(ntime, nbl)
    extra = int(rng.integers(0, 3))                      # trailing rows that belong to no bin
    nrow = ntime * nbl + extra
    idx_t = (np.int32, np.int64)[seed % 2]
    tbi = (np.arange(ntime) * nbl + 1000 * (seed % 3)).astype(idx_t)      # chunk offset
    tbc = counts.astype(idx_t)
    ant1 = np.concatenate([np.tile(a1, ntime), np.zeros(extra, int)]).astype(idx_t)
    ant2 = np.concatenate([np.tile(a2, ntime), np.ones(extra, int)]).astype(idx_t)
    rc = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    jones = rc(ntime, nant, nchan, ndir, *jcorr) + 1.0
    model = rc(nrow, nchan, ndir, *corr)
    data = rc(nrow, nchan, *corr)
    flag = rng.random(data.shape) < 0.2
    assert_array_equal(corrupt_vis(tbi, tbc, ant1, ant2, jones, model), oracle.corrupt_vis(tbi, tbc, ant1, ant2, jones, model))
    assert_array_equal(residual_vis(tbi, tbc, ant1, ant2, jones, data, flag, model),
                       oracle.residual_vis(tbi, tbc, ant1, ant2, jones, data, flag, model))
    j1 = np.ascontiguousarray(jones[:, :, :, :1])
    assert_array_equal(correct_vis(tbi, tbc, ant1, ant2, j1, data, flag), oracle.correct_vis(tbi, tbc, ant1, ant2, j1, data, flag))


@pytest.mark.parametrize("layout", range(4))
def test_two_directions_fetch_both_gain_records_per_gather(layout, monkeypatch):
    """With exactly two directions corrupt_vis / residual_vis fetch both records of a gain in one cooperative gather (whole
    cache lines per load instruction; csrc/af_calibration.hip `calib_kernel<..., 2>`): bit-equal to the oracle and to the
    one-record-per-direction form (AFHIP_CALIB_PAIR=0) in all four (visibility, gain) layouts, more than one wave per
    block, flags, rows outside every bin."""
    rng = np.random.default_rng(40 + layout)
    ntime, nant, nchan, ndir = 3, 9, 37, 2
    corr, jcorr = [((1,), (1,)), ((2,), (2,)), ((2, 2), (2,)), ((2, 2), (2, 2))][layout]
    a1, a2 = np.triu_indices(nant, 1)
    nbl = a1.shape[0]
    nrow = ntime * nbl + 2
    tbi, tbc = (np.arange(ntime) * nbl).astype(np.int64), np.full(ntime, nbl, np.int64)
    ant1 = np.concatenate([np.tile(a1, ntime), [0, 1]]).astype(np.int64)
    ant2 = np.concatenate([np.tile(a2, ntime), [1, 2]]).astype(np.int64)
    rc = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    jones, model, data = rc(ntime, nant, nchan, ndir, *jcorr) + 1.0, rc(nrow, nchan, ndir, *corr), rc(nrow, nchan, *corr)
    flag = rng.random(data.shape) < 0.2
    ref_c = oracle.corrupt_vis(tbi, tbc, ant1, ant2, jones, model)
    ref_r = oracle.residual_vis(tbi, tbc, ant1, ant2, jones, data, flag, model)
    for pair in ("1", "0"):
        monkeypatch.setenv("AFHIP_CALIB_PAIR", pair)
        assert_array_equal(corrupt_vis(tbi, tbc, ant1, ant2, jones, model), ref_c)
        assert_array_equal(residual_vis(tbi, tbc, ant1, ant2, jones, data, flag, model), ref_r)


def test_calibration_round_trip_and_errors(g9):
    """correct_vis undoes a direction-independent corrupt_vis (calibration/utils/tests/test_utils.py:117-164);
    the reference's argument errors"""
    a = (g9["tbin_idx"], g9["tbin_counts"], g9["ant1"], g9["ant2"])
    for tag in TAGS:
        j1 = np.ascontiguousarray(g9[tag + "_jones"][:, :, :, :1])
        m1 = np.ascontiguousarray(g9[tag + "_model"][:, :, :1])
        vis = corrupt_vis(*a, j1, m1)
        back = correct_vis(*a, j1, vis, np.zeros(vis.shape, bool))
        np.testing.assert_allclose(back, m1[:, :, 0], rtol=1e-10, atol=1e-12)
        assert check_type(j1, vis) == {"dd1": 0, "dd2": 0, "diag": 1, "full": 2}[tag]
    with pytest.raises(ValueError, match="n_dir > 1"):
        correct_vis(*a, g9["full_jones"], g9["full_data"], g9["full_flag"])
    with pytest.raises(ValueError, match="ncorr cant be larger than 2"):
        corrupt_vis(*a, np.ones((5, 4, 6, 3, 4), complex), np.ones((30, 6, 3, 4), complex))
    with pytest.raises(RuntimeError, match="Jones axes not compatible"):
        corrupt_vis(*a, g9["full_jones"], g9["dd2_model"])
    chunks, tbi, tbc = chunkify_rows(g9["time"], 2)
    assert chunks == (12, 12, 6) and tbi.dtype == np.int32
    assert_array_equal(tbi, g9["tbin_idx"])
    assert_array_equal(tbc, g9["tbin_counts"])


def test_calibration_torch_device_resident(g9):
    import torch
    dev = torch.device("cuda:0")
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    out = residual_vis(T(g9["tbin_idx"]), T(g9["tbin_counts"]), T(g9["ant1"]), T(g9["ant2"]), T(g9["full_jones"]),
                       T(g9["full_data"]), T(g9["full_flag"]), T(g9["full_model"]))
    assert isinstance(out, torch.Tensor) and out.is_cuda
    assert_array_equal(out.cpu().numpy(), g9["full_residual"])


@pytest.mark.parametrize("tag", TAGS)
def test_compute_and_corrupt_vis_golden(g9, tag):
    """the predict (phase x time-variable model / n) fused into corrupt_vis; device sin/cos are within 1 ulp of
    libm's, everything else is the reference's operation order"""
    from codex_africanus_amd.calibration.utils import compute_and_corrupt_vis
    a = (g9["tbin_idx"], g9["tbin_counts"], g9["ant1"], g9["ant2"])
    out = compute_and_corrupt_vis(*a, g9[tag + "_jones"], g9[tag + "_tmodel"], g9["cc_uvw"], g9["cc_freq"], g9["cc_lm"])
    ref = g9[tag + "_ccvis"]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert np.abs(out - ref).max() <= 1e-14 * np.abs(ref).max()
    # equals corrupt_vis on the materialised coherencies (compute_and_corrupt_vis.py:14-22)
    uvw, freq, lm, tm = g9["cc_uvw"], g9["cc_freq"], g9["cc_lm"], g9[tag + "_tmodel"]
    tidx = np.repeat(np.arange(5), 6)
    l, m = lm[..., 0], lm[..., 1]
    n = np.sqrt(1 - l**2 - m**2)
    ph = -2 * np.pi / 299792458.0 * freq[None, :, None] * (uvw[:, 0, None, None] * l[tidx][:, None, :]
                                                            + uvw[:, 1, None, None] * m[tidx][:, None, :]
                                                            + uvw[:, 2, None, None] * (n[tidx][:, None, :] - 1))
    extra = (None,) * (tm.ndim - 3)
    model = tm[tidx] * (np.exp(1j * ph) / n[tidx][:, None, :])[(..., ) + extra]
    np.testing.assert_allclose(out, corrupt_vis(*a, g9[tag + "_jones"], model), rtol=1e-12, atol=1e-12)
