"""GPU tests of the chunk contract: chunked == unchunked for the reference's
chunkings (africanus/rime/tests/test_predict.py:20-31, africanus/dft/tests/test_dft.py:218-250),
streams=True and False, plus the dask front-ends when dask is importable."""
import numpy as np
import pytest
from numpy.testing import assert_array_equal, assert_array_almost_equal

from codex_africanus_amd import chunked, rime, dft

pytestmark = pytest.mark.gpu

CHUNKS = {"source": (2, 3, 4, 2, 2, 2, 2, 2, 2), "time": (2, 1, 1), "row": (4, 4, 2), "chan": (3, 2)}
CORR = {"c1": (1,), "c2": (2,), "c22": (2, 2)}
DDE = {"ddecoh": (True, True, True), "dde": (True, False, True), "coh": (False, True, False)}
DIE = {"diebv": (True, True, True), "die": (True, False, True), "bv": (False, True, False)}


@pytest.mark.parametrize("ck", list(CORR))
@pytest.mark.parametrize("dk", list(DDE))
@pytest.mark.parametrize("gk", list(DIE))
@pytest.mark.parametrize("streams", [True, False])
def test_chunked_predict_vis_equals_unchunked(g2, ck, dk, gk, streams):
    get = lambda k: g2["%s_%s" % (ck, k)]
    ti, a1, a2 = g2["time_idx"], g2["ant1"], g2["ant2"]
    a1j, blj, a2j = DDE[dk]
    g1j, bvis, g2j = DIE[gk]
    args = (get("a1") if a1j else None, get("bl") if blj else None, get("a2") if a2j else None,
            get("g1") if g1j else None, get("bv") if bvis else None, get("g2") if g2j else None)
    out = chunked.predict_vis(ti, a1, a2, *args, streams=streams, chunks=CHUNKS)
    ref = g2["%s_%s_%s_vis" % (ck, dk, gk)]
    assert out.shape == ref.shape
    assert_array_almost_equal(out, ref, decimal=12)
    if streams and not (g1j or bvis):
        # the serial chain adds sources in the reference's ascending order: bit-identical
        assert_array_equal(out, ref)


def test_chunked_phase_delay_exact(g1):
    out = chunked.phase_delay(g1["lm"], g1["uvw"], g1["frequency"], chunks={"source": 3, "row": 10, "chan": 2})
    assert_array_equal(out, rime.phase_delay(g1["lm"], g1["uvw"], g1["frequency"]))


def test_chunked_im_to_vis(g3):
    """rows chunked as in africanus/dft/tests/test_dft.py:240-243; row chunking is exact."""
    out = chunked.im_to_vis(g3["img_r4"], g3["uvw"], g3["lm"], g3["frequency"], chunks={"row": 7})
    assert_array_equal(out, dft.im_to_vis(g3["img_r4"], g3["uvw"], g3["lm"], g3["frequency"]))
    out2 = chunked.im_to_vis(g3["img_r70"], g3["uvw"], g3["lm"], g3["frequency70"],
                             chunks={"row": 20, "chan": 26})
    assert_array_almost_equal(out2, g3["vis_r70_fourier"], decimal=11)


def test_dask_wrappers_match_numpy(g2, g3):
    da = pytest.importorskip("dask.array")
    from codex_africanus_amd.rime import dask as rdask
    from codex_africanus_amd.dft import dask as ddask
    get = lambda k: g2["c22_%s" % k]
    ti, a1, a2 = g2["time_idx"], g2["ant1"], g2["ant2"]
    s, t, r, c = CHUNKS["source"], CHUNKS["time"], CHUNKS["row"], CHUNKS["chan"]
    dde = lambda x: da.from_array(x, chunks=(s, t, 4, c, 2, 2))
    coh = da.from_array(get("bl"), chunks=(s, r, c, 2, 2))
    die = lambda x: da.from_array(x, chunks=(t, 4, c, 2, 2))
    bv = da.from_array(get("bv"), chunks=(r, c, 2, 2))
    idx = [da.from_array(x, chunks=(r,)) for x in (ti, a1, a2)]
    for streams in (True, False):
        out = rdask.predict_vis(*idx, dde(get("a1")), coh, dde(get("a2")), die(get("g1")), bv,
                                die(get("g2")), streams=streams).compute(scheduler="sync")
        assert_array_almost_equal(out, g2["c22_ddecoh_diebv_vis"], decimal=12)
    vis = ddask.im_to_vis(da.from_array(g3["img_r4"], chunks=(13, 3, 4)),
                          da.from_array(g3["uvw"], chunks=(10, 3)),
                          da.from_array(g3["lm"], chunks=(13, 2)),
                          da.from_array(g3["frequency"], chunks=3)).compute(scheduler="sync")
    assert_array_almost_equal(vis, g3["vis_r4_fourier"], decimal=11)
    with pytest.raises(ValueError, match="lm chunks must match"):
        ddask.im_to_vis(da.from_array(g3["img_r4"], chunks=(13, 3, 4)),
                        da.from_array(g3["uvw"], chunks=(10, 3)),
                        da.from_array(g3["lm"], chunks=(5, 2)), da.from_array(g3["frequency"], chunks=3))


def test_chunked_vis_to_im():
    from conftest import load_golden
    g6 = load_golden("g6_vis_to_im.npz")
    ref = g6["im70"]
    out = chunked.vis_to_im(g6["vis70"], g6["uvw300"], g6["lm"], g6["frequency70"], g6["flags70"],
                            chunks={"row": 77, "chan": 26})
    # same bound as the unchunked parity test: 1e-11 relative to sum_r |vis|
    assert np.abs(out - ref).max() <= 1e-11 * np.abs(g6["vis70"]).sum(axis=0).max()


def test_dask_vis_to_im():
    da = pytest.importorskip("dask.array")
    from conftest import load_golden
    from codex_africanus_amd.dft import dask as ddask
    g6 = load_golden("g6_vis_to_im.npz")
    r, c = (100, 100, 100), (35, 35)
    out = ddask.vis_to_im(da.from_array(g6["vis70"], chunks=(r, c, 4)), da.from_array(g6["uvw300"], chunks=(r, 3)),
                          da.from_array(g6["lm"], chunks=(11, 2)), da.from_array(g6["frequency70"], chunks=c),
                          da.from_array(g6["flags70"], chunks=(r, c, 4))).compute(scheduler="sync")
    assert np.abs(out - g6["im70"]).max() <= 1e-11 * np.abs(g6["vis70"]).sum(axis=0).max()
    with pytest.raises(ValueError, match="Vis chunks must match flags"):
        ddask.vis_to_im(da.from_array(g6["vis70"], chunks=(r, c, 4)), da.from_array(g6["uvw300"], chunks=(r, 3)),
                        da.from_array(g6["lm"], chunks=(11, 2)), da.from_array(g6["frequency70"], chunks=c),
                        da.from_array(g6["flags70"], chunks=(r, (70,), 4)))


def test_dask_producers_and_calibration():
    """dask front-ends of the producers / calibration consumers (africanus/rime/dask.py:144-163,
    model/shape/dask.py, model/spectral/dask.py, calibration/utils/dask.py) against the array-level calls"""
    da = pytest.importorskip("dask.array")
    from conftest import load_golden
    from codex_africanus_amd.rime import dask as rdask, feed_rotation
    from codex_africanus_amd.model.shape import gaussian
    from codex_africanus_amd.model.shape import dask as sdask
    from codex_africanus_amd.model.spectral import spectral_model
    from codex_africanus_amd.model.spectral import dask as pdask
    from codex_africanus_amd.calibration import utils as cu
    from codex_africanus_amd.calibration.utils import dask as cdask
    g8, g9 = load_golden("g8_producers.npz"), load_golden("g9_calibration.npz")
    out = rdask.feed_rotation(da.from_array(g8["pa"], chunks=(2, 3)), "circular").compute(scheduler="sync")
    assert_array_equal(out, feed_rotation(g8["pa"], "circular"))
    out = sdask.gaussian(da.from_array(g8["uvw"], chunks=(13, 3)), da.from_array(g8["freq"], chunks=5),
                         da.from_array(g8["shape_params"], chunks=(9, 3))).compute(scheduler="sync")
    assert_array_equal(out, gaussian(g8["uvw"], g8["freq"], g8["shape_params"]))
    out = pdask.spectral_model(da.from_array(g8["stokes"], chunks=(4, 4)), da.from_array(g8["spi"], chunks=(4, 3, 4)),
                               da.from_array(g8["spec_ref_freq"], chunks=4), da.from_array(g8["freq"], chunks=7),
                               base=[0, 1, 2]).compute(scheduler="sync")
    assert_array_equal(out, spectral_model(g8["stokes"], g8["spi"], g8["spec_ref_freq"], g8["freq"], base=[0, 1, 2]))
    # calibration: 5 time bins of 6 rows, chunked 2 + 2 + 1 bins (chunkify_rows), chunk-local bin starts
    row_chunks, tbi, tbc = cu.chunkify_rows(g9["time"], 2)
    tchunks = (2, 2, 1)
    d = lambda x, c: da.from_array(x, chunks=c)
    jones, model = g9["full_jones"], g9["full_model"]
    dj = d(jones, (tchunks,) + jones.shape[1:])
    dm = d(model, (row_chunks,) + model.shape[1:])
    idx = [d(tbi, (tchunks,)), d(tbc, (tchunks,)), d(g9["ant1"], (row_chunks,)), d(g9["ant2"], (row_chunks,))]
    vis = cdask.corrupt_vis(*idx, dj, dm).compute(scheduler="sync")
    assert_array_equal(vis, g9["full_vis"])
    dv, df = d(g9["full_data"], (row_chunks,) + vis.shape[1:]), d(g9["full_flag"], (row_chunks,) + vis.shape[1:])
    res = cdask.residual_vis(*idx, dj, dv, df, dm).compute(scheduler="sync")
    assert_array_equal(res, g9["full_residual"])
    j1 = np.ascontiguousarray(jones[:, :, :, :1])
    cor = cdask.correct_vis(*idx, d(j1, (tchunks,) + j1.shape[1:]), dv, df).compute(scheduler="sync")
    assert_array_equal(cor, g9["full_corrected"])
    with pytest.raises(ValueError, match="Cannot chunk jones over antenna"):
        cdask.corrupt_vis(*idx, d(jones, (tchunks, 2) + jones.shape[2:]), dm)
