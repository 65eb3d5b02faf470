"""
The dask layer without dask: the build's per-block wrappers -- ``_coh_block`` / ``_die_block``
(codex_africanus_amd/rime/dask.py; reference africanus/rime/dask_predict.py:257-308), ``_im_to_vis_block`` /
``_vis_to_im_block`` (codex_africanus_amd/dft/dask.py; reference africanus/dft/dask.py:20-24,54-57), ``_phase_block``,
``_beam_block`` -- are driven with EXACTLY the arguments ``da.blockwise`` hands them (nested lists for contracted
axes, row blocks against time blocks by position, one-element row-block ids), by tests/blockwise_emulator.py, which
tests/golden/make_golden_dask.py verified against the real dask.  Expected values: G12 (tests/golden/g12_dask.npz),
produced by the REFERENCE's dask wrappers with real dask on the chunkings of africanus/rime/tests/test_predict.py:20-31
and africanus/dft/tests/test_dft.py:218-250, for ``streams`` True and False; and the unchunked goldens G2 / G3 / G6.
Blocks also run concurrently on a thread pool, as under dask's threaded scheduler.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
from numpy.testing import assert_array_equal

from blockwise_emulator import Chunked, blockwise
from conftest import load_golden
from codex_africanus_amd import placement
from codex_africanus_amd.rime import dask as rdask
from codex_africanus_amd.dft import dask as ddask

pytestmark = pytest.mark.gpu

CHUNKS = {"source": (2, 3, 4, 2, 2, 2, 2, 2, 2), "time": (2, 1, 1), "row": (4, 4, 2), "ant": (4,), "chan": (3, 2)}
CORR = {"c1": (1,), "c2": (2,), "c22": (2, 2)}
DDE = {"ddecoh": (True, True, True), "dde": (True, False, True), "coh": (False, True, False)}
DIE = {"diebv": (True, True, True), "die": (True, False, True), "bv": (False, True, False)}


@pytest.fixture(scope="module")
def g12():
    return load_golden("g12_dask.npz")


def _ids(n):
    return Chunked(np.arange(n, dtype=np.int64), ((1,) * n,))


def _predict(g2, ck, dk, gk, streams, executor=None):
    """The graph of rime.dask.predict_vis (parallel_reduction / linear_reduction / apply_dies of the reference),
    block calls made through the emulator."""
    cs = CORR[ck]
    s, t, r, a, c = (CHUNKS[k] for k in ("source", "time", "row", "ant", "chan"))
    cd = tuple((n,) for n in cs)
    cn = tuple("corr-%d" % i for i in range(len(cs)))
    jones_ix, coh_ix = ("src", "row", "ant", "chan") + cn, ("src", "row", "chan") + cn
    g_ix, v_ix = ("row", "ant", "chan") + cn, ("row", "chan") + cn
    get = lambda k: g2["%s_%s" % (ck, k)]
    a1j, blj, a2j = DDE[dk]
    g1j, bvis, g2j = DIE[gk]
    idx = [Chunked(g2[k], (r,)) for k in ("time_idx", "ant1", "ant2")]
    ids = _ids(len(r))

    edges = np.concatenate([[0], np.cumsum(s)])

    if streams:
        running = None
        for k in range(len(s)):
            lo, hi = int(edges[k]), int(edges[k + 1])
            sel = slice(lo, hi)
            # one source chunk: Chunked over exactly that chunk
            d1 = Chunked(get("a1")[sel], ((hi - lo,), t, a, c) + cd) if a1j else None
            d2 = Chunked(get("a2")[sel], ((hi - lo,), t, a, c) + cd) if a2j else None
            co = Chunked(get("bl")[sel], ((hi - lo,), r, c) + cd) if blj else None
            run = None if running is None else Chunked(running, ((1,), r, c) + cd)
            running = blockwise(rdask._coh_block, coh_ix, idx[0], ("row",), idx[1], ("row",), idx[2], ("row",),
                                d1, None if d1 is None else jones_ix, co, None if co is None else coh_ix,
                                d2, None if d2 is None else jones_ix, run, None if run is None else coh_ix,
                                ids, ("row",), executor=executor)
        summed = running[0]
    else:
        d1 = Chunked(get("a1"), (s, t, a, c) + cd) if a1j else None
        d2 = Chunked(get("a2"), (s, t, a, c) + cd) if a2j else None
        co = Chunked(get("bl"), (s, r, c) + cd) if blj else None
        per_chunk = blockwise(rdask._coh_block, coh_ix, idx[0], ("row",), idx[1], ("row",), idx[2], ("row",),
                              d1, None if d1 is None else jones_ix, co, None if co is None else coh_ix,
                              d2, None if d2 is None else jones_ix, None, None, ids, ("row",), executor=executor)
        assert per_chunk.shape[0] == len(s)          # one slab per source chunk (adjust_chunks={"src": 1})
        summed = per_chunk.sum(axis=0)
    base = summed + get("bv") if bvis else summed
    d1 = Chunked(get("g1"), (t, a, c) + cd) if g1j else None
    d2 = Chunked(get("g2"), (t, a, c) + cd) if g2j else None
    return blockwise(rdask._die_block, v_ix, idx[0], ("row",), idx[1], ("row",), idx[2], ("row",),
                     d1, None if d1 is None else g_ix, Chunked(base, (r, c) + cd), v_ix,
                     d2, None if d2 is None else g_ix, ids, ("row",), executor=executor)


@pytest.mark.parametrize("ck", list(CORR))
@pytest.mark.parametrize("dk", list(DDE))
@pytest.mark.parametrize("gk", list(DIE))
def test_predict_vis_blocks_equal_the_reference_dask_graph(g2, g12, ck, dk, gk):
    ref_unchunked = g2["%s_%s_%s_vis" % (ck, dk, gk)]
    # streams=True: the serial source-chunk chain adds in the reference's order -> every bit equal to the
    # reference's dask result (the HIP predict_vis is bit-identical to the numba kernel block by block)
    out = _predict(g2, ck, dk, gk, True)
    assert out.shape == ref_unchunked.shape
    assert_array_equal(out, g12["%s_%s_%s_streams1" % (ck, dk, gk)])
    # streams=False: per-chunk results are bit-identical, the tree sum over source chunks is dask's own
    out = _predict(g2, ck, dk, gk, False)
    np.testing.assert_allclose(out, g12["%s_%s_%s_streams0" % (ck, dk, gk)], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out, ref_unchunked, rtol=0, atol=1e-12)


def test_predict_vis_blocks_on_a_thread_pool(g2, g12):
    """blocks called concurrently from worker threads (dask's threaded scheduler): per-thread streams, pooled
    scratch, thread-local error text -- same bits as the serial run"""
    with ThreadPoolExecutor(6) as ex:
        for streams in (True, False) * 3:
            out = _predict(g2, "c22", "ddecoh", "diebv", streams, executor=ex)
            if streams:
                assert_array_equal(out, g12["c22_ddecoh_diebv_streams1"])
            else:
                np.testing.assert_allclose(out, g12["c22_ddecoh_diebv_streams0"], rtol=0, atol=1e-12)


def test_dies_only_block_takes_no_base_vis(g2):
    """a dies-only call hands the block function base_vis=None (the blockwise call must pair it with a
    None index)"""
    r, t, a, c = (CHUNKS[k] for k in ("row", "time", "ant", "chan"))
    idx = [Chunked(g2[k], (r,)) for k in ("time_idx", "ant1", "ant2")]
    g_ix, v_ix = ("row", "ant", "chan", "c1", "c2"), ("row", "chan", "c1", "c2")
    out = blockwise(rdask._die_block, v_ix, idx[0], ("row",), idx[1], ("row",), idx[2], ("row",),
                    Chunked(g2["c22_g1"], (t, a, c, (2,), (2,))), g_ix, None, None,
                    Chunked(g2["c22_g2"], (t, a, c, (2,), (2,))), g_ix, _ids(3), ("row",))
    from codex_africanus_amd import rime
    ref = rime.predict_vis(g2["time_idx"], g2["ant1"], g2["ant2"], None, None, None, g2["c22_g1"], None, g2["c22_g2"])
    assert_array_equal(out, ref)


def test_im_to_vis_and_vis_to_im_blocks(g3, g12):
    g6 = load_golden("g6_vis_to_im.npz")
    r = (10,) * 5
    out = blockwise(ddask._im_to_vis_block, ("row", "chan", "corr"),
                    Chunked(g3["img_r4"], ((13,), (3, 3), (4,))), ("src", "chan", "corr"),
                    Chunked(g3["uvw"], (r, (3,))), ("row", "uvwc"), Chunked(g3["lm"], ((13,), (2,))), ("src", "lmc"),
                    Chunked(g3["frequency"], ((3, 3),)), ("chan",), _ids(5), ("row",),
                    convention="fourier", dtype_=np.complex128)
    ref = g12["im_to_vis_r4_rows10_chans3"]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert np.abs(out - ref).max() <= 1e-11 * np.abs(g3["img_r4"]).sum(axis=0).max()
    rr, cc = (100, 100, 100), (35, 35)
    with ThreadPoolExecutor(3) as ex:
        ims = blockwise(ddask._vis_to_im_block, ("row", "src", "chan", "corr"),
                        Chunked(g6["vis70"], (rr, cc, (4,))), ("row", "chan", "corr"),
                        Chunked(g6["uvw300"], (rr, (3,))), ("row", "uvwc"), Chunked(g6["lm"], ((11,), (2,))), ("src", "lmc"),
                        Chunked(g6["frequency70"], (cc,)), ("chan",), Chunked(g6["flags70"], (rr, cc, (4,))),
                        ("row", "chan", "corr"), _ids(3), ("row",), executor=ex,
                        convention="fourier", dtype_=np.float64)
    assert ims.shape[0] == 3                         # one image per row block (adjust_chunks={"row": 1})
    im = ims.sum(axis=0)
    assert np.abs(im - g12["vis_to_im_70_rows100_chans35"]).max() <= 1e-11 * np.abs(g6["vis70"]).sum(axis=0).max()


def test_phase_and_beam_blocks(g1, g4):
    from codex_africanus_amd import rime
    out = blockwise(rdask._phase_block, ("s", "r", "c"), Chunked(g1["lm"], ((3, 4), (2,))), ("s", "x"),
                    Chunked(g1["uvw"], ((10, 10, 13), (3,))), ("r", "y"), Chunked(g1["frequency"], ((2, 3),)), ("c",),
                    convention="casa")
    assert_array_equal(out, rime.phase_delay(g1["lm"], g1["uvw"], g1["frequency"], convention="casa"))


def test_block_ids_select_devices():
    """row block k -> device k % n (one visible device here: always 0); policy 'none' leaves the selection alone"""
    from codex_africanus_amd import _lib
    devs = placement.devices()
    assert devs and devs[0] == 0
    with placement.block(np.array([5])):
        assert placement.choose() == devs[5 % len(devs)]
        assert placement.choose(devs=(0, 1, 2, 3)) == 1
    assert placement.choose(policy="none") is None
    assert _lib.get_device() in devs


def test_wgridder_blocks_row_chunks_times_bands():
    """The wgridder front-ends' block functions (codex_africanus_amd/gridding/wgridder/dask.py; reference
    africanus/gridding/wgridder/dask.py:22-463) under the blockwise calling convention: blocks are (row chunk, band);
    uvw arrives as a one-element list (its "three" axis is contracted), the image of ``model`` as a doubly nested one
    (nx, ny contracted); ``dirty`` gives one image per row chunk, summed.  Every row chunk picks its own w-planes, so
    chunked == unchunked to epsilon (test_wgridder.py:357-518), not to rounding."""
    from codex_africanus_amd.gridding.wgridder import dask as wdask, model, dirty
    rng = np.random.default_rng(420)
    nx, ny, nrow, nchan, nband, eps = 30, 64, 3333, 8, 2, 1e-6
    cell = 5.0 * np.pi / 180 / nx
    freq = 1e9 + np.arange(nchan) * (1e9 / nchan)
    uvw = (rng.random((nrow, 3)) - 0.5) / (cell * freq[-1] / 2.99792458e8)
    step = nchan // nband
    fbi = np.arange(0, nchan, step)
    fbc = np.full(nband, step)
    image = rng.standard_normal((nband, nx, ny))
    ms = rng.standard_normal((nrow, nchan)) + 1j * rng.standard_normal((nrow, nchan))
    wgt = rng.random((nrow, nchan))
    flag = (rng.random((nrow, nchan)) > 0.1).astype(np.uint8)
    rows = (1111, 1111, 1111)
    C = Chunked
    c_uvw, c_freq = C(uvw, (rows, (3,))), C(freq, ((step,) * nband,))
    c_fbi, c_fbc = C(fbi, ((1,) * nband,)), C(fbc, ((1,) * nband,))
    c_img = C(image, ((1,) * nband, (nx,), (ny,)))
    c_ms, c_wgt, c_flag = (C(a, (rows, (step,) * nband)) for a in (ms, wgt, flag))
    ids = C(np.arange(len(rows)), ((1,) * len(rows),))
    kw = dict(cell=cell, celly=None, epsilon=eps, do_wstacking=True)
    vis = blockwise(wdask._model_block, ("row", "chan"), c_uvw, ("row", "three"), c_freq, ("chan",),
                    c_img, ("chan", "nx", "ny"), c_fbi, ("chan",), c_fbc, ("chan",), c_wgt, ("row", "chan"),
                    c_flag, ("row", "chan"), ids, ("row",), **kw)
    ref = model(uvw, freq, image, fbi, fbc, cell, weights=wgt, flag=flag, epsilon=eps)
    assert vis.shape == ref.shape
    assert np.sqrt(np.sum(np.abs(vis - ref) ** 2) / np.sum(np.abs(ref) ** 2)) <= 2 * eps
    ims = blockwise(wdask._dirty_block, ("row", "chan", "nx", "ny"), c_uvw, ("row", "three"), c_freq, ("chan",),
                    c_ms, ("row", "chan"), c_fbi, ("chan",), c_fbc, ("chan",), c_wgt, ("row", "chan"),
                    c_flag, ("row", "chan"), ids, ("row",), nx=nx, ny=ny, **kw)
    assert ims.shape == (len(rows), nband, nx, ny)
    ref = dirty(uvw, freq, ms, fbi, fbc, nx, ny, cell, weights=wgt, flag=flag, epsilon=eps)
    got = ims.sum(axis=0)
    assert np.sqrt(np.sum((got - ref) ** 2) / np.sum(ref ** 2)) <= 2 * eps
    # weights / flags absent: passed as None with index None
    vis0 = blockwise(wdask._model_block, ("row", "chan"), c_uvw, ("row", "three"), c_freq, ("chan",),
                     c_img, ("chan", "nx", "ny"), c_fbi, ("chan",), c_fbc, ("chan",), None, None, None, None,
                     ids, ("row",), **kw)
    ref0 = model(uvw, freq, image, fbi, fbc, cell, epsilon=eps)
    assert np.sqrt(np.sum(np.abs(vis0 - ref0) ** 2) / np.sum(np.abs(ref0) ** 2)) <= 2 * eps
