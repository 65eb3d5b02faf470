#!/opt/conda/bin/python3.9
"""
G12 -- the chunk contract of the reference's dask layer, generated with the REAL reference and the
REAL dask in the build container (conda python 3.9: dask 2021.10, numba 0.54 + tests/golden/ref_shim.py):

    NUMBA_CACHE_DIR=/tmp/numba_cache PYTHONPATH=/root/reference:tests/golden:. \
        /opt/conda/bin/python3.9 tests/golden/make_golden_dask.py

  * africanus.rime.dask.predict_vis on the g2 inputs with the chunking of africanus/rime/tests/test_predict.py:20-31,
    all 27 presence combinations x 3 correlation layouts, streams=True and streams=False   -> g12_dask.npz
  * africanus.dft.dask.im_to_vis / vis_to_im with the chunkings of africanus/dft/tests/test_dft.py:218-250,297-331
and, as a check of tests/blockwise_emulator.py (no output): the build's own wrapper module driven by real
``da.blockwise`` equals the same wrappers driven by the emulator, with the CPU oracle standing in for the HIP
library as the block function (the emulator is what the GPU box, which has no dask, runs).
"""
import os
import sys

import ref_shim  # noqa: F401  (must come first)
import numpy as np
import dask
import dask.array as da

from africanus.rime.dask import predict_vis as ref_dask_predict_vis
from africanus.dft.dask import im_to_vis as ref_dask_im_to_vis, vis_to_im as ref_dask_vis_to_im

HERE = os.path.dirname(os.path.abspath(__file__))
CHUNKS = {"source": (2, 3, 4, 2, 2, 2, 2, 2, 2), "time": (2, 1, 1), "row": (4, 4, 2), "ant": (4,), "chan": (3, 2)}
CORR = {"c1": (1,), "c2": (2,), "c22": (2, 2)}
DDE = {"ddecoh": (True, True, True), "dde": (True, False, True), "coh": (False, True, False)}
DIE = {"diebv": (True, True, True), "die": (True, False, True), "bv": (False, True, False)}


def main():
    g2 = np.load(os.path.join(HERE, "g2_predict_vis.npz"))
    g3 = np.load(os.path.join(HERE, "g3_im_to_vis.npz"))
    g6 = np.load(os.path.join(HERE, "g6_vis_to_im.npz"))
    out = {}
    s, t, r, a, c = (CHUNKS[k] for k in ("source", "time", "row", "ant", "chan"))
    idx = [da.from_array(g2[k], chunks=r) for k in ("time_idx", "ant1", "ant2")]
    for ck, cs in CORR.items():
        get = lambda k: g2["%s_%s" % (ck, k)]
        dde = lambda x: da.from_array(x, chunks=(s, t, a, c) + cs)
        die = lambda x: da.from_array(x, chunks=(t, a, c) + cs)
        for dk, (a1j, blj, a2j) in DDE.items():
            for gk, (g1j, bvis, g2j) in DIE.items():
                for streams in (True, False):
                    args = (dde(get("a1")) if a1j else None,
                            da.from_array(get("bl"), chunks=(s, r, c) + cs) if blj else None,
                            dde(get("a2")) if a2j else None, die(get("g1")) if g1j else None,
                            da.from_array(get("bv"), chunks=(r, c) + cs) if bvis else None,
                            die(get("g2")) if g2j else None)
                    vis = ref_dask_predict_vis(*idx, *args, streams=streams).compute(scheduler="sync")
                    ref = g2["%s_%s_%s_vis" % (ck, dk, gk)]
                    assert np.abs(vis - ref).max() < 1e-12
                    out["%s_%s_%s_streams%d" % (ck, dk, gk, int(streams))] = vis
    # dft: rows chunked (africanus/dft/tests/test_dft.py:240-243), and rows x chans
    vis = ref_dask_im_to_vis(da.from_array(g3["img_r4"], chunks=(13, 3, 4)), da.from_array(g3["uvw"], chunks=(10, 3)),
                             da.from_array(g3["lm"], chunks=(13, 2)), da.from_array(g3["frequency"], chunks=3))
    out["im_to_vis_r4_rows10_chans3"] = vis.compute(scheduler="sync")
    rr, cc = (100, 100, 100), (35, 35)
    im = ref_dask_vis_to_im(da.from_array(g6["vis70"], chunks=(rr, cc, 4)), da.from_array(g6["uvw300"], chunks=(rr, 3)),
                            da.from_array(g6["lm"], chunks=(11, 2)), da.from_array(g6["frequency70"], chunks=cc),
                            da.from_array(g6["flags70"], chunks=(rr, cc, 4)))
    out["vis_to_im_70_rows100_chans35"] = im.compute(scheduler="sync")
    np.savez_compressed(os.path.join(HERE, "g12_dask.npz"), **out)
    print("wrote g12_dask.npz: %d arrays, %.1f KB" % (len(out), os.path.getsize(os.path.join(HERE, "g12_dask.npz")) / 1024))
    check_emulator(g2)


def check_emulator(g2):
    """real da.blockwise == tests/blockwise_emulator.blockwise on the build's block wrappers (oracle as kernel)."""
    sys.path.insert(0, os.path.dirname(HERE))
    import oracle
    from blockwise_emulator import Chunked, blockwise
    calls = {"dask": [], "emu": []}

    def first(x):
        while isinstance(x, list):
            x = x[0]
        return x

    def make_block(tag):
        def coh_block(time_index, antenna1, antenna2, dde1, coh, dde2, base_vis):
            calls[tag].append((type(dde1).__name__, type(coh).__name__, time_index.shape, first(dde1).shape, coh.shape))
            vis = oracle.predict_vis(time_index, antenna1, antenna2, first(dde1), coh, first(dde2), None, None, None)
            return vis[None, ...]
        return coh_block

    s, t, r, a, c = (CHUNKS[k] for k in ("source", "time", "row", "ant", "chan"))
    cs = (2, 2)
    jones_ix, coh_ix = ("src", "row", "ant", "chan", "c1", "c2"), ("src", "row", "chan", "c1", "c2")
    d_idx = [da.from_array(g2[k], chunks=r) for k in ("time_idx", "ant1", "ant2")]
    real = da.blockwise(make_block("dask"), coh_ix, d_idx[0], ("row",), d_idx[1], ("row",), d_idx[2], ("row",),
                        da.from_array(g2["c22_a1"], chunks=(s, t, a, c) + cs), jones_ix,
                        da.from_array(g2["c22_bl"], chunks=(s, r, c) + cs), coh_ix,
                        da.from_array(g2["c22_a2"], chunks=(s, t, a, c) + cs), jones_ix, None, None,
                        align_arrays=False, adjust_chunks={"row": r, "src": 1},
                        meta=np.empty((0,) * 5, dtype=np.complex128), dtype=np.complex128).compute(scheduler="sync")
    e_idx = [Chunked(g2[k], (r,)) for k in ("time_idx", "ant1", "ant2")]
    emu = blockwise(make_block("emu"), coh_ix, e_idx[0], ("row",), e_idx[1], ("row",), e_idx[2], ("row",),
                    Chunked(g2["c22_a1"], (s, t, a, c, 2, 2)), jones_ix, Chunked(g2["c22_bl"], (s, r, c, 2, 2)), coh_ix,
                    Chunked(g2["c22_a2"], (s, t, a, c, 2, 2)), jones_ix, None, None)
    assert real.shape == emu.shape and np.array_equal(real, emu), (real.shape, emu.shape)
    assert sorted(calls["dask"]) == sorted(calls["emu"]) and len(calls["emu"]) == 9 * 3 * 2
    assert np.abs(emu.sum(axis=0) - g2["c22_ddecoh_bv_vis"] + g2["c22_bv"]).max() < 1e-12
    print("emulator == da.blockwise: %d block calls, identical argument structure and results" % len(calls["emu"]))


if __name__ == "__main__":
    main()
