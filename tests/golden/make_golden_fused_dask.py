#!/opt/conda/bin/python3.9
"""
G14 -- the REFERENCE's dask predict graph from source-level inputs, as africanus/rime/examples/predict.py:404-525
composes it, computed with the real reference and the real dask in the build container (conda python 3.9:
dask 2021.10, numba 0.54 + tests/golden/ref_shim.py):

    NUMBA_CACHE_DIR=/tmp/numba_cache PYTHONPATH=/root/reference:tests/golden:. \
        /opt/conda/bin/python3.9 tests/golden/make_golden_fused_dask.py

    phase  = africanus.rime.dask.phase_delay(lm, uvw, frequency)
    [shape = africanus.model.shape.dask.gaussian(uvw, frequency, shape_params)]
    coh    = da.einsum("srf,[srf,]sfij->srfij", phase, [shape,] brightness)
    dde    = africanus.rime.dask.beam_cube_dde(beam, extents, freq_map, lm, parangles, point_errors, scaling, frequency)
    [dde   = da.einsum("stafij,tajk->stafik", dde, africanus.rime.dask.feed_rotation(parangles, "linear"))]
    vis    = africanus.rime.dask.predict_vis(time_index, antenna1, antenna2, dde, coh, dde, die, base_vis, die)

on several (source, row/time, chan) chunkings; with brightness either given or made by the reference's
``spectral_model`` (africanus/model/spectral/spec_model.py:102) followed by the Stokes -> linear-feed correlations of
africanus/model/coherency/conversion.py:18-27 (restated here in one line: that module does not import under python 3.9).
What the build's ``rime.dask.fused_predict_vis`` / ``chunked.fused_predict_vis`` / ``sharding.fused_predict_shard``
have to reproduce (tests/test_gpu_fused_frontends.py), block for block, without materialising ``coh`` or ``dde``.
Stores inputs and results in g14_fused_dask.npz.
"""
import os

import ref_shim  # noqa: F401  (must come first)
import numpy as np
import dask.array as da

from africanus.rime.dask import phase_delay, beam_cube_dde, feed_rotation, predict_vis
from africanus.model.shape.dask import gaussian as gaussian_shape
from africanus.model.spectral.dask import spectral_model

HERE = os.path.dirname(os.path.abspath(__file__))

NSRC, NTIME, NANT, NCHAN = 9, 6, 5, 6
# (source chunks, row chunks, time chunks, chan chunks): rows of whole timesteps (10 baselines each), as the
# reference requires (africanus/rime/dask_predict.py:494-499: a row chunk's times live in its own time chunk)
CHUNKINGS = {
    "one": ((9,), (60,), (6,), (6,)),
    "rows3": ((9,), (20, 20, 20), (2, 2, 2), (6,)),
    "rows3u_src2_chan2": ((4, 5), (30, 10, 20), (3, 1, 2), (4, 2)),
}


def inputs():
    rng = np.random.default_rng(140)
    a1, a2 = np.triu_indices(NANT, 1)
    nbl = a1.shape[0]
    nrow = nbl * NTIME
    d = {}
    d["antenna1"] = np.tile(a1, NTIME).astype(np.int32)
    d["antenna2"] = np.tile(a2, NTIME).astype(np.int32)
    d["time_index"] = np.repeat(np.arange(NTIME), nbl).astype(np.int32)
    rad, ang = 0.05 * np.sqrt(rng.random(NSRC)), 2 * np.pi * rng.random(NSRC)
    d["lm"] = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1)
    d["uvw"] = rng.uniform(-1, 1, (nrow, 3)) * np.array([4000.0, 4000.0, 400.0])
    d["frequency"] = np.linspace(0.856e9, 1.712e9, NCHAN)
    d["stokes"] = np.stack([rng.lognormal(0, 1, NSRC)] + [0.1 * rng.standard_normal(NSRC) for _ in range(3)], axis=1)
    d["spi"] = rng.uniform(-1.0, 0.2, (NSRC, 2, 4))
    d["ref_freq"] = rng.uniform(0.9e9, 1.5e9, NSRC)
    X = rng.standard_normal((NSRC, NCHAN, 2, 2)) + 1j * rng.standard_normal((NSRC, NCHAN, 2, 2))
    d["brightness"] = X
    g = np.linspace(-1, 1, 9)
    ll, mm = np.meshgrid(g, g, indexing="ij")
    pattern = np.exp(-(ll**2 + mm**2) / 0.5) * np.exp(1j * (0.3 * ll + 0.2 * mm))
    gains = (1 + 0.1 * np.arange(4))[:, None] * np.array([1.0, 0.05j, -0.04j, 0.95])[None, :]
    d["beam"] = (pattern[:, :, None, None] * gains[None, None]).reshape(9, 9, 4, 2, 2)
    d["beam_lm_extents"] = np.array([[-0.06, 0.06], [-0.06, 0.06]])
    d["beam_freq_map"] = np.linspace(0.8e9, 1.6e9, 4)            # the top channels lie above the cube
    d["parallactic_angles"] = rng.uniform(0, np.pi / 6, (NTIME, NANT))
    d["point_errors"] = 1e-3 * rng.standard_normal((NTIME, NANT, NCHAN, 2))
    d["antenna_scaling"] = 1.0 + 1e-3 * rng.standard_normal((NANT, NCHAN, 2))
    shp = (NTIME, NANT, NCHAN, 2, 2)
    d["die"] = np.eye(2)[None, None, None] + 0.1 * (rng.standard_normal(shp) + 1j * rng.standard_normal(shp))
    d["base_vis"] = 0.1 * (rng.standard_normal((nrow, NCHAN, 2, 2)) + 1j * rng.standard_normal((nrow, NCHAN, 2, 2)))
    d["gauss_shape"] = np.stack([rng.uniform(1e-5, 2e-4, NSRC), rng.uniform(1e-5, 1e-4, NSRC),
                                 rng.uniform(0, np.pi, NSRC)], axis=1)
    d["gauss_shape"][::3] = 0.0                                   # every third source is a point source
    return d


def graph(d, chunking, beam=True, feed=False, gauss=False, die=False, model=False, local_time=False, streams=None):
    s, r, t, c = chunking
    lm = da.from_array(d["lm"], chunks=(s, 2))
    uvw = da.from_array(d["uvw"], chunks=(r, 3))
    freq = da.from_array(d["frequency"], chunks=(c,))
    ti = d["time_index"]
    if local_time:            # africanus/rime/examples/predict.py:507-516: "The index is not global"
        edges = np.concatenate([[0], np.cumsum(r)])
        ti = np.concatenate([ti[lo:hi] - ti[lo:hi].min() for lo, hi in zip(edges[:-1], edges[1:])])
    idx = [da.from_array(x, chunks=(r,)) for x in (ti, d["antenna1"], d["antenna2"])]
    if model:
        st = spectral_model(da.from_array(d["stokes"], chunks=(s, 4)), da.from_array(d["spi"], chunks=(s, 2, 4)),
                            da.from_array(d["ref_freq"], chunks=(s,)), freq, base=0)          # (src, chan, 4)
        # africanus/model/coherency/conversion.py:18-27, linear feeds: XX = I + Q, XY = U + iV, YX = U - iV, YY = I - Q
        I, Q, U, V = (st[..., k] for k in range(4))                                            # noqa: E741
        X = da.stack([I + Q, U + 1j * V, U - 1j * V, I - Q], axis=-1)
        X = X.reshape(X.shape[:2] + (2, 2)).rechunk({2: 2, 3: 2})     # the correlation axes stay whole
    else:
        X = da.from_array(d["brightness"], chunks=(s, c, 2, 2))
    phase = phase_delay(lm, uvw, freq)
    if gauss:
        shape = gaussian_shape(uvw, freq, da.from_array(d["gauss_shape"], chunks=(s, 3)))
        coh = da.einsum("srf,srf,sfij->srfij", phase, shape, X)
    else:
        coh = da.einsum("srf,sfij->srfij", phase, X)
    dde = None
    if beam:
        pa = da.from_array(d["parallactic_angles"], chunks=(t, NANT))
        dde = beam_cube_dde(da.from_array(d["beam"], chunks=d["beam"].shape),
                            da.from_array(d["beam_lm_extents"], chunks=(2, 2)),
                            da.from_array(d["beam_freq_map"], chunks=d["beam_freq_map"].shape), lm, pa,
                            da.from_array(d["point_errors"], chunks=(t, NANT, c, 2)),
                            da.from_array(d["antenna_scaling"], chunks=(NANT, c, 2)), freq)
        if feed:
            dde = da.einsum("stafij,tajk->stafik", dde, feed_rotation(pa, "linear"))
    g = da.from_array(d["die"], chunks=(t, NANT, c, 2, 2)) if die else None
    bv = da.from_array(d["base_vis"], chunks=(r, c, 2, 2)) if die else None
    return predict_vis(*idx, dde, coh, dde, g, bv, g, streams=streams)


CASES = {
    # name: keyword arguments of graph()
    "beam": dict(),
    "beam_feed": dict(feed=True),
    "beam_die": dict(die=True),
    "beam_feed_model": dict(feed=True, model=True),
    "beam_gauss": dict(gauss=True),
    "nobeam": dict(beam=False),
    "nobeam_gauss_die": dict(beam=False, gauss=True, die=True),
    "nobeam_model": dict(beam=False, model=True),
    "beam_localtime": dict(local_time=True),
}


def main():
    d = inputs()
    out = dict(d)
    for name, kw in CASES.items():
        ref = None
        for ck, chunking in CHUNKINGS.items():
            vis = graph(d, chunking, **kw).compute(scheduler="sync")
            assert vis.shape == (d["uvw"].shape[0], NCHAN, 2, 2) and vis.dtype == np.complex128
            if ref is None:
                ref = vis
            # chunked == unchunked up to the association of the source-chunk sum
            assert np.abs(vis - ref).max() <= 1e-12 * np.abs(ref).max(), (name, ck)
            out["vis_%s_%s" % (name, ck)] = vis
        st = graph(d, CHUNKINGS["rows3u_src2_chan2"], streams=True, **kw).compute(scheduler="sync")
        assert np.abs(st - ref).max() <= 1e-12 * np.abs(ref).max()
    path = os.path.join(HERE, "g14_fused_dask.npz")
    np.savez_compressed(path, **out)
    print("wrote g14_fused_dask.npz: %d arrays, %.1f KB" % (len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
