"""
Driven by tests/test_gpu_dask_conda.py in an interpreter that has dask (python >= 3.8, numpy >= 1.20; no pytest, no
torch needed): the build's dask front-ends (codex_africanus_amd/rime/dask.py, dft/dask.py; reference
africanus/rime/dask_predict.py:443-593, africanus/dft/dask.py:26-90) on the HIP library, computed with dask's THREADED
scheduler, against G12 (the reference's own dask results) and the unchunked goldens.  Prints DASK_CASES_OK.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
# libafhip (and the HIP runtime under it) first: conda's scipy, which dask imports, ships an older libstdc++ than
# the ROCm runtime needs, and whichever libstdc++ is loaded first serves the whole process
from codex_africanus_amd import _lib as _lib_first             # noqa: E402
_lib_first.load()

import numpy as np                                              # noqa: E402
import dask                                                     # noqa: E402
import dask.array as da                                         # noqa: E402

from codex_africanus_amd.rime import dask as rdask            # noqa: E402
from codex_africanus_amd.dft import dask as ddask              # noqa: E402
from codex_africanus_amd import placement, _lib                # noqa: E402

CHUNKS = {"source": (2, 3, 4, 2, 2, 2, 2, 2, 2), "time": (2, 1, 1), "row": (4, 4, 2), "ant": (4,), "chan": (3, 2)}
CORR = {"c1": (1,), "c2": (2,), "c22": (2, 2)}
DDE = {"ddecoh": (True, True, True), "dde": (True, False, True), "coh": (False, True, False)}
DIE = {"diebv": (True, True, True), "die": (True, False, True), "bv": (False, True, False)}


def load(name):
    return np.load(os.path.join(HERE, "golden", name))


def raises(exc, match, fn):
    try:
        fn()
    except exc as e:
        assert match in str(e), (match, str(e))
        return
    raise AssertionError("no %s raised" % exc.__name__)


def main():
    g2, g3, g6, g12 = load("g2_predict_vis.npz"), load("g3_im_to_vis.npz"), load("g6_vis_to_im.npz"), load("g12_dask.npz")
    s, t, r, a, c = (CHUNKS[k] for k in ("source", "time", "row", "ant", "chan"))
    idx = [da.from_array(g2[k], chunks=r) for k in ("time_idx", "ant1", "ant2")]
    ncase = 0
    with dask.config.set(scheduler="threads", num_workers=6):
        for ck, cs in CORR.items():
            get = lambda k: g2["%s_%s" % (ck, k)]
            dde = lambda x: da.from_array(x, chunks=(s, t, a, c) + cs)
            die = lambda x: da.from_array(x, chunks=(t, a, c) + cs)
            for dk, (a1j, blj, a2j) in DDE.items():
                for gk, (g1j, bvis, g2j) in DIE.items():
                    args = (dde(get("a1")) if a1j else None,
                            da.from_array(get("bl"), chunks=(s, r, c) + cs) if blj else None,
                            dde(get("a2")) if a2j else None, die(get("g1")) if g1j else None,
                            da.from_array(get("bv"), chunks=(r, c) + cs) if bvis else None,
                            die(get("g2")) if g2j else None)
                    st = rdask.predict_vis(*idx, *args, streams=True)
                    fan = rdask.predict_vis(*idx, *args, streams=False)
                    assert st.chunks[0] == r and fan.chunks[0] == r and st.dtype == np.complex128
                    st, fan = dask.compute(st, fan)
                    key = "%s_%s_%s" % (ck, dk, gk)
                    # serial chain: bit for bit the reference's dask result
                    assert np.array_equal(st, g12[key + "_streams1"]), key
                    assert np.abs(fan - g12[key + "_streams0"]).max() < 1e-12, key
                    assert np.abs(fan - g2[key + "_vis"]).max() < 1e-12, key
                    ncase += 2
        # dies only: base_vis is None
        g = rdask.predict_vis(*idx, None, None, None, da.from_array(g2["c22_g1"], chunks=(t, a, c, 2, 2)), None,
                              da.from_array(g2["c22_g2"], chunks=(t, a, c, 2, 2))).compute()
        from codex_africanus_amd import rime
        assert np.array_equal(g, rime.predict_vis(g2["time_idx"], g2["ant1"], g2["ant2"], None, None, None,
                                                  g2["c22_g1"], None, g2["c22_g2"]))
        # chunk errors of the reference (africanus/rime/dask_predict.py:478-524)
        raises(ValueError, "Subdivision of antenna dimension",
               lambda: rdask.predict_vis(*idx, da.from_array(g2["c22_a1"], chunks=(s, t, (2, 2), c, 2, 2)), None,
                                         da.from_array(g2["c22_a2"], chunks=(s, t, (2, 2), c, 2, 2))))
        raises(ValueError, "does not equal number of time chunks",
               lambda: rdask.predict_vis(*idx, da.from_array(g2["c22_a1"], chunks=(s, (2, 2), a, c, 2, 2)), None,
                                         da.from_array(g2["c22_a2"], chunks=(s, (2, 2), a, c, 2, 2))))
        # dft
        vis = ddask.im_to_vis(da.from_array(g3["img_r4"], chunks=(13, 3, 4)), da.from_array(g3["uvw"], chunks=(10, 3)),
                              da.from_array(g3["lm"], chunks=(13, 2)), da.from_array(g3["frequency"], chunks=3))
        assert vis.chunks == ((10,) * 5, (3, 3), (4,))
        vis = vis.compute()
        assert np.abs(vis - g12["im_to_vis_r4_rows10_chans3"]).max() <= 1e-11 * np.abs(g3["img_r4"]).sum(axis=0).max()
        raises(ValueError, "lm chunks must match",
               lambda: ddask.im_to_vis(da.from_array(g3["img_r4"], chunks=(13, 3, 4)), da.from_array(g3["uvw"], chunks=(10, 3)),
                                       da.from_array(g3["lm"], chunks=(5, 2)), da.from_array(g3["frequency"], chunks=3)))
        rr, cc = (100, 100, 100), (35, 35)
        im = ddask.vis_to_im(da.from_array(g6["vis70"], chunks=(rr, cc, 4)), da.from_array(g6["uvw300"], chunks=(rr, 3)),
                             da.from_array(g6["lm"], chunks=(11, 2)), da.from_array(g6["frequency70"], chunks=cc),
                             da.from_array(g6["flags70"], chunks=(rr, cc, 4))).compute()
        assert np.abs(im - g12["vis_to_im_70_rows100_chans35"]).max() <= 1e-11 * np.abs(g6["vis70"]).sum(axis=0).max()
        raises(ValueError, "Vis chunks must match flags",
               lambda: ddask.vis_to_im(da.from_array(g6["vis70"], chunks=(rr, cc, 4)), da.from_array(g6["uvw300"], chunks=(rr, 3)),
                                       da.from_array(g6["lm"], chunks=(11, 2)), da.from_array(g6["frequency70"], chunks=cc),
                                       da.from_array(g6["flags70"], chunks=(rr, (70,), 4))))
        # phase_delay
        g1 = load("g1_phase_delay.npz")
        ph = rdask.phase_delay(da.from_array(g1["lm"], chunks=(3, 2)), da.from_array(g1["uvw"], chunks=(10, 3)),
                               da.from_array(g1["frequency"], chunks=2)).compute()
        assert np.array_equal(ph, rime.phase_delay(g1["lm"], g1["uvw"], g1["frequency"]))
    producers_and_calibration()
    wgridder_cases()
    nfused = fused_cases()
    stats = _lib.pool_stats(0)
    print("predict_vis cases: %d; fused predict cases: %d; placement devices %s policy %s; pool hits %d misses %d"
          % (ncase, nfused, placement.devices(), placement.get_policy(), stats["hits"], stats["misses"]))
    _lib.shutdown()
    print("DASK_CASES_OK")


def fused_cases():
    """rime.dask.fused_predict_vis (one fused device call per block) against G14: the REFERENCE's dask graph
    phase_delay -> einsum -> beam_cube_dde [-> feed rotation] -> predict_vis of africanus/rime/examples/predict.py:404-525
    on the same chunkings, and the reference's chunk errors (africanus/rime/dask_predict.py:478-524)"""
    from fused_cases import CASES, CHUNKINGS, NANT, case_arrays, scale_of
    g14 = load("g14_fused_dask.npz")

    def dask_args(name, ck, override=None):
        s, r, t, c = CHUNKINGS[ck]
        if override:
            s, r, t, c = (override.get(k, v) for k, v in zip("srtc", (s, r, t, c)))
        a = case_arrays(g14, name, ck)
        ch = {"time_index": (r,), "antenna1": (r,), "antenna2": (r,), "lm": (s, 2), "uvw": (r, 3), "frequency": (c,),
              "brightness": (s, c, 2, 2), "stokes": (s, 4), "spi": (s, 2, 4), "ref_freq": (s,), "gauss_shape": (s, 3),
              "beam": a["beam"].shape if "beam" in a else None, "beam_lm_extents": (2, 2), "beam_freq_map": (4,),
              "parallactic_angles": (t, NANT), "point_errors": (t, NANT, c, 2), "antenna_scaling": (NANT, c, 2),
              "feed_rotation": (t, NANT, 2, 2), "die1_jones": (t, NANT, c, 2, 2), "die2_jones": (t, NANT, c, 2, 2),
              "base_vis": (r, c, 2, 2)}
        return {k: da.from_array(v, chunks=ch[k]) for k, v in a.items()}

    n = 0
    with dask.config.set(scheduler="threads", num_workers=6):
        for name in CASES:
            for ck in CHUNKINGS:
                ref = g14["vis_%s_%s" % (name, ck)]
                for streams in (None, True):
                    vis = rdask.fused_predict_vis(streams=streams, **dask_args(name, ck))
                    assert vis.shape == ref.shape and vis.dtype == ref.dtype and vis.chunks[0] == CHUNKINGS[ck][1], (name, ck)
                    assert vis.chunks[1] == CHUNKINGS[ck][3]
                    out = vis.compute()
                    assert np.abs(out - ref).max() <= 1e-9 * scale_of(g14, name), (name, ck, streams)
                    n += 1
        # many row blocks computed concurrently, each placed by its block id, plans cached per row chunk
        many = dask_args("beam_feed", "one", {"r": (10,) * 6, "t": (1,) * 6})
        out = rdask.fused_predict_vis(**many).compute()
        assert np.abs(out - g14["vis_beam_feed_one"]).max() <= 1e-9 * scale_of(g14, "beam_feed")
    raises(ValueError, "does not equal number of time chunks",
           lambda: rdask.fused_predict_vis(**dask_args("beam", "rows3", {"t": (3, 3)})))
    raises(ValueError, "Subdivision of antenna dimension", lambda: rdask.fused_predict_vis(
        **dict(dask_args("beam", "one"), parallactic_angles=da.from_array(g14["parallactic_angles"], chunks=(6, (2, 3))))))
    raises(ValueError, "Beam chunking unsupported", lambda: rdask.fused_predict_vis(
        **dict(dask_args("beam", "one"), beam=da.from_array(g14["beam"], chunks=(5, 9, 4, 2, 2)))))
    raises(ValueError, "row chunks", lambda: rdask.fused_predict_vis(
        **dict(dask_args("beam", "rows3"), uvw=da.from_array(g14["uvw"], chunks=(30, 3)))))
    raises(ValueError, "chan chunks", lambda: rdask.fused_predict_vis(
        **dict(dask_args("beam", "one"), frequency=da.from_array(g14["frequency"], chunks=(3,)))))
    raises(ValueError, "source chunks", lambda: rdask.fused_predict_vis(
        **dict(dask_args("beam", "one"), lm=da.from_array(g14["lm"], chunks=(3, 2)))))
    raises(ValueError, "Both die1_jones and die2_jones",
           lambda: rdask.fused_predict_vis(**dict(dask_args("beam_die", "one"), die2_jones=None)))
    return n


def producers_and_calibration():
    """dask front-ends of the producers / calibration consumers (africanus/rime/dask.py:144-163, model/shape/dask.py,
    model/spectral/dask.py, calibration/utils/dask.py, rime/dask_predict.py:609-658) against the array-level calls"""
    from codex_africanus_amd.rime import feed_rotation, wsclean_predict
    from codex_africanus_amd.model.shape import gaussian
    from codex_africanus_amd.model.shape import dask as sdask
    from codex_africanus_amd.model.spectral import spectral_model
    from codex_africanus_amd.model.spectral import dask as pdask
    from codex_africanus_amd.calibration import utils as cu
    from codex_africanus_amd.calibration.utils import dask as cdask
    g8, g9, g7 = load("g8_producers.npz"), load("g9_calibration.npz"), load("g7_wsclean.npz")
    with dask.config.set(scheduler="threads", num_workers=4):
        out = rdask.feed_rotation(da.from_array(g8["pa"], chunks=(2, 3)), "circular").compute()
        assert np.array_equal(out, feed_rotation(g8["pa"], "circular"))
        out = sdask.gaussian(da.from_array(g8["uvw"], chunks=(13, 3)), da.from_array(g8["freq"], chunks=5),
                             da.from_array(g8["shape_params"], chunks=(9, 3))).compute()
        assert np.array_equal(out, gaussian(g8["uvw"], g8["freq"], g8["shape_params"]))
        out = pdask.spectral_model(da.from_array(g8["stokes"], chunks=(4, 4)), da.from_array(g8["spi"], chunks=(4, 3, 4)),
                                   da.from_array(g8["spec_ref_freq"], chunks=4), da.from_array(g8["freq"], chunks=7),
                                   base=[0, 1, 2]).compute()
        assert np.array_equal(out, spectral_model(g8["stokes"], g8["spi"], g8["spec_ref_freq"], g8["freq"], base=[0, 1, 2]))
        # calibration: 5 time bins of 6 rows, chunked 2 + 2 + 1 bins (chunkify_rows), chunk-local bin starts
        row_chunks, tbi, tbc = cu.chunkify_rows(g9["time"], 2)
        tchunks = (2, 2, 1)
        d = lambda x, c: da.from_array(x, chunks=c)
        jones, model = g9["full_jones"], g9["full_model"]
        dj = d(jones, (tchunks,) + jones.shape[1:])
        dm = d(model, (row_chunks,) + model.shape[1:])
        idx = [d(tbi, (tchunks,)), d(tbc, (tchunks,)), d(g9["ant1"], (row_chunks,)), d(g9["ant2"], (row_chunks,))]
        vis = cdask.corrupt_vis(*idx, dj, dm).compute()
        assert np.array_equal(vis, g9["full_vis"])
        dv, df = d(g9["full_data"], (row_chunks,) + vis.shape[1:]), d(g9["full_flag"], (row_chunks,) + vis.shape[1:])
        res = cdask.residual_vis(*idx, dj, dv, df, dm).compute()
        assert np.array_equal(res, g9["full_residual"])
        j1 = np.ascontiguousarray(jones[:, :, :, :1])
        cor = cdask.correct_vis(*idx, d(j1, (tchunks,) + j1.shape[1:]), dv, df).compute()
        assert np.array_equal(cor, g9["full_corrected"])
        raises(ValueError, "Cannot chunk jones over antenna",
               lambda: cdask.corrupt_vis(*idx, d(jones, (tchunks, 2) + jones.shape[2:]), dm))
        # wsclean_predict: source / row / chan chunks summed over the source chunks (rime/dask_predict.py:609-658)
        a = (g7["big_uvw"], g7["big_lm"], np.where(g7["big_is_gauss"], "GAUSSIAN", "POINT"), g7["big_flux"],
             g7["big_coeffs"], g7["big_log_poly"], g7["big_ref_freq"], g7["big_gauss_shape"], g7["big_freq"])
        if True:
            s_, r_, c_ = 10, 50, (40, 30)
            dd = [da.from_array(a[0], chunks=(r_, 3)), da.from_array(a[1], chunks=(s_, 2)), da.from_array(a[2], chunks=s_),
                  da.from_array(a[3], chunks=s_), da.from_array(a[4], chunks=(s_, a[4].shape[1])),
                  da.from_array(a[5], chunks=s_), da.from_array(a[6], chunks=s_), da.from_array(a[7], chunks=(s_, 3)),
                  da.from_array(a[8], chunks=(c_,))]
            out = rdask.wsclean_predict(*dd).compute()
            scale = np.abs(g7["big_spectrum"]).sum(axis=0).max()
            assert out.shape == g7["big_vis"].shape and np.abs(out - g7["big_vis"]).max() <= 1e-11 * scale
        # Stokes <-> correlation convert (model/coherency/tests/test_convert.py:143-160): chunked == unchunked
        import json
        from codex_africanus_amd.model.coherency import convert
        from codex_africanus_amd.model.coherency.dask import convert as da_convert
        g11 = load("g11_convert.npz")
        for chunks in (((10, 5, 3), (2, 3), (3,)), ((6, 8), (3, 3), (4, 4)), ((5, 5, 5),)):
            vis_shape = tuple(sum(c) for c in chunks)
            for isch, osch, implicit in json.loads(str(g11["cases"]))[:11]:
                ishape = np.asarray(isch).shape
                n = int(np.prod(vis_shape + ishape))
                vis = np.arange(1.0, n + 1.0).reshape(vis_shape + ishape)
                dvis = da.from_array(vis, chunks=chunks + tuple((s,) for s in ishape))
                assert np.array_equal(da_convert(dvis, isch, osch).compute(), convert(vis, isch, osch))


def wgridder_cases():
    """dask front-ends of the wgridder operators (africanus/gridding/wgridder/dask.py:53-463) with the reference's test
    chunking (tests/test_wgridder.py:357-600: rows in 3 chunks, one band per frequency chunk) against the array-level
    calls: to epsilon, since every row chunk picks its own w-planes"""
    from codex_africanus_amd.gridding.wgridder import model, dirty, residual, hessian
    from codex_africanus_amd.gridding.wgridder import dask as wdask
    rng = np.random.default_rng(420)
    nx, ny, nrow, nchan, nband, eps = 30, 128, 3333, 8, 2, 1e-6
    cell = 5.0 * np.pi / 180 / nx
    freq = 1e9 + np.arange(nchan) * (1e9 / nchan)
    uvw = (rng.random((nrow, 3)) - 0.5) / (cell * freq[-1] / 2.99792458e8)
    step = nchan // nband
    fbi, fbc = np.arange(0, nchan, step), np.full(nband, step)
    image = rng.standard_normal((nband, nx, ny))
    ms = rng.standard_normal((nrow, nchan)) + 1j * rng.standard_normal((nrow, nchan))
    wgt = rng.random((nrow, nchan))
    flag = (rng.random((nrow, nchan)) > 0.1).astype(np.uint8)
    rows = (1111, 1111, 1111)
    d_uvw, d_freq = da.from_array(uvw, chunks=(rows, 3)), da.from_array(freq, chunks=step)
    d_fbi, d_fbc = da.from_array(fbi, chunks=1), da.from_array(fbc, chunks=1)
    d_img = da.from_array(image, chunks=(1, nx, ny))
    d_ms, d_wgt, d_flag = (da.from_array(a, chunks=(rows, step)) for a in (ms, wgt, flag))
    l2 = lambda a, b: np.sqrt(np.sum(np.abs(a - b) ** 2) / np.sum(np.abs(b) ** 2))
    with dask.config.set(scheduler="threads", num_workers=4):
        vis = wdask.model(d_uvw, d_freq, d_img, d_fbi, d_fbc, cell, weights=d_wgt, flag=d_flag, epsilon=eps)
        assert vis.shape == (nrow, nchan) and vis.dtype == np.complex128
        assert l2(vis.compute(), model(uvw, freq, image, fbi, fbc, cell, weights=wgt, flag=flag, epsilon=eps)) <= 2 * eps
        img = wdask.dirty(d_uvw, d_freq, d_ms, d_fbi, d_fbc, nx, ny, cell, weights=d_wgt, flag=d_flag, epsilon=eps)
        assert img.shape == (nband, nx, ny) and img.dtype == np.float64
        assert l2(img.compute(), dirty(uvw, freq, ms, fbi, fbc, nx, ny, cell, weights=wgt, flag=flag, epsilon=eps)) <= 2 * eps
        img32 = wdask.dirty(d_uvw, d_freq, d_ms.astype(np.complex64), d_fbi, d_fbc, nx, ny, cell, epsilon=1e-4)
        assert img32.dtype == np.float32 and img32.compute().dtype == np.float32
        res = wdask.residual(d_uvw, d_freq, d_img, d_ms, d_fbi, d_fbc, cell, weights=d_wgt, flag=d_flag, epsilon=eps)
        want = residual(uvw, freq, image, ms, fbi, fbc, cell, weights=wgt, flag=flag, epsilon=eps)
        assert res.shape == image.shape and l2(res.compute(), want) <= 4 * eps
        hes = wdask.hessian(d_uvw, d_freq, d_img, d_fbi, d_fbc, cell, weights=d_wgt, flag=d_flag, epsilon=eps)
        want = hessian(uvw, freq, image, fbi, fbc, cell, weights=wgt, flag=flag, epsilon=eps)
        assert l2(hes.compute(), want) <= 4 * eps


if __name__ == "__main__":
    main()
