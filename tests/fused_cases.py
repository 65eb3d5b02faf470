"""
The cases of G14 (tests/golden/make_golden_fused_dask.py): which arguments of ``fused_predict_vis`` each case of the
reference's dask predict graph corresponds to.  Shared by tests/test_gpu_fused_frontends.py (no dask) and
tests/dask_cases.py (real dask, conda interpreter): numpy only.
"""
import numpy as np

NSRC, NTIME, NANT, NCHAN = 9, 6, 5, 6
# (source chunks, row chunks, time chunks, chan chunks), as in make_golden_fused_dask.py
CHUNKINGS = {
    "one": ((9,), (60,), (6,), (6,)),
    "rows3": ((9,), (20, 20, 20), (2, 2, 2), (6,)),
    "rows3u_src2_chan2": ((4, 5), (30, 10, 20), (3, 1, 2), (4, 2)),
}
CASES = {
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


def linear_feed_rotation(pa):
    """africanus/rime/feeds.py:19-31 (linear feeds): [[cos, sin], [-sin, cos]] of the parallactic angle"""
    c, s = np.cos(pa), np.sin(pa)
    return np.stack([np.stack([c, s], -1), np.stack([-s, c], -1)], -2).astype(np.complex128)


def case_arrays(g, name, ck):
    """keyword arguments of fused_predict_vis for case `name` on chunking `ck` (the chunking only matters for the
    chunk-local time index of the ``local_time`` case)"""
    kw = dict(beam=True, feed=False, gauss=False, die=False, model=False, local_time=False)
    kw.update(CASES[name])
    ti = g["time_index"]
    if kw["local_time"]:
        r = CHUNKINGS[ck][1]
        edges = np.concatenate([[0], np.cumsum(r)])
        ti = np.concatenate([ti[lo:hi] - ti[lo:hi].min() for lo, hi in zip(edges[:-1], edges[1:])])
    a = dict(time_index=ti, antenna1=g["antenna1"], antenna2=g["antenna2"], lm=g["lm"], uvw=g["uvw"],
             frequency=g["frequency"])
    if kw["model"]:
        a.update(stokes=g["stokes"], spi=g["spi"], ref_freq=g["ref_freq"])
    else:
        a["brightness"] = g["brightness"]
    if kw["beam"]:
        a.update(beam=g["beam"], beam_lm_extents=g["beam_lm_extents"], beam_freq_map=g["beam_freq_map"],
                 parallactic_angles=g["parallactic_angles"], point_errors=g["point_errors"],
                 antenna_scaling=g["antenna_scaling"])
        if kw["feed"]:
            a["feed_rotation"] = linear_feed_rotation(g["parallactic_angles"])
    if kw["gauss"]:
        a["gauss_shape"] = g["gauss_shape"]
    if kw["die"]:
        a.update(die1_jones=g["die"], die2_jones=g["die"], base_vis=g["base_vis"])
    return a


def scale_of(g, name):
    """sum over sources of the largest |brightness| entry: the magnitude an error of the phasor is multiplied by"""
    ref = np.abs(g["vis_%s_one" % name]).max()
    return max(ref, np.abs(g["brightness"]).sum(axis=0).max())


def oracle_fused_predict_vis(time_index, antenna1, antenna2, lm, uvw, frequency, brightness=None,
                             beam=None, beam_lm_extents=None, beam_freq_map=None, parallactic_angles=None,
                             point_errors=None, antenna_scaling=None, die1_jones=None, base_vis=None, die2_jones=None,
                             convention="fourier", feed_rotation=None, gauss_shape=None, stokes=None, spi=None,
                             ref_freq=None, corr_schema=(("XX", "XY"), ("YX", "YY")), spectral_base=0, plan=None):
    """The reference chain on the CPU oracle with the signature of ``rime.fused_predict_vis`` (checker / CPU stand-in
    for the device call in the ``-m "not gpu"`` tests of the front-ends' chunk logic):
    spectral_model -> linear-feed correlations -> phase_delay [x gaussian shape] -> einsum -> beam_cube_dde
    [-> feed rotation einsum] -> predict_vis (africanus/rime/examples/predict.py:404-525)."""
    import oracle
    if stokes is not None:
        st = oracle.spectral_model(stokes, spi, ref_freq, frequency, base=spectral_base)
        I, Q, U, V = (st[..., k] for k in range(4))                                            # noqa: E741
        brightness = np.stack([I + Q, U + 1j * V, U - 1j * V, I - Q], axis=-1).reshape(st.shape[:2] + (2, 2))
    if brightness.ndim == 3:
        brightness = np.broadcast_to(brightness[:, None], (lm.shape[0], frequency.shape[0], 2, 2))
    phase = oracle.phase_delay(lm, uvw, frequency, convention)
    if gauss_shape is not None:
        phase = phase * oracle.gaussian_shape(uvw, frequency, gauss_shape)
    coh = np.einsum("srf,sfij->srfij", phase, brightness)
    dde = None
    if beam is not None:
        dde = oracle.beam_cube_dde(beam, beam_lm_extents, beam_freq_map, lm, parallactic_angles, point_errors,
                                   antenna_scaling, frequency)
        if feed_rotation is not None:
            dde = np.einsum("stafij,tajk->stafik", dde, feed_rotation)
    return oracle.predict_vis(time_index, antenna1, antenna2, dde, coh, dde, die1_jones, base_vis, die2_jones)
