"""
dask.array front-ends of the HIP hot path with the signatures of ``africanus.rime.dask``
(africanus/rime/dask.py:38-52,166-212; africanus/rime/dask_predict.py:443-593).

dask is optional (the reference guards it with ``requires_optional``,
africanus/util/requirements.py:31): importing this module without dask works, calling a wrapper
raises ``ImportError``.  Each dask block becomes one call of the array-level HIP function; the
chunk rules are the reference's (single antenna chunk, #row chunks == #time chunks, sources
reduced either by a serial chain ``streams=True`` or by a sum of per-chunk results).
"""
import numpy as np

try:
    import dask.array as da
    _dask_error = None
except ImportError as e:  # pragma: no cover - depends on the environment
    da = None
    _dask_error = e

from .phase import phase_delay as _np_phase_delay
from .predict import predict_vis as _np_predict_vis, predict_checks
from .fast_beam_cubes import beam_cube_dde as _np_beam_cube_dde
from .wsclean_predict import wsclean_predict as _np_wsclean_predict
from .feeds import feed_rotation as _np_feed_rotation
from .. import placement


def _need_dask():
    if da is None:
        raise ImportError("dask.array is required for codex_africanus_amd.rime.dask: %s" % (_dask_error,))


def _first(x):
    """Blocks of a contracted, single-chunk axis arrive as nested lists."""
    while isinstance(x, list):
        x = x[0]
    return x


def _row_block_ids(nblocks):
    """One element per row block, holding the block's number: rides along the "row" axis of a blockwise call
    (matched by position, ``align_arrays=False``) so that the block function knows which row block it is and
    ``placement`` can map row block k to GPU k % n_devices (north star: dask row chunks -> GPUs of one node)."""
    return da.arange(nblocks, chunks=1, dtype=np.int64)


# ---------------------------------------------------------------------------- phase_delay
def _phase_block(lm, uvw, frequency, convention):
    return _np_phase_delay(_first(lm), _first(uvw), frequency, convention=convention)


def phase_delay(lm, uvw, frequency, convention="fourier"):
    _need_dask()
    dtype = np.result_type(np.complex64, lm.dtype, uvw.dtype, frequency.dtype)
    return da.blockwise(_phase_block, ("s", "r", "c"), lm, ("s", "x"), uvw, ("r", "y"),
                        frequency, ("c",), convention=convention, dtype=dtype)


# ---------------------------------------------------------------------------- feed_rotation
def feed_rotation(parallactic_angles, feed_type="linear"):
    """africanus/rime/dask.py:144-163: one block per chunk of parallactic angles."""
    _need_dask()
    pa_dims = tuple("pa-%d" % i for i in range(parallactic_angles.ndim))
    dtype = np.result_type(parallactic_angles.dtype, np.complex64)
    return da.blockwise(_np_feed_rotation, pa_dims + ("corr-1", "corr-2"), parallactic_angles, pa_dims,
                        feed_type=feed_type, new_axes={"corr-1": 2, "corr-2": 2}, dtype=dtype)


# ---------------------------------------------------------------------------- wsclean_predict
def _wsclean_block(uvw, lm, source_type, flux, coeffs, log_poly, ref_freq, gauss_shape, frequency):
    return _np_wsclean_predict(_first(uvw), _first(lm), source_type, flux, _first(coeffs), log_poly, ref_freq,
                               _first(gauss_shape), frequency)[None]


def wsclean_predict(uvw, lm, source_type, flux, coeffs, log_poly, ref_freq, gauss_shape, frequency):
    """africanus/rime/dask_predict.py:609-658: one block per (source, row, chan) chunk, summed over the
    source chunks.  Spectrum and predict of a block are one fused device call."""
    _need_dask()
    dtype = np.result_type(np.complex64, uvw.dtype, lm.dtype, flux.dtype, coeffs.dtype, ref_freq.dtype,
                           frequency.dtype)
    vis = da.blockwise(_wsclean_block, ("source", "row", "chan", "corr"), uvw, ("row", "uvw"), lm, ("source", "lm"),
                       source_type, ("source",), flux, ("source",), coeffs, ("source", "comp"),
                       log_poly, ("source",), ref_freq, ("source",), gauss_shape, ("source", "gauss"),
                       frequency, ("chan",), adjust_chunks={"source": 1}, new_axes={"corr": 1}, dtype=dtype)
    return vis.sum(axis=0)


# ---------------------------------------------------------------------------- beam_cube_dde
def _beam_block(beam, extents, freq_map, lm, pa, pe, ascale, frequency):
    return _np_beam_cube_dde(_first(beam), _first(extents), _first(freq_map), _first(lm), pa,
                             _first(pe), _first(ascale), frequency)


def beam_cube_dde(beam, beam_lm_extents, beam_freq_map, lm, parallactic_angles, point_errors,
                  antenna_scaling, frequency):
    _need_dask()
    if any(len(c) != 1 for c in beam.chunks):
        raise ValueError("beam must have a single chunk")
    corrs = tuple("corr-%d" % i for i in range(beam.ndim - 3))
    return da.blockwise(
        _beam_block, ("src", "time", "ant", "chan") + corrs,
        beam, ("bl", "bm", "bf") + corrs, beam_lm_extents, ("e1", "e2"), beam_freq_map, ("bf",),
        lm, ("src", "lmc"), parallactic_angles, ("time", "ant"),
        point_errors, ("time", "ant", "chan", "pec"), antenna_scaling, ("ant", "chan", "asc"),
        frequency, ("chan",), dtype=beam.dtype)


# ---------------------------------------------------------------------------- predict_vis
def _coh_block(time_index, antenna1, antenna2, dde1, coh, dde2, base_vis, block_id=None):
    # dde blocks lose the single-chunk 'ant' axis into a list; a chained running sum arrives
    # with the length-1 source axis the previous link added
    if base_vis is not None:
        base_vis = base_vis[0]
    with placement.block(block_id):
        vis = _np_predict_vis(time_index, antenna1, antenna2,
                              None if dde1 is None else _first(dde1), coh,
                              None if dde2 is None else _first(dde2), None, base_vis, None)
    return vis[None, ...]


def _die_block(time_index, antenna1, antenna2, die1, base_vis, die2, block_id=None):
    with placement.block(block_id):
        return _np_predict_vis(time_index, antenna1, antenna2, None, None, None,
                               None if die1 is None else _first(die1), base_vis,
                               None if die2 is None else _first(die2))


def _check_jones_chunks(name, arr, ant_axis, time_axis, time_index, pair):
    if arr.shape[ant_axis] != arr.chunks[ant_axis][0]:
        raise ValueError("Subdivision of antenna dimension into multiple chunks is not supported.")
    if arr.chunks != pair.chunks:
        raise ValueError("%s1_jones.chunks != %s2_jones.chunks" % (name, name))
    if len(arr.chunks[time_axis]) != len(time_index.chunks[0]):
        raise ValueError("Number of row chunks (%s) does not equal number of time chunks (%s)."
                         % (time_index.chunks[0], arr.chunks[time_axis]))


def predict_vis(time_index, antenna1, antenna2, dde1_jones=None, source_coh=None, dde2_jones=None,
                die1_jones=None, base_vis=None, die2_jones=None, streams=None):
    _need_dask()
    tup = predict_checks(time_index, antenna1, antenna2, dde1_jones, source_coh, dde2_jones,
                         die1_jones, base_vis, die2_jones)
    have_ddes1, have_coh, have_ddes2, have_dies1, have_bvis, have_dies2 = tup
    have_ddes, have_dies = have_ddes1 and have_ddes2, have_dies1 and have_dies2
    if have_ddes:
        _check_jones_chunks("dde", dde1_jones, 2, 1, time_index, dde2_jones)
    if have_dies:
        _check_jones_chunks("die", die1_jones, 1, 0, time_index, die2_jones)
    present = [a for a in (dde1_jones, source_coh, dde2_jones, die1_jones, die2_jones) if a is not None]
    out_dtype = np.result_type(*[a.dtype for a in present]) if present else base_vis.dtype
    row_chunks = time_index.chunks[0]
    block_ids = _row_block_ids(len(row_chunks))

    summed = None
    if have_ddes or have_coh:
        ncorr_dims = (dde1_jones.ndim - 4) if have_ddes else (source_coh.ndim - 3)
        cd = tuple("corr-%d" % i for i in range(ncorr_dims))
        jones_ix, coh_ix = ("src", "row", "ant", "chan") + cd, ("src", "row", "chan") + cd
        nsrc_chunks = len((dde1_jones if have_ddes else source_coh).chunks[0])

        def blocks(src_sel, bvis):
            d1 = None if not have_ddes else dde1_jones.blocks[src_sel]
            d2 = None if not have_ddes else dde2_jones.blocks[src_sel]
            co = None if not have_coh else source_coh.blocks[src_sel]
            return da.blockwise(
                _coh_block, coh_ix, time_index, ("row",), antenna1, ("row",), antenna2, ("row",),
                d1, None if d1 is None else jones_ix, co, None if co is None else coh_ix,
                d2, None if d2 is None else jones_ix,
                bvis, None if bvis is None else coh_ix, block_ids, ("row",),
                align_arrays=False, adjust_chunks={"row": row_chunks, "src": 1},
                meta=np.empty((0,) * len(coh_ix), dtype=out_dtype), dtype=out_dtype)

        if streams is True:
            # serial chain over source chunks: chunk k's sum rides in as base_vis of chunk k+1
            running = None
            for k in range(nsrc_chunks):
                running = blocks(slice(k, k + 1), running)
            summed = running[0]
        else:
            summed = blocks(slice(None), None).sum(axis=0)

    if not have_dies and not have_bvis:
        return summed
    if summed is not None:
        base_vis = summed if not have_bvis else base_vis + summed
    ncorr_dims = (die1_jones.ndim - 3) if have_dies else (base_vis.ndim - 2)
    cd = tuple("corr-%d" % i for i in range(ncorr_dims))
    g_ix, v_ix = ("row", "ant", "chan") + cd, ("row", "chan") + cd
    return da.blockwise(
        _die_block, v_ix, time_index, ("row",), antenna1, ("row",), antenna2, ("row",),
        die1_jones, None if die1_jones is None else g_ix, base_vis, None if base_vis is None else v_ix,
        die2_jones, None if die2_jones is None else g_ix, block_ids, ("row",),
        align_arrays=False, adjust_chunks={"row": row_chunks},
        meta=np.empty((0,) * len(v_ix), dtype=out_dtype), dtype=out_dtype)


# ---------------------------------------------------------------------------- fused predict from source-level inputs
def _fused_block(time_index, antenna1, antenna2, lm, uvw, frequency, brightness, beam, extents, freq_map, pa, pe,
                 ascale, feed_rot, gauss_shape, stokes, spi, ref_freq, running, block_id=None, convention="fourier",
                 corr_schema=None, spectral_base=0):
    """One (source chunk, row chunk, chan chunk) block: every contracted axis (lm / uvw components, antennas, the
    beam cube's axes) arrives as nested one-element lists; ``running`` is the previous link of a ``streams=True``
    chain (its length-1 source axis still on)."""
    from .fused import fused_predict_vis as _np_fused, cached_plan, _all_single
    u = lambda x: None if x is None else _first(x)
    pa_ = u(pa)
    with placement.block(block_id):
        uvw_ = u(uvw)
        single = _all_single(u(lm), uvw_, frequency, u(brightness), u(feed_rot), u(beam), u(extents), u(freq_map),
                             pa_, u(pe), u(ascale), u(stokes), u(spi), u(ref_freq))
        plan = None if beam is None else cached_plan(time_index, antenna1, antenna2, pa_.shape[1],
                                                      uvw=None if gauss_shape is not None else uvw_, single=single)
        vis = _np_fused(time_index, antenna1, antenna2, u(lm), uvw_, frequency, u(brightness), u(beam), u(extents),
                        u(freq_map), pa_, u(pe), u(ascale), None, None, None, convention, u(feed_rot), u(gauss_shape),
                        u(stokes), u(spi), u(ref_freq), corr_schema, spectral_base, plan)
    if running is not None:
        vis = vis + running[0]
    return vis[None, ...]


def _single_chunk(arr, axes, message):
    if arr is not None and any(len(arr.chunks[ax]) != 1 for ax in axes):
        raise ValueError(message)


def fused_predict_vis(time_index, antenna1, antenna2, lm, uvw, frequency, brightness=None,
                      beam=None, beam_lm_extents=None, beam_freq_map=None, parallactic_angles=None,
                      point_errors=None, antenna_scaling=None,
                      die1_jones=None, base_vis=None, die2_jones=None, convention="fourier",
                      feed_rotation=None, gauss_shape=None, stokes=None, spi=None, ref_freq=None,
                      corr_schema=(("XX", "XY"), ("YX", "YY")), spectral_base=0, streams=None):
    """
    dask front-end of :func:`codex_africanus_amd.rime.fused_predict_vis`: the graph the reference builds from
    source-level inputs (africanus/rime/examples/predict.py:404-525: ``rime.dask.phase_delay`` -> ``da.einsum`` ->
    ``rime.dask.beam_cube_dde`` [-> feed rotation einsum] -> ``rime.dask.predict_vis``) as ONE blockwise call whose
    block is one fused device call: neither the (source, row, chan, 2, 2) coherencies nor the (source, time, ant, chan,
    2, 2) beam terms ever exist.  The chunk contract is the reference's: row chunks of ``time_index`` / ``antenna1`` /
    ``antenna2`` / ``uvw`` / ``base_vis`` agree; every per-time array (``parallactic_angles``, ``point_errors``,
    ``feed_rotation``, ``die{1,2}_jones``) has as many time chunks as there are row chunks, row chunk k indexing into
    time chunk k (every block normalises ``time_index`` by its own minimum, africanus/rime/predict.py:597); the antenna
    axis, the beam cube, its extents and its frequency map are single chunks (africanus/rime/dask_predict.py:478-524,
    africanus/rime/dask.py:177-185); source chunks are summed -- by a serial chain when ``streams=True``, else by
    ``sum(axis=0)`` -- and ``base_vis`` / the DIEs are applied after that sum (africanus/rime/dask_predict.py:372-439).
    Row block k runs on GPU k % n_devices (``placement``); the row layout's plan is cached per row chunk.
    """
    _need_dask()
    if (die1_jones is None) != (die2_jones is None):
        raise ValueError("Both die1_jones and die2_jones must be present or absent")
    beam_args = (beam, beam_lm_extents, beam_freq_map, parallactic_angles, point_errors, antenna_scaling)
    have_beam = beam is not None
    if any((a is None) != (not have_beam) for a in beam_args):
        raise ValueError("beam, beam_lm_extents, beam_freq_map, parallactic_angles, point_errors and "
                         "antenna_scaling must all be present or all absent")
    model = stokes is not None or spi is not None or ref_freq is not None
    if model and (brightness is not None or stokes is None or spi is None or ref_freq is None):
        raise ValueError("pass either brightness or all of stokes, spi and ref_freq")
    if not model and brightness is None:
        raise ValueError("pass either brightness or all of stokes, spi and ref_freq")
    if feed_rotation is not None and not have_beam:
        raise ValueError("feed_rotation multiplies the beam term: pass the beam arguments as well")
    row_chunks = time_index.chunks[0]
    for name, a in (("antenna1", antenna1), ("antenna2", antenna2), ("uvw", uvw), ("base_vis", base_vis)):
        if a is not None and a.chunks[0] != row_chunks:
            raise ValueError("%s row chunks %s do not match time_index row chunks %s" % (name, a.chunks[0], row_chunks))
    _single_chunk(beam, range(beam.ndim) if have_beam else (), "Beam chunking unsupported")
    _single_chunk(beam_freq_map, (0,), "Beam frequency map chunking unsupported")
    _single_chunk(beam_lm_extents, (0, 1), "Chunking of beam_lm_extents unsupported")
    for name, a, ant_axis in (("parallactic_angles", parallactic_angles, 1), ("point_errors", point_errors, 1),
                              ("feed_rotation", feed_rotation, 1), ("die1_jones", die1_jones, 1),
                              ("die2_jones", die2_jones, 1), ("antenna_scaling", antenna_scaling, 0)):
        if a is None:
            continue
        if len(a.chunks[ant_axis]) != 1:
            raise ValueError("Subdivision of antenna dimension into multiple chunks is not supported.")
        if name != "antenna_scaling" and len(a.chunks[0]) != len(row_chunks):
            raise ValueError("Number of row chunks (%s) does not equal number of time chunks (%s)."
                             % (row_chunks, a.chunks[0]))
    src_chunks = lm.chunks[0]
    for name, a in (("brightness", brightness), ("gauss_shape", gauss_shape), ("stokes", stokes), ("spi", spi),
                    ("ref_freq", ref_freq)):
        if a is not None and a.chunks[0] != src_chunks:
            raise ValueError("%s source chunks %s do not match lm source chunks %s" % (name, a.chunks[0], src_chunks))
    chan_chunks = frequency.chunks[0]
    flat = brightness is not None and brightness.ndim == 3
    for name, a, ax in (("brightness", None if flat else brightness, 1), ("point_errors", point_errors, 2),
                        ("antenna_scaling", antenna_scaling, 1), ("die1_jones", die1_jones, 2),
                        ("die2_jones", die2_jones, 2), ("base_vis", base_vis, 1)):
        if a is not None and a.chunks[ax] != chan_chunks:
            raise ValueError("%s chan chunks %s do not match frequency chunks %s" % (name, a.chunks[ax], chan_chunks))

    out_ix = ("src", "row", "chan", "corr-1", "corr-2")
    ix = lambda a, names: None if a is None else names
    block_ids = _row_block_ids(len(row_chunks))
    b_ix = ("src", "corr-1", "corr-2") if flat else ("src", "chan", "corr-1", "corr-2")
    new_axes = {"corr-1": 2, "corr-2": 2} if brightness is None else None
    # the promoted type of the floating-point inputs (rime.fused_predict_vis's rule, africanus/util/type_inference.py:24-26)
    dtype = np.result_type(np.complex64, *[a.dtype for a in (lm, uvw, frequency, brightness, beam, beam_lm_extents, beam_freq_map,
                                                             parallactic_angles, point_errors, antenna_scaling, feed_rotation,
                                                             gauss_shape, stokes, spi, ref_freq) if a is not None])

    def blocks(src_sel, running):
        sel = lambda a: None if a is None else a.blocks[src_sel]
        lm_, b_, gs_, st_, sp_, rf_ = (sel(a) for a in (lm, brightness, gauss_shape, stokes, spi, ref_freq))
        return da.blockwise(
            _fused_block, out_ix, time_index, ("row",), antenna1, ("row",), antenna2, ("row",),
            lm_, ("src", "lmc"), uvw, ("row", "uvwc"), frequency, ("chan",), b_, ix(b_, b_ix),
            beam, ix(beam, ("bl", "bm", "bf", "bc-1", "bc-2")), beam_lm_extents, ix(beam_lm_extents, ("e1", "e2")),
            beam_freq_map, ix(beam_freq_map, ("bf",)), parallactic_angles, ix(parallactic_angles, ("row", "ant")),
            point_errors, ix(point_errors, ("row", "ant", "chan", "pec")),
            antenna_scaling, ix(antenna_scaling, ("ant", "chan", "asc")),
            feed_rotation, ix(feed_rotation, ("row", "ant", "fr-1", "fr-2")), gs_, ix(gs_, ("src", "gsc")),
            st_, ix(st_, ("src", "pol")), sp_, ix(sp_, ("src", "spi", "pol")), rf_, ix(rf_, ("src",)),
            running, ix(running, out_ix), block_ids, ("row",),
            align_arrays=False, adjust_chunks={"row": row_chunks, "src": 1}, new_axes=new_axes,
            convention=convention, corr_schema=tuple(tuple(r) for r in corr_schema), spectral_base=spectral_base,
            meta=np.empty((0,) * 5, dtype=dtype), dtype=dtype)

    if streams is True:
        running = None
        for k in range(len(src_chunks)):
            running = blocks(slice(k, k + 1), running)
        summed = running[0]
    else:
        summed = blocks(slice(None), None).sum(axis=0)
    if die1_jones is None and base_vis is None:
        return summed
    # base_vis is added, then the DIEs applied, in the reference's order (africanus/rime/predict.py:605-612)
    return predict_vis(time_index, antenna1, antenna2, None, None, None, die1_jones,
                       summed if base_vis is None else base_vis + summed, die2_jones)
