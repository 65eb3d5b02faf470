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
