"""
dask.array front-ends of the wgridder operators with the signatures of africanus/gridding/wgridder/dask.py:53-463.

Blocks are (row chunk, band): ``freq`` and the channel axis of ``vis`` / ``weights`` / ``flag`` are chunked one imaging
band per chunk, ``freq_bin_idx`` / ``freq_bin_counts`` (and the band axis of ``image``) one entry per chunk, ``uvw``
over rows only.  ``model`` returns (row, chan) visibilities; ``dirty`` / ``residual`` / ``hessian`` compute one image
per row chunk and sum them (every row chunk picks its own w-planes from its own w range, so chunked and unchunked
results agree to ``epsilon``, not to rounding -- as in the reference, tests/test_wgridder.py:357-600).  Row block k runs
on GPU k % n_devices (codex_africanus_amd/placement.py).
"""
import numpy as np

try:
    import dask.array as da
    _dask_error = None
except ImportError as e:  # pragma: no cover - depends on the environment
    da = None
    _dask_error = e

from ... import placement
from . import im2vis as _fwd, vis2im as _adj


def _first(x):
    while isinstance(x, list):
        x = x[0]
    return x


def _need_dask():
    if da is None:
        raise ImportError("dask.array is required for codex_africanus_amd.gridding.wgridder.dask: %s" % (_dask_error,))


def _model_block(uvw, freq, image, fbi, fbc, weights, flag, block_id, cell, celly, epsilon, do_wstacking):
    with placement.block(block_id):
        return _fwd.model(_first(uvw), freq, _first(image), fbi, fbc, cell, weights, flag, celly, epsilon, 1, do_wstacking)


def _dirty_block(uvw, freq, vis, fbi, fbc, weights, flag, block_id, nx, ny, cell, celly, epsilon, do_wstacking):
    with placement.block(block_id):
        return _adj.dirty(_first(uvw), freq, vis, fbi, fbc, nx, ny, cell, weights, flag, celly, epsilon, 1,
                          do_wstacking)[None]


def _residual_block(uvw, freq, image, vis, fbi, fbc, weights, flag, block_id, cell, celly, epsilon, do_wstacking):
    with placement.block(block_id):
        return _adj.residual(_first(uvw), freq, image, vis, fbi, fbc, cell, weights, flag, celly, epsilon, 1,
                             do_wstacking)[None]


def _hessian_block(uvw, freq, image, fbi, fbc, weights, flag, block_id, cell, celly, epsilon, do_wstacking):
    with placement.block(block_id):
        return _adj.hessian(_first(uvw), freq, image, fbi, fbc, cell, weights, flag, celly, epsilon, 1,
                            do_wstacking)[None]


def _opt(x):
    return (x, None if x is None else ("row", "chan"))


def _row_ids(uvw):
    return da.arange(len(uvw.chunks[0]), chunks=1, dtype=np.int64), ("row",)


def model(uvw, freq, image, freq_bin_idx, freq_bin_counts, cell, weights=None, flag=None, celly=None, epsilon=1e-5,
          nthreads=1, do_wstacking=True):
    """africanus/gridding/wgridder/dask.py:53-117"""
    _need_dask()
    return da.blockwise(_model_block, ("row", "chan"), uvw, ("row", "three"), freq, ("chan",),
                        image, ("chan", "nx", "ny"), freq_bin_idx, ("chan",), freq_bin_counts, ("chan",),
                        *_opt(weights), *_opt(flag), *_row_ids(uvw),
                        cell=cell, celly=celly, epsilon=epsilon, do_wstacking=do_wstacking,
                        adjust_chunks={"chan": freq.chunks[0]}, dtype=np.result_type(image.dtype, np.complex64),
                        align_arrays=False)


def dirty(uvw, freq, vis, freq_bin_idx, freq_bin_counts, nx, ny, cell, weights=None, flag=None, celly=None,
          epsilon=1e-5, nthreads=1, do_wstacking=True, double_accum=False):
    """africanus/gridding/wgridder/dask.py:157-239"""
    _need_dask()
    if vis.dtype == np.complex128:
        real_type = np.float64
    elif vis.dtype == np.complex64:
        real_type = np.float32
    else:
        raise ValueError("Vis of incorrect type")
    ims = da.blockwise(_dirty_block, ("row", "chan", "nx", "ny"), uvw, ("row", "three"), freq, ("chan",),
                       vis, ("row", "chan"), freq_bin_idx, ("chan",), freq_bin_counts, ("chan",),
                       *_opt(weights), *_opt(flag), *_row_ids(uvw),
                       nx=nx, ny=ny, cell=cell, celly=celly, epsilon=epsilon, do_wstacking=do_wstacking,
                       adjust_chunks={"chan": freq_bin_idx.chunks[0], "row": (1,) * len(vis.chunks[0])},
                       new_axes={"nx": nx, "ny": ny}, dtype=real_type, align_arrays=False)
    return ims.sum(axis=0)


def residual(uvw, freq, image, vis, freq_bin_idx, freq_bin_counts, cell, weights=None, flag=None, celly=None,
             epsilon=1e-5, nthreads=1, do_wstacking=True, double_accum=False):
    """africanus/gridding/wgridder/dask.py:276-351"""
    _need_dask()
    ims = da.blockwise(_residual_block, ("row", "chan", "nx", "ny"), uvw, ("row", "three"), freq, ("chan",),
                       image, ("chan", "nx", "ny"), vis, ("row", "chan"), freq_bin_idx, ("chan",),
                       freq_bin_counts, ("chan",), *_opt(weights), *_opt(flag), *_row_ids(uvw),
                       cell=cell, celly=celly, epsilon=epsilon, do_wstacking=do_wstacking,
                       adjust_chunks={"chan": freq_bin_idx.chunks[0], "row": (1,) * len(vis.chunks[0])},
                       dtype=image.dtype, align_arrays=False)
    return ims.sum(axis=0)


def hessian(uvw, freq, image, freq_bin_idx, freq_bin_counts, cell, weights=None, flag=None, celly=None, epsilon=1e-5,
            nthreads=1, do_wstacking=True, double_accum=False):
    """africanus/gridding/wgridder/dask.py:386-456"""
    _need_dask()
    ims = da.blockwise(_hessian_block, ("row", "chan", "nx", "ny"), uvw, ("row", "three"), freq, ("chan",),
                       image, ("chan", "nx", "ny"), freq_bin_idx, ("chan",), freq_bin_counts, ("chan",),
                       *_opt(weights), *_opt(flag), *_row_ids(uvw),
                       cell=cell, celly=celly, epsilon=epsilon, do_wstacking=do_wstacking,
                       adjust_chunks={"chan": freq_bin_idx.chunks[0], "row": (1,) * len(uvw.chunks[0])},
                       dtype=image.dtype, align_arrays=False)
    return ims.sum(axis=0)
