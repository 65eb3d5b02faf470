"""
dask.array front-end of ``im_to_vis`` with the signature of ``africanus.dft.dask.im_to_vis``
(africanus/dft/dask.py:26-51): blocks over (row, chan); the source axis must be a single chunk
and image / frequency channel chunks must agree (same ``ValueError``s, :29-36).
"""
import numpy as np

try:
    import dask.array as da
    _dask_error = None
except ImportError as e:  # pragma: no cover - depends on the environment
    da = None
    _dask_error = e

from .kernels import im_to_vis as _np_im_to_vis, vis_to_im as _np_vis_to_im
from .. import placement


def _first(x):
    while isinstance(x, list):
        x = x[0]
    return x


def _im_to_vis_block(image, uvw, lm, frequency, block_id=None, convention="fourier", dtype_=None):
    # block_id: the row block's number (a one-element array riding along the "row" axis): row block k runs on
    # GPU k % n_devices (codex_africanus_amd/placement.py)
    with placement.block(block_id):
        return _np_im_to_vis(_first(image), _first(uvw), _first(lm), frequency,
                             convention=convention, dtype=dtype_)


def im_to_vis(image, uvw, lm, frequency, convention="fourier", dtype=np.complex128):
    if da is None:
        raise ImportError("dask.array is required for codex_africanus_amd.dft.dask: %s" % (_dask_error,))
    if lm.chunks[0][0] != lm.shape[0]:
        raise ValueError("lm chunks must match lm shape on first axis")
    if image.chunks[0][0] != image.shape[0]:
        raise ValueError("Image chunks must match image shape on first axis")
    if image.chunks[0][0] != lm.chunks[0][0]:
        raise ValueError("Image chunks and lm chunks must match on first axis")
    if image.chunks[1] != frequency.chunks[0]:
        raise ValueError("Image chunks must match frequency chunks on second axis")
    # blocks are matched by position (align_arrays=False): the checks above already made the chunkings agree
    return da.blockwise(_im_to_vis_block, ("row", "chan", "corr"), image, ("src", "chan", "corr"),
                        uvw, ("row", "uvwc"), lm, ("src", "lmc"), frequency, ("chan",),
                        da.arange(len(uvw.chunks[0]), chunks=1, dtype=np.int64), ("row",),
                        align_arrays=False, convention=convention, dtype_=dtype, dtype=dtype)


def _vis_to_im_block(vis, uvw, lm, frequency, flags, block_id=None, convention="fourier", dtype_=None):
    with placement.block(block_id):
        return _np_vis_to_im(vis, _first(uvw), _first(lm), frequency, flags, convention=convention,
                             dtype=dtype_)[None, ...]


def vis_to_im(vis, uvw, lm, frequency, flags, convention="fourier", dtype=np.float64):
    """``africanus.dft.dask.vis_to_im`` (africanus/dft/dask.py:60-90): one image per row chunk,
    summed over the row chunks; same chunk checks."""
    if da is None:
        raise ImportError("dask.array is required for codex_africanus_amd.dft.dask: %s" % (_dask_error,))
    if vis.chunks[0] != uvw.chunks[0]:
        raise ValueError("Vis chunks and uvw chunks must match on first axis")
    if vis.chunks[1] != frequency.chunks[0]:
        raise ValueError("Vis chunks must match frequency chunks on second axis")
    if vis.chunks != flags.chunks:
        raise ValueError("Vis chunks must match flags chunks on all axes")
    ims = da.blockwise(_vis_to_im_block, ("row", "src", "chan", "corr"), vis, ("row", "chan", "corr"),
                       uvw, ("row", "uvwc"), lm, ("src", "lmc"), frequency, ("chan",),
                       flags, ("row", "chan", "corr"),
                       da.arange(len(uvw.chunks[0]), chunks=1, dtype=np.int64), ("row",),
                       align_arrays=False, adjust_chunks={"row": 1},
                       convention=convention, dtype_=dtype, dtype=dtype)
    return ims.sum(axis=0)
