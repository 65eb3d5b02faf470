"""dask.array front-ends of the calibration consumers with the signatures and chunk rules of
africanus/calibration/utils/dask.py:25-295: rows are chunked in whole time bins (``time_bin_indices`` /
``time_bin_counts`` are chunked along the same 'row' index as the per-row arrays and carry chunk-local offsets that
the kernels normalise), jones is chunked over time bins but never over antenna or direction."""
try:
    import dask.array as da
    _dask_error = None
except ImportError as e:  # pragma: no cover - depends on the environment
    da = None
    _dask_error = e

from . import corrupt_vis as _np_corrupt_vis, residual_vis as _np_residual_vis, correct_vis as _np_correct_vis
from . import check_type, DIAG_DIAG, DIAG, FULL


def _need_dask():
    if da is None:
        raise ImportError("dask.array is required for codex_africanus_amd.calibration.utils.dask: %s" % (_dask_error,))


def _shapes(mode):
    if mode == DIAG_DIAG:
        return ("row", "chan", "corr1"), ("row", "chan", "dir", "corr1"), ("row", "ant", "chan", "dir", "corr1")
    if mode == DIAG:
        return (("row", "chan", "corr1", "corr2"), ("row", "chan", "dir", "corr1", "corr2"),
                ("row", "ant", "chan", "dir", "corr1"))
    if mode == FULL:
        return (("row", "chan", "corr1", "corr2"), ("row", "chan", "dir", "corr1", "corr2"),
                ("row", "ant", "chan", "dir", "corr1", "corr2"))
    raise ValueError("Unknown mode argument of %s" % mode)


def _check_jones(jones):
    if jones.chunks[1][0] != jones.shape[1]:
        raise ValueError("Cannot chunk jones over antenna")
    if jones.chunks[3][0] != jones.shape[3]:
        raise ValueError("Cannot chunk jones over direction")


def _corrupt_block(tbi, tbc, a1, a2, jones, model):
    return _np_corrupt_vis(tbi, tbc, a1, a2, jones[0][0], model[0])


def corrupt_vis(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, model):
    _need_dask()
    mode = check_type(jones, model, vis_type="model")
    _check_jones(jones)
    if model.chunks[2][0] != model.shape[2]:
        raise ValueError("Cannot chunk model over direction")
    out_shape, model_shape, jones_shape = _shapes(mode)
    return da.blockwise(_corrupt_block, out_shape, time_bin_indices, ("row",), time_bin_counts, ("row",),
                        antenna1, ("row",), antenna2, ("row",), jones, jones_shape, model, model_shape,
                        adjust_chunks={"row": antenna1.chunks[0]}, dtype=model.dtype, align_arrays=False)


def _residual_block(tbi, tbc, a1, a2, jones, vis, flag, model):
    return _np_residual_vis(tbi, tbc, a1, a2, jones[0][0], vis, flag, model[0])


def residual_vis(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, vis, flag, model):
    _need_dask()
    mode = check_type(jones, vis)
    _check_jones(jones)
    if model.chunks[2][0] != model.shape[2]:
        raise ValueError("Cannot chunk model over direction")
    out_shape, model_shape, jones_shape = _shapes(mode)
    return da.blockwise(_residual_block, out_shape, time_bin_indices, ("row",), time_bin_counts, ("row",),
                        antenna1, ("row",), antenna2, ("row",), jones, jones_shape, vis, out_shape, flag, out_shape,
                        model, model_shape, adjust_chunks={"row": antenna1.chunks[0]}, dtype=vis.dtype,
                        align_arrays=False)


def _correct_block(tbi, tbc, a1, a2, jones, vis, flag):
    return _np_correct_vis(tbi, tbc, a1, a2, jones[0][0], vis, flag)


def correct_vis(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, vis, flag):
    _need_dask()
    mode = check_type(jones, vis)
    _check_jones(jones)
    out_shape, _, jones_shape = _shapes(mode)
    return da.blockwise(_correct_block, out_shape, time_bin_indices, ("row",), time_bin_counts, ("row",),
                        antenna1, ("row",), antenna2, ("row",), jones, jones_shape, vis, out_shape, flag, out_shape,
                        adjust_chunks={"row": antenna1.chunks[0]}, dtype=vis.dtype, align_arrays=False)
