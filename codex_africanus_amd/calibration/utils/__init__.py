"""corrupt_vis / residual_vis / correct_vis / check_type / chunkify_rows with the signatures of
africanus/calibration/utils (corrupt_vis.py:58, residual_vis.py:63, correct_vis.py:63, utils.py:11,48)."""
import numpy as np

from ... import _lib
from ..._device import Call, np_dtype_of, _is_torch

DIAG_DIAG, DIAG, FULL = 0, 1, 2


def check_type(jones, vis, vis_type="vis"):
    """Calibration scenario of a (jones, vis) pair: 0 DIAG_DIAG, 1 DIAG, 2 FULL
    (africanus/calibration/utils/utils.py:11-45, same errors)."""
    if vis_type == "vis":
        vis_ndim = (3, 4)
    elif vis_type == "model":
        vis_ndim = (4, 5)
    else:
        raise ValueError("Unknown vis_type")
    vdim, jdim = len(vis.shape), len(jones.shape)
    if vdim == vis_ndim[0]:
        if jdim != 5:
            raise RuntimeError("Jones axes not compatible with visibility axes. Expected length 5 but got "
                               "length %d" % jdim)
        return DIAG_DIAG
    if vdim == vis_ndim[1]:
        if jdim == 5:
            return DIAG
        if jdim == 6:
            return FULL
        raise RuntimeError("Jones term has incorrect shape")
    raise RuntimeError("Visibility data has incorrect shape")


def chunkify_rows(time, utimes_per_chunk):
    """Row chunks holding whole unique times, with the bin starts and counts the kernels take
    (africanus/calibration/utils/utils.py:48-61).  Host-side bookkeeping on the TIME column."""
    time = time.detach().cpu().numpy() if _is_torch(time) else np.asarray(time)
    utimes, time_bin_counts = np.unique(time, return_counts=True)
    n_time = len(utimes)
    if utimes_per_chunk <= 0:
        utimes_per_chunk = n_time
    row_chunks = [int(np.sum(time_bin_counts[i:i + utimes_per_chunk])) for i in range(0, n_time, utimes_per_chunk)]
    time_bin_indices = np.zeros(n_time, dtype=np.int32)
    time_bin_indices[1:] = np.cumsum(time_bin_counts)[:-1]
    return tuple(row_chunks), time_bin_indices, time_bin_counts.astype(np.int32)


def _prepare(jones, arrays, vis_like, vis_type):
    mode = check_type(jones, vis_like, vis_type)
    for a in (jones,) + tuple(arrays):
        if int(a.shape[-1]) > 2:
            raise ValueError("ncorr cant be larger than 2")
    ncorr = int(vis_like.shape[-1])
    J = (ncorr,) if mode == DIAG_DIAG else ((2,) if mode == DIAG else (2, 2))
    if tuple(int(s) for s in jones.shape[4:]) != J:
        raise ValueError("jones correlation shape %s does not fit the visibilities" % (tuple(jones.shape[4:]),))
    return mode, ncorr


def _check_bins(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, nrow):
    """What the kernels rely on: one count per bin start, as many gain time slots as bins, antenna numbers inside the
    gain array.  Shapes always; values only for host-resident (numpy) arrays -- a device tensor is trusted rather than
    synchronised on.  The reference indexes out of bounds in these cases (numba has no bounds check)."""
    if tuple(time_bin_indices.shape) != tuple(time_bin_counts.shape) or len(time_bin_indices.shape) != 1:
        raise ValueError("time_bin_indices and time_bin_counts must be 1-D and equally long")
    if int(jones.shape[0]) < int(time_bin_indices.shape[0]):
        raise ValueError("jones holds %d time slots for %d time bins" % (int(jones.shape[0]), int(time_bin_indices.shape[0])))
    if tuple(antenna1.shape) != (nrow,) or tuple(antenna2.shape) != (nrow,):
        raise ValueError("antenna1 and antenna2 must have one entry per row")
    nant = int(jones.shape[1])
    if nrow and isinstance(antenna1, np.ndarray) and isinstance(antenna2, np.ndarray):
        lo, hi = min(int(antenna1.min()), int(antenna2.min())), max(int(antenna1.max()), int(antenna2.max()))
        if lo < 0 or hi >= nant:
            raise ValueError("antenna indices span [%d, %d], jones holds %d antennas" % (lo, hi, nant))
    if isinstance(time_bin_counts, np.ndarray) and time_bin_counts.size and int(time_bin_counts.min()) < 0:
        raise ValueError("time_bin_counts must not be negative")


def _result_dtype(*arrays):
    return np.result_type(np.complex64, *[np_dtype_of(a) for a in arrays])


def corrupt_vis(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, model):
    """
    ``vis[row,chan] = sum_dir G_p[t,chan,dir] . model[row,chan,dir] . G_q[t,chan,dir]^H`` with ``t`` the time
    bin of the row.  Same contract as ``africanus.calibration.utils.corrupt_vis``
    (africanus/calibration/utils/corrupt_vis.py:58-101): ``jones`` (time, ant, chan, dir, corr[, corr]),
    ``model`` (row, chan, dir, corr[, corr]) -> (row, chan, corr[, corr]) of ``model``'s dtype.  The caller's
    ``time_bin_indices`` is NOT modified (the reference normalises it in place).
    """
    mode, ncorr = _prepare(jones, (model,), model, "model")
    nrow, nchan, ndir = (int(s) for s in model.shape[:3])
    ntime, nant = int(jones.shape[0]), int(jones.shape[1])
    out_shape = tuple(int(s) for s in model.shape[:2]) + tuple(int(s) for s in model.shape[3:])
    out_dtype = np_dtype_of(model)
    _check_bins(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, nrow)
    with Call(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, model) as c:
        p_tbi, p_tbc = c.inp(time_bin_indices, np.int64), c.inp(time_bin_counts, np.int64)
        p_a1, p_a2 = c.inp(antenna1, np.int64), c.inp(antenna2, np.int64)
        p_j, p_m = c.inp(jones, np.complex128), c.inp(model, np.complex128)
        p_out, h = c.out(out_shape, np.complex128)
        ws = int(_lib.load().af_calibration_workspace_bytes(nrow))
        p_ws = c.scratch(ws)
        _lib.call("af_corrupt_vis_c128", p_tbi, p_tbc, int(time_bin_indices.shape[0]), p_a1, p_a2, p_j, p_m, nrow, nant,
                  nchan, ndir, mode, ncorr, p_out, p_ws, max(ws, 256), c.stream)
        return c.result(h, cast=None if out_dtype == np.complex128 else out_dtype)


def compute_and_corrupt_vis(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, model, uvw, freq, lm):
    """
    ``corrupt_vis`` with the model coherencies of a time-variable point-source model formed on the fly:
    ``source_vis = model[t,chan,dir] exp(-2 pi i nu/c (u l + v m + w (n-1))) / n``.  Same contract as
    ``africanus.calibration.utils.compute_and_corrupt_vis``
    (africanus/calibration/utils/compute_and_corrupt_vis.py:73-152): ``model`` (time, chan, dir, corr[, corr]),
    ``uvw`` (row, 3), ``freq`` (chan,), ``lm`` (time, dir, 2) -> (row, chan, corr[, corr]) of ``jones``'s dtype.
    """
    mode, ncorr = _prepare(jones, (model,), model, "model")
    ntime, nant = int(jones.shape[0]), int(jones.shape[1])
    nrow, nchan, ndir = int(uvw.shape[0]), int(model.shape[1]), int(model.shape[2])
    if tuple(int(s) for s in lm.shape) != (int(model.shape[0]), ndir, 2) or int(freq.shape[0]) != nchan:
        raise ValueError("model (time, chan, dir, corr...), lm (time, dir, 2) and freq (chan,) disagree")
    out_shape = (nrow, nchan) + tuple(int(s) for s in model.shape[3:])
    out_dtype = np_dtype_of(jones)
    _check_bins(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, nrow)
    with Call(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, model, uvw, freq, lm) as c:
        p_tbi, p_tbc = c.inp(time_bin_indices, np.int64), c.inp(time_bin_counts, np.int64)
        p_a1, p_a2 = c.inp(antenna1, np.int64), c.inp(antenna2, np.int64)
        p_j, p_m = c.inp(jones, np.complex128), c.inp(model, np.complex128)
        p_uvw, p_fr, p_lm = c.inp(uvw, np.float64), c.inp(freq, np.float64), c.inp(lm, np.float64)
        p_out, h = c.out(out_shape, np.complex128)
        ws = int(_lib.load().af_calibration_workspace_bytes(nrow))
        p_ws = c.scratch(ws)
        _lib.call("af_compute_and_corrupt_vis_c128", p_tbi, p_tbc, int(time_bin_indices.shape[0]), p_a1, p_a2, p_j, p_m,
                  p_uvw, p_fr, p_lm, nrow, nant, nchan, ndir, mode, ncorr, p_out, p_ws, max(ws, 256), c.stream)
        return c.result(h, cast=None if out_dtype == np.complex128 else out_dtype)


def residual_vis(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, vis, flag, model):
    """
    ``residual = vis - sum_dir G_p model_dir G_q^H`` where no correlation of the (row, chan) cell is flagged,
    0 elsewhere.  Same contract as ``africanus.calibration.utils.residual_vis``
    (africanus/calibration/utils/residual_vis.py:63-119); result in ``vis``'s dtype.
    """
    mode, ncorr = _prepare(jones, (vis, model), vis, "vis")
    nrow, nchan, ndir = (int(s) for s in model.shape[:3])
    if tuple(vis.shape) != tuple(flag.shape) or tuple(vis.shape[:2]) != (nrow, nchan):
        raise ValueError("vis, flag (row, chan, corr...) and model (row, chan, dir, corr...) disagree")
    ntime, nant = int(jones.shape[0]), int(jones.shape[1])
    out_dtype = np_dtype_of(vis)
    _check_bins(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, nrow)
    with Call(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, vis, flag, model) as c:
        p_tbi, p_tbc = c.inp(time_bin_indices, np.int64), c.inp(time_bin_counts, np.int64)
        p_a1, p_a2 = c.inp(antenna1, np.int64), c.inp(antenna2, np.int64)
        p_j, p_v, p_m = c.inp(jones, np.complex128), c.inp(vis, np.complex128), c.inp(model, np.complex128)
        p_f = c.inp(flag, np.uint8)
        p_out, h = c.out(tuple(int(s) for s in vis.shape), np.complex128)
        ws = int(_lib.load().af_calibration_workspace_bytes(nrow))
        p_ws = c.scratch(ws)
        _lib.call("af_residual_vis_c128", p_tbi, p_tbc, int(time_bin_indices.shape[0]), p_a1, p_a2, p_j, p_v, p_f, p_m,
                  nrow, nant, nchan, ndir, mode, ncorr, p_out, p_ws, max(ws, 256), c.stream)
        return c.result(h, cast=None if out_dtype == np.complex128 else out_dtype)


def correct_vis(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, vis, flag):
    """
    ``corrected = G_p^-1 . vis . G_q^-H`` for direction-independent gains (``jones.shape[3] == 1``) where
    unflagged, 0 elsewhere.  Same contract as ``africanus.calibration.utils.correct_vis``
    (africanus/calibration/utils/correct_vis.py:63-115); result in ``vis``'s dtype.
    """
    mode, ncorr = _prepare(jones, (vis,), vis, "vis")
    if int(jones.shape[3]) > 1:
        raise ValueError("Jones has n_dir > 1. Cannot correct for direction dependent gains")
    nrow, nchan = int(vis.shape[0]), int(vis.shape[1])
    if tuple(vis.shape) != tuple(flag.shape):
        raise ValueError("vis and flag must have the same shape")
    ntime, nant = int(jones.shape[0]), int(jones.shape[1])
    out_dtype = np_dtype_of(vis)
    _check_bins(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, nrow)
    with Call(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, vis, flag) as c:
        p_tbi, p_tbc = c.inp(time_bin_indices, np.int64), c.inp(time_bin_counts, np.int64)
        p_a1, p_a2 = c.inp(antenna1, np.int64), c.inp(antenna2, np.int64)
        p_j, p_v = c.inp(jones, np.complex128), c.inp(vis, np.complex128)
        p_f = c.inp(flag, np.uint8)
        p_out, h = c.out(tuple(int(s) for s in vis.shape), np.complex128)
        # FULL gains: room for the per-(time, antenna, chan) inverse gains (inverted once instead of once per baseline)
        ws = int(_lib.load().af_correct_vis_workspace_bytes(nrow, ntime, nant, nchan)) if mode == 2 else \
            int(_lib.load().af_calibration_workspace_bytes(nrow))
        p_ws = c.scratch(ws)
        _lib.call("af_correct_vis_c128", p_tbi, p_tbc, ntime, p_a1, p_a2, p_j, p_v, p_f, nrow, nant, nchan,
                  int(jones.shape[3]), mode, ncorr, p_out, p_ws, max(ws, 256), c.stream)
        return c.result(h, cast=None if out_dtype == np.complex128 else out_dtype)
