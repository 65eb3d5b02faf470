"""predict_vis / apply_gains with the signatures of africanus/rime/predict.py:466-488,622-626."""
import numpy as np

from .. import _lib
from .._device import Call, np_dtype_of

JONES_NOT_PRESENT = 0
JONES_1_OR_2 = 1
JONES_2X2 = 2


def _ndim(a):
    return len(a.shape)


def predict_checks(time_index, antenna1, antenna2, dde1_jones, source_coh, dde2_jones,
                   die1_jones, base_vis, die2_jones):
    """
    Argument validation with the behaviour of ``africanus.rime.predict.predict_checks``
    (africanus/rime/predict.py:380-463): same conditions, same ``ValueError`` messages,
    returns the six presence booleans.
    """
    have_ddes1 = dde1_jones is not None
    have_coh = source_coh is not None
    have_ddes2 = dde2_jones is not None
    have_dies1 = die1_jones is not None
    have_bvis = base_vis is not None
    have_dies2 = die2_jones is not None

    for name, arr in (("time_index", time_index), ("antenna1", antenna1), ("antenna2", antenna2)):
        if _ndim(arr) != 1:
            raise ValueError("%s.ndim != 1" % name)

    if have_ddes1 != have_ddes2:
        raise ValueError("Both dde1_jones and dde2_jones must be present or absent")
    if have_dies1 != have_dies2:
        raise ValueError("Both die1_jones and die2_jones must be present or absent")
    have_ddes = have_ddes1 and have_ddes2
    have_dies = have_dies1 and have_dies2

    def _in(name, arr, allowed):
        if _ndim(arr) not in allowed:
            raise ValueError("%s.ndim %d not in %s" % (name, _ndim(arr), (allowed,)[0]))

    if have_ddes1:
        _in("dde1_jones", dde1_jones, (5, 6))
    if have_ddes2:
        _in("dde2_jones", dde2_jones, (5, 6))
    if have_ddes and _ndim(dde1_jones) != _ndim(dde2_jones):
        raise ValueError("dde1_jones.ndim != dde2_jones.ndim")
    if have_coh:
        _in("source_coh", source_coh, (4, 5))
    if have_dies1:
        _in("die1_jones", die1_jones, (4, 5))
    if have_bvis:
        _in("base_vis", base_vis, (3, 4))
    if have_dies2:
        _in("die2_jones", die2_jones, (4, 5))
    if have_dies and _ndim(die1_jones) != _ndim(die2_jones):
        raise ValueError("die1_jones.ndim != die2_jones.ndim")

    # every present term must imply the same dde ndim
    implied = []
    if have_ddes:
        implied.append(_ndim(dde1_jones))
    if have_coh:
        implied.append(_ndim(source_coh) + 1)
    if have_dies:
        implied.append(_ndim(die1_jones) + 1)
    if have_bvis:
        implied.append(_ndim(base_vis) + 2)
    if any(i != implied[0] for i in implied[1:]):
        raise ValueError(
            "One of the following pre-conditions is broken "
            "(missing values are ignored):\n"
            "dde_jones{1,2}.ndim == source_coh.ndim + 1\n"
            "dde_jones{1,2}.ndim == base_vis.ndim + 2\n"
            "dde_jones{1,2}.ndim == die_jones{1,2}.ndim + 1")

    return (have_ddes1, have_coh, have_ddes2, have_dies1, have_bvis, have_dies2)


def _jones_type(name, arr, corr_1_dims, corr_2_dims):
    # africanus/rime/predict.py:15-53
    if arr is None:
        return JONES_NOT_PRESENT
    if _ndim(arr) == corr_1_dims:
        return JONES_1_OR_2
    if _ndim(arr) == corr_2_dims:
        return JONES_2X2
    raise ValueError("%s.ndim not in (%d, %d)" % (name, corr_1_dims, corr_2_dims))


def _index_array(c, a):
    dt = np_dtype_of(a)
    if dt == np.dtype(np.int32):
        return c.inp(a, np.int32), 4
    if dt.kind not in "iu":
        raise TypeError("time_index/antenna arrays must be integer, got %s" % dt)
    return c.inp(a, np.int64), 8


def _check_indices(time_index, antenna1, antenna2, ntime, nant, have_jones):
    """Index ranges the kernel relies on when per-antenna terms are gathered.  The reference reads out of bounds (or
    raises an IndexError under numba's boundscheck); on the device a bad index is an out-of-bounds read that can kill
    the context, so host-resident (numpy) index arrays are checked here -- O(row), negligible next to the upload.
    Device-resident index tensors are checked by the kernels themselves (clamped read, NaN rows, a status word the
    wrapper turns into a ValueError when it next synchronises: ``_index_status_message``)."""
    if not have_jones or any(not isinstance(a, np.ndarray) for a in (time_index, antenna1, antenna2)):
        return
    if time_index.size == 0:
        return
    span = int(time_index.max()) - int(time_index.min())
    if span >= ntime:
        raise ValueError("time_index spans %d timesteps, the Jones terms hold %d" % (span + 1, ntime))
    lo = min(int(antenna1.min()), int(antenna2.min()))
    hi = max(int(antenna1.max()), int(antenna2.max()))
    if lo < 0 or hi >= nant:
        raise ValueError("antenna indices span [%d, %d], the Jones terms hold %d antennas" % (lo, hi, nant))


def _index_status_message(flags, ntime, nant):
    parts = []
    if flags & _lib.AF_STATUS_TIME_INDEX:
        parts.append("time_index - min(time_index) reaches beyond the %d timesteps of the Jones terms" % ntime)
    if flags & _lib.AF_STATUS_ANTENNA:
        parts.append("antenna1 / antenna2 hold indices outside the %d antennas of the Jones terms" % nant)
    return "predict_vis: " + " and ".join(parts)


def predict_vis(time_index, antenna1, antenna2, dde1_jones=None, source_coh=None, dde2_jones=None,
                die1_jones=None, base_vis=None, die2_jones=None):
    """
    Jones-chain reduction ``V_pq = G_p (B_pq + sum_s E_ps X_pqs E_qs^H) G_q^H``.

    Same contract as ``africanus.rime.predict_vis`` (africanus/rime/predict.py:466-619):
    ``time_index``/``antenna1``/``antenna2`` (row,) integer; ``dde{1,2}_jones``
    (source, time, ant, chan, corr...); ``source_coh`` (source, row, chan, corr...);
    ``die{1,2}_jones`` (time, ant, chan, corr...); ``base_vis`` (row, chan, corr...) with
    corr... one of (1,), (2,), (2, 2).  Any term may be None (both members of a dde / die
    pair together).  ``time_index`` is normalised by its minimum inside the call
    (predict.py:597).  Output (row, chan, corr...) of the promoted dtype of the inputs;
    the shape comes from the first present of dde, coh, die, base_vis (predict.py:255-326).
    Results are bit-identical to the numba path (same operation order, no contraction).
    """
    tup = predict_checks(time_index, antenna1, antenna2, dde1_jones, source_coh, dde2_jones,
                         die1_jones, base_vis, die2_jones)
    have_ddes1, have_coh, have_ddes2, have_dies1, have_bvis, have_dies2 = tup
    arrays = (dde1_jones, source_coh, dde2_jones, die1_jones, base_vis, die2_jones)
    present = [a for a in arrays if a is not None]
    if not present:
        raise ValueError("No Jones Matrices were supplied")

    jones_types = [
        _jones_type("dde1_jones", dde1_jones, 5, 6), _jones_type("source_coh", source_coh, 4, 5),
        _jones_type("dde2_jones", dde2_jones, 5, 6), _jones_type("die1_jones", die1_jones, 4, 5),
        _jones_type("base_vis", base_vis, 3, 4), _jones_type("die2_jones", die2_jones, 4, 5)]
    ptypes = [t for t in jones_types if t != JONES_NOT_PRESENT]
    if not all(ptypes[0] == p for p in ptypes[1:]):
        raise ValueError("Jones Matrix Correlations were mismatched")
    jones_type = ptypes[0]

    out_dtype = np.result_type(*[np_dtype_of(a) for a in present])
    if out_dtype in (np.dtype(np.complex64), np.dtype(np.float32)):
        ct, fn = np.complex64, "af_predict_vis_c64"
    else:
        ct, fn = np.complex128, "af_predict_vis_c128"

    have_ddes = have_ddes1 and have_ddes2
    have_dies = have_dies1 and have_dies2
    nrow = int(time_index.shape[0])
    nsrc = ntime = nant = 0
    # output shape from the first present of dde / coh / die / base_vis (predict.py:255-326)
    if have_ddes:
        nsrc, ntime, nant, nchan = (int(s) for s in dde1_jones.shape[:4])
        corrs = tuple(int(s) for s in dde1_jones.shape[4:])
    elif have_coh:
        nsrc, nchan = int(source_coh.shape[0]), int(source_coh.shape[2])
        corrs = tuple(int(s) for s in source_coh.shape[3:])
    elif have_dies:
        ntime, nant, nchan = (int(s) for s in die1_jones.shape[:3])
        corrs = tuple(int(s) for s in die1_jones.shape[3:])
    else:
        nchan = int(base_vis.shape[1])
        corrs = tuple(int(s) for s in base_vis.shape[2:])
    if have_dies:
        ntime, nant = int(die1_jones.shape[0]), int(die1_jones.shape[1])
    if corrs not in ((1,), (2,), (2, 2)):
        raise ValueError("correlation shape %s not in ((1,), (2,), (2, 2))" % (corrs,))
    ncorr = int(np.prod(corrs))

    # extents the kernel relies on (the reference reads out of bounds here; we refuse)
    def _expect(name, arr, shape):
        if arr is not None and tuple(int(s) for s in arr.shape) != shape:
            raise ValueError("%s has shape %s, expected %s" % (name, tuple(arr.shape), shape))

    if tuple(antenna1.shape) != (nrow,) or tuple(antenna2.shape) != (nrow,):
        raise ValueError("time_index, antenna1 and antenna2 must have the same length")
    _expect("dde1_jones", dde1_jones, (nsrc, ntime, nant, nchan) + corrs)
    _expect("dde2_jones", dde2_jones, (nsrc, ntime, nant, nchan) + corrs)
    _expect("source_coh", source_coh, (nsrc, nrow, nchan) + corrs)
    _expect("die1_jones", die1_jones, (ntime, nant, nchan) + corrs)
    _expect("die2_jones", die2_jones, (ntime, nant, nchan) + corrs)
    _expect("base_vis", base_vis, (nrow, nchan) + corrs)
    _check_indices(time_index, antenna1, antenna2, ntime, nant, have_ddes or have_dies)

    with Call(time_index, antenna1, antenna2, *arrays) as c:
        p_ti, ib = _index_array(c, time_index)
        if ib == 4 and not (np_dtype_of(antenna1) == np.int32 and np_dtype_of(antenna2) == np.int32):
            p_ti, ib = c.inp(time_index, np.int64), 8
        idt = np.int32 if ib == 4 else np.int64
        p_a1, p_a2 = c.inp(antenna1, idt), c.inp(antenna2, idt)
        ptrs = [c.inp(a, ct) for a in arrays]
        p_out, h = c.out((nrow, nchan) + corrs, ct)
        ws_bytes = int(_lib.load().af_predict_vis_workspace_bytes())
        p_ws = c.scratch(ws_bytes)
        _lib.call(fn, p_ti, p_a1, p_a2, ib, nrow, *ptrs, nsrc, ntime, nant, nchan, ncorr,
                  _lib.AF_JONES_2X2 if jones_type == JONES_2X2 else _lib.AF_JONES_DIAG,
                  p_out, p_ws, ws_bytes, c.stream)
        if have_ddes or have_dies:
            c.watch_status(_lib.AF_PREDICT_VIS_STATUS_OFFSET, lambda flags: _index_status_message(flags, ntime, nant))
        return c.result(h)


def apply_gains(time_index, antenna1, antenna2, die1_jones, corrupted_vis, die2_jones):
    """``africanus.rime.predict.apply_gains`` (africanus/rime/predict.py:622-649):
    ``predict_vis`` with only the direction-independent terms and ``base_vis``."""
    return predict_vis(time_index, antenna1, antenna2, die1_jones=die1_jones,
                       base_vis=corrupted_vis, die2_jones=die2_jones)
