"""
Row sharding of the predict across the GPUs of one node.

Rows are independent, so each rank (one process per GPU, ``torch.distributed`` with the ``nccl``
backend = RCCL over xGMI on ROCm, ``gloo`` in CPU tests) predicts a contiguous block of rows that
starts and ends on timestep boundaries -- the reference's rule that a unique time never straddles
row chunks (africanus/rime/dask_predict.py:694-712, africanus/util/shapes.py:4-69).  Visibilities
never cross GPUs; the only exchange is one all-reduce of the per-channel chi-squared vector.
"""
import numpy as np


def shard_bounds(nrow, world_size, time_index=None):
    """Row range [start, stop) of every rank: near-equal contiguous blocks; with ``time_index``
    (non-decreasing) block edges are moved to the nearest timestep boundary so that no timestep
    straddles two shards.  Returns a list of (start, stop) of length ``world_size``."""
    nrow, world_size = int(nrow), int(world_size)
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    edges = [(nrow * r) // world_size for r in range(world_size + 1)]
    if time_index is not None and nrow > 0:
        ti = np.asarray(time_index)
        if ti.shape[0] != nrow:
            raise ValueError("time_index length does not match nrow")
        if np.any(np.diff(ti) < 0):
            raise ValueError("time_index must be non-decreasing to shard on timestep boundaries")
        starts = np.concatenate([[0], np.nonzero(np.diff(ti))[0] + 1, [nrow]])  # timestep starts + end
        snapped = [0]
        for e in edges[1:-1]:
            k = int(np.searchsorted(starts, e))
            lo, hi = starts[max(k - 1, 0)], starts[min(k, len(starts) - 1)]
            cand = lo if (e - lo) <= (hi - e) else hi
            snapped.append(int(max(cand, snapped[-1])))
        edges = snapped + [nrow]
    return [(int(edges[r]), int(edges[r + 1])) for r in range(world_size)]


def time_slice(time_index, start, stop):
    """[t0, t1) range of (global) time indices touched by rows [start, stop): the slice of the
    (time, ant, ...) DDE/DIE arrays this shard needs."""
    if stop <= start:
        return (0, 0)
    ti = np.asarray(time_index[start:stop])
    return (int(ti.min()), int(ti.max()) + 1)


def allreduce_chi2(chi2, group=None):
    """Sum the per-channel chi-squared vector over all ranks, in place (RCCL/gloo all-reduce of
    nchan float64 values: latency bound).  No-op without an initialised process group; with one -- also a group
    of ONE rank -- the collective runs (a world-size-1 ``nccl`` group still builds an RCCL communicator and launches
    the reduction on the device, which is how the one-GPU test box exercises the path: tests/test_gpu_rccl.py)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(chi2, op=dist.ReduceOp.SUM, group=group)
    return chi2


def allreduce_image(image, group=None):
    """Sum a (source, chan, corr) dirty image over all ranks, in place: the cross-GPU step of the
    row-sharded ``vis_to_im`` (the reference sums row-chunk images, africanus/dft/dask.py:90).
    No-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(image, op=dist.ReduceOp.SUM, group=group)
    return image


def vis_to_im_shard(rank, world_size, vis, uvw, lm, frequency, flags, convention="fourier", group=None):
    """One rank's part of the row-sharded adjoint transform: ``vis_to_im`` of its row block
    (torch ROCm tensors), then the all-reduced image.  ``vis``/``uvw``/``flags`` hold the FULL row
    range on every rank only in tests; in production each rank passes its own rows with
    ``world_size=1``-style slicing done by the caller."""
    from .dft.kernels import vis_to_im
    start, stop = shard_bounds(uvw.shape[0], world_size)[rank]
    im = vis_to_im(vis[start:stop], uvw[start:stop], lm, frequency, flags[start:stop], convention=convention)
    return allreduce_image(im, group=group), (start, stop)


def chi2(model, data, weight=None):
    """chi2[nu] = sum_{r,c} w |data - model|^2 on the device (af_chi2_c128); torch ROCm tensors
    (row, chan, corr) complex128 in, (chan,) float64 tensor out."""
    import ctypes
    import torch
    from . import _lib
    if not (model.is_cuda and data.is_cuda):
        raise ValueError("chi2 expects ROCm tensors")
    if tuple(model.shape) != tuple(data.shape) or model.dim() < 2:
        raise ValueError("chi2: model %s and data %s must share one (row, chan, ...) shape"
                         % (tuple(model.shape), tuple(data.shape)))
    # af_chi2_c128 reads 16-byte complex values: complex64 operands (im_to_vis of all-float32 inputs) are
    # widened here, anything that is not complex is refused rather than reinterpreted
    for name, x in (("model", model), ("data", data)):
        if x.dtype not in (torch.complex64, torch.complex128):
            raise ValueError("chi2: %s must be complex64 or complex128, got %s" % (name, x.dtype))
    model = model.to(torch.complex128).contiguous()
    data = data.to(torch.complex128).contiguous()
    if weight is not None:
        if weight.is_complex() or tuple(weight.shape) != tuple(model.shape):
            raise ValueError("chi2: weight must be real with the shape of the visibilities %s, got %s %s"
                             % (tuple(model.shape), weight.dtype, tuple(weight.shape)))
        if weight.device != model.device:
            raise ValueError("chi2: weight lives on %s, the visibilities on %s" % (weight.device, model.device))
    nrow, nchan = int(model.shape[0]), int(model.shape[1])
    ncorr = int(np.prod(model.shape[2:])) if model.dim() > 2 else 1
    out = torch.empty(nchan, dtype=torch.float64, device=model.device)
    w = None if weight is None else weight.to(torch.float64).contiguous()
    with torch.cuda.device(model.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(model.device).cuda_stream)
        _lib.call("af_chi2_c128", ctypes.c_void_p(model.data_ptr()), ctypes.c_void_p(data.data_ptr()),
                  None if w is None else ctypes.c_void_p(w.data_ptr()), nrow, nchan, ncorr,
                  ctypes.c_void_p(out.data_ptr()), stream)
    return out


def predict_shard(rank, world_size, image, uvw, lm, frequency, data=None, convention="fourier", group=None,
                  time_index=None):
    """One rank's part of the row-sharded direct-transform predict: im_to_vis on its row block
    (device resident) and, if ``data`` (the rank's rows) is given, the all-reduced chi-squared.
    ``time_index`` (row,), non-decreasing, optional: the block then starts and ends on timestep boundaries
    (``shard_bounds(nrow, world_size, time_index)``: the reference's rule for row chunks that meet (time, ant, ...) arrays,
    africanus/rime/dask_predict.py:494-499), so that the same bounds serve the DIE / DDE stages of the job.
    Returns (vis_shard, chi2 or None, (start, stop))."""
    from .dft.kernels import im_to_vis, im_to_vis_chi2
    if time_index is not None:
        from .rime.fused import _host
        time_index = np.asarray(_host(time_index))
    start, stop = shard_bounds(uvw.shape[0], world_size, time_index)[rank]
    if data is None:
        return im_to_vis(image, uvw[start:stop], lm, frequency, convention=convention), None, (start, stop)
    # transform and chi^2 in one device call (summed in the transform's epilogue where the MFMA kernels run)
    vis, c2 = im_to_vis_chi2(image, uvw[start:stop], lm, frequency, data, convention=convention)
    return vis, allreduce_chi2(c2, group=group), (start, stop)


def fused_predict_shard(rank, world_size, time_index, antenna1, antenna2, lm, uvw, frequency, brightness=None,
                        beam=None, beam_lm_extents=None, beam_freq_map=None, parallactic_angles=None,
                        point_errors=None, antenna_scaling=None, die1_jones=None, base_vis=None, die2_jones=None,
                        data=None, weight=None, group=None, bounds=None, **kwargs):
    """
    One rank's part of the row-sharded FUSED predict (BASELINE configs[3]: 8e6 rows x 64 chan x 1000 sources over 8
    GPUs, the only predict that exists at that size): the rank's rows are a contiguous block that starts and ends on
    timestep boundaries (``shard_bounds(nrow, world_size, time_index)``: the reference's rule that a time never
    straddles two row chunks, africanus/rime/dask_predict.py:494-499), its slice of every (time, ant, ...) array is
    ``time_slice`` of those rows, sources / frequency / beam cube are replicated.  One ``fused_predict_vis`` call on
    the rank's device, then -- if ``data`` (the observed visibilities of the SAME rows as passed in) is given -- the
    per-channel chi-squared of the shard, all-reduced over the ranks (RCCL over xGMI / gloo): the path's only collective.

    Arrays are FULL-length (every rank sees all rows; the rank's block is sliced here) unless ``bounds=(start, stop)``
    is given: then ``time_index`` / ``antenna1`` / ``antenna2`` / ``uvw`` / ``base_vis`` / ``data`` / ``weight`` hold
    only this rank's rows -- rows [start, stop) of the job -- and the per-time arrays only this rank's timesteps, which
    is how a job that cannot hold 8e6 rows in one place feeds its ranks.  Further keyword arguments go to
    ``fused_predict_vis`` (``feed_rotation`` -- sliced like the other per-time arrays -- ``gauss_shape``, ``stokes``,
    ``spi``, ``ref_freq``, ``corr_schema``, ``convention``...).  Returns (vis_shard, chi2 or None, (start, stop)).
    """
    from .rime.fused import fused_predict_vis, cached_plan, _host
    feed_rotation = kwargs.pop("feed_rotation", None)
    if bounds is None:
        ti_h = np.asarray(_host(time_index))
        start, stop = shard_bounds(ti_h.shape[0], world_size, ti_h)[rank]
        t0, t1 = time_slice(ti_h, start, stop)
        tmin = int(ti_h.min()) if ti_h.size else 0
        t0, t1 = t0 - tmin, t1 - tmin                 # time_index may carry an offset; the per-time arrays start at 0
        rows, times = slice(start, stop), slice(t0, t1)
    else:
        start, stop = (int(b) for b in bounds)
        rows, times = slice(None), slice(None)
    cut = lambda a, *sl: None if a is None else a[sl]
    ti, a1, a2 = time_index[rows], antenna1[rows], antenna2[rows]
    plan = None
    if beam is not None and stop > start:
        from .rime.fused import _all_single
        single = _all_single(lm, uvw, frequency, brightness, feed_rotation, beam, beam_lm_extents, beam_freq_map, parallactic_angles,
                             point_errors, antenna_scaling, kwargs.get("stokes"), kwargs.get("spi"), kwargs.get("ref_freq"))
        plan = cached_plan(ti, a1, a2, int(parallactic_angles.shape[1]),
                           uvw=None if kwargs.get("gauss_shape") is not None else uvw[rows], single=single)
    vis = fused_predict_vis(ti, a1, a2, lm, uvw[rows], frequency, brightness, beam, beam_lm_extents, beam_freq_map,
                            cut(parallactic_angles, times), cut(point_errors, times), antenna_scaling,
                            cut(die1_jones, times), cut(base_vis, rows), cut(die2_jones, times),
                            feed_rotation=cut(feed_rotation, times), plan=plan, **kwargs)
    c2 = None
    if data is not None:
        c2 = allreduce_chi2(chi2(vis, data[rows], cut(weight, rows)), group=group)
    return vis, c2, (start, stop)
