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
    nchan float64 values: latency bound).  No-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(chi2, op=dist.ReduceOp.SUM, group=group)
    return chi2


def allreduce_image(image, group=None):
    """Sum a (source, chan, corr) dirty image over all ranks, in place: the cross-GPU step of the
    row-sharded ``vis_to_im`` (the reference sums row-chunk images, africanus/dft/dask.py:90).
    No-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
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


def predict_shard(rank, world_size, image, uvw, lm, frequency, data=None, convention="fourier", group=None):
    """One rank's part of the row-sharded direct-transform predict: im_to_vis on its row block
    (device resident) and, if ``data`` (the rank's rows) is given, the all-reduced chi-squared.
    Returns (vis_shard, chi2 or None, (start, stop))."""
    from .dft.kernels import im_to_vis
    start, stop = shard_bounds(uvw.shape[0], world_size)[rank]
    vis = im_to_vis(image, uvw[start:stop], lm, frequency, convention=convention)
    c2 = None
    if data is not None:
        c2 = allreduce_chi2(chi2(vis, data), group=group)
    return vis, c2, (start, stop)
