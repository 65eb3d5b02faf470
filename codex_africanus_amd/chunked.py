"""
Chunked execution of the hot path with the block -> kernel-call contract of the reference's
dask wrappers, without needing dask:

    africanus/rime/dask_predict.py:443-593  predict_vis (chunk checks :478-524,
        parallel_reduction :311-369, linear_reduction :181-254, apply_dies :372-439)
    africanus/rime/dask.py:38-52            phase_delay
    africanus/dft/dask.py:26-51,60-90       im_to_vis, vis_to_im

Arrays are plain numpy arrays (or torch ROCm tensors); chunking is described like
``dask.array.Array.chunks``: a tuple of block lengths per axis.  Every block is one call of the
array-level function (one HIP launch sequence), exactly what a dask task would do, so results for a
given chunking equal dask's.  The rules the reference enforces are enforced here with the same
``ValueError``s: the antenna axis is never chunked; the number of row chunks equals the number of
time chunks and a row chunk's ``time_index`` range maps to its own time chunk (every block
normalises ``time_index`` by its own minimum, africanus/rime/predict.py:597); im_to_vis keeps the
source axis in one chunk.

``streams=True`` reproduces the serial source-chunk chain (each chunk's result is fed as
``base_vis`` of the next); otherwise per-source-chunk results are summed.
"""
import numpy as np

from .rime.predict import predict_vis as _predict_vis, predict_checks
from .rime.phase import phase_delay as _phase_delay
from .dft.kernels import im_to_vis as _im_to_vis, vis_to_im as _vis_to_im


def normalize_chunks(chunks, size, name="axis"):
    """int -> uniform blocks; tuple -> validated; None -> one block."""
    if chunks is None:
        return (int(size),)
    if isinstance(chunks, (int, np.integer)):
        c = int(chunks)
        if c <= 0:
            raise ValueError("%s chunk size must be positive" % name)
        full, rem = divmod(int(size), c)
        return (c,) * full + ((rem,) if rem else ()) or (0,)
    chunks = tuple(int(c) for c in chunks)
    if sum(chunks) != int(size):
        raise ValueError("%s chunks %s do not sum to %d" % (name, chunks, size))
    return chunks


def _bounds(chunks):
    edges = np.concatenate([[0], np.cumsum(chunks)]).astype(np.int64)
    return [(int(edges[i]), int(edges[i + 1])) for i in range(len(chunks))]


def _cat(blocks, axis):
    if type(blocks[0]).__module__.split(".")[0] == "torch":
        import torch
        return torch.cat(blocks, dim=axis)
    return np.concatenate(blocks, axis=axis)


def phase_delay(lm, uvw, frequency, convention="fourier", chunks=None):
    """Blockwise phase_delay over (source, row, chan) chunks (africanus/rime/dask.py:38-52)."""
    chunks = chunks or {}
    sb = _bounds(normalize_chunks(chunks.get("source"), lm.shape[0], "source"))
    rb = _bounds(normalize_chunks(chunks.get("row"), uvw.shape[0], "row"))
    cb = _bounds(normalize_chunks(chunks.get("chan"), frequency.shape[0], "chan"))
    return _cat([_cat([_cat([_phase_delay(lm[s0:s1], uvw[r0:r1], frequency[c0:c1], convention=convention)
                             for (c0, c1) in cb], 2) for (r0, r1) in rb], 1) for (s0, s1) in sb], 0)


def im_to_vis(image, uvw, lm, frequency, convention="fourier", dtype=np.complex128, chunks=None):
    """Blockwise im_to_vis over (row, chan) chunks (africanus/dft/dask.py:26-51).  The source
    axis must stay in one chunk, as in the reference (:29-36)."""
    chunks = chunks or {}
    src = normalize_chunks(chunks.get("source"), lm.shape[0], "source")
    if src[0] != lm.shape[0]:
        raise ValueError("lm chunks must match lm shape on first axis")
    if src[0] != image.shape[0]:
        raise ValueError("Image chunks must match image shape on first axis")
    rb = _bounds(normalize_chunks(chunks.get("row"), uvw.shape[0], "row"))
    cb = _bounds(normalize_chunks(chunks.get("chan"), frequency.shape[0], "chan"))
    return _cat([_cat([_im_to_vis(image[:, c0:c1], uvw[r0:r1], lm, frequency[c0:c1],
                                  convention=convention, dtype=dtype)
                       for (c0, c1) in cb], 1) for (r0, r1) in rb], 0)


def vis_to_im(vis, uvw, lm, frequency, flags, convention="fourier", dtype=np.float64, chunks=None):
    """Blockwise vis_to_im over (row, chan) chunks, summed over the row chunks
    (africanus/dft/dask.py:60-90: ``ims.sum(axis=0)``).  The source axis stays whole."""
    chunks = chunks or {}
    rb = _bounds(normalize_chunks(chunks.get("row"), uvw.shape[0], "row"))
    cb = _bounds(normalize_chunks(chunks.get("chan"), frequency.shape[0], "chan"))
    chan_blocks = []
    for (c0, c1) in cb:
        acc = None
        for (r0, r1) in rb:
            part = _vis_to_im(vis[r0:r1, c0:c1], uvw[r0:r1], lm, frequency[c0:c1], flags[r0:r1, c0:c1],
                              convention=convention, dtype=dtype)
            acc = part if acc is None else acc + part
        chan_blocks.append(acc)
    return _cat(chan_blocks, 1)


def wsclean_predict(uvw, lm, source_type, flux, coeffs, log_poly, ref_freq, gauss_shape, frequency, chunks=None):
    """Blockwise wsclean_predict over (source, row, chan) chunks, summed over the source chunks
    (africanus/rime/dask_predict.py:609-658: ``vis.sum(axis=0)``)."""
    from .rime.wsclean_predict import wsclean_predict as _wsclean_predict
    chunks = chunks or {}
    nsrc = lm.shape[0]
    sb = _bounds(normalize_chunks(chunks.get("source"), nsrc, "source")) if nsrc else [(0, 0)]
    rb = _bounds(normalize_chunks(chunks.get("row"), uvw.shape[0], "row"))
    cb = _bounds(normalize_chunks(chunks.get("chan"), frequency.shape[0], "chan"))
    lp = np.broadcast_to(np.asarray(log_poly), (nsrc,))
    row_blocks = []
    for (r0, r1) in rb:
        chan_blocks = []
        for (c0, c1) in cb:
            acc = None
            for (s0, s1) in sb:
                part = _wsclean_predict(uvw[r0:r1], lm[s0:s1], source_type[s0:s1], flux[s0:s1], coeffs[s0:s1],
                                        lp[s0:s1], ref_freq[s0:s1], gauss_shape[s0:s1], frequency[c0:c1])
                acc = part if acc is None else acc + part
            chan_blocks.append(acc)
        row_blocks.append(_cat(chan_blocks, 1))
    return _cat(row_blocks, 0)


def predict_vis(time_index, antenna1, antenna2, dde1_jones=None, source_coh=None, dde2_jones=None,
                die1_jones=None, base_vis=None, die2_jones=None, streams=None, chunks=None):
    """
    Chunked predict_vis with the contract of ``africanus.rime.dask.predict_vis``
    (africanus/rime/dask_predict.py:443-593).

    ``chunks``: dict with any of "source", "row", "time", "chan" (block lengths or a block
    size).  Row and time chunk COUNTS must match (:494-499,:519-524).
    """
    tup = predict_checks(time_index, antenna1, antenna2, dde1_jones, source_coh, dde2_jones,
                         die1_jones, base_vis, die2_jones)
    have_ddes1, have_coh, have_ddes2, have_dies1, have_bvis, have_dies2 = tup
    have_ddes, have_dies = have_ddes1 and have_ddes2, have_dies1 and have_dies2
    chunks = chunks or {}
    nrow = int(time_index.shape[0])

    if have_ddes:
        nsrc, ntime, nchan = dde1_jones.shape[0], dde1_jones.shape[1], dde1_jones.shape[3]
    elif have_coh:
        nsrc, ntime, nchan = source_coh.shape[0], None, source_coh.shape[2]
    elif have_dies:
        nsrc, ntime, nchan = 0, die1_jones.shape[0], die1_jones.shape[2]
    elif have_bvis:
        nsrc, ntime, nchan = 0, None, base_vis.shape[1]
    else:
        raise ValueError("No Jones Matrices were supplied")
    if have_dies:
        ntime = die1_jones.shape[0]

    row_chunks = normalize_chunks(chunks.get("row"), nrow, "row")
    chan_chunks = normalize_chunks(chunks.get("chan"), nchan, "chan")
    src_chunks = normalize_chunks(chunks.get("source"), nsrc, "source") if nsrc else (0,)
    if ntime is not None:
        time_chunks = normalize_chunks(chunks.get("time"), ntime, "time")
        if len(time_chunks) != len(row_chunks):
            raise ValueError("Number of row chunks (%s) does not equal number of time chunks (%s)."
                             % (row_chunks, time_chunks))
        tb = _bounds(time_chunks)
    else:
        tb = [(0, 0)] * len(row_chunks)
    if "ant" in chunks or "antenna" in chunks:
        na = (dde1_jones.shape[2] if have_ddes else die1_jones.shape[1] if have_dies else None)
        ac = chunks.get("ant", chunks.get("antenna"))
        if na is not None and normalize_chunks(ac, na, "ant") != (na,):
            raise ValueError("Subdivision of antenna dimension into multiple chunks is not supported.")

    rb, cb, sb = _bounds(row_chunks), _bounds(chan_chunks), _bounds(src_chunks)
    row_blocks = []
    for (r0, r1), (t0, t1) in zip(rb, tb):
        ti, a1, a2 = time_index[r0:r1], antenna1[r0:r1], antenna2[r0:r1]
        chan_blocks = []
        for (c0, c1) in cb:
            acc = None
            if have_ddes or have_coh:
                partial = []
                for (s0, s1) in sb:
                    d1 = dde1_jones[s0:s1, t0:t1, :, c0:c1] if have_ddes else None
                    d2 = dde2_jones[s0:s1, t0:t1, :, c0:c1] if have_ddes else None
                    co = source_coh[s0:s1, r0:r1, c0:c1] if have_coh else None
                    if streams is True:
                        # serial chain: the running sum rides in as base_vis (dask_predict.py:121-156)
                        acc = _predict_vis(ti, a1, a2, d1, co, d2, None, acc, None)
                    else:
                        partial.append(_predict_vis(ti, a1, a2, d1, co, d2, None, None, None))
                if streams is not True:
                    acc = partial[0]
                    for p in partial[1:]:
                        acc = acc + p
            if have_dies or have_bvis:
                bv = base_vis[r0:r1, c0:c1] if have_bvis else None
                if acc is not None:
                    bv = acc if bv is None else bv + acc
                g1 = die1_jones[t0:t1, :, c0:c1] if have_dies else None
                g2 = die2_jones[t0:t1, :, c0:c1] if have_dies else None
                acc = _predict_vis(ti, a1, a2, None, None, None, g1, bv, g2)
            chan_blocks.append(acc)
        row_blocks.append(_cat(chan_blocks, 1))
    return _cat(row_blocks, 0)


def fused_predict_vis(time_index, antenna1, antenna2, lm, uvw, frequency, brightness=None,
                      beam=None, beam_lm_extents=None, beam_freq_map=None, parallactic_angles=None,
                      point_errors=None, antenna_scaling=None,
                      die1_jones=None, base_vis=None, die2_jones=None, convention="fourier",
                      feed_rotation=None, gauss_shape=None, stokes=None, spi=None, ref_freq=None,
                      corr_schema=(("XX", "XY"), ("YX", "YY")), spectral_base=0, streams=None, chunks=None):
    """
    Chunked fused predict from source-level inputs, block for block what ``rime.dask.fused_predict_vis`` computes (the
    reference graph of africanus/rime/examples/predict.py:404-525 with the chunk rules of
    africanus/rime/dask_predict.py:478-524) on numpy arrays or torch ROCm tensors.

    ``chunks``: dict with any of "source", "row", "time", "chan".  Row chunk k is paired with time chunk k of
    ``parallactic_angles`` / ``point_errors`` / ``feed_rotation`` / ``die{1,2}_jones`` (counts must agree; every block
    normalises ``time_index`` by its own minimum, africanus/rime/predict.py:597); the antenna axis and the beam cube
    are never chunked.  Source chunks are summed (serial chain when ``streams=True``); ``base_vis`` and the DIEs are
    applied to that sum.  The row layout's plan is made once per row chunk and re-used for its source and channel blocks.
    """
    from .rime.fused import fused_predict_vis as _fused, cached_plan, _all_single
    # every input single precision: plans of float32 rows decompose at their own precision (the single-precision GEMM form)
    single = _all_single(lm, uvw, frequency, brightness, feed_rotation, beam, beam_lm_extents, beam_freq_map,
                         parallactic_angles, point_errors, antenna_scaling, stokes, spi, ref_freq)
    chunks = chunks or {}
    if (die1_jones is None) != (die2_jones is None):
        raise ValueError("Both die1_jones and die2_jones must be present or absent")
    nrow, nsrc, nchan = int(uvw.shape[0]), int(lm.shape[0]), int(frequency.shape[0])
    per_time = [a for a in (parallactic_angles, point_errors, feed_rotation, die1_jones, die2_jones) if a is not None]
    ntime = int(per_time[0].shape[0]) if per_time else None
    row_chunks = normalize_chunks(chunks.get("row"), nrow, "row")
    chan_chunks = normalize_chunks(chunks.get("chan"), nchan, "chan")
    src_chunks = normalize_chunks(chunks.get("source"), nsrc, "source")
    if ntime is not None:
        time_chunks = normalize_chunks(chunks.get("time"), ntime, "time")
        if len(time_chunks) != len(row_chunks):
            raise ValueError("Number of row chunks (%s) does not equal number of time chunks (%s)."
                             % (row_chunks, time_chunks))
        tb = _bounds(time_chunks)
    else:
        tb = [(0, 0)] * len(row_chunks)
    if "ant" in chunks or "antenna" in chunks:
        na = int(per_time[0].shape[1]) if per_time else None
        if na is not None and normalize_chunks(chunks.get("ant", chunks.get("antenna")), na, "ant") != (na,):
            raise ValueError("Subdivision of antenna dimension into multiple chunks is not supported.")
    flat = brightness is not None and len(brightness.shape) == 3
    cut = lambda a, *sl: None if a is None else a[sl]
    row_blocks = []
    for (r0, r1), (t0, t1) in zip(_bounds(row_chunks), tb):
        ti, a1, a2, uv = time_index[r0:r1], antenna1[r0:r1], antenna2[r0:r1], uvw[r0:r1]
        plan = None
        if beam is not None:
            plan = cached_plan(ti, a1, a2, int(parallactic_angles.shape[1]), uvw=None if gauss_shape is not None else uv,
                               single=single)
        chan_blocks = []
        for (c0, c1) in _bounds(chan_chunks):
            acc = None
            for (s0, s1) in _bounds(src_chunks):
                b = None if brightness is None else (brightness[s0:s1] if flat else brightness[s0:s1, c0:c1])
                part = _fused(ti, a1, a2, lm[s0:s1], uv, frequency[c0:c1], b, beam, beam_lm_extents, beam_freq_map,
                              cut(parallactic_angles, slice(t0, t1)),
                              cut(point_errors, slice(t0, t1), slice(None), slice(c0, c1)),
                              cut(antenna_scaling, slice(None), slice(c0, c1)), None, None, None, convention,
                              cut(feed_rotation, slice(t0, t1)), cut(gauss_shape, slice(s0, s1)),
                              cut(stokes, slice(s0, s1)), cut(spi, slice(s0, s1)), cut(ref_freq, slice(s0, s1)),
                              corr_schema, spectral_base, plan)
                acc = part if acc is None else acc + part
            if die1_jones is not None or base_vis is not None:
                bv = acc if base_vis is None else base_vis[r0:r1, c0:c1] + acc
                acc = _predict_vis(ti, a1, a2, None, None, None, cut(die1_jones, slice(t0, t1), slice(None), slice(c0, c1)),
                                   bv, cut(die2_jones, slice(t0, t1), slice(None), slice(c0, c1)))
            chan_blocks.append(acc)
        row_blocks.append(_cat(chan_blocks, 1))
    return _cat(row_blocks, 0)
