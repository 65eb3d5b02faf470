"""
The calling convention of ``dask.array.blockwise`` without dask: what arguments a block function receives.

The reference's dask layer (africanus/rime/dask_predict.py:257-308,311-439, africanus/dft/dask.py:20-51,
africanus/rime/dask.py:34-52) is ``da.blockwise`` over per-block wrappers; the build's own wrappers
(codex_africanus_amd/rime/dask.py, dft/dask.py) keep that contract.  dask is not installed next to the GPU, so
the GPU tests drive the block wrappers through this emulator, which reproduces exactly the part of blockwise
that the wrappers depend on:

  * the output has one block per combination of block indices of the OUTPUT index names;
  * an input axis whose name appears in the output index contributes its block of that number (blocks are
    matched by POSITION, sizes may differ: ``align_arrays=False`` -- row chunks against time chunks);
  * an input axis whose name does NOT appear in the output index is "contracted": the function receives a
    list over all blocks of that axis -- one list level per contracted axis, in the order of the input's own
    index -- even when there is a single block (``lm[0]``, ``dde1_jones[0]`` in the reference's wrappers);
  * an argument paired with index ``None`` is passed through as is; keyword arguments go to every call;
  * ``new_axes`` / ``adjust_chunks`` only affect the declared chunk sizes of the result, which here are
    whatever the blocks turn out to be; the result blocks are concatenated along every output axis.

tests/golden/make_golden_dask.py checks this emulator against the real ``da.blockwise`` (dask 2021.10 under the
build container's conda interpreter) on the same wrappers with the CPU oracle as the block function.
"""
import itertools

import numpy as np


def _bounds(chunks):
    edges = np.concatenate([[0], np.cumsum(chunks)]).astype(np.int64)
    return [(int(edges[i]), int(edges[i + 1])) for i in range(len(chunks))]


class Chunked(object):
    """An array with dask-style chunks: a tuple of block lengths per axis."""

    def __init__(self, array, chunks):
        self.array = array
        self.chunks = tuple(tuple(int(c) for c in ch) if not isinstance(ch, (int, np.integer)) else (int(ch),)
                            for ch in chunks)
        assert len(self.chunks) == array.ndim, (self.chunks, array.shape)
        for ax, ch in enumerate(self.chunks):
            assert sum(ch) == array.shape[ax], (ax, ch, array.shape)
        self.bounds = [_bounds(ch) for ch in self.chunks]

    @property
    def numblocks(self):
        return tuple(len(c) for c in self.chunks)

    def block(self, idx):
        sl = tuple(slice(*self.bounds[ax][i]) for ax, i in enumerate(idx))
        return self.array[sl]


def blockwise(func, out_ind, *args, executor=None, **kwargs):
    """``blockwise(func, out_ind, a, a_ind, b, b_ind, ..., **kwargs)`` -> numpy array (blocks concatenated).
    ``a`` is a :class:`Chunked`, or anything at all when ``a_ind`` is None.  ``executor`` (a
    concurrent.futures executor) runs the block calls concurrently, as dask's threaded scheduler does."""
    pairs = list(zip(args[0::2], args[1::2]))
    nblocks = {}
    for a, ind in pairs:
        if ind is None:
            continue
        for name, n in zip(ind, a.numblocks):
            if nblocks.setdefault(name, n) != n:
                raise ValueError("axis %r has %d blocks in one argument and %d in another" % (name, nblocks[name], n))
    out_counts = [nblocks.get(name, 1) for name in out_ind]

    def arg_for(a, ind, coord):
        if ind is None:
            return a
        dummies = [k for k, name in enumerate(ind) if name not in out_ind]
        fixed = [coord[out_ind.index(name)] if name in out_ind else None for name in ind]

        def build(level, idx):
            if level == len(dummies):
                return a.block(tuple(idx))
            ax = dummies[level]
            out = []
            for b in range(a.numblocks[ax]):
                idx2 = list(idx)
                idx2[ax] = b
                out.append(build(level + 1, idx2))
            return out
        return build(0, fixed)

    coords = list(itertools.product(*[range(n) for n in out_counts]))

    def run(coord):
        return func(*[arg_for(a, ind, coord) for a, ind in pairs], **kwargs)

    results = list(executor.map(run, coords)) if executor is not None else [run(c) for c in coords]
    grid = np.empty(out_counts, dtype=object)
    for coord, r in zip(coords, results):
        grid[coord] = np.asarray(r)

    def assemble(sub, axis):
        if axis == len(out_counts):
            return sub
        return np.concatenate([assemble(sub[i], axis + 1) for i in range(out_counts[axis])], axis=axis)
    return assemble(grid, 0)
