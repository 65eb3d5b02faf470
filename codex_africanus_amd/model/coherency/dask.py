"""dask.array front-end of convert (africanus/model/coherency/dask.py:27-66): blocks over the leading dimensions,
the correlation dimensions in one chunk each."""
try:
    import dask.array as da
    _dask_error = None
except ImportError as e:  # pragma: no cover
    da = None
    _dask_error = e

from .conversion import convert_setup, convert_impl


def _block(block, mapping=None, in_shape=None, out_shape=None, dtype_=None):
    return convert_impl(block, mapping, in_shape, out_shape, dtype_)


def convert(input, input_schema, output_schema, implicit_stokes=False):
    if da is None:
        raise ImportError("dask.array is required for codex_africanus_amd.model.coherency.dask: %s" % (_dask_error,))
    mapping, in_shape, out_shape, dtype = convert_setup(input, input_schema, output_schema, implicit_stokes)
    nfree = input.ndim - len(in_shape)
    if any(len(c) != 1 for c in input.chunks[nfree:]):
        input = input.rechunk(input.chunks[:nfree] + tuple((s,) for s in in_shape))
    return da.map_blocks(_block, input, mapping=mapping, in_shape=in_shape, out_shape=out_shape, dtype_=dtype,
                         dtype=dtype, drop_axis=list(range(nfree, input.ndim)),
                         new_axis=list(range(nfree, nfree + len(out_shape))),
                         chunks=input.chunks[:nfree] + tuple((s,) for s in out_shape))
