"""Stokes <-> correlation conversion with the signature of africanus/model/coherency/conversion.py:207."""
import ctypes
from pprint import pformat

import numpy as np

from ... import _lib
from ..._device import Call, np_dtype_of

# casacore Stokes enumeration (africanus/util/casa_types.py:4-38)
STOKES_TYPES = ("Undefined I Q U V RR RL LR LL XX XY YX YY RX RY LX LY XR XL YR YL PP PQ QP QQ "
                "RCircular LCircular Linear Ptotal Plinear PFtotal PFlinear Pangle").split()
STOKES_TYPE_MAP = {k: i for i, k in enumerate(STOKES_TYPES)}
STOKES_ID_MAP = {i: k for i, k in enumerate(STOKES_TYPES)}

# product codes of include/afhip.h
ADD, SUB, ADDJ, SUBJ, HALF_ADD, HALF_SUB, HALF_SUB_OVER_J = range(7)

# output name -> candidate (operand pair, product), in the reference's order of preference
# (africanus/model/coherency/conversion.py:18-48); correlations come first so that the "known outputs"
# message lists them as the reference does
PRODUCTS = {
    "RR": ((("I", "V"), ADD),), "RL": ((("Q", "U"), ADDJ),), "LR": ((("Q", "U"), SUBJ),), "LL": ((("I", "V"), SUB),),
    "XX": ((("I", "Q"), ADD),), "XY": ((("U", "V"), ADDJ),), "YX": ((("U", "V"), SUBJ),), "YY": ((("I", "Q"), SUB),),
    "I": ((("XX", "YY"), HALF_ADD), (("RR", "LL"), HALF_ADD)),
    "Q": ((("XX", "YY"), HALF_SUB), (("RL", "LR"), HALF_ADD)),
    "U": ((("XY", "YX"), HALF_ADD), (("RL", "LR"), HALF_SUB_OVER_J)),
    "V": ((("XY", "YX"), HALF_SUB_OVER_J), (("RR", "LL"), HALF_SUB)),
}
_CORRELATIONS = ("RR", "RL", "LR", "LL", "XX", "XY", "YX", "YY")
_KINDS = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.complex64): 2, np.dtype(np.complex128): 3}


class DimensionMismatch(Exception):
    pass


class MissingConversionInputs(Exception):
    pass


def _schema_positions(schema):
    """{name: flat C-order position} and the shape of a (possibly nested) schema; level by level as the
    reference walks it (conversion.py:91-140), so that the same malformed schema raises the same error."""
    if not isinstance(schema, (tuple, list)):
        schema = [schema]
    shape, where = [], {}
    level = [(schema, ())]
    depth = 0
    while level:
        below = []
        for seq, at in level:
            if len(shape) <= depth:
                shape.append(len(seq))
            elif shape[depth] != len(seq):
                raise DimensionMismatch("Dimension mismatch %d != %d at depth %d" % (shape[depth], len(seq), depth))
            for i, e in enumerate(seq):
                if isinstance(e, (tuple, list)):
                    below.append((e, at + (i,)))
                    continue
                if not isinstance(e, str):
                    if not np.issubdtype(type(e), np.integer):
                        raise TypeError("Invalid type '%s' for element '%s'" % (type(e), e))
                    try:
                        e = STOKES_ID_MAP[e]
                    except KeyError:
                        raise ValueError("Invalid id '%s'. Valid id's '%s'" % (e, pformat(STOKES_ID_MAP)))
                if e in where:
                    raise ValueError("'%s' defined multiple times" % e)
                where[e] = at + (i,)
        level = below
        depth += 1
    shape = tuple(shape)
    flat = {k: int(np.ravel_multi_index(idx, shape)) for k, idx in where.items()}
    return flat, shape


def convert_setup(input, input_schema, output_schema, implicit_stokes):
    """Resolves the schemas: ((src1, src2, op, out_pos) per output, input corr shape, output corr shape, dtype).
    Same checks and messages as conversion.py:143-204."""
    in_pos, in_shape = _schema_positions(input_schema)
    out_pos, out_shape = _schema_positions(output_schema)
    if tuple(input.shape[len(input.shape) - len(in_shape):]) != in_shape:
        raise ValueError("Last dimension of input doesn't match input schema")
    in_dtype = np_dtype_of(input)
    real_in = in_dtype.kind != "c"
    mapping, all_real = [], True
    for name, pos in out_pos.items():
        if name not in PRODUCTS:
            raise ValueError("Unknown output %s. Known outputs: %s" % (name, list(PRODUCTS.keys())))
        defaults = implicit_stokes and name in _CORRELATIONS   # a missing Stokes parameter counts as 0
        chosen = None
        for (n1, n2), op in PRODUCTS[name]:
            if (n1 in in_pos or defaults) and (n2 in in_pos or defaults):
                score = (n1 in in_pos) + (n2 in in_pos)
                if chosen is None or score > chosen[0]:
                    chosen = (score, in_pos.get(n1, -1), in_pos.get(n2, -1), op)
        if chosen is None:
            # the reference formats its own schema objects into this message
            raise MissingConversionInputs(
                "None of the supplied inputs '%s' can produce output '%s'. It can be produced by the following "
                "combinations '%s'." % (input_schema, name, dict.fromkeys(p for p, _ in PRODUCTS[name]).keys()))
        mapping.append((chosen[1], chosen[2], chosen[3], pos))
        all_real = all_real and real_in and chosen[3] in (HALF_ADD, HALF_SUB)
    # numpy >= 2 promotion of the reference's lambdas applied to input.dtype.type(0) (conversion.py:183-203)
    if in_dtype.kind in "biu":
        base = np.dtype(np.float64)
    elif in_dtype in _KINDS:
        base = np.dtype(np.float32) if in_dtype.itemsize == (8 if in_dtype.kind == "c" else 4) else np.dtype(np.float64)
    else:
        raise TypeError("convert: unsupported input dtype %s" % in_dtype)
    out_dtype = base if all_real else np.result_type(base, np.complex64)
    return mapping, in_shape, out_shape, out_dtype


def convert_impl(input, mapping, in_shape, out_shape, dtype):
    """Applies a resolved mapping on the device (conversion.py:207-216 `convert_impl`)."""
    dtype = np.dtype(dtype)
    if dtype not in _KINDS:
        raise TypeError("convert: unsupported output dtype %s" % dtype)
    # the arithmetic runs in the output's precision: on complex numbers if the input is complex, else on reals
    if np_dtype_of(input).kind == "c" or dtype.kind != "c":
        work_in = dtype
    else:
        work_in = np.dtype(np.float32 if dtype == np.complex64 else np.float64)
    lead = tuple(int(s) for s in input.shape[:len(input.shape) - len(in_shape)])
    nelem = int(np.prod(lead, dtype=np.int64)) if lead else 1
    nin = int(np.prod(in_shape, dtype=np.int64))
    nout = int(np.prod(out_shape, dtype=np.int64))
    if len(mapping) != nout:
        raise ValueError("mapping has %d entries for %d outputs" % (len(mapping), nout))
    src1 = (ctypes.c_int * nout)()
    src2 = (ctypes.c_int * nout)()
    ops = (ctypes.c_int * nout)()
    for s1, s2, op, pos in mapping:
        src1[pos], src2[pos], ops[pos] = s1, s2, op
    with Call(input) as c:
        p_in = c.inp(input, work_in)
        p_out, h = c.out(lead + tuple(out_shape), dtype)
        _lib.call("af_coherency_convert", p_in, _KINDS[work_in], nelem, nin, nout, src1, src2, ops, p_out,
                  _KINDS[dtype], c.stream)
        return c.result(h)


def convert(input, input_schema, output_schema, implicit_stokes=False):
    """
    Converts between Stokes parameters ``I,Q,U,V`` and linear ``XX,XY,YX,YY`` / circular ``RR,RL,LR,LL``
    correlations, in either direction.  ``input`` has shape ``(dim_1, ..., dim_n, icorr_1, ..., icorr_m)`` where the
    trailing dimensions match ``input_schema`` (a possibly nested list of names or casacore Stokes ids); the result
    has shape ``(dim_1, ..., dim_n, ocorr_1, ...)`` following ``output_schema``.  With ``implicit_stokes`` missing
    Stokes inputs count as zero when forming correlations.

    Same contract as ``africanus.model.coherency.convert`` (africanus/model/coherency/conversion.py:207-216): same
    products (:18-48), same choice among candidate operand pairs, same result dtype (real only if every requested
    output is a real product of real inputs), same exceptions.  numpy in -> numpy out; ROCm torch tensor in -> tensor out.
    """
    mapping, in_shape, out_shape, dtype = convert_setup(input, input_schema, output_schema, implicit_stokes)
    return convert_impl(input, mapping, in_shape, out_shape, dtype)
