"""feed_rotation with the signature of africanus/rime/feeds.py:50."""
import numpy as np

from .. import _lib
from .._device import Call, np_dtype_of

_FEEDS = {"linear": 0, "circular": 1}


def feed_rotation(parallactic_angles, feed_type="linear"):
    """
    2x2 feed rotation matrices: linear ``[[cos pa, sin pa], [-sin pa, cos pa]]``, circular
    ``diag(exp(-i pa), exp(+i pa))``.

    Same contract as ``africanus.rime.feed_rotation`` (africanus/rime/feeds.py:50-73): ``parallactic_angles``
    of any shape, float32 or float64 (anything else raises ``ValueError``) -> complex (..., 2, 2) of matching
    precision; an unknown ``feed_type`` raises ``ValueError("Invalid feed_type '...'")``.
    """
    if feed_type not in _FEEDS:
        raise ValueError("Invalid feed_type '%s'" % feed_type)
    dt = np_dtype_of(parallactic_angles)
    if dt == np.float32:
        rt, ct, fn = np.float32, np.complex64, "af_feed_rotation_f32"
    elif dt == np.float64:
        rt, ct, fn = np.float64, np.complex128, "af_feed_rotation_f64"
    else:
        raise ValueError("parallactic_angles has none-floating point type %s" % dt)
    shape = tuple(int(s) for s in parallactic_angles.shape)
    n = int(np.prod(shape, dtype=np.int64))
    with Call(parallactic_angles) as c:
        p_pa = c.inp(parallactic_angles, rt)
        p_out, h = c.out(shape + (2, 2), ct)
        _lib.call(fn, p_pa, n, _FEEDS[feed_type], p_out, c.stream)
        return c.result(h)
