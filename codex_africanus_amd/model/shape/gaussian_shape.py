"""gaussian with the signature of africanus/model/shape/gaussian_shape.py:11-13."""
import numpy as np

from ... import _lib
from ..._device import Call, np_dtype_of


def gaussian(uvw, frequency, shape_params):
    """
    Gaussian shape function ``shape[s,r,f] = exp(-(u1^2 + v1^2) (nu_f gs)^2)`` with
    ``u1 = (u em - v el) er``, ``v1 = u el + v em``, ``el = emaj sin(pa)``, ``em = emaj cos(pa)``,
    ``er = emin / (emaj or 1)``, ``gs = sqrt(2) pi / (fwhm c)``.

    Same contract as ``africanus.model.shape.gaussian`` (africanus/model/shape/gaussian_shape.py:11-62):
    ``uvw`` (row, 3), ``frequency`` (chan,), ``shape_params`` (source, 3) = (major, minor, orientation) in
    radians -> float (source, row, chan) of dtype ``result_type(uvw, frequency, shape_params)``.
    Arithmetic is float64 on the device.
    """
    if len(uvw.shape) != 2 or uvw.shape[1] != 3:
        raise ValueError("uvw must have shape (row, 3)")
    if len(shape_params.shape) != 2 or shape_params.shape[1] != 3:
        raise ValueError("shape_params must have shape (source, 3)")
    if len(frequency.shape) != 1:
        raise ValueError("frequency must have shape (chan,)")
    nsrc, nrow, nchan = int(shape_params.shape[0]), int(uvw.shape[0]), int(frequency.shape[0])
    out_dtype = np.result_type(*[np_dtype_of(a) for a in (uvw, frequency, shape_params)])
    with Call(uvw, frequency, shape_params) as c:
        p_uvw, p_fr, p_sp = (c.inp(a, np.float64) for a in (uvw, frequency, shape_params))
        p_out, h = c.out((nsrc, nrow, nchan), np.float64)
        ws_bytes = max(nsrc * 32, 256)
        p_ws = c.scratch(ws_bytes)
        _lib.call("af_gaussian_shape_f64", p_uvw, p_fr, p_sp, nsrc, nrow, nchan, p_out, p_ws, ws_bytes, c.stream)
        return c.result(h, cast=None if out_dtype == np.float64 else out_dtype)
