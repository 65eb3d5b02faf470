"""gridder with the signature of africanus/gridding/perleypolyhedron/gridder.py:12-30."""
import numpy as np

from ... import _lib
from ..._device import Call, _is_torch

# corr2stokes policies as per-correlation factors (policies/stokes_conversion_policies.py:143-180)
CORR_TO_STOKES = {
    "I_FROM_XXYY": [.5, .5], "I_FROM_XXXYYXYY": [.5, 0, 0, .5], "I_FROM_RRLL": [.5, .5], "I_FROM_RRRLLRLL": [.5, 0, 0, .5],
    "Q_FROM_XXYY": [.5, -.5], "Q_FROM_XXXYYXYY": [.5, 0, 0, -.5], "Q_FROM_RRRLLRLL": [0, .5, .5, 0],
    "U_FROM_XYYX": [.5, .5], "U_FROM_XXXYYXYY": [0, .5, .5, 0], "U_FROM_RLLR": [-.5j, .5j],
    "U_FROM_RRRLLRLL": [0, -.5j, .5j, 0], "V_FROM_RRLL": [.5, -.5], "V_FROM_RRRLLRLL": [.5, 0, 0, -.5],
    "V_FROM_XYYX": [-.5j, .5j], "V_FROM_XXXYYXYY": [0, -.5j, .5j, 0],
}
_CONV = {"conv_1d_axisymmetric_unpacked_scatter": 0, "conv_1d_axisymmetric_packed_scatter": 1, "conv_nn_scatter": 2}


def gridder(uvw, vis, wavelengths, chanmap, npix, cell, image_centre, phase_centre, convolution_kernel,
            convolution_kernel_width, convolution_kernel_oversampling, baseline_transform_policy,
            phase_transform_policy, stokes_conversion_policy, convolution_policy, grid_dtype=np.complex128,
            do_normalize=False):
    """
    2-D convolutional gridder, visibilities -> (band, npix, npix) grid; the adjoint of ``degridder``.

    Same contract as ``africanus.gridding.perleypolyhedron.gridder.gridder``
    (africanus/gridding/perleypolyhedron/gridder.py:12-117): ``vis`` (row, chan, corr) complex, ``chanmap`` the band
    of every channel (``nband = max + 1``), ``convolution_policy`` 'conv_1d_axisymmetric_unpacked_scatter',
    'conv_1d_axisymmetric_packed_scatter' or 'conv_nn_scatter', ``stokes_conversion_policy`` any of the 15
    '<stokes>_FROM_<corrs>', ``phase_transform_policy`` 'None' / None / 'phase_rotate' (sign +1, applied to a copy:
    the reference rotates ``vis`` in place), ``baseline_transform_policy`` 'None', ``do_normalize`` to divide every
    band by its summed tap weights.  The adds are hardware atomics: reproducible to rounding.  Off-grid points of the
    nearest-neighbour policy are dropped (the reference indexes out of bounds there).
    """
    if baseline_transform_policy not in ("None", None):
        raise ValueError("Invalid baseline transform policy type" if baseline_transform_policy not in
                         ("rotate", "wlinapprox") else
                         "baseline_transform_policy '%s' has no defined result in the reference" % baseline_transform_policy)
    if phase_transform_policy not in ("None", None, "phase_rotate"):
        raise ValueError("Invalid baseline transform policy type")
    if stokes_conversion_policy not in CORR_TO_STOKES:
        raise ValueError("Invalid stokes conversion")
    if convolution_policy not in _CONV:
        raise ValueError("Invalid convolution policy type")
    nchan = int(np.prod(tuple(wavelengths.shape), dtype=np.int64))
    if int(np.prod(tuple(chanmap.shape), dtype=np.int64)) != nchan:
        raise ValueError("Chanmap and corresponding wavelengths must match in shape")
    if len(uvw.shape) != 2 or uvw.shape[1] != 3:
        raise ValueError("UVW array must be array of tripples")
    if len(vis.shape) != 3 or int(uvw.shape[0]) != int(vis.shape[0]):
        raise ValueError("UVW array must have same number of rows as vis array")
    if int(vis.shape[1]) != nchan:
        raise ValueError("Chanmap must correspond to visibility channels")
    coef = np.asarray(CORR_TO_STOKES[stokes_conversion_policy], dtype=np.complex128)
    ncorr = int(vis.shape[2])
    if coef.shape[0] != ncorr:
        raise ValueError("stokes_conversion_policy '%s' needs %d correlations" % (stokes_conversion_policy, coef.shape[0]))
    W, OS = int(convolution_kernel_width), int(convolution_kernel_oversampling)
    cm = chanmap.detach().cpu().numpy() if _is_torch(chanmap) else np.asarray(chanmap)
    if cm.size and int(cm.min()) < 0:
        raise ValueError("chanmap holds negative band numbers")
    nband = int(cm.max()) + 1 if cm.size else 0
    if convolution_policy != "conv_nn_scatter" and \
            int(np.prod(tuple(convolution_kernel.shape), dtype=np.int64)) != OS * (W + 2):
        raise ValueError("convolution_kernel must hold oversampling * (width + 2) taps")
    nrow, npix = int(uvw.shape[0]), int(npix)
    ic = np.ascontiguousarray(image_centre, dtype=np.float64).reshape(2)
    pc = np.ascontiguousarray(phase_centre, dtype=np.float64).reshape(2)
    out_dtype = np.dtype(grid_dtype)
    with Call(uvw, vis, wavelengths, chanmap, convolution_kernel) as c:
        p_uvw, p_wl, p_k = c.inp(uvw, np.float64), c.inp(wavelengths, np.float64), c.inp(convolution_kernel, np.float64)
        p_v, p_cm, p_cf = c.inp(vis, np.complex128), c.inp(chanmap, np.int64), c.inp(coef, np.complex128)
        p_out, h = c.out((nband, npix, npix), np.complex128)
        ws_bytes = int(_lib.load().af_gridder_workspace_bytes(nrow, nchan, nband, npix))
        p_ws = c.scratch(ws_bytes)
        _lib.call("af_gridder_c128", p_uvw, p_v, p_wl, p_cm, npix, float(cell), ic.ctypes.data, pc.ctypes.data, p_k, W, OS,
                  int(phase_transform_policy == "phase_rotate"), p_cf, ncorr, _CONV[convolution_policy],
                  int(bool(do_normalize)), nrow, nchan, nband, p_out, p_ws, max(ws_bytes, 256), c.stream)
        return c.result(h, cast=None if out_dtype == np.complex128 else out_dtype)
