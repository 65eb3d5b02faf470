"""degridder with the signature of africanus/gridding/perleypolyhedron/degridder.py:79-94."""
import numpy as np

from ... import _lib
from ..._device import Call, _is_torch

# stokes2corr policies as per-correlation factors (policies/stokes_conversion_policies.py:8-137)
STOKES_TO_CORR = {
    "XXYY_FROM_I": [1, 1], "XXXYYXYY_FROM_I": [1, 0, 0, 1], "RRLL_FROM_I": [1, 1], "RRRLLRLL_FROM_I": [1, 0, 0, 1],
    "XXYY_FROM_Q": [1, -1], "XXXYYXYY_FROM_Q": [1, 0, 0, -1], "RLLR_FROM_Q": [1, 1], "RRRLLRLL_FROM_Q": [0, 1, 1, 0],
    "XYYX_FROM_U": [1, 1], "XXXYYXYY_FROM_U": [0, 1, 1, 0], "RLLR_FROM_U": [1j, -1j],
    "RRRLLRLL_FROM_U": [0, 1j, -1j, 0], "XYYX_FROM_V": [1j, -1j], "XXXYYXYY_FROM_V": [0, 1j, -1j, 0],
    "RRLL_FROM_V": [1, -1], "RRRLLRLL_FROM_V": [1, 0, 0, -1],
}
_CONV = {"conv_1d_axisymmetric_packed_gather": 1, "conv_1d_axisymmetric_unpacked_gather": 0}


def degridder(uvw, gridstack, wavelengths, chanmap, cell, image_centre, phase_centre, convolution_kernel,
              convolution_kernel_width, convolution_kernel_oversampling, baseline_transform_policy,
              phase_transform_policy, stokes_conversion_policy, convolution_policy, vis_dtype=np.complex128):
    """
    2-D convolutional degridder, grid -> visibilities.

    Same contract as ``africanus.gridding.perleypolyhedron.degridder.degridder``
    (africanus/gridding/perleypolyhedron/degridder.py:79-175): ``uvw`` (row, 3) [m], ``gridstack``
    (band, npix, npix) complex, ``wavelengths`` (chan,) [m], ``chanmap`` (chan,) band of every channel, ``cell``
    [arcsec], ``image_centre`` / ``phase_centre`` (ra, dec) [rad], ``convolution_kernel`` as produced by
    ``kernels`` (packed for the packed policy), its width (odd) and oversampling -> (row, chan, ncorr) of
    ``vis_dtype``; the same ``ValueError``s.  Policies: ``convolution_policy`` 'conv_1d_axisymmetric_packed_gather' or
    'conv_1d_axisymmetric_unpacked_gather'; ``stokes_conversion_policy`` any of the 16 '<corrs>_FROM_<stokes>';
    ``phase_transform_policy`` 'None' / None / 'phase_rotate'; ``baseline_transform_policy`` 'None' (the
    reference's 'rotate' indexes uvw[3] and its 'wlinapprox' does not compile under numba 0.54: neither has a
    defined result to reproduce).  ``uvw`` is not modified.
    """
    if baseline_transform_policy not in ("None", None):
        raise ValueError("Invalid baseline transform policy type" if baseline_transform_policy not in
                         ("rotate", "wlinapprox") else
                         "baseline_transform_policy '%s' has no defined result in the reference" % baseline_transform_policy)
    if phase_transform_policy not in ("None", None, "phase_rotate"):
        raise ValueError("Invalid baseline transform policy type")
    if stokes_conversion_policy not in STOKES_TO_CORR:
        raise ValueError("Invalid stokes conversion")
    if convolution_policy not in _CONV:
        raise ValueError("Invalid convolution policy type")
    nchan = int(np.prod(tuple(wavelengths.shape), dtype=np.int64))
    if int(np.prod(tuple(chanmap.shape), dtype=np.int64)) != nchan:
        raise ValueError("Chanmap and corresponding wavelengths must match in shape")
    if len(gridstack.shape) != 3 or gridstack.shape[1] != gridstack.shape[2]:
        raise ValueError("Grid must be square")
    if len(uvw.shape) != 2 or uvw.shape[1] != 3:
        raise ValueError("UVW array must be array of tripples")
    cm = chanmap.detach().cpu().numpy() if _is_torch(chanmap) else np.asarray(chanmap)
    if cm.size and int(cm.min()) < 0:
        raise ValueError("chanmap holds negative band numbers")
    if cm.size and int(gridstack.shape[0]) < int(cm.max()) + 1:
        raise ValueError("Not enough channel bands in grid stack to match mfs band mapping")
    W, OS = int(convolution_kernel_width), int(convolution_kernel_oversampling)
    if int(np.prod(tuple(convolution_kernel.shape), dtype=np.int64)) != OS * (W + 2):
        raise ValueError("convolution_kernel must hold oversampling * (width + 2) taps")
    nrow, npix = int(uvw.shape[0]), int(gridstack.shape[1])
    coef = np.asarray(STOKES_TO_CORR[stokes_conversion_policy], dtype=np.complex128)
    ncorr = coef.shape[0]
    ic = np.ascontiguousarray(image_centre, dtype=np.float64).reshape(2)
    pc = np.ascontiguousarray(phase_centre, dtype=np.float64).reshape(2)
    out_dtype = np.dtype(vis_dtype)
    with Call(uvw, gridstack, wavelengths, chanmap, convolution_kernel) as c:
        p_uvw, p_wl, p_k = c.inp(uvw, np.float64), c.inp(wavelengths, np.float64), c.inp(convolution_kernel, np.float64)
        p_g, p_cm, p_cf = c.inp(gridstack, np.complex128), c.inp(chanmap, np.int64), c.inp(coef, np.complex128)
        p_out, h = c.out((nrow, nchan, ncorr), np.complex128)
        ws_bytes = int(_lib.load().af_degridder_workspace_bytes(nrow))
        p_ws = c.scratch(ws_bytes)
        _lib.call("af_degridder_c128", p_uvw, p_g, p_wl, p_cm, float(cell), ic.ctypes.data, pc.ctypes.data, p_k, W, OS,
                  int(phase_transform_policy == "phase_rotate"), p_cf, ncorr, _CONV[convolution_policy], nrow, nchan,
                  npix, p_out, p_ws, max(ws_bytes, 256), c.stream)
        return c.result(h, cast=None if out_dtype == np.complex128 else out_dtype)
