"""wsclean_predict / spectra with the signatures of africanus/rime/wsclean_predict.py:86-92 and
africanus/model/wsclean/spec_model.py:70-71."""
import numpy as np

from .. import _lib
from .._device import Call, np_dtype_of, _is_torch
from ..dft import kernels as _dft


def _byte_flags(x, nsrc, what):
    """(nsrc,) boolean-like -> uint8; a scalar bool broadcasts (spec_model.py:33-66)."""
    if _is_torch(x):
        return x
    a = np.asarray(x)
    if a.ndim == 0:
        a = np.full((nsrc,), bool(a))
    if a.shape != (nsrc,):
        raise ValueError("%s must have shape (source,)" % what)
    return np.ascontiguousarray(a != 0, dtype=np.uint8)


def _gaussian_flags(source_type, nsrc):
    """'POINT' / 'GAUSSIAN' strings (wsclean_predict.py:34,48,79-80) -> uint8; boolean arrays /
    tensors are taken as ``is_gaussian`` directly."""
    if _is_torch(source_type):
        return source_type
    st = np.asarray(source_type)
    if st.shape != (nsrc,):
        raise ValueError("source_type must have shape (source,)")
    if st.dtype.kind in "US":
        st = st.astype(str)
        point, gauss = st == "POINT", st == "GAUSSIAN"
        if not np.all(point | gauss):
            raise ValueError("source_type must be POINT or GAUSSIAN")
        return np.ascontiguousarray(gauss, dtype=np.uint8)
    return np.ascontiguousarray(st != 0, dtype=np.uint8)


def _check_spectral_args(flux, coeffs, ref_freq):
    if len(coeffs.shape) != 2 or len(flux.shape) != 1 or len(ref_freq.shape) != 1:
        raise ValueError("flux (source,), coeffs (source, coeffs) and ref_freq (source,) expected")
    if not (int(flux.shape[0]) == int(coeffs.shape[0]) == int(ref_freq.shape[0])):
        raise ValueError("first dimensions of I, coeffs and ref_freq don't match.")


def spectra(I, coeffs, log_poly, ref_freq, frequency):  # noqa: E741
    """
    WSClean spectral model, (source, chan) float.  Same contract as
    ``africanus.model.wsclean.spectra`` (africanus/model/wsclean/spec_model.py:70-126): ordinary
    polynomials ``I + sum_k coeffs[k] (nu/ref - 1)^(k+1)`` or, where ``log_poly`` is set, logarithmic
    ones ``I exp(sum_k coeffs[k] log(nu/ref)^(k+1))``; ``log_poly`` is a (source,) array or one bool.
    """
    _check_spectral_args(I, coeffs, ref_freq)
    nsrc, ncoeffs = int(coeffs.shape[0]), int(coeffs.shape[1])
    if not _is_torch(log_poly) and np.ndim(log_poly) == 1 and np.shape(log_poly)[0] != nsrc:
        raise ValueError("coeffs.shape[0] != log_poly.shape[0]")
    nchan = int(frequency.shape[0])
    out_dtype = np.result_type(*[np_dtype_of(a) for a in (I, coeffs, ref_freq, frequency)])
    lp = _byte_flags(log_poly, nsrc, "log_poly")
    with Call(I, coeffs, ref_freq, frequency, lp) as c:
        p_i, p_co, p_rf, p_fr = (c.inp(a, np.float64) for a in (I, coeffs, ref_freq, frequency))
        p_lp = c.inp(lp, np.uint8)
        p_out, h = c.out((nsrc, nchan), np.float64)
        _lib.call("af_wsclean_spectra_f64", p_i, p_co, p_lp, p_rf, p_fr, nsrc, ncoeffs, nchan, p_out, c.stream)
        return c.result(h, cast=None if out_dtype == np.float64 else out_dtype)


def wsclean_predict(uvw, lm, source_type, flux, coeffs, log_poly, ref_freq, gauss_shape, frequency):
    """
    Predict single-correlation visibilities from a WSClean component list.

    Same contract as ``africanus.rime.wsclean_predict`` (africanus/rime/wsclean_predict.py:86-171):
    ``uvw`` (row, 3), ``lm`` (source, 2), ``source_type`` (source,) strings ``"POINT"`` / ``"GAUSSIAN"``
    (anything else raises ``ValueError``), ``flux`` (source,), ``coeffs`` (source, coeffs), ``log_poly``
    (source,) bool, ``ref_freq`` (source,), ``gauss_shape`` (source, 3) = (major, minor, orientation) in
    radians, ``frequency`` (chan,) -> complex (row, chan, 1) of dtype
    ``result_type(complex64, uvw, lm, flux, coeffs, ref_freq, frequency)``.  CASA sign, ``n`` unclamped.
    A boolean array / tensor is accepted for ``source_type`` (True = Gaussian) so that the whole call
    can stay on the device.  Phasor / envelope evaluation follows :func:`codex_africanus_amd.dft.set_mode`.
    """
    if len(uvw.shape) != 2 or uvw.shape[1] != 3:
        raise ValueError("uvw must have shape (row, 3)")
    if len(lm.shape) != 2 or lm.shape[1] != 2:
        raise ValueError("lm must have shape (source, 2)")
    _check_spectral_args(flux, coeffs, ref_freq)
    nsrc, nrow, nchan, ncoeffs = int(lm.shape[0]), int(uvw.shape[0]), int(frequency.shape[0]), int(coeffs.shape[1])
    if int(flux.shape[0]) != nsrc or tuple(gauss_shape.shape) != (nsrc, 3):
        raise ValueError("lm (source, 2), flux (source,) and gauss_shape (source, 3) disagree")
    out_dtype = np.result_type(np.complex64, *[np_dtype_of(a) for a in (uvw, lm, flux, coeffs, ref_freq, frequency)])
    isg = _gaussian_flags(source_type, nsrc)
    lp = _byte_flags(log_poly, nsrc, "log_poly")
    with Call(uvw, lm, isg, flux, coeffs, lp, ref_freq, gauss_shape, frequency) as c:
        p_uvw, p_lm, p_fl, p_co, p_rf, p_gs, p_fr = (
            c.inp(a, np.float64) for a in (uvw, lm, flux, coeffs, ref_freq, gauss_shape, frequency))
        p_isg, p_lp = c.inp(isg, np.uint8), c.inp(lp, np.uint8)
        p_out, h = c.out((nrow, nchan, 1), np.complex128)
        ws_bytes = int(_lib.load().af_wsclean_predict_workspace_bytes(nsrc, nchan))
        p_ws = c.scratch(ws_bytes)
        _lib.call("af_wsclean_predict_f64", p_uvw, p_lm, p_isg, p_fl, p_co, p_lp, p_rf, p_gs, p_fr, nsrc, nrow,
                  nchan, ncoeffs, _dft._MODES[_dft.get_mode()] & ~_lib.AF_DFT_VALU_ONLY, p_out, p_ws, max(ws_bytes, 256), c.stream)
        return c.result(h, cast=None if out_dtype == np.complex128 else out_dtype)
