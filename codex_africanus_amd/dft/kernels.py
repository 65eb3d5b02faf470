"""im_to_vis with the signature of africanus/dft/kernels.py:14-16."""
import contextlib
import os
import threading

import numpy as np

from .. import _lib
from .._device import Call, np_dtype_of

_MODES = {"auto": _lib.AF_DFT_AUTO, "exact": _lib.AF_DFT_EXACT, "recurrence": _lib.AF_DFT_RECURRENCE,
          "valu": _lib.AF_DFT_AUTO | _lib.AF_DFT_VALU_ONLY}
_default_mode = os.environ.get("AFHIP_DFT_MODE", "auto")
if _default_mode not in _MODES:
    raise ValueError("AFHIP_DFT_MODE must be one of %s" % sorted(_MODES))


class _ThreadMode(threading.local):
    """The phasor mode is per THREAD: the transforms are called concurrently from dask worker threads (the
    reference kernels are nogil and stateless, africanus/util/numba.py:9-12), so one caller's set_mode must not
    change what another thread's call in flight computes.  A thread that never called set_mode sees the
    process default (AFHIP_DFT_MODE, else 'auto')."""
    value = None


_tls = _ThreadMode()


def set_mode(mode):
    """Phasor evaluation for the CALLING THREAD: 'auto' (channel recurrence when ``frequency`` is uniformly
    spaced, decided on the device; otherwise the exact path), 'exact' (reference operation
    order + full-accuracy sincos per (row, source, chan)), 'recurrence' (force), 'valu' ('auto' with
    im_to_vis kept on the VALU recurrence kernels instead of the MFMA-accumulator ones).  ``None`` returns the
    thread to the process default."""
    if mode is not None and mode not in _MODES:
        raise ValueError("mode must be one of %s" % sorted(_MODES))
    _tls.value = mode


def get_mode():
    return _tls.value if _tls.value is not None else _default_mode


@contextlib.contextmanager
def mode(name):
    """``with dft.mode('exact'): ...`` -- set_mode for the duration of a block, this thread only."""
    previous = _tls.value
    set_mode(name)
    try:
        yield
    finally:
        _tls.value = previous


def im_to_vis(image, uvw, lm, frequency, convention="fourier", dtype=None):
    """
    Direct Fourier transform image -> visibilities,
    ``vis[r,nu,c] = sum_s exp(-+2 pi i (u l + v m + w (n-1)) nu / c) * image[s,nu,c]``.

    Same contract as ``africanus.dft.im_to_vis`` (africanus/dft/kernels.py:14-69):
    ``image`` (source, chan, corr) real or complex, ``uvw`` (row, 3), ``lm`` (source, 2),
    ``frequency`` (chan,) -> complex (row, chan, corr); output dtype ``dtype`` or
    ``result_type(complex64, image, uvw, lm, frequency)``; ``n`` is NOT clamped (NaN outside
    the unit disc, kernels.py:54) and zero pixels are skipped (kernels.py:64).
    Arithmetic is float64 on the device, with one exception: when EVERY input is single precision (float32 / complex64
    image, float32 uvw, lm and frequency) and the result is complex64 -- the case in which the reference runs its whole
    loop in float32 -- ``af_im_to_vis_f32`` computes phases in float64 and phasors, recurrence and sums in float32
    (csrc/af_im_to_vis_f32.hip: at least as close to the float64 transform of those inputs as the reference's float32
    loop, at about twice the float64 rate).  ``AFHIP_DFT_F32=0`` keeps such calls on the float64 path.
    """
    if convention not in _lib.CONVENTION:
        raise ValueError("convention not in ('fourier', 'casa')")
    img_dt = np_dtype_of(image)
    if dtype is None:
        out_dtype = np.result_type(np.complex64, img_dt, *[np_dtype_of(a) for a in (uvw, lm, frequency)])
    else:
        out_dtype = np.dtype(dtype)
    if len(image.shape) != 3:
        raise ValueError("image must have shape (source, chan, corr)")
    if len(uvw.shape) != 2 or uvw.shape[1] != 3:
        raise ValueError("uvw must have shape (row, 3)")
    if len(lm.shape) != 2 or lm.shape[1] != 2:
        raise ValueError("lm must have shape (source, 2)")
    nsrc, nchan, ncorr = (int(s) for s in image.shape)
    nrow = int(uvw.shape[0])
    if int(lm.shape[0]) != nsrc or tuple(frequency.shape) != (nchan,):
        raise ValueError("image (source, chan, corr), lm (source, 2) and frequency (chan,) disagree")
    is_cplx = img_dt.kind == "c"
    single = (out_dtype == np.dtype(np.complex64) and img_dt in (np.dtype(np.float32), np.dtype(np.complex64))
              and all(np_dtype_of(a) == np.dtype(np.float32) for a in (uvw, lm, frequency))
              and ncorr in (1, 2, 4) and get_mode() != "valu" and os.environ.get("AFHIP_DFT_F32", "1") != "0")
    if single:
        with Call(image, uvw, lm, frequency) as c:
            p_img = c.inp(image, np.complex64 if is_cplx else np.float32)
            p_uvw, p_lm, p_fr = c.inp(uvw, np.float32), c.inp(lm, np.float32), c.inp(frequency, np.float32)
            p_out, h = c.out((nrow, nchan, ncorr), np.complex64)
            ws_bytes = int(_lib.load().af_im_to_vis_f32_workspace_bytes(nsrc, nchan, ncorr, int(is_cplx)))
            p_ws = c.scratch(ws_bytes)
            _lib.call("af_im_to_vis_f32", p_img, int(is_cplx), p_uvw, p_lm, p_fr, nsrc, nrow, nchan, ncorr,
                      _lib.CONVENTION[convention], _MODES[get_mode()], p_out, p_ws, max(ws_bytes, 256), c.stream)
            return c.result(h)
    with Call(image, uvw, lm, frequency) as c:
        p_img = c.inp(image, np.complex128 if is_cplx else np.float64)
        p_uvw, p_lm, p_fr = c.inp(uvw, np.float64), c.inp(lm, np.float64), c.inp(frequency, np.float64)
        p_out, h = c.out((nrow, nchan, ncorr), np.complex128)
        ws_bytes = _lib.load().af_im_to_vis_workspace_bytes(nsrc, nchan, ncorr, int(is_cplx))
        p_ws = c.scratch(ws_bytes)
        conv, mode_, ws_n = _lib.CONVENTION[convention], _MODES[get_mode()], max(int(ws_bytes), 256)

        def launch(r0, r1, p_rows):     # rows are independent: a row chunk is the same call on a slice of uvw
            import ctypes
            _lib.call("af_im_to_vis_f64", p_img, int(is_cplx), ctypes.c_void_p(p_uvw.value + r0 * 24), p_lm, p_fr, nsrc,
                      r1 - r0, nchan, ncorr, conv, mode_, p_rows, p_ws, ws_n, c.stream)

        # numpy callers: the result comes back in row chunks whose downloads overlap the next chunk's transform
        return c.result_rows(h, launch, cast=None if out_dtype == np.complex128 else out_dtype)


def im_to_vis_chi2(image, uvw, lm, frequency, data, weight=None, convention="fourier"):
    """
    ``vis = im_to_vis(image, uvw, lm, frequency)`` and ``chi2[nu] = sum_{row, corr} [weight] |data - vis|^2`` in ONE
    device call (``af_im_to_vis_chi2_f64``): the step of the row-sharded predict -- the transform
    (africanus/dft/kernels.py:14-69), the shard's per-channel chi-squared, then one all-reduce of that vector
    (``sharding.allreduce_chi2``).  With 4 correlations on a uniformly spaced band the sum is formed in the transform's
    epilogue from the visibilities still in registers; otherwise by the separate pass (``sharding.chi2``'s kernel).
    float64 / complex128.  ``data`` (row, chan, corr) complex, ``weight`` the same shape, real, or None.
    Returns ``(vis, chi2)``: complex128 (row, chan, corr), float64 (chan,).
    """
    if convention not in _lib.CONVENTION:
        raise ValueError("convention not in ('fourier', 'casa')")
    if len(image.shape) != 3 or len(uvw.shape) != 2 or uvw.shape[1] != 3 or len(lm.shape) != 2 or lm.shape[1] != 2:
        raise ValueError("image (source, chan, corr), uvw (row, 3), lm (source, 2) expected")
    nsrc, nchan, ncorr = (int(s) for s in image.shape)
    nrow = int(uvw.shape[0])
    if int(lm.shape[0]) != nsrc or tuple(frequency.shape) != (nchan,):
        raise ValueError("image (source, chan, corr), lm (source, 2) and frequency (chan,) disagree")
    if tuple(int(x) for x in data.shape) != (nrow, nchan, ncorr):
        raise ValueError("data must have the shape of the visibilities %s" % ((nrow, nchan, ncorr),))
    if weight is not None and tuple(int(x) for x in weight.shape) != (nrow, nchan, ncorr):
        raise ValueError("weight must have the shape of the visibilities %s" % ((nrow, nchan, ncorr),))
    is_cplx = np_dtype_of(image).kind == "c"
    with Call(image, uvw, lm, frequency, data, weight) as c:
        p_img = c.inp(image, np.complex128 if is_cplx else np.float64)
        p_uvw, p_lm, p_fr = c.inp(uvw, np.float64), c.inp(lm, np.float64), c.inp(frequency, np.float64)
        p_d, p_w = c.inp(data, np.complex128), c.inp(weight, np.float64)
        p_out, h = c.out((nrow, nchan, ncorr), np.complex128)
        p_chi, hc = c.out((nchan,), np.float64)
        ws_bytes = _lib.load().af_im_to_vis_workspace_bytes(nsrc, nchan, ncorr, int(is_cplx))
        p_ws = c.scratch(ws_bytes)
        _lib.call("af_im_to_vis_chi2_f64", p_img, int(is_cplx), p_uvw, p_lm, p_fr, nsrc, nrow, nchan, ncorr,
                  _lib.CONVENTION[convention], _MODES[get_mode()], p_out, p_d, p_w, p_chi, p_ws, max(int(ws_bytes), 256),
                  c.stream)
        return c.result(h), c.result(hc)


def vis_to_im(vis, uvw, lm, frequency, flags, convention="fourier", dtype=None):
    """
    Adjoint direct Fourier transform visibilities -> image,
    ``im[s,nu,c] = sum_r Re(vis[r,nu,c] exp(+-2 pi i (u l + v m + w (n-1)) nu / c))`` over the
    unflagged (row, chan).

    Same contract as ``africanus.dft.vis_to_im`` (africanus/dft/kernels.py:72-148): ``vis``
    (row, chan, corr) real or complex, ``uvw`` (row, 3), ``lm`` (source, 2), ``frequency`` (chan,),
    ``flags`` (row, chan, corr) boolean -- a (row, chan) is dropped when ANY of its correlations is
    flagged (:139-140) -> float (source, chan, corr); output dtype ``dtype`` or
    ``result_type(real type of vis, uvw, lm, frequency)``; 'fourier' is exp(+2 pi i ...) here, the
    opposite of ``im_to_vis`` (:113-118).  Arithmetic is float64 on the device.
    """
    if convention not in _lib.CONVENTION:
        raise ValueError("convention not in ('fourier', 'casa')")
    vdt = np_dtype_of(vis)
    if dtype is None:
        vreal = np.empty(0, vdt).real.dtype
        out_dtype = np.result_type(vreal, *[np_dtype_of(a) for a in (uvw, lm, frequency)])
    else:
        out_dtype = np.dtype(dtype)
        if out_dtype.kind == "c":
            raise TypeError("dtype must be real")
    if len(vis.shape) != 3 or tuple(vis.shape) != tuple(flags.shape):
        raise ValueError("vis and flags must both have shape (row, chan, corr)")
    if len(uvw.shape) != 2 or uvw.shape[1] != 3 or len(lm.shape) != 2 or lm.shape[1] != 2:
        raise ValueError("uvw must be (row, 3) and lm (source, 2)")
    nrow, nchan, ncorr = (int(s) for s in vis.shape)
    nsrc = int(lm.shape[0])
    if int(uvw.shape[0]) != nrow or tuple(frequency.shape) != (nchan,):
        raise ValueError("vis (row, chan, corr), uvw (row, 3) and frequency (chan,) disagree")
    single = (out_dtype == np.dtype(np.float32) and vdt in (np.dtype(np.complex64), np.dtype(np.float32))
              and all(np_dtype_of(a) == np.dtype(np.float32) for a in (uvw, lm, frequency))
              and ncorr in (1, 2, 4) and os.environ.get("AFHIP_DFT_F32", "1") != "0")
    if single:   # every input single precision: phasors in float64, products and sums in float32 (af_vis_to_im_f32)
        with Call(vis, uvw, lm, frequency, flags) as c:
            p_vis = c.inp(vis, np.complex64)
            p_uvw, p_lm, p_fr = c.inp(uvw, np.float32), c.inp(lm, np.float32), c.inp(frequency, np.float32)
            p_fl = c.inp(flags, np.uint8)
            p_out, h = c.out((nsrc, nchan, ncorr), np.float32)
            ws_bytes = int(_lib.load().af_vis_to_im_f32_workspace_bytes(nsrc, nrow, nchan, ncorr))
            p_ws = c.scratch(ws_bytes)
            _lib.call("af_vis_to_im_f32", p_vis, p_uvw, p_lm, p_fr, p_fl, nsrc, nrow, nchan, ncorr,
                      _lib.CONVENTION[convention], _MODES[get_mode()] & ~_lib.AF_DFT_VALU_ONLY, p_out, p_ws,
                      max(ws_bytes, 256), c.stream)
            return c.result(h)
    with Call(vis, uvw, lm, frequency, flags) as c:
        p_vis = c.inp(vis, np.complex128)
        p_uvw, p_lm, p_fr = c.inp(uvw, np.float64), c.inp(lm, np.float64), c.inp(frequency, np.float64)
        p_fl = c.inp(flags, np.uint8)
        p_out, h = c.out((nsrc, nchan, ncorr), np.float64)
        ws_bytes = int(_lib.load().af_vis_to_im_workspace_bytes(nsrc, nrow, nchan, ncorr))
        p_ws = c.scratch(ws_bytes)
        _lib.call("af_vis_to_im_f64", p_vis, p_uvw, p_lm, p_fr, p_fl, nsrc, nrow, nchan, ncorr,
                  _lib.CONVENTION[convention], _MODES[get_mode()] & ~_lib.AF_DFT_VALU_ONLY, p_out, p_ws,
                  max(ws_bytes, 256), c.stream)
        return c.result(h, cast=None if out_dtype == np.float64 else out_dtype)


# ---------------------------------------------------------------------------------------- predict from the sky model
_STOKES = ("I", "Q", "U", "V")
_SPECTRAL_BASES = {0: 0, 1: 1, 2: 2, "std": 0, "log": 1, "log10": 2}


def model_tables(stokes, spi, corr_schema, base, stokes_schema=None, implicit_stokes=False):
    """Host-side resolution shared by the model-level entry points: validates (stokes, spi) like ``spectral_model``
    (africanus/model/spectral/spec_model.py:102-236), resolves ``corr_schema`` like ``convert``
    (africanus/model/coherency/conversion.py:143-204) and returns (base int32 array, ctypes tables src1 / src2 / op,
    npol, ncorr, correlation shape, image_is_complex)."""
    import ctypes
    from ..model.coherency.conversion import convert_setup
    if len(stokes.shape) != 2 or len(spi.shape) != 3:
        raise ValueError("stokes must be (source, pol) and spi (source, spi-comps, pol)")
    npol = int(stokes.shape[1])
    if int(spi.shape[2]) != npol:
        raise ValueError("Correlations on stokes and spi don't agree")
    if stokes_schema is None:
        stokes_schema = list(_STOKES[:npol])
    bl = list(base) if isinstance(base, (list, tuple)) else [base] * npol
    bl = bl + [bl[-1]] * (npol - len(bl))
    try:
        b = np.array([_SPECTRAL_BASES[x] for x in bl[:npol]], dtype=np.int32)
    except (KeyError, TypeError):
        raise ValueError("Invalid base")
    probe = np.empty((1, npol), dtype=np.float64)       # the spectra are float64: the dtype convert() would see
    mapping, in_shape, out_shape, out_dtype = convert_setup(probe, stokes_schema, corr_schema, implicit_stokes)
    ncorr = int(np.prod(out_shape, dtype=np.int64))
    src1, src2, ops = (ctypes.c_int * ncorr)(), (ctypes.c_int * ncorr)(), (ctypes.c_int * ncorr)()
    for s1, s2, op, pos in mapping:
        src1[pos], src2[pos], ops[pos] = s1, s2, op
    return b, (src1, src2, ops), npol, ncorr, tuple(out_shape), np.dtype(out_dtype).kind == "c"


def im_to_vis_from_model(stokes, spi, ref_freq, uvw, lm, frequency, corr_schema=(("XX", "XY"), ("YX", "YY")), base=0,
                         convention="fourier", dtype=None, stokes_schema=None):
    """
    ``im_to_vis(convert(spectral_model(stokes, spi, ref_freq, frequency, base), stokes_schema, corr_schema), uvw, lm,
    frequency)`` -- the model steps of africanus/rime/examples/predict.py:494-498 in front of the direct transform
    (africanus/dft/kernels.py:14-69) -- in ONE device call: the spectra and the correlations are evaluated on the
    device into the call's workspace, the caller never forms (or uploads) a (source, chan, corr) image.

    ``stokes`` (source, pol), ``spi`` (source, spi-comps, pol), ``ref_freq`` (source,), ``stokes_schema`` default
    ``["I", "Q", "U", "V"][:pol]``, ``corr_schema`` e.g. ``[["XX", "XY"], ["YX", "YY"]]``, ``["XX", "YY"]``,
    ``["RR", "LL"]``; returns (row, chan) + the shape of ``corr_schema``.  Same values, bit for bit, as the chain of the
    three stand-alone functions (same kernels); the same exceptions as ``spectral_model`` / ``convert`` / ``im_to_vis``.
    """
    if convention not in _lib.CONVENTION:
        raise ValueError("convention not in ('fourier', 'casa')")
    b, (src1, src2, ops), npol, ncorr, corr_shape, cplx = model_tables(stokes, spi, corr_schema, base, stokes_schema)
    if len(uvw.shape) != 2 or uvw.shape[1] != 3:
        raise ValueError("uvw must have shape (row, 3)")
    if len(lm.shape) != 2 or lm.shape[1] != 2:
        raise ValueError("lm must have shape (source, 2)")
    nsrc, nspi, nchan, nrow = int(stokes.shape[0]), int(spi.shape[1]), int(frequency.shape[0]), int(uvw.shape[0])
    if int(lm.shape[0]) != nsrc or int(spi.shape[0]) != nsrc or tuple(ref_freq.shape) != (nsrc,):
        raise ValueError("stokes, spi, ref_freq and lm disagree on the number of sources")
    if dtype is None:
        out_dtype = np.result_type(np.complex64, *[np_dtype_of(a) for a in (stokes, spi, ref_freq, uvw, lm, frequency)])
    else:
        out_dtype = np.dtype(dtype)
    with Call(stokes, spi, ref_freq, uvw, lm, frequency) as c:
        p_st, p_sp, p_rf = c.inp(stokes, np.float64), c.inp(spi, np.float64), c.inp(ref_freq, np.float64)
        p_uvw, p_lm, p_fr = c.inp(uvw, np.float64), c.inp(lm, np.float64), c.inp(frequency, np.float64)
        p_b = c.inp(b, np.int32)
        p_out, h = c.out((nrow, nchan) + corr_shape, np.complex128)
        ws_bytes = int(_lib.load().af_im_to_vis_model_workspace_bytes(nsrc, nchan, npol, ncorr, int(cplx)))
        p_ws = c.scratch(ws_bytes)
        _lib.call("af_im_to_vis_model_f64", p_st, p_sp, p_rf, p_b, nspi, npol, src1, src2, ops, ncorr, int(cplx), p_uvw,
                  p_lm, p_fr, nsrc, nrow, nchan, _lib.CONVENTION[convention], _MODES[get_mode()], p_out, p_ws,
                  max(ws_bytes, 256), c.stream)
        return c.result(h, cast=None if out_dtype == np.complex128 else out_dtype)
