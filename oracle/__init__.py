"""
TEST INFRASTRUCTURE ONLY -- CPU oracle for the RIME predict hot path.

ctypes front-end of ``oracle/rime_oracle.c``, a plain-C restatement of the
reference's numba kernels (file:line citations live in the C sources).  The
functions keep the reference's numpy signatures so parity tests read like the
reference's own tests.

Parity pinned: ``tests/test_oracle_golden.py`` checks every function against
golden vectors generated from the real reference in the build container
(``tests/golden/make_golden.py``) and against the reference tests' own
known-answer values.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; ``codex_africanus_amd`` never does.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIBS = {}

_i64 = ctypes.c_int64
_int = ctypes.c_int
_vp = ctypes.c_void_p


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    targets = [os.path.join(_BUILD, n) for n in ("liboracle.so", "liboracle_omp.so")]
    srcs = [os.path.join(_HERE, n) for n in ("rime_oracle.c", "rime_oracle_impl.h")]
    stale = force or not all(os.path.exists(t) for t in targets) or any(
        os.path.getmtime(s) > min(os.path.getmtime(t) for t in targets) for s in srcs
    )
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return targets


def _lib(omp=False):
    key = bool(omp)
    if key not in _LIBS:
        name = "liboracle_omp.so" if omp else "liboracle.so"
        path = os.path.join(_BUILD, name)
        if not os.path.exists(path):
            build()
        _LIBS[key] = ctypes.CDLL(path)
    return _LIBS[key]


def num_threads(omp=True):
    return int(_lib(omp).orc_num_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _sign(convention):
    # africanus/rime/phase.py:29-34, africanus/dft/kernels.py:34-39
    if convention == "fourier":
        return -1
    elif convention == "casa":
        return 1
    raise ValueError("convention not in ('fourier', 'casa')")


def phase_delay(lm, uvw, frequency, convention="fourier"):
    """africanus/rime/phase.py:11-63."""
    sign = _sign(convention)
    out_dtype = np.result_type(np.complex64, lm.dtype, uvw.dtype, frequency.dtype)
    # all-float32 inputs compute in float32 (constants are cast to lm.dtype, phase.py:23-25);
    # anything mixed is promoted to float64 first (numba's per-operation promotion
    # of mixed inputs is not restated).
    if out_dtype == np.complex64:
        rt, fn = np.float32, "orc_phase_delay_f32"
    else:
        rt, fn = np.float64, "orc_phase_delay_f64"
    lm_, uvw_, fr_ = _c(lm, rt), _c(uvw, rt), _c(frequency, rt)
    nsrc, nrow, nchan = lm_.shape[0], uvw_.shape[0], fr_.shape[0]
    out = np.empty((nsrc, nrow, nchan), dtype=out_dtype)
    rc = getattr(_lib(), fn)(_p(lm_), _i64(nsrc), _p(uvw_), _i64(nrow), _p(fr_), _i64(nchan),
                             _int(sign), _p(out))
    assert rc == 0
    return out


def im_to_vis(image, uvw, lm, frequency, convention="fourier", dtype=None, omp=False):
    """africanus/dft/kernels.py:14-69."""
    sign = _sign(convention)
    if dtype is None:
        out_dtype = np.result_type(np.complex64, image.dtype, uvw.dtype, lm.dtype, frequency.dtype)
    else:
        out_dtype = np.dtype(dtype)
    is_cplx = np.iscomplexobj(image)
    img = _c(image, np.complex128 if is_cplx else np.float64)
    uvw_, lm_, fr_ = _c(uvw, np.float64), _c(lm, np.float64), _c(frequency, np.float64)
    nsrc, nchan, ncorr = img.shape
    nrow = uvw_.shape[0]
    out = np.empty((nrow, nchan, ncorr), dtype=np.complex128)
    rc = _lib(omp).orc_im_to_vis_f64(_p(img), _int(int(is_cplx)), _p(uvw_), _p(lm_), _p(fr_),
                                     _i64(nsrc), _i64(nrow), _i64(nchan), _i64(ncorr),
                                     _int(sign), _int(int(out_dtype == np.complex64)), _p(out))
    assert rc == 0
    return out.astype(out_dtype, copy=False)


def vis_to_im(vis, uvw, lm, frequency, flags, convention="fourier", dtype=None, omp=False):
    """africanus/dft/kernels.py:72-148."""
    sign = _sign(convention)
    if dtype is None:
        vdt = np.dtype(vis.dtype)
        vreal = np.empty(0, vdt).real.dtype
        out_dtype = np.result_type(vreal, uvw.dtype, lm.dtype, frequency.dtype)
    else:
        out_dtype = np.dtype(dtype)
    assert vis.shape == flags.shape
    v_ = _c(vis, np.complex128)
    uvw_, lm_, fr_ = _c(uvw, np.float64), _c(lm, np.float64), _c(frequency, np.float64)
    fl_ = _c(flags, np.uint8)
    nrow, nchan, ncorr = v_.shape
    nsrc = lm_.shape[0]
    out = np.empty((nsrc, nchan, ncorr), dtype=np.float64)
    rc = _lib(omp).orc_vis_to_im_f64(_p(v_), _p(uvw_), _p(lm_), _p(fr_), _p(fl_), _i64(nsrc), _i64(nrow),
                                     _i64(nchan), _i64(ncorr), _int(sign), _p(out))
    assert rc == 0
    return out.astype(out_dtype, copy=False)


def spectra(I, coeffs, log_poly, ref_freq, frequency):  # noqa: E741
    """africanus/model/wsclean/spec_model.py:70-126."""
    I_, co, rf, fr = (_c(a, np.float64) for a in (I, coeffs, ref_freq, frequency))
    if co.ndim != 2 or not (I_.shape[0] == co.shape[0] == rf.shape[0]):
        raise ValueError("first dimensions of I, coeffs and ref_freq don't match.")
    lp = np.asarray(log_poly)
    if lp.ndim == 1 and lp.shape[0] != co.shape[0]:
        raise ValueError("coeffs.shape[0] != log_poly.shape[0]")
    lp = _c(np.broadcast_to(lp.astype(bool), (co.shape[0],)), np.uint8)
    out = np.empty((co.shape[0], fr.shape[0]), dtype=np.float64)
    rc = _lib().orc_spectra_f64(_p(I_), _p(co), _p(lp), _p(rf), _p(fr), _i64(co.shape[0]), _i64(co.shape[1]),
                                _i64(fr.shape[0]), _p(out))
    assert rc == 0
    return out


def wsclean_predict(uvw, lm, source_type, flux, coeffs, log_poly, ref_freq, gauss_shape, frequency):
    """africanus/rime/wsclean_predict.py:11-120."""
    st = np.asarray(source_type)
    if not np.all((st == "POINT") | (st == "GAUSSIAN")):
        raise ValueError("source_type must be POINT or GAUSSIAN")
    spec = _c(spectra(flux, coeffs, log_poly, ref_freq, frequency), np.float64)
    uvw_, lm_, gs_, fr_ = (_c(a, np.float64) for a in (uvw, lm, gauss_shape, frequency))
    isg = _c(st == "GAUSSIAN", np.uint8)
    nsrc, nrow, nchan = lm_.shape[0], uvw_.shape[0], fr_.shape[0]
    out = np.empty((nrow, nchan, 1), dtype=np.complex128)
    rc = _lib().orc_wsclean_predict_f64(_p(uvw_), _p(lm_), _p(isg), _p(gs_), _p(fr_), _p(spec), _i64(nsrc),
                                        _i64(nrow), _i64(nchan), _p(out))
    assert rc == 0
    return out


_BASES = {0: 0, 1: 1, 2: 2, "std": 0, "log": 1, "log10": 2}


def spectral_model(stokes, spi, ref_freq, frequency, base=0):
    """africanus/model/spectral/spec_model.py:102-236."""
    stokes, spi = np.asarray(stokes), np.asarray(spi)
    if spi.ndim - 2 != stokes.ndim - 1:
        raise ValueError("Dimensions on stokes and spi don't agree")
    pol_shape = stokes.shape[1:]
    npol = int(np.prod(pol_shape, dtype=np.int64)) if pol_shape else 1
    if npol != (int(np.prod(spi.shape[2:], dtype=np.int64)) if spi.shape[2:] else 1):
        raise ValueError("Correlations on stokes and spi don't agree")
    bl = list(base) if isinstance(base, (list, tuple)) else [base] * npol
    bl = bl + [bl[-1]] * (npol - len(bl))
    try:
        b = _c(np.array([_BASES[x] for x in bl[:npol]]), np.int32)
    except KeyError:
        raise ValueError("Invalid base")
    st, sp = _c(stokes.reshape(stokes.shape[0], npol), np.float64), _c(spi.reshape(spi.shape[0], spi.shape[1], npol), np.float64)
    rf, fr = _c(ref_freq, np.float64), _c(frequency, np.float64)
    out = np.empty((st.shape[0], fr.shape[0], npol), dtype=np.float64)
    rc = _lib().orc_spectral_model_f64(_p(st), _p(sp), _p(rf), _p(fr), _p(b), _i64(st.shape[0]), _i64(sp.shape[1]),
                                       _i64(npol), _i64(fr.shape[0]), _p(out))
    assert rc == 0
    return out.reshape((st.shape[0], fr.shape[0]) + tuple(pol_shape))


def feed_rotation(parallactic_angles, feed_type="linear"):
    """africanus/rime/feeds.py:50-73."""
    if feed_type not in ("linear", "circular"):
        raise ValueError("Invalid feed_type '%s'" % feed_type)
    pa = np.asarray(parallactic_angles)
    if pa.dtype not in (np.float32, np.float64):
        raise ValueError("parallactic_angles has none-floating point type %s" % pa.dtype)
    pa64 = _c(pa, np.float64)
    out = np.empty(pa.shape + (2, 2), dtype=np.complex128)
    rc = _lib().orc_feed_rotation_f64(_p(pa64), _i64(pa64.size), ctypes.c_int(0 if feed_type == "linear" else 1),
                                      _p(out))
    assert rc == 0
    return out if pa.dtype == np.float64 else out.astype(np.complex64)


def gaussian_shape(uvw, frequency, shape_params):
    """africanus/model/shape/gaussian_shape.py:11-62."""
    uvw_, fr_, sp_ = (_c(a, np.float64) for a in (uvw, frequency, shape_params))
    nsrc, nrow, nchan = sp_.shape[0], uvw_.shape[0], fr_.shape[0]
    out = np.empty((nsrc, nrow, nchan), dtype=np.float64)
    rc = _lib().orc_gaussian_shape_f64(_p(uvw_), _p(fr_), _p(sp_), _i64(nsrc), _i64(nrow), _i64(nchan), _p(out))
    assert rc == 0
    return out


# ---- convolutional degridder (africanus/gridding/perleypolyhedron) --------------------------------------
# stokes2corr policies as per-correlation factors (policies/stokes_conversion_policies.py:8-137)
STOKES2CORR = {
    "XXYY_FROM_I": [1, 1], "XXXYYXYY_FROM_I": [1, 0, 0, 1], "RRLL_FROM_I": [1, 1], "RRRLLRLL_FROM_I": [1, 0, 0, 1],
    "XXYY_FROM_Q": [1, -1], "XXXYYXYY_FROM_Q": [1, 0, 0, -1], "RLLR_FROM_Q": [1, 1], "RRRLLRLL_FROM_Q": [0, 1, 1, 0],
    "XYYX_FROM_U": [1, 1], "XXXYYXYY_FROM_U": [0, 1, 1, 0], "RLLR_FROM_U": [1j, -1j],
    "RRRLLRLL_FROM_U": [0, 1j, -1j, 0], "XYYX_FROM_V": [1j, -1j], "XXXYYXYY_FROM_V": [0, 1j, -1j, 0],
    "RRLL_FROM_V": [1, -1], "RRRLLRLL_FROM_V": [1, 0, 0, -1],
}


def degridder(uvw, gridstack, wavelengths, chanmap, cell, image_centre, phase_centre, convolution_kernel,
              convolution_kernel_width, convolution_kernel_oversampling, baseline_transform_policy,
              phase_transform_policy, stokes_conversion_policy, convolution_policy, vis_dtype=np.complex128):
    """africanus/gridding/perleypolyhedron/degridder.py:79-175 (gather policies; 'None' / 'wlinapprox' baseline
    transforms)."""
    if np.size(chanmap) != np.size(wavelengths):
        raise ValueError("Chanmap and corresponding wavelengths must match in shape")
    if gridstack.shape[1] != gridstack.shape[2]:
        raise ValueError("Grid must be square")
    chanmap = _c(np.ravel(chanmap), np.int64)
    wl = _c(np.ravel(wavelengths), np.float64)
    if gridstack.shape[0] < chanmap.max() + 1:
        raise ValueError("Not enough channel bands in grid stack to match mfs band mapping")
    if uvw.shape[1] != 3:
        raise ValueError("UVW array must be array of tripples")
    bpol = {"None": 0, "wlinapprox": 1}[baseline_transform_policy]
    ppol = {"None": 0, None: 0, "phase_rotate": 1}[phase_transform_policy]
    cpol = {"conv_1d_axisymmetric_packed_gather": 0, "conv_1d_axisymmetric_unpacked_gather": 1}[convolution_policy]
    coef = _c(np.asarray(STOKES2CORR[stokes_conversion_policy], dtype=np.complex128), np.complex128)
    uvw_, g = _c(uvw, np.float64), _c(gridstack, np.complex128)
    k = _c(convolution_kernel, np.float64)
    ic, pc = _c(image_centre, np.float64), _c(phase_centre, np.float64)
    nrow, nchan, npix = uvw_.shape[0], wl.shape[0], g.shape[1]
    out = np.empty((nrow, nchan, coef.shape[0]), dtype=np.complex128)
    rc = _lib().orc_degridder_c128(_p(uvw_), _p(g), _p(wl), _p(chanmap), ctypes.c_double(cell), _p(ic), _p(pc), _p(k),
                                   _i64(convolution_kernel_width), _i64(convolution_kernel_oversampling),
                                   ctypes.c_int(bpol), ctypes.c_int(ppol), _p(coef), ctypes.c_int(coef.shape[0]),
                                   ctypes.c_int(cpol), _i64(nrow), _i64(nchan), _i64(npix), _p(out))
    assert rc == 0
    return out.astype(vis_dtype, copy=False)


# corr2stokes policies as per-correlation factors (policies/stokes_conversion_policies.py:143-180)
CORR2STOKES = {
    "I_FROM_XXYY": [.5, .5], "I_FROM_XXXYYXYY": [.5, 0, 0, .5], "I_FROM_RRLL": [.5, .5], "I_FROM_RRRLLRLL": [.5, 0, 0, .5],
    "Q_FROM_XXYY": [.5, -.5], "Q_FROM_XXXYYXYY": [.5, 0, 0, -.5], "Q_FROM_RRRLLRLL": [0, .5, .5, 0],
    "U_FROM_XYYX": [.5, .5], "U_FROM_XXXYYXYY": [0, .5, .5, 0], "U_FROM_RLLR": [-.5j, .5j],
    "U_FROM_RRRLLRLL": [0, -.5j, .5j, 0], "V_FROM_RRLL": [.5, -.5], "V_FROM_RRRLLRLL": [.5, 0, 0, -.5],
    "V_FROM_XYYX": [-.5j, .5j], "V_FROM_XXXYYXYY": [0, -.5j, .5j, 0],
}


def gridder(uvw, vis, wavelengths, chanmap, npix, cell, image_centre, phase_centre, convolution_kernel,
            convolution_kernel_width, convolution_kernel_oversampling, baseline_transform_policy,
            phase_transform_policy, stokes_conversion_policy, convolution_policy, grid_dtype=np.complex128,
            do_normalize=False):
    """africanus/gridding/perleypolyhedron/gridder.py:12-117 (scatter policies; 'None' baseline transform)."""
    if np.size(chanmap) != np.size(wavelengths):
        raise ValueError("Chanmap and corresponding wavelengths must match in shape")
    chanmap = _c(np.ravel(chanmap), np.int64)
    wl = _c(np.ravel(wavelengths), np.float64)
    if uvw.shape[1] != 3:
        raise ValueError("UVW array must be array of tripples")
    if uvw.shape[0] != vis.shape[0]:
        raise ValueError("UVW array must have same number of rows as vis array")
    if vis.shape[1] != wl.size:
        raise ValueError("Chanmap must correspond to visibility channels")
    assert baseline_transform_policy == "None"
    ppol = {"None": 0, None: 0, "phase_rotate": 1}[phase_transform_policy]
    cpol = {"conv_1d_axisymmetric_unpacked_scatter": 0, "conv_1d_axisymmetric_packed_scatter": 1,
            "conv_nn_scatter": 2}[convolution_policy]
    coef = _c(np.asarray(CORR2STOKES[stokes_conversion_policy], dtype=np.complex128), np.complex128)
    if coef.shape[0] != vis.shape[2]:
        raise ValueError("stokes_conversion_policy does not fit the correlations of vis")
    nband = int(chanmap.max()) + 1
    uvw_, vs = _c(uvw, np.float64), _c(vis, np.complex128)
    k = _c(convolution_kernel, np.float64)
    ic, pc = _c(image_centre, np.float64), _c(phase_centre, np.float64)
    out = np.empty((nband, npix, npix), dtype=np.complex128)
    wt = np.empty(nband, dtype=np.float64)
    rc = _lib().orc_gridder_c128(_p(uvw_), _p(vs), _p(wl), _p(chanmap), _i64(npix), ctypes.c_double(cell), _p(ic), _p(pc),
                                 _p(k), _i64(convolution_kernel_width), _i64(convolution_kernel_oversampling),
                                 ctypes.c_int(ppol), _p(coef), ctypes.c_int(coef.shape[0]), ctypes.c_int(cpol),
                                 ctypes.c_int(int(bool(do_normalize))), _i64(uvw_.shape[0]), _i64(wl.shape[0]),
                                 _i64(nband), _p(out), _p(wt))
    assert rc == 0
    return out.astype(grid_dtype, copy=False)


# ---- calibration consumers (africanus/calibration/utils) ------------------------------------------------
def _calib_mode(jones, vis, vis_type):
    """africanus/calibration/utils/utils.py:11-45 (check_type)."""
    vis_ndim = (3, 4) if vis_type == "vis" else (4, 5)
    if vis.ndim == vis_ndim[0]:
        if jones.ndim != 5:
            raise RuntimeError("Jones axes not compatible with visibility axes. Expected length 5 but got "
                               "length %d" % jones.ndim)
        return 0
    if vis.ndim == vis_ndim[1]:
        if jones.ndim == 5:
            return 1
        if jones.ndim == 6:
            return 2
        raise RuntimeError("Jones term has incorrect shape")
    raise RuntimeError("Visibility data has incorrect shape")


def _calib_common(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, arrays):
    for a in (jones,) + tuple(arrays):
        if a.shape[-1] > 2:
            raise ValueError("ncorr cant be larger than 2")
    tbi = _c(time_bin_indices, np.int64)
    tbi = tbi - tbi.min() if tbi.size else tbi
    return tbi, _c(time_bin_counts, np.int64), _c(antenna1, np.int64), _c(antenna2, np.int64)


def corrupt_vis(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, model):
    """africanus/calibration/utils/corrupt_vis.py:58-101."""
    mode = _calib_mode(jones, model, "model")
    tbi, tbc, a1, a2 = _calib_common(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, (model,))
    jn, md = _c(jones, np.complex128), _c(model, np.complex128)
    nrow, nchan, ndir = md.shape[:3]
    out = np.empty(md.shape[:2] + md.shape[3:], dtype=np.complex128)
    rc = _lib().orc_corrupt_vis_c128(_p(tbi), _p(tbc), _i64(tbi.shape[0]), _p(a1), _p(a2), _p(jn), _p(md), _i64(nrow),
                                     _i64(jn.shape[1]), _i64(nchan), _i64(ndir), ctypes.c_int(mode),
                                     ctypes.c_int(md.shape[-1]), _p(out))
    assert rc == 0
    return out


def compute_and_corrupt_vis(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, model, uvw, freq, lm):
    """africanus/calibration/utils/compute_and_corrupt_vis.py:73-152."""
    mode = _calib_mode(jones, model, "model")
    tbi, tbc, a1, a2 = _calib_common(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, (model,))
    jn, md = _c(jones, np.complex128), _c(model, np.complex128)
    uvw_, fr, lm_ = _c(uvw, np.float64), _c(freq, np.float64), _c(lm, np.float64)
    nrow, nchan, ndir = uvw_.shape[0], md.shape[1], md.shape[2]
    out = np.empty((nrow, fr.shape[0]) + md.shape[3:], dtype=np.complex128)
    rc = _lib().orc_compute_and_corrupt_vis_c128(_p(tbi), _p(tbc), _i64(tbi.shape[0]), _p(a1), _p(a2), _p(jn), _p(md),
                                                 _p(uvw_), _p(fr), _p(lm_), _i64(nrow), _i64(jn.shape[1]), _i64(nchan),
                                                 _i64(ndir), ctypes.c_int(mode), ctypes.c_int(md.shape[-1]), _p(out))
    assert rc == 0
    return out


def residual_vis(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, vis, flag, model):
    """africanus/calibration/utils/residual_vis.py:63-119."""
    mode = _calib_mode(jones, vis, "vis")
    tbi, tbc, a1, a2 = _calib_common(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, (vis, model))
    jn, md, vs = _c(jones, np.complex128), _c(model, np.complex128), _c(vis, np.complex128)
    fl = _c(np.asarray(flag) != 0, np.uint8)
    nrow, nchan, ndir = md.shape[:3]
    out = np.empty(vs.shape, dtype=np.complex128)
    rc = _lib().orc_residual_vis_c128(_p(tbi), _p(tbc), _i64(tbi.shape[0]), _p(a1), _p(a2), _p(jn), _p(vs), _p(fl),
                                      _p(md), _i64(nrow), _i64(jn.shape[1]), _i64(nchan), _i64(ndir),
                                      ctypes.c_int(mode), ctypes.c_int(vs.shape[-1]), _p(out))
    assert rc == 0
    return out


def correct_vis(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, vis, flag):
    """africanus/calibration/utils/correct_vis.py:63-115."""
    mode = _calib_mode(jones, vis, "vis")
    tbi, tbc, a1, a2 = _calib_common(time_bin_indices, time_bin_counts, antenna1, antenna2, jones, (vis,))
    if jones.shape[3] > 1:
        raise ValueError("Jones has n_dir > 1. Cannot correct for direction dependent gains")
    jn, vs = _c(jones, np.complex128), _c(vis, np.complex128)
    fl = _c(np.asarray(flag) != 0, np.uint8)
    nrow, nchan = vs.shape[:2]
    out = np.empty(vs.shape, dtype=np.complex128)
    rc = _lib().orc_correct_vis_c128(_p(tbi), _p(tbc), _i64(jn.shape[0]), _p(a1), _p(a2), _p(jn), _p(vs), _p(fl),
                                     _i64(nrow), _i64(jn.shape[1]), _i64(nchan), ctypes.c_int(mode),
                                     ctypes.c_int(vs.shape[-1]), _p(out))
    assert rc == 0
    return out


def predict_vis(time_index, antenna1, antenna2, dde1_jones=None, source_coh=None,
                dde2_jones=None, die1_jones=None, base_vis=None, die2_jones=None):
    """africanus/rime/predict.py:466-619 (checks are NOT restated here: the
    product's host wrapper ports them and is tested against the reference's
    error behaviour separately)."""
    arrays = [dde1_jones, source_coh, dde2_jones, die1_jones, base_vis, die2_jones]
    present = [a for a in arrays if a is not None]
    if not present:
        raise ValueError("No Jones Matrices were supplied")
    out_dtype = np.result_type(*[a.dtype for a in present])
    rt = np.float32 if out_dtype == np.complex64 else np.float64
    ct = np.complex64 if rt == np.float32 else np.complex128
    fn = "orc_predict_vis_f32" if rt == np.float32 else "orc_predict_vis_f64"

    have_ddes = dde1_jones is not None
    have_coh = source_coh is not None
    have_dies = die1_jones is not None
    nrow = time_index.shape[0]
    if have_ddes:
        nsrc, ntime, nant, nchan = dde1_jones.shape[:4]
        corrs = dde1_jones.shape[4:]
    elif have_coh:
        nsrc, _, nchan = source_coh.shape[:3]
        corrs = source_coh.shape[3:]
        ntime = nant = 0
    elif have_dies:
        ntime, nant, nchan = die1_jones.shape[:3]
        corrs = die1_jones.shape[3:]
        nsrc = 0
    else:
        nchan = base_vis.shape[1]
        corrs = base_vis.shape[2:]
        nsrc = ntime = nant = 0
    if have_dies:
        ntime, nant = die1_jones.shape[:2]
    ncorr = int(np.prod(corrs))
    jones_2x2 = int(len(corrs) == 2)

    cs = [None if a is None else _c(a, ct) for a in arrays]
    ti, a1, a2 = (_c(x, np.int64) for x in (time_index, antenna1, antenna2))
    out = np.empty((nrow, nchan) + tuple(corrs), dtype=ct)
    rc = getattr(_lib(), fn)(_p(ti), _p(a1), _p(a2), _i64(nrow), *[_p(c) for c in cs],
                             _i64(nsrc), _i64(ntime), _i64(nant), _i64(nchan),
                             _int(ncorr), _int(jones_2x2), _p(out))
    assert rc == 0
    return out


def apply_gains(time_index, antenna1, antenna2, die1_jones, corrupted_vis, die2_jones):
    """africanus/rime/predict.py:622-649."""
    return predict_vis(time_index, antenna1, antenna2, die1_jones=die1_jones,
                       base_vis=corrupted_vis, die2_jones=die2_jones)


def freq_grid_interp(frequency, beam_freq_map):
    """africanus/rime/fast_beam_cubes.py:10-54."""
    rt = np.float32 if frequency.dtype == np.float32 else np.float64
    fn = "orc_freq_grid_interp_f32" if rt == np.float32 else "orc_freq_grid_interp_f64"
    fr_, fm_ = _c(frequency, rt), _c(beam_freq_map, rt)
    out = np.empty((fr_.shape[0], 3), dtype=rt)
    rc = getattr(_lib(), fn)(_p(fr_), _i64(fr_.shape[0]), _p(fm_), _i64(fm_.shape[0]), _p(out))
    assert rc == 0
    return out


def beam_cube_dde(beam, beam_lm_extents, beam_freq_map, lm, parallactic_angles,
                  point_errors, antenna_scaling, frequency):
    """africanus/rime/fast_beam_cubes.py:57-240."""
    if beam.dtype == np.complex64:
        rt, ct, fn = np.float32, np.complex64, "orc_beam_cube_dde_f32"
    else:
        rt, ct, fn = np.float64, np.complex128, "orc_beam_cube_dde_f64"
    beam_lw, beam_mh, beam_nud = beam.shape[:3]
    corrs = beam.shape[3:]
    ncorr = int(np.prod(corrs, dtype=np.int64)) if corrs else 1
    if beam_lw < 2 or beam_mh < 2 or beam_nud < 2:
        raise ValueError("beam_lw, beam_mh and beam_nud must be >= 2")
    b_, ex_, fm_ = _c(beam, ct), _c(beam_lm_extents, rt), _c(beam_freq_map, rt)
    lm_, pa_ = _c(lm, rt), _c(parallactic_angles, rt)
    pe_, as_, fr_ = _c(point_errors, rt), _c(antenna_scaling, rt), _c(frequency, rt)
    nsrc = lm_.shape[0]
    ntime, nant = pa_.shape
    nchan = fr_.shape[0]
    out = np.empty((nsrc, ntime, nant, nchan) + tuple(corrs), dtype=ct)
    rc = getattr(_lib(), fn)(_p(b_), _i64(beam_lw), _i64(beam_mh), _i64(beam_nud), _int(ncorr),
                             _p(ex_), _p(fm_), _p(lm_), _i64(nsrc), _p(pa_), _i64(ntime),
                             _i64(nant), _p(pe_), _p(as_), _p(fr_), _i64(nchan), _p(out))
    assert rc == 0
    return out


# ---- Stokes <-> correlation conversion (africanus/model/coherency/conversion.py) ---------------------------
# the products of conversion.py:18-48, evaluated by numpy's own (complex) add / multiply / divide loops so that
# the dtype promotion and the propagation of non-finite values are the reference's
_PRODUCT_FN = {
    "sum": lambda a, b: a + b + 0j, "diff": lambda a, b: a - b + 0j,
    "sum_j": lambda a, b: a + b * 1j, "diff_j": lambda a, b: a - b * 1j,
    "half_sum": lambda a, b: (a + b) / 2, "half_diff": lambda a, b: (a - b) / 2,
    "half_diff_over_j": lambda a, b: (a - b) / 2j,
}
CONVERT_PRODUCTS = {
    "RR": [("I", "V", "sum")], "LL": [("I", "V", "diff")], "RL": [("Q", "U", "sum_j")], "LR": [("Q", "U", "diff_j")],
    "XX": [("I", "Q", "sum")], "YY": [("I", "Q", "diff")], "XY": [("U", "V", "sum_j")], "YX": [("U", "V", "diff_j")],
    "I": [("XX", "YY", "half_sum"), ("RR", "LL", "half_sum")],
    "Q": [("XX", "YY", "half_diff"), ("RL", "LR", "half_sum")],
    "U": [("XY", "YX", "half_sum"), ("RL", "LR", "half_diff_over_j")],
    "V": [("XY", "YX", "half_diff_over_j"), ("RR", "LL", "half_diff")],
}
_CASA_STOKES = ("Undefined I Q U V RR RL LR LL XX XY YX YY").split()


def convert(input, input_schema, output_schema, implicit_stokes=False):
    """africanus/model/coherency/conversion.py:143-216 for well-formed schemas (names or casacore ids, nested
    lists): first candidate pair whose operands are all present (with implicit_stokes a missing Stokes operand of a
    correlation counts as 0); result real only if every output is a real combination of real input."""
    def names(schema):
        arr = np.asarray(schema, dtype=object)
        flat = [x if isinstance(x, str) else _CASA_STOKES[int(x)] for x in arr.ravel()]
        return flat, arr.shape
    inames, ishape = names(input_schema)
    onames, oshape = names(output_schema)
    x = np.asarray(input)
    if x.dtype.kind in "biu":
        x = x.astype(np.float64)
    lead = x.shape[:x.ndim - len(ishape)]
    x = x.reshape(lead + (len(inames),))
    cols = []
    for name in onames:
        for n1, n2, product in CONVERT_PRODUCTS[name]:
            corr = name not in "IQUV"
            have1, have2 = n1 in inames, n2 in inames
            if (have1 or (corr and implicit_stokes)) and (have2 or (corr and implicit_stokes)):
                a = x[..., inames.index(n1)] if have1 else x.dtype.type(0)
                b = x[..., inames.index(n2)] if have2 else x.dtype.type(0)
                cols.append(np.broadcast_to(_PRODUCT_FN[product](a, b), lead))
                break
        else:
            raise KeyError(name)
    dtype = np.result_type(*[c.dtype for c in cols])
    return np.stack([c.astype(dtype) for c in cols], axis=-1).reshape(lead + oshape)
