"""``model`` (image -> visibilities) with the signature of africanus/gridding/wgridder/im2vis.py:63-99."""
import math

import numpy as np

from ... import _lib
from ..._device import Call, _is_torch, np_dtype_of

import contextlib
import threading

LIGHTSPEED = 2.99792458e8
_NQUAD = 48
PLANE_BUDGET = 48 << 30      # bytes of w-plane grids kept resident per call (288 GB of HBM per GPU)
PLANES_F64, PLANES_F32 = 0, 1                                  # include/afhip.h: AF_WGRID_PLANES_*
_planes = threading.local()


@contextlib.contextmanager
def plane_precision(mode):
    """``with plane_precision("single"):`` -- ``model`` calls of this thread keep their w-planes in float32 where the
    requested accuracy allows (epsilon >= 1e-5): half the bytes of every pass of the plane transforms (configs[4]: 39 ->
    27 ms, l2 error 1.2905e-6 -> 1.2973e-6).  Not the default for float64 images: <R x, y> = <x, R^H y> then holds to
    ~1e-7 instead of 1e-12 (the reference's adjointness test pins 1e-12 for double precision,
    africanus/gridding/wgridder/tests/test_wgridder.py:125-188); float32 images always take float32 planes, as the
    reference's single-precision calls take float grids in ducc0.  ``"double"`` restores the default."""
    if mode not in ("single", "double"):
        raise ValueError("plane_precision is 'single' or 'double'")
    prev = getattr(_planes, "mode", PLANES_F64)
    _planes.mode = PLANES_F32 if mode == "single" else PLANES_F64
    try:
        yield
    finally:
        _planes.mode = prev


def kernel_parameters(epsilon):
    """Taps per axis W and shape beta of the exponential-of-semicircle kernel exp(beta (sqrt(1 - (2t/W)^2) - 1)) for a
    requested accuracy at an oversampling of 2 (Barnett, Magland & af Klinteberg 2019: error ~ 10^(1-W) per axis with
    beta = 2.3 W; one more tap for the three axes u, v, w)."""
    if not (epsilon > 0):
        raise ValueError("epsilon must be positive")
    W = int(math.ceil(math.log10(1.0 / min(epsilon, 0.1)))) + 2
    if W > 16:
        # 16 taps reach ~1e-14, the floor of a float64 transform of this size: a smaller epsilon cannot be met, and
        # clamping silently would break the documented contract (ducc0 refuses such accuracies too)
        raise ValueError("epsilon = %g is below what the float64 kernel can reach (>= 1e-14)" % epsilon)
    W = max(4, W)
    return W, 2.30 * W


def _quadrature():
    x, w = np.polynomial.legendre.leggauss(_NQUAD)
    return 0.5 * (x + 1.0), 0.5 * w          # nodes / weights on (0, 1)


def kernel_correction(n, n_padded, W, beta):
    """1 / psihat(xi) at xi = (i - n // 2) / n_padded, psihat(xi) = W int_0^1 exp(beta (sqrt(1 - t^2) - 1))
    cos(pi W xi t) dt: the Fourier transform of the kernel sampled where the image's pixels sit."""
    t, w = _quadrature()
    xi = (np.arange(n) - n // 2) / float(n_padded)
    phi = np.exp(beta * (np.sqrt(1.0 - t * t) - 1.0))
    psihat = W * (np.cos(np.pi * W * xi[:, None] * t[None, :]) * (w * phi)[None, :]).sum(axis=1)
    return 1.0 / psihat


def _bins(freq_bin_idx, freq_bin_counts, nband, nchan):
    fbi = np.asarray(freq_bin_idx.cpu() if _is_torch(freq_bin_idx) else freq_bin_idx).astype(np.int64)
    fbc = np.asarray(freq_bin_counts.cpu() if _is_torch(freq_bin_counts) else freq_bin_counts).astype(np.int64)
    if nband is None:
        nband = int(fbi.size)
    if fbi.shape != (nband,) or fbc.shape != (nband,):
        raise ValueError("freq_bin_idx and freq_bin_counts must have one entry per band")
    fbi = fbi - fbi.min() if nband else fbi
    if nband and (fbi.min() < 0 or (fbi + fbc).max() > nchan or fbc.min() < 0):
        raise ValueError("frequency bins exceed the channel axis")
    return fbi, fbc


def _operator(adjoint, uvw, freq, image, vis, freq_bin_idx, freq_bin_counts, nx, ny, cell, weights, flag, celly,
              epsilon, do_wstacking, out_dtype, w_bounds=None):
    """Both directions of the wgridder operator, band by band (csrc/af_wgridder.hip): ``adjoint`` False: image (band, nx,
    ny) -> visibilities (row, chan); True: visibilities -> image.  Geometry, planes and taps are the same for both, so
    the two are exact transposes of each other."""
    import ctypes
    if celly is None:
        celly = cell
    nrow, nchan = int(uvw.shape[0]), int(freq.shape[0])
    if len(uvw.shape) != 2 or uvw.shape[1] != 3:
        raise ValueError("uvw must have shape (row, 3)")
    nband = None if adjoint else int(image.shape[0])
    fbi, fbc = _bins(freq_bin_idx, freq_bin_counts, nband, nchan)
    nband = int(fbi.size)
    for name, a in (("weights", weights), ("flag", flag), ("vis", vis)):
        if a is not None and tuple(int(s) for s in a.shape) != (nrow, nchan):
            raise ValueError("%s must have shape (row, chan)" % name)
    W, beta = kernel_parameters(float(epsilon))
    if nx % 2 or ny % 2:
        # ducc0 requires even image sizes (the phase centre is the pixel nx/2, ny/2); so does this entry
        raise ValueError("image dimensions must be even (got %d x %d)" % (nx, ny))
    lib = _lib.load()
    nu, nv = int(lib.af_wgrid_padded(nx)), int(lib.af_wgrid_padded(ny))
    corr_u, corr_v = kernel_correction(nx, nu, W, beta), kernel_correction(ny, nv, W, beta)
    qt, qw = _quadrature()
    eps_max = (nx / 2.0 * cell) ** 2 + (ny / 2.0 * celly) ** 2
    if do_wstacking and eps_max >= 1.0:
        raise ValueError("the image extends beyond the horizon (l^2 + m^2 >= 1)")
    max_nm1 = eps_max / (math.sqrt(1.0 - eps_max) + 1.0) if do_wstacking else 0.0
    # range of w nu / c per band (host scalars: they size the w-plane loop)
    # The number of w-planes sizes the workspace and the plane loop, so the range of w is needed on the HOST.  numpy
    # inputs: free.  Device-resident uvw: reading min / max back waits for the tensor's producer (one host
    # synchronisation per call) -- callers that want the call fully asynchronous pass ``w_bounds=(wmin, wmax)`` in
    # metres (any superset of the true range gives the same accuracy; the planes cover the stated range).
    if w_bounds is not None:
        wmin, wmax = float(w_bounds[0]), float(w_bounds[1])
        if not (wmin <= wmax):
            raise ValueError("w_bounds must be (wmin, wmax) with wmin <= wmax")
    elif nrow:
        wcol = uvw[:, 2]
        wmin, wmax = (float(wcol.min()), float(wcol.max()))
    else:
        wmin = wmax = 0.0
    fhost = np.asarray(freq.cpu() if _is_torch(freq) else freq, dtype=np.float64)
    with Call(uvw, freq, image, vis, weights, flag) as c:
        p_uvw, p_fr = c.inp(uvw, np.float64), c.inp(freq, np.float64)
        p_wgt = c.inp(weights, np.float64)
        if flag is not None:
            flag = (flag != 0)
        p_mask = c.inp(flag, np.uint8 if not _is_torch(flag) else np.bool_)
        p_cu, p_cv, p_qt, p_qw = (c.inp(a, np.float64) for a in (corr_u, corr_v, qt, qw))
        if adjoint:
            p_in = c.inp(vis, np.complex128)
            p_out, h = c.out((nband, nx, ny), np.float64)
            entry = "af_wgrid_vis2im_f64"
        else:
            p_in = c.inp(image, np.float64)
            p_out, h = c.out((nrow, nchan), np.complex128)
            entry = "af_wgrid_im2vis_f64"
            if nrow * nchan:
                _lib.call("af_memset", p_out, 0, nrow * nchan * 16, c.stream)     # channels outside every band stay 0
        # per band: the range of w nu / c; the workspace holds as many w-plane grids as the largest band needs, within
        # PLANE_BUDGET bytes (beyond it the planes are worked through in batches)
        bands = []
        for b in range(nband):
            c0, nc = int(fbi[b]), int(fbc[b])
            if nc == 0 or nrow == 0:
                if adjoint:
                    _lib.call("af_memset", ctypes.c_void_p(p_out.value + 8 * b * nx * ny), 0, nx * ny * 8, c.stream)
                continue
            f = fhost[c0:c0 + nc] / LIGHTSPEED
            cands = (wmin * f.min(), wmin * f.max(), wmax * f.min(), wmax * f.max())
            npl = int(lib.af_wgrid_planes(float(min(cands)), float(max(cands)), float(max_nm1), W, int(bool(do_wstacking))))
            if npl < 1:
                raise ValueError("w range of band %d is not finite or needs more than 1e6 w-planes" % b)
            bands.append((b, c0, nc, cands, npl))
        want = max([x[4] for x in bands] + [1])
        resident = max(1, min(want, PLANE_BUDGET // (nu * nv * 16)))
        ws_bytes = int(lib.af_wgrid_workspace_bytes(nx, ny, resident, nrow, max([x[2] for x in bands] + [1]), want, W))
        p_ws = c.scratch(ws_bytes)
        # float32 planes: the reference's single-precision call (float32 image), or the caller's plane_precision("single")
        single = (not adjoint) and (np_dtype_of(image) == np.float32 or getattr(_planes, "mode", PLANES_F64) == PLANES_F32)
        prev = lib.af_wgrid_plane_precision(PLANES_F32 if single else PLANES_F64)        # per thread
        try:
            for b, c0, nc, cands, npl in bands:
                img = ctypes.c_void_p((p_out if adjoint else p_in).value + 8 * b * nx * ny)
                _lib.call(entry, p_uvw, ctypes.c_void_p(p_fr.value + 8 * c0), nrow, nc, c0, nchan,
                          p_in if adjoint else img, nx, ny, float(cell), float(celly), p_cu, p_cv, p_qt,
                          p_qw, W, beta, float(min(cands)), float(max(cands)), float(max_nm1), int(bool(do_wstacking)),
                          p_wgt, p_mask, img if adjoint else p_out, p_ws, max(ws_bytes, 256), c.stream)
        finally:
            lib.af_wgrid_plane_precision(prev)
        native = np.float64 if adjoint else np.complex128
        return c.result(h, cast=None if out_dtype == native else out_dtype)


def model(uvw, freq, image, freq_bin_idx, freq_bin_counts, cell, weights=None, flag=None, celly=None, epsilon=1e-5,
          nthreads=1, do_wstacking=True, w_bounds=None):
    """
    ``V = R x``: visibilities (row, chan) of the model image ``x`` (band, nx, ny), channels ``freq_bin_idx[b] ..
    + freq_bin_counts[b]`` taken from band ``b`` (bin starts are normalised by their minimum, as the reference does
    for row chunks); ``cell`` / ``celly`` pixel sizes in radians; ``weights`` (row, chan) multiply the result
    (whitened model); ``flag`` (row, chan): only visibilities with ``flag != 0`` are computed, the rest are 0;
    ``epsilon``: accuracy with respect to the direct Fourier transform; ``do_wstacking`` False ignores w and n.
    ``nthreads`` is accepted and ignored (the work runs on the GPU).  ``w_bounds=(wmin, wmax)`` (metres; an extension,
    keyword only in spirit) spares a device-resident call the host read-back of the w range -- with it, and ``freq``
    given as a numpy array, nothing in the call waits for the device.

    Same contract as ``africanus.gridding.wgridder.model`` (africanus/gridding/wgridder/im2vis.py:63-99).  The
    reference delegates the arithmetic to ``ducc0.wgridder.dirty2ms`` (not vendored, not installed here: parity
    unpinned); what its tests pin, and what holds here, is the accuracy contract of
    africanus/gridding/wgridder/tests/test_wgridder.py:18-113: relative l2 error <= ``epsilon`` against
    ``sum_xy x[x,y]/n exp(-2 pi i nu/c (u x + v y - w (n - 1)))``.  Algorithm: improved w-stacking (a separable
    exponential-of-semicircle kernel in u, v and w; one zero-padded FFT per w-plane), csrc/af_wgridder.hip.
    """
    if len(image.shape) != 3:
        raise ValueError("image must have shape (band, nx, ny)")
    nx, ny = int(image.shape[1]), int(image.shape[2])
    out_dtype = np.result_type(np_dtype_of(image), np.complex64)
    return _operator(False, uvw, freq, image, None, freq_bin_idx, freq_bin_counts, nx, ny, cell, weights, flag, celly,
                     epsilon, do_wstacking, out_dtype, w_bounds)
