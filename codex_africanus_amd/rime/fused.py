"""
Fused RIME predict: the reference's chain

    phase  = phase_delay(lm, uvw, frequency)                              africanus/rime/phase.py:11
    coh    = einsum("srf,sfij->srfij", phase, brightness)                 africanus/rime/examples/predict.py:107-134
    ddes   = beam_cube_dde(beam, ..., lm, parangles, point_errors, ...)   africanus/rime/fast_beam_cubes.py:57
    vis    = predict_vis(time_index, a1, a2, ddes, coh, ddes, die1, base_vis, die2)   africanus/rime/predict.py:466

evaluated on the device without materialising ``coh`` (4.1 TB at 1e6 rows x 64 chan x 1000 src)
or ``ddes`` (130 GB with 64 antennas): the form in which BASELINE configs 2-4 are feasible at all.
The reference's own counterpart is the experimental fused RIME
(africanus/experimental/rime/fused/core.py:88-120).
"""
import collections
import ctypes
import hashlib
import os
import threading

import numpy as np

from .. import _lib
from .._device import Call, _is_torch, np_dtype_of
from .predict import predict_vis


def _host(a):
    return a.detach().cpu().numpy() if _is_torch(a) else np.asarray(a)


class FusedPlan(object):
    """Row layout of a fused predict with DDEs: how the rows of every timestep are dealt to the lanes of a workgroup.
    Built by :func:`fused_plan`; holds host copies of the plan arrays and uploads them once per device."""

    def __init__(self, nrow, nant, nsteps, items, groups, antenna1, antenna2, step):
        self.nrow, self.nant, self.nsteps = nrow, nant, nsteps
        self.items, self.groups = items, groups
        self.antenna1, self.antenna2 = antenna1, antenna2
        self.step = step            # (row,) int32: time_index - min(time_index) -- what af_fused_plan_check compares
        # rows in time order?  then step_first[t] = first row of step t (and nrow at the end): where a result may be cut
        self.time_sorted = bool(nrow) and bool(np.all(np.diff(step) >= 0))
        self.step_first = np.searchsorted(step, np.arange(nsteps + 1)) if self.time_sorted else None
        self.n_items = int(items.shape[0]) if nrow else 0
        # antenna decomposition of uvw (fused_plan(..., uvw=...)): per-antenna coordinates (nsteps, nant, 3), the row of
        # every (step, antenna1, antenna2) (nsteps, nap, nap) and the largest |x_p - x_q - uvw_pq| [m]; None = the rows
        # are not (known to be) antenna-decomposable and the call runs on the lane-per-row kernel
        self.ant_uvw = self.rowmap = self.residual = None
        self.tol = DECOMPOSE_TOL
        self.single_tol = False     # the tolerance is single precision's (fused_plan(..., single=True) on float32 rows)
        # rows over the baseline slots the GEMM form evaluates (nsteps x upper block triangle of 8-antenna blocks): the
        # GEMM form pays for every slot, the lane-per-row kernel for every row (fused_predict_vis picks by this)
        self.fill = 1.0
        self._dev = {}

    @property
    def decomposable(self):
        return self.ant_uvw is not None

    def device(self, arr, call):
        """`arr` (one of this plan's arrays) for the call: the numpy array in host mode, a cached tensor in device mode"""
        if not call.device_mode:
            return arr
        import torch
        key = (id(arr), str(call.torch_device))
        t = self._dev.get(key)
        if t is None:
            t = self._dev[key] = torch.from_numpy(arr).to(call.torch_device)
        return t


DECOMPOSE_TOL = 1e-10      # metres: see fused_plan
GEMM_MAX_ANTENNAS = 512    # af_fused_predict_antennas_c128: 64 blocks of 8 antennas in <= 8 super-blocks (SKA-low: 512 stations)


def fused_plan(time_index, antenna1, antenna2, nant, grouped=True, uvw=None, decompose_tol=None, single=False):
    """
    Plan of a row layout for :func:`fused_predict_vis` with DDEs (host side, O(row)): runs of consecutive rows with
    equal ``time_index`` become workgroup items.  ``grouped`` (default): rows are dealt in 2 x 2 blocks of baselines that
    share their antennas' Jones terms (``af_fused_plan_groups``: 5 instead of 8 LDS reads per (row, source); up to ~230
    antennas); otherwise plain row ranges (``af_fused_plan_rows``).  ``nant`` = the antenna extent
    of the per-antenna arrays (parallactic_angles.shape[1]).

    ``uvw`` (row, 3), optional: the plan then also tests whether the rows are ANTENNA-DECOMPOSABLE,
    ``uvw_pq = uvw_p - uvw_q`` per timestep -- true of every real Measurement Set, not of uvw drawn at random per row --
    by solving for per-antenna coordinates (``af_fused_plan_antennas``) and accepting them when
    ``max |uvw_p - uvw_q - uvw_pq| <= decompose_tol`` metres (default 1e-10: a phase error below
    ``2 pi nu / c |lmn| 1e-10`` = 2e-10 rad at 1.7 GHz, 0.05 rad off axis).  A decomposable plan (``plan.decomposable``;
    at most 512 antennas, every (time, antenna1, antenna2) at most once) sends the call to the GEMM form of the predict,
    ``V(t, nu) = G H^H`` on the matrix cores (csrc/af_fused_gemm.hip: about half the arithmetic of the general kernel);
    such a plan is bound to these ``uvw``.  ``AFHIP_FUSED_GEMM=0`` switches the test off.

    ``single`` (with float32 ``uvw``; :func:`fused_predict_vis` sets it when EVERY input of the call is single precision):
    the test is made at the rows' own precision, ``decompose_tol = 2^-22 max |uvw|`` -- float32 differences of antenna
    coordinates carry 2^-24 of rounding each, which no antenna coordinates reproduce -- and such a plan serves the
    single-precision GEMM form (``af_fused_predict_antennas_c64``).
    """
    ti = np.ascontiguousarray(_host(time_index), dtype=np.int64)
    a1h = np.ascontiguousarray(_host(antenna1), dtype=np.int32)
    a2h = np.ascontiguousarray(_host(antenna2), dtype=np.int32)
    nrow, nant = int(ti.shape[0]), int(nant)
    if a1h.shape != (nrow,) or a2h.shape != (nrow,):
        raise ValueError("time_index, antenna1 and antenna2 must have shape (row,)")
    if nrow and (min(a1h.min(), a2h.min()) < 0 or max(a1h.max(), a2h.max()) >= nant):
        raise ValueError("antenna index out of range")
    nsteps = int(ti.max()) - int(ti.min()) + 1 if nrow else 0
    grouped = grouped and nant <= 230 and os.environ.get("AFHIP_FUSED_WS", "1") != "0" and \
        os.environ.get("AFHIP_FUSED_GROUPS", "1") != "0"
    n_items = ctypes.c_int64(0)
    tip = ti.ctypes.data_as(ctypes.c_void_p)
    groups = None
    if grouped:
        n_groups = ctypes.c_int64(0)
        pa1, pa2 = a1h.ctypes.data_as(ctypes.c_void_p), a2h.ctypes.data_as(ctypes.c_void_p)
        _lib.call("af_fused_plan_groups", tip, pa1, pa2, nrow, nant, None, 0, ctypes.byref(n_items), None, 0,
                  ctypes.byref(n_groups))
        items = np.zeros((max(n_items.value, 1), 4), dtype=np.int32)
        groups = np.zeros((max(n_groups.value, 1), 8), dtype=np.int32)
        _lib.call("af_fused_plan_groups", tip, pa1, pa2, nrow, nant, items.ctypes.data_as(ctypes.c_void_p),
                  n_items.value, ctypes.byref(n_items), groups.ctypes.data_as(ctypes.c_void_p), n_groups.value,
                  ctypes.byref(n_groups))
    else:
        _lib.call("af_fused_plan_rows", tip, nrow, None, 0, ctypes.byref(n_items))
        items = np.zeros((max(n_items.value, 1), 4), dtype=np.int32)
        _lib.call("af_fused_plan_rows", tip, nrow, items.ctypes.data_as(ctypes.c_void_p), n_items.value,
                  ctypes.byref(n_items))
    step = (ti - (ti.min() if nrow else 0)).astype(np.int32)
    plan = FusedPlan(nrow, nant, nsteps, items, groups, a1h, a2h, step)
    if uvw is not None and nrow and nant <= GEMM_MAX_ANTENNAS and os.environ.get("AFHIP_FUSED_GEMM", "1") != "0":
        uvw_h = np.ascontiguousarray(_host(uvw), dtype=np.float64)
        if uvw_h.shape != (nrow, 3):
            raise ValueError("uvw must have shape (row, 3)")
        nap = 8 * ((nant + 7) // 8)
        ant_uvw = np.zeros((nsteps, nant, 3), dtype=np.float64)
        rowmap = np.zeros((nsteps, nap, nap), dtype=np.int32)
        resid, ok = ctypes.c_double(0.0), ctypes.c_int(0)
        if decompose_tol is not None:
            tol = float(decompose_tol)
        elif single and np_dtype_of(uvw) == np.float32 and nrow:
            tol = float(np.abs(uvw_h).max()) * 2.0 ** -22
        else:
            tol = DECOMPOSE_TOL
        _lib.call("af_fused_plan_antennas", tip, a1h.ctypes.data_as(ctypes.c_void_p), a2h.ctypes.data_as(ctypes.c_void_p),
                  uvw_h.ctypes.data_as(ctypes.c_void_p), nrow, nant, tol, nsteps, ant_uvw.ctypes.data_as(ctypes.c_void_p),
                  rowmap.ctypes.data_as(ctypes.c_void_p), ctypes.byref(resid), ctypes.byref(ok))
        plan.residual, plan.tol = resid.value, tol
        plan.single_tol = decompose_tol is None and tol != DECOMPOSE_TOL
        plan.fill = nrow / float(nsteps * int(_lib.load().af_fused_gemm_slots(nant)))
        if ok.value:
            plan.ant_uvw, plan.rowmap = ant_uvw, rowmap
    return plan


# the GEMM form costs ~73 flop per baseline SLOT of the upper block triangle, the lane-per-row kernel ~126 per ROW
# (csrc/af_fused_gemm.hip header): below this fill (sub-arrays, baseline selections, an antenna axis much larger than
# the antennas present) the row kernel is the faster one.  A full 64-antenna step has fill 2016 / 2304 = 0.875; the
# slots of an array come from af_fused_gemm_slots (the super-tiles the kernel runs: 136 tiles at 128 antennas).
GEMM_MIN_FILL = 0.5


_plan_cache = collections.OrderedDict()
_plan_ident = collections.OrderedDict()     # identity of device tensors -> plan (see cached_plan)
_plan_lock = threading.Lock()


def _tensor_ident(arrays):
    """Identity key of a set of device tensors: (data_ptr, shape, dtype, torch's in-place version counter) each, or None
    when one of them is not a torch tensor.  The cache entry keeps weak references and is a hit only for the SAME tensor
    objects at the same version: a new tensor that happens to land on a freed address is a different object."""
    key = []
    for a in arrays:
        if a is None:
            key.append(None)
            continue
        if not _is_torch(a):
            return None
        key.append((a.data_ptr(), tuple(a.shape), str(a.dtype), a._version))
    return tuple(key)


def cached_plan(time_index, antenna1, antenna2, nant, grouped=True, uvw=None, single=False):
    """:func:`fused_plan` memoised on the CONTENTS of the three index arrays (a 16-byte digest: ~1 ms per 1e6 rows):
    what the row-chunk front-ends call (``rime.dask.fused_predict_vis``, ``chunked.fused_predict_vis``,
    ``sharding.fused_predict_shard``), where the same row chunk comes back for every channel block, every source chunk
    and every imaging cycle, each time as a fresh array.  ``AFHIP_PLAN_CACHE`` = number of plans kept (default 16,
    least recently used first out; 0 = no cache).  With ``uvw`` the plan carries the antenna decomposition (see
    :func:`fused_plan`) and the digest covers ``uvw`` as well.

    Device tensors are first looked up by IDENTITY -- the same tensor objects at the same in-place version (torch's
    ``_version``) as a previous call: no device -> host copy, no stream synchronisation, no O(row) host work --, and only
    on a miss copied to the host and digested.  Either way ``fused_predict_vis(plan=...)`` verifies the plan
    against the call's arrays on the device, so a wrong hit cannot produce a wrong result."""
    limit = int(os.environ.get("AFHIP_PLAN_CACHE", "16"))
    if limit <= 0:
        return fused_plan(time_index, antenna1, antenna2, nant, grouped, uvw, single=single)
    arrays = (time_index, antenna1, antenna2, uvw)
    ident = _tensor_ident(arrays)
    if ident is not None:
        ikey = (int(nant), bool(grouped), bool(single), ident)
        with _plan_lock:
            hit = _plan_ident.get(ikey)
            if hit is not None and all((r is None and a is None) or (r is not None and r() is a)
                                       for r, a in zip(hit[0], arrays)):
                plan = hit[1]()
                if plan is not None:                 # (None: evicted from _plan_cache since -- digest and rebuild)
                    _plan_ident.move_to_end(ikey)
                    if hit[2] in _plan_cache:
                        _plan_cache.move_to_end(hit[2])      # an identity hit is a use of the plan
                    return plan
                del _plan_ident[ikey]
    h = hashlib.blake2b(digest_size=16)
    n = 0
    for a in arrays:
        if a is None:
            continue
        a = np.ascontiguousarray(_host(a))
        n = int(a.shape[0])
        h.update(a.dtype.str.encode())
        h.update(a.view(np.uint8).reshape(-1).data if a.size else b"")
    key = (n, int(nant), bool(grouped), uvw is not None, bool(single), h.digest())
    with _plan_lock:
        plan = _plan_cache.get(key)
        if plan is not None:
            _plan_cache.move_to_end(key)
    if plan is None:
        plan = fused_plan(time_index, antenna1, antenna2, nant, grouped, uvw, single=single)
        with _plan_lock:
            _plan_cache[key] = plan
            while len(_plan_cache) > limit:
                _plan_cache.popitem(last=False)
    if ident is not None:
        import weakref
        refs = tuple(None if a is None else weakref.ref(a) for a in arrays)
        with _plan_lock:
            # entries whose tensors are gone can never hit again and would keep their plan (host arrays + device copies)
            # alive past its eviction from _plan_cache: dropped here.  The entry holds the plan WEAKLY --
            # _plan_cache alone decides how many plans live.
            for k in [k for k, (rs, _, _) in _plan_ident.items() if any(r is not None and r() is None for r in rs)]:
                del _plan_ident[k]
            _plan_ident[ikey] = (refs, weakref.ref(plan), key)
            while len(_plan_ident) > 4 * limit:
                _plan_ident.popitem(last=False)
    return plan


def fused_predict_vis(time_index, antenna1, antenna2, lm, uvw, frequency, brightness=None,
                      beam=None, beam_lm_extents=None, beam_freq_map=None, parallactic_angles=None,
                      point_errors=None, antenna_scaling=None,
                      die1_jones=None, base_vis=None, die2_jones=None, convention="fourier",
                      feed_rotation=None, gauss_shape=None, stokes=None, spi=None, ref_freq=None,
                      corr_schema=(("XX", "XY"), ("YX", "YY")), spectral_base=0, plan=None):
    """See :func:`_fused_predict_vis` (below) for the arguments.  Result type: the reference's rule for the chain this
    call replaces -- the promoted type of the inputs (africanus/util/type_inference.py:24-26): complex64 when EVERY
    floating-point input is single precision, complex128 otherwise.  Single-precision calls with a beam are COMPUTED in
    single precision (``af_fused_predict_antennas_c64`` on antenna-decomposable rows, ``af_fused_predict_c64`` on any others;
    phases in double; a sky model given as ``stokes`` / ``spi`` / ``ref_freq`` is first turned into the complex64 brightness
    array); without a beam and without Gaussian shapes: the single-precision direct transform (``af_im_to_vis_f32``, phases
    in double); Gaussian sources without a beam and the sky model without a beam compute in double and round once."""
    vis = _fused_predict_vis(time_index, antenna1, antenna2, lm, uvw, frequency, brightness, beam, beam_lm_extents, beam_freq_map,
                             parallactic_angles, point_errors, antenna_scaling, die1_jones, base_vis, die2_jones, convention,
                             feed_rotation, gauss_shape, stokes, spi, ref_freq, corr_schema, spectral_base, plan)
    if np_dtype_of(vis) == np.complex128 and _all_single(lm, uvw, frequency, brightness, beam, beam_lm_extents, beam_freq_map,
                                                         parallactic_angles, point_errors, antenna_scaling, die1_jones, base_vis,
                                                         die2_jones, feed_rotation, gauss_shape, stokes, spi, ref_freq):
        if _is_torch(vis):
            import torch
            return vis.to(torch.complex64)
        return vis.astype(np.complex64)
    return vis


def _fused_predict_vis(time_index, antenna1, antenna2, lm, uvw, frequency, brightness=None,
                       beam=None, beam_lm_extents=None, beam_freq_map=None, parallactic_angles=None,
                       point_errors=None, antenna_scaling=None,
                       die1_jones=None, base_vis=None, die2_jones=None, convention="fourier",
                       feed_rotation=None, gauss_shape=None, stokes=None, spi=None, ref_freq=None,
                       corr_schema=(("XX", "XY"), ("YX", "YY")), spectral_base=0, plan=None):
    """
    ``V_pq = G_p ( B_pq + sum_s E_ps (K_pqs X_s) E_qs^H ) G_q^H`` from source-level inputs.

    ``time_index``/``antenna1``/``antenna2`` (row,); ``lm`` (source, 2); ``uvw`` (row, 3);
    ``frequency`` (chan,); ``brightness`` (source, chan, 2, 2) -- or (source, 2, 2), flat in
    frequency -- the per-source coherency matrix X_s (africanus.model.coherency.convert output);
    optional beam cube arguments exactly as ``beam_cube_dde`` (all or none); optional
    ``die{1,2}_jones`` (time, ant, chan, 2, 2) and ``base_vis`` (row, chan, 2, 2) exactly as
    ``predict_vis``; optional ``feed_rotation`` (time, ant, 2, 2) multiplied onto the beam term,
    ``E <- E R`` (the ``einsum("stafij,tajk->stafik")`` of africanus/rime/examples/predict.py:472; needs the
    beam arguments) and ``gauss_shape`` (source, 3) = (major, minor, orientation) in radians, the Gaussian
    shape function of africanus/model/shape/gaussian_shape.py multiplied onto the phase term (rows of zeros
    are point sources).  Instead of ``brightness`` the SKY MODEL may be given: ``stokes`` (source, 4), ``spi``
    (source, spi-comps, 4), ``ref_freq`` (source,), ``corr_schema`` (2 x 2 nested: linear or circular feeds) and
    ``spectral_base``; ``brightness = convert(spectral_model(stokes, spi, ref_freq, frequency, spectral_base),
    ["I","Q","U","V"], corr_schema)`` (africanus/rime/examples/predict.py:494-498) is then evaluated on the device inside
    the call and no (source, chan, 2, 2) array exists on the caller's side.  ``plan``: a :func:`fused_plan` of the
    row layout (time_index, antenna1, antenna2) -- made once and re-used for every call on that layout (every channel
    block, every imaging cycle); without it the plan is rebuilt per call, which costs a device -> host copy of the three
    index arrays and O(row) host work.  A plan made with ``uvw=`` is bound to those ``uvw`` as well.  A plan passed in is
    VERIFIED against this call's ``time_index`` / ``antenna1`` / ``antenna2`` / ``uvw`` on the device
    (``af_fused_plan_check``, O(row), no host round trip): on a mismatch the result is NaN and ``ValueError`` is raised
    -- at once for numpy arguments, at the next call / ``check_status()`` for device tensors.  Returns (row, chan, 2, 2) complex
    (the type rule is :func:`fused_predict_vis`'s).  With a beam, rows should be grouped by
    ``time_index`` (Measurement-Set order): every run of equal ``time_index`` shares its
    per-antenna Jones terms on the device.
    """
    if convention not in _lib.CONVENTION:
        raise ValueError("convention not in ('fourier', 'casa')")
    beam_args = (beam, beam_lm_extents, beam_freq_map, parallactic_angles, point_errors, antenna_scaling)
    have_beam = beam is not None
    if any((a is None) != (not have_beam) for a in beam_args):
        raise ValueError("beam, beam_lm_extents, beam_freq_map, parallactic_angles, point_errors and "
                         "antenna_scaling must all be present or all absent")
    if (die1_jones is None) != (die2_jones is None):
        raise ValueError("Both die1_jones and die2_jones must be present or absent")
    if feed_rotation is not None and not have_beam:
        raise ValueError("feed_rotation multiplies the beam term: pass the beam arguments as well")
    nsrc, nrow, nchan = int(lm.shape[0]), int(uvw.shape[0]), int(frequency.shape[0])
    if tuple(lm.shape) != (nsrc, 2) or tuple(uvw.shape) != (nrow, 3):
        raise ValueError("lm must be (source, 2) and uvw (row, 3)")
    model = stokes is not None or spi is not None or ref_freq is not None
    if model:
        if brightness is not None or stokes is None or spi is None or ref_freq is None:
            raise ValueError("pass either brightness or all of stokes, spi and ref_freq")
        from ..dft.kernels import model_tables
        m_base, m_tabs, m_npol, m_ncorr, m_shape, _ = model_tables(stokes, spi, corr_schema, spectral_base)
        if m_shape != (2, 2):
            raise ValueError("corr_schema must be a 2 x 2 schema, e.g. [['XX', 'XY'], ['YX', 'YY']]")
        if int(stokes.shape[0]) != nsrc or int(spi.shape[0]) != nsrc or tuple(ref_freq.shape) != (nsrc,):
            raise ValueError("stokes, spi, ref_freq and lm disagree on the number of sources")
        if beam is None and gauss_shape is None:
            # no DDEs: the model-level direct transform, phase_delay's clamped n
            vis = _model_dft(stokes, spi, ref_freq, uvw, lm, frequency, m_base, m_tabs, m_npol, convention)
            if die1_jones is None and base_vis is None:
                return vis
            return predict_vis(time_index, antenna1, antenna2, None, vis[None], None, die1_jones, base_vis, die2_jones)
        brightness = None
        flat_spectrum = False
    elif brightness is None:
        raise ValueError("pass either brightness or all of stokes, spi and ref_freq")
    if gauss_shape is not None and not have_beam:
        # Gaussian sources without a beam: sum_s shape K X_s -- the einsum("srf,srf,sfij->srfij") + source sum of
        # africanus/rime/examples/predict.py:107-134 -- by the direct transform whose phasor carries the envelope
        # (csrc/af_gauss_dft.hip).  (Until round 4 this ran through the beam kernel with a cube of identity matrices.)
        if tuple(gauss_shape.shape) != (nsrc, 3):
            raise ValueError("gauss_shape must have shape (source, 3)")
        if model:
            from ..model.spectral import spectral_model
            from ..model.coherency import convert
            brightness = convert(spectral_model(stokes, spi, ref_freq, frequency, base=spectral_base),
                                 ["I", "Q", "U", "V"], [list(r) for r in corr_schema])
        bshape = tuple(int(x) for x in brightness.shape)
        if bshape not in ((nsrc, 2, 2), (nsrc, nchan, 2, 2)):
            raise ValueError("brightness must have shape (source, chan, 2, 2) or (source, 2, 2)")
        with Call(lm, uvw, frequency, brightness, gauss_shape) as c:
            if bshape == (nsrc, 2, 2):
                if _is_torch(brightness):
                    brightness = brightness[:, None].expand(nsrc, nchan, 2, 2)
                else:
                    brightness = np.broadcast_to(np.asarray(brightness)[:, None], (nsrc, nchan, 2, 2))
            p_lm, p_uvw, p_fr = c.inp(lm, np.float64), c.inp(uvw, np.float64), c.inp(frequency, np.float64)
            p_b, p_gs = c.inp(brightness, np.complex128), c.inp(gauss_shape, np.float64)
            p_out, h = c.out((nrow, nchan, 2, 2), np.complex128)
            ws_bytes = int(_lib.load().af_gauss_predict_workspace_bytes(nsrc, nchan))
            p_ws = c.scratch(ws_bytes)
            _lib.call("af_gauss_predict_c128", p_lm, p_uvw, p_fr, p_b, p_gs, nsrc, nrow, nchan,
                      _lib.CONVENTION[convention], p_out, p_ws, max(ws_bytes, 256), c.stream)
            vis = c.result(h)
        if die1_jones is None and base_vis is None:
            return vis
        return predict_vis(time_index, antenna1, antenna2, None, vis[None], None, die1_jones, base_vis, die2_jones)
    model_made = False
    if model and have_beam and os.environ.get("AFHIP_FUSED_C64", "1") != "0" and \
            _all_single(lm, uvw, frequency, stokes, spi, ref_freq, feed_rotation, gauss_shape, *beam_args):
        # a sky model with EVERY input single precision: the single-precision kernels take a brightness array, so it is made
        # first -- the reference's own order of work (africanus/rime/examples/predict.py:494-498), 32 bytes per (source, channel),
        # complex64 -- and the call goes on as one with `brightness` (Hermitian by construction: real Stokes parameters)
        from ..model.spectral import spectral_model
        from ..model.coherency import convert
        brightness = convert(spectral_model(stokes, spi, ref_freq, frequency, base=spectral_base),
                             ["I", "Q", "U", "V"], [list(r) for r in corr_schema])
        stokes = spi = ref_freq = None
        model, model_made = False, True
    bshape = tuple(int(s) for s in brightness.shape) if brightness is not None else (nsrc, nchan, 2, 2)
    if bshape == (nsrc, 2, 2):
        flat_spectrum = True
    elif bshape == (nsrc, nchan, 2, 2):
        flat_spectrum = False
    else:
        raise ValueError("brightness must have shape (source, chan, 2, 2) or (source, 2, 2)")
    for name, a in (("time_index", time_index), ("antenna1", antenna1), ("antenna2", antenna2)):
        if tuple(a.shape) != (nrow,):
            raise ValueError("%s must have shape (row,)" % name)

    if gauss_shape is not None and tuple(gauss_shape.shape) != (nsrc, 3):
        raise ValueError("gauss_shape must have shape (source, 3)")
    with Call(time_index, antenna1, antenna2, lm, uvw, frequency, brightness, feed_rotation, gauss_shape, stokes, spi,
              ref_freq, *beam_args) as c:
        brightness_given = brightness           # (the caller's object: what _hermitian remembers verdicts by)
        if flat_spectrum:
            if _is_torch(brightness):
                brightness = brightness[:, None].expand(nsrc, nchan, 2, 2)
            else:
                brightness = np.broadcast_to(np.asarray(brightness)[:, None], (nsrc, nchan, 2, 2))
        explicit_plan = plan is not None
        gemm = single_route = False
        if have_beam:
            if len(beam.shape) != 5 or tuple(beam.shape[3:]) != (2, 2):
                raise ValueError("beam must have shape (beam_lw, beam_mh, beam_nud, 2, 2)")
            beam_lw, beam_mh, beam_nud = (int(s) for s in beam.shape[:3])
            if beam_lw < 2 or beam_mh < 2 or beam_nud < 2:
                raise ValueError("beam_lw, beam_mh and beam_nud must be >= 2")
            ntime, nant = (int(s) for s in parallactic_angles.shape)
            if tuple(point_errors.shape) != (ntime, nant, nchan, 2):
                raise ValueError("point_errors must have shape (time, ant, chan, 2)")
            if tuple(antenna_scaling.shape) != (nant, nchan, 2):
                raise ValueError("antenna_scaling must have shape (ant, chan, 2)")
            if feed_rotation is not None and tuple(feed_rotation.shape) != (ntime, nant, 2, 2):
                raise ValueError("feed_rotation must have shape (time, ant, 2, 2)")
            all_single = not model and _all_single(lm, uvw, frequency, brightness, feed_rotation, gauss_shape, *beam_args)
            if plan is None:
                plan = fused_plan(time_index, antenna1, antenna2, nant, grouped=True,
                                  uvw=None if gauss_shape is not None else uvw, single=all_single)
            if plan.nrow != nrow or plan.nant != nant or plan.nsteps > ntime:
                raise ValueError("plan was made for %d rows, %d antennas, %d timesteps; the call has %d, %d, %d"
                                 % (plan.nrow, plan.nant, plan.nsteps, nrow, nant, ntime))
            # antenna-decomposable rows: the GEMM form (Gaussian shapes depend on the baseline: general kernel)
            gemm = plan.decomposable and gauss_shape is None and nrow > 0 and \
                plan.fill >= float(os.environ.get("AFHIP_GEMM_MIN_FILL", GEMM_MIN_FILL)) and \
                os.environ.get("AFHIP_FUSED_GEMM", "1") != "0"
            # ... and Hermitian brightness matrices (_hermitian: the sky model's always are)
            gemm = gemm and (model or model_made or _hermitian(brightness_given))
            # every input single precision: the reference computes this chain in float32 / complex64
            # (africanus/util/type_inference.py:24-26); the GEMM form has a single-precision kernel of its own
            # (and the lane-per-row kernel too: rows that do not decompose, Gaussian shapes, non-Hermitian brightness)
            single_route = all_single and os.environ.get("AFHIP_FUSED_C64", "1") != "0"
            # a plan whose rows decompose only at single precision's tolerance (fused_plan(..., single=True)) serves the
            # single-precision GEMM form alone: in double its residual (~1e-4 m) would be the error of the result
            if gemm and plan.single_tol and not single_route:
                gemm = False
        if single_route and not gemm:
            vis = _rows_c64(c, plan, explicit_plan, time_index, antenna1, antenna2, lm, uvw, frequency, brightness, beam,
                            beam_lm_extents, beam_freq_map, parallactic_angles, point_errors, antenna_scaling, feed_rotation,
                            gauss_shape, convention, nsrc, nrow, nchan)
            if die1_jones is None and base_vis is None:
                return vis
            return predict_vis(time_index, antenna1, antenna2, None, vis[None], None, die1_jones, base_vis, die2_jones)
        if single_route:
            vis = _gemm_c64(c, plan, explicit_plan, time_index, antenna1, antenna2, lm, uvw, frequency, brightness, beam,
                            beam_lm_extents, beam_freq_map, parallactic_angles, point_errors, antenna_scaling, feed_rotation,
                            convention, nsrc, nrow, nchan, allow_chunks=die1_jones is None and base_vis is None)
            if die1_jones is None and base_vis is None:
                return vis
            return predict_vis(time_index, antenna1, antenna2, None, vis[None], None, die1_jones, base_vis, die2_jones)
        if not have_beam and _all_single(lm, uvw, frequency, brightness) and os.environ.get("AFHIP_FUSED_C64", "1") != "0":
            # no DDEs, every input single precision: sum_s K X_s by the single-precision direct transform (complex image,
            # phase_delay's clamped n, phases in double: csrc/af_im_to_vis_f32.hip)
            f32 = np.float32
            p_lm, p_uvw, p_fr, p_b = c.inp(lm, f32), c.inp(uvw, f32), c.inp(frequency, f32), c.inp(brightness, np.complex64)
            p_out, h = c.out((nrow, nchan, 2, 2), np.complex64)
            ws_bytes = int(_lib.load().af_im_to_vis_f32_workspace_bytes(nsrc, nchan, 4, 1))
            p_ws = c.scratch(ws_bytes)
            _lib.call("af_im_to_vis_f32", p_b, 1, p_uvw, p_lm, p_fr, nsrc, nrow, nchan, 4, _lib.CONVENTION[convention],
                      _lib.AF_DFT_AUTO | _lib.AF_DFT_CLAMP_N, p_out, p_ws, max(ws_bytes, 256), c.stream)
            vis = c.result(h)
            if die1_jones is None and base_vis is None:
                return vis
            return predict_vis(time_index, antenna1, antenna2, None, vis[None], None, die1_jones, base_vis, die2_jones)
        p_lm, p_uvw, p_fr = c.inp(lm, np.float64), c.inp(uvw, np.float64), c.inp(frequency, np.float64)
        p_b = c.inp(brightness, np.complex128)
        if model:
            p_st, p_sp, p_rf = c.inp(stokes, np.float64), c.inp(spi, np.float64), c.inp(ref_freq, np.float64)
            p_mb = c.inp(m_base, np.int32)
        p_out, h = c.out((nrow, nchan, 2, 2), np.complex128)
        conv = _lib.CONVENTION[convention]
        if not have_beam:
            # sum_s K X_s: the direct transform with a complex image and phase_delay's clamped n
            ws_bytes = int(_lib.load().af_im_to_vis_workspace_bytes(nsrc, nchan, 4, 1))
            p_ws = c.scratch(ws_bytes)
            _lib.call("af_im_to_vis_f64", p_b, 1, p_uvw, p_lm, p_fr, nsrc, nrow, nchan, 4, conv,
                      _lib.AF_DFT_AUTO | _lib.AF_DFT_CLAMP_N, p_out, p_ws, max(ws_bytes, 256), c.stream)
        else:
            if gemm:
                p_au = c.inp(plan.device(plan.ant_uvw, c), np.float64)
                p_rm = c.inp(plan.device(plan.rowmap, c), np.int32)
            else:
                n_items = ctypes.c_int64(plan.n_items)
                p_items = c.inp(plan.device(plan.items, c), np.int32)
                p_groups = None if plan.groups is None else c.inp(plan.device(plan.groups, c), np.int32)
                p_a1 = c.inp(plan.device(plan.antenna1, c), np.int32)
                p_a2 = c.inp(plan.device(plan.antenna2, c), np.int32)
            p_beam, p_ext, p_map = c.inp(beam, np.complex128), c.inp(beam_lm_extents, np.float64), \
                c.inp(beam_freq_map, np.float64)
            p_pa, p_pe, p_as = c.inp(parallactic_angles, np.float64), c.inp(point_errors, np.float64), \
                c.inp(antenna_scaling, np.float64)
            p_fr_rot = c.inp(feed_rotation, np.complex128)
            p_gs = c.inp(gauss_shape, np.float64)
            if model and gemm:
                ws_bytes = int(_lib.load().af_fused_predict_model_workspace_bytes(nsrc, nchan, m_npol, beam_lw, beam_mh,
                                                                                  beam_nud))
                p_ws = c.scratch(ws_bytes)
                _lib.call("af_fused_predict_antennas_model_c128", p_st, p_sp, p_rf, p_mb, int(spi.shape[1]), m_npol,
                          m_tabs[0], m_tabs[1], m_tabs[2], p_au, p_rm, plan.nsteps, nrow, p_lm, p_fr, nsrc, nchan, p_beam,
                          beam_lw, beam_mh, beam_nud, p_ext, p_map, p_pa, ntime, nant, p_pe, p_as, p_fr_rot, conv, p_out,
                          p_ws, max(ws_bytes, 256), c.stream)
            elif gemm:
                ws_bytes = int(_lib.load().af_fused_predict_workspace_bytes(nsrc, nchan, beam_lw, beam_mh, beam_nud))
                p_ws = c.scratch(ws_bytes)
                nap = int(plan.rowmap.shape[1])
                if not c.device_mode and plan.time_sorted and not explicit_plan and die1_jones is None and base_vis is None:
                    # numpy caller, rows in time order: the result comes back in timestep-aligned row chunks whose
                    # downloads overlap the next chunk's kernels (Call.result_rows).  A chunk = the steps [t0, t1): the
                    # same entry on slices of the per-step arrays; the row map holds absolute rows, `out` stays the base
                    def launch(r0, r1, p_rows):
                        t0, t1 = int(plan.step[r0]), int(plan.step[r1 - 1]) + 1
                        off = lambda ptr, nbytes: None if ptr is None else ctypes.c_void_p(ptr.value + nbytes)
                        _lib.call("af_fused_predict_antennas_c128", off(p_au, t0 * nant * 24), off(p_rm, t0 * nap * nap * 4),
                                  t1 - t0, nrow, p_lm, p_fr, p_b, nsrc, nchan, p_beam, beam_lw, beam_mh, beam_nud, p_ext, p_map,
                                  off(p_pa, t0 * nant * 8), ntime - t0, nant, off(p_pe, t0 * nant * nchan * 16), p_as,
                                  off(p_fr_rot, t0 * nant * 64), conv, p_out, p_ws, max(ws_bytes, 256), c.stream)
                    return c.result_rows(h, launch, edges=plan.step_first)
                _lib.call("af_fused_predict_antennas_c128", p_au, p_rm, plan.nsteps, nrow, p_lm, p_fr, p_b, nsrc, nchan,
                          p_beam, beam_lw, beam_mh, beam_nud, p_ext, p_map, p_pa, ntime, nant, p_pe, p_as, p_fr_rot, conv,
                          p_out, p_ws, max(ws_bytes, 256), c.stream)
            elif model:
                ws_bytes = int(_lib.load().af_fused_predict_model_workspace_bytes(nsrc, nchan, m_npol, beam_lw, beam_mh,
                                                                                  beam_nud))
                p_ws = c.scratch(ws_bytes)
                _lib.call("af_fused_predict_model_c128", p_st, p_sp, p_rf, p_mb, int(spi.shape[1]), m_npol, m_tabs[0],
                          m_tabs[1], m_tabs[2], p_items, n_items.value, p_a1, p_a2, p_groups, nrow, p_lm, p_uvw, p_fr, nsrc, nchan,
                          p_beam, beam_lw, beam_mh, beam_nud, p_ext, p_map, p_pa, ntime, nant, p_pe, p_as, p_fr_rot, p_gs,
                          conv, p_out, p_ws, max(ws_bytes, 256), c.stream)
            else:
                ws_bytes = int(_lib.load().af_fused_predict_workspace_bytes(nsrc, nchan, beam_lw, beam_mh, beam_nud))
                p_ws = c.scratch(ws_bytes)
                _lib.call("af_fused_predict_c128", p_items, n_items.value, p_a1, p_a2, p_groups, nrow, p_lm, p_uvw, p_fr, p_b,
                          nsrc, nchan, p_beam, beam_lw, beam_mh, beam_nud, p_ext, p_map, p_pa, ntime, nant, p_pe,
                          p_as, p_fr_rot, p_gs, conv, p_out, p_ws, max(ws_bytes, 256), c.stream)
            if explicit_plan and nrow:
                _check_plan(c, plan, gemm, time_index, antenna1, antenna2, p_uvw, nrow, p_out, nrow * nchan * 8)
        vis = c.result(h)
    if die1_jones is None and base_vis is None:
        return vis
    # base_vis is added, then the DIEs applied, in the reference's order (africanus/rime/predict.py:605-612)
    return predict_vis(time_index, antenna1, antenna2, None, vis[None], None, die1_jones, base_vis, die2_jones)


_herm_ident = collections.OrderedDict()      # identity of a device brightness tensor -> verdict (see _hermitian)


def _hermitian(brightness):
    """Are the (..., 2, 2) brightness matrices Hermitian -- X[1,0] == conj(X[0,1]), real diagonal -- exactly?  Coherency
    matrices made from real Stokes parameters (africanus.model.coherency.convert) always are; the reference's chain accepts
    any complex matrices (``einsum("srf,sfij->srfij", phase, brightness)``).  The GEMM form needs it: it evaluates the upper
    block triangle of ``M = G H^H`` and serves the baseline stored the other way round with the conjugate transpose of the
    computed element, ``V_qp = (A_p X A_q^H)^H = A_q X^H A_p^H`` -- which is ``A_q X A_p^H`` only for Hermitian ``X``.  Anything
    else takes the lane-per-row kernel.  numpy: checked directly; a device tensor: one small device reduction and ONE host
    read per tensor object and in-place version (the verdict is remembered by identity, as plans are)."""
    if brightness is None:
        return True
    if _is_torch(brightness):
        import weakref
        key = (brightness.data_ptr(), tuple(brightness.shape), str(brightness.dtype), brightness._version)
        with _plan_lock:
            hit = _herm_ident.get(key)
            if hit is not None and hit[0]() is brightness:
                _herm_ident.move_to_end(key)
                return hit[1]
        x = brightness
        ok = bool(((x[..., 1, 0] == x[..., 0, 1].conj()) & (x[..., 0, 0].imag == 0) & (x[..., 1, 1].imag == 0)).all().item()) \
            if x.is_complex() else bool((x[..., 1, 0] == x[..., 0, 1]).all().item())
        with _plan_lock:
            for k in [k for k, (r, _) in _herm_ident.items() if r() is None]:
                del _herm_ident[k]
            _herm_ident[key] = (weakref.ref(brightness), ok)
            while len(_herm_ident) > 64:
                _herm_ident.popitem(last=False)
        return ok
    x = np.asarray(brightness)
    if not np.iscomplexobj(x):
        return bool(np.array_equal(x[..., 1, 0], x[..., 0, 1]))
    return bool(np.array_equal(x[..., 1, 0], np.conj(x[..., 0, 1])) and not x[..., 0, 0].imag.any() and not x[..., 1, 1].imag.any())


def _all_single(*arrays):
    """every array present is float32 / complex64"""
    single = (np.dtype(np.float32), np.dtype(np.complex64))
    return all(a is None or np_dtype_of(a) in single for a in arrays)


def _gemm_c64(c, plan, explicit_plan, time_index, antenna1, antenna2, lm, uvw, frequency, brightness, beam, beam_lm_extents,
              beam_freq_map, parallactic_angles, point_errors, antenna_scaling, feed_rotation, convention, nsrc, nrow, nchan,
              allow_chunks=True):
    """The single-precision GEMM form (af_fused_predict_antennas_c64) inside an open Call: complex64 result."""
    bshape = tuple(int(s) for s in brightness.shape)
    if bshape == (nsrc, 2, 2):
        if _is_torch(brightness):
            brightness = brightness[:, None].expand(nsrc, nchan, 2, 2)
        else:
            brightness = np.broadcast_to(np.asarray(brightness)[:, None], (nsrc, nchan, 2, 2))
    elif bshape != (nsrc, nchan, 2, 2):
        raise ValueError("brightness must have shape (source, chan, 2, 2) or (source, 2, 2)")
    beam_lw, beam_mh, beam_nud = (int(s) for s in beam.shape[:3])
    ntime, nant = (int(s) for s in parallactic_angles.shape)
    f32, c64 = np.float32, np.complex64
    p_au = c.inp(plan.device(plan.ant_uvw, c), np.float64)
    p_rm = c.inp(plan.device(plan.rowmap, c), np.int32)
    p_lm, p_fr, p_b = c.inp(lm, f32), c.inp(frequency, f32), c.inp(brightness, c64)
    p_beam, p_ext, p_map = c.inp(beam, c64), c.inp(beam_lm_extents, f32), c.inp(beam_freq_map, f32)
    p_pa, p_pe, p_as = c.inp(parallactic_angles, f32), c.inp(point_errors, f32), c.inp(antenna_scaling, f32)
    p_rot = c.inp(feed_rotation, c64)
    p_out, h = c.out((nrow, nchan, 2, 2), c64)
    ws_bytes = int(_lib.load().af_fused_predict_c64_workspace_bytes(nsrc, nchan, beam_lw, beam_mh, beam_nud))
    p_ws = c.scratch(ws_bytes)
    conv = _lib.CONVENTION[convention]
    if not c.device_mode and plan.time_sorted and not explicit_plan and allow_chunks:
        # numpy caller, rows in time order: timestep-aligned row chunks whose downloads overlap the next chunk's kernels
        # (Call.result_rows), as on the double-precision GEMM route
        nap = int(plan.rowmap.shape[1])

        def launch(r0, r1, p_rows):
            t0, t1 = int(plan.step[r0]), int(plan.step[r1 - 1]) + 1
            off = lambda ptr, nbytes: None if ptr is None else ctypes.c_void_p(ptr.value + nbytes)
            _lib.call("af_fused_predict_antennas_c64", off(p_au, t0 * nant * 24), off(p_rm, t0 * nap * nap * 4), t1 - t0, nrow,
                      p_lm, p_fr, p_b, nsrc, nchan, p_beam, beam_lw, beam_mh, beam_nud, p_ext, p_map, off(p_pa, t0 * nant * 4),
                      ntime - t0, nant, off(p_pe, t0 * nant * nchan * 8), p_as, off(p_rot, t0 * nant * 32), conv, p_out, p_ws,
                      max(ws_bytes, 256), c.stream)
        return c.result_rows(h, launch, edges=plan.step_first)
    _lib.call("af_fused_predict_antennas_c64", p_au, p_rm, plan.nsteps, nrow, p_lm, p_fr, p_b, nsrc, nchan, p_beam, beam_lw,
              beam_mh, beam_nud, p_ext, p_map, p_pa, ntime, nant, p_pe, p_as, p_rot, conv, p_out, p_ws,
              max(ws_bytes, 256), c.stream)
    if explicit_plan and nrow:
        # (the guard fills float64 words with NaN: a complex64 result is nrow * nchan * 4 of them)
        _check_plan(c, plan, True, time_index, antenna1, antenna2, c.inp(uvw, np.float64), nrow, p_out, nrow * nchan * 4)
    return c.result(h)


def _rows_c64(c, plan, explicit_plan, time_index, antenna1, antenna2, lm, uvw, frequency, brightness, beam, beam_lm_extents,
              beam_freq_map, parallactic_angles, point_errors, antenna_scaling, feed_rotation, gauss_shape, convention, nsrc, nrow,
              nchan):
    """The single-precision lane-per-row form (af_fused_predict_c64) inside an open Call: complex64 result."""
    beam_lw, beam_mh, beam_nud = (int(s) for s in beam.shape[:3])
    ntime, nant = (int(s) for s in parallactic_angles.shape)
    f32, c64 = np.float32, np.complex64
    p_items = c.inp(plan.device(plan.items, c), np.int32)
    p_groups = None if plan.groups is None else c.inp(plan.device(plan.groups, c), np.int32)
    p_a1 = c.inp(plan.device(plan.antenna1, c), np.int32)
    p_a2 = c.inp(plan.device(plan.antenna2, c), np.int32)
    p_lm, p_uvw, p_fr, p_b = c.inp(lm, f32), c.inp(uvw, f32), c.inp(frequency, f32), c.inp(brightness, c64)
    p_beam, p_ext, p_map = c.inp(beam, c64), c.inp(beam_lm_extents, f32), c.inp(beam_freq_map, f32)
    p_pa, p_pe, p_as = c.inp(parallactic_angles, f32), c.inp(point_errors, f32), c.inp(antenna_scaling, f32)
    p_rot, p_gs = c.inp(feed_rotation, c64), c.inp(gauss_shape, f32)
    p_out, h = c.out((nrow, nchan, 2, 2), c64)
    ws_bytes = int(_lib.load().af_fused_predict_c64_workspace_bytes(nsrc, nchan, beam_lw, beam_mh, beam_nud))
    p_ws = c.scratch(ws_bytes)
    _lib.call("af_fused_predict_c64", p_items, plan.n_items, p_a1, p_a2, p_groups, nrow, p_lm, p_uvw, p_fr, p_b, nsrc, nchan,
              p_beam, beam_lw, beam_mh, beam_nud, p_ext, p_map, p_pa, ntime, nant, p_pe, p_as, p_rot, p_gs,
              _lib.CONVENTION[convention], p_out, p_ws, max(ws_bytes, 256), c.stream)
    if explicit_plan and nrow:
        # (the guard fills float64 words with NaN: a complex64 result is nrow * nchan * 4 of them)
        _check_plan(c, plan, False, time_index, antenna1, antenna2, c.inp(uvw, np.float64), nrow, p_out, nrow * nchan * 4)
    return c.result(h)


def _plan_status_message(flags):
    what = []
    if flags & _lib.AF_STATUS_PLAN_INDEX:
        what.append("time_index / antenna1 / antenna2 differ from the arrays the plan was made from")
    if flags & _lib.AF_STATUS_PLAN_UVW:
        what.append("uvw differ from the uvw the plan's antenna decomposition was solved from")
    return "fused_predict_vis: stale plan: " + " and ".join(what)


def _check_plan(c, plan, gemm, time_index, antenna1, antenna2, p_uvw, nrow, p_out, out_doubles):
    """A caller-supplied plan against the call's own arrays (af_fused_plan_check, after the predict on its stream)."""
    from .predict import _index_array
    p_ti, ib = _index_array(c, time_index)
    if ib == 4 and not (np_dtype_of(antenna1) == np.int32 and np_dtype_of(antenna2) == np.int32):
        p_ti, ib = c.inp(time_index, np.int64), 8
    idt = np.int32 if ib == 4 else np.int64
    p_a1, p_a2 = c.inp(antenna1, idt), c.inp(antenna2, idt)
    p_ps = c.inp(plan.device(plan.step, c), np.int32)
    p_pa1, p_pa2 = c.inp(plan.device(plan.antenna1, c), np.int32), c.inp(plan.device(plan.antenna2, c), np.int32)
    p_au = c.inp(plan.device(plan.ant_uvw, c), np.float64) if gemm else None
    p_status = c.scratch(256)
    _lib.call("af_fused_plan_check", p_ti, p_a1, p_a2, ib, p_uvw, nrow, p_ps, p_pa1, p_pa2, p_au, plan.nant,
              plan.tol, p_out, out_doubles, p_status, c.stream)
    c.watch_status(0, _plan_status_message)


def _model_dft(stokes, spi, ref_freq, uvw, lm, frequency, base, tabs, npol, convention):
    """Model-level direct transform with phase_delay's clamped n (the no-DDE route of fused_predict_vis): the four
    complex correlations, (row, chan, 2, 2)."""
    nsrc, nspi, nchan, nrow = int(stokes.shape[0]), int(spi.shape[1]), int(frequency.shape[0]), int(uvw.shape[0])
    with Call(stokes, spi, ref_freq, uvw, lm, frequency) as c:
        p_st, p_sp, p_rf = c.inp(stokes, np.float64), c.inp(spi, np.float64), c.inp(ref_freq, np.float64)
        p_uvw, p_lm, p_fr = c.inp(uvw, np.float64), c.inp(lm, np.float64), c.inp(frequency, np.float64)
        p_b = c.inp(base, np.int32)
        p_out, h = c.out((nrow, nchan, 2, 2), np.complex128)
        ws_bytes = int(_lib.load().af_im_to_vis_model_workspace_bytes(nsrc, nchan, npol, 4, 1))
        p_ws = c.scratch(ws_bytes)
        _lib.call("af_im_to_vis_model_f64", p_st, p_sp, p_rf, p_b, nspi, npol, tabs[0], tabs[1], tabs[2], 4, 1, p_uvw,
                  p_lm, p_fr, nsrc, nrow, nchan, _lib.CONVENTION[convention], _lib.AF_DFT_AUTO | _lib.AF_DFT_CLAMP_N,
                  p_out, p_ws, max(ws_bytes, 256), c.stream)
        return c.result(h)
