"""
ctypes binding of libafhip.so (C ABI declared in include/afhip.h).

The shared library is built in-tree (codex_africanus_amd/lib/libafhip.so) by
``build()`` / ``make -C codex_africanus_amd/csrc``; there is no CPU fallback: if
the library is missing, loading raises.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# AFHIP_LIB: another build of the same library (an experimental or the profiling build, tools/ab_lib.sh,
# tools/profile_gemm_stage.sh) -- chosen per process, nothing is overwritten in place
LIB_PATH = os.path.abspath(os.environ["AFHIP_LIB"]) if os.environ.get("AFHIP_LIB") else \
    os.path.join(_HERE, "lib", "libafhip.so")
CSRC = os.path.join(_HERE, "csrc")

AF_OK, AF_EINVAL, AF_ENOMEM, AF_ENOTSUP, AF_EHIP_BASE = 0, 1, 2, 3, 1000
CONVENTION = {"fourier": -1, "casa": 1}
AF_DFT_AUTO, AF_DFT_EXACT, AF_DFT_RECURRENCE = 0, 1, 2
AF_DFT_CLAMP_N = 0x100
AF_DFT_VALU_ONLY = 0x200
AF_JONES_DIAG, AF_JONES_2X2 = 1, 2
AF_STATUS_TIME_INDEX, AF_STATUS_ANTENNA = 1, 2
AF_STATUS_PLAN_INDEX, AF_STATUS_PLAN_UVW = 4, 8
AF_PREDICT_VIS_STATUS_OFFSET = 8

_vp, _i64, _int, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/afhip.h one to one
_SIGNATURES = {
    "af_version": (_int, []),
    "af_last_error": (ctypes.c_char_p, []),
    "af_device_count": (_int, [ctypes.POINTER(_int)]),
    "af_set_device": (_int, [_int]),
    "af_get_device": (_int, [ctypes.POINTER(_int)]),
    "af_device_info": (_int, [_int, ctypes.c_char_p, _sz, ctypes.c_char_p, _sz, ctypes.POINTER(_int),
                              ctypes.POINTER(_sz)]),
    "af_malloc": (_int, [ctypes.POINTER(_vp), _sz]),
    "af_free": (_int, [_vp]),
    "af_malloc_host": (_int, [ctypes.POINTER(_vp), _sz]),
    "af_free_host": (_int, [_vp]),
    "af_pool_malloc": (_int, [ctypes.POINTER(_vp), _sz]),
    "af_pool_free": (_int, [_vp]),
    "af_pool_malloc_host": (_int, [ctypes.POINTER(_vp), _sz]),
    "af_pool_free_host": (_int, [_vp]),
    "af_pool_trim": (_int, [_int, _sz]),
    "af_pool_stats": (_int, [_int, ctypes.POINTER(_sz), ctypes.POINTER(_sz), ctypes.POINTER(_i64),
                             ctypes.POINTER(_i64)]),
    "af_thread_stream": (_int, [ctypes.POINTER(_vp)]),
    "af_shutdown": (_int, []),
    "af_memcpy_h2d": (_int, [_vp, _vp, _sz, _vp]),
    "af_memcpy_d2h": (_int, [_vp, _vp, _sz, _vp]),
    "af_memcpy_d2d": (_int, [_vp, _vp, _sz, _vp]),
    "af_memset": (_int, [_vp, _int, _sz, _vp]),
    "af_stream_create": (_int, [ctypes.POINTER(_vp)]),
    "af_stream_destroy": (_int, [_vp]),
    "af_stream_synchronize": (_int, [_vp]),
    "af_device_synchronize": (_int, []),
    "af_event_create": (_int, [ctypes.POINTER(_vp)]),
    "af_event_destroy": (_int, [_vp]),
    "af_event_record": (_int, [_vp, _vp]),
    "af_stream_wait_event": (_int, [_vp, _vp]),
    "af_event_synchronize": (_int, [_vp]),
    "af_event_elapsed_ms": (_int, [_vp, _vp, ctypes.POINTER(ctypes.c_float)]),
    "af_profile_events": (_int, [_vp, _vp]),
    "af_convert_f32_to_f64": (_int, [_vp, _vp, _i64, _vp]),
    "af_convert_f64_to_f32": (_int, [_vp, _vp, _i64, _vp]),
    "af_phase_delay_f64": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _int, _vp, _vp]),
    "af_phase_delay_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _int, _vp, _vp]),
    "af_im_to_vis_workspace_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "af_im_to_vis_f64": (_int, [_vp, _int, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _vp, _sz,
                                _vp]),
    "af_im_to_vis_chi2_f64": (_int, [_vp, _int, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp,
                                     _sz, _vp]),
    "af_im_to_vis_f32_workspace_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "af_im_to_vis_f32": (_int, [_vp, _int, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _vp, _sz,
                                _vp]),
    "af_vis_to_im_f32_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "af_vis_to_im_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _vp, _sz, _vp]),
    "af_vis_to_im_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "af_vis_to_im_f64": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _vp, _sz, _vp]),
    "af_predict_vis_workspace_bytes": (_sz, []),
    "af_predict_vis_c128": (_int, [_vp, _vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                   _i64, _int, _int, _vp, _vp, _sz, _vp]),
    "af_predict_vis_c64": (_int, [_vp, _vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                  _i64, _int, _int, _vp, _vp, _sz, _vp]),
    "af_freq_grid_interp_f64": (_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "af_freq_grid_interp_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "af_beam_cube_dde_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64, _i64, _i64, _int]),
    "af_beam_cube_dde_c128": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp,
                                     _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "af_beam_cube_dde_c64": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp,
                                    _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "af_fused_plan_rows": (_int, [_vp, _i64, _vp, _i64, ctypes.POINTER(_i64)]),
    "af_fused_predict_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "af_fused_plan_groups": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, ctypes.POINTER(_i64), _vp, _i64,
                                    ctypes.POINTER(_i64)]),
    "af_fused_predict_c128": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64,
                                     _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _int, _vp, _vp, _sz, _vp]),
    "af_chi2_c128": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "af_feed_rotation_f64": (_int, [_vp, _i64, _int, _vp, _vp]),
    "af_feed_rotation_f32": (_int, [_vp, _i64, _int, _vp, _vp]),
    "af_gaussian_shape_f64": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "af_calibration_workspace_bytes": (_sz, [_i64]),
    "af_correct_vis_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "af_corrupt_vis_c128": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _vp,
                                   _sz, _vp]),
    "af_residual_vis_c128": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int,
                                    _vp, _vp, _sz, _vp]),
    "af_correct_vis_c128": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp,
                                   _vp, _sz, _vp]),
    "af_degridder_workspace_bytes": (_sz, [_i64]),
    "af_degridder_c128": (_int, [_vp, _vp, _vp, _vp, ctypes.c_double, _vp, _vp, _vp, _i64, _i64, _int, _vp, _int, _int,
                                 _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "af_spectral_model_f64": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "af_compute_and_corrupt_vis_c128": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64,
                                               _int, _int, _vp, _vp, _sz, _vp]),
    "af_im_to_vis_model_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _int]),
    "af_im_to_vis_model_f64": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _i64, _i64,
                                      _i64, _int, _int, _vp, _vp, _sz, _vp]),
    "af_gauss_predict_workspace_bytes": (_sz, [_i64, _i64]),
    "af_gauss_predict_c128": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp, _vp, _sz, _vp]),
    "af_gauss_predict_chi2_c128": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "af_fused_plan_antennas": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_double, _i64, _vp, _vp,
                                      ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "af_fused_gemm_slots": (_i64, [_i64]),
    "af_fused_plan_check": (_int, [_vp, _vp, _vp, _int, _vp, _i64, _vp, _vp, _vp, _vp, _i64, ctypes.c_double, _vp, _i64, _vp,
                                   _vp]),
    "af_fused_predict_antennas_c128": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp,
                                              _vp, _vp, _i64, _i64, _vp, _vp, _vp, _int, _vp, _vp, _sz, _vp]),
    "af_fused_predict_c64_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "af_fused_predict_antennas_c64": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp,
                                             _vp, _vp, _i64, _i64, _vp, _vp, _vp, _int, _vp, _vp, _sz, _vp]),
    "af_fused_predict_c64": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64,
                                    _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _int, _vp, _vp, _sz, _vp]),
    "af_fused_predict_antennas_model_c128": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp,
                                                    _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp,
                                                    _vp, _vp, _int, _vp, _vp, _sz, _vp]),
    "af_fused_predict_model_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64, _i64]),
    "af_fused_predict_model_c128": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp,
                                           _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp,
                                           _vp, _vp, _int, _vp, _vp, _sz, _vp]),
    "af_wgrid_padded": (_i64, [_i64]),
    "af_wgrid_plane_precision": (_int, [_int]),
    "af_wgrid_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64, _i64, _int]),
    "af_wgrid_planes": (_i64, [ctypes.c_double, ctypes.c_double, ctypes.c_double, _int, _int]),
    "af_wgrid_im2vis_f64": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, ctypes.c_double, ctypes.c_double,
                                   _vp, _vp, _vp, _vp, _int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "af_wgrid_vis2im_f64": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, ctypes.c_double, ctypes.c_double,
                                   _vp, _vp, _vp, _vp, _int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "af_gridder_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "af_gridder_c128": (_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_double, _vp, _vp, _vp, _i64, _i64, _int, _vp, _int, _int,
                               _int, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "af_coherency_convert": (_int, [_vp, _int, _i64, _int, _int, _vp, _vp, _vp, _vp, _int, _vp]),
    "af_wsclean_spectra_f64": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "af_wsclean_predict_workspace_bytes": (_sz, [_i64, _i64]),
    "af_wsclean_predict_f64": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int,
                                      _vp, _vp, _sz, _vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lock = threading.Lock()
_lib = None


class AfHipError(RuntimeError):
    """A HIP runtime failure reported by libafhip (status >= AF_EHIP_BASE)."""


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into codex_africanus_amd/lib/libafhip.so."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "afhip.h"))
    if os.environ.get("AFHIP_LIB"):
        return LIB_PATH             # somebody else's build: used as it is
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        cmd = ["make", "-C", CSRC, "-j", "4", "all"]
        if force:
            cmd.insert(1, "-B")
        subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def _preload_hip_runtime():
    """One HIP runtime per process: when PyTorch-ROCm is installed it bundles its own
    libamdhip64.so.7 (+ HSA runtime); loading ROCm's copy first and torch's afterwards leaves
    torch without devices.  Bind libafhip to torch's copy (same SONAME) by loading that one
    first -- without importing torch.  Without torch, /opt/rocm's runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        return ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    return None


def load():
    """Load libafhip.so (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise ImportError(
                        "libafhip.so not found at %s: build it with "
                        "`python -c 'import __graft_entry__ as g; g.build()'` or "
                        "`make -C codex_africanus_amd/csrc`" % LIB_PATH)
                _preload_hip_runtime()
                lib = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in _SIGNATURES.items():
                    fn = getattr(lib, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = lib
    return _lib


def check(status, lib=None):
    """Map a libafhip status to the exception the reference raises for it."""
    if status == AF_OK:
        return
    lib = lib or load()
    msg = lib.af_last_error().decode("utf-8", "replace")
    if status == AF_EINVAL:
        raise ValueError(msg)
    if status == AF_ENOMEM:
        raise MemoryError(msg)
    if status == AF_ENOTSUP:
        raise NotImplementedError(msg)
    raise AfHipError("libafhip status %d: %s" % (status, msg))


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), lib)


def device_count():
    n = _int(0)
    call("af_device_count", ctypes.byref(n))
    return n.value


def set_device(device):
    call("af_set_device", int(device))


def get_device():
    d = _int(0)
    call("af_get_device", ctypes.byref(d))
    return d.value


def device_info(device=None):
    device = get_device() if device is None else device
    name, arch = ctypes.create_string_buffer(256), ctypes.create_string_buffer(256)
    cus, mem = _int(0), _sz(0)
    call("af_device_info", device, name, 256, arch, 256, ctypes.byref(cus), ctypes.byref(mem))
    return dict(name=name.value.decode(), arch=arch.value.decode(), compute_units=cus.value,
                total_mem=mem.value)


def thread_stream():
    """The calling thread's own stream on its current device (af_thread_stream), as a c_void_p."""
    st = _vp()
    call("af_thread_stream", ctypes.byref(st))
    return st


def pool_stats(device=None):
    """Scratch-pool counters of `device` (None = current, -1 = the page-locked host list)."""
    device = get_device() if device is None else device
    cached, used, hits, misses = _sz(0), _sz(0), _i64(0), _i64(0)
    call("af_pool_stats", int(device), ctypes.byref(cached), ctypes.byref(used), ctypes.byref(hits),
         ctypes.byref(misses))
    return dict(cached_bytes=cached.value, in_use_bytes=used.value, hits=hits.value, misses=misses.value)


def shutdown():
    """Release every cached pool block and the per-thread streams (af_shutdown)."""
    if _lib is not None:
        call("af_shutdown")
