"""
codex_africanus_amd -- MI355X (gfx950) implementation of codex-africanus' RIME
visibility-predict hot path, behind the reference's own function signatures:

    codex_africanus_amd.rime.phase_delay      (africanus/rime/phase.py:11)
    codex_africanus_amd.rime.predict_vis      (africanus/rime/predict.py:466)
    codex_africanus_amd.rime.apply_gains      (africanus/rime/predict.py:622)
    codex_africanus_amd.rime.beam_cube_dde    (africanus/rime/fast_beam_cubes.py:57)
    codex_africanus_amd.dft.im_to_vis         (africanus/dft/kernels.py:14)

All arithmetic runs in hand-written HIP kernels (csrc/*.hip) reached through the C ABI
of include/afhip.h; there is no CPU fallback.
"""
from ._lib import build, device_count, device_info, get_device, set_device  # noqa: F401


def check_status(wait=True):
    """Raise ``ValueError`` if a device-mode call made so far met an out-of-range index (its rows are NaN).  Such a
    call cannot raise by itself -- nothing synchronises on the device path -- so the error surfaces at the next call
    into the package, or here (``wait=True`` waits for the outstanding calls first)."""
    from ._device import check_status as _cs
    _cs(wait)

__version__ = "0.1.0"
