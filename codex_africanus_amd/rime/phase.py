"""phase_delay with the signature of africanus/rime/phase.py:11-13."""
import numpy as np

from .. import _lib
from .._device import Call, np_dtype_of


def phase_delay(lm, uvw, frequency, convention="fourier"):
    """
    Phase delay (K) term exp(-+2 pi i (u l + v m + w (n - 1)) nu / c).

    Same contract as ``africanus.rime.phase_delay`` (africanus/rime/phase.py:11-63):
    ``lm`` (source, 2), ``uvw`` (row, 3), ``frequency`` (chan,) -> complex
    (source, row, chan); ``convention`` 'fourier' (exp(-2 pi i)) or 'casa' (exp(+2 pi i));
    output dtype ``result_type(complex64, lm, uvw, frequency)``; ``n`` is clamped at the
    unit circle (phase.py:43).  numpy in -> numpy out; torch ROCm tensors in -> torch out.
    """
    if convention not in _lib.CONVENTION:
        raise ValueError("convention not in ('fourier', 'casa')")
    dts = [np_dtype_of(a) for a in (lm, uvw, frequency)]
    out_dtype = np.result_type(np.complex64, *dts)
    # all-float32 inputs compute in float32 like the reference (constants cast to lm.dtype,
    # phase.py:23-25); mixed precisions are promoted to float64 first.
    if out_dtype == np.complex64:
        rt, fn = np.float32, "af_phase_delay_f32"
    else:
        rt, fn, out_dtype = np.float64, "af_phase_delay_f64", np.dtype(np.complex128)
    if len(lm.shape) != 2 or lm.shape[1] != 2:
        raise ValueError("lm must have shape (source, 2)")
    if len(uvw.shape) != 2 or uvw.shape[1] != 3:
        raise ValueError("uvw must have shape (row, 3)")
    if len(frequency.shape) != 1:
        raise ValueError("frequency must have shape (chan,)")
    nsrc, nrow, nchan = int(lm.shape[0]), int(uvw.shape[0]), int(frequency.shape[0])
    with Call(lm, uvw, frequency) as c:
        p_lm, p_uvw, p_fr = c.inp(lm, rt), c.inp(uvw, rt), c.inp(frequency, rt)
        p_out, h = c.out((nsrc, nrow, nchan), out_dtype)
        _lib.call(fn, p_lm, nsrc, p_uvw, nrow, p_fr, nchan, _lib.CONVENTION[convention], p_out, c.stream)
        return c.result(h)
