"""spectral_model with the signature of africanus/model/spectral/spec_model.py:102-103."""
import numpy as np

from ... import _lib
from ..._device import Call, np_dtype_of

_BASES = {0: 0, 1: 1, 2: 2, "std": 0, "log": 1, "log10": 2}


def spectral_model(stokes, spi, ref_freq, frequency, base=0):
    """
    Per-polarisation spectral model: ``base`` 0 / "std" ``I prod_i (nu/nu0)^spi_i``, 1 / "log"
    ``I exp(sum_i spi_i ln(nu/nu0)^(i+1))``, 2 / "log10" the same in base 10; a list gives one base per
    polarisation (the last entry repeats).

    Same contract as ``africanus.model.spectral.spectral_model`` (africanus/model/spectral/spec_model.py:102-236):
    ``stokes`` (source,) or (source, pol...), ``spi`` (source, spi-comps) or (source, spi-comps, pol...),
    ``ref_freq`` (source,), ``frequency`` (chan,) -> (source, chan[, pol...]) of dtype
    ``result_type(stokes, spi, ref_freq, frequency)``; the same ``ValueError``s.
    """
    if len(spi.shape) - 2 != len(stokes.shape) - 1:
        raise ValueError("Dimensions on stokes and spi don't agree")
    pol_shape = tuple(int(s) for s in stokes.shape[1:])
    npol = int(np.prod(pol_shape, dtype=np.int64)) if pol_shape else 1
    spi_pol = tuple(int(s) for s in spi.shape[2:])
    if npol != (int(np.prod(spi_pol, dtype=np.int64)) if spi_pol else 1):
        raise ValueError("Correlations on stokes and spi don't agree")
    if not isinstance(base, (list, tuple, int, str, np.integer)):
        raise TypeError("base '%s' should be a string or integer" % (base,))
    bl = list(base) if isinstance(base, (list, tuple)) else [base] * npol
    bl = bl + [bl[-1]] * (npol - len(bl))
    try:
        b = np.array([_BASES[x] for x in bl[:npol]], dtype=np.int32)
    except (KeyError, TypeError):
        raise ValueError("Invalid base")
    nsrc, nspi, nchan = int(stokes.shape[0]), int(spi.shape[1]), int(frequency.shape[0])
    out_dtype = np.result_type(*[np_dtype_of(a) for a in (stokes, spi, ref_freq, frequency)])
    with Call(stokes, spi, ref_freq, frequency) as c:
        p_st, p_sp, p_rf, p_fr = (c.inp(a, np.float64) for a in (stokes, spi, ref_freq, frequency))
        p_b = c.inp(b, np.int32)
        p_out, h = c.out((nsrc, nchan) + pol_shape, np.float64)
        _lib.call("af_spectral_model_f64", p_st, p_sp, p_rf, p_fr, p_b, nsrc, nspi, npol, nchan, p_out, c.stream)
        return c.result(h, cast=None if out_dtype == np.float64 else out_dtype)
