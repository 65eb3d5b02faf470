"""Anti-aliasing windows of the Perley-polyhedron (de)gridder: host-side set-up data (a few hundred doubles).

Interface of africanus/gridding/perleypolyhedron/kernels.py:16-119 (``uspace``, ``sinc``, ``kbsinc``,
``hanningsinc``, ``pack_kernel``, ``unpack_kernel``: same names, arguments, defaults and values to rounding), built
here around one generator: every window is ``sinc(u) * taper(x)`` sampled on the tap grid and scaled to unit
sum, with ``x = 2 u / (W + 3)`` the tap position relative to the padded support.

Tap grid: a W-tap kernel carries one guard tap either side, ``P = W + 2`` taps, each split into ``oversample``
phases; sample k sits at ``u_k = (k - oversample * (P // 2)) / oversample`` pixels from the kernel centre.
"""
import numpy as np

# window parameter against padded support P = W + 2: tabulated optima, interpolated by a least-squares polynomial
# (the same tables and degrees as the reference, which is where the numbers come from: kernels.py:44-49,70-72)
_TAPER_TABLES = {
    "kaiser_bessel": (1, ((1.5, 1.9980), (2.0, 2.3934), (2.5, 3.3800), (3.0, 4.2054),
                          (3.5, 4.9107), (4.0, 5.7567), (4.5, 6.6291), (5.0, 7.4302))),
    "hanning": (3, ((1.5, 0.7600), (2.0, 0.7146), (2.5, 0.6185), (3.0, 0.5534), (3.5, 0.5185))),
}


def _default_parameter(taper, support):
    degree, table = _TAPER_TABLES[taper]
    x, y = np.array(table).T
    return float(np.polyval(np.polyfit(x, y, degree), support))


def _padded(W):
    if W % 2 != 1:
        raise AssertionError("W must be odd: the taps are centred on the origin")
    return W + 2


def uspace(W, oversample):
    """Positions (pixels from the kernel centre) of the ``oversample * (W + 2)`` samples of a W-tap kernel."""
    P = _padded(W)
    k = np.arange(oversample * P)
    return k / float(oversample) - P // 2


def _window(W, oversample, taper=None, sinc_scale=1.0):
    u = uspace(W, oversample)
    values = np.sinc(u * sinc_scale)
    values = values / values.sum()
    if taper is not None:
        values = values * taper(u, _padded(W) + 1.0)
        values = values / values.sum()
    return values


def sinc(W, oversample=5, a=1.0):
    """Oversampled sinc, unit sum."""
    return _window(W, oversample, None, a)


def kbsinc(W, b=None, oversample=5, order=15):
    """Sinc tapered by ``J_order(b sqrt(1 - x^2))`` (a Kaiser-Bessel-like window with a high-order Bessel
    function, which behaves better at few taps); ``b`` defaults to the tabulated optimum.  Needs scipy."""
    from scipy.special import jn
    beta = _default_parameter("kaiser_bessel", _padded(W)) if b is None else b

    def taper(u, span):
        t = jn(order, beta * np.sqrt(1 - (2 * u / span) ** 2)) / span
        return t * t.sum()   # scale only: the window is normalised afterwards
    return _window(W, oversample, taper)


def hanningsinc(W, a=None, oversample=5):
    """Sinc tapered by the raised cosine ``a + (1 - a) cos(pi x)``; ``a`` defaults to the tabulated optimum."""
    alpha = _default_parameter("hanning", _padded(W)) if a is None else a
    return _window(W, oversample, lambda u, span: alpha + (1 - alpha) * np.cos(2 * np.pi / span * u))


def pack_kernel(K, W, oversample=5):
    """Sample order (tap, phase) -> (phase, tap): the ``W + 2`` taps of one oversampling phase become contiguous,
    which is the order the degridder reads them in."""
    K = np.asarray(K)
    return np.ascontiguousarray(K.reshape(W + 2, oversample).T).reshape(-1)


def unpack_kernel(K, W, oversample=5):
    """Inverse of :func:`pack_kernel`: (phase, tap) -> (tap, phase)."""
    K = np.asarray(K)
    return np.ascontiguousarray(K.reshape(oversample, W + 2).T).reshape(-1)
