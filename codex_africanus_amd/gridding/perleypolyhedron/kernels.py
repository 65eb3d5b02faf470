"""Anti-aliasing kernels of the Perley-polyhedron (de)gridder, numpy host utilities with the signatures of
africanus/gridding/perleypolyhedron/kernels.py:16-127 (uspace, sinc, kbsinc, hanningsinc, pack_kernel,
unpack_kernel).  A kernel is a 1-D oversampled window of ``oversample * (W + 2)`` taps; these few hundred values
are set-up data for the degridder, generated on the host exactly as the reference does."""
import numpy as np


def uspace(W, oversample):
    """Tap positions ``|...|...|`` of a W-tap kernel, one pad tap either side (kernels.py:16-33)."""
    assert W % 2 == 1, "the taps must be centred on the origin"
    return np.arange(oversample * (W + 2)) / float(oversample) - (W + 2) // 2


def sinc(W, oversample=5, a=1.0):
    """Oversampled sinc window, unit sum (kernels.py:36-42)."""
    res = np.sinc(uspace(W, oversample) * a)
    return res / np.sum(res)


_KBSINC_AUTOCOEFFS = np.polyfit([1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0],
                                [1.9980, 2.3934, 3.3800, 4.2054, 4.9107, 5.7567, 6.6291, 7.4302], 1)


def kbsinc(W, b=None, oversample=5, order=15):
    """Kaiser-Bessel windowed sinc with a high-order Bessel function (kernels.py:52-67); needs scipy."""
    from scipy.special import jn
    if b is None:
        b = np.poly1d(_KBSINC_AUTOCOEFFS)((W + 2))
    u = uspace(W, oversample)
    wnd = jn(order, b * np.sqrt(1 - (2 * u / ((W + 2) + 1)) ** 2)) * 1 / ((W + 2) + 1)
    res = sinc(W, oversample=oversample) * wnd * np.sum(wnd)
    return res / np.sum(res)


_HANNING_AUTOCOEFFS = np.polyfit([1.5, 2.0, 2.5, 3.0, 3.5], [0.7600, 0.7146, 0.6185, 0.5534, 0.5185], 3)


def hanningsinc(W, a=None, oversample=5):
    """Hanning windowed sinc (kernels.py:75-85)."""
    if a is None:
        a = np.poly1d(_HANNING_AUTOCOEFFS)((W + 2))
    u = uspace(W, oversample)
    wnd = a + (1 - a) * np.cos(2 * np.pi / ((W + 2) + 1) * u)
    res = sinc(W, oversample=oversample) * wnd
    return res / np.sum(res)


def pack_kernel(K, W, oversample=5):
    """Regroup the taps by oversampling phase: [phase 0 taps | phase 1 taps | ...] (kernels.py:88-102)."""
    K = np.asarray(K)
    return np.ascontiguousarray(K.reshape(W + 2, oversample).T).reshape(-1)


def unpack_kernel(K, W, oversample=5):
    """Inverse of :func:`pack_kernel` (kernels.py:105-119)."""
    K = np.asarray(K)
    return np.ascontiguousarray(K.reshape(oversample, W + 2).T).reshape(-1)
