"""
Seeded synthetic inputs for the RIME predict hot path: lm in a
0.05 rad disc, MeerKAT-like uvw, linspace(0.856, 1.712) GHz, lognormal Stokes I
with small Q/U/V and linear feeds.  Counterpart of the reference's
``africanus/testing`` helpers; used by tests/, bench.py and the golden-vector
generator (tests/golden/make_golden.py draws the same numbers in the same
order, so fixtures only need to store seeds, sampled rows and checksums).
"""
import numpy as np


def synthetic_inputs(seed=0, nrow=10000, nchan=16, nsrc=100, nant=7):
    rng = np.random.default_rng(seed)
    rad = 0.05 * np.sqrt(rng.random(nsrc))
    ang = 2 * np.pi * rng.random(nsrc)
    lm = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1)
    uvw = np.empty((nrow, 3))
    uvw[:, 0] = rng.uniform(-4000, 4000, nrow)
    uvw[:, 1] = rng.uniform(-4000, 4000, nrow)
    uvw[:, 2] = rng.uniform(-400, 400, nrow)
    freq = np.linspace(0.856e9, 1.712e9, nchan)
    stokes_i = rng.lognormal(0.0, 1.0, nsrc)
    q, u, v = (0.1 * rng.standard_normal(nsrc) for _ in range(3))
    # linear feeds: [I+Q, U+iV, U-iV, I-Q]
    bright = np.stack([stokes_i + q, u + 1j * v, u - 1j * v, stokes_i - q], axis=1)
    nbl = nant * (nant - 1) // 2
    a1, a2 = np.triu_indices(nant, 1)
    ntime = -(-nrow // nbl)
    ant1 = np.tile(a1, ntime)[:nrow].astype(np.int32)
    ant2 = np.tile(a2, ntime)[:nrow].astype(np.int32)
    time_index = np.repeat(np.arange(ntime, dtype=np.int32), nbl)[:nrow]
    return dict(lm=lm, uvw=uvw, frequency=freq, brightness=bright, ant1=ant1, ant2=ant2,
                time_index=time_index, ntime=ntime, nant=nant, rng=rng)


def real_image(d, nchan=None):
    """The real image pattern [I+Q, U, U, I-Q] broadcast over channels
    (flat spectrum), shape (src, chan, 4) float64."""
    nchan = d["frequency"].shape[0] if nchan is None else nchan
    b = d["brightness"].real
    return np.ascontiguousarray(np.broadcast_to(b[:, None, :], (b.shape[0], nchan, 4)))
