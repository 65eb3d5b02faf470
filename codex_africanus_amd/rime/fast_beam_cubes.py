"""beam_cube_dde / freq_grid_interp with the signatures of africanus/rime/fast_beam_cubes.py."""
import numpy as np

from .. import _lib
from .._device import Call, np_dtype_of


def _real_type(*arrays):
    """float32 only if every array is single precision, else float64."""
    dts = [np_dtype_of(a) for a in arrays]
    single = all(d in (np.dtype(np.float32), np.dtype(np.complex64)) for d in dts)
    return (np.float32, np.complex64) if single else (np.float64, np.complex128)


def freq_grid_interp(frequency, beam_freq_map):
    """``africanus.rime.fast_beam_cubes.freq_grid_interp`` (fast_beam_cubes.py:10-54):
    per channel (frequency scale, lower weight, lower grid position) -> (chan, 3)."""
    rt, _ = _real_type(frequency, beam_freq_map)
    fn = "af_freq_grid_interp_f32" if rt == np.float32 else "af_freq_grid_interp_f64"
    nchan, nud = int(frequency.shape[0]), int(beam_freq_map.shape[0])
    with Call(frequency, beam_freq_map) as c:
        p_fr, p_map = c.inp(frequency, rt), c.inp(beam_freq_map, rt)
        p_out, h = c.out((nchan, 3), rt)
        _lib.call(fn, p_fr, nchan, p_map, nud, p_out, c.stream)
        return c.result(h)


def beam_cube_dde(beam, beam_lm_extents, beam_freq_map, lm, parallactic_angles, point_errors,
                  antenna_scaling, frequency):
    """
    Per-antenna direction-dependent Jones terms from a complex beam cube.

    Same contract as ``africanus.rime.beam_cube_dde`` (africanus/rime/fast_beam_cubes.py:57-240):
    ``beam`` (beam_lw, beam_mh, beam_nud, corr...) complex, ``beam_lm_extents`` (2, 2),
    ``beam_freq_map`` (beam_nud,), ``lm`` (source, 2), ``parallactic_angles`` (time, ant),
    ``point_errors`` (time, ant, chan, 2), ``antenna_scaling`` (ant, chan, 2),
    ``frequency`` (chan,) -> (source, time, ant, chan, corr...) of ``beam``'s dtype.
    """
    if len(beam.shape) < 3:
        raise ValueError("beam must have at least 3 dimensions")
    beam_lw, beam_mh, beam_nud = (int(s) for s in beam.shape[:3])
    if beam_lw < 2 or beam_mh < 2 or beam_nud < 2:
        raise ValueError("beam_lw, beam_mh and beam_nud must be >= 2")
    corrs = tuple(int(s) for s in beam.shape[3:])
    ncorr = int(np.prod(corrs, dtype=np.int64)) if corrs else 1
    rt, ct = _real_type(beam, beam_lm_extents, beam_freq_map, lm, parallactic_angles, point_errors,
                        antenna_scaling, frequency)
    fn = "af_beam_cube_dde_c64" if rt == np.float32 else "af_beam_cube_dde_c128"
    nsrc = int(lm.shape[0])
    ntime, nant = (int(s) for s in parallactic_angles.shape)
    nchan = int(frequency.shape[0])
    if tuple(point_errors.shape) != (ntime, nant, nchan, 2):
        raise ValueError("point_errors must have shape (time, ant, chan, 2)")
    if tuple(antenna_scaling.shape) != (nant, nchan, 2):
        raise ValueError("antenna_scaling must have shape (ant, chan, 2)")
    if tuple(beam_freq_map.shape) != (beam_nud,):
        raise ValueError("beam_freq_map must have shape (beam_nud,)")
    with Call(beam, beam_lm_extents, beam_freq_map, lm, parallactic_angles, point_errors, antenna_scaling,
              frequency) as c:
        p_beam, p_ext, p_map = c.inp(beam, ct), c.inp(beam_lm_extents, rt), c.inp(beam_freq_map, rt)
        p_lm, p_pa = c.inp(lm, rt), c.inp(parallactic_angles, rt)
        p_pe, p_as, p_fr = c.inp(point_errors, rt), c.inp(antenna_scaling, rt), c.inp(frequency, rt)
        p_out, h = c.out((nsrc, ntime, nant, nchan) + corrs, ct)
        ws_bytes = int(_lib.load().af_beam_cube_dde_workspace_bytes(beam_lw, beam_mh, beam_nud, ncorr, ntime, nant,
                                                                    nchan, int(rt == np.float32)))
        p_ws = c.scratch(ws_bytes)
        _lib.call(fn, p_beam, beam_lw, beam_mh, beam_nud, ncorr, p_ext, p_map, p_lm, nsrc, p_pa, ntime, nant,
                  p_pe, p_as, p_fr, nchan, p_out, p_ws, max(ws_bytes, 256), c.stream)
        return c.result(h)
