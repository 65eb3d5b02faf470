"""
Calls whose arrays cross 2^31 elements (VERDICT r4 "missing" 2).  The reference has no size limit
(africanus/rime/phase.py:36-61, africanus/dft/kernels.py:45-67, africanus/rime/predict.py:199-212 index with Python
integers); on a 288 GB device 41 GB arrays are ordinary requests, and every flat index of the kernels must be 64-bit.
One call per hot function, device resident, sampled elements -- the LAST ones first: that is where a 32-bit index
wraps -- against the CPU oracle:

    phase_delay   40 sources x 1e6 rows x 64 chan       = 2.56e9 complex128 (41 GB)
    im_to_vis     9e6 rows x 64 chan x 4 corr           = 2.30e9 complex128 (37 GB)
    predict_vis   coherencies 10 x 1e6 x 64 x 2 x 2     = 2.56e9 complex128 (41 GB), with DIEs and base_vis
"""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free():
    import torch
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def _inputs(nrow, nsrc, nchan=64, nant=64):
    from codex_africanus_amd.testing import synthetic_inputs
    d = synthetic_inputs(seed=5, nrow=16, nchan=nchan, nsrc=nsrc, nant=nant)
    rng = np.random.default_rng(77)
    uvw = np.empty((nrow, 3))
    uvw[:, 0] = rng.uniform(-4000, 4000, nrow)
    uvw[:, 1] = rng.uniform(-4000, 4000, nrow)
    uvw[:, 2] = rng.uniform(-400, 400, nrow)
    return d, uvw, rng


def test_phase_delay_2_56e9_elements():
    import torch
    import oracle
    from codex_africanus_amd import rime
    nsrc, nrow, nchan = 40, 1000000, 64
    assert nsrc * nrow * nchan > 2 ** 31
    d, uvw, rng = _inputs(nrow, nsrc)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = rime.phase_delay(t(d["lm"]), t(uvw), t(d["frequency"]))
    assert tuple(out.shape) == (nsrc, nrow, nchan) and out.dtype == torch.complex128
    for s in (nsrc - 1, 33, 34, 0):       # source 33 / 34: flat index 2^31 = (33, 554432, 0) lies between them
        rows = np.unique(np.concatenate([[0, nrow - 1, 554431, 554432, 554433], rng.integers(0, nrow, 40)]))
        ref = oracle.phase_delay(d["lm"][s:s + 1], uvw[rows], d["frequency"])[0]
        got = out[s][torch.from_numpy(rows).to(dev)].cpu().numpy()
        assert np.abs(got - ref).max() < 1e-12, s
    # nothing was left unwritten anywhere: a phasor has modulus 1 (41 GB scanned on the device)
    assert float((out.abs() - 1.0).abs().max()) < 1e-14
    del out
    _free()


def test_im_to_vis_9e6_rows():
    import torch
    import oracle
    from codex_africanus_amd import dft
    from codex_africanus_amd.testing import real_image
    nrow, nsrc, nchan = 9000000, 300, 64
    assert nrow * nchan * 4 > 2 ** 31
    d, uvw, rng = _inputs(nrow, nsrc)
    image = real_image(d)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for img in (image, image * (1 + 0.25j)):                   # the real-image and the complex-image kernels
        vis = dft.im_to_vis(t(img), t(uvw), t(d["lm"]), t(d["frequency"]))
        assert tuple(vis.shape) == (nrow, nchan, 4)
        # flat element 2^31 is row 8388608; rows around it, the first and the last ones, and a random sample
        rows = np.unique(np.concatenate([[0, 1, nrow - 2, nrow - 1], np.arange(8388600, 8388616), rng.integers(0, nrow, 40)]))
        ref = oracle.im_to_vis(img, uvw[rows], d["lm"], d["frequency"], omp=True)
        got = vis[torch.from_numpy(rows).to(dev)].cpu().numpy()
        assert np.abs(got - ref).max() < 1e-8
        # every row written: a block of rows that a wrapped index skipped would still hold torch.empty's garbage or
        # another block's values; compare a strided quarter of a million rows against a second call on just those rows
        sel = torch.arange(nrow - 250000 * 36, nrow, 36, device=dev)
        again = dft.im_to_vis(t(img), t(uvw)[sel], t(d["lm"]), t(d["frequency"]))
        assert torch.equal(again, vis[sel])
        del vis, again
        _free()


def test_predict_vis_coherencies_beyond_2_31():
    import torch
    import oracle
    from codex_africanus_amd import rime
    nsrc, nrow, nchan, nant = 10, 1000000, 64, 64
    assert nsrc * nrow * nchan * 4 > 2 ** 31
    d, uvw, rng = _inputs(nrow, nsrc)
    nbl = nant * (nant - 1) // 2
    ntime = -(-nrow // nbl)
    a1, a2 = np.triu_indices(nant, 1)
    ant1, ant2 = np.tile(a1, ntime)[:nrow].astype(np.int32), np.tile(a2, ntime)[:nrow].astype(np.int32)
    time_index = np.repeat(np.arange(ntime, dtype=np.int32), nbl)[:nrow]
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_uvw = t(uvw)
    phase = rime.phase_delay(t(d["lm"]), d_uvw, t(d["frequency"]))                   # (10, 1e6, 64)
    # the caller's einsum("srf,si->srfi") (africanus/rime/examples/predict.py:107-134), on the device: plumbing
    coh = (phase[:, :, :, None] * t(d["brightness"])[:, None, None, :]).reshape(nsrc, nrow, nchan, 2, 2).contiguous()
    del phase
    assert coh.numel() > 2 ** 31
    die = 1.0 + 0.1 * (rng.standard_normal((ntime, nant, nchan, 2, 2)) + 1j * rng.standard_normal((ntime, nant, nchan, 2, 2)))
    bvis = 0.1 * (rng.standard_normal((nrow, nchan, 2, 2)) + 1j * rng.standard_normal((nrow, nchan, 2, 2)))
    d_ti, d_a1, d_a2, d_die, d_bvis = t(time_index), t(ant1), t(ant2), t(die), t(bvis)
    # flat element 2^31 of the coherencies is (source 8, row 388608); sources 9 start beyond it
    rows = np.unique(np.concatenate([[0, nrow - 1], rng.integers(0, nrow, 30), np.arange(388604, 388612)]))
    d_rows = torch.from_numpy(rows).to(dev)
    coh_rows = coh[:, d_rows].cpu().numpy()
    tsel, tinv = np.unique(time_index[rows], return_inverse=True)
    for with_die in (False, True):
        out = rime.predict_vis(d_ti, d_a1, d_a2, None, coh, None, d_die if with_die else None,
                               d_bvis if with_die else None, d_die if with_die else None)
        assert tuple(out.shape) == (nrow, nchan, 2, 2)
        ref = oracle.predict_vis(tinv, ant1[rows], ant2[rows], None, coh_rows, None, die[tsel] if with_die else None,
                                 bvis[rows] if with_die else None, die[tsel] if with_die else None)
        got = out[d_rows].cpu().numpy()
        assert np.array_equal(got, ref)                          # the API kernels reproduce the reference bit for bit
        if not with_die:
            # the whole output against a torch reduction over sources of the same 41 GB (association differs: ~1e-13)
            tot = coh.sum(dim=0)
            assert float((out - tot).abs().max()) < 1e-11
            del tot
        del out
        _free()
    import codex_africanus_amd as af
    af.check_status()
    del coh
    _free()
