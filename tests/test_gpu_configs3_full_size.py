"""
BASELINE configs[3] at ITS OWN size on the one device of the test box: 8e6 rows (3969 timesteps x 2016 baselines, the
last one short) x 64 channels x 1000 sources, rows cut into 8 timestep-aligned shards exactly as 8 ranks of one node
would hold them (`sharding.shard_bounds(nrow, 8, time_index)`: the reference's rule that a time never straddles two row
chunks, africanus/rime/dask_predict.py:343-367,494-499), every shard through the front-ends a rank runs --
`sharding.predict_shard` (the direct transform, BASELINE's headline kernel) and `sharding.fused_predict_shard` (beam
DDEs, Measurement-Set uvw: the GEMM form) -- with all eight outputs kept on the device (33 GB each way; one MI355X has
288 GB).  What a single box cannot show is the RCCL all-reduce between eight devices; everything else of the job runs.

Checks: every shard edge on a timestep boundary; sampled rows of shards 0 / 3 / 7 against the CPU oracle < 1e-8; shard k
equal, bit for bit, to the same rows of a 2-shard split; the sum of the eight chi^2 partials equal to one pass over the
concatenation to 1e-12 (float atomics: the order of the adds differs, nothing else).
"""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NROW, NCHAN, NSRC, NANT, WORLD = 8000000, 64, 1000, 64, 8
NBL = NANT * (NANT - 1) // 2
NTIME = -(-NROW // NBL)


def _free():
    import torch
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def _layout():
    a1, a2 = np.triu_indices(NANT, 1)
    ant1 = np.tile(a1, NTIME)[:NROW].astype(np.int32)
    ant2 = np.tile(a2, NTIME)[:NROW].astype(np.int32)
    time_index = np.repeat(np.arange(NTIME, dtype=np.int32), NBL)[:NROW]
    return time_index, ant1, ant2


def _check_bounds(bounds, time_index):
    from codex_africanus_amd import sharding
    assert bounds[0][0] == 0 and bounds[-1][1] == NROW
    for (a, b), (c, d) in zip(bounds[:-1], bounds[1:]):
        assert b == c and b % NBL == 0                           # contiguous, every inner edge starts a timestep
        assert time_index[b - 1] != time_index[b]
    sizes = [b - a for a, b in bounds]
    assert max(sizes) - min(sizes) <= 2 * NBL and min(sizes) > 990000
    spans = [sharding.time_slice(time_index, a, b) for a, b in bounds]
    assert spans[0][0] == 0 and spans[-1][1] == NTIME
    assert all(s[1] == t[0] for s, t in zip(spans[:-1], spans[1:]))     # the time slices tile [0, ntime) without overlap


def test_direct_transform_8e6_rows_in_eight_shards():
    import torch
    import oracle
    from codex_africanus_amd import sharding
    from codex_africanus_amd.testing import synthetic_inputs, real_image
    assert NTIME == 3969
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = synthetic_inputs(seed=0, nrow=16, nchan=NCHAN, nsrc=NSRC, nant=NANT)
    image = real_image(d)
    time_index, _, _ = _layout()
    rng = np.random.default_rng(31)
    uvw = np.empty((NROW, 3))
    uvw[:, 0] = rng.uniform(-4000, 4000, NROW)
    uvw[:, 1] = rng.uniform(-4000, 4000, NROW)
    uvw[:, 2] = rng.uniform(-400, 400, NROW)
    bounds = sharding.shard_bounds(NROW, WORLD, time_index)
    _check_bounds(bounds, time_index)
    d_img, d_uvw, d_lm, d_fr = t(image), t(uvw), t(d["lm"]), t(d["frequency"])
    # "observed" data of every shard: its model + a fixed offset, so that chi^2 is known in closed form as well
    shards, partial = [], torch.zeros(NCHAN, dtype=torch.float64, device=dev)
    for k in range(WORLD):
        vis0, _, bk = sharding.predict_shard(k, WORLD, d_img, d_uvw, d_lm, d_fr, time_index=time_index)
        assert bk == bounds[k] and tuple(vis0.shape) == (bk[1] - bk[0], NCHAN, 4)
        data = vis0 + 0.01
        vis, c2, _ = sharding.predict_shard(k, WORLD, d_img, d_uvw, d_lm, d_fr, data=data, time_index=time_index)
        assert torch.equal(vis, vis0)                        # the chi^2 epilogue changes no visibility
        # |data - vis|^2 = 1e-4 per correlation up to the rounding of vis + 0.01 (|vis| ~ 1e3: ~1e-13 absolute)
        expect = 4 * (bk[1] - bk[0]) * 1e-4
        assert np.allclose(c2.cpu().numpy(), expect, rtol=1e-6)
        partial += c2
        shards.append((vis, data))
        del vis0
    # one pass over the concatenation of the eight shards (33 GB model + 33 GB data on the device)
    model = torch.cat([s[0] for s in shards])
    obs = torch.cat([s[1] for s in shards])
    assert tuple(model.shape) == (NROW, NCHAN, 4)
    whole = sharding.chi2(model, obs)
    assert torch.allclose(whole, partial, rtol=1e-12, atol=0)
    del obs
    shards = [s[0] for s in shards]
    _free()
    # sampled rows of shards 0 / 3 / 7 against the oracle (the reference's loop order), north-star tolerance
    for k in (0, 3, 7):
        a, b = bounds[k]
        rows = np.unique(np.concatenate([[a, b - 1], rng.integers(a, b, 30)]))
        ref = oracle.im_to_vis(image, uvw[rows], d["lm"], d["frequency"], omp=True)
        got = shards[k][torch.from_numpy(rows - a).to(dev)].cpu().numpy()
        err = np.abs(got - ref).max()
        assert err < 1e-8, (k, err)
    # shard k of 8 == the same rows of a 2-shard split, bit for bit
    b2 = sharding.shard_bounds(NROW, 2, time_index)
    assert b2[0][1] == bounds[3][1]                           # the halves meet where shards 3 and 4 meet
    for h in range(2):
        half, _, bh = sharding.predict_shard(h, 2, d_img, d_uvw, d_lm, d_fr, time_index=time_index)
        for k in range(4 * h, 4 * h + 4):
            a, b = bounds[k]
            assert torch.equal(half[a - bh[0]:b - bh[0]], shards[k]), k
        del half
        _free()
    del model, shards
    _free()


def test_fused_predict_8e6_rows_in_eight_shards():
    """the same job with per-antenna beam-cube DDEs (the only predict of the reference chain that exists at 1000 sources:
    its coherencies would be 33 TB): Measurement-Set uvw, the GEMM form on every shard, the rank's slice of every
    (time, ant, ...) array cut by time_slice of its rows"""
    import torch
    from benchlib.workloads_fused import FusedDde
    from codex_africanus_amd import sharding, _lib
    from codex_africanus_amd.rime import fused
    import argparse
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = argparse.Namespace(rows=NROW, chans=NCHAN, sources=NSRC, seed=0, workload="fused_dde_ant", pa="random",
                              uvw="antennas", check_rows=0)
    wl = FusedDde(args, 0, dev, _lib.load(), _lib, t)        # bench.py's inputs at 8e6 rows (plans made on the host)
    h, v = wl.h, wl.dv
    assert wl.ntime == NTIME and h["time_index"].shape == (NROW,)
    bounds = sharding.shard_bounds(NROW, WORLD, h["time_index"])
    _check_bounds(bounds, h["time_index"])
    d_ti = t(h["time_index"])
    args8 = (d_ti, v["a1"], v["a2"], v["lm"], v["uvw"], v["freq"], v["X"], v["beam"], v["ext"], v["fmap"], v["pa"],
             v["pe"], v["asc"])
    shards, partial = [], torch.zeros(NCHAN, dtype=torch.float64, device=dev)
    for k in range(WORLD):
        a, b = bounds[k]
        # a rank's own arrays: its rows and its timesteps only (bounds=...), as a job that cannot hold 8e6 rows in one
        # place feeds its ranks
        t0, t1 = sharding.time_slice(h["time_index"], a, b)
        own = (d_ti[a:b], v["a1"][a:b], v["a2"][a:b], v["lm"], v["uvw"][a:b], v["freq"], v["X"], v["beam"], v["ext"],
               v["fmap"], v["pa"][t0:t1], v["pe"][t0:t1], v["asc"])
        vis, _, bk = sharding.fused_predict_shard(k, WORLD, *own, bounds=(a, b))
        assert bk == (a, b) and tuple(vis.shape) == (b - a, NCHAN, 2, 2)
        data = vis + 0.01
        _, c2, _ = sharding.fused_predict_shard(k, WORLD, *own, bounds=(a, b), data=data)
        partial += c2
        shards.append((vis, data))
    import codex_africanus_amd as af
    af.check_status()                                         # no stale plan, no index out of range anywhere
    plan = fused.cached_plan(d_ti[bounds[7][0]:], v["a1"][bounds[7][0]:], v["a2"][bounds[7][0]:], NANT,
                             uvw=v["uvw"][bounds[7][0]:])
    assert plan.decomposable and plan.fill > 0.8              # the shards took the GEMM form
    model = torch.cat([s[0] for s in shards])
    obs = torch.cat([s[1] for s in shards])
    whole = sharding.chi2(model, obs)
    assert torch.allclose(whole, partial, rtol=1e-12, atol=0)
    del obs, model
    shards = [s[0] for s in shards]
    _free()
    # the oracle chain on rows of the first timestep of shard 0, a middle one of shard 3, the last (short) one of shard 7
    for k, tstep in ((0, 0), (3, h["time_index"][(bounds[3][0] + bounds[3][1]) // 2]), (7, NTIME - 1)):
        a, b = bounds[k]
        lo, hi = int(tstep) * NBL, min((int(tstep) + 1) * NBL, NROW)
        assert a <= lo and hi <= b
        rows = np.unique(np.linspace(lo, hi - 1, 8).astype(np.int64))
        ref = wl._chain(rows).reshape(len(rows), NCHAN, 2, 2)
        got = shards[k][torch.from_numpy(rows - a).to(dev)].cpu().numpy()
        err = np.abs(got - ref).max()
        assert err < 1e-8, (k, err)
    # full-length arrays on every "rank" (bounds computed inside), 2-shard split: shard k of 8 is the same bits
    for hh in range(2):
        half, _, bh = sharding.fused_predict_shard(hh, 2, *args8)
        for k in range(4 * hh, 4 * hh + 4):
            a, b = bounds[k]
            assert torch.equal(half[a - bh[0]:b - bh[0]], shards[k]), k
        del half
        _free()
    del shards
    _free()
