"""CPU-side tests of the chunk contract (errors, chunk arithmetic) and the row sharding,
including a world-size-2 gloo all-reduce of the chi-squared vector."""
import os
import socket

import numpy as np
import pytest

from codex_africanus_amd import chunked, sharding


def test_normalize_chunks():
    assert chunked.normalize_chunks(None, 10) == (10,)
    assert chunked.normalize_chunks(4, 10) == (4, 4, 2)
    assert chunked.normalize_chunks((4, 4, 2), 10) == (4, 4, 2)
    with pytest.raises(ValueError):
        chunked.normalize_chunks((4, 4), 10)


def z(*shape):
    return np.zeros(shape, np.complex128)


IDX = (np.zeros(10, np.int32),) * 3


def test_predict_vis_row_time_chunk_count_mismatch():
    """africanus/rime/dask_predict.py:494-499."""
    with pytest.raises(ValueError, match="Number of row chunks"):
        chunked.predict_vis(*IDX, dde1_jones=z(3, 4, 4, 5, 2, 2), source_coh=z(3, 10, 5, 2, 2),
                            dde2_jones=z(3, 4, 4, 5, 2, 2),
                            chunks={"row": (4, 4, 2), "time": (2, 2)})
    with pytest.raises(ValueError, match="Number of row chunks"):
        chunked.predict_vis(*IDX, die1_jones=z(4, 4, 5, 2, 2), base_vis=z(10, 5, 2, 2),
                            die2_jones=z(4, 4, 5, 2, 2), chunks={"row": (5, 5), "time": (4,)})


def test_predict_vis_antenna_chunking_refused():
    """africanus/rime/dask_predict.py:478-489."""
    with pytest.raises(ValueError, match="Subdivision of antenna dimension"):
        chunked.predict_vis(*IDX, dde1_jones=z(3, 4, 4, 5, 2, 2), dde2_jones=z(3, 4, 4, 5, 2, 2),
                            chunks={"ant": (2, 2)})


def test_predict_vis_inherits_predict_checks():
    with pytest.raises(ValueError, match="Both dde1_jones and dde2_jones"):
        chunked.predict_vis(*IDX, dde1_jones=z(3, 4, 4, 5, 2, 2))


def test_im_to_vis_source_axis_single_chunk():
    """africanus/dft/dask.py:29-36."""
    with pytest.raises(ValueError, match="lm chunks must match lm shape"):
        chunked.im_to_vis(np.zeros((6, 4, 2)), np.zeros((5, 3)), np.zeros((6, 2)), np.ones(4),
                          chunks={"source": (3, 3)})


# ----------------------------------------------------------------------------- sharding
def test_shard_bounds_plain():
    b = sharding.shard_bounds(10, 3)
    assert b == [(0, 3), (3, 6), (6, 10)]
    assert sharding.shard_bounds(0, 2) == [(0, 0), (0, 0)]
    assert sharding.shard_bounds(5, 8)[-1][1] == 5


def test_shard_bounds_on_timestep_boundaries():
    nbl, ntime = 21, 50
    ti = np.repeat(np.arange(ntime), nbl)
    for world in (2, 3, 4, 8):
        b = sharding.shard_bounds(ti.shape[0], world, ti)
        assert b[0][0] == 0 and b[-1][1] == ti.shape[0]
        for (s0, e0), (s1, e1) in zip(b[:-1], b[1:]):
            assert e0 == s1
        for (s, e) in b:
            assert s % nbl == 0 and e % nbl == 0          # whole timesteps only
            if e > s:
                t0, t1 = sharding.time_slice(ti, s, e)
                assert (t0, t1) == (s // nbl, e // nbl)
        sizes = [e - s for s, e in b]
        assert max(sizes) - min(sizes) <= 2 * nbl
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, np.array([0, 1, 0, 1]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gloo_worker(rank, world, port, nchan, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # each rank's shard of a fake residual: chi2 of its own rows (numpy stands in for the kernel)
        rng = np.random.default_rng(5)
        resid = rng.standard_normal((40, nchan, 4)) + 1j * rng.standard_normal((40, nchan, 4))
        (s, e) = sharding.shard_bounds(40, world)[rank]
        part = torch.from_numpy((np.abs(resid[s:e]) ** 2).sum(axis=(0, 2)))
        total = sharding.allreduce_chi2(part.clone())
        expect = (np.abs(resid) ** 2).sum(axis=(0, 2))
        q.put((rank, np.allclose(total.numpy(), expect, rtol=1e-13), float(total.sum())))
    finally:
        dist.destroy_process_group()


def test_chi2_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, 16, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert abs(res[0][2] - res[1][2]) == 0.0     # every rank holds the same reduced vector


def _gloo_image_worker(rank, world, port, q):
    """Row-sharded adjoint transform: every rank images its own row block (the CPU oracle stands in for the kernel),
    allreduce_image sums the (source, chan, corr) images -- the one data-path collective of vis_to_im."""
    import torch
    import torch.distributed as dist
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)
        nrow, nchan, nsrc = 90, 5, 7
        uvw = rng.standard_normal((nrow, 3)) * 500.0
        lm = rng.standard_normal((nsrc, 2)) * 0.02
        freq = np.linspace(1.0e9, 1.1e9, nchan)
        vis = rng.standard_normal((nrow, nchan, 2)) + 1j * rng.standard_normal((nrow, nchan, 2))
        flags = rng.random((nrow, nchan, 2)) < 0.1
        (s, e) = sharding.shard_bounds(nrow, world)[rank]
        part = torch.from_numpy(oracle.vis_to_im(vis[s:e], uvw[s:e], lm, freq, flags[s:e]))
        total = sharding.allreduce_image(part.clone()).numpy()
        full = oracle.vis_to_im(vis, uvw, lm, freq, flags)
        q.put((rank, float(np.abs(total - full).max()), float(np.abs(full).max()), float(total.sum())))
    finally:
        dist.destroy_process_group()


def test_vis_to_im_row_shards_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_image_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, err, scale, _ in res:
        assert err <= 1e-12 * scale        # shard sums reassociate the row sum: rounding only
    assert res[0][3] == res[1][3]          # every rank holds the same reduced image


def test_allreduce_is_noop_without_process_group():
    import torch
    x = torch.arange(4, dtype=torch.float64)
    assert sharding.allreduce_chi2(x.clone()).equal(x)


# ------------------------------------------------------------------------------ block -> GPU placement
def test_placement_row_blocks_round_robin_over_devices():
    """row block k -> devices[k % n] (north star: dask row chunks map to the GPUs of one node); arithmetic only"""
    from codex_africanus_amd import placement
    devs = tuple(range(8))
    assert [placement.device_for_block(k, devs) for k in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert [placement.device_for_block(k, (3, 5)) for k in range(5)] == [3, 5, 3, 5, 3]
    # 8e6 rows in 1e6-row chunks (BASELINE configs[3]): one chunk per GPU, none shared, none idle
    assert sorted(placement.device_for_block(k, devs) for k in range(8)) == list(devs)
    assert placement.parse_device_list(None, 4) == (0, 1, 2, 3)
    assert placement.parse_device_list("2,0", 4) == (2, 0)
    for bad in ("4", "0,0", "-1", ","):
        with pytest.raises(ValueError):
            placement.parse_device_list(bad, 4)


def test_placement_block_context_and_policies():
    from codex_africanus_amd import placement
    devs = (0, 1, 2, 3)
    assert placement.choose(devs=devs, policy="none") is None
    with placement.block(np.array([6])):                 # the one-element array dask hands a block function
        assert placement.choose(devs=devs, policy="block") == 2
        with placement.block(1):
            assert placement.choose(devs=devs, policy="block") == 1
        assert placement.choose(devs=devs, policy="block") == 2
        # policy 'thread' ignores block indices
        assert placement.choose(devs=devs, policy="thread") == placement.device_for_thread(devs)
    # without a block index the default policy leaves the thread's current device alone (a plain call must
    # not override the caller's af_set_device / a rank's own device); per-thread devices are opt-in
    assert placement.choose(devs=devs, policy="block") is None
    assert placement.choose(devs=devs, policy="thread") == placement.device_for_thread(devs)


def test_placement_threads_cover_all_devices():
    """N >= n_devices worker threads drive every device: threads are numbered in order of first use"""
    import threading
    from codex_africanus_amd import placement
    devs = tuple(range(4))
    seen, lock, gate = [], threading.Lock(), threading.Barrier(8)

    def worker():
        gate.wait()
        d = placement.device_for_thread(devs)
        assert placement.device_for_thread(devs) == d            # sticky
        with lock:
            seen.append(d)
    ts = [threading.Thread(target=worker) for _ in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert sorted(set(seen)) == list(devs) and all(seen.count(d) == 2 for d in devs)


def _placement_rank(rank, world, port, q):
    """world-2 check: each rank owns its shard's row blocks; block -> device is the same arithmetic on both"""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from codex_africanus_amd import placement, sharding
    import torch
    bounds = sharding.shard_bounds(8000, world)
    lo, hi = bounds[rank]
    blocks = list(range(lo // 1000, hi // 1000))                 # 1000-row dask chunks of this rank's shard
    mine = torch.tensor([placement.device_for_block(k, tuple(range(8))) for k in blocks])
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    q.put((rank, torch.cat(gathered).tolist()))
    dist.destroy_process_group()


def test_placement_world2_gloo():
    import multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_placement_rank, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    assert res[0] == res[1] == list(range(8))            # 8 row chunks over 2 ranks -> 8 distinct devices, in order
