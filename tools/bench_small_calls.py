#!/usr/bin/env python
"""Per-call latency at small (dask-chunk-like) shapes, where launch count and allocation matter more than kernel
time: C1 (10k rows x 16 chan x 100 src x 4 corr) and a 1k-row chunk.
  * device mode: torch ROCm tensors in, tensor out (torch's caching allocator, torch's stream);
  * host mode:   numpy in, numpy out -- what a dask block does.  AFHIP_POOL=0 re-creates round 1's path
                 (hipMalloc / hipFree per array per call, NULL stream); default = scratch pool + per-thread stream.
    python tools/bench_small_calls.py            # pool
    AFHIP_POOL=0 python tools/bench_small_calls.py   # before
"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import dft, rime, _lib
from codex_africanus_amd.testing import synthetic_inputs, real_image

dev = torch.device("cuda:0")
out = {"AFHIP_POOL": os.environ.get("AFHIP_POOL", "1")}
n = 200


def timeit(fn, sync):
    for _ in range(5):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e6


for name, nrow, nchan, nsrc in (("C1 10k x 16 x 100", 10000, 16, 100), ("chunk 1k x 64 x 100", 1000, 64, 100),
                                ("chunk 10k x 64 x 100", 10000, 64, 100)):
    d = synthetic_inputs(seed=0, nrow=nrow, nchan=nchan, nsrc=nsrc, nant=7)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    img = real_image(d)
    a = (T(img), T(d["uvw"]), T(d["lm"]), T(d["frequency"]))
    out["device im_to_vis " + name] = dict(us_per_call=timeit(lambda: dft.im_to_vis(*a), torch.cuda.synchronize))
    lm, uvw, fr = a[2][:16], a[1], a[3]
    out["device phase_delay(16 src) " + name] = dict(us_per_call=timeit(lambda: rime.phase_delay(lm, uvw, fr),
                                                                        torch.cuda.synchronize))
    h = (img, d["uvw"], d["lm"], d["frequency"])
    out["host im_to_vis " + name] = dict(us_per_call=timeit(lambda: dft.im_to_vis(*h), lambda: None))
    coh = (np.random.default_rng(0).standard_normal((8, nrow, nchan, 2, 2))
           + 1j * np.random.default_rng(1).standard_normal((8, nrow, nchan, 2, 2)))
    pv = lambda: rime.predict_vis(d["time_index"], d["ant1"], d["ant2"], None, coh, None, None, None, None)
    out["host predict_vis(8 src coh) " + name] = dict(us_per_call=timeit(pv, lambda: None))
out["pool_stats_device0"] = _lib.pool_stats(0)
out["pool_stats_pinned"] = _lib.pool_stats(-1)
print(json.dumps(out, indent=1))
