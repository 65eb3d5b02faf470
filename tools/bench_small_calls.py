#!/usr/bin/env python
"""Per-call latency of the device-resident entry points at small (dask-chunk-like) shapes, where launch count
matters more than kernel time: C1 (10k rows x 16 chan x 100 src x 4 corr) and a 1k-row chunk."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import dft, rime
from codex_africanus_amd.testing import synthetic_inputs, real_image

dev = torch.device("cuda:0")
out = {}
for name, nrow, nchan, nsrc in (("C1 10k x 16 x 100", 10000, 16, 100), ("chunk 1k x 64 x 100", 1000, 64, 100)):
    d = synthetic_inputs(seed=0, nrow=nrow, nchan=nchan, nsrc=nsrc, nant=7)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    a = (T(real_image(d)), T(d["uvw"]), T(d["lm"]), T(d["frequency"]))
    for _ in range(3):
        dft.im_to_vis(*a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        dft.im_to_vis(*a)
    torch.cuda.synchronize()
    out["im_to_vis " + name] = dict(us_per_call=(time.perf_counter() - t0) / n * 1e6)
    lm, uvw, fr = a[2][:16], a[1], a[3]
    for _ in range(3):
        rime.phase_delay(lm, uvw, fr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        rime.phase_delay(lm, uvw, fr)
    torch.cuda.synchronize()
    out["phase_delay(16 src) " + name] = dict(us_per_call=(time.perf_counter() - t0) / n * 1e6)
print(json.dumps(out, indent=1))
