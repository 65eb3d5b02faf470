#!/usr/bin/env python
"""im_to_vis at C2's counts with 1 / 2 / 4 correlations (VALU kernels with wide tiles for 1 and 2, MFMA kernel for 4)."""
import sys, os, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import dft
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
nrow, nchan, nsrc = 1000000, 64, 1000
uvw = torch.from_numpy(np.c_[rng.uniform(-4000, 4000, nrow), rng.uniform(-4000, 4000, nrow), rng.uniform(-400, 400, nrow)]).to(dev)
lm = torch.from_numpy(rng.uniform(-0.03, 0.03, (nsrc, 2))).to(dev)
fr = torch.linspace(0.856e9, 1.712e9, nchan, dtype=torch.float64, device=dev)
out = {}
for nc in (1, 2, 4):
    img = torch.from_numpy(rng.standard_normal((nsrc, nchan, nc))).to(dev)
    f = lambda: dft.im_to_vis(img, uvw, lm, fr)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    out["ncorr=%d" % nc] = dict(ms=ms, Mvis_per_s=nrow * nchan / ms / 1e3)
print(json.dumps(out, indent=1))
