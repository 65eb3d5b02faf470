#!/usr/bin/env python
"""wsclean_predict on one MI355X: device-resident inputs, HIP-event timing on torch's stream.
Workload: C2's row/channel/component counts (1e6 rows x 64 chan x 1000 components), all points, half
Gaussians, all Gaussians; fp64 op model per (row, component, tile of CT channels):
  setup 2 sincos (~75 ops) + 4 ops per channel (point) or 8 (Gaussian, + 3 exp ~ 90 ops per tile)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import rime, dft
import oracle

dev = torch.device("cuda:0")
nrow, nchan, nsrc = int(os.environ.get("NROW", 1000000)), 64, 1000
rs = np.random.RandomState(0)
uvw = rs.normal(size=(nrow, 3)) * 2000.0
lm = rs.normal(size=(nsrc, 2)) * 1e-2
flux, coeffs = rs.uniform(0.1, 2, nsrc), rs.normal(size=(nsrc, 2)) * [0.7, 0.1]
log_poly = rs.randint(0, 2, nsrc).astype(bool)
gshape = np.stack([rs.uniform(0, 3e-4, nsrc), rs.uniform(0, 2e-4, nsrc), rs.uniform(0, np.pi, nsrc)], axis=1)
freq = np.linspace(0.856e9, 1.712e9, nchan)
ref_freq = np.full(nsrc, 1.284e9)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
out = {}
for name, frac in (("all points", 0.0), ("half Gaussians", 0.5), ("all Gaussians", 1.0)):
    isg = np.arange(nsrc) < int(frac * nsrc)
    rs.shuffle(isg)
    args = [T(uvw), T(lm), T(isg), T(flux), T(coeffs), T(log_poly), T(ref_freq), T(gshape), T(freq)]
    for mode in ("recurrence", "exact"):
        if mode == "exact" and frac != 0.5:
            continue
        dft.set_mode(mode)
        if mode == "exact":
            args[0] = args[0][:100000]
        fn = lambda: rime.wsclean_predict(*args)
        vis = fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        n = args[0].shape[0]
        # error against the oracle on 64 sampled rows
        idx = np.linspace(0, n - 1, 64).astype(int)
        st = np.where(isg, "GAUSSIAN", "POINT")
        ref = oracle.wsclean_predict(uvw[idx], lm, st, flux, coeffs, log_poly, ref_freq, gshape, freq)
        err = float(np.abs(vis[torch.from_numpy(idx).to(dev)].cpu().numpy() - ref).max())
        out["%s / %s" % (name, mode)] = dict(ms=ms, rows=int(n), Mvis_per_s=n * nchan / ms / 1e3,
                                              G_row_comp_chan_per_s=n * nchan * nsrc / ms / 1e6, max_abs_err=err)
print(json.dumps(out, indent=1))
