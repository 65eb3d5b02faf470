#!/usr/bin/env python
"""PCIe-inclusive rate of the numpy-in / numpy-out FUSED predict (BASELINE configs[2] counts, Measurement-Set uvw: the GEMM
form): plan on the host, upload, kernels, 4.1 GB download -- in one piece (AFHIP_D2H_PIPELINE=0) and in timestep-aligned
chunks whose downloads overlap the next chunk's kernels (the default)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codex_africanus_amd import _lib, rime
from benchlib.workloads_fused import FusedDde

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
args = argparse.Namespace(rows=1000000, chans=64, sources=1000, seed=0, workload="fused_dde_ant", pa="random", uvw="antennas", antennas=64)
wl = FusedDde(args, 0, dev, _lib.load(), _lib, t)
h = wl.h
call = lambda: rime.fused_predict_vis(h["time_index"], h["ant1"], h["ant2"], h["lm"], h["uvw"], h["freq"], h["X"], h["beam"],
                                      h["extents"], h["beam_freq_map"], h["pa"], h["pe"], h["asc"])
res = {}
for mode in ("0", "1"):
    os.environ["AFHIP_D2H_PIPELINE"] = mode
    ts = []
    for k in range(4):
        t0 = time.perf_counter(); vis = call(); ts.append(time.perf_counter() - t0)
    res["pipeline_" + mode] = {"seconds": ts, "best_ms": 1e3 * min(ts[1:]), "Mvis_s": 1e6 * 64 / min(ts[1:]) / 1e6}
    if mode == "0":
        ref = vis.copy()
    else:
        res["bit_equal"] = bool(np.array_equal(vis, ref))
print(json.dumps(res))
