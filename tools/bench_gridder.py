#!/usr/bin/env python
"""Convolutional gridding (the adjoint of BASELINE configs[4]) on one MI355X: 1e6 rows x 64 chan onto a 4096^2
grid, 7x7 taps, Stokes I from XX,YY; device-resident, HIP-event timing on torch's stream."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd.gridding.perleypolyhedron import kernels
from codex_africanus_amd.gridding.perleypolyhedron.gridder import gridder

dev = torch.device("cuda:0")
npix, nrow, nchan, W, OS = 4096, int(os.environ.get("NROW", 1000000)), 64, 7, 63
cell = 2.0
wl = 299792458.0 / np.linspace(0.856e9, 1.712e9, nchan)
rng = np.random.default_rng(0)
umax = 0.45 / np.deg2rad(cell / 3600.0) * wl.min()
uvw = np.zeros((nrow, 3))
uvw[:, :2] = rng.uniform(-1, 1, (nrow, 2)) * umax
uvw[:, 2] = rng.uniform(-400, 400, nrow)
vis = torch.randn(nrow, nchan, 2, dtype=torch.complex128, device=dev)
k = kernels.pack_kernel(kernels.kbsinc(W, oversample=OS), W, OS)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
args = [T(uvw), vis, T(wl), T(np.zeros(nchan, np.int64)), npix, cell, (0.0, 0.0), (0.0, 0.0), T(k), W, OS, "None", "None",
        "I_FROM_XXYY", "conv_1d_axisymmetric_packed_scatter"]
g = gridder(*args); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    gridder(*args)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
taps = nrow * nchan * W * W
print(json.dumps(dict(ms=ms, Mvis_per_s=nrow * nchan / ms / 1e3, Gtaps_per_s=taps / ms / 1e6,
                      atomic_TBs=taps * 16 / ms / 1e9, grid_abs_sum=float(g.abs().sum()))))
