#!/usr/bin/env python
"""BASELINE configs[4] through the wgridder-shaped entry (gridding.wgridder.model): 4096^2 model image, 1e6 rows x 64
chan, epsilon 1e-5 (7 taps per axis), w-stacking on; device-resident, HIP-event timing; accuracy of sampled rows against
the direct transform of the image's non-zero pixels (CPU oracle)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd.gridding.wgridder import model
from codex_africanus_amd.gridding.wgridder.im2vis import kernel_parameters
import oracle

dev = torch.device("cuda:0")
npix, nrow, nchan = int(os.environ.get("NPIX", 4096)), int(os.environ.get("NROW", 1000000)), 64
eps = float(os.environ.get("EPS", 1e-5))
cell = np.deg2rad(2.0 / 3600.0)
freq = np.linspace(0.856e9, 1.712e9, nchan)
rng = np.random.default_rng(0)
umax = 0.45 / cell * (299792458.0 / freq.max())
uvw = np.zeros((nrow, 3))
uvw[:, :2] = rng.uniform(-1, 1, (nrow, 2)) * umax
uvw[:, 2] = rng.uniform(-400, 400, nrow)
image = np.zeros((1, npix, npix))
nz = rng.integers(0, npix, (5000, 2))
image[0, nz[:, 0], nz[:, 1]] = rng.lognormal(0, 1, 5000)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
args = (T(uvw), T(freq), T(image), np.array([0]), np.array([nchan]), cell)
for _ in range(2):
    vis = model(*args, epsilon=eps)
torch.cuda.synchronize()
# per-call HIP-event times; the median is reported (on some boxes of the pool one call in a few stalls for ~0.5 s on the
# host side of an allocation, which a mean over a handful of calls would report as the kernel's time)
times = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    vis = model(*args, epsilon=eps)
    e1.record(); torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
ms = float(np.median(times))
rows = np.linspace(0, nrow - 1, 64).astype(int)
ix, iy = np.nonzero(image[0])
x, y = (ix - npix / 2) * cell, (iy - npix / 2) * cell
n = np.sqrt(1 - x * x - y * y)
src = np.broadcast_to((image[0, ix, iy] / n)[:, None, None], (ix.size, nchan, 1)).copy()
ref = oracle.im_to_vis(src, uvw[rows] * np.array([1, 1, -1.0]), np.stack([x, y], 1), freq, omp=True)[:, :, 0]
got = vis[torch.from_numpy(rows).to(dev)].cpu().numpy()
l2 = float(np.sqrt(np.sum(np.abs(got - ref) ** 2) / np.sum(np.abs(ref) ** 2)))
wl = np.abs(uvw[:, 2]).max() * freq.max() / 299792458.0
emax = 2 * (npix / 2 * cell) ** 2
print(json.dumps(dict(ms=ms, ms_calls=[round(t, 2) for t in times], Mvis_per_s=nrow * nchan / ms / 1e3, epsilon=eps, taps=kernel_parameters(eps)[0],
                      l2_error_vs_direct_transform=l2, npix=npix, rows=nrow, chans=nchan,
                      w_planes=int(np.ceil(2 * wl * 4 * emax / (np.sqrt(1 - emax) + 1))) + kernel_parameters(eps)[0] + 1)))
