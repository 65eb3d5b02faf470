#!/usr/bin/env python
"""The adjoint of tools/bench_wgridder.py: gridding.wgridder.dirty at BASELINE configs[4]'s counts (1e6 rows x 64 chan ->
4096^2 image, epsilon 1e-5, w-stacking on); device-resident, per-call HIP-event times (median); accuracy of sampled
pixels against the direct transform (CPU oracle vis_to_im on the pixel sample)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd.gridding.wgridder import dirty
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
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
d_vis = torch.randn(nrow, nchan, dtype=torch.complex128, device=dev)
args = (T(uvw), T(freq), d_vis, np.array([0]), np.array([nchan]), npix, npix, cell)
for _ in range(2):
    img = dirty(*args, epsilon=eps)
torch.cuda.synchronize()
times = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    img = dirty(*args, epsilon=eps)
    e1.record(); torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
ms = float(np.median(times))
# 24 pixels against the direct sum over a row sample would not be the same sum: use ALL rows on few pixels
pix = rng.integers(0, npix, (6, 2))
x, y = (pix[:, 0] - npix / 2) * cell, (pix[:, 1] - npix / 2) * cell
n = np.sqrt(1 - x * x - y * y)
vis = d_vis.cpu().numpy()
im = oracle.vis_to_im(vis[:, :, None], uvw * np.array([1, 1, -1.0]), np.stack([x, y], 1), freq,
                      np.zeros(vis.shape + (1,), np.uint8), omp=True)[:, :, 0].sum(axis=1) / n
got = img[0].cpu().numpy()[pix[:, 0], pix[:, 1]]
scale = float(np.sqrt(np.mean(img[0].cpu().numpy() ** 2)))
print(json.dumps(dict(ms=ms, ms_calls=[round(t, 2) for t in times], Mvis_per_s=nrow * nchan / ms / 1e3, epsilon=eps,
                      taps=kernel_parameters(eps)[0], max_pixel_error_over_image_rms=float(np.abs(got - im).max() / scale),
                      npix=npix, rows=nrow, chans=nchan)))
