#!/usr/bin/env python
"""vis_to_im on one MI355X at C2's counts (1e6 rows x 64 chan x 4 corr -> 1000 sources), device-resident
inputs, HIP-event timing on torch's stream; error against the oracle on a source sample."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import dft
import oracle

dev = torch.device("cuda:0")
nrow, nchan, nsrc = int(os.environ.get("NROW", 1000000)), 64, int(os.environ.get("NSRC", 1000))
rs = np.random.default_rng(0)
uvw = np.stack([rs.uniform(-4000, 4000, nrow), rs.uniform(-4000, 4000, nrow), rs.uniform(-400, 400, nrow)], axis=1)
lm = rs.uniform(-0.035, 0.035, (nsrc, 2))
freq = np.linspace(0.856e9, 1.712e9, nchan)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
d_vis = torch.randn(nrow, nchan, 4, dtype=torch.complex128, device=dev)
d_flags = torch.rand(nrow, nchan, 4, device=dev) < 0.01
args = (d_vis, T(uvw), T(lm), T(freq), d_flags)
out = {}
for mode in ("auto",):
    dft.set_mode(mode)
    im = dft.vis_to_im(*args); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        dft.vis_to_im(*args)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    idx = np.linspace(0, nsrc - 1, 8).astype(int)
    sub = slice(0, 20000)
    ref = oracle.vis_to_im(d_vis[sub].cpu().numpy(), uvw[sub], lm[idx], freq, d_flags[sub].cpu().numpy())
    got = dft.vis_to_im(d_vis[sub], T(uvw[sub]), T(lm[idx]), T(freq), d_flags[sub]).cpu().numpy()
    out[mode] = dict(ms=ms, G_row_src_chan_per_s=nrow * nsrc * nchan / ms / 1e6,
                     max_abs_err_20k_rows=float(np.abs(got - ref).max()), scale=float(np.abs(ref).max()))
# single precision (af_vis_to_im_f32): complex64 visibilities, float32 coordinates; both classes of float32 band
vis32 = d_vis.to(torch.complex64)
for label, fr in (("float32, linspace cast to float32 (corrected recurrence)", freq.astype(np.float32)),
                  ("float32, float32-exact grid (plain recurrence)", (0.856e9 + np.arange(nchan) * 13586432.0).astype(np.float32))):
    a32 = (vis32, T(uvw.astype(np.float32)), T(lm.astype(np.float32)), T(fr), d_flags)
    dft.vis_to_im(*a32); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        dft.vis_to_im(*a32)
    e1.record(); torch.cuda.synchronize()
    ms32 = e0.elapsed_time(e1) / 3
    idx = np.linspace(0, nsrc - 1, 8).astype(int)
    sub = slice(0, 20000)
    ref = oracle.vis_to_im(vis32[sub].cpu().numpy().astype(np.complex128), uvw.astype(np.float32).astype(np.float64)[sub],
                           lm.astype(np.float32).astype(np.float64)[idx], fr.astype(np.float64), d_flags[sub].cpu().numpy())
    got = dft.vis_to_im(vis32[sub], T(uvw.astype(np.float32)[sub]), T(lm.astype(np.float32)[idx]), T(fr), d_flags[sub]).cpu().numpy()
    out[label] = dict(ms=ms32, G_row_src_chan_per_s=nrow * nsrc * nchan / ms32 / 1e6, speedup_vs_float64=out["auto"]["ms"] / ms32,
                      max_abs_err_20k_rows=float(np.abs(got - ref).max()), scale=float(np.abs(ref).max()))
print(json.dumps(out, indent=1))
