#!/usr/bin/env python
"""im_to_vis at C2's counts (1e6 rows x 64 chan x 1000 src x 4 corr) for real and complex images, MFMA
('auto') against VALU-only ('valu') kernels; device-resident, HIP-event timing on torch's stream."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import dft
from codex_africanus_amd.testing import synthetic_inputs, real_image

dev = torch.device("cuda:0")
nrow = int(os.environ.get("NROW", 1000000))
d = synthetic_inputs(seed=0, nrow=16, nchan=64, nsrc=1000, nant=64)
rng = np.random.default_rng(1)
uvw = np.stack([rng.uniform(-4000, 4000, nrow), rng.uniform(-4000, 4000, nrow), rng.uniform(-400, 400, nrow)], axis=1)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
img = real_image(d)
cimg = np.broadcast_to(d["brightness"][:, None, :], (1000, 64, 4)).copy()
out = {}
for name, image in (("real", img), ("complex", cimg)):
    a = (T(image), T(uvw), T(d["lm"]), T(d["frequency"]))
    res = {}
    for mode in ("auto", "valu"):
        dft.set_mode(mode)
        res[mode + "_vis"] = dft.im_to_vis(*a); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            dft.im_to_vis(*a)
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 3
    out[name] = dict(mfma_ms=res["auto"], valu_ms=res["valu"],
                     max_abs_diff=float((res["auto_vis"] - res["valu_vis"]).abs().max()))
    del res
print(json.dumps(out, indent=1))
