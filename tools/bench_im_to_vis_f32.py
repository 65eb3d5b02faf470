#!/usr/bin/env python
"""im_to_vis at BASELINE configs[1]'s counts (1e6 rows x 64 chan x 1000 sources x 4 corr) in single precision
(af_im_to_vis_f32) next to the float64 kernel on the same box: the rounded float32 linspace band (corrected
recurrence), an exactly representable float32 grid (plain recurrence), both tile widths, real and complex pixels;
errors against the float64 transform of the float32 inputs on a row sample."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from codex_africanus_amd import dft
from codex_africanus_amd.testing import synthetic_inputs, real_image

dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


nrow, nchan, nsrc = int(os.environ.get("AF_BENCH_ROWS", 1000000)), 64, 1000
d = synthetic_inputs(seed=0, nrow=16, nchan=nchan, nsrc=nsrc, nant=64)
rng = np.random.default_rng(1000)
uvw = np.empty((nrow, 3))
uvw[:, 0] = rng.uniform(-4000, 4000, nrow); uvw[:, 1] = rng.uniform(-4000, 4000, nrow); uvw[:, 2] = rng.uniform(-400, 400, nrow)
img_r = real_image(d)
img_c = np.ascontiguousarray(np.broadcast_to(d["brightness"][:, None, :], (nsrc, nchan, 4)))
lm = d["lm"]
bands = {"linspace cast to float32 (rounded: corrected recurrence)": d["frequency"].astype(np.float32),
         "float32-exact grid (plain recurrence)": (0.856e9 + np.arange(nchan) * 13586432.0).astype(np.float32)}
rows = np.linspace(0, nrow - 1, 48).astype(np.int64)
out = {}
t64 = [torch.from_numpy(a).to(dev) for a in (img_r, uvw, lm, d["frequency"])]
dt = timeit(lambda: dft.im_to_vis(*t64))
out["float64 kernel, real image"] = dict(ms=dt * 1e3, Mvis_s=nrow * nchan / dt / 1e6)
base = dt
t64c = [torch.from_numpy(img_c).to(dev)] + t64[1:]
dtc = timeit(lambda: dft.im_to_vis(*t64c))
out["float64 kernel, complex image"] = dict(ms=dtc * 1e3, Mvis_s=nrow * nchan / dtc / 1e6)
del t64, t64c
ct = os.environ.get("AFHIP_F32_CT", "16")      # the library reads the tile width once per process: one width per run
for label, fr in bands.items():
    for name, img, ref_t in (("real", img_r.astype(np.float32), base), ("complex", img_c.astype(np.complex64), dtc)):
        t = [torch.from_numpy(a).to(dev) for a in (img, uvw.astype(np.float32), lm.astype(np.float32), fr)]
        v = dft.im_to_vis(*t)
        dt = timeit(lambda: dft.im_to_vis(*t))
        truth = oracle.im_to_vis(img.astype(np.complex128 if name == "complex" else np.float64),
                                 uvw.astype(np.float32).astype(np.float64)[rows], lm.astype(np.float32).astype(np.float64),
                                 fr.astype(np.float64), omp=True)
        got = v.cpu().numpy()[rows].astype(np.complex128)
        err = np.abs(got - np.asarray(truth, dtype=np.complex128))
        out["float32 CT=%s, %s, %s image" % (ct, label, name)] = dict(
            ms=dt * 1e3, Mvis_s=nrow * nchan / dt / 1e6, speedup_vs_float64=ref_t / dt,
            max_abs_err_vs_float64_transform=float(err.max()), peak_visibility=float(np.abs(truth).max()),
            max_rel_err_vs_float64_transform=float(err.max() / np.abs(truth).max()))
        del t, v
print(json.dumps(out, indent=1))
