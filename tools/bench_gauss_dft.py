#!/usr/bin/env python
"""Gaussian sources without a beam at BASELINE configs[1]'s counts (1e6 rows x 64 chan x 1000 sources, half of them
extended): af_gauss_predict_c128 (csrc/af_gauss_dft.hip) against the route such calls took until round 4 (the fused
beam kernel with a cube of identity matrices).  Prints one JSON line."""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codex_africanus_amd import _lib, rime
from codex_africanus_amd.testing import synthetic_inputs

nrow, nchan, nsrc = int(os.environ.get("ROWS", 1000000)), 64, 1000
dev = torch.device("cuda:0")
d = synthetic_inputs(seed=0, nrow=nrow, nchan=nchan, nsrc=nsrc, nant=64)
rng = np.random.default_rng(1)
sp = np.stack([rng.uniform(0, 3e-4, nsrc), rng.uniform(0, 2e-4, nsrc), rng.uniform(0, np.pi, nsrc)], axis=1)
sp[::2] = 0.0
X = np.ascontiguousarray(np.broadcast_to(d["brightness"][:, None, :], (nsrc, nchan, 4))).reshape(nsrc, nchan, 2, 2)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
lm, uvw, fr, Xd, gs = t(d["lm"]), t(d["uvw"]), t(d["frequency"]), t(X), t(sp)
out = torch.empty((nrow, nchan, 2, 2), dtype=torch.complex128, device=dev)
lib = _lib.load()
nb = int(lib.af_gauss_predict_workspace_bytes(nsrc, nchan))
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
P = lambda x: ctypes.c_void_p(x.data_ptr())
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timed(lambda: _lib.call("af_gauss_predict_c128", P(lm), P(uvw), P(fr), P(Xd), P(gs), nsrc, nrow, nchan, -1, P(out), P(ws),
                             nb, stream), 3)
res = {"gauss_direct_ms": ms, "Mvis_per_s": nrow * nchan / ms / 1e3, "units": float(nrow) * nchan * nsrc}
if os.environ.get("BEAM_ROUTE", "1") != "0":
    ident = np.zeros((2, 2, 2, 2, 2), dtype=np.complex128)
    ident[..., 0, 0] = ident[..., 1, 1] = 1.0
    ntime = d["ntime"]
    args = (t(d["time_index"]), t(d["ant1"]), t(d["ant2"]), lm, uvw, fr, Xd, t(ident), t(np.array([[-2.0, 2.0], [-2.0, 2.0]])),
            t(np.array([0.4e9, 3.5e9])), t(np.zeros((ntime, 64))), t(np.zeros((ntime, 64, 64, 2))), t(np.ones((64, 64, 2))))
    plan = rime.fused_plan(d["time_index"], d["ant1"], d["ant2"], 64)
    other = None
    def beam():
        global other
        other = rime.fused_predict_vis(*args, gauss_shape=gs, plan=plan)
    res["identity_beam_route_ms"] = timed(beam, 2)
    res["max_abs_diff"] = float((other - out).abs().max())
print(json.dumps(res))
