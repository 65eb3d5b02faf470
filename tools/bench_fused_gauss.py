#!/usr/bin/env python
"""Fused DDE predict with Gaussian source shapes (half of 1000 sources extended) at a fifth of C3's rows (201600 rows x
64 chan, 64 antennas): the kernel's HIP-event time.  AFHIP_FUSED_WS=0 selects the 8-wave kernel for comparison."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time, torch, os, argparse
from codex_africanus_amd import rime, _lib
from bench import FusedDde
args = argparse.Namespace(gpus=1, steps=1, warmup=0, rows=201600, chans=64, sources=1000, seed=0, mode="auto", workload="fused_dde", pa="random", npix=4096, backend="nccl", no_cpu_baseline=True, cpu_seconds=1.0, check_rows=0)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
wl = FusedDde(args, 0, dev, _lib.load(), _lib, t)
h = wl.h; dv = wl.dv
gs = np.zeros((1000, 3)); gs[::2] = [2e-4, 1e-4, 0.3]
ti = t(h["time_index"]); g = t(gs)
f = lambda: rime.fused_predict_vis(ti, dv["a1"], dv["a2"], dv["lm"], dv["uvw"], dv["freq"], dv["X"], dv["beam"], dv["ext"], dv["fmap"], dv["pa"], dv["pe"], dv["asc"], gauss_shape=g)
out = f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a = ctypes = None
import ctypes
ea, eb = ctypes.c_void_p(), ctypes.c_void_p()
_lib.call("af_event_create", ctypes.byref(ea)); _lib.call("af_event_create", ctypes.byref(eb))
_lib.call("af_profile_events", ea, eb)
out = f(); torch.cuda.synchronize()
ms = ctypes.c_float(0); _lib.call("af_event_elapsed_ms", ea, eb, ctypes.byref(ms))
print("gauss half of the sources, 201600 rows, AFHIP_FUSED_WS=%s: kernel %.2f ms, checksum %.6e" % (os.environ.get("AFHIP_FUSED_WS", "1"), ms.value, float(out.abs().sum())))
