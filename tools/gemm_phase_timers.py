#!/usr/bin/env python
"""Phase timers of the GEMM-form fused predict (profiling build: make -C codex_africanus_amd/csrc HOOKS=1; AFHIP_LIB is set
here).  Shader-clock cycles per batch, averaged over every 16th workgroup of channel 0:
matrix wave 0 (MFMA loop, barrier wait) and sampling wave 8 (geometry + issue, gather wait, rounds, barrier wait)."""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("AFHIP_LIB", os.path.join(ROOT, "codex_africanus_amd", "lib", "prof", "libafhip.so"))
sys.path.insert(0, ROOT)
import argparse
import numpy as np
import torch
from codex_africanus_amd import _lib
from benchlib.workloads_fused import FusedDde

lib = _lib.load()
lib.af_debug_gemm_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
args = argparse.Namespace(rows=1000000, chans=64, sources=1000, seed=0, workload="fused_dde_ant", pa="random", uvw="antennas",
                          antennas=int(os.environ.get("ANT", "64")))
wl = FusedDde(args, 0, dev, lib, _lib, t)
P = lambda x: ctypes.c_void_p(x.data_ptr())
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
vis = torch.empty((args.rows, 64, 4), dtype=torch.complex128, device=dev)
wl.predict(vis, stream, P); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
lib.af_debug_gemm_prof(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); wl.predict(vis, stream, P); e1.record(); torch.cuda.synchronize()
lib.af_debug_gemm_prof(out, 0)
v = [int(x) for x in out]
nb = max(v[7], 1)
names = ["matrix_mfma_loop", "matrix_barrier_wait", "sampler_geometry_issue", "sampler_gather_wait", "sampler_rounds", "sampler_barrier_wait"]
res = {n: round(v[i] / nb, 1) for i, n in enumerate(names)}
res["workgroups_sampled"], res["batches"], res["call_ms"] = v[6], v[7], e0.elapsed_time(e1)
res["unit"] = "shader-clock ticks per batch (s_memtime)"
print(json.dumps(res))
