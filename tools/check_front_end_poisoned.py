#!/usr/bin/env python3
"""The bench's multi-GPU front-end check (sharding.fused_predict_shard == the direct C-ABI call, bit for bit) at the full
configs[2] shape with every scratch / result block of the front-end starting as 0xFF bytes (_device.POISON): a cell read
before it is written, or a result cell left unwritten, fails here every time instead of once in a while.
    python tools/check_front_end_poisoned.py [--workload fused_dde_ant] [--rows 1000000] [--reps 3]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from benchlib.workloads import WORKLOADS
from codex_africanus_amd import _device, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="fused_dde_ant")
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--sources", type=int, default=1000)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
args = bench.parse(["--workload", a.workload, "--rows", str(a.rows), "--sources", str(a.sources)])
dev = torch.device("cuda:0")
lib = _lib.load()
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
w = WORKLOADS[a.workload](args, 0, dev, lib, _lib, t)
P = lambda x: ctypes.c_void_p(x.data_ptr())
d_vis = torch.empty((args.rows, args.chans, 4), dtype=torch.complex128, device=dev)
d_vis.view(torch.uint8).fill_(0xFF)
w.d_ws.fill_(0xFF)
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
w.predict(d_vis, stream, P)
torch.cuda.synchronize()
print("direct call: NaNs", int(torch.isnan(torch.view_as_real(d_vis)).sum()))
for byte in [None, 0xFF, 0x01, 0x40, 0x7F] * a.reps:
    _device.POISON = byte is not None
    _device.POISON_BYTE = byte or 0
    junk = torch.empty(int(6e9), dtype=torch.uint8, device=dev).fill_(0x3C if byte is None else byte)   # what the allocator will hand out next
    del junk
    print("poison", byte, w.front_end_check(d_vis, 0, 1, dev))
