#!/usr/bin/env python3
"""Do results depend on what the LDS held before the kernel started?  (A workgroup's LDS is whatever the previous workgroup
on that CU left -- normally one of the same kernel, i.e. plausible values; after another process or another kernel,
anything.)  Profiling build only (make -C codex_africanus_amd/csrc HOOKS=1; AFHIP_LIB=codex_africanus_amd/lib/prof/libafhip.so):
af_debug_fill_lds(pattern) fills every CU's LDS, then the workload's predict runs; the results for several patterns must
be the same bits.
    AFHIP_LIB=codex_africanus_amd/lib/prof/libafhip.so python tools/check_lds_independence.py [--workloads fused_dde_ant,...]"""
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
from codex_africanus_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--workloads", default="fused_dde_ant,fused_dde,fused_dde_ant128,dft,dft_complex,gauss,degrid,wgrid")
ap.add_argument("--rows", type=int, default=200000)
ap.add_argument("--sources", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
lib.af_debug_fill_lds.argtypes = [ctypes.c_uint, ctypes.c_void_p]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
P = lambda x: ctypes.c_void_p(x.data_ptr())
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
bad = 0
for name in a.workloads.split(","):
    args = bench.parse(["--workload", name, "--rows", str(a.rows), "--sources", str(a.sources)])
    w = WORKLOADS[name](args, 0, dev, lib, _lib, t)
    shape = (args.rows, args.chans, w.ncorr)
    ref = None
    for pattern in (0x00000000, 0xFFFFFFFF, 0x7FF80000, 0x40404040, 0x00000001, 0xC0000000):
        out = torch.empty(shape, dtype=getattr(torch, getattr(w, "vis_dtype", "complex128")), device=dev)
        out.view(torch.uint8).fill_(0xFF)
        assert lib.af_debug_fill_lds(pattern, stream) == 0
        w.predict(out, stream, P)
        torch.cuda.synchronize()
        if ref is None:
            ref = out
            continue
        if not torch.equal(torch.view_as_real(ref).view(torch.uint8), torch.view_as_real(out).view(torch.uint8)):
            d = (torch.view_as_real(ref) != torch.view_as_real(out)).any(-1) if ref.is_complex() else (ref != out)
            idx = d.nonzero()
            print("%s: LDS pattern %08x changes %d cells (rows %d..%d)" % (name, pattern, idx.shape[0], int(idx[:, 0].min()), int(idx[:, 0].max())), flush=True)
            bad += 1
    print("%s: checked" % name, flush=True)
sys.exit(1 if bad else 0)
