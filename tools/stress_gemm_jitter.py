#!/usr/bin/env python3
"""Timing-perturbation stress of the GEMM-form kernel's panel-buffer protocol, PROFILING build only:
    make -C codex_africanus_amd/csrc HOOKS=1
    AFHIP_LIB=codex_africanus_amd/lib/prof/libafhip.so python tools/stress_gemm_jitter.py [--seeds 6]
For array sizes that run every instantiation kind -- DIAG with one super-round per batch (64 antennas), DIAG with padded
super-rounds (40, 24 antennas), RECT with the flat term stream over three panel buffers (STRADDLE: 128 antennas), RECT with
a short last column block (100, 197 antennas) -- the unperturbed result, then `seeds` runs in which every wave sleeps
pseudo-random 0 .. 8000 cycles around the batch barriers (af_debug_gemm_jitter): every run must reproduce the
unperturbed bits.  A missing barrier, or a buffer re-used one batch too early, shows as a mismatch here."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=6)
    ap.add_argument("--sources", type=int, default=203)       # odd batch count incl. a partial last batch
    ap.add_argument("--steps", type=int, default=12, help="timesteps per case")
    ap.add_argument("--antennas", default="64,128,100,40,24,197")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from benchlib.workloads import WORKLOADS
    from codex_africanus_amd import _lib
    lib = _lib.load()
    if not hasattr(lib, "af_debug_gemm_jitter"):
        raise SystemExit("needs the profiling build: AFHIP_LIB=codex_africanus_amd/lib/prof/libafhip.so")
    lib.af_debug_gemm_jitter.argtypes = [ctypes.c_uint]
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    report, bad = [], 0
    for nant in (int(x) for x in a.antennas.split(",")):
        rows = a.steps * (nant * (nant - 1) // 2) - 7
        args = bench.parse(["--workload", "fused_dde_ant", "--rows", str(rows), "--sources", str(a.sources), "--antennas", str(nant)])
        w = WORKLOADS["fused_dde_ant"](args, 0, dev, lib, _lib, t)
        base = torch.empty((rows, args.chans, 4), dtype=torch.complex128, device=dev)
        other = torch.empty_like(base)
        assert lib.af_debug_gemm_jitter(0) == 0
        w.predict(base, stream, P)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w.predict(other, stream, P)
        torch.cuda.synchronize()
        plain = time.perf_counter() - t0
        assert torch.equal(base, other)
        case = {"antennas": nant, "rows": rows, "sources": a.sources, "plain_ms": 1e3 * plain, "seeds": []}
        for seed in range(1, a.seeds + 1):
            assert lib.af_debug_gemm_jitter((0x9E3779B9 * seed) & 0xFFFFFFFF or 1) == 0
            other.fill_(complex(float("nan"), 0.0))
            t0 = time.perf_counter()
            w.predict(other, stream, P)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            same = bool(torch.equal(base, other))
            cells = 0 if same else int(((torch.view_as_real(base) != torch.view_as_real(other)).any(-1) | torch.isnan(other.real)).sum())
            case["seeds"].append({"seed": seed, "ms": 1e3 * dt, "bit_equal": same, "cells_differ": cells})
            bad += 0 if same else 1
        lib.af_debug_gemm_jitter(0)
        report.append(case)
        print(json.dumps(case), flush=True)
        del w, base, other
    out = {"cases": report, "mismatches": bad}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gemm_jitter.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("mismatching runs: %d" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
