#!/usr/bin/env python3
"""Run-to-run determinism of the GEMM-form fused predict under contention: P processes on one device, each launching
af_fused_predict_antennas_c128 K times on the same inputs and comparing every result with its first, bit for bit.
    python tools/stress_gemm_determinism.py [--procs 8] [--reps 12] [--rows 1000000] [--sources 1000]
Prints one line per process; a mismatch prints where (timestep, baseline, channel, correlation) and exits 1."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(a):
    import numpy as np
    import torch
    import ctypes
    import bench
    from benchlib.workloads import WORKLOADS
    from codex_africanus_amd import _lib
    args = bench.parse(["--workload", "fused_dde_ant", "--rows", str(a.rows), "--sources", str(a.sources), "--seed", str(a.seed)])
    dev = torch.device("cuda:0")
    lib = _lib.load()
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    w = WORKLOADS["fused_dde_ant"](args, a.rank, dev, lib, _lib, t)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    first = torch.empty((args.rows, args.chans, 4), dtype=torch.complex128, device=dev)
    other = torch.empty_like(first)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    w.predict(first, stream, P)
    torch.cuda.synchronize()
    bad = 0
    for k in range(a.reps):
        other.fill_(complex(float("nan"), 0.0))
        w.predict(other, stream, P)
        torch.cuda.synchronize()
        if not torch.equal(first, other):
            diff = (torch.view_as_real(first) != torch.view_as_real(other)).any(-1) | torch.isnan(other.real)
            idx = diff.nonzero()
            rows = idx[:, 0].unique()
            nbl = w.nbl
            print("proc %d rep %d: %d cells differ; rows %d..%d (steps %s) baselines-in-step %s chans %s corrs %s maxdiff %.3e" % (
                a.rank, k, idx.shape[0], int(rows.min()), int(rows.max()), sorted(set((rows // nbl).tolist()))[:8],
                sorted(set((rows % nbl).tolist()))[:16], sorted(set(idx[:, 1].tolist()))[:16], sorted(set(idx[:, 2].tolist())),
                float((first - other)[diff].abs().max())), flush=True)
            bad += 1
    print("proc %d: %d of %d repeats differ" % (a.rank, bad, a.reps), flush=True)
    # the multi-GPU front-end on the same arrays (plan from the device arrays, plan guard, pooled scratch): as bench.py checks it
    for k in range(a.front_end):
        try:
            w.front_end_check(first, a.rank, a.procs, dev)
        except SystemExit as e:
            print("proc %d front-end rep %d: %s" % (a.rank, k, e), flush=True)
            bad += 1
    print("proc %d: front-end checks done" % a.rank, flush=True)
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--sources", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--front-end", type=int, default=6, help="repeats of the sharding front-end check per process")
    a = ap.parse_args()
    if a.rank >= 0:
        sys.exit(worker(a))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), "--reps", str(a.reps), "--rows", str(a.rows),
                               "--sources", str(a.sources), "--seed", str(a.seed), "--procs", str(a.procs),
                               "--front-end", str(a.front_end)]) for r in range(a.procs)]
    rc = 0
    for p in procs:
        rc |= p.wait()
    sys.exit(rc)


if __name__ == "__main__":
    main()
