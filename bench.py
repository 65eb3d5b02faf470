#!/usr/bin/env python
"""
bench.py -- benchmarks of the RIME visibility-predict hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM:
    vis = predict(...)                               # the workload's transform (below)
    chi2[nu] = sum |data - vis|^2                    # per-channel chi^2 of the shard
    (N > 1) all-reduce of chi2                        # the only cross-GPU exchange of the path
PER GPU (rows shard across GPUs, weak scaling: BASELINE configs[3] is 8e6 rows on 8 GPUs).
Metric: Mvis/s = rows x chans / second / 1e6 (whole job).

Workloads (--workload), one per BASELINE config a single GPU can run:
    dft          im_to_vis, real 4-correlation image: BASELINE configs[1], the HEADLINE and the default
                 (1e6 rows x 64 chan x 1000 point sources x 4 corr, fp64)
    dft_complex  the same transform with complex brightness matrices (= the fused predict without DDEs)
    dft_f32      the same transform for single-precision callers (float32 in, complex64 out; af_im_to_vis_f32)
    fused_dde    fused predict with per-antenna beam-cube DDEs, 64 antennas: BASELINE configs[2]
    fused_dde_ant  the same with antenna-decomposable uvw (uvw_pq = uvw_p - uvw_q, as in a Measurement Set): one complex
                 GEMM per (timestep, channel) on the fp64 matrix cores (= fused_dde --uvw antennas)
    fused_dde_ant128  the same on a 128-antenna array: the GEMM form in super-tiles (65 .. 256 antennas)
    fused_dde_ant_c64  fused_dde_ant with every input single precision (complex64 out): the GEMM form on the fp32 matrix cores
    fused_dde_c64      fused_dde (uvw drawn per row) with every input single precision: the lane-per-row kernel in packed float32
    degrid       convolutional degridding of a 4096^2 grid, 1e6 rows x 64 chan, 7x7 taps: BASELINE configs[4]
    wgrid        wgridder-style degridding of a 4096^2 image at epsilon 1e-5: BASELINE configs[4] as named
The default run (N = 1, headline shape) also times every other workload for a few steps and reports them under
"workloads" in the same JSON line (--extras none switches that off; --extras a,b picks some).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload W] [--rows R --chans C --sources S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: rows shard, nothing else moves.  Two executors:
  --executor ranks    (default) one process per GPU, torch.distributed, chi^2 all-reduced by RCCL over xGMI.  Under
                      a launcher (WORLD_SIZE set) this process is one rank.  WITHOUT a launcher `--gpus N` starts
                      its N rank processes itself -- before this process makes any GPU call -- relays rank 0's
                      JSON line and exits non-zero if any rank failed: it never falls through to one rank.
  --executor threads  ONE process, N worker threads, row block k on device k % N through
                      codex_africanus_amd.placement.block(k) -- the reference's "dask chunks on a thread pool"
                      (africanus/rime/dask_predict.py:311-369, africanus/dft/dask.py:37-51) mapped to the GPUs of
                      one node; the chi^2 partials are peer-copied to the first device and summed there.
Fewer visible devices than N is an error unless AFHIP_BENCH_DEVICE=d puts every rank / worker on device d
(then the ranks use gloo: RCCL refuses two ranks on one device).  "n_gpus" is the number of ranks / workers that
actually reported.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  "roofline"     for the workload's dominant kernel: algorithmic flops (or bytes) per launch / its average
                 duration measured with HIP events on the stream it is launched on (the library's measurement
                 hook af_profile_events brackets exactly that kernel), against the peak that bounds it
                 ("Rooflines"); "traffic" = HBM bytes per launch from the PMC passes of the same
                 command committed under profiles/ (null when no such summary exists);
  "cpu_baseline" the CPU oracle (C restatement of the reference's numba loops) timed on this box's host
                 cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib.common import DEFAULT_SHAPE, EXTRA_WORKLOADS, sized_cpu_sample  # noqa: E402,F401
from benchlib.workloads import METRIC, WORKLOADS  # noqa: E402,F401
from benchlib.measure import extras_requested  # noqa: E402,F401
from benchlib.launcher import launch_ranks, rank_environments, supervise  # noqa: E402,F401


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--rows", type=int, default=DEFAULT_SHAPE["rows"], help="rows PER GPU")
    p.add_argument("--chans", type=int, default=DEFAULT_SHAPE["chans"])
    p.add_argument("--sources", type=int, default=DEFAULT_SHAPE["sources"])
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--mode", default="auto", choices=["auto", "exact", "recurrence"])
    p.add_argument("--workload", default="dft", choices=["dft", "dft_complex", "dft_f32", "gauss", "fused_dde", "fused_dde_ant", "fused_dde_ant128", "fused_dde_ant_c64", "fused_dde_c64", "degrid", "wgrid",
                                                         "wgrid_f32planes"])
    p.add_argument("--uvw", default="random", choices=["random", "antennas"],
                   help="fused_dde: uvw drawn per row (BASELINE's recipe: not antenna-decomposable, lane-per-row kernel) or "
                        "differences of per-(time, antenna) coordinates as in a Measurement Set (the GEMM form on the "
                        "matrix cores); workload fused_dde_ant = fused_dde --uvw antennas")
    p.add_argument("--extras", default="auto",
                   help="other workloads timed for a few steps into \"workloads\" of the same JSON line: auto (all of "
                        "them when N = 1, the workload is the headline and the shape is the default), all, none, or a "
                        "comma list of " + ",".join(EXTRA_WORKLOADS))
    p.add_argument("--extra-steps", type=int, default=5)
    p.add_argument("--executor", default="ranks", choices=["ranks", "threads"],
                   help="N > 1: one process per GPU (torch.distributed) or one process with N worker threads")
    p.add_argument("--pa", default="random", choices=["random", "common"],
                   help="fused_dde: parallactic angles iid U(0, pi/6) per (time, antenna) (SURVEY 8(d), the "
                        "reference's own test recipe) or one angle per timestep + 1e-3 rad antenna jitter "
                        "(a real array: coherent beam gathers)")
    p.add_argument("--antennas", type=int, default=64,
                   help="fused_dde / fused_dde_ant: antennas of the array (64 = BASELINE configs[2]; the GEMM form runs up to "
                        "256 in super-tiles, the lane-per-row kernel up to 664)")
    p.add_argument("--npix", type=int, default=DEFAULT_SHAPE["npix"], help="degrid / wgrid: grid size")
    p.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"],
                   help="torch.distributed backend; nccl = RCCL over xGMI.  auto = nccl, or gloo when "
                        "AFHIP_BENCH_DEVICE puts the ranks on one device (the N > 1 code path on a one-GPU box)")
    p.add_argument("--force-dist", action="store_true",
                   help="N = 1: still create the process group (world size 1) and all-reduce the chi^2 vector every step -- "
                        "on a one-GPU box this loads librccl, builds a communicator and runs the collective on the device")
    p.add_argument("--launch-timeout", type=int, default=3600, help="self-launched ranks: seconds before they are killed")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-end-to-end", action="store_true",
                   help="skip the numpy-in -> numpy-out measurement (profiling runs: its chunked calls launch the same "
                        "kernels at other sizes and would mix into per-kernel averages)")
    p.add_argument("--cpu-seconds", type=float, default=2.0,
                   help="minimum wall time of the all-cores CPU baseline sample (the single-thread probe runs "
                        ">= a quarter of it)")
    p.add_argument("--check-rows", type=int, default=256, help="rows checked against the oracle")
    return p.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.executor == "threads":
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--executor threads is one process; do not start it under a multi-rank launcher")
        from benchlib.executors import run_threads
        return run_threads(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))        # this process makes no GPU call, before or after
    from benchlib.executors import run_ranks
    return run_ranks(args)


if __name__ == "__main__":
    main()

