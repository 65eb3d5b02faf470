#!/usr/bin/env python3
"""The condition under which the GEMM-form front-end once differed from the direct call: bench.py's
8-rank job with all ranks aliased onto device 0, started again and again with captured pipes as pytest does.
    python tools/stress_bench_ranks.py [--runs 60] [--repeats 4] [--seconds 1500] [--gpus 8] [--rows 1000000]
Every failing run's full text goes to gpurun_out/stress_ranks_failure_<k>.txt (the ranks write their own forensics to
gpurun_out/front_end_mismatch_*.json); one summary line per run, a JSON summary at the end; exit 1 on any failure."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=60)
    ap.add_argument("--repeats", type=int, default=4, help="front-end checks per rank and run (AFHIP_BENCH_FRONT_END_REPEATS)")
    ap.add_argument("--seconds", type=float, default=1500.0, help="stop starting new runs after this long")
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--sources", type=int, default=1000)
    ap.add_argument("--workload", default="fused_dde_ant")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AFHIP_DEVICES", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", AFHIP_BENCH_DEVICE="0", AFHIP_BENCH_FRONT_END_REPEATS=str(a.repeats))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(a.gpus), "--executor", "ranks", "--workload", a.workload,
           "--steps", "2", "--warmup", "1", "--rows", str(a.rows), "--sources", str(a.sources), "--no-cpu-baseline", "--check-rows", "32",
           "--launch-timeout", "850"]
    t00 = time.time()
    failures, done = [], 0
    for k in range(a.runs):
        if time.time() - t00 > a.seconds:
            break
        t0 = time.time()
        p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        done += 1
        ok = p.returncode == 0
        print("run %d: rc %d in %.1f s" % (k, p.returncode, time.time() - t0), flush=True)
        if not ok:
            failures.append(k)
            with open(os.path.join(a.out, "stress_ranks_failure_%d.txt" % k), "wb") as f:
                f.write(p.stdout + b"\n---- stderr ----\n" + p.stderr)
    summary = {"runs": done, "checks": done * a.gpus * a.repeats, "failures": failures, "seconds": time.time() - t00,
               "command": " ".join(cmd[1:]), "repeats": a.repeats}
    print(json.dumps(summary), flush=True)
    with open(os.path.join(a.out, "stress_ranks_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
