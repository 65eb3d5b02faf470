#!/usr/bin/env python
"""Print the per-round reading table of profiles/README.md from the committed summary files
(profiles/<round>_<workload>_pmc_summary.json + _bench_line.json):   python tools/profiles_readme_table.py r04
VERDICT r3 found the hand-written round-3 table stale against the files; this prints it from them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
print("| workload | kernel | bench HIP events | kernel-trace avg | HBM traffic (2 FETCH + WRITE) vs algorithmic | clock | LDS conflict ratio |")
print("|---|---|---|---|---|---|---|")
for w in ("dft", "dft_complex", "dft_f32", "gauss", "fused_dde", "fused_dde_ant", "degrid", "wgrid", "wgrid_f32planes"):
    ps = os.path.join(ROOT, "profiles", "%s_%s_pmc_summary.json" % (tag, w))
    pb = os.path.join(ROOT, "profiles", "%s_%s_bench_line.json" % (tag, w))
    if not (os.path.exists(ps) and os.path.exists(pb)):
        continue
    s, b = json.load(open(ps)), json.loads(open(pb).read())
    r = b["roofline"]
    traffic = (2 * s.get("FETCH_SIZE", 0) + s.get("WRITE_SIZE", 0)) * 1024 / 1e9
    print("| `%s` | `%s` | %.2f ms | %.2f ms (%d calls) | %.1f GB vs %.2f GB | %.2f GHz | %.2f |" % (
        w, r["kernel"], r["kernel_ms"], s["avg_ns_kernel_trace_stats"] / 1e6, s["calls_kernel_trace_stats"], traffic,
        r["hbm"]["algorithmic_bytes"] / 1e9, s.get("clock_GHz", float("nan")), s.get("lds_conflict_ratio", float("nan"))))
