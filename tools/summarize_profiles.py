#!/usr/bin/env python
"""Condense gpurun_out/prof (tools/profile_bench.sh) into profiles/<round>_*: the kernel-stats table, the
per-dispatch counter rows of the dominant kernel, and <round>_pmc_summary.json (per-launch means).
    python tools/summarize_profiles.py r01 [dominant-kernel-substring]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def find(sub, suffix):
    hits = sorted(glob.glob(os.path.join(SRC, sub, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    dom = sys.argv[2] if len(sys.argv) > 2 else "dft_mfma_kernel"
    summary = {"kernel_substring": dom}
    stats = find("stats", "kernel_stats.csv")
    if stats:
        shutil.copy(stats, os.path.join(DST, "%s_bench_kernel_stats.csv" % tag))
        for row in csv.DictReader(open(stats)):
            if dom in row["Name"]:
                summary["kernel"] = row["Name"]
                summary["calls_kernel_trace_stats"] = int(row["Calls"])
                summary["avg_ns_kernel_trace_stats"] = float(row["AverageNs"])
    for sub in ("fetch", "write", "sq", "sq2"):
        cc = find(sub, "counter_collection.csv")
        if not cc:
            continue
        rows = [r for r in csv.DictReader(open(cc)) if dom in r["Kernel_Name"]]
        if not rows:
            continue
        out = os.path.join(DST, "%s_pmc_%s_dft_kernel.csv" % (tag, sub))
        with open(out, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        per = {}
        for r in rows:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in per.items():
            summary[k] = sum(v) / len(v)
        kt = find(sub, "kernel_trace.csv")
        if kt:
            d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))
                 if dom in r["Kernel_Name"]]
            if d:
                summary["avg_ns_under_pmc_%s" % sub] = sum(d) / len(d)
    bl = os.path.join(SRC, "bench_line.json")
    if os.path.exists(bl):
        lines = [x for x in open(bl).read().splitlines() if x.startswith("{")]
        if lines:
            open(os.path.join(DST, "%s_bench_line.json" % tag), "w").write(lines[-1] + "\n")
    json.dump(summary, open(os.path.join(DST, "%s_pmc_summary.json" % tag), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
