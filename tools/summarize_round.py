#!/usr/bin/env python
"""Condense gpurun_out/prof_round (tools/profile_round.sh) into profiles/<round>_*:
  <round>_<workload>_bench_line.json      the un-profiled JSON line of bench.py --workload <workload>
  <round>_<workload>_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command (3 timed steps)
  <round>_<workload>_pmc_summary.json     per-launch means of the PMC passes for the workload's dominant kernel
  <round>_<workload>_pmc_<pass>.csv       the counter rows of that kernel, per dispatch
  <round>_aux_<tool>.json / _kernel_stats.csv   the auxiliary benches
    python tools/summarize_round.py r02"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_round")
DST = os.path.join(ROOT, "profiles")
DOMINANT = {"dft": "dft_mfma_kernel", "dft_complex": "dft_mfma_kernel", "dft_f32": "dft_f32_kernel", "gauss": "dft_mfma_kernel", "fused_dde": "fused_predict_kernel",
            "fused_dde_ant": "fused_gemm3_kernel", "fused_dde_ant128": "fused_gemm3_kernel", "fused_dde_ant_c64": "fused_gemm_c64_kernel", "fused_dde_c64": "fused_rows_c64_kernel",
            "degrid": "degrid_coop_kernel", "wgrid": "wg_degrid_tiles", "wgrid_f32planes": "wg_degrid_tiles"}


def find(base, suffix):
    hits = sorted(glob.glob(os.path.join(base, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    dl = os.path.join(SRC, "default_line.json")
    if os.path.exists(dl):
        lines = [x for x in open(dl).read().splitlines() if x.startswith("{")]
        if lines:
            open(os.path.join(DST, "%s_bench_default_line.json" % tag), "w").write(lines[-1] + "\n")
    for w, dom in DOMINANT.items():
        base = os.path.join(SRC, w)
        if not os.path.isdir(base):
            continue
        summary = {"workload": w, "kernel_substring": dom}
        bl = os.path.join(base, "bench_line.json")
        if os.path.exists(bl):
            lines = [x for x in open(bl).read().splitlines() if x.startswith("{")]
            if lines:
                open(os.path.join(DST, "%s_%s_bench_line.json" % (tag, w)), "w").write(lines[-1] + "\n")
                d = json.loads(lines[-1])
                summary["bench_kernel_ms_hip_events"] = d["roofline"]["kernel_ms"]
                summary["bench_value_Mvis_s"] = d["value"]
                ex = d["roofline"].get("executed", {})
                if "mfma_instructions" in ex:
                    summary["bench_mfma_instructions"] = ex["mfma_instructions"]
        stats = find(os.path.join(base, "stats"), "kernel_stats.csv")
        if stats:
            shutil.copy(stats, os.path.join(DST, "%s_%s_kernel_stats.csv" % (tag, w)))
            for row in csv.DictReader(open(stats)):
                if dom in row["Name"]:
                    summary["kernel"] = row["Name"]
                    summary["calls_kernel_trace_stats"] = int(row["Calls"])
                    summary["avg_ns_kernel_trace_stats"] = float(row["AverageNs"])
                    break
        for sub in ("fetch", "write", "sq", "sq2"):
            cc = find(os.path.join(base, sub), "counter_collection.csv")
            if not cc:
                continue
            # the dominant INSTANTIATION (the stats row with the largest total): some entries launch several
            # instantiations of which all but one return at once (af_im_to_vis_f32: band class x phasor form)
            full = summary.get("kernel")
            allrows = [r for r in csv.DictReader(open(cc)) if dom in r["Kernel_Name"]]
            rows = [r for r in allrows if r["Kernel_Name"] == full] or allrows
            if not rows:
                continue
            names = set(r["Kernel_Name"] for r in rows)
            with open(os.path.join(DST, "%s_%s_pmc_%s.csv" % (tag, w, sub)), "w", newline="") as f:
                wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                wr.writeheader()
                wr.writerows(rows)
            per = {}
            for r in rows:
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, v in per.items():
                summary[k] = sum(v) / len(v)
            # an entry that launches several instantiations per step (the GEMM form beyond 64 antennas: DIAG and RECT
            # super-tiles): the matrix instructions of ALL of them, one launch of each per step
            byname = {}
            for r in allrows:
                if r["Counter_Name"] == "SQ_INSTS_MFMA":
                    byname.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            if len(byname) > 1:
                summary["SQ_INSTS_MFMA_all_instantiations"] = sum(sum(v) / len(v) for v in byname.values())
            kt = find(os.path.join(base, sub), "kernel_trace.csv")
            if kt:
                dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))
                       if r["Kernel_Name"] in names]
                if dur:
                    summary["avg_ns_under_pmc_%s" % sub] = sum(dur) / len(dur)
        if "SQ_LDS_BANK_CONFLICT" in summary and summary.get("SQ_LDS_IDX_ACTIVE"):
            summary["lds_conflict_ratio"] = summary["SQ_LDS_BANK_CONFLICT"] / summary["SQ_LDS_IDX_ACTIVE"]
        if "bench_mfma_instructions" in summary and "SQ_INSTS_MFMA" in summary:
            # the line's "executed" block must describe the kernel that ran (it counted the removed 4M
            # kernel's schedule for a round): matrix instructions claimed == matrix instructions counted
            ratio = summary["bench_mfma_instructions"] / summary.get("SQ_INSTS_MFMA_all_instantiations", summary["SQ_INSTS_MFMA"])
            summary["mfma_claimed_over_counted"] = ratio
            assert abs(ratio - 1.0) < 0.01, "%s: bench line claims %g MFMA per launch, SQ_INSTS_MFMA counted %g" % (
                w, summary["bench_mfma_instructions"], summary["SQ_INSTS_MFMA"])
        if "GRBM_GUI_ACTIVE" in summary and "avg_ns_under_pmc_sq" in summary:
            summary["clock_GHz"] = summary["GRBM_GUI_ACTIVE"] / 8.0 / summary["avg_ns_under_pmc_sq"]
        json.dump(summary, open(os.path.join(DST, "%s_%s_pmc_summary.json" % (tag, w)), "w"), indent=1)
        print(json.dumps(summary, indent=1))
    for d in sorted(glob.glob(os.path.join(SRC, "aux", "*"))):
        t = os.path.basename(d)
        res = os.path.join(d, "result.json")
        if os.path.exists(res) and os.path.getsize(res):
            shutil.copy(res, os.path.join(DST, "%s_aux_%s.json" % (tag, t)))
        # HBM traffic per dispatch of the library's kernels: FETCH_SIZE / WRITE_SIZE (KB) from their own passes, the
        # durations of the same dispatches from those passes' kernel traces; per kernel name: the LARGEST dispatch (the
        # bench-sized one, not the small verification calls)
        per_kernel = {}
        for sub, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            cc = find(os.path.join(d, sub), "counter_collection.csv")
            kt = find(os.path.join(d, sub), "kernel_trace.csv")
            if not cc or not kt:
                continue
            dur = {r["Dispatch_Id"]: float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
            for r in csv.DictReader(open(cc)):
                if "anonymous namespace" not in r["Kernel_Name"] or r["Counter_Name"] != key:
                    continue
                name = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]
                e = per_kernel.setdefault(name, {})
                ns = dur.get(r["Dispatch_Id"], 0.0)
                if ns >= e.get(key + "_ns", -1.0):
                    e[key + "_ns"] = ns
                    e[key + "_KB"] = float(r["Counter_Value"])
        if per_kernel:
            for name, e in per_kernel.items():
                if "FETCH_SIZE_KB" in e and "WRITE_SIZE_KB" in e:
                    ns = max(e["FETCH_SIZE_ns"], e["WRITE_SIZE_ns"])
                    e["hbm_GB_2fetch_plus_write"] = (2.0 * e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"]) * 1024.0 / 1e9
                    e["hbm_GBs"] = e["hbm_GB_2fetch_plus_write"] / (ns * 1e-9) if ns else None
                    e["ms"] = ns / 1e6
            json.dump(per_kernel, open(os.path.join(DST, "%s_aux_%s_pmc_hbm.json" % (tag, t)), "w"), indent=1, sort_keys=True)
        stats = find(os.path.join(d, "stats"), "kernel_stats.csv")
        if stats:
            # keep the library's own kernels (and the FFT) only
            rows = [r for r in csv.DictReader(open(stats)) if "anonymous namespace" in r["Name"] or "fft" in r["Name"]
                    or "transpose" in r["Name"]]
            if rows:
                with open(os.path.join(DST, "%s_aux_%s_kernel_stats.csv" % (tag, t)), "w", newline="") as f:
                    wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                    wr.writeheader()
                    wr.writerows(rows)


if __name__ == "__main__":
    main()
