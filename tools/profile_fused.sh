#!/bin/bash
# rocprofv3 counter passes on the fused DDE predict (BASELINE configs[2]); run on the GPU box from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/profile_fused.sh'
# Counters are collected in their own runs (never combined with other trace domains).
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_fused
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="bench.py --workload fused_dde --steps 2 --warmup 1 --no-cpu-baseline --check-rows 0"
python3 bench.py --workload fused_dde --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d "$OUT/sq" -o sq -- python3 $ARGS > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA \
    --kernel-trace --output-format csv -d "$OUT/sq2" -o sq2 -- python3 $ARGS > "$OUT/sq2.log" 2>&1
python3 - <<'PY'
import csv, glob, json
out = {}
for sub in ("sq", "sq2"):
    for cc in glob.glob("gpurun_out/prof_fused/%s/**/*counter_collection.csv" % sub, recursive=True):
        per = {}
        for r in csv.DictReader(open(cc)):
            if "fused_predict_kernel" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in per.items():
            out[k] = sum(v) / len(v)
    for kt in glob.glob("gpurun_out/prof_fused/%s/**/*kernel_trace.csv" % sub, recursive=True):
        d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))
             if "fused_predict_kernel" in r["Kernel_Name"]]
        if d:
            out["avg_ns_under_pmc_%s" % sub] = sum(d) / len(d)
for st in glob.glob("gpurun_out/prof_fused/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(st)):
        if "fused_predict_kernel" in r["Name"]:
            out["avg_ns_kernel_trace_stats"] = float(r["AverageNs"]); out["calls"] = int(r["Calls"])
json.dump(out, open("gpurun_out/prof_fused/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
for f in "$OUT"/*.log; do tail -n 2 "$f"; done
