#!/bin/bash
# Is the lane-per-row fused kernel (BASELINE configs[2], per-row uvw) bound by its beam-gather traffic?  The same kernel
# with incoherent (--pa random: BASELINE's recipe) and coherent (--pa common: one angle per timestep) gathers: kernel time
# (HIP events, un-profiled), then FETCH_SIZE and the LDS conflict counters of each.
#   gpurun --timeout 1500 -- 'bash tools/profile_fused_traffic.sh'
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_fused_traffic
rm -rf "$OUT"; mkdir -p "$OUT"
for pa in random common; do
  ARGS="bench.py --workload fused_dde --pa $pa --steps 3 --warmup 1 --no-cpu-baseline --check-rows 0 --extras none"
  python3 bench.py --workload fused_dde --pa $pa --steps 5 --warmup 2 --no-cpu-baseline --extras none > "$OUT/line_$pa.json" 2> "$OUT/line_$pa.err"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch_$pa" -o fetch -- python3 $ARGS > "$OUT/fetch_$pa.log" 2>&1
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq_$pa" -o sq -- python3 $ARGS > "$OUT/sq_$pa.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
base = sys.argv[1]
res = {}
for pa in ("random", "common"):
    e = {}
    for ln in open("%s/line_%s.json" % (base, pa)):
        if ln.startswith("{"):
            d = json.loads(ln)
            e["kernel_ms"] = d["roofline"]["kernel_ms"]; e["frac"] = d["roofline"]["frac"]; e["err"] = d["fp64_max_abs_err"]
    for sub in ("fetch", "sq"):
        for cc in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (base, sub, pa), recursive=True):
            per = {}
            for r in csv.DictReader(open(cc)):
                if "fused_predict_kernel" in r["Kernel_Name"]:
                    per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, v in per.items():
                e[k] = sum(v) / len(v)
    if "FETCH_SIZE" in e:
        e["fetch_GB_raw"] = e["FETCH_SIZE"] * 1024 / 1e9
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_ratio"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
    res[pa] = e
json.dump(res, open(base + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
