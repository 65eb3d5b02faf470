#!/bin/bash
# (TCP / TA counter passes are NOT part of this script: one hung a box for 25 minutes in round 4.)
# rocprofv3 counter passes on the GEMM-form fused predict (fused_dde_ant): whole kernel or one stage (AFHIP_FUSED_STAGE).
#   gpurun --timeout 1500 -- 'bash tools/profile_gemm_stage.sh [stage]'
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
STAGE=${1:-0}
export AFHIP_FUSED_STAGE=$STAGE
# the stage hook exists only in the profiling build: make -C codex_africanus_amd/csrc HOOKS=1 (before gpurun: it travels)
if [ "$STAGE" != 0 ]; then
  export AFHIP_LIB=codex_africanus_amd/lib/prof/libafhip.so
  [ -f "$AFHIP_LIB" ] || { echo "build the profiling library first: make -C codex_africanus_amd/csrc HOOKS=1"; exit 1; }
fi
OUT=gpurun_out/prof_gemm_s$STAGE
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="bench.py --workload fused_dde_ant --steps 2 --warmup 1 --no-cpu-baseline --check-rows 0 --extras none"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d "$OUT/sq" -o sq -- python3 $ARGS > "$OUT/sq.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES \
    --kernel-trace --output-format csv -d "$OUT/sq2" -o sq2 -- python3 $ARGS > "$OUT/sq2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
base = sys.argv[1]
out = {}
for sub in ("sq", "sq2"):
    for cc in glob.glob("%s/%s/**/*counter_collection.csv" % (base, sub), recursive=True):
        per = {}
        for r in csv.DictReader(open(cc)):
            if "fused_gemm" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in per.items():
            out[k] = sum(v) / len(v)
    for kt in glob.glob("%s/%s/**/*kernel_trace.csv" % (base, sub), recursive=True):
        d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "fused_gemm" in r["Kernel_Name"]]
        if d:
            out["avg_ns_under_pmc_%s" % sub] = sum(d) / len(d)
json.dump(out, open(base + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
for f in "$OUT"/*.log; do tail -n 1 "$f"; done
