#!/bin/bash
# rocprofv3 counter passes on the GEMM-form fused predict (fused_dde_ant): whole kernel or one stage (AFHIP_FUSED_STAGE).
#   gpurun --timeout 1500 -- 'bash tools/profile_gemm_stage.sh [stage]'
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
STAGE=${1:-0}
export AFHIP_FUSED_STAGE=$STAGE
OUT=gpurun_out/prof_gemm_s$STAGE
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="bench.py --workload fused_dde_ant --steps 2 --warmup 1 --no-cpu-baseline --check-rows 0 --extras none"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d "$OUT/sq" -o sq -- python3 $ARGS > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES \
    --kernel-trace --output-format csv -d "$OUT/sq2" -o sq2 -- python3 $ARGS > "$OUT/sq2.log" 2>&1
rocprofv3 --pmc TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum \
    --kernel-trace --output-format csv -d "$OUT/tcp" -o tcp -- python3 $ARGS > "$OUT/tcp.log" 2>&1
rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum \
    --kernel-trace --output-format csv -d "$OUT/tcp2" -o tcp2 -- python3 $ARGS > "$OUT/tcp2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
base = sys.argv[1]
out = {}
for sub in ("sq", "sq2", "tcp", "tcp2"):
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
