#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: run on the GPU box from the repo root,
#   gpurun --timeout 1500 -- 'bash tools/profile_bench.sh'
# then `python tools/summarize_profiles.py rNN` here.  Kernel trace and every counter group are
# separate runs (counters are never combined with other trace domains).
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --check-rows 0"
python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python3 $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o write -- python3 $ARGS > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d "$OUT/sq" -o sq -- python3 $ARGS > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS \
    --kernel-trace --output-format csv -d "$OUT/sq2" -o sq2 -- python3 $ARGS > "$OUT/sq2.log" 2>&1
find "$OUT" -name "*.csv" | head -40
tail -2 "$OUT"/*.log | tail -30
