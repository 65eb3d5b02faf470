#!/bin/bash
# The pytest-session condition of the front-end mismatch: the three 8-rank jobs of tests/test_gpu_bench_ranks.py in
# the file's order -- ranks-dft, threads-dft (one process, eight threads, 64 GB, exits immediately before), ranks-fused_dde_ant
# (eight processes, the bit-for-bit front-end check, four times per rank) -- again and again inside pytest sessions.
#   tools/stress_pytest_ranks.sh [seconds=2400]
# A failure leaves the ranks' forensics in gpurun_out/front_end_mismatch_*.json; the loop stops at the first one.
cd "$(dirname "$0")/.." || exit 1
LIMIT=${1:-2400}; T0=$(date +%s); N=0; mkdir -p gpurun_out/r6
export AFHIP_BENCH_FRONT_END_REPEATS=4
while [ $(( $(date +%s) - T0 )) -lt $LIMIT ]; do
  N=$((N+1))
  timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -k "eight_ranks" -x -q > gpurun_out/r6/pytest_ranks_loop.log 2>&1
  rc=$?
  echo "loop $N rc $rc $(tail -1 gpurun_out/r6/pytest_ranks_loop.log)"
  if [ $rc -ne 0 ]; then cp gpurun_out/r6/pytest_ranks_loop.log gpurun_out/r6/pytest_ranks_FAILED_$N.log; echo "FAILED in loop $N"; break; fi
done
echo "{\"loops\": $N, \"seconds\": $(( $(date +%s) - T0 )), \"failed\": $([ ${rc:-0} -ne 0 ] && echo true || echo false)}" | tee gpurun_out/r6/pytest_ranks_loop_summary.json
