line() { python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'))
"; }
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_fused_gemm_c64.py -x -q > gpurun_out/r6/tests_f.log 2>&1; echo "tests rc $?"; tail -6 gpurun_out/r6/tests_f.log
for round in 1 2; do for l in codex_africanus_amd/lib/ab/libafhip_prev.so codex_africanus_amd/lib/libafhip.so; do
  echo -n "c64 $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant_c64 --steps 5 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
done; done
for l in codex_africanus_amd/lib/ab/libafhip_prev.so codex_africanus_amd/lib/libafhip.so; do
  echo -n "c64 128 $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant_c64 --antennas 128 --steps 5 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
done
AFHIP_STRESS_ONLY=fused_gemm_c64_sweep timeout 900 python tools/stress_random.py 800000 6000 > gpurun_out/r6/stress_random_c64.log 2>&1; echo "stress rc $?"; tail -3 gpurun_out/r6/stress_random_c64.log
