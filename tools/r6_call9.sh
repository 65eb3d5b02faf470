line() { python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'))
"; }
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_fused_gemm_c64.py -x -q > gpurun_out/r6/tests_c64.log 2>&1; echo "c64 tests rc $?"; tail -5 gpurun_out/r6/tests_c64.log
for a in 64 128; do echo -n "c64 $a antennas: "; timeout 600 python3 bench.py --workload fused_dde_ant_c64 --antennas $a --steps 5 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line; done
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6/full_gpu_suite.log 2>&1; echo "full suite rc $?"; tail -15 gpurun_out/r6/full_gpu_suite.log
