line() { python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'))
"; }
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_fused_gemm_c64.py tests/test_gpu_wgridder.py tests/test_gpu_fused.py tests/test_gpu_fused_frontends.py -x -q > gpurun_out/r6/tests_d.log 2>&1; echo "tests rc $?"; tail -12 gpurun_out/r6/tests_d.log
for a in 64 128 197 512; do echo -n "c64 $a antennas: "; timeout 600 python3 bench.py --workload fused_dde_ant_c64 --antennas $a --steps 3 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end --check-rows 16 2>/dev/null | line; done
