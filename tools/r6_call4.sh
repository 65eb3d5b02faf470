mkdir -p gpurun_out/r6
line() { python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'), d['config'].get('front_end','')[:60])
"; }
timeout 1200 python -m pytest tests/test_gpu_fused_gemm_c64.py -x -q > gpurun_out/r6/tests_c64.log 2>&1; echo "c64 tests rc $?"; tail -25 gpurun_out/r6/tests_c64.log
timeout 1500 python -m pytest tests/test_gpu_fused_gemm.py -x -q > gpurun_out/r6/tests_b.log 2>&1; echo "gemm tests rc $?"; tail -8 gpurun_out/r6/tests_b.log
for round in 1 2; do for w in fused_dde_ant fused_dde_ant_c64; do
  echo -n "$w: "; timeout 600 python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2> gpurun_out/r6/bench_$w.err | line
done; done 2>&1 | tee gpurun_out/r6/ab_c64.log
tail -3 gpurun_out/r6/bench_fused_dde_ant_c64.err
echo -n "c64 128 antennas: "; timeout 600 python3 bench.py --workload fused_dde_ant_c64 --antennas 128 --steps 3 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
echo -n "fp64 512 antennas: "; timeout 600 python3 bench.py --workload fused_dde_ant --antennas 512 --steps 2 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end --check-rows 16 2>gpurun_out/r6/bench_512.err | line
tail -2 gpurun_out/r6/bench_512.err
echo -n "row kernel 512 antennas: "; AFHIP_FUSED_GEMM=0 timeout 900 python3 bench.py --workload fused_dde --uvw antennas --antennas 512 --steps 2 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end --check-rows 16 2>/dev/null | line
