mkdir -p gpurun_out/r6
line() { python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'), d['config'].get('front_end','')[:60])
"; }
timeout 1200 python -m pytest tests/test_gpu_fused_gemm_c64.py -x -q > gpurun_out/r6/tests_c64.log 2>&1; echo "c64 tests rc $?"; tail -25 gpurun_out/r6/tests_c64.log
timeout 1200 python -m pytest tests/test_gpu_fused_gemm.py tests/test_gpu_wgridder.py tests/test_gpu_fused_frontends.py -x -q > gpurun_out/r6/tests_b.log 2>&1; echo "gemm+wgrid tests rc $?"; tail -5 gpurun_out/r6/tests_b.log
for round in 1 2; do for l in codex_africanus_amd/lib/ab/libafhip_base.so codex_africanus_amd/lib/libafhip.so; do
  echo -n "ant128 $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant128 --steps 5 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
done; done 2>&1 | tee gpurun_out/r6/ab_ant128_b.log
for l in codex_africanus_amd/lib/ab/libafhip_base.so codex_africanus_amd/lib/libafhip.so; do
  echo -n "ant197 $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant --antennas 197 --steps 3 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
  echo -n "ant256 $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant --antennas 256 --steps 3 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
done 2>&1 | tee -a gpurun_out/r6/ab_ant128_b.log
for round in 1 2; do for w in fused_dde_ant fused_dde_ant_c64; do
  echo -n "$w: "; timeout 600 python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2> gpurun_out/r6/bench_$w.err | line
done; done 2>&1 | tee gpurun_out/r6/ab_c64.log
tail -3 gpurun_out/r6/bench_fused_dde_ant_c64.err
for round in 1 2; do for l in codex_africanus_amd/lib/ab/libafhip_base.so codex_africanus_amd/lib/libafhip.so; do
  echo -n "wgrid $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload wgrid --extras none --no-cpu-baseline 2>/dev/null | line
done; done 2>&1 | tee gpurun_out/r6/ab_wgrid_b.log
