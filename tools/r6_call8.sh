line() { python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'))
"; }
mkdir -p gpurun_out/r6
for round in 1 2; do for l in codex_africanus_amd/lib/libafhip.so codex_africanus_amd/lib/ab/libafhip_c64_3m.so codex_africanus_amd/lib/ab/libafhip_c64_pk.so codex_africanus_amd/lib/ab/libafhip_c64_pkdaf_c64_3m.so; do
  echo -n "$l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant_c64 --steps 5 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
done; done 2>&1 | tee gpurun_out/r6/ab_c64_variants2.log
timeout 2000 python -m pytest tests -m gpu -x -q > gpurun_out/r6/full_gpu_suite.log 2>&1; echo "full suite rc $?"; tail -15 gpurun_out/r6/full_gpu_suite.log
