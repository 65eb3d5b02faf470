line() { python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'))
"; }
mkdir -p gpurun_out/r6
for round in 1 2; do for l in codex_africanus_amd/lib/libafhip.so codex_africanus_amd/lib/ab/libafhip_c64_x4.so codex_africanus_amd/lib/ab/libafhip_c64_orderb.so codex_africanus_amd/lib/ab/libafhip_c64_noprio.so codex_africanus_amd/lib/ab/libafhip_c64_x4daf_c64_orderb.so; do
  echo -n "$l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant_c64 --steps 5 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
done; done 2>&1 | tee gpurun_out/r6/ab_c64_variants.log
for l in codex_africanus_amd/lib/ab/libafhip_c64_stage_sample.so codex_africanus_amd/lib/ab/libafhip_c64_stage_matrix.so; do
  echo -n "$l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant_c64 --steps 5 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end --check-rows 0 2>/dev/null | line
done 2>&1 | tee -a gpurun_out/r6/ab_c64_variants.log
