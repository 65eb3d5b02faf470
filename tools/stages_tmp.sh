run() { python bench.py --workload fused_dde_ant --steps 3 --warmup 1 --extras none --no-cpu-baseline --check-rows ${CR:-0} | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['ms_per_step'], r['roofline']['kernel_ms'], r['fp64_max_abs_err'])"; }
CR=64 run full
AFHIP_FUSED_STAGE=1 run stage1
AFHIP_FUSED_STAGE=2 run stage2
AFHIP_GEMM_PRIO=-1 run noprio
