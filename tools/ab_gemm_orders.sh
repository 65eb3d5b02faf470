cd /root/repo
for cfg in "0 0" "6 32" "0 0"; do set -- $cfg; echo -n "order_b=$1 delay=$2: "; AFHIP_GEMM_ORDER_B=$1 AFHIP_GEMM_BURST_DELAY=$2 python3 bench.py --workload fused_dde_ant --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --extras none 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']; print('kernel_ms', round(r['kernel_ms'],2), 'err', d['fp64_max_abs_err'])
"; done
