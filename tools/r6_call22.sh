line() { python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'))
"; }
mkdir -p gpurun_out/r6
for round in 1 2; do for c in 512 1024 2048; do
  echo -n "wgrid SORT_BINS=$c: "; AFHIP_WGRID_SORT_BINS=$c timeout 600 python3 bench.py --workload wgrid --extras none --no-cpu-baseline 2>/dev/null | line
done; done 2>&1 | tee gpurun_out/r6/ab_wgrid_sortbins.log
