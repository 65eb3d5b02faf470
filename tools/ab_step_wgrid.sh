for round in 1 2; do for l in codex_africanus_amd/lib/ab/libafhip_head.so codex_africanus_amd/lib/libafhip.so; do
echo -n "$l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload wgrid --extras none --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), d.get('accuracy') or d.get('l2_error') or {k:v for k,v in d.items() if 'err' in k or 'l2' in k})
"; done; done
