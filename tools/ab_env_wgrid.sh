#!/bin/bash
# same-box A/B of one env hook of the profiling build on the wgridder workload: tools/ab_env_wgrid.sh VAR "v1 v2 ..." [workload]
set -u
cd "$(dirname "$0")/.."
VAR=$1; VALS=$2; W=${3:-wgrid}
for round in 1 2; do for v in $VALS; do
echo -n "$VAR=$v: "; env AFHIP_LIB=codex_africanus_amd/lib/prof/libafhip.so $VAR=$v timeout 600 python3 bench.py --workload $W --extras none --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'))
"; done; done
