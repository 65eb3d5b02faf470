#!/bin/bash
# Same-box A/B of one environment switch: tools/ab_env.sh VAR "v1 v2 ..." <bench args...>
# Alternates the values twice (v1 v2 v1 v2) so that clock drift of the box shows; prints the dominant kernel's ms.
set -u
cd "$(dirname "$0")/.."
VAR=$1; VALUES=$2; shift 2
for round in 1 2; do
  for v in $VALUES; do
    echo -n "$VAR=$v: "
    env "$VAR=$v" timeout 900 python3 bench.py "$@" 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']
        print('ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(r['kernel_ms'],3), 'frac', round(r['frac'],4), 'err', d.get('fp64_max_abs_err'))
"
  done
done
