#!/bin/bash
# Same-box comparison of several builds of libafhip.so: tools/ab_libs.sh "a.so b.so ..." <bench args...>  (two rounds)
set -u
cd "$(dirname "$0")/.."
LIBS=$1; shift
for round in 1 2; do
  for l in $LIBS; do
    echo -n "$l: "
    AFHIP_LIB=$l timeout 600 python3 bench.py "$@" 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']
        print('kernel_ms', round(r['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'))
"
  done
done
