#!/bin/bash
# Same-box A/B of two builds of libafhip.so: tools/ab_lib.sh <exp.so> <bench args...>
# (the experimental library is built beside the shipped one, e.g. codex_africanus_amd/lib/libafhip_exp.so; built .so
# files travel with the gpurun snapshot).  Prints the kernel ms of base / exp / base / exp.  The library is chosen per
# process through AFHIP_LIB (codex_africanus_amd/_lib.py): nothing is overwritten, an interrupted run leaves no trace.
set -u
cd "$(dirname "$0")/.."
EXP=$1; shift
for round in 1 2; do
  for which in base exp; do
    if [ $which = base ]; then unset AFHIP_LIB; else export AFHIP_LIB=$EXP; fi
    echo -n "$which: "
    timeout 600 python3 bench.py "$@" 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']
        print('ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', r.get('kernel_ms', r.get('avg_ms')), 'err', d.get('max_abs_err', d.get('parity')))
"
  done
done
