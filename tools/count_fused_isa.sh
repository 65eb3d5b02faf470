#!/bin/bash
# Instruction mix of the fused predict's hot loops (the unrolled, grouped, wave-specialised instantiation that runs
# BASELINE configs[2]): the consumer loop covers one batch = 8 sources x 4 rows per lane, the producer loop one batch
# = two super-rounds of 256 Jones terms per 4 sampling waves.  bench.py's roofline.executed quotes these counts.
set -e
cd "$(dirname "$0")/../codex_africanus_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only af_fused_predict.hip -o /tmp/af_fused.s 2>/dev/null
awk '/^_ZN12_GLOBAL__N_120fused_predict_kernelILb0ELb0ELi64ELb1ELi8ELb1E.*:/{on=1} on{print} /\.amdhsa_kernel _ZN12_GLOBAL__N_120fused_predict_kernelILb0ELb0ELi64ELb1ELi8ELb1E/{on=0}' /tmp/af_fused.s > /tmp/af_fused_k.s
python3 - <<'PY'
import re
lines = open('/tmp/af_fused_k.s').read().splitlines()
bar = [i for i, l in enumerate(lines) if 's_barrier' in l]
# barriers: [setup, producer-loop barrier, consumer-loop barrier]; loops are the code between a loop header and its barrier
def mix(lo, hi):
    from collections import Counter
    c = Counter()
    for l in lines[lo:hi]:
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        c[t.split()[0]] += 1
    return c
def summary(name, c):
    f64 = sum(v for k, v in c.items() if k.endswith('_f64') or '_f64_' in k)
    lds = sum(v for k, v in c.items() if k.startswith('ds_'))
    vmem = sum(v for k, v in c.items() if k.startswith('global_'))
    valu = sum(v for k, v in c.items() if k.startswith('v_')) - f64
    print("%s: fp64 VALU %d, other VALU %d, LDS %d, global %d, s_waitcnt %d" % (name, f64, valu, lds, vmem, c.get('s_waitcnt', 0)))
prod_end = max(i for i in range(bar[1], bar[2]) if 'ds_write_b128' in lines[i]) + 1
summary("producer batch (2 super-rounds, per sampling lane)", mix(bar[1], prod_end))
cons_hdr = bar[2]
cons_end = next(i for i in range(cons_hdr, len(lines)) if re.match(r'\.LBB\d+_\d+:', lines[i]) and i > cons_hdr + 1000)
summary("consumer batch (8 sources x 4 rows per lane)", mix(cons_hdr, cons_end))
PY
