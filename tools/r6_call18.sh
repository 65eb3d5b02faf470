mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6/full_gpu_suite2.log 2>&1; echo "full suite rc $?"; tail -4 gpurun_out/r6/full_gpu_suite2.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
( time python bench.py > gpurun_out/r6/default_line.json 2> gpurun_out/r6/default_line.err ) 2>&1 | tail -3
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6/default_line.json').read().splitlines() if l.startswith('{')][-1])
print(d['metric'][:40], d['value'], d['ms_per_step'], d['roofline']['frac'], list(d['workloads'].keys()))
print(json.dumps(d['summary'])[:900])
PY
