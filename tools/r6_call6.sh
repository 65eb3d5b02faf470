line() { python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'))
"; }
mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_gpu_fused_gemm_c64.py tests/test_gpu_wgridder.py -x -q > gpurun_out/r6/tests_c.log 2>&1; echo "c64 + wgrid tests rc $?"; tail -8 gpurun_out/r6/tests_c.log
for round in 1 2; do for l in codex_africanus_amd/lib/ab/libafhip_head.so codex_africanus_amd/lib/libafhip.so; do
  echo -n "wgrid $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload wgrid --extras none --no-cpu-baseline 2>/dev/null | line
done; done 2>&1 | tee gpurun_out/r6/ab_wgrid_c.log
for l in codex_africanus_amd/lib/ab/libafhip_head.so codex_africanus_amd/lib/libafhip.so; do
  echo -n "wgrid_f32planes $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload wgrid_f32planes --extras none --no-cpu-baseline 2>/dev/null | line
  echo -n "c64 128 antennas $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant_c64 --antennas 128 --steps 3 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
  echo -n "c64 197 antennas $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant_c64 --antennas 197 --steps 3 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
done 2>&1 | tee -a gpurun_out/r6/ab_wgrid_c.log
export TMPDIR=/tmp
ARGS="bench.py --workload wgrid --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --check-rows 0 --extras none"
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
     --kernel-trace --output-format csv -d gpurun_out/r6/wgrid_lds_new -o lds -- python3 $ARGS > gpurun_out/r6/wgrid_lds_new.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for f in glob.glob('gpurun_out/r6/wgrid_lds_new/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'wg_degrid_tiles' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
    print('new', {k:round(v/max(n[k],1)) for k,v in acc.items()})
PY
