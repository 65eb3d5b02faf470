set -x
mkdir -p gpurun_out/r6
line() { python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('step_ms', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'err', d.get('fp64_max_abs_err'))
"; }
# A. parity of what changed
timeout 1200 python -m pytest tests/test_gpu_fused_gemm.py tests/test_gpu_wgridder.py tests/test_gpu_chunked.py -x -q > gpurun_out/r6/tests_a.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r6/tests_a.log
# B. 128 antennas: base library (round 5's kernel) against the brightness hoist, two rounds
for round in 1 2; do for l in codex_africanus_amd/lib/ab/libafhip_base.so codex_africanus_amd/lib/libafhip.so; do
  echo -n "ant128 $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant128 --steps 5 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
done; done 2>&1 | tee gpurun_out/r6/ab_ant128.log
for l in codex_africanus_amd/lib/ab/libafhip_base.so codex_africanus_amd/lib/libafhip.so; do
  echo -n "ant197 $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant --antennas 197 --steps 3 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
  echo -n "ant40 $l: "; AFHIP_LIB=$l timeout 600 python3 bench.py --workload fused_dde_ant --antennas 40 --steps 3 --warmup 1 --no-cpu-baseline --extras none --no-end-to-end 2>/dev/null | line
done 2>&1 | tee -a gpurun_out/r6/ab_ant128.log
# C. wgridder tile pass: surplus lanes concentrated / chunks closed at bucket boundaries
for round in 1 2; do for v in "0 0" "1 0" "1 160" "1 224" "0 192"; do set -- $v
  echo -n "wgrid CONCENTRATE=$1 CLOSE=$2: "; AFHIP_WGRID_CONCENTRATE=$1 AFHIP_WGRID_CLOSE=$2 timeout 600 python3 bench.py --workload wgrid --extras none --no-cpu-baseline 2>/dev/null | line
done; done 2>&1 | tee gpurun_out/r6/ab_wgrid.log
# D. LDS counters of the tile pass, both dealings
export TMPDIR=/tmp
ARGS="bench.py --workload wgrid --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --check-rows 0 --extras none"
for c in 0 1; do
  AFHIP_WGRID_CONCENTRATE=$c timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
     --kernel-trace --output-format csv -d gpurun_out/r6/wgrid_lds_c$c -o lds -- python3 $ARGS > gpurun_out/r6/wgrid_lds_c$c.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for c in (0,1):
    for f in glob.glob('gpurun_out/r6/wgrid_lds_c%d/**/*counter_collection.csv'%c, recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            if 'wg_degrid_tiles' in r['Kernel_Name']:
                acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
        print('CONCENTRATE',c,{k:round(v/max(n[k],1)) for k,v in acc.items()})
PY
# E. f32 MFMA probe (layout, rate, overlap with VALU of another wave)
./tools/probe/probe_mfma_f32_16x16x4 > gpurun_out/r6/probe_mfma_f32.log 2>&1; cat gpurun_out/r6/probe_mfma_f32.log
# F. the bench-ranks test file (forensics test included)
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -x -q > gpurun_out/r6/tests_ranks.log 2>&1; echo "ranks tests rc $?"; tail -3 gpurun_out/r6/tests_ranks.log
