python bench.py --workload wgrid --no-cpu-baseline 2>&1 | tail -1 | grep -o "\"ms_per_step\": [0-9.]*\|\"kernel_ms\": [0-9.]*\|\"fp64_max_abs_err\": [0-9.e-]*"
timeout 600 python -m pytest tests/test_gpu_wgridder.py tests/test_gpu_full_size.py -q -x -k "wgrid or model or tile" 2>&1 | tail -2
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
A="$R/bench.py --workload wgrid --steps 3 --warmup 1 --no-cpu-baseline --check-rows 0"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/wgp/s2 -o s2 -- python3 $A > /dev/null 2>&1
cd $R; python - <<PY
import csv, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/wgp/s2/s2_counter_collection.csv")):
    if "wg_degrid_tiles" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
