#!/bin/bash
# rocprofv3 evidence for one round: every bench.py workload (kernel-trace stats + PMC passes, each in its own run:
# counters are never combined with other trace domains) and the auxiliary benches.  On the GPU box, from the repo root:
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh'
# then here: python tools/summarize_round.py r05
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_round
rm -rf "$OUT"; mkdir -p "$OUT"
WORKLOADS="${WORKLOADS:-dft dft_complex dft_f32 gauss fused_dde fused_dde_ant fused_dde_ant128 fused_dde_ant_c64 fused_dde_c64 degrid wgrid wgrid_f32planes}"
T="timeout 900"    # a profiler pass that hangs must not take the box with it
# the line the driver gets: headline + every other single-GPU workload under "workloads"
# (SKIP_DEFAULT=1 / SKIP_AUX=1: re-profile a subset of WORKLOADS only)
[ "${SKIP_DEFAULT:-0}" = 1 ] || python3 bench.py > "$OUT/default_line.json" 2> "$OUT/default_stderr.log"
for w in $WORKLOADS; do
    mkdir -p "$OUT/$w"
    # 10 timed steps after 2 warm-up steps: the kernel-trace average then covers mostly warm launches (the first launch
    # of a process is 5-10 % slower: clocks, instruction cache) and agrees with the HIP-event figure of the bench line
    ARGS="bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --check-rows 0 --extras none"
    python3 bench.py --workload $w --extras none > "$OUT/$w/bench_line.json" 2> "$OUT/$w/bench_stderr.log"
    $T rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w/stats" -o stats -- python3 $ARGS > "$OUT/$w/stats.log" 2>&1
    $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/$w/fetch" -o fetch -- python3 $ARGS > "$OUT/$w/fetch.log" 2>&1
    $T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/$w/write" -o write -- python3 $ARGS > "$OUT/$w/write.log" 2>&1
    $T rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d "$OUT/$w/sq" -o sq -- python3 $ARGS > "$OUT/$w/sq.log" 2>&1
    $T rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM \
        --kernel-trace --output-format csv -d "$OUT/$w/sq2" -o sq2 -- python3 $ARGS > "$OUT/$w/sq2.log" 2>&1
done
# auxiliary benches: kernel-trace stats of each (per-kernel durations of the real launches)
[ "${SKIP_AUX:-0}" = 1 ] && exit 0
for t in bench_degridder bench_wgridder bench_vis_to_im bench_wsclean bench_api_kernels bench_im_to_vis_ncorr bench_predict_tile bench_im_to_vis_f32 bench_apply_gains ab_beam_cube bench_gauss_dft; do
    [ -f tools/$t.py ] || continue
    mkdir -p "$OUT/aux/$t"
    python3 tools/$t.py > "$OUT/aux/$t/result.json" 2> "$OUT/aux/$t/stderr.log"
    $T rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/aux/$t/stats" -o stats -- python3 tools/$t.py > "$OUT/aux/$t/stats.log" 2>&1
    $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/aux/$t/fetch" -o fetch -- python3 tools/$t.py > "$OUT/aux/$t/fetch.log" 2>&1
    $T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/aux/$t/write" -o write -- python3 tools/$t.py > "$OUT/aux/$t/write.log" 2>&1
done
find "$OUT" -name "*kernel_stats.csv" | wc -l
for f in "$OUT"/*/bench_line.json; do tail -c 300 "$f"; echo; done
