set -x
mkdir -p gpurun_out/r6
# 1. the forensics path itself, on a forced one-ulp mismatch (2 ranks, small)
AFHIP_BENCH_FORCE_MISMATCH=1 AFHIP_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --executor ranks --workload fused_dde_ant --steps 1 --warmup 0 --rows 20160 --sources 60 --no-cpu-baseline > gpurun_out/r6/forced.out 2> gpurun_out/r6/forced.err; echo "forced rc $?"
ls gpurun_out/front_end_mismatch_* | head; mkdir -p gpurun_out/r6/forced; mv gpurun_out/front_end_mismatch_* gpurun_out/r6/forced/
# 2. timing-perturbation stress (profiling build)
AFHIP_LIB=codex_africanus_amd/lib/prof/libafhip.so timeout 900 python tools/stress_gemm_jitter.py --seeds 6 > gpurun_out/r6/jitter.log 2>&1; echo "jitter rc $?"
tail -3 gpurun_out/r6/jitter.log
# 3. the 8-ranks-on-one-device job, again and again
timeout 1500 python tools/stress_bench_ranks.py --runs 200 --repeats 4 --seconds 1300 > gpurun_out/r6/stress_ranks.log 2>&1; echo "stress rc $?"
tail -5 gpurun_out/r6/stress_ranks.log
