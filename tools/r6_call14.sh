mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_fused_gemm_c64.py tests/test_gpu_chunked.py -x -q > gpurun_out/r6/tests_e.log 2>&1; echo "tests rc $?"; tail -6 gpurun_out/r6/tests_e.log
AFHIP_STRESS_ONLY=fused_gemm_sweep,fused_gemm_c64_sweep,fused_sweep,wgridder_sweep timeout 1700 python tools/stress_random.py 700000 10000 > gpurun_out/r6/stress_random_gemm2.log 2>&1; echo "stress rc $?"; tail -3 gpurun_out/r6/stress_random_gemm2.log
