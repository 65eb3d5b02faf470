mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_fused_gemm.py tests/test_gpu_fused_gemm_c64.py tests/test_gpu_fused_frontends.py tests/test_gpu_fused.py -x -q > gpurun_out/r6/tests_g.log 2>&1; echo "tests rc $?"; tail -12 gpurun_out/r6/tests_g.log
