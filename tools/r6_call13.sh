mkdir -p gpurun_out/r6
AFHIP_STRESS_ONLY=fused_gemm_sweep,fused_gemm_c64_sweep,fused_sweep,wgridder_sweep timeout 1500 python tools/stress_random.py 600000 1500 > gpurun_out/r6/stress_random_gemm.log 2>&1; echo "stress rc $?"; tail -3 gpurun_out/r6/stress_random_gemm.log
