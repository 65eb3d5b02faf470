mkdir -p gpurun_out/r6
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r6/tests_full_k.log 2>&1; echo "tests rc $?"; tail -8 gpurun_out/r6/tests_full_k.log
