#!/bin/bash
# The fused predict's HOST planners (af_fused_plan_antennas / af_fused_plan_groups, csrc/af_fused_gemm.hip and
# af_fused_predict.hip) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (host code only; GPU sanitizers are
# not available on the pool): the bench's row layouts, 200 times, identical plan bytes required every time.
#   tools/sanitize/run_planners.sh [reps]        (needs a built lib/obj: python -c "import __graft_entry__ as g; g.build()")
set -e
cd "$(dirname "$0")/../../codex_africanus_amd/csrc"
T=${TMPDIR:-/tmp}/afhip_asan; mkdir -p $T
F="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer"
/opt/rocm/bin/hipcc $F -c af_fused_gemm.hip -o $T/gemm.o
/opt/rocm/bin/hipcc $F -c af_fused_predict.hip -o $T/pred.o
/opt/rocm/lib/llvm/bin/clang++ -fsanitize=address,undefined -O1 -g -c ../../tools/sanitize/planners_driver.cpp -o $T/drive.o
OTHERS=$(ls ../lib/obj/*.o | grep -v "af_fused_gemm.o\|af_fused_predict.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address,undefined $T/drive.o $T/gemm.o $T/pred.o $OTHERS -o $T/drive -L/opt/rocm/lib -lhipfft
ASAN_OPTIONS=detect_leaks=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $T/drive ${1:-200}
