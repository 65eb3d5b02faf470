#!/bin/bash
# Timeline evidence for the overlapped download: kernel trace + memory-copy trace (no counters) of the
# chunked two-stream pipeline of tools/bench_d2h_overlap.py at BASELINE configs[1].
#   gpurun --timeout 900 -- 'bash tools/profile_d2h_overlap.sh'   then   python tools/summarize_d2h_overlap.py r05
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_d2h
rm -rf "$OUT"; mkdir -p "$OUT"
python3 tools/bench_d2h_overlap.py --mode both > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 tools/bench_host_path.py > "$OUT/host_path.json" 2> "$OUT/host_path.err"
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -o d2h -- \
    python3 tools/bench_d2h_overlap.py --mode chunked --repeats 1 > "$OUT/trace.log" 2>&1
find "$OUT" -name "*.csv" | head; cat "$OUT/bench.json" "$OUT/host_path.json"
