#!/bin/bash
# rocprofv3 kernel trace of one tabled MSM in R equal point ranges: bash tools/equal_ranges_trace.sh <log_n> <R>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/eqtrace_$1_$2
rm -rf "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT" -o t --output-format csv -- python3 tools/equal_ranges_bench.py "$1" "$2" 1 1 > "$OUT.log" 2>&1
python3 tools/timeline.py $(find "$OUT" -name "*kernel_trace.csv" | head -1) k_digits > "$OUT.txt"
