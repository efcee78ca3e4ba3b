#!/bin/bash
# rocprofv3 kernel trace of the tabled MSM with the sort split over two streams (development aid):
#   bash tools/overlap_trace.sh <log_n> <front:wgs>      -> gpurun_out/ovtrace_<log_n>_<front>_<wgs>.txt (timeline of the last call)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=$1
COMBO=$2
TAG=$(echo "$COMBO" | tr ':' '_')
OUT=gpurun_out/ovtrace_${K}_${TAG}
rm -rf "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT" -o t --output-format csv -- python3 tools/overlap_bench.py "$K" "$COMBO" 2 > "$OUT.log" 2>&1
python3 tools/timeline.py $(find "$OUT" -name "*kernel_trace.csv" | head -1) k_digits > "$OUT.txt"
tail -30 "$OUT.txt"
