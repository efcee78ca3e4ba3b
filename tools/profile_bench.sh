#!/bin/bash
# The three rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own
# passes (counters are never combined with API traces).  Run on the GPU box from the repository root:
#   bash tools/profile_bench.sh      (outputs under gpurun_out/, summarised by tools/summarize_profiles.py <tag>)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o bench --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-compare > gpurun_out/prof_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o bench --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-compare > gpurun_out/prof_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -o bench --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-compare > gpurun_out/prof_write.log 2>&1
grep "^{" gpurun_out/prof_stats.log | tail -1
