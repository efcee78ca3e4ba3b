#!/bin/bash
# The rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and the SQ issue counters in
# passes of their own (counters are never combined with API traces; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Run on the GPU box from the repository root:
#   bash tools/profile_bench.sh      (outputs under gpurun_out/, summarised by tools/summarize_profiles.py <tag>)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
PMC_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-compare --no-config4 --no-extra-configs --no-ntt-sweep"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o bench --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-compare --no-config4 --no-extra-configs --no-ntt-sweep > gpurun_out/prof_stats.log 2>&1
echo "stats pass done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o bench --output-format csv -- python3 bench.py $PMC_ARGS > gpurun_out/prof_fetch.log 2>&1
echo "fetch pass done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -o bench --output-format csv -- python3 bench.py $PMC_ARGS > gpurun_out/prof_write.log 2>&1
echo "write pass done"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/prof_sq -o bench --output-format csv -- python3 bench.py $PMC_ARGS > gpurun_out/prof_sq.log 2>&1
echo "sq pass done"
grep "^{" gpurun_out/prof_stats.log | tail -1
