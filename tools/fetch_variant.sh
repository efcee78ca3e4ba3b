#!/bin/bash
# FETCH_SIZE of k_accumulate for an accumulate variant against the built-in kernel (tools/overlap_bench.py with PANDA_ACC_VARIANT), 2^24 tabled BN254
# usage (on the GPU box): bash tools/fetch_variant.sh <variant>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 $1; do
    rm -rf gpurun_out/fv_$v
    PANDA_ACC_VARIANT=$v timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/fv_$v -o f --output-format csv -- python3 tools/overlap_bench.py 24 0:0 5 > gpurun_out/fv_$v.log 2>&1
    python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
x = [float(r["Counter_Value"]) * 1024 for r in csv.DictReader(open(glob.glob(f"gpurun_out/fv_{v}/**/*counter_collection.csv", recursive=True)[0])) if r["Counter_Name"] == "FETCH_SIZE" and "k_accumulate" in r["Kernel_Name"]][2:]
print("accumulate variant %s: k_accumulate FETCH_SIZE %.3f GB per launch (%d launches)" % (v, sum(x) / len(x) / 1e9, len(x)))
PY
done
