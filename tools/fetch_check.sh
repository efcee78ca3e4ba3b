#!/bin/bash
# FETCH_SIZE of k_accumulate (tabled BN254 MSM 2^24, tools/clock_check.py) per library build: rocprofv3 --pmc FETCH_SIZE, bytes per launch
# (counter KB x 1024; the kernel's reads are 64-byte sectors, which the counter tallies exactly: profiles/r02_fetch_size_calibration.txt).
# usage (on the GPU box): bash tools/fetch_check.sh <tag> <lib> [<lib> ...]     writes gpurun_out/fetch_check_<tag>.txt
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fetch_check_$tag.txt
: > $out
for lib in "$@"; do
    name=$(basename $lib .so)
    rm -rf gpurun_out/fch_$name
    PANDA_LIB=$PWD/$lib timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/fch_$name -o f --output-format csv -- python3 tools/clock_check.py 24 6 ${CURVE:-0} > gpurun_out/fch_$name.log 2>&1
    python3 - $name >> $out <<'PY'
import csv, glob, sys
name = sys.argv[1]
v = [float(r["Counter_Value"]) * 1024 for r in csv.DictReader(open(glob.glob(f"gpurun_out/fch_{name}/**/*counter_collection.csv", recursive=True)[0]))
     if r["Counter_Name"] == "FETCH_SIZE" and "k_accumulate" in r["Kernel_Name"]]
v = v[3:]
print("%-28s k_accumulate FETCH_SIZE %.3f GB per launch (%d launches)" % (name + ".so", sum(v) / len(v) / 1e9, len(v)))
PY
done
cat $out
