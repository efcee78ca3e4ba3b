#!/bin/bash
# effective shader clock of k_accumulate (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / duration) with the rows of every gather in HBM (the
# library) and in the caches (a build with -DPANDA_ROW_MASK=0x0003ffffu: 16 MiB of rows): is the difference between the two the clock?
# usage (on the GPU box): bash tools/accumulate_clock.sh <lib> [<lib> ...]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
    tag=$(basename $lib .so)
    rm -rf gpurun_out/acc_clk_$tag
    PANDA_LIB=$PWD/$lib timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/acc_clk_$tag -o acc --output-format csv -- python3 tools/overlap_bench.py 24 0:0 9 > gpurun_out/acc_clk_$tag.log 2>&1
    python3 - $tag <<'PY'
import csv, glob, sys
tag = sys.argv[1]
cc = glob.glob(f"gpurun_out/acc_clk_{tag}/**/*counter_collection.csv", recursive=True)[0]
v = []
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "k_accumulate" not in r["Kernel_Name"]:
        continue
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    v.append((float(r["Counter_Value"]) / 8 / dur, dur / 1e6))
v = v[3:]
print(tag, "launches", len(v), "mean clock %.3f GHz" % (sum(x[0] for x in v) / len(v)), "mean duration %.3f ms" % (sum(x[1] for x in v) / len(v)), flush=True)
PY
done
