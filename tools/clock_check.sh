#!/bin/bash
# Cycles of k_accumulate (tabled BN254 MSM, 2^24 points) per library build, from two independent sources:
#   (1) the in-kernel stamps of this round's library (s_memtime / s_memrealtime around the launch, tools/clock_check.py), un-profiled;
#   (2) rocprofv3 --pmc GRBM_GUI_ACTIVE of the same command: counter / 8 = shader cycles of the launch (the profiler sums the 8 XCDs),
#       counter / 8 / duration = the clock it ran at -- the cross-check of (1), and the only source for builds that predate the stamps.
# The builds alternate (A B A B ...) so that a warming chip does not favour one of them.
# usage (on the GPU box): [CURVE=1] bash tools/clock_check.sh <tag> <lib> [<lib> ...]      writes gpurun_out/clock_check_<tag>.txt
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/clock_check_$tag.txt
: > $out
for round in 1 2; do
for lib in "$@"; do
    name=$(basename $lib .so)
    PANDA_LIB=$PWD/$lib timeout -k 10 200 python3 tools/clock_check.py 24 12 ${CURVE:-0} 2>/dev/null | grep CLOCK_CHECK | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().split(' ', 1)[1]); m = d['mean']
print('round $round  %-26s un-profiled : k_accumulate %.3f ms (HIP events)' % (d['lib'], m['k_accumulate_ms']) + ('   stamps: slowest XCD %.3f Mcycles, mean of the XCDs %.3f Mcycles at %.0f MHz (XCDs %.0f .. %.0f MHz; %.3f ms, %d XCDs)' % (m['mcycles'], m['mcycles_mean'], m['sclk_mhz'], m['mhz_min'], m['mhz_max'], m['stamp_ms'], m['xcds']) if 'mcycles' in m else '   (no stamps in this build)'))
" >> $out
    rm -rf gpurun_out/clk_$name
    PANDA_LIB=$PWD/$lib timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/clk_$name -o clk --output-format csv -- python3 tools/clock_check.py 24 12 ${CURVE:-0} > gpurun_out/clk_$name.log 2>&1
    python3 - $name $round >> $out <<'PY'
import csv, glob, json, sys
name, rnd = sys.argv[1], sys.argv[2]
cc = glob.glob(f"gpurun_out/clk_{name}/**/*counter_collection.csv", recursive=True)[0]
v = []
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "k_accumulate" not in r["Kernel_Name"]:
        continue
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    v.append((float(r["Counter_Value"]) / 8, dur))
v = v[3:]
cyc = sum(x[0] for x in v) / len(v) / 1e6
ms = sum(x[1] for x in v) / len(v) / 1e6
line = [l for l in open(f"gpurun_out/clk_{name}.log") if l.startswith("CLOCK_CHECK")]
m = json.loads(line[-1].split(" ", 1)[1])["mean"] if line else {}
s = "round %s  %-26s profiled    : GRBM_GUI_ACTIVE/8 = %.3f Mcycles, %.3f ms -> %.0f MHz (%d launches)" % (rnd, "libpanda-" + name.replace("libpanda-", "") + ".so", cyc, ms, cyc / ms * 1e3, len(v))
if "mcycles" in m:
    s += "   stamps in the same run: mean of the XCDs %.3f Mcycles at %.0f MHz (stamps / counter = %.4f), slowest XCD %.3f" % (m["mcycles_mean"], m["sclk_mhz"], m["mcycles_mean"] / cyc, m["mcycles"])
print(s)
PY
done
done
cat $out
