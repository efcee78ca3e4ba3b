#!/bin/bash
# How many bytes does the memory system FETCH for random 64-byte row gathers, by table size?  rocprofv3 --pmc FETCH_SIZE over
# tools/bin/ubench_gather_rate (4.295 GB requested per launch: 2^18 lanes x 256 rows x 64 B; table sizes 0.125 ... 12 GiB; per size first the
# per-lane gather, then four lanes to a row).  Anything above the requested bytes at the large sizes is not data: address translation.
# usage (on the GPU box): bash tools/gather_fetch.sh      writes gpurun_out/gather_fetch.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/gfetch
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/gfetch -o g --output-format csv -- tools/bin/ubench_gather_rate > gpurun_out/gfetch.log 2>&1
python3 - > gpurun_out/gather_fetch.txt <<'PY'
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob("gpurun_out/gfetch/**/*counter_collection.csv", recursive=True)[0])) if r["Counter_Name"] == "FETCH_SIZE" and "k_gather" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
sizes = [0.125, 0.5, 1.0, 2.0, 3.0, 4.0, 6.0, 8.0, 12.0]
req = 262144 * 4 * 256 * 64 / 4  # threads = 256 CUs * 4 SIMDs * 64 lanes * 4 waves
req = 256 * 4 * 64 * 4 * 256 * 64
print("table GiB   kernel            FETCH_SIZE GB / launch   requested GB   ratio   launch ms")
i = 0
for gib in sizes:
    for name in ("per-lane rows", "four lanes/row"):
        grp = rows[i:i + 6][1:]
        i += 6
        f = sum(float(r["Counter_Value"]) for r in grp) / len(grp) * 1024
        ms = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp) / len(grp) / 1e6
        print("%8.3f    %-16s  %10.3f               %6.3f        %5.3f   %7.3f" % (gib, name, f / 1e9, req / 1e9, f / req, ms))
PY
cat gpurun_out/gather_fetch.txt
