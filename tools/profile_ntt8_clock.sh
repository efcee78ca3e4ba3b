#!/bin/bash
# effective shader clock of k_ntt_pass8 (MI355X_MICROARCH.md, "DVFS give-back"): GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BIN=${1:-tools/bin/ntt8_pb5_w3_B}
rm -rf gpurun_out/p8_clk
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/p8_clk -o p8 --output-format csv -- $BIN 24 20 > gpurun_out/p8_clk.log 2>&1
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/p8_clk/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "k_ntt_pass8" not in r["Kernel_Name"]:
        continue
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    role = "first" if "true, false" in r["Kernel_Name"] else ("last" if "false, true" in r["Kernel_Name"] else "middle")
    acc[role].append((float(r["Counter_Value"]) / 8 / dur, dur / 1e6))
for role, v in acc.items():
    v = v[4:]  # skip warm-up launches
    print(role, "launches", len(v), "mean clock %.3f GHz" % (sum(x[0] for x in v) / len(v)), "mean duration %.3f ms" % (sum(x[1] for x in v) / len(v)))
PY
