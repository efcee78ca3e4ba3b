#!/bin/bash
# SQ counters of k_ntt_pass8 (development aid): two rocprofv3 --pmc passes over tools/bin/ntt8_pb5_w3 (tools/ntt8_variants.hip)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BIN=${1:-tools/bin/ntt8_pb5_w3}
rm -rf gpurun_out/p8_sq1 gpurun_out/p8_sq2
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/p8_sq1 -o p8 --output-format csv -- $BIN 24 4 > gpurun_out/p8_sq1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM -d gpurun_out/p8_sq2 -o p8 --output-format csv -- $BIN 24 4 > gpurun_out/p8_sq2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/p8_sq1", "gpurun_out/p8_sq2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("<")[1][:40] if "<" in r["Kernel_Name"] else r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
        for k in acc:
            n = max(cnt[k], 1)
            print(k, " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(acc[k].items())))
PY
