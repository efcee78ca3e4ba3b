#!/bin/bash
# SQ / LDS counters of the radix-512 pass (development aid): two rocprofv3 --pmc passes over tools/ntt_bench.py 26 (9 + 9 + 8), summarised per kernel
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/p9_sq1 gpurun_out/p9_sq2
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/p9_sq1 -o p9 --output-format csv -- python3 tools/ntt_bench.py 26 2 > gpurun_out/p9_sq1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM -d gpurun_out/p9_sq2 -o p9 --output-format csv -- python3 tools/ntt_bench.py 26 2 > gpurun_out/p9_sq2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
for d in ("gpurun_out/p9_sq1", "gpurun_out/p9_sq2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_ntt_pass" not in k: continue
            m = re.search(r"(k_ntt_pass\d)<[^,]+, (true|false), (true|false), (\d), (\d)", k)
            k = "%s first=%s last=%s PB=%s" % m.groups()[:4] if m else k[:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
        for k in sorted(acc):
            n = max(cnt[k], 1)
            print(k, "launches=%d" % n, " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(acc[k].items())))
PY
