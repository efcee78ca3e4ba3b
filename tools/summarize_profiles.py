"""Turns the rocprofv3 CSVs merged into gpurun_out/ (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into the small
summaries committed under profiles/.  usage: python tools/summarize_profiles.py <tag>   e.g. r01b"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GATHER_KERNELS = {"k_accumulate"}  # dominated by 64-byte row gathers
GATHER_FACTOR = 1.0


def latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, pattern), recursive=True), key=os.path.getmtime)
    return files[-1]


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+)", k)
    name = m.group(1) if m else k[:30]
    name = name.split("::")[-1]
    p8 = re.search(r"k_ntt_pass8<[^,]+, (true|false), (true|false)", k)
    if p8:  # the three roles of the register-resident radix-256 pass are three kernels
        name += "_first" if p8.group(1) == "true" else ("_last" if p8.group(2) == "true" else "_middle")
    return name


def by_grid(tag):
    """rocprofv3's kernel_stats.csv averages a kernel over ALL its launches -- k_accumulate's 2^24-point launches together with the 2^16-point
    launch of the config-1 check (the printed 12.75 ms was not the 2^24 figure).  This table splits every kernel's launches by
    grid size, from the kernel trace of the same pass."""
    trace = latest("gpurun_out/prof_stats/**/*kernel_trace.csv")
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        grid = "x".join(str(r.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else str(r.get("Grid_Size", ""))
        wg = "x".join(str(r.get(k, "")) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")) if "Workgroup_Size_X" in r else str(r.get("Workgroup_Size", ""))
        groups[(short(r["Kernel_Name"]), grid, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    path = os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_by_grid.csv")
    with open(path, "w") as o:
        o.write("# rocprofv3 --kernel-trace of the stats pass (tools/profile_bench.sh), every kernel's launches split by grid size; durations in ms\n")
        o.write("kernel,grid_threads,workgroup,launches,avg_ms,min_ms,max_ms,total_ms\n")
        for (k, grid, wg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            if re.match(r"k\d?_", k):
                o.write("%s,%s,%s,%d,%.4f,%.4f,%.4f,%.3f\n" % (k, grid, wg, len(v), sum(v) / len(v), min(v), max(v), sum(v)))
    print(open(path).read()[:2500])


def main():
    tag = sys.argv[1]
    shutil.copy(latest("gpurun_out/prof_stats/**/*kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats.csv"))
    by_grid(tag)
    res = {}
    for name in ("fetch", "write"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(latest(f"gpurun_out/prof_{name}/**/*counter_collection.csv"))):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        res[name] = agg
    rows = []
    for k in sorted(res["fetch"]):
        if not re.match(r"k\d?_", k):
            continue
        fv, wv = res["fetch"][k], res["write"].get(k, [0])
        f, w = sum(fv) / len(fv) * 1024, sum(wv) / len(wv) * 1024
        # FETCH_SIZE tallies a 128-byte request of a wide streaming read at 64 bytes (factor 2, MI355X_MICROARCH.md) but a
        # 64-byte row gather exactly (factor 1: tools/ubench_gather.hip, profiles/*_fetch_size_calibration.txt)
        factor = GATHER_FACTOR if k in GATHER_KERNELS else 2.0
        rows.append((k, len(fv), f, factor * f, w, factor * f + w))
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_bytes.csv"), "w") as o:
        o.write("# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/profile_bench.sh) of: python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compare --no-config4 --no-extra-configs (BN254 MSM 2^24, precomputed tables; NTT 2^24)\n")
        o.write("# bytes per launch = counter (KB) * 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16 B/lane streaming reads),\n# except k_accumulate: its reads are 64-byte row gathers, which the counter tallies exactly (calibrated with tools/ubench_gather.hip: 4.295 GB requested, 4.295 GB counted);\n")
        o.write("# k_convert_bases calibrates the correction: it reads 2^24 * 64 B = 1.074 GB\n")
        o.write("kernel,launches,fetch_size_raw_bytes,fetch_bytes_corrected,write_bytes,hbm_bytes_per_launch\n")
        for r in rows:
            o.write("%s,%d,%.0f,%.0f,%.0f,%.0f\n" % r)
    sq_files = glob.glob(os.path.join(ROOT, "gpurun_out/prof_sq/**/*counter_collection.csv"), recursive=True)
    if sq_files:
        # SQ issue counters per kernel: mean over the launches of (sum over the chip).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
        # quad-cycles per wave (MI355X_MICROARCH.md); the ratios to SQ_WAVE_CYCLES are what DESIGN.md quotes.
        sq = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(latest("gpurun_out/prof_sq/**/*counter_collection.csv"))):
            sq[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
        with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_sq.csv"), "w") as o:
            o.write("# rocprofv3 --pmc " + " ".join(names) + " (one pass, tools/profile_bench.sh) of: python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compare --no-config4 --no-extra-configs\n")
            o.write("# per launch, summed over the chip, mean over launches; the last columns are ratios to SQ_WAVE_CYCLES and VALU instructions per wave\n")
            o.write("kernel,launches," + ",".join(names) + ",active_valu_over_wave_cycles,wait_any_over_wave_cycles,wait_inst_any_over_wave_cycles,insts_valu_per_wave\n")
            for k in sorted(sq):
                if not re.match(r"k\d?_", k):
                    continue
                m = {nm: (sum(sq[k][nm]) / len(sq[k][nm]) if sq[k][nm] else 0.0) for nm in names}
                wc = m["SQ_WAVE_CYCLES"] or 1.0
                o.write("%s,%d,%s,%.4f,%.4f,%.4f,%.1f\n" % (k, len(sq[k]["SQ_WAVE_CYCLES"]), ",".join("%.0f" % m[nm] for nm in names), m["SQ_ACTIVE_INST_VALU"] / wc,
                                                           m["SQ_WAIT_ANY"] / wc, m["SQ_WAIT_INST_ANY"] / wc, m["SQ_INSTS_VALU"] / (m["SQ_WAVES"] or 1.0)))
        print(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_sq.csv")).read())
    acc = [r for r in rows if r[0] == "k_accumulate"][0]
    json.dump({"round": tag, "log_n": 24, "k_accumulate_hbm_bytes_per_launch": acc[5], "fetch_corrected": acc[3], "write": acc[4],
               "source": f"profiles/{tag}_pmc_hbm_bytes.csv"}, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    line = [l for l in open(os.path.join(ROOT, "gpurun_out", "prof_stats.log")) if l.startswith("{")]
    if line:
        open(os.path.join(ROOT, "profiles", f"{tag}_bench_line.json"), "w").write(line[-1])
    print(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_bytes.csv")).read())
    print(open(os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats.csv")).read()[:3000])


if __name__ == "__main__":
    main()
