"""Kernel timeline of the LAST call in a rocprofv3 --kernel-trace csv (development aid): start offset, duration and the gap to
the previous kernel's end, for the launches after the last occurrence of <first_kernel_substring>.
usage: timeline.py <kernel_trace.csv> [first_kernel_substring=k_digits]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_digits"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if first in r["Kernel_Name"])
# the sample check / memset in front of the digits kernel belong to the call too
while last > 0 and ("k_check_samples" in rows[last - 1]["Kernel_Name"]):
    last -= 1
t0 = int(rows[last]["Start_Timestamp"])
prev_end = t0
tot_busy = 0
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    print("%-48s start=%9.1f us  dur=%8.1f us  gap=%6.1f us  grid=%s wg=%s vgpr=%s" % (nm, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?"))))
    prev_end = max(prev_end, e)
    tot_busy += e - s
print("span %.1f us, busy %.1f us, launches %d" % ((prev_end - t0) / 1e3, tot_busy / 1e3, len(rows) - last))
