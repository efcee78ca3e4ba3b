"""Per-kernel summary of a rocprofv3 rocpd database (development aid): python tools/kstats.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(grid_x), max(grid_y), max(workgroup_x), max(vgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
for r in rows:
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    nm = nm.split("(")[0][:60]
    print("%-60s n=%3d avg=%9.1fus min=%9.1f grid=%dx%d wg=%d vgpr=%d lds=%d" % (nm, r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]))
