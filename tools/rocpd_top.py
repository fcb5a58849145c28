"""Per-kernel summary (calls, average us, total ms) of a rocprofv3 rocpd database, grouped by kernel name and grid.
Usage: python tools/rocpd_top.py <results.db> [rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 25
q = ("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6, grid_x, workgroup_x from kernels "
     "group by name, grid_x order by 4 desc limit %d" % rows)
tot = db.execute("select sum(end-start)/1e6 from kernels").fetchone()[0]
print("total kernel ms: %.2f" % tot)
for r in db.execute(q):
    print("%-62s calls %5d  avg %8.1f us  total %8.2f ms  grid %d/%d" % (r[0][:62], r[1], r[2], r[3], r[4] // r[5], r[5]))
