"""Print per-kernel PMC counter sums from a rocprofv3 rocpd database. Usage: pmc_dump.py <results.db> [name-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print([t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()]); sys.exit(0)
cols = [d[0] for d in db.execute("select * from %s limit 1" % view).description]
kn = "kernel_name" if "kernel_name" in cols else "name"
q = "select %s, counter_name, count(*), sum(value) from %s where %s like ? group by 1, 2 order by 1, 2" % (kn, view, kn)
for r in db.execute(q, ("%" + sub + "%",)):
    print("%-50s %-28s n=%4d  sum=%.4g  avg=%.4g" % (r[0][:50], r[1], r[2], r[3], r[3] / r[2]))
