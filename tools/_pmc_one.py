"""Print per-kernel sums of every counter in a rocprofv3 --pmc results db.  usage: python tools/_pmc_one.py <db>"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cols = [d[0] for d in db.execute("select * from counters_collection limit 1").description]
kn = "kernel_name" if "kernel_name" in cols else "name"
rows = db.execute("select %s, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by 1,2" % kn).fetchall()
for name, c, n, v in rows:
    if "conv" in name or "gemm" in name or "halo" in name:
        print(re.sub(r"\(.*$", "", name)[-60:], c, n, v / max(n, 1))
