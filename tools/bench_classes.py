"""Print value / ms_per_step and the per-class kernel times of one bench.py JSON line (stdin). usage: bench.py ... | bench_classes.py [tag]"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read())
print(tag, d["value"], "img/s", d["ms_per_step"], "ms/step")
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print("   %-12s %7.3f ms  %5.1f calls" % (k, v["ms_per_step"], v["calls_per_step"]))
