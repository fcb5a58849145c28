#!/bin/bash
# GPU box: bn_bwd_reduce / bn_bwd_apply kernel time (rocprofv3 --stats) per ICAMD_BNBWD_BLOCKS value and activation shape
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for nb in 512 1024 2048 4096; do
  for shp in "256 56 56 64" "256 28 28 128" "256 14 14 256" "256 14 14 1024" "256 56 56 256"; do
    rm -rf /tmp/bnp; ICAMD_BNBWD_BLOCKS=$nb rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bnp -- python3 $R/tools/bench_bn.py $shp > /tmp/bnp.log 2>&1
    f=$(find /tmp/bnp -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "$nb" "$shp" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
get = lambda k: next((float(r["AverageNs"]) / 1e3 for r in rows if k in r["Name"]), float("nan"))
print("blocks=%s shape=[%s] reduce %.1f us  apply %.1f us  bn_apply %.1f us" % (sys.argv[2], sys.argv[3], get("bn_bwd_reduce"), get("bn_bwd_apply"), get("bn_apply_kernel")))
PY
  done
done
