#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of one bench.py configuration; prints the top kernels per step.
# Usage: tools/stats_run.sh <tag> <steps> <bench args...>
TAG=$1; STEPS=$2; shift 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline "$@" > $O/run.log 2>&1
cd $R
F=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$F" "$STEPS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
nst = 2 * steps + 2            # warm-up 2 + timed + profiled pass (bench.py runs min(steps, 10) more) -- approximate: printed per CALL too
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over the run" % (tot / 1e6))
for r in rows[:40]:
    print("%-100s calls %6d  total %9.3f ms  avg %8.1f us  %5.1f %%" % (r["Name"][:100], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
