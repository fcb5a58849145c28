set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for op in fwd dgrad; do
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/res_$op -- python3 $R/tools/one_layer.py 64 64 3 1 56 10 $op > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/res_$op/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'conv' in r['Name']: print("$op", r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
cd $R
for v in 0 1; do ICAMD_CONV3X3_RESIDENT=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('resident=$v', d['ms_per_step'])"; done
