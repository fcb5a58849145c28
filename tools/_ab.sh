set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for shape in "64 64 3 1 56" "256 256 3 1 14"; do
rm -rf $R/gpurun_out/gl
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gl -- python3 $R/tools/one_layer.py $shape 10 wgrad > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/gl/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'wgrad' in r['Name'] or 'slab' in r['Name']: print("$shape", r['Name'][27:75], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
