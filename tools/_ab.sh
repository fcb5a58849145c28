set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for shape in "256 1024 1 1 14" "1024 256 1 1 14" "512 2048 1 1 7" "2048 512 1 1 7" "512 128 1 1 28" "256 512 1 1 28"; do
for v in 0 1; do
rm -rf $R/gpurun_out/gl
ICAMD_GEMM_NT=$v timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gl -- python3 $R/tools/one_layer.py $shape 10 fwd,dgrad > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/gl/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'conv' in r['Name'] or 'gemm' in r['Name']: print("$shape nt=$v", r['Name'][27:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
done
