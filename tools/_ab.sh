set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "pool" 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -x -q -k "resnet18" 2>&1 | tail -3
for i in 1 2; do timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('pool', d['ms_per_step'], {k:v.get('ms_per_step') for k,v in d.get('kernels',{}).items() if k in ('pool','pack','bn_bwd')})"; done
