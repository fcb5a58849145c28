set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "maxpool or bn_bwd or bn_backward" 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -x -q -k "resnet18 or resnet50_whole" 2>&1 | tail -2
for v in 0 1 0 1; do ICAMD_FUSED_POOL_BWD=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('fused_pool_bwd=$v', d['ms_per_step'], {k:v.get('ms_per_step') for k,v in d.get('kernels',{}).items() if k in ('bn_bwd','pool')})"; done
