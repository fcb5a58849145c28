set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "dgrad or pointwise" 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -x -q -k "resnet18 or resnet50_whole" 2>&1 | tail -2
for v in 2 1 2 1; do ICAMD_IGEMM_LEAN=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('lean=$v', d['ms_per_step'], d['kernels']['conv_dgrad']['ms_per_step'])"; done
