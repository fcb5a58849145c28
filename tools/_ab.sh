set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -k "even_grid or pointwise" 2>&1 | tail -3
timeout -k 10 200 python -m pytest tests/test_model_gpu.py -x -q -k "resnet50_whole" 2>&1 | tail -3
for v in 1 1; do ICAMD_SUB2_SHORTCUT=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('sub2=$v', d['ms_per_step'])"; done
