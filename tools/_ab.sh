set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "stem7x7s2" 2>&1 | tail -3
timeout -k 10 300 python tools/bench_layers.py 256 10 2>&1 | grep -E "stem|total"
for v in 0 1 0 1; do ICAMD_STEM_RESIDENT=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('stem_resident=$v', d['ms_per_step'], d['kernels']['conv_fwd']['ms_per_step'], d['kernels']['conv_wgrad']['ms_per_step'])"; done
