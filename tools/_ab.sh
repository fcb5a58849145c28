set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "register_resident_filter" 2>&1 | tail -2
for v in 1 3 1 3; do ICAMD_PW_RESIDENT=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('pw_resident=$v', d['ms_per_step'], d['kernels']['conv_dgrad']['ms_per_step'])"; done
