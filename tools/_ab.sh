set -e
R=$GRAFT_REPO_ROOT
cd $R
for v in 0 1 0 1; do ICAMD_MAIN_HIGH_PRIO=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('main_high_prio=$v', d['ms_per_step'])"; done
