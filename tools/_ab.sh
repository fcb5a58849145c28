set -e
R=$GRAFT_REPO_ROOT
cd $R
for v in 0 1 0 1; do ICAMD_PW_RESIDENT=$v timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('pw_resident=$v', d['ms_per_step'], {k:v.get('ms_per_step') for k,v in d.get('kernels',{}).items() if k.startswith('conv')})"; done
