#!/bin/bash
# the routing switches of earlier rounds against the current defaults, ResNet-50 step (two rounds)
for rep in 1 2; do
for v in "X=0" "ICAMD_FUSED_BNBWD=1" "ICAMD_FUSED_POOL_BWD=1" "ICAMD_DUAL_BNBWD=0" "ICAMD_BNRED=0" "ICAMD_GEMM_K1024=0" "ICAMD_PW_XCD=0" "ICAMD_WGRAD_STREAM=0" "ICAMD_IGEMM_LEAN=0" "ICAMD_WGRAD_HALO=0"; do
  echo -n "$v  "
  env $v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 20 steps: //'
done; done
