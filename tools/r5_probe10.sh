#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_p10.log
: > $O
run() { echo "## $*" >> $O; env "$@" >> $O 2>&1; }
for w in 2.5e10 1e10; do
  for s in "50176 256 1024" "50176 1024 256" "12544 512 2048" "12544 2048 512" "200704 512 256"; do
    run ICAMD_WGRAD_WORK=$w python3 tools/wgrad_probe.py $s
  done
done
run ICAMD_WGRAD_WORK=1e10 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
run python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
run python3 bench.py --arch convnext_tiny --mixup --steps 10 --warmup 3 --no-cpu-baseline
run ICAMD_WGRAD_8PHASE=0 python3 bench.py --arch convnext_tiny --mixup --steps 10 --warmup 3 --no-cpu-baseline
echo "## vit + fullsize tests" >> $O
timeout -k 10 900 python3 -m pytest tests/test_vit_gpu.py tests/test_fullsize_gpu.py tests/test_convnext_gpu.py -x -q >> $O 2>&1
grep -v "amdgpu.ids" $O > $O.tmp; mv $O.tmp $O
