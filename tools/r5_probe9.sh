#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_p9.log
: > $O
run() { echo "## $*" >> $O; env "$@" >> $O 2>&1; }
echo "## wgrad tests (default + forced ring/8-phase child)" >> $O
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "test_conv_wgrad" >> $O 2>&1 || exit 1
for e in 0 1; do
  for s in "50432 768 3072" "50432 3072 768" "50432 768 2304" "50432 768 768" "200704 512 256" "50176 1024 512"; do
    run ICAMD_WGRAD_8PHASE=$e python3 tools/wgrad_probe.py $s
  done
done
run python3 bench.py --arch vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline
run ICAMD_WGRAD_8PHASE=0 python3 bench.py --arch vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline
grep -v "amdgpu.ids" $O > $O.tmp; mv $O.tmp $O
