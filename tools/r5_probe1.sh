#!/bin/bash
# round-5 probe 1 (GPU box): weight-gradient block order / split floor, GEMM output-store policy, bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_p1.log
: > $O
run() { echo "## $*" >> $O; env "$@" >> $O 2>&1; }
for x in 0 1; do
  for s in "50432 768 3072" "50432 3072 768" "50432 768 2304" "50432 768 768"; do
    run ICAMD_WGRAD_XCD=$x python3 tools/wgrad_probe.py $s
  done
done
for m in 256 384 512 768; do
  run ICAMD_WGRAD_MINWG=$m python3 tools/wgrad_probe.py 50176 256 1024
  run ICAMD_WGRAD_MINWG=$m python3 tools/wgrad_probe.py 50176 1024 256
  run ICAMD_WGRAD_MINWG=$m python3 tools/wgrad_probe.py 12544 512 2048
  run ICAMD_WGRAD_MINWG=$m python3 tools/wgrad_probe.py 200704 128 512
done
for pol in 0 1 2; do
  for g in 4 0; do
    for s in "50432 2304 768" "50432 768 768" "50432 3072 768" "50432 768 3072"; do
      run ICAMD_GEMM_OUT_POLICY=$pol ICAMD_GEMM_GROUP_N=$g python3 tools/gemm_probe.py $s
    done
  done
done
echo "## tests" >> $O
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad or large_tile or resident" >> $O 2>&1
run python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
run python3 bench.py --arch vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline
run ICAMD_WGRAD_XCD=0 python3 bench.py --arch vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline
run python3 bench.py --arch convnext_tiny --mixup --steps 10 --warmup 3 --no-cpu-baseline
grep -v "amdgpu.ids" $O > $O.tmp; mv $O.tmp $O
