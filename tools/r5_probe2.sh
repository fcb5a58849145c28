#!/bin/bash
# round-5 probe 2 (GPU box): the fused conv3 + bn3 backward -- parity, isolated time, model tests, bench A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_p2.log
: > $O
run() { echo "## $*" >> $O; env "$@" >> $O 2>&1; }
echo "## fused test" >> $O
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "bn_bwd_fused" >> $O 2>&1 || exit 1
run timeout -k 10 300 python3 tools/fused_bwd_probe.py 20
echo "## model tests" >> $O
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_zz_loss_curve_gpu.py -x -q >> $O 2>&1
run python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ICAMD_FUSED_CONV_BN_BWD=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
run python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ICAMD_FUSED_CONV_BN_BWD=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
grep -v "amdgpu.ids" $O > $O.tmp; mv $O.tmp $O
