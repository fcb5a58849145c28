#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_p15.log
: > $O
run() { echo "## $*" >> $O; env "$@" 2>&1 | grep -E "^.bench.*timed|passed|failed|Error|error" >> $O; }
echo "## model tests" >> $O
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_zz_loss_curve_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -4 >> $O
for i in 1 2; do
run ICAMD_FUSED_APPLY_CONV=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ICAMD_FUSED_APPLY_CONV=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ICAMD_FUSED_APPLY_CONV=1 ICAMD_FUSED_NT=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ICAMD_FUSED_APPLY_CONV=1 ICAMD_FUSED_NT=1 ICAMD_PW_NT=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ICAMD_FUSED_APPLY_CONV=0 ICAMD_FUSED_NT=1 ICAMD_PW_NT=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
done
