#!/bin/bash
# GPU box: single-stream rocprofv3 kernel stats of the ResNet-50 bench -> gpurun_out/r5_stats_1s.csv (+ two-stream)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RAW=/tmp/icamd_r5_stats; rm -rf $RAW; mkdir -p $RAW
export ICAMD_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/s1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r5_stats_1s.log 2>&1
unset ICAMD_WGRAD_STREAM
cp $(find $RAW/s1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r5_stats_1s.csv
