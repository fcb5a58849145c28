#!/bin/bash
# A/B of the folded layer scale on ConvNeXt-T + mixup + EMA (two pairs) and on convnext eval-free paths
for rep in 1 2; do for v in 0 1; do
  echo "ICAMD_FUSED_LAYERSCALE=$v"
  ICAMD_FUSED_LAYERSCALE=$v python3 bench.py --arch convnext_tiny --mixup --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^.bench.*timed"
done; done
