#!/bin/bash
for rep in 1 2; do for v in 0 1 2 3; do
  echo "ICAMD_LN_NV3=$v"
  ICAMD_LN_NV3=$v python3 bench.py --arch convnext_tiny --mixup --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^.bench.*timed"
  ICAMD_LN_NV3=$v python3 bench.py --arch vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^.bench.*timed"
done; done
