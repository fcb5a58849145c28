#!/bin/bash
# GPU box: gemm_nt.hip at the ViT-B/16 Linear shapes (forward and data-gradient orientations)
for s in "50432 2304 768" "50432 3072 768" "50432 768 3072" "50432 768 768" "50432 768 2304"; do
  python tools/gemm_probe.py $s 2>&1 | grep TFLOP
done
