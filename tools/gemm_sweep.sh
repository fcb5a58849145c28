#!/bin/bash
# GPU box: gemm_nt.hip at the ViT-B/16 Linear shapes, both tile widths, next to the vendor GEMM (comparison only)
for tn in 8 128 256; do
  for s in "50432 3072 768" "50432 768 3072" "50432 2304 768" "50432 768 768"; do
    echo -n "ICAMD_GEMM_TN=$tn "; ICAMD_GEMM_TN=$tn python tools/gemm_probe.py $s
  done
done
python tools/vendor_gemm_probe.py
