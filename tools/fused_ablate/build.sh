#!/bin/bash
# Build the ablation variants of the fused conv3 + bn3 backward kernel (build container; run them on the GPU box).
cd "$(dirname "$0")"
mkdir -p variants
for a in ${@:-0 1 8 2 4 6 14 15}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../imageclassification_amd/csrc -DICAMD_FUSED_ABLATE=$a -Wno-unused-value \
    -x hip probe.cpp -o variants/probe_$a 2> variants/build_$a.log &
done
wait
ls -la variants/
