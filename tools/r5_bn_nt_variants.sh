#!/bin/bash
# Build container: libicamd variants with non-temporal loads / stores in the BatchNorm streaming kernels (A/B runs on the GPU box:
# copy one over imageclassification_amd/csrc/libicamd.so inside the gpurun command, never in the tree).
cd "$(dirname "$0")/../imageclassification_amd/csrc"
mkdir -p build/variants
OBJS=$(ls build/*.o | grep -v norm_pool)
for v in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DICAMD_BN_NT=$v -c norm_pool.hip -o build/variants/norm_pool_$v.o 2>/dev/null &
done
wait
for v in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/variants/norm_pool_$v.o -ldl -o build/variants/libicamd_bnnt$v.so
done
ls -la build/variants/*.so
