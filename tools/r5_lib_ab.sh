#!/bin/bash
# same-box A/B of builds of the library (tools/attn_ab/lib_*.so): args = variant names
L=imageclassification_amd/csrc/libicamd.so
cp $L /tmp/lib_keep.so
for v in "$@"; do cp tools/attn_ab/lib_$v.so $L; echo "$v"; python3 tools/bench_dwconv.py 2>&1 | grep -v amdgpu.ids; done
for rep in 1 2 3; do for v in "$@"; do
  cp tools/attn_ab/lib_$v.so $L
  echo -n "$v  cnx "; python3 bench.py --arch convnext_tiny --mixup --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 10 steps: //'
done; done
cp /tmp/lib_keep.so $L
