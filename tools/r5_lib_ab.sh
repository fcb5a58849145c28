#!/bin/bash
# Same-box A/B of builds of the library: args = variant names.  Prepare (build container) tools/attn_ab/lib_<name>.so, e.g.
#   bash imageclassification_amd/csrc/build.sh && cp imageclassification_amd/csrc/libicamd.so tools/attn_ab/lib_new.so
#   git stash; bash imageclassification_amd/csrc/build.sh; cp imageclassification_amd/csrc/libicamd.so tools/attn_ab/lib_old.so; git stash pop
# (tools/attn_ab/ is git-ignored but travels to the GPU box), then: gpurun -- 'bash tools/r5_lib_ab.sh old new'.
# The body below is the last comparison of the round (depthwise weight gradient); edit the bench lines for another one.
L=imageclassification_amd/csrc/libicamd.so
cp $L /tmp/lib_keep.so
for v in "$@"; do cp tools/attn_ab/lib_$v.so $L; echo "$v"; python3 tools/bench_dwconv.py 2>&1 | grep -v amdgpu.ids; done
for rep in 1 2 3; do for v in "$@"; do
  cp tools/attn_ab/lib_$v.so $L
  echo -n "$v  cnx "; python3 bench.py --arch convnext_tiny --mixup --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 10 steps: //'
done; done
cp /tmp/lib_keep.so $L
