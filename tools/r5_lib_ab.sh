#!/bin/bash
# Same-box A/B of builds of the library: args = variant names.  Prepare (build container) tools/attn_ab/lib_<name>.so, e.g.
#   bash imageclassification_amd/csrc/build.sh && cp imageclassification_amd/csrc/libicamd.so tools/attn_ab/lib_new.so
#   git stash; bash imageclassification_amd/csrc/build.sh; cp imageclassification_amd/csrc/libicamd.so tools/attn_ab/lib_old.so; git stash pop
# (tools/attn_ab/ is git-ignored but travels to the GPU box), then: gpurun -- 'bash tools/r5_lib_ab.sh old new'.
# The body below is the last comparison of the round (attention forward); edit the bench lines for another one.
L=imageclassification_amd/csrc/libicamd.so
cp $L /tmp/lib_keep.so
for rep in 1 2 3; do for v in "$@"; do
  cp tools/attn_ab/lib_$v.so $L
  echo -n "$v  "; python3 tools/bench_attn.py 256 20 2>&1 | grep -v amdgpu.ids | cut -c1-100
  echo -n "$v  vit "; python3 bench.py --arch vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['roofline_classes']
print(d['ms_per_step'], 'attn_bwd', c['attn_bwd']['ms_per_step'], 'attn_fwd', c['attn_fwd']['ms_per_step'])"
done; done
cp /tmp/lib_keep.so $L
