#!/bin/bash
# same-box A/B of two builds of the library (tools/attn_ab/lib_{old,new}.so): attention alone and the ViT-B/16 step
L=imageclassification_amd/csrc/libicamd.so
cp $L /tmp/lib_keep.so
for rep in 1 2 3; do for v in old new; do
  cp tools/attn_ab/lib_$v.so $L
  echo -n "$v  "; python3 tools/bench_attn.py 256 20 2>&1 | grep -v amdgpu.ids | cut -c1-110
  echo -n "$v  vit "; python3 bench.py --arch vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 10 steps: //'
done; done
cp /tmp/lib_keep.so $L
