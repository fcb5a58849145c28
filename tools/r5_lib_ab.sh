#!/bin/bash
# same-box A/B of builds of the library (tools/attn_ab/lib_*.so): args = variant names
L=imageclassification_amd/csrc/libicamd.so
cp $L /tmp/lib_keep.so
for rep in 1 2; do for v in "$@"; do
  cp tools/attn_ab/lib_$v.so $L
  echo -n "$v  vit "; python3 bench.py --arch vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 10 steps: //'
  echo -n "$v  cnx "; python3 bench.py --arch convnext_tiny --mixup --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 10 steps: //'
  echo -n "$v  r50 "; python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 20 steps: //'
  echo -n "$v  eval "; python3 bench.py --mode eval --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 20 steps: //'
done; done
cp /tmp/lib_keep.so $L
