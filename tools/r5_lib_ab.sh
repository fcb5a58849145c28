#!/bin/bash
# same-box A/B of builds of the library (tools/attn_ab/lib_*.so): args = variant names
L=imageclassification_amd/csrc/libicamd.so
cp $L /tmp/lib_keep.so
for rep in 1 2 3; do for v in "$@"; do
  cp tools/attn_ab/lib_$v.so $L
  echo -n "$v  r50 "; python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['roofline_classes']
print(d['ms_per_step'], 'bn_apply', c['bn_apply']['ms_per_step'], 'bn_bwd', c['bn_bwd']['ms_per_step'], 'finalize', c['bn_finalize']['ms_per_step'])"
done; done
cp /tmp/lib_keep.so $L
