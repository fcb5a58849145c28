#!/bin/bash
# same-box A/B of builds of the library (tools/attn_ab/lib_*.so): args = variant names
L=imageclassification_amd/csrc/libicamd.so
cp $L /tmp/lib_keep.so
for rep in 1 2 3; do for v in "$@"; do
  cp tools/attn_ab/lib_$v.so $L
  echo -n "$v  r50 "; python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['roofline_classes']
print(d['ms_per_step'], 'dgrad', c['conv_dgrad']['ms_per_step'], 'fwd', c['conv_fwd']['ms_per_step'])"
  echo -n "$v  eval "; python3 bench.py --mode eval --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 20 steps: //'
done; done
cp /tmp/lib_keep.so $L
