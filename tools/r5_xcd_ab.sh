#!/bin/bash
for rep in 1 2 3; do for v in 1 0; do
  echo -n "ICAMD_PW_XCD=$v  r50 "; ICAMD_PW_XCD=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 20 steps: //'
  echo -n "ICAMD_PW_XCD=$v  eval "; ICAMD_PW_XCD=$v python3 bench.py --mode eval --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 20 steps: //'
done; done
for rep in 1 2; do for v in 1 0; do
  echo -n "ICAMD_PW_XCD=$v  cnx "; ICAMD_PW_XCD=$v python3 bench.py --arch convnext_tiny --mixup --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" | sed 's/.*timed 10 steps: //'
done; done
