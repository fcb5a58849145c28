#!/bin/bash
for rep in 1 2; do for v in 0 1; do
  echo "ICAMD_PW_S2=$v"
  ICAMD_PW_S2=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^.bench.*timed"
  ICAMD_PW_S2=$v python3 bench.py --mode eval --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^.bench.*timed"
done; done
for v in 0 1; do echo "ICAMD_PW_S2=$v"; ICAMD_PW_S2=$v python3 tools/bench_layers.py 256 10 2>&1 | grep "k1 s2"; done
