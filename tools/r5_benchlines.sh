#!/bin/bash
# GPU box: the four bench lines of the round into gpurun_out/r5_bench/ (copy to profiles/r05_bench_*.json)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_bench; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_r50.json 2> $O/bench_r50.err
python3 bench.py --arch vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_vit.json 2> $O/bench_vit.err
python3 bench.py --arch convnext_tiny --mixup --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cnx.json 2> $O/bench_cnx.err
python3 bench.py --mode eval --steps 20 --warmup 5 > $O/bench_eval.json 2> $O/bench_eval.err
grep -h "timed" $O/*.err
