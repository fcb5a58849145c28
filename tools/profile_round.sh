# Round-2 evidence: rocprofv3 kernel stats (two-stream and single-stream), PMC traffic / MFMA-busy passes, bench lines for
# the three model families.  Run on the GPU box from the repo root; writes under gpurun_out/prof_r02.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${ROUND:-r02}
mkdir -p $O
echo "== bench lines"; date
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_r50.json 2> $O/bench_r50.err
python3 $R/bench.py --arch vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_vit.json 2> $O/bench_vit.err
python3 $R/bench.py --arch convnext_tiny --mixup --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cnx.json 2> $O/bench_cnx.err
echo "== kernel stats"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_r50 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats_r50.log 2>&1
export ICAMD_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_r50_1s -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats_r50_1s.log 2>&1
unset ICAMD_WGRAD_STREAM
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_vit -- python3 $R/bench.py --arch vit_base_patch16_224 --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_vit.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cnx -- python3 $R/bench.py --arch convnext_tiny --mixup --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_cnx.log 2>&1
echo "== pmc"; date
export ICAMD_WGRAD_STREAM=0
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc TCC_BUSY_avr TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_tcc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_tcc.log 2>&1
unset ICAMD_WGRAD_STREAM
echo "== per-layer table and L2 requests of the 64->64 3x3 layer (implicit GEMM vs register-resident filter)"; date
python3 $R/tools/bench_layers.py 256 10 > $O/layers.txt 2>&1
for v in 0 1; do
  ICAMD_CONV3X3_RESIDENT=$v ICAMD_WGRAD_HALO=$v rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_l2_c64_$v -- python3 $R/tools/one_layer.py 64 64 3 1 56 3 fwd,dgrad,wgrad > $O/pmc_l2_c64_$v.log 2>&1
done
date
find $O -name "*.db" | head; find $O -name "*kernel_stats.csv" | head
cd $R && python3 tools/pmc_traffic.py $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) 7 $O/r02_pmc_traffic.json resnet50 256
