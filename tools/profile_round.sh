#!/bin/bash
# Round evidence, part A / B (one gpurun call each: a call is capped at 20 minutes).  Run on the GPU box from the repo root:
#   ROUND=r05 bash tools/profile_round.sh A     bench lines of the three model families + rocprofv3 kernel stats
#   ROUND=r05 bash tools/profile_round.sh C     the three bench lines again, once part B's summaries are in profiles/
#   ROUND=r05 bash tools/profile_round.sh B     PMC passes (HBM traffic per arch, MFMA busy), per-layer / GEMM / depthwise tables
# then, here: python tools/summarize_profiles.py gpurun_out/prof_r04 r04
PART=${1:-A}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${ROUND:-r05}
mkdir -p $O
VIT="--arch vit_base_patch16_224"
CNX="--arch convnext_tiny --mixup"
if [ "$PART" = "C" ]; then
  # bench lines only (after part B's PMC summaries have been copied into profiles/: roofline.traffic is then non-null)
  python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_r50.json 2> $O/bench_r50.err
  python3 $R/bench.py $VIT --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_vit.json 2> $O/bench_vit.err
  python3 $R/bench.py $CNX --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cnx.json 2> $O/bench_cnx.err
  python3 $R/bench.py --mode eval --steps 20 --warmup 5 > $O/bench_eval.json 2> $O/bench_eval.err
elif [ "$PART" = "A" ]; then
  echo "== bench lines"; date
  python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_r50.json 2> $O/bench_r50.err
  python3 $R/bench.py $VIT --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_vit.json 2> $O/bench_vit.err
  python3 $R/bench.py $CNX --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cnx.json 2> $O/bench_cnx.err
  python3 $R/bench.py --mode eval --steps 20 --warmup 5 > $O/bench_eval.json 2> $O/bench_eval.err
  echo "== kernel stats"; date
  RAWA=/tmp/icamd_prof_raw_a; rm -rf $RAWA; mkdir -p $RAWA
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAWA/stats_r50 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats_r50.log 2>&1
  export ICAMD_WGRAD_STREAM=0
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAWA/stats_r50_1s -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats_r50_1s.log 2>&1
  unset ICAMD_WGRAD_STREAM
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAWA/stats_vit -- python3 $R/bench.py $VIT --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_vit.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAWA/stats_cnx -- python3 $R/bench.py $CNX --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_cnx.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAWA/stats_eval -- python3 $R/bench.py --mode eval --steps 10 --warmup 2 --no-cpu-baseline > $O/stats_eval.log 2>&1
  for d in stats_r50 stats_r50_1s stats_vit stats_cnx stats_eval; do
    mkdir -p $O/$d
    cp $(find $RAWA/$d -name "*kernel_stats.csv" | head -1) $O/$d/kernel_stats.csv
  done
  date
else
  # raw rocprofv3 output (databases, per-dispatch CSVs: > 64 MiB in all) stays in /tmp on the GPU box; only the summaries and the
  # counter CSVs the MFMA-busy tables are built from go to gpurun_out/ (gpurun merges at most 64 MiB back)
  RAW=/tmp/icamd_prof_raw
  rm -rf $RAW; mkdir -p $RAW
  echo "== pmc"; date
  export ICAMD_WGRAD_STREAM=0
  for a in r50 vit cnx eval; do
    case $a in r50) ARGS="";; vit) ARGS="$VIT";; cnx) ARGS="$CNX";; eval) ARGS="--mode eval";; esac
    rocprofv3 --pmc FETCH_SIZE -d $RAW/pmc_fetch_$a -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_$a.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $RAW/pmc_write_$a -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write_$a.log 2>&1
    date
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $RAW/pmc_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_mfma.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $RAW/pmc_mfma_vit -- python3 $R/bench.py $VIT --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_mfma_vit.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $RAW/pmc_mfma_cnx -- python3 $R/bench.py $CNX --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_mfma_cnx.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $RAW/pmc_mfma_eval -- python3 $R/bench.py --mode eval --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_mfma_eval.log 2>&1
  unset ICAMD_WGRAD_STREAM
  for d in pmc_mfma pmc_mfma_vit pmc_mfma_cnx pmc_mfma_eval; do
    mkdir -p $O/$d
    cp $(find $RAW/$d -name "*counter_collection.csv" | head -1) $O/$d/counter_collection.csv
  done
  echo "== tables"; date
  cd $R
  python3 tools/bench_layers.py 256 10 > $O/layers.txt 2>&1
  bash tools/gemm_ablate.sh > $O/gemm_shapes.txt 2>&1
  python3 tools/vendor_gemm_probe.py >> $O/gemm_shapes.txt 2>&1
  python3 tools/bench_dwconv.py > $O/dwconv.txt 2>&1
  python3 tools/bench_cnx_layers.py 256 10 2>&1 | grep -v amdgpu.ids > $O/cnx_layers.txt
  python3 tools/bench_attn.py 256 20 2>&1 | grep -v amdgpu.ids > $O/attention.txt
  # round 5: the fused conv3 + bn3 backward against the three launches it replaces; the Linear-layer weight gradients (8-phase vs ring)
  python3 tools/fused_bwd_probe.py 20 2>&1 | grep -v amdgpu.ids > $O/fused_bwd.txt
  python3 tools/fused_fwd_probe.py 20 2>&1 | grep -v amdgpu.ids > $O/fused_fwd.txt
  python3 tools/ls_fuse_probe.py 20 2>&1 | grep -v amdgpu.ids > $O/ls_fuse.txt
  for e in 1 0; do for s in "50432 768 3072" "50432 3072 768" "50432 768 2304" "50432 768 768" "12544 768 3072" "50176 1024 512"; do
    echo -n "ICAMD_WGRAD_8PHASE=$e "; ICAMD_WGRAD_8PHASE=$e python3 tools/wgrad_probe.py $s 2>&1 | grep TFLOP; done; done > $O/wgrad_shapes.txt
  # 10 steps per PMC run: 1 warm-up + 3 timed + the 3-step host-enqueue burst + 3 in the per-class timing pass
  python3 tools/pmc_traffic.py $(find $RAW/pmc_fetch_r50 -name "*.db" | head -1) $(find $RAW/pmc_write_r50 -name "*.db" | head -1) 10 $O/pmc_traffic_r50.json resnet50 256
  python3 tools/pmc_traffic.py $(find $RAW/pmc_fetch_vit -name "*.db" | head -1) $(find $RAW/pmc_write_vit -name "*.db" | head -1) 10 $O/pmc_traffic_vit.json vit_base_patch16_224 256
  python3 tools/pmc_traffic.py $(find $RAW/pmc_fetch_cnx -name "*.db" | head -1) $(find $RAW/pmc_write_cnx -name "*.db" | head -1) 10 $O/pmc_traffic_cnx.json convnext_tiny 256
  python3 tools/pmc_traffic.py $(find $RAW/pmc_fetch_eval -name "*.db" | head -1) $(find $RAW/pmc_write_eval -name "*.db" | head -1) 10 $O/pmc_traffic_eval.json resnet50 384 eval
  du -sh $O
  date
fi
