cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/p1
for mode in 0 1; do
 for shape in "1024 256 1 1 14" "256 256 3 1 14" "256 128 1 1 56" "64 64 3 1 56"; do
  tag=$(echo $shape | tr ' ' '_')_m$mode
  ICAMD_WGRAD_RING=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p1/$tag -- python3 $R/tools/one_layer.py $shape 10 wgrad > /dev/null 2>&1
  f=$(find $R/gpurun_out/p1/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"; head -4 "$f" | cut -d, -f1-4 | cut -c1-150
 done
done
