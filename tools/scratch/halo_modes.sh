for mode in 0 2 3; do
  echo "== ICAMD_CONV3X3_HALO=$mode"
  ICAMD_CONV3X3_HALO=$mode timeout -k 5 200 python tools/bench_layers.py 256 10 2>&1 | grep "k3 s1"
done
