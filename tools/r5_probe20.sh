#!/bin/bash
cd $GRAFT_REPO_ROOT
L=imageclassification_amd/csrc/libicamd.so
V=imageclassification_amd/csrc/build/variants
cp $L /tmp/base.so
O=gpurun_out/r5_p20.log
: > $O
for rep in 1 2; do
  for v in base attention; do
    if [ $v = base ]; then cp /tmp/base.so $L; else cp $V/libicamd_$v.so $L; fi
    echo "variant $v (vit)" >> $O
    python3 bench.py --arch vit_base_patch16_224 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" >> $O
  done
  for v in base conv_stem conv3x3_halo; do
    if [ $v = base ]; then cp /tmp/base.so $L; else cp $V/libicamd_$v.so $L; fi
    echo "variant $v (resnet)" >> $O
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^.bench.*timed" >> $O
  done
done
cp /tmp/base.so $L
