#!/bin/bash
# Build libicamd.so (gfx950 only) in-tree. Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include"
OBJS=""
PIDS=""
COMPILED=""
REUSED=""
for f in conv_igemm conv3x3_halo conv1x1_resident conv_stem conv_wgrad norm_pool loss_optim token_ops attention dwconv gemm_nt image_ops collective capi; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ icamd_internal.h -nt build/$f.o ] || [ ../../include/icamd.h -nt build/$f.o ]; then
    mkdir -p build
    rm -f build/$f.o   # a failed compile must not leave a stale object for the link step
    $HIPCC $FLAGS "$@" -c $f.hip -o build/$f.o &
    PIDS="$PIDS $!"
    COMPILED="$COMPILED $f"
  else
    REUSED="$REUSED $f"
  fi
  OBJS="$OBJS build/$f.o"
done
for p in $PIDS; do wait $p; done   # `set -e`: the first failed compile aborts the build
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o libicamd.so
echo "compiled for gfx950:${COMPILED:- (none)}; reused up-to-date objects:${REUSED:- (none)}"
echo "built $(pwd)/libicamd.so"
