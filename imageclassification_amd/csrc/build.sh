#!/bin/bash
# Build libicamd.so (gfx950 only) in-tree. Usage: build.sh
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include"
UNITS="conv_igemm conv3x3_halo conv1x1_resident conv_fused_bwd conv_fused_fwd conv_stem conv_wgrad norm_pool loss_optim token_ops attention dwconv gemm_nt image_ops collective capi"
asm_of() { echo "build/$1-hip-amdgcn-amd-amdhsa-gfx950.s"; }
OBJS=""
ASMS=""
PIDS=""
COMPILED=""
REUSED=""
for f in $UNITS; do
  # An object is reused only together with the device assembly it was built with: the ISA lint below reads that file, and an
  # object whose .s is missing (a build directory of an older tree, a deleted file) would otherwise be linked unlinted.
  if [ ! -f build/$f.o ] || [ ! -f "$(asm_of $f)" ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ icamd_internal.h -nt build/$f.o ] || [ ../../include/icamd.h -nt build/$f.o ]; then
    mkdir -p build
    rm -f build/$f.o "$(asm_of $f)"   # a failed compile must not leave a stale object (or its assembly) for the link step
    # -save-temps=obj leaves build/$f-hip-amdgcn-amd-amdhsa-gfx950.s next to the object: the device assembly tools/isa_lint.py reads
    $HIPCC $FLAGS -save-temps=obj -c $f.hip -o build/$f.o 2> build/$f.log &
    PIDS="$PIDS $!"
    COMPILED="$COMPILED $f"
  else
    REUSED="$REUSED $f"
  fi
  OBJS="$OBJS build/$f.o"
  ASMS="$ASMS $(asm_of $f)"
done
FAILED=0
for p in $PIDS; do wait $p || FAILED=1; done
for f in $COMPILED; do grep -v "argument unused during compilation" build/$f.log >&2 || true; done
if [ $FAILED = 1 ]; then echo "build.sh: a translation unit failed to compile" >&2; exit 1; fi
# keep only the objects and the device assembly of -save-temps (the rest is ~80 MB that would travel to the GPU box)
rm -f build/*.bc build/*.hipi build/*.out build/*.resolution.txt build/*.hipfb build/*-host-x86_64-unknown-linux-gnu.s build/*-gfx950.o
# ISA lint over the device assembly of EVERY translation unit that is about to be linked (rule PK32-OPSEL: a hardware hazard of
# this part, see tools/isa_lint.py): the explicit list, one file per object -- a missing file fails the build.  A hit removes the
# library and the offending objects (with their assembly), so that no later run can link them without recompiling.
for a in $ASMS; do
  if [ ! -f "$a" ]; then echo "build.sh: $a is missing; libicamd.so NOT linked" >&2; rm -f libicamd.so; exit 1; fi
done
if ! python3 ../../tools/isa_lint.py $ASMS > build/isa_lint.log; then
  cat build/isa_lint.log >&2
  echo "build.sh: ISA lint failed; libicamd.so NOT linked" >&2
  rm -f libicamd.so
  for a in $(grep -o '^build/[^:]*\.s' build/isa_lint.log | sort -u); do
    f=$(basename "$a" -hip-amdgcn-amd-amdhsa-gfx950.s)
    rm -f "build/$f.o" "$a"
  done
  exit 1
fi
cat build/isa_lint.log
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o libicamd.so
echo "compiled for gfx950:${COMPILED:- (none)}; reused up-to-date objects:${REUSED:- (none)}"
echo "built $(pwd)/libicamd.so"
