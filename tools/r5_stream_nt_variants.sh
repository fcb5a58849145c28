#!/bin/bash
# Build container: libicamd variants with non-temporal streaming loads in token_ops / dwconv / loss_optim (A/B runs on the GPU box).
cd "$(dirname "$0")/../imageclassification_amd/csrc"
mkdir -p build/variants
for f in token_ops dwconv loss_optim; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DICAMD_STREAM_NT=1 -c $f.hip -o build/variants/${f}_nt.o 2>/dev/null &
done
wait
link() { # name, replaced units...
  name=$1; shift
  OBJS=$(ls build/*.o)
  for f in "$@"; do OBJS=$(echo "$OBJS" | grep -v "build/$f.o"); OBJS="$OBJS build/variants/${f}_nt.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o build/variants/libicamd_$name.so
}
link tok token_ops
link dw dwconv
link opt loss_optim
link all token_ops dwconv loss_optim
ls -la build/variants/libicamd_*.so | awk '{print $5, $9}'
