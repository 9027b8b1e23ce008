#!/bin/bash
# Build libmbistft_vits.so for gfx950 (in-tree, next to the sources).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
OBJS=""
for f in conv1d ops sdp attention istft_pqmf capi; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ kernels.h -nt $f.o ] || [ ../../include/mbistft_vits.h -nt $f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o $f.o &
  fi
  OBJS="$OBJS $f.o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmbistft_vits.so $OBJS
