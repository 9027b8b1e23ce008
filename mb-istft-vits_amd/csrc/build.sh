#!/bin/bash
# Build libmbistft_vits.so for gfx950 (in-tree, next to the sources).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
OBJS=""
PIDS=""
for f in conv1d conv1d_narrow wn_fused ops sdp attention istft_pqmf capi; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ kernels.h -nt $f.o ] || [ ../../include/mbistft_vits.h -nt $f.o ]; then
    rm -f $f.o                     # a failed compile must not leave a stale object to link
    $HIPCC $FLAGS -c $f.hip -o $f.o &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $f.o"
done
for pid in $PIDS; do
  wait $pid || { echo "build.sh: a hipcc job failed" >&2; exit 1; }
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmbistft_vits.so $OBJS
