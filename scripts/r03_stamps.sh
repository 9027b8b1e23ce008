#!/bin/bash
# In-kernel timeline of the decoder's conv kernel (scripts/conv_stamps.hip, built with -DMBV_CONV_STAMPS) on the five conv
# geometries of the bench batch.  usage (inside gpurun): bash scripts/r03_stamps.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/${1:-r03}; mkdir -p $O; cd $R
for args in "128 7 9056 64 1 0" "128 7 9056 64 1 1" "128 3 9056 64 1 0" "128 11 9056 64 5 0" "256 7 2264 64 1 0"; do
  echo "=== $args"
  timeout -k 10 120 scripts/conv_stamps $args || exit 1
done 2>&1 | tee $O/stamps.txt
