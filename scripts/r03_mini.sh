#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/${1:-r03s}; mkdir -p $O; cd $R
: > $O/mini_ab.txt
for cb in "ljs_mini_mb_istft_vits 64" "ljs_mini_istft_vits 64" "ljs_mini_mb_istft_vits 16"; do
  set -- $cb
  for v in 0 1; do MBV_CONV_HALF=$v timeout -k 10 200 python scripts/stage_ab.py $1 $2 2>&1 | grep total | tee -a $O/mini_ab.txt; done
done
