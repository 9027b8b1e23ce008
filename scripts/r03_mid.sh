#!/bin/bash
# mid-size launches: batch split between the two conv tile shapes on/off (same box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/${1:-r03h}; mkdir -p $O; cd $R
: > $O/mid_ab.txt
for cb in "uudb_ms_istft_vits_ms 32" "ljs_mb_istft_vits 16" "ljs_mb_istft_vits 32" "ljs_mb_istft_vits 48" "ljs_mb_istft_vits 96" "ljs_ms_istft_vits 32"; do
  set -- $cb
  for v in 0 1; do MBV_CONV_BATCH_SPLIT=$v timeout -k 10 200 python scripts/stage_ab.py $1 $2 2>&1 | grep total | tee -a $O/mid_ab.txt; done
done
