#!/bin/bash
# (CFGS="cfg batch;cfg batch" overrides the configuration list)
# usage: r03_cfg_ab.sh <tag> "<ENV=a>" "<ENV=b>" ...   other configurations' ms per step / conv-stack fraction per variant (same box, alternating)
set -o pipefail
VARS=("${@:2}")
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
: > $O/cfg_ab.txt
IFS=";" read -ra CF <<< "${CFGS:-uudb_ms_istft_vits_ms 32;ljs_mini_mb_istft_vits 64;ljs_mb_istft_vits 16;ljs_mb_istft_vits 48;ljs_mini_istft_vits 64}"
for cb in "${CF[@]}"; do
  set -- $cb; cfg=$1; b=$2
  for rep in 1 2; do
    for v in "${VARS[@]}"; do
      env $v timeout -k 10 300 python bench.py --config $cfg --batch $b --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg B=$b [$v]: %.3f ms/step  conv %.4f' % (d['ms_per_step'], d['roofline_conv']['frac']))" | tee -a $O/cfg_ab.txt
    done
  done
done
