#!/bin/bash
# usage: r03_ab2.sh <tag> "<ENV=a ENV2=b>" "<...>" ...   (each quoted group is one variant; same box)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
: > $O/stage_ab.txt
for rep in 1 2; do
  for v in "$@"; do env $v timeout -k 10 200 python scripts/stage_ab.py 2>&1 | grep total | tee -a $O/stage_ab.txt; done
done
