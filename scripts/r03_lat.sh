#!/bin/bash
# usage: r03_lat.sh <tag> "<ENV=a>" ...   single-utterance / batch-8 latency per variant (same box), default and low-latency mode
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
: > $O/latency_ab.txt
for v in "$@"; do
  for sk in 0 1; do
    echo "== $v MBV_CONV_SPLITK=$sk" | tee -a $O/latency_ab.txt
    env $v MBV_CONV_SPLITK=$sk timeout -k 10 200 python scripts/latency_b1.py 2>&1 | grep "ms/call" | tee -a $O/latency_ab.txt
  done
done
