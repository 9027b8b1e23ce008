#!/bin/bash
# usage: r03_ks.sh <tag> "<ENV=..>" ...  -> per-kernel stats side by side (same box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0; dirs=""
for v in "$@"; do
  i=$((i+1))
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$i -o ks -- python3 $R/scripts/run_infer.py ljs_mb_istft_vits 64 4 > $O/ks$i.log 2>&1 || { tail -5 $O/ks$i.log; exit 1; }
  rm -f $O/ks$i/*kernel_trace.csv $O/ks$i/*/*kernel_trace.csv
  dirs="$dirs $O/ks$i"
done
cd $R; python scripts/kstats.py $dirs | tee $O/kstats.txt
