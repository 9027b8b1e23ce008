#!/bin/bash
# A/B of the conv kernel variants under rocprofv3 --kernel-trace (run on the GPU box)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "0 0" "0 1" "1 0" "1 1"; do
  set -- $cfg
  export MBV_CONV_PIPE=$1 MBV_CONV_WIDE=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_p$1_w$2 -- python3 $R/scripts/prof_kernels.py conv 3 > $R/gpurun_out/ab_p$1_w$2.log 2>&1
  echo "== PIPE=$1 WIDE=$2"; grep conv1d $R/gpurun_out/ab_p$1_w$2/*/*kernel_stats.csv | cut -d, -f1-4 | sed 's/.*mbv:://'
done
