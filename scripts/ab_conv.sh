#!/bin/bash
# A/B of the conv kernel variants under rocprofv3 --kernel-trace (run on the GPU box)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in ${AB_CFGS:-"0,0 0,1 1,0 1,1"}; do
  export MBV_CONV_PIPE=${cfg%,*} MBV_CONV_WIDE=${cfg#*,}
  d=$R/gpurun_out/ab_p${MBV_CONV_PIPE}_w${MBV_CONV_WIDE}
  rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/scripts/prof_kernels.py conv 3 > $d.log 2>&1
  echo "== PIPE=$MBV_CONV_PIPE WIDE=$MBV_CONV_WIDE"
  python3 - <<PY
import csv,glob
f=glob.glob("$d/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "conv1d" in r["Name"]:
        print("  %-44s calls=%s avg=%.1f us" % (r["Name"].split("(")[0][-42:], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
