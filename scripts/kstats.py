#!/usr/bin/env python3
"""Per-kernel averages from rocprofv3 --kernel-trace --stats output directories, side by side.
usage: kstats.py dirA [dirB ...]   (each holds a *kernel_stats.csv somewhere below it)"""
import csv, glob, sys
tabs = []
for d in sys.argv[1:]:
    f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
    tabs.append({r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))})
names = sorted(tabs[0], key=lambda n: -tabs[0][n][2])
for n in names[:28]:
    short = n.replace("void mbv::", "").replace("(anonymous namespace)::", "").split("(")[0][:46]
    print("%-46s" % short + "".join("  %4d x %9.1f us = %7.2f ms" % t.get(n, (0, 0, 0)) for t in tabs))
print("%-46s" % "total" + "".join("  %30.2f ms" % sum(v[2] for v in t.values()) for t in tabs))
