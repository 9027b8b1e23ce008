#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch per kernel."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "mbv" not in k: continue
            print(k)
            for c, v in cs.items():
                print("   %-34s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
