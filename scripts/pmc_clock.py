#!/usr/bin/env python3
"""Per kernel from one rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY pass:
duration, shader clock (GUI_ACTIVE is summed over the 8 XCDs), MFMA-busy share of the launch, share of wave life spent waiting.
usage: pmc_clock.py <dir with *counter_collection.csv>"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[0]
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    d = rows[r["Dispatch_Id"]]
    d["name"] = r["Kernel_Name"].replace("void mbv::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    if "Start_Timestamp" in r: d["dur"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in rows.values():
    for k, v in d.items():
        if k != "name": agg[d["name"]][k].append(v)
print("%-44s %5s %9s %8s %8s %8s" % ("kernel", "n", "us", "GHz", "mfma%", "wait%"))
for n, a in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("dur", [0]))):
    m = lambda k: sum(a[k]) / len(a[k]) if k in a else float("nan")
    cyc = m("GRBM_GUI_ACTIVE") / 8
    print("%-44s %5d %9.1f %8.2f %8.1f %8.1f" % (n, len(a["GRBM_GUI_ACTIVE"]), m("dur"), cyc / (m("dur") * 1e3) if m("dur") == m("dur") else float("nan"),
          100 * m("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 8 * 256 * 4) * 8 if cyc else 0, 100 * m("SQ_WAIT_INST_ANY") / m("SQ_WAVE_CYCLES") if m("SQ_WAVE_CYCLES") else 0))
