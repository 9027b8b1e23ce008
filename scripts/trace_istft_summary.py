#!/usr/bin/env python3
"""Per-dispatch view of the fused iSTFT+PQMF launch in a rocprofv3 --kernel-trace CSV of `bench.py`:
the dispatches come in the order [per-step all-outputs launches][1000 untimed warm-up launches]
[7 x 50 timed waveform-only launches]; prints the averages of each group.
usage: trace_istft_summary.py <..._kernel_trace.csv>"""
import csv
import sys

import numpy as np

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "istft_pqmf_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
big = d > 1.6 * np.median(d)
w = d[~big]
print("istft_pqmf_kernel dispatches: %d, average %.2f us" % (len(d), d.mean()))
print("  all-outputs variant (inside infer): n=%d  avg %.1f us" % (big.sum(), d[big].mean() if big.any() else 0))
print("  waveform-only: n=%d" % len(w))
print("    first 100 (clocks coming up from idle)  avg %.2f us" % w[:100].mean())
print("    rest of the untimed warm-up              avg %.2f us" % w[100:-350].mean())
print("    last 350 = the launches bench.py times   avg %.2f us  median %.2f us" % (w[-350:].mean(), np.median(w[-350:])))
