#!/usr/bin/env python3
"""Audit of compiled kernels for MFMA regions broken up by scalar branches: a run-time test on an index inside an
unrolled MFMA nest (`if (i != skip) acc[i] = mfma(..)`) makes hipcc emit a branch around every MFMA.
Prints, per kernel, the MFMA count and the number of s_cbranch instructions that sit between two MFMAs less than
12 instructions apart.  usage: asm_branchy_mfma.py file.hip [kernel-name-regex]"""
import re, subprocess, sys, os
src = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
out = "/tmp/asm/%s.s" % os.path.basename(src).replace(".hip", "")
os.makedirs("/tmp/asm", exist_ok=True)
if not (os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(src)):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S", src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
name, lines, kernels = None, [], {}
for ln in open(out):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name, lines = m.group(1), []
        kernels[name] = lines
    elif name:
        lines.append(ln)
        if "s_endpgm" in ln:
            name = None
for k, ls in kernels.items():
    if not pat.search(k):
        continue
    ins = [l.strip() for l in ls if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
    br = 0
    for a, b in zip(mf, mf[1:]):
        if b - a <= 12:
            br += sum(1 for l in ins[a:b] if l.startswith("s_cbranch"))
    if mf:
        print("%-110s mfma %4d  branches between adjacent MFMAs: %3d" % (k[:110], len(mf), br))
