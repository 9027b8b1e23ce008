#!/usr/bin/env python3
"""Audit of compiled kernels for SERIALISED vector-memory loads: a load followed (within a few instructions, no other
load in between) by `s_waitcnt vmcnt(0)` is one exposed memory latency; a kernel with dozens of them in a row has a
loop that hipcc could not batch (typically `cond ? ptr[i] : 0` or `if (cond) v += ptr[i]` inside an unrolled loop).
usage: asm_serial_loads.py file.hip [kernel-name-regex]     (device-only compile to assembly, like kernel_regs.sh)"""
import re, subprocess, sys, os
src = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
out = "/tmp/asm/%s.s" % os.path.basename(src).replace(".hip", "")
os.makedirs("/tmp/asm", exist_ok=True)
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
    serial = 0
    last_load = None
    for i, l in enumerate(ins):
        if re.match(r"(global_load|buffer_load|flat_load)", l) and "lds" not in l:
            if last_load is not None and i - last_load <= 2:
                last_load = None          # part of a batch of loads
                batch = True
            last_load = i
        elif l.startswith("s_waitcnt") and "vmcnt(0)" in l:
            if last_load is not None and i - last_load <= 6 and not any(
                    re.match(r"(global_load|buffer_load)", x) for x in ins[max(0, last_load - 2):last_load]):
                serial += 1
            last_load = None
    nload = sum(1 for l in ins if re.match(r"(global_load|buffer_load|flat_load)", l))
    print("%-110s loads %4d  single load -> vmcnt(0): %3d" % (k[:110], nload, serial))
