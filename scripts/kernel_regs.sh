#!/bin/bash
# Register / spill / LDS usage of every kernel in one .hip file (device-only compile to assembly, metadata parsed).
# usage: scripts/kernel_regs.sh mb-istft-vits_amd/csrc/conv1d.hip [grep-pattern]
set -e
f=${1:?file.hip}; pat=${2:-.}
out=/tmp/asm/$(basename $f .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only -S $f -o $out 2>/dev/null
python3 - "$out" "$pat" <<'P'
import re, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
md = txt[txt.rfind('amdhsa.kernels:'):]
for blk in md.split('- .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    name = g('name')
    if not pat.search(name): continue
    print('%-95s vgpr %s agpr %s sgpr %s vspill %s sspill %s scratch %s lds %s' % (
        name[:95], g('vgpr_count'), blk.split()[0], g('sgpr_count'), g('vgpr_spill_count'),
        g('sgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
P
