#!/usr/bin/env python3
"""Stage times of one configuration (A/B of run-time switches through the environment).
usage: python scripts/stage_ab.py [config] [batch] [--ragged]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gpu_util import make_net
from mb_istft_vits_amd import synth
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "ljs_mb_istft_vits"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ragged = "--ragged" in sys.argv
net, sd = make_net(cfg_name)
x, xl, sid = synth.synthetic_batch(net.cfg, B, 200, seed=0, ragged=ragged)
xg, xlg = torch.from_numpy(x).cuda(), torch.from_numpy(xl).cuda()
sg = torch.from_numpy(sid).cuda() if sid is not None else None
acc = {}
n = 12
for i in range(n + 3):
    r = net.infer(xg, xlg, sg, noise_scale=0, length_scale=1)
    t = dict(r[7])
    if i >= 3:
        for k, v in t.items():
            acc.setdefault(k, []).append(v * 1e3)
tags = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("MBV_"))
print("%s B=%d%s [%s]: " % (cfg_name, B, " ragged" if ragged else "", tags) +
      "  ".join("%s %.3f" % (k, float(np.median(v))) for k, v in acc.items()) +
      "  | total %.3f ms" % sum(float(np.median(v)) for v in acc.values()))
