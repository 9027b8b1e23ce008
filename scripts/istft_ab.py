#!/usr/bin/env python3
"""A/B of the fused iSTFT+PQMF launch inside `infer` (all outputs written) — the launch the product
issues by default.  usage: [MBV_ISTFT_NT=0|1] python scripts/istft_ab.py [config] [batch] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gpu_util import make_net
from mb_istft_vits_amd import synth
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "ljs_mb_istft_vits"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
net, sd = make_net(cfg_name)
x, xl, sid = synth.synthetic_batch(net.cfg, B, 200, seed=0)
xg, xlg = torch.from_numpy(x).cuda(), torch.from_numpy(xl).cuda()
sg = torch.from_numpy(sid).cuda() if sid is not None else None
for mode, outs in (("all outputs", None), ("waveform only", ("o",))):
    ts, cs = [], []
    for i in range(steps + 5):
        r = net.infer(xg, xlg, sg, noise_scale=0, length_scale=1, outputs=outs)
        c, t = net.kernel_times_ms()
        if i >= 5:
            ts.append(t); cs.append(c)
    Tp = r[0].shape[-1] // 256
    per_frame = 11264 if outs is None else 5632
    ts = np.array(ts)
    print("%s NT=%s %s B=%d T'=%d: istft launch mean %.2f us median %.2f us min %.2f -> %.0f GB/s (median); conv stack %.3f ms"
          % (mode, os.environ.get("MBV_ISTFT_NT", "default"), cfg_name, B, Tp, ts.mean() * 1e3, np.median(ts) * 1e3,
             ts.min() * 1e3, per_frame * B * Tp / (np.median(ts) * 1e-3) / 1e9, np.mean(cs)))
