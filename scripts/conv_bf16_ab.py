#!/usr/bin/env python3
"""Exact fp32 vs the opt-in split-bf16 conv mode (`conv_bf16 = 3`) on a few configurations: ms per infer (sum of the
stage timers), decoder stage, and the distance between the two waveforms.  usage: python scripts/conv_bf16_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gpu_util import make_net
from mb_istft_vits_amd import synth
def rms(a): return float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
for cfg_name, B in (("ljs_mb_istft_vits", 64), ("ljs_mb_istft_vits", 16), ("ljs_mb_istft_vits", 4), ("uudb_ms_istft_vits_ms", 32), ("ljs_istft_vits", 32)):
    net, sd = make_net(cfg_name)
    x, xl, sid = synth.synthetic_batch(net.cfg, B, 200, seed=0)
    xg, xlg = torch.from_numpy(x).cuda(), torch.from_numpy(xl).cuda()
    sg = torch.from_numpy(sid).cuda() if sid is not None else None
    outs = {}
    for mode in (0, 3):
        net.set_option("conv_bf16", mode)
        for i in range(3):
            r = net.infer(xg, xlg, sg, noise_scale=0, length_scale=1)
        torch.cuda.synchronize()
        ts = []
        for i in range(6):
            r = net.infer(xg, xlg, sg, noise_scale=0, length_scale=1)
            ts.append(sum(dict(r[7]).values()) * 1e3)
        outs[mode] = (r[0].cpu().numpy(), float(np.median(ts)), dict(r[7]))
    o0, o3 = outs[0][0], outs[3][0]
    print("%s B=%d: fp32 %.2f ms  split-bf16 %.2f ms (decoder %.2f -> %.2f)  | waveform rms %.4f  rms diff %.3e  max diff %.3e" % (
        cfg_name, B, outs[0][1], outs[3][1], outs[0][2]["waveform_decoder"] * 1e3, outs[3][2]["waveform_decoder"] * 1e3, rms(o0), rms(o3 - o0), float(np.abs(o3 - o0).max())))
