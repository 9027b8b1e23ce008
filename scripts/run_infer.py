#!/usr/bin/env python3
"""N x infer of one configuration (workload for rocprofv3 runs). usage: run_infer.py [config] [batch] [n] [t_text]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import make_net
from mb_istft_vits_amd import synth
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "ljs_mb_istft_vits"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
T_text = int(sys.argv[4]) if len(sys.argv) > 4 else 200
net, sd = make_net(cfg_name)
x, xl, sid = synth.synthetic_batch(net.cfg, B, T_text, seed=0 if T_text == 200 else 1)
xg, xlg = torch.from_numpy(x).cuda(), torch.from_numpy(xl).cuda()
sg = torch.from_numpy(sid).cuda() if sid is not None else None
for i in range(n):
    r = net.infer(xg, xlg, sg, noise_scale=0, length_scale=1)
torch.cuda.synchronize()
print("done", r[0].shape)
