#!/usr/bin/env python3
"""One configuration of scripts/latency_b1.py (for rocprofv3): LAT_CFG, LAT_B, LAT_T env."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import make_net
from mb_istft_vits_amd import synth
cfg_name = os.environ.get("LAT_CFG", "ljs_mb_istft_vits")
B, T = int(os.environ.get("LAT_B", "1")), int(os.environ.get("LAT_T", "100"))
net, _ = make_net(cfg_name)
x, xl, _ = synth.synthetic_batch(net.cfg, B, T, seed=1)
x, xl = torch.from_numpy(x).cuda(), torch.from_numpy(xl).cuda()
for _ in range(5):
    o = net.infer(x, xl, noise_scale=0)[0]
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 20
for _ in range(n):
    out = net.infer(x, xl, noise_scale=0)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("%s B=%d T'=%d: %.3f ms/call stages(ms) %s" % (cfg_name, B, out[0].shape[-1] // 256, dt * 1e3,
      {k: round(v * 1e3, 2) for k, v in dict(out[7]).items()}))
