#!/usr/bin/env python3
"""Single-utterance latency of infer() (the reference's service use-case, tts_vits.py): wall time
per call incl. the host sync, T_text = 100, ljs_mb / mini configs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import make_net
from mb_istft_vits_amd import synth
for cfg_name in ("ljs_mini_mb_istft_vits", "ljs_mb_istft_vits"):
    net, _ = make_net(cfg_name)
    for B in (1, 8):
        x, xl, _ = synth.synthetic_batch(net.cfg, B, 100, seed=1)
        x, xl = torch.from_numpy(x).cuda(), torch.from_numpy(xl).cuda()
        for _ in range(5):
            o = net.infer(x, xl, noise_scale=0)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 30
        for _ in range(n):
            o = net.infer(x, xl, noise_scale=0)[0]
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("%-24s B=%d T'=%d: %.3f ms/call  audio %.2f s  RTF %.5f" % (cfg_name, B, o.shape[-1] // 256, dt * 1e3,
              B * o.shape[-1] / 22050, dt / (B * o.shape[-1] / 22050)))
