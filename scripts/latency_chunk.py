#!/usr/bin/env python3
"""Latency of the chunked decoder entry `net.dec(z_chunk)` (streaming use: inferz_test.ipynb cell 7)
for a few chunk lengths, default vs split-K ("splitk" option)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import make_net
net, _ = make_net("ljs_mb_istft_vits")
for frames in (16, 32, 64, 128):
    z = torch.randn(1, 192, frames, device="cuda")
    row = []
    for sk in (0, 1):
        net.set_option("splitk", sk)
        for _ in range(5):
            net.dec(z)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 30
        for _ in range(n):
            o = net.dec(z)[0]
            torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / n * 1e3)
    print("dec chunk of %3d frames (%.2f s audio): %.2f ms default, %.2f ms splitk" %
          (frames, frames * 256 / 22050, row[0], row[1]))
