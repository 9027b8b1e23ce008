#!/usr/bin/env python3
"""Launch the two kernels that matter in isolation (for rocprofv3 --pmc / --kernel-trace):
  istft : fused iSTFT+PQMF, waveform-only, B=64 T'=566 (the bench workload) x N
  conv  : conv1d_mfma C=128 k=11 dil=5 (ResBlock1 conv at 16T'), B=64, T=9056 x N
usage: prof_kernels.py [istft|conv|both] [iters]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np      # noqa: E402
import torch            # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(os.environ.get("PROF_B", "64"))
Tp = int(os.environ.get("PROF_TP", "566"))

from gpu_util import make_net, op_conv1d     # noqa: E402
from mb_istft_vits_amd.benchutil import istft_waveform_only_ms   # noqa: E402

net = make_net("ljs_mini_mb_istft_vits")[0]
if which in ("istft", "both"):
    ms = istft_waveform_only_ms(net, B, Tp, iters=iters)
    print("istft_pqmf waveform-only B=%d T'=%d: %.4f ms  -> %.1f GB/s algorithmic" %
          (B, Tp, ms, 5632 * B * Tp / (ms * 1e-3) / 1e9))
if which in ("conv", "both"):
    rs = np.random.RandomState(0)
    for (C, K, dil, T) in ((128, 11, 5, 16 * Tp), (256, 7, 3, 4 * Tp), (128, 3, 1, 16 * Tp)):
        x = torch.randn(B, C, T, device="cuda")
        w = (rs.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
        b = rs.standard_normal(C).astype(np.float32)
        op_conv1d(net, x, w, b, K, dil, 0.1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        t0 = time.perf_counter()
        for _ in range(iters):
            op_conv1d(net, x, w, b, K, dil, 0.1)      # includes weight upload + sync per call
        torch.cuda.synchronize()
        print("conv C=%d K=%d dil=%d T=%d: host-timed %.3f ms/call (incl. upload) ; %.2f GFLOP" %
              (C, K, dil, T, (time.perf_counter() - t0) / iters * 1e3, 2.0 * B * T * C * C * K / 1e9))
