#!/usr/bin/env python3
"""Numerics of a split-bf16 ("bf16x3") dot product against sequential fp32 accumulation — evidence for
the DESIGN.md note on a possible split-bf16 conv kernel (NOT used by the product path, which is exact
fp32).  a = a_hi + a_mid + a_lo with bf16 (round-to-nearest-even) pieces; products of bf16 pairs are
exact in fp32; accumulation in fp32 as the MFMA does.  Dot length 1408 = 128 channels x 11 taps."""
import numpy as np


def bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def split3(x):
    h = bf16(x)
    r1 = (x - h).astype(np.float32)
    m = bf16(r1)
    return h, m, bf16((r1 - m).astype(np.float32))


def dot32(x, y):
    acc = np.zeros(x.shape[0], np.float32)
    for k in range(x.shape[1]):
        acc = (acc + (x[:, k] * y[:, k]).astype(np.float32)).astype(np.float32)
    return acc


rs = np.random.RandomState(0)
K, N = 1408, 4000
a = rs.standard_normal((N, K)).astype(np.float32) / np.sqrt(K)
b = (rs.standard_normal((N, K)) * np.where(rs.rand(N, K) < 0.5, 1.0, 0.1)).astype(np.float32)
exact = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
(ah, am, al), (bh, bm, bl) = split3(a), split3(b)
six = sum((dot32(x, y) for x, y in ((al, bh), (am, bm), (ah, bl), (am, bh), (ah, bm), (ah, bh))),
          np.zeros(N, np.float32))
three = sum((dot32(x, y) for x, y in ((am, bh), (ah, bm), (ah, bh))), np.zeros(N, np.float32))
scale = np.sqrt((exact ** 2).mean())
for name, v in (("fp32, sequential accumulation", dot32(a, b)), ("split bf16, 6 terms", six),
                ("split bf16, 3 terms", three), ("plain bf16", dot32(ah, bh))):
    print("%-32s relative rms error %.2e" % (name, np.sqrt(((v - exact) ** 2).mean()) / scale))
