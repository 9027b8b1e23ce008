#!/usr/bin/env python3
"""Prints the PQMF_C / PQMF_G constant tables embedded in csrc/istft_pqmf.hip
(float64 design of pqmf.py:15-75, rounded to fp32)."""
import numpy as np


def kaiser(n, beta):
    a = (n - 1) / 2.0
    x = np.arange(n, dtype=np.float64)
    return np.i0(beta * np.sqrt(np.clip(1.0 - ((x - a) / a) ** 2, 0.0, None))) / np.i0(beta)


n = np.arange(63.0)
centre = n - 31
with np.errstate(all="ignore"):
    proto = np.sin(np.pi * 0.15 * centre) / (np.pi * centre)
proto[31] = 0.15
proto *= kaiser(63, 9.0)
c = np.array([[np.cos((2 * k + 1) * (np.pi / 8) * (q - 30.5) - (-1) ** k * np.pi / 4) for q in range(8)]
              for k in range(4)])
g = np.concatenate([8 * proto * np.array([(-1) ** (j // 8) for j in range(63)], float), [0.0]])
for name, arr in (("PQMF_C[32]", c.reshape(-1)), ("PQMF_G[64]", g)):
    print("__device__ constexpr float %s = {" % name)
    vals = ["%.9ef" % np.float32(v) for v in arr]
    for i in range(0, len(vals), 4):
        print("    " + ", ".join(vals[i:i + 4]) + ",")
    print("};")
