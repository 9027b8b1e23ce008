#!/usr/bin/env python3
"""What plain streaming kernels reach on this box (torch fill_/copy_ as the yardstick): the
write-heavy all-outputs iSTFT launch (163 MB read from cache, 241 MB written) is compared with these."""
import torch
def t(fn, n=50, warm=20):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (41, 241, 482, 964):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    ms = t(lambda: a.fill_(1.0))
    print("fill  %4d MB: %.1f us -> %.0f GB/s written" % (mb, ms * 1e3, mb / ms))
    ms = t(lambda: b.copy_(a))
    print("copy  %4d MB: %.1f us -> %.0f GB/s read+written" % (mb, ms * 1e3, 2 * mb / ms))
    ms = t(lambda: torch.sum(a))
    print("sum   %4d MB: %.1f us -> %.0f GB/s read" % (mb, ms * 1e3, mb / ms))
