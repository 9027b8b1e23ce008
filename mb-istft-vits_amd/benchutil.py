"""Measurement helper shared by bench.py and scripts/prof_kernels.py (not on the product path)."""
import ctypes as C

import torch

from . import _capi


def istft_waveform_only_ms(net, B, Tp, iters=50, x_post=None, prescaled=True, warm=1000, rotate=1):
    """Average duration (ms) of the fused iSTFT+PQMF launch in waveform-only mode on a
    [B, 72, 16 Tp + 1] input, HIP events on the launch stream.  prescaled=True times the variant
    the decoder stack uses (x_post in the library's internal units, see mbistft_vits.h).
    rotate > 1: consecutive launches walk `rotate` distinct (input, output) buffer sets, so that with
    rotate * working set >> 256 MiB no launch finds its data in the Infinity Cache."""
    dev = net._device()
    h = net._ensure_handle()
    L = _capi.lib()
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    if x_post is None:
        x_posts = [torch.randn(B, 72, 16 * Tp + 1, device=dev, generator=g) * 0.5 for _ in range(rotate)]
    else:
        x_posts = [x_post] + [x_post.clone() for _ in range(rotate - 1)]
    outs = [torch.empty(B, 1, 256 * Tp, device=dev) for _ in range(rotate)]
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    ptrs = [(C.c_void_p(xp.data_ptr()), C.c_void_p(o.data_ptr())) for xp, o in zip(x_posts, outs)]
    state = {"i": 0}

    def launch():
        xp, op = ptrs[state["i"] % rotate]
        state["i"] += 1
        rc = L.mbv_istft_pqmf(h, xp, B, Tp, None, 2 if prescaled else 0, op, None, None, None, sp)
        _capi.check(h, rc, "mbv_istft_pqmf")
    # untimed warm-up long enough (~35 ms of launches) for the memory / fabric clocks to come up
    # from idle: measured right after an idle period the same launch is 12-15 % slower
    # (scripts/istft_state.py)
    for _ in range(warm):
        launch()
    # several event-bracketed batches; the median batch is reported (single batches on this pool
    # scatter by +-5 % with clock / cache state)
    times = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            launch()
        e1.record(stream)
        e1.synchronize()
        times.append(e0.elapsed_time(e1) / iters)
    times.sort()
    return times[len(times) // 2]
