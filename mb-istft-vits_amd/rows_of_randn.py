"""Rows lo:hi of `torch.randn(b_all, *tail, device=cuda)` without drawing the other rows.

A sharded run wants every rank to use ITS rows of the draw a single process would make for the whole batch
(dist.sharded_infer: ranks seeded alike then reproduce the single-process result), and to leave the device generator
where that full draw would leave it.  Drawing everything and slicing costs B_global x I x T' floats per rank and step
(222 MB at 8 x 64 utterances).  torch's CUDA/HIP normal kernel is a grid-stride loop over a fixed grid: thread `idx`
(Philox subsequence idx) writes, in its k-th call of the 4-wide normal generator, the elements idx + S (4 k + ii),
ii = 0..3, with S = 256 * grid threads and grid = min(SMs * (max threads per SM / 256), ceil(numel / 256)).  So a run of
elements [a, b) of a large draw lies inside the calls k_lo .. k_hi of every thread, and those calls are exactly what a
draw of (k_hi - k_lo + 1) * 4 S elements produces when the generator's Philox offset is advanced by 4 k_lo first.

That is knowledge of torch internals, so it is checked, not trusted: the first use per (device, torch build) compares the
fast path with the full draw once; on any mismatch the fast path is switched off for the process and the plain
slice of the full draw is used (same results, old cost)."""
import torch

_ok = {}          # device index -> True (verified) / False (mismatch: use the full draw)


def _grid_threads(dev, numel):
    p = torch.cuda.get_device_properties(dev)
    blocks = min(p.multi_processor_count * (p.max_threads_per_multi_processor // 256), (numel + 255) // 256)
    return 256 * blocks


def _full(lo, hi, b_all, tail, dev):
    return torch.randn(b_all, *tail, device=dev, dtype=torch.float32)[lo:hi]


def _fast(lo, hi, b_all, tail, dev):
    row = 1
    for d in tail:
        row *= int(d)
    n_all = b_all * row
    S = _grid_threads(dev, n_all)
    if S < 256 * 2048 // 8 or (hi - lo) * 2 > b_all or n_all < 8 * S:      # small draws: nothing to save
        return None
    a, b = lo * row, hi * row
    k_lo, k_hi = (a // S) // 4, ((b - 1) // S) // 4
    m = (k_hi - k_lo + 1) * 4 * S
    if _grid_threads(dev, m) != S:                    # the partial draw must run on the same grid
        return None
    gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
    off0 = gen.get_offset()
    gen.set_offset(off0 + 4 * k_lo)
    part = torch.randn(m, device=dev, dtype=torch.float32)
    gen.set_offset(off0 + ((n_all - 1) // (4 * S) + 1) * 4)      # where the full draw would have left the generator
    return part[a - k_lo * 4 * S: b - k_lo * 4 * S].view(hi - lo, *tail)


def randn_rows(lo, hi, b_all, tail, dev):
    """== torch.randn(b_all, *tail, device=dev)[lo:hi], generator state included; cheaper when lo:hi is a small part."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    state = _ok.get(key)
    if state is False:
        return _full(lo, hi, b_all, tail, dev)
    gen = torch.cuda.default_generators[key]
    if state is None:                                # verify once: both ways from the same generator state
        st = gen.get_state()
        want = _full(lo, hi, b_all, tail, dev).clone()
        end = gen.get_state()
        gen.set_state(st)
        got = _fast(lo, hi, b_all, tail, dev)
        if got is None:
            gen.set_state(end)
            return want                               # too small to bother (not a verdict on the fast path)
        same = bool(torch.equal(got, want)) and bool(torch.equal(gen.get_state(), end))
        _ok[key] = same
        gen.set_state(end)
        return want
    got = _fast(lo, hi, b_all, tail, dev)
    return got if got is not None else _full(lo, hi, b_all, tail, dev)
