"""Utterance-batch sharding across the GPUs of one node.

Utterances are independent, so the path shards by contiguous blocks of the
batch, one process per GPU.  Collectives (torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in the CPU tests):
  * `broadcast_arena` — rank 0 loads the checkpoint, folds and packs it ONCE; the folded weight
    arena (one flat fp32 device buffer, ~130 MB for the mb config) is broadcast and the other
    ranks import it (`mbv_import_arena`): no state dict, no host-side fold, no upload on them.
    (`broadcast_state_dict` — the r01 form, rank 0's raw checkpoint to every rank — stays for
    callers that want module parameters on every rank.)
  * one 2-element all-reduce(MAX) of [T', error flag] between phase A and phase B
    (on the device tensor, before its single host read), so every shard pads to
    the GLOBAL T'max: the decoder is unmasked, and an
    utterance's last ~15 frames depend on the padded length of its batch
    (SURVEY §7 "batch-padding dependence") — with the global pad the gathered
    result is identical to a single-GPU run of the whole batch;
  * `all_gather` of the fixed-stride waveform rows [B/N, 256 T'max] and of
    y_lengths — on a side stream when asked (`overlap=`), so that it runs under compute.
"""
import torch
import torch.distributed as dist


def shard_bounds(batch, world, rank):
    """Contiguous block [lo, hi) of `batch` utterances for `rank` (sizes differ by <= 1)."""
    q, r = divmod(batch, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_state_dict(state_dict_or_none, keys_shapes, device, src=0):
    """rank `src` passes its state dict (name -> tensor); the others pass None and receive
    it.  One flat buffer, one broadcast (a 110 MB fp32 blob for the mb config)."""
    total = sum(int(torch.Size(s).numel()) for s in keys_shapes.values())
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        off = 0
        for k, shp in keys_shapes.items():
            n = int(torch.Size(shp).numel())
            flat[off:off + n] = torch.as_tensor(state_dict_or_none[k]).reshape(-1).to(device, torch.float32)
            off += n
    dist.broadcast(flat, src=src)
    out, off = {}, 0
    for k, shp in keys_shapes.items():
        n = int(torch.Size(shp).numel())
        out[k] = flat[off:off + n].reshape(tuple(shp)).clone()
        off += n
    return out


def broadcast_arena(net, src=0):
    """Rank `src` has loaded its checkpoint into `net` (on its GPU); every other rank passes a model of the
    same configuration that holds no checkpoint yet.  The folded, packed weight arena of `src` goes out in
    ONE broadcast and the receivers import it.  Returns the arena size in floats."""
    dev = next(net.parameters()).device
    size = torch.zeros(1, dtype=torch.int64, device=dev)
    flat = None
    if dist.get_rank() == src:
        flat = net.export_arena()
        size[0] = flat.numel()
    dist.broadcast(size, src=src)
    n = int(size.item())
    if flat is None:
        flat = torch.empty(n, dtype=torch.float32, device=dev)
    dist.broadcast(flat, src=src)
    if dist.get_rank() != src:
        net.import_arena(flat)
    return n


def reduce_frames_and_status(stat):
    """In-place all-reduce(MAX) of the device tensor [T'max, error flag] between phase A and phase B:
    every shard pads to the global T'max, and a shard whose kernels flagged a bad token / length /
    speaker id makes EVERY rank raise (no rank is left waiting in the all-gather).  The reduce runs
    on the device tensor before its one host read, so the sharded path keeps the single host sync
    per call of the unsharded one."""
    dist.all_reduce(stat, op=dist.ReduceOp.MAX)


class StepTimes:
    """HIP-event timeline of one `sharded_infer` call (pass `timing=StepTimes()`): events are recorded on the
    streams the work is issued on and read by `ms()` after the caller has synchronised.  Keys:
      all_reduce_ms      the [T', status] all-reduce between phase A and phase B
      gather_o_ms        all-gather(s) of the waveform rows (on the side stream when overlapped)
      gather_ylen_ms     all-gather of y_lengths
      compute_ms         everything else between the call's first and last event on the caller's stream
      exposed_gather_ms  how long the caller's stream waited for gathers at the end of the call
                         (== gather time without overlap; ~0 when the gather finished under compute)
      total_ms           first event -> last event"""

    def __init__(self):
        self._ev = {}
        self.world = None

    def mark(self, name, stream=None):
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream if stream is not None else torch.cuda.current_stream())
        self._ev.setdefault(name, []).append(e)

    def _span(self, a, b):
        return sum(x.elapsed_time(y) for x, y in zip(self._ev.get(a, []), self._ev.get(b, [])))

    def ms(self):
        ar = self._span("ar0", "ar1")
        go = self._span("go0", "go1")
        gy = self._span("gy0", "gy1")
        total = self._span("t0", "t1")
        exposed = self._span("wait0", "t1")
        overlapped = bool(self._ev.get("side"))
        on_main = ar + (0.0 if overlapped else go + gy)
        return {"all_reduce_ms": ar, "gather_o_ms": go, "gather_ylen_ms": gy,
                "compute_ms": max(0.0, total - on_main - (exposed if overlapped else 0.0)),
                "exposed_gather_ms": exposed if overlapped else go + gy, "total_ms": total,
                "overlap": overlapped, "world_size": self.world}


class Gathered:
    """Result of an overlapped `sharded_infer`: the gathers were issued on a side stream.  `result()` makes the
    CURRENT stream wait for them and returns (o, y_lengths); until then the tensors must not be read."""

    def __init__(self, o, ylen, event, keep):
        self._o, self._y, self._ev, self._keep = o, ylen, event, keep

    def result(self):
        if self._ev is not None:
            torch.cuda.current_stream().wait_event(self._ev)
            self._ev = None
        return self._o, self._y


_side_streams = {}


def _side_stream(dev):
    s = _side_streams.get(dev)
    if s is None:
        s = _side_streams[dev] = torch.cuda.Stream(device=dev)
    return s


def _pad_rows(rows, n):
    pad = n - rows.shape[0]
    return rows if pad == 0 else torch.cat([rows, rows.new_zeros((pad,) + tuple(rows.shape[1:]))])


def gather_waveforms(o_local, ylen_local, shard_sizes, timing=None):
    """all_gather of equal-stride rows; shards of unequal batch are padded to the largest."""
    world = dist.get_world_size()
    bmax = max(shard_sizes)
    dev = o_local.device
    rows = _pad_rows(o_local.reshape(o_local.shape[0], -1), bmax)
    ylen_local = _pad_rows(ylen_local, bmax)
    o_all = torch.empty(world * bmax, rows.shape[1], dtype=rows.dtype, device=dev)
    y_all = torch.empty(world * bmax, dtype=ylen_local.dtype, device=dev)
    if timing is not None:
        timing.mark("go0")
    dist.all_gather_into_tensor(o_all, rows.contiguous())
    if timing is not None:
        timing.mark("go1")
        timing.mark("gy0")
    dist.all_gather_into_tensor(y_all, ylen_local.contiguous())
    if timing is not None:
        timing.mark("gy1")
    if all(n == bmax for n in shard_sizes):          # equal shards: the gathered buffer is the result
        return o_all.unsqueeze(1), y_all
    keep = torch.cat([torch.arange(r * bmax, r * bmax + n, device=dev) for r, n in enumerate(shard_sizes)])
    return o_all[keep].unsqueeze(1), y_all[keep]


def sharded_infer(net, x, x_lengths, sid=None, noise_scale=1, length_scale=1, max_len=None,
                  noise_scale_w=1., outputs=("o",), overlap=None, timing=None):
    """Every rank passes the SAME full batch; each synthesises its contiguous block and all ranks
    return the full-batch waveform [B, 1, 256 T'max] and y_lengths [B].

    By default only the waveform is materialised on each shard (`outputs=("o",)`: this entry
    returns nothing else); `outputs=None` makes every shard write all eight tensors of the
    reference tuple, as `infer` does (bench.py times that, so that N = 1 and N > 1 do equal work).
    Equality with a single-process run of the whole batch: bitwise at noise_scale == 0 (every shard
    pads to the global T'max).  With noise_scale != 0 each rank draws the prior noise for the WHOLE
    batch on its device generator and uses its rows (also at noise_scale == 0, so that the generator
    advances as in a single-process run), and with a StochasticDurationPredictor each
    rank draws the full-batch duration noise on the CPU generator (models.py:94) and uses its
    block — so ranks that are seeded alike (torch.manual_seed) reproduce the single-process draws;
    ranks seeded differently produce valid but different samples.

    overlap (GPU, any backend):
      None     gathers on the caller's stream after the shard's last kernel; returns (o, y_lengths).
      "next"   gathers on a side stream, returns a `Gathered` handle at once: the caller goes on to its next
               call (the next step's encoder / flows / decoder run while RCCL moves this step's rows) and takes
               `.result()` when it needs the tensors.  Same kernels, same arguments: bitwise the same result.
      "halves" the shard's decoder runs in two halves and the first half's rows travel (side stream) under the
               second half's decode; returns (o, y_lengths).  Rows are computed exactly as in one piece (the
               decoder's arithmetic does not depend on the batch, tests assert it); worth it when half a shard
               still fills the chip (>= 128 utterances per GPU: at 64 the two half-batch decodes cost more than
               the gather they hide).
    timing: a `StepTimes` to fill (HIP events)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    B = x.shape[0]
    if B < world:                                    # same test on every rank, before any collective
        raise ValueError("batch %d smaller than world size %d" % (B, world))
    if overlap not in (None, "next", "halves"):
        raise ValueError("overlap must be None, 'next' or 'halves'")
    sizes = [shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world)]
    lo, hi = shard_bounds(B, world, rank)
    extra = {}
    if getattr(net.cfg, "use_sdp", False):
        extra = dict(noise_scale_w=noise_scale_w, noise_w=torch.randn(B, 2, x.shape[1])[lo:hi])
    if timing is not None:
        timing.world = world
        timing.mark("t0")

    def stat_reduce(stat):
        if timing is not None:
            timing.mark("ar0")
        reduce_frames_and_status(stat)
        if timing is not None:
            timing.mark("ar1")

    sid_l = sid[lo:hi] if sid is not None else None
    if overlap == "halves" and min(sizes) >= 2:
        return _sharded_halves(net, x[lo:hi], x_lengths[lo:hi], sid_l, noise_scale, length_scale, max_len, outputs,
                               sizes, (lo, hi, B), stat_reduce, extra, timing)
    r = net._run(x[lo:hi], x_lengths[lo:hi], sid_l, noise_scale, length_scale, max_len, True,
                 stat_reduce=stat_reduce, outputs=outputs, prior_rows=(lo, hi, B), **extra)
    o_local, ylen_local = r[0], r[8]
    if overlap == "next" and o_local.is_cuda:
        main = torch.cuda.current_stream()
        side = _side_stream(o_local.device)
        ready = torch.cuda.Event()
        ready.record(main)
        side.wait_event(ready)
        if timing is not None:
            timing.mark("side", side)
            timing.mark("wait0")
            timing.mark("t1")                        # the caller's stream is done with this call here
        with torch.cuda.stream(side):
            o_all, y_all = gather_waveforms(o_local, ylen_local, sizes, None if timing is None else _SideTiming(timing, side))
            done = torch.cuda.Event()
            done.record(side)
        for t in (o_local, ylen_local, o_all, y_all):
            t.record_stream(side)
        return Gathered(o_all, y_all, done, (o_local, ylen_local))
    out = gather_waveforms(o_local, ylen_local, sizes, timing)
    if timing is not None:
        timing.mark("t1")
    return out


class _SideTiming:
    """StepTimes marks recorded on the side stream."""

    def __init__(self, timing, stream):
        self._t, self._s = timing, stream

    def mark(self, name):
        self._t.mark(name, self._s)


def _sharded_halves(net, x, x_lengths, sid, noise_scale, length_scale, max_len, outputs, sizes, prior_rows,
                    stat_reduce, extra, timing):
    """overlap="halves": encoder + flows for the whole shard, then the decoder per half; the first half's rows are
    gathered on a side stream while the second half decodes."""
    world = dist.get_world_size()
    want = set(net._OUTPUT_NAMES) if outputs is None else set(outputs)
    dec_names = ("o", "o_mb", "spec", "phase")
    keep = tuple(n for n in net._OUTPUT_NAMES if n in want and n not in dec_names) + (() if "z" in want else ("z",))
    r = net._run(x, x_lengths, sid, noise_scale, length_scale, None, False, stat_reduce=stat_reduce,
                 outputs=keep, prior_rows=prior_rows, **extra)
    z, ylen_local = r[6][0], r[8]
    n, Tp = z.shape[0], z.shape[2]
    Td = Tp if max_len is None else max(0, min(Tp, int(max_len)))
    if Td <= 0:
        raise ValueError("max_len leaves no frames to decode")
    zd = z if Td == Tp else z[:, :, :Td].contiguous()
    dev = z.device
    g = net._speaker_embedding(sid) if (sid is not None and net.cfg.gin_channels) else None
    full = net._alloc_decoder_outputs(n, Td, dev, want | {"o"})
    bmax = max(sizes)
    hA = bmax // 2                                   # rows of the first half on EVERY rank (min(sizes) >= hA)
    stride = full[0].shape[-1]
    o_all = torch.empty(world, bmax, stride, dtype=torch.float32, device=dev)
    y_all = torch.empty(world * bmax, dtype=ylen_local.dtype, device=dev)
    main = torch.cuda.current_stream()
    side = _side_stream(dev)
    st = None if timing is None else _SideTiming(timing, side)
    if timing is not None:
        timing.mark("side", side)

    def gather_part(rows, n_rows, dst_lo):
        tmp = torch.empty(world, n_rows, stride, dtype=torch.float32, device=dev)
        if st is not None:
            st.mark("go0")
        dist.all_gather_into_tensor(tmp.view(world * n_rows, stride), _pad_rows(rows, n_rows).contiguous())
        o_all[:, dst_lo:dst_lo + n_rows].copy_(tmp)
        if st is not None:
            st.mark("go1")
        return tmp

    held = []
    for k, (a, b) in enumerate(((0, hA), (hA, n))):
        net._decode_into(zd[a:b], None if g is None else g[a:b], tuple(None if t is None else t[a:b] for t in full))
        ready = torch.cuda.Event()
        ready.record(main)
        side.wait_event(ready)
        with torch.cuda.stream(side):
            held.append(gather_part(full[0][a:b].reshape(b - a, stride), hA if k == 0 else bmax - hA, 0 if k == 0 else hA))
            if k == 1:
                if st is not None:
                    st.mark("gy0")
                dist.all_gather_into_tensor(y_all, _pad_rows(ylen_local, bmax).contiguous())
                if st is not None:
                    st.mark("gy1")
    done = torch.cuda.Event()
    done.record(side)
    if timing is not None:
        timing.mark("wait0")
    main.wait_event(done)
    for t in held + [o_all, y_all, full[0], ylen_local]:
        t.record_stream(side)
    if timing is not None:
        timing.mark("t1")
    o_all = o_all.view(world * bmax, stride)
    if all(s == bmax for s in sizes):
        return o_all.unsqueeze(1), y_all
    keep_idx = torch.cat([torch.arange(r_ * bmax, r_ * bmax + s, device=dev) for r_, s in enumerate(sizes)])
    return o_all[keep_idx].unsqueeze(1), y_all[keep_idx]
