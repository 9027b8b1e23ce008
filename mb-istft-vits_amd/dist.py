"""Utterance-batch sharding across the GPUs of one node.

Utterances are independent, so the path shards by contiguous blocks of the
batch, one process per GPU.  Collectives (torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in the CPU tests):
  * `broadcast_state_dict` — rank 0's checkpoint to every rank, once;
  * one 2-element all-reduce(MAX) of [T', error flag] between phase A and phase B
    (on the device tensor, before its single host read), so every shard pads to
    the GLOBAL T'max: the decoder is unmasked, and an
    utterance's last ~15 frames depend on the padded length of its batch
    (SURVEY §7 "batch-padding dependence") — with the global pad the gathered
    result is identical to a single-GPU run of the whole batch;
  * `all_gather` of the fixed-stride waveform rows [B/N, 256 T'max] and of
    y_lengths.
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


def reduce_frames_and_status(stat):
    """In-place all-reduce(MAX) of the device tensor [T'max, error flag] between phase A and phase B:
    every shard pads to the global T'max, and a shard whose kernels flagged a bad token / length /
    speaker id makes EVERY rank raise (no rank is left waiting in the all-gather).  The reduce runs
    on the device tensor before its one host read, so the sharded path keeps the single host sync
    per call of the unsharded one."""
    dist.all_reduce(stat, op=dist.ReduceOp.MAX)


def gather_waveforms(o_local, ylen_local, shard_sizes):
    """all_gather of equal-stride rows; shards of unequal batch are padded to the largest."""
    world = dist.get_world_size()
    bmax = max(shard_sizes)
    dev = o_local.device
    rows = o_local.reshape(o_local.shape[0], -1)
    pad = bmax - rows.shape[0]
    if pad:
        rows = torch.cat([rows, rows.new_zeros(pad, rows.shape[1])])
        ylen_local = torch.cat([ylen_local, ylen_local.new_zeros(pad)])
    o_all = torch.empty(world * bmax, rows.shape[1], dtype=rows.dtype, device=dev)
    y_all = torch.empty(world * bmax, dtype=ylen_local.dtype, device=dev)
    dist.all_gather_into_tensor(o_all, rows.contiguous())
    dist.all_gather_into_tensor(y_all, ylen_local.contiguous())
    if all(n == bmax for n in shard_sizes):          # equal shards: the gathered buffer is the result
        return o_all.unsqueeze(1), y_all
    keep = torch.cat([torch.arange(r * bmax, r * bmax + n, device=dev) for r, n in enumerate(shard_sizes)])
    return o_all[keep].unsqueeze(1), y_all[keep]


def sharded_infer(net, x, x_lengths, sid=None, noise_scale=1, length_scale=1, max_len=None,
                  noise_scale_w=1., outputs=("o",)):
    """Every rank passes the SAME full batch; each synthesises its contiguous block and all ranks
    return the full-batch waveform [B, 1, 256 T'max] and y_lengths [B].

    By default only the waveform is materialised on each shard (`outputs=("o",)`: this entry
    returns nothing else); `outputs=None` makes every shard write all eight tensors of the
    reference tuple, as `infer` does (bench.py times that, so that N = 1 and N > 1 do equal work).
    Equality with a single-process run of the whole batch: bitwise at noise_scale == 0 (every shard
    pads to the global T'max).  With noise_scale != 0 each rank draws the prior noise for the WHOLE
    batch on its device generator and uses its rows, and with a StochasticDurationPredictor each
    rank draws the full-batch duration noise on the CPU generator (models.py:94) and uses its
    block — so ranks that are seeded alike (torch.manual_seed) reproduce the single-process draws;
    ranks seeded differently produce valid but different samples."""
    world, rank = dist.get_world_size(), dist.get_rank()
    B = x.shape[0]
    if B < world:                                    # same test on every rank, before any collective
        raise ValueError("batch %d smaller than world size %d" % (B, world))
    sizes = [shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world)]
    lo, hi = shard_bounds(B, world, rank)
    extra = {}
    if getattr(net.cfg, "use_sdp", False):
        extra = dict(noise_scale_w=noise_scale_w, noise_w=torch.randn(B, 2, x.shape[1])[lo:hi])
    r = net._run(x[lo:hi], x_lengths[lo:hi], sid[lo:hi] if sid is not None else None, noise_scale,
                 length_scale, max_len, True, stat_reduce=reduce_frames_and_status, outputs=outputs,
                 prior_rows=(lo, hi, B), **extra)
    o_local, ylen_local = r[0], r[8]
    return gather_waveforms(o_local, ylen_local, sizes)
