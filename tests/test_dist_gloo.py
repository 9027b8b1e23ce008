"""world_size-2 `gloo` tests (CPU) of the sharded path's host logic (`mb_istft_vits_amd.dist`):
shard bounds, checkpoint broadcast, global-T' all-reduce and the waveform all-gather.
The per-shard compute is the ORACLE here (the HIP path needs a GPU); what is checked is the
property SURVEY §8e demands: shards padded to the global T'max reproduce the full-batch result
exactly, while shards padded to their local T'max would not."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import config_for
from mb_istft_vits_amd import dist as mdist, spec as mspec, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleNet:
    """Stands in for SynthesizerTrn._run on CPU: same signature/return layout, oracle compute."""

    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg

    def _device(self):
        return torch.device("cpu")

    def _run(self, x, xl, sid, noise_scale, length_scale, max_len, decode, frames_hook=None,
             stat_reduce=None, outputs=None, prior_rows=None):
        from oracle import ref_infer as R
        first = R.infer(self.sd, self.cfg, x, xl, sid, noise_scale=0.0, length_scale=length_scale)
        stat = torch.stack((first["y_lengths"].max(), torch.zeros((), dtype=torch.int64)))
        if stat_reduce is not None:
            stat_reduce(stat)
        tp = int(stat[0])
        if frames_hook is not None:
            tp = frames_hook(tp)
        r = R.infer(self.sd, self.cfg, x, xl, sid, noise_scale=0.0, length_scale=length_scale, t_frames=tp)
        return (r["o"], r["o_mb"], r["spec"], r["phase"], r["attn"], r["y_mask"],
                (r["z"], r["z_p"], r["m_p"], r["logs_p"]), {}, r["y_lengths"])


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        _, cfg = config_for("ljs_mini_mb_istft_vits")
        shapes = mspec.param_shapes(cfg)
        sd0 = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 1234).items()} if rank == 0 else None
        sd = mdist.broadcast_state_dict(sd0, shapes, torch.device("cpu"))
        ref_sd = synth.make_state_dict(cfg, 1234)
        assert all(torch.equal(sd[k], torch.from_numpy(ref_sd[k])) for k in shapes)

        x, xl, _ = synth.synthetic_batch(cfg, 5, 14, seed=9, ragged=True)       # uneven shards: 3 + 2
        x, xl = torch.from_numpy(x), torch.from_numpy(xl)
        net = OracleNet(sd, cfg)
        o, ylen = mdist.sharded_infer(net, x, xl, None, noise_scale=0, length_scale=1)
        torch.save({"o": o, "ylen": ylen}, os.path.join(tmp, "out%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_shard_bounds():
    assert [mdist.shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [mdist.shard_bounds(64, 8, r)[1] - mdist.shard_bounds(64, 8, r)[0] for r in range(8)] == [8] * 8


@pytest.mark.timeout(300)
def test_sharded_path_equals_full_batch(tmp_path):
    from oracle import ref_infer as R
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(tmp_path, "out%d.pt" % r)) for r in range(2)]
    assert torch.equal(outs[0]["o"], outs[1]["o"]) and torch.equal(outs[0]["ylen"], outs[1]["ylen"])
    _, cfg = config_for("ljs_mini_mb_istft_vits")
    sd = synth.make_state_dict(cfg, 1234)
    x, xl, _ = synth.synthetic_batch(cfg, 5, 14, seed=9, ragged=True)
    torch.set_num_threads(4)
    full = R.infer(sd, cfg, x, xl)
    assert torch.equal(outs[0]["ylen"], full["y_lengths"])
    assert outs[0]["o"].shape == full["o"].shape
    # identical up to conv reduction-order noise between batch sizes (oneDNN picks kernels per shape)
    assert float((outs[0]["o"] - full["o"]).abs().max()) < 1e-5
    # and the global pad is what makes it so: a shard run at its LOCAL T'max differs in its tail
    lo, hi = mdist.shard_bounds(5, 2, 1)
    local = R.infer(sd, cfg, x[lo:hi], xl[lo:hi])
    tp_local = local["o"].shape[-1]
    if tp_local < full["o"].shape[-1]:
        tail = (local["o"] - full["o"][lo:hi, :, :tp_local]).abs().max()
        assert float(tail) > 1e-4
