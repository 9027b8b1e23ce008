"""Child rank of tests/test_gpu_dist.py: one process per rank, all ranks on cuda:0 (a one-GPU box),
`gloo` backend (RCCL cannot put two ranks on one device; the collectives still run on CUDA tensors).
Runs the PRODUCT path — `mb_istft_vits_amd.dist.sharded_infer` over the HIP kernels — and saves what
every rank got.

usage: dist_gpu_worker.py RANK WORLD PORT OUTDIR
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from gpu_util import make_net
    from mb_istft_vits_amd import dist as mdist, spec as mspec, synth
    dev = torch.device("cuda", 0)
    res = {}
    try:
        for name, cfg_name, B, T in (("mini_b5", "ljs_mini_mb_istft_vits", 5, 23),
                                     ("uudb_b5", "uudb_ms_istft_vits_ms", 5, 17),
                                     ("mini_b4", "ljs_mini_mb_istft_vits", 4, 31)):
            if name == "uudb_b5":
                # weights: rank 0's raw checkpoint through the state-dict broadcast (the other ranks load what arrives)
                net, sd = make_net(cfg_name, seed=1300)
                shapes = mspec.param_shapes(net.cfg)
                got = mdist.broadcast_state_dict({k: torch.from_numpy(v) for k, v in sd.items()} if rank == 0 else None,
                                                 shapes, dev)
                net.load_state_dict(got)
            else:
                # weights: rank 0 loads + folds, the others import the folded arena (never see a state dict:
                # their module parameters keep another seed's values, which the kernels must not use)
                net, sd = make_net(cfg_name, seed=1300 if rank == 0 else 7)
                res[name + "_arena_floats"] = mdist.broadcast_arena(net, src=0)
            x, xl, sid = synth.synthetic_batch(net.cfg, B, T, seed=40 + B, ragged=True)
            xg, xlg = torch.from_numpy(x).to(dev), torch.from_numpy(xl).to(dev)
            sidg = torch.from_numpy(sid).to(dev) if sid is not None else None
            o, ylen = mdist.sharded_infer(net, xg, xlg, sidg, noise_scale=0, length_scale=1)
            res[name] = {"o": o.cpu(), "ylen": ylen.cpu()}
            if name == "mini_b5":
                # every shard writing all eight tensors (what bench.py times) returns the same waveform
                o_all, _ = mdist.sharded_infer(net, xg, xlg, sidg, noise_scale=0, length_scale=1, outputs=None)
                res["mini_b5_all_outputs_equal"] = bool(torch.equal(o_all, o))
                # noise_scale > 0: ranks seeded alike draw the full-batch prior noise and use their rows
                torch.manual_seed(77)
                torch.cuda.manual_seed(77)
                o_n, _ = mdist.sharded_infer(net, xg, xlg, sidg, noise_scale=0.6, length_scale=1)
                res["mini_b5_noise"] = {"o": o_n.cpu()}
                # a bad token id in the LAST utterance (rank world-1's shard): every rank must raise,
                # none may be left waiting in a collective
                bad = xg.clone()
                bad[B - 1, 0] = net.cfg.n_vocab
                try:
                    mdist.sharded_infer(net, bad, xlg, sidg, noise_scale=0, length_scale=1)
                    res["bad_token"] = "no error"
                except IndexError:
                    res["bad_token"] = "IndexError"
                # batch smaller than the world: the same ValueError on every rank, before any collective
                try:
                    mdist.sharded_infer(net, xg[:world - 1], xlg[:world - 1], None, noise_scale=0)
                    res["small_batch"] = "no error"
                except ValueError:
                    res["small_batch"] = "ValueError"
                # the group is still usable afterwards
                o2, _ = mdist.sharded_infer(net, xg, xlg, sidg, noise_scale=0, length_scale=1)
                res["after_errors_equal"] = bool(torch.equal(o2, o))
                # gathers on a side stream: handle now, tensors at result(); and the decoder in two halves with the
                # first half's rows travelling under the second half's decode — both bitwise the plain call
                tm = mdist.StepTimes()
                h = mdist.sharded_infer(net, xg, xlg, sidg, noise_scale=0, length_scale=1, overlap="next", timing=tm)
                o3, y3 = h.result()
                torch.cuda.synchronize()
                res["overlap_next_equal"] = bool(torch.equal(o3, o) and torch.equal(y3, ylen))
                res["timing_next"] = tm.ms()
                tm = mdist.StepTimes()
                o4, y4 = mdist.sharded_infer(net, xg, xlg, sidg, noise_scale=0, length_scale=1, overlap="halves", timing=tm,
                                             outputs=None)
                torch.cuda.synchronize()
                res["overlap_halves_equal"] = bool(torch.equal(o4, o) and torch.equal(y4, ylen))
                res["overlap_halves_maxdiff"] = float((o4 - o).abs().max())
                res["timing_halves"] = tm.ms()
                tm = mdist.StepTimes()
                mdist.sharded_infer(net, xg, xlg, sidg, noise_scale=0, length_scale=1, timing=tm)
                torch.cuda.synchronize()
                res["timing_plain"] = tm.ms()
                # a quiet call (noise_scale 0) between the seed and a noisy call: the generator must have advanced as
                # in a single process, so the noisy call still reproduces the single-process draw
                torch.manual_seed(78)
                torch.cuda.manual_seed(78)
                mdist.sharded_infer(net, xg, xlg, sidg, noise_scale=0, length_scale=1)
                o_n2, _ = mdist.sharded_infer(net, xg, xlg, sidg, noise_scale=0.6, length_scale=1)
                res["mini_b5_noise_after_quiet"] = {"o": o_n2.cpu()}
        import ctypes as C
        loaded = [ln.split()[-1] for ln in open("/proc/self/maps") if "libmbistft_vits.so" in ln]
        res["native_loaded"] = bool(loaded)
        torch.save(res, os.path.join(outdir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
