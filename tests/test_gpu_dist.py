"""-m gpu: the sharded path with TWO ranks running the HIP kernels.

A one-GPU box cannot host two RCCL ranks, so both child ranks use cuda:0 and the `gloo` backend;
everything else is the product path: `dist.broadcast_state_dict`, `dist.sharded_infer` (its
[T', error flag] all-reduce and the waveform all-gather run on CUDA tensors), `SynthesizerTrn._run`
over `libmbistft_vits.so`.  The gathered result must equal, bitwise, a single-process `infer` of the
whole batch — uneven shards (5 = 3 + 2), a multi-speaker model with `sid`, and the error paths.
The real-RCCL N > 1 run is the driver's (`bench.py --gpus N` on an 8-GPU node)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(900)
def test_two_rank_hip_sharded_infer_equals_full_batch(tmp_path):
    from gpu_util import make_net
    from mb_istft_vits_amd import synth
    port = _free_port()
    world = 2
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_gpu_worker.py"), str(r), str(world),
                               str(port), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=800)[0])
    finally:
        for p in procs:                         # exact PIDs we started; never by pattern
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-4000:])
    res = [torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(world)]
    assert all(r["native_loaded"] for r in res)
    # every rank holds the same gathered result
    for k in ("mini_b5", "uudb_b5", "mini_b4", "mini_b5_noise"):
        assert torch.equal(res[0][k]["o"], res[1][k]["o"]), k
    for r in res:
        assert r["bad_token"] == "IndexError" and r["small_batch"] == "ValueError"
        assert r["after_errors_equal"] and r["mini_b5_all_outputs_equal"]
        assert r["overlap_next_equal"]
        if os.environ.get("MBV_CONV_SPLITK", "0") not in ("", "0"):
            assert r["overlap_halves_maxdiff"] < 2e-5      # low-latency mode: kernels picked by launch size, equal within rounding
        else:
            assert r["overlap_halves_equal"]
        assert r["mini_b5_arena_floats"] == res[0]["mini_b5_arena_floats"] > 0
        for k in ("timing_plain", "timing_next", "timing_halves"):
            t = r[k]
            assert t["world_size"] == world and t["total_ms"] > 0 and t["all_reduce_ms"] >= 0 and t["gather_o_ms"] > 0, (k, t)
            assert t["overlap"] == (k != "timing_plain")

    # single-process full-batch reference on the same device, same kernels
    for name, cfg_name, B, T in (("mini_b5", "ljs_mini_mb_istft_vits", 5, 23),
                                 ("uudb_b5", "uudb_ms_istft_vits_ms", 5, 17),
                                 ("mini_b4", "ljs_mini_mb_istft_vits", 4, 31)):
        net, sd = make_net(cfg_name, seed=1300)
        x, xl, sid = synth.synthetic_batch(net.cfg, B, T, seed=40 + B, ragged=True)
        xg, xlg = torch.from_numpy(x).cuda(), torch.from_numpy(xl).cuda()
        sidg = torch.from_numpy(sid).cuda() if sid is not None else None
        (o, *_), ylen = net.infer_with_lengths(xg, xlg, sidg, noise_scale=0, length_scale=1)
        assert torch.equal(res[0][name]["ylen"], ylen.cpu()), name
        assert res[0][name]["o"].shape == o.shape, name
        if os.environ.get("MBV_CONV_SPLITK", "0") not in ("", "0"):
            # the opt-in low-latency mode picks kernels by launch size: a shard equals the full batch within rounding
            assert float((res[0][name]["o"] - o.cpu()).abs().max()) < 2e-5, name
        else:
            assert torch.equal(res[0][name]["o"], o.cpu()), name      # bitwise: global T' pad, same kernels
        if name == "mini_b5":
            torch.manual_seed(77)
            torch.cuda.manual_seed(77)
            o_n = net.infer(xg, xlg, sidg, noise_scale=0.6, length_scale=1)[0]
            if os.environ.get("MBV_CONV_SPLITK", "0") not in ("", "0"):
                assert float((res[0]["mini_b5_noise"]["o"] - o_n.cpu()).abs().max()) < 2e-5
            else:
                assert torch.equal(res[0]["mini_b5_noise"]["o"], o_n.cpu())
            assert not torch.equal(o_n, o)
            # generator advance at noise_scale == 0: quiet call, then noisy call == single process doing the same
            torch.manual_seed(78)
            torch.cuda.manual_seed(78)
            net.infer(xg, xlg, sidg, noise_scale=0, length_scale=1)
            o_n2 = net.infer(xg, xlg, sidg, noise_scale=0.6, length_scale=1)[0]
            if os.environ.get("MBV_CONV_SPLITK", "0") in ("", "0"):
                assert torch.equal(res[0]["mini_b5_noise_after_quiet"]["o"], o_n2.cpu())
                assert torch.equal(res[1]["mini_b5_noise_after_quiet"]["o"], o_n2.cpu())


@pytest.mark.timeout(900)
def test_bench_self_launch_two_ranks_rehearsal(tmp_path):
    """`python bench.py --gpus 2` as typed from a plain shell: the launcher starts the ranks as child
    processes before any GPU call; on a box with fewer devices than ranks it says so in the JSON
    (`rehearsal`), shares the device and uses gloo."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--t-text", "40", "--config", "ljs_mini_mb_istft_vits", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=800, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dist"]["world_size"] == 2 and len(j["dist"]["devices"]) == 2
    assert j["config"]["global_batch"] == 8
    if torch.cuda.device_count() < 2:
        assert j["dist"]["rehearsal"] and j["dist"]["backend"] == "gloo"
    assert j["value"] > 0


def test_rows_of_a_full_batch_draw_without_the_full_draw():
    """`rows_of_randn.randn_rows` == `torch.randn(b_all, I, T')[lo:hi]` bitwise, generator state included, through the
    partial-draw path (verified against the full draw on first use) — the sharded path's prior noise at 8 x 64."""
    from mb_istft_vits_amd import rows_of_randn as rr
    dev = torch.device("cuda", 0)
    rr._ok.clear()
    for b_all, lo, hi, tail in ((512, 64, 128, (192, 566)), (512, 448, 512, (192, 523)), (256, 0, 32, (192, 301)), (8, 2, 4, (192, 40))):
        for rep in range(2):                      # first use verifies, second takes the fast path
            torch.cuda.manual_seed(1234 + rep)
            want = torch.randn(b_all, *tail, device=dev)[lo:hi].clone()
            after = torch.randn(5, device=dev).clone()
            torch.cuda.manual_seed(1234 + rep)
            got = rr.randn_rows(lo, hi, b_all, tail, dev)
            after2 = torch.randn(5, device=dev)
            assert torch.equal(got, want), (b_all, lo, hi, rep)
            assert torch.equal(after, after2), "generator left elsewhere than the full draw leaves it"
    assert rr._ok.get(0) is True, "the partial-draw path was not verified on this torch build (it falls back to the full draw)"
