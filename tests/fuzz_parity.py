#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random configuration family, batch, text length, raggedness,
speaker ids, length_scale, max_len, SDP on/off, split-K on/off — HIP path vs the oracle.
usage: python tests/fuzz_parity.py [n_cases] [seed]     (exit code 1 on the first violation);
also driven with a small case count by tests/test_gpu_fuzz.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))       # (test infrastructure: uses the oracle)
import numpy as np
import torch
from gpu_util import make_net
from helpers import rms
from mb_istft_vits_amd import synth
from oracle import ref_infer as R

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(16)
FAMILIES = [("ljs_mini_mb_istft_vits", {}), ("ljs_mini_istft_vits", {}), ("ljs_mini_mb_istft_vits", {"use_sdp": True}),
            ("ljs_mini_mb_istft_vits", {"resblock": "2", "resblock_dilation_sizes": [[1, 3], [1, 3], [1, 3]]}),
            ("uudb_ms_istft_vits_ms", {}), ("ljs_mb_istft_vits", {}), ("ljs_ms_istft_vits", {})]
def vc_case(net, sd, rs):
    """voice_conversion (models.py:790-798) on random spectrogram lengths / speaker pairs; the
    posterior draw is handed to the shim by patching torch.randn for the call."""
    B, T = int(rs.randint(1, 4)), int(rs.randint(3, 60))
    y = (np.abs(rs.standard_normal((B, net.cfg.spec_channels, T))) * 2.0).astype(np.float32)
    yl = rs.randint(max(1, T // 2), T + 1, size=(B,)).astype(np.int64)
    yl[rs.randint(B)] = T
    src = rs.randint(0, net.cfg.n_speakers, size=(B,)).astype(np.int64)
    tgt = rs.randint(0, net.cfg.n_speakers, size=(B,)).astype(np.int64)
    noise = rs.standard_normal((B, net.cfg.inter_channels, T)).astype(np.float32)
    ref = R.voice_conversion(sd, net.cfg, y, yl, src, tgt, noise=noise)
    ng = torch.from_numpy(noise).cuda()
    real = torch.randn
    torch.randn = lambda *a, **k: ng if tuple(a) == tuple(ng.shape) else real(*a, **k)
    try:
        o, o_mb, y_mask, (z, z_p, z_hat) = net.voice_conversion(
            torch.from_numpy(y).cuda(), torch.from_numpy(yl).cuda(), torch.from_numpy(src).cuda(),
            torch.from_numpy(tgt).cuda())
    finally:
        torch.randn = real
    err = rms(o.cpu().numpy() - ref["o"].numpy())
    zerr = rms(z_hat.cpu().numpy() - ref["z_hat"].numpy()) / max(rms(ref["z_hat"].numpy()), 1e-3)
    return "voice_conversion B=%d T=%d" % (B, T), err, zerr


nets = {}
worst = 0.0
done = 0
attempt = 0
while done < n_cases:
    attempt += 1
    fi = rs.randint(len(FAMILIES))
    cfg_name, ov = FAMILIES[fi]
    big = cfg_name in ("uudb_ms_istft_vits_ms", "ljs_mb_istft_vits", "ljs_ms_istft_vits")
    if fi not in nets:
        nets[fi] = make_net(cfg_name, seed=1300 + fi, overrides=ov or None)
    net, sd = nets[fi]
    if cfg_name == "uudb_ms_istft_vits_ms" and rs.randint(2):
        net.set_option("splitk", int(rs.randint(2)))
        what, err, zerr = vc_case(net, sd, rs)
        worst = max(worst, err)
        done += 1
        print("%3d %-24s %s  o rms err %.2e  z_hat rel %.2e" % (done, cfg_name, what, err, zerr), flush=True)
        if err >= 1e-4 or zerr >= 5e-5:
            print("VIOLATION"); sys.exit(1)
        continue
    B = int(rs.randint(1, 4 if big else 9))
    T = int(rs.randint(1, 40 if big else 150))
    ragged = bool(rs.randint(2))
    x, xl, sid = synth.synthetic_batch(net.cfg, B, T, seed=int(rs.randint(1 << 30)), ragged=ragged and T >= 2)
    ls = float(rs.choice([1.0, 0.8, 1.3]))
    splitk = int(rs.randint(2))
    nsw = float(rs.choice([0.0, 0.667]))
    noise_w = torch.from_numpy(rs.standard_normal((B, 2, T)).astype(np.float32))
    ref = R.infer(sd, net.cfg, x, xl, sid, length_scale=ls, noise_w=noise_w, noise_scale_w=nsw)
    w = (torch.exp(ref["logw"]) * ref["x_mask"] * ls).numpy()[ref["x_mask"].numpy() > 0]
    if w.size and np.min(np.abs(w - np.round(w))) < 3e-4:
        continue                                  # a duration on a ceil() edge: not a parity question
    Tp = int(ref["y_lengths"].max())
    max_len = None if rs.randint(3) else int(rs.randint(1, Tp + 1))
    if max_len is not None:
        ref = R.infer(sd, net.cfg, x, xl, sid, length_scale=ls, noise_w=noise_w, noise_scale_w=nsw, max_len=max_len)
    net.set_option("splitk", splitk)
    xg, xlg = torch.from_numpy(x).cuda(), torch.from_numpy(xl).cuda()
    sg = torch.from_numpy(sid).cuda() if sid is not None else None
    r = net._run(xg, xlg, sg, 0, ls, max_len, True, noise_scale_w=nsw, noise_w=noise_w if net.cfg.use_sdp else None)
    o, z, ylen = r[0], r[6][0], r[8]
    ok = np.array_equal(ylen.cpu().numpy(), ref["y_lengths"].numpy()) and o.shape == ref["o"].shape
    err = rms(o.cpu().numpy() - ref["o"].numpy()) if ok else float("inf")
    zerr = rms(z.cpu().numpy() - ref["z"].numpy()) / max(rms(ref["z"].numpy()), 1e-3) if ok else float("inf")
    worst = max(worst, err)
    done += 1
    print("%3d %-24s %-12s B=%d T=%3d ragged=%d ls=%.1f max_len=%-5s splitk=%d nsw=%.3f  T'=%4d  o rms err %.2e  z rel %.2e"
          % (done, cfg_name, ",".join(ov) or "-", B, T, ragged, ls, max_len, splitk, nsw, Tp, err, zerr), flush=True)
    if not ok or err >= 1e-4 or zerr >= 5e-5:
        print("VIOLATION"); sys.exit(1)
print("fuzz: %d cases, worst waveform rms error %.2e (bar 1e-4)" % (done, worst))
