"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (the reference is not on the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

It imports `/root/reference` in-process (read-only, no bytecode written) with
the three shims SURVEY §8c lists — stub `monotonic_align` (training only),
stub `librosa` (dead `STFT` class only) and a CPU `.cuda()` no-op for
`pqmf.py:78-86` — builds `models.SynthesizerTrn` for each BASELINE config,
loads the deterministic synthetic checkpoint (`mb_istft_vits_amd.synth`) through
the reference's own `load_state_dict`, runs `infer(noise_scale=0,
length_scale=1)` on small ragged batches and stores inputs plus every stage
boundary.  Only data is stored; no reference source text.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MBV_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def import_reference():
    sys.path.insert(0, REF)
    sys.modules["monotonic_align"] = types.ModuleType("monotonic_align")
    lib = types.ModuleType("librosa")
    libu = types.ModuleType("librosa.util")
    libf = types.ModuleType("librosa.filters")
    for n in ("pad_center", "tiny", "normalize"):
        setattr(libu, n, lambda *a, **k: None)
    lib.util, lib.filters = libu, libf
    sys.modules.update({"librosa": lib, "librosa.util": libu, "librosa.filters": libf})
    import torch
    torch.Tensor.cuda = lambda self, device=None, **k: self.to(device) if device is not None else self
    import models    # noqa: the reference's
    import utils     # noqa
    import stft      # noqa
    import pqmf      # noqa
    return models, utils, stft, pqmf


def build_reference_model(models, utils, cfg_name, n_vocab, overrides=None):
    hps = utils.get_hparams_from_file(os.path.join(REF, "configs", cfg_name + ".json"))
    for k, v in (overrides or {}).items():          # e.g. resblock "2" (models.py:317), no such config ships
        hps.model[k] = v
    net = models.SynthesizerTrn(n_vocab, hps.data.filter_length // 2 + 1,
                                hps.train.segment_size // hps.data.hop_length,
                                n_speakers=hps.data.n_speakers, **hps.model).eval()
    return hps, net


def capture(net, x, x_lengths, sid, noise_seed=None):
    import torch
    taps = {}
    hooks = []

    def grab(name, idx=None):
        def fn(_m, _inp, out):
            t = out if idx is None else out[idx]
            taps[name] = t.detach().clone()
        return fn

    hooks.append(net.enc_p.register_forward_hook(
        lambda m, i, o: taps.update(x_enc=o[0].clone(), m_text=o[1].clone(),
                                    logs_text=o[2].clone(), x_mask=o[3].clone())))
    hooks.append(net.dp.register_forward_hook(grab("logw")))
    if getattr(net, "use_sdp", False):              # models.py:89-100: conditioning + reversed flows
        hooks.append(net.dp.proj.register_forward_hook(grab("sdp_proj")))
        for f in (7, 5, 3):
            hooks.append(net.dp.flows[f].register_forward_hook(grab("sdp_flow_%d" % f)))
    for f in range(4):
        hooks.append(net.flow.flows[2 * f].register_forward_hook(grab("flow_after_%d" % f)))
    hooks.append(net.dec.conv_pre.register_forward_hook(grab("dec_conv_pre")))
    for i in range(2):
        hooks.append(net.dec.ups[i].register_forward_hook(grab("dec_up_%d" % i)))
    for j in range(6):
        hooks.append(net.dec.resblocks[j].register_forward_hook(grab("_rb%d" % j)))
    post = net.dec.subband_conv_post if hasattr(net.dec, "subband_conv_post") else net.dec.conv_post
    hooks.append(post.register_forward_hook(grab("x_post")))
    with torch.no_grad():
        if noise_seed is not None:
            # the SDP draws torch.randn(B, 2, T) on the default CPU generator (models.py:94), the
            # first draw of infer: re-seeding reproduces it as data for the fixture
            torch.manual_seed(noise_seed)
            taps["noise_w"] = torch.randn(x.size(0), 2, x.size(1))
            torch.manual_seed(noise_seed)
        o, o_mb, spec, phase, attn, y_mask, (z, z_p, m_p, logs_p), timings = net.infer(
            x, x_lengths, sid=sid, noise_scale=0, length_scale=1, noise_scale_w=0.8)
    for h in hooks:
        h.remove()
    for i in range(2):
        xs = taps["_rb%d" % (3 * i)]
        xs = xs + taps["_rb%d" % (3 * i + 1)]
        xs = xs + taps["_rb%d" % (3 * i + 2)]
        taps["dec_res_%d" % i] = xs / 3
    for j in range(6):
        del taps["_rb%d" % j]
    taps.update(o=o, spec=spec, phase=phase, attn=attn, y_mask=y_mask, z=z, z_p=z_p,
                m_p=m_p, logs_p=logs_p)
    if o_mb is not None:                            # iSTFT_Generator returns None (models.py:300)
        taps["o_mb"] = o_mb
    assert sorted(timings) == sorted(["text_encoder", "duration_predictor",
                                      "alignment_and_projection", "flow", "waveform_decoder"])
    return taps


def thin(name, a):
    """Keep fixtures small: big decoder intermediates are stored subsampled."""
    if name in ("dec_conv_pre", "dec_up_0", "dec_up_1", "dec_res_0", "dec_res_1"):
        return a[:, ::16, :]                       # every 16th channel
    if name in ("spec", "phase"):
        return a[..., ::5]                         # every 5th frame
    if name == "attn":
        return a.sum(2)                            # row sums == w_ceil (models.py:680)
    return a


CASES = [
    # (fixture, config, n_vocab, batch, T_text, lengths, weight seed)
    ("mini_b1", "ljs_mini_mb_istft_vits", 59, 1, 20, [20], 1234),
    ("mb_b3", "ljs_mb_istft_vits", 59, 3, 28, [28, 17, 23], 1234),
    ("ms_b2", "ljs_ms_istft_vits", 59, 2, 24, [24, 15], 1234),
    ("uudb_b2", "uudb_ms_istft_vits_ms", 59, 2, 24, [19, 24], 1234),
    ("mb_short", "ljs_mb_istft_vits", 59, 2, 3, [1, 3], 1234),
    # single-band iSTFT_Generator family and ResBlock2 (row a15)
    ("sb_mini_b2", "ljs_mini_istft_vits", 59, 2, 12, [12, 7], 1234),
    ("rb2_mini_b2", "ljs_mini_mb_istft_vits", 59, 2, 16, [16, 11], 1234,
     {"resblock": "2", "resblock_dilation_sizes": [[1, 3], [1, 3], [1, 3]]}),
    # StochasticDurationPredictor, reverse (no reference config sets use_sdp)
    ("sdp_mini_b2", "ljs_mini_mb_istft_vits", 59, 2, 18, [18, 11], 1234, {"use_sdp": True}),
    ("sdp_uudb_b2", "uudb_ms_istft_vits_ms", 59, 2, 14, [9, 14], 1234, {"use_sdp": True}),
]


def make_voice_conversion_golden(models, utils):
    """SynthesizerTrn.voice_conversion (models.py:790-798) on the multi-speaker config; the
    posterior encoder's randn_like draw is pinned by patching it for the duration of the call."""
    import torch
    from mb_istft_vits_amd import synth, spec as mspec, utils as mutils
    cfg_name, n_vocab, B, T = "uudb_ms_istft_vits_ms", 59, 2, 26
    hps, net = build_reference_model(models, utils, cfg_name, n_vocab)
    my_hps = mutils.get_hparams_from_file(mutils.builtin_config(cfg_name))
    cfg = mspec.config_from_ctor(n_vocab, my_hps.data.filter_length // 2 + 1,
                                 my_hps.train.segment_size // my_hps.data.hop_length,
                                 n_speakers=my_hps.data.n_speakers, **my_hps.model)
    sd = synth.make_state_dict(cfg, 1234)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    rs = np.random.RandomState(77)
    y = np.abs(rs.standard_normal((B, 513, T))).astype(np.float32) * 2.0       # magnitude spectrogram
    yl = np.asarray([26, 17], np.int64)
    sid_src, sid_tgt = np.asarray([3, 9], np.int64), np.asarray([7, 0], np.int64)
    noise = rs.standard_normal((B, 192, T)).astype(np.float32)
    real = torch.randn_like
    torch.randn_like = lambda t, **k: torch.from_numpy(noise) if tuple(t.shape) == noise.shape else real(t, **k)
    try:
        with torch.no_grad():
            o, o_mb, y_mask, (z, z_p, z_hat) = net.voice_conversion(
                torch.from_numpy(y), torch.from_numpy(yl), torch.from_numpy(sid_src), torch.from_numpy(sid_tgt))
    finally:
        torch.randn_like = real
    np.savez_compressed(os.path.join(HERE, "vc_uudb_b2.npz"), y=y, y_lengths=yl, sid_src=sid_src,
                        sid_tgt=sid_tgt, noise=noise, o=o.numpy(), o_mb=o_mb.numpy(), y_mask=y_mask.numpy(),
                        z=z.numpy(), z_p=z_p.numpy(), z_hat=z_hat.numpy(), weight_seed=np.int64(1234),
                        n_vocab=np.int64(n_vocab))
    print("vc_uudb_b2  o", tuple(o.shape), "|o|rms=%.4f" % float(o.pow(2).mean().sqrt()),
          "%.0f KB" % (os.path.getsize(os.path.join(HERE, "vc_uudb_b2.npz")) / 1024))


def make_params_and_dec_golden(models, utils):
    """The call-parameter surface of `infer` with the real reference: noise_scale > 0 (the
    randn_like draw of models.py:729 pinned by patching it), length_scale != 1, max_len
    truncation (models.py:733), and the decoder-only entry `net.dec(z_chunk)` the chunked
    decoding notebooks use."""
    import torch
    from mb_istft_vits_amd import synth, spec as mspec, utils as mutils
    cfg_name, n_vocab, B, T = "ljs_mb_istft_vits", 59, 2, 14
    hps, net = build_reference_model(models, utils, cfg_name, n_vocab)
    my_hps = mutils.get_hparams_from_file(mutils.builtin_config(cfg_name))
    cfg = mspec.config_from_ctor(n_vocab, my_hps.data.filter_length // 2 + 1,
                                 my_hps.train.segment_size // my_hps.data.hop_length,
                                 n_speakers=my_hps.data.n_speakers, **my_hps.model)
    length_scale, noise_scale, max_len = 1.2, 0.667, 30
    seed = 1234
    while True:
        sd = synth.make_state_dict(cfg, seed)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        rs = np.random.RandomState(211)
        x = rs.randint(1, n_vocab, size=(B, T)).astype(np.int64)
        xl = np.asarray([14, 9], np.int64)
        for b in range(B):
            x[b, xl[b]:] = 0
        with torch.no_grad():
            taps = {}
            hk = net.dp.register_forward_hook(lambda m, i, o: taps.update(logw=o.clone()))
            probe = net.infer(torch.from_numpy(x), torch.from_numpy(xl), noise_scale=0, length_scale=length_scale)
            hk.remove()
        w = (torch.exp(taps["logw"]) * probe[5].new_ones(1)).numpy()[:, 0] * length_scale
        w = np.concatenate([w[b, :xl[b]] for b in range(B)])
        margin = float(np.min(np.abs(w - np.round(w))))
        if margin >= 1e-3:
            break
        seed += 1
    Tp = probe[6][0].shape[-1]
    noise = rs.standard_normal((B, 192, Tp)).astype(np.float32)
    real = torch.randn_like
    torch.randn_like = lambda t, **k: torch.from_numpy(noise) if tuple(t.shape) == noise.shape else real(t, **k)
    try:
        with torch.no_grad():
            o, o_mb, spec, phase, attn, y_mask, (z, z_p, m_p, logs_p), _ = net.infer(
                torch.from_numpy(x), torch.from_numpy(xl), noise_scale=noise_scale,
                length_scale=length_scale, max_len=max_len)
            zc = z[:, :, 7:29].contiguous()                 # a chunk, as inferz_test.ipynb cell 7
            do, do_mb, dspec, dphase = net.dec(zc)
    finally:
        torch.randn_like = real
    np.savez_compressed(
        os.path.join(HERE, "params_mb_b2.npz"), x=x, x_lengths=xl, noise=noise,
        length_scale=np.float32(length_scale), noise_scale=np.float32(noise_scale), max_len=np.int64(max_len),
        weight_seed=np.int64(seed), n_vocab=np.int64(n_vocab), ceil_margin=np.float64(margin),
        o=o.numpy(), o_mb=o_mb.numpy(), attn=attn.sum(2).numpy(), y_mask=y_mask.numpy(), z=z.numpy(),
        z_p=z_p.numpy(), spec=spec.numpy()[..., ::5], phase=phase.numpy()[..., ::5],
        dec_chunk_lo=np.int64(7), dec_chunk_hi=np.int64(29), dec_o=do.numpy(), dec_o_mb=do_mb.numpy(),
        dec_spec=dspec.numpy()[..., ::5], dec_phase=dphase.numpy()[..., ::5])
    print("params_mb_b2 seed=%d T'=%d o %s dec_o %s margin=%.3g %.0f KB" % (
        seed, Tp, tuple(o.shape), tuple(do.shape), margin,
        os.path.getsize(os.path.join(HERE, "params_mb_b2.npz")) / 1024))


def main():
    import torch
    from mb_istft_vits_amd import synth, spec as mspec, utils as mutils
    torch.manual_seed(0)
    torch.set_num_threads(4)
    models, utils, stft, pqmf = import_reference()

    only = set(sys.argv[1:])
    for case in CASES:
        fixture, cfg_name, n_vocab, B, T, lens, wseed = case[:7]
        overrides = case[7] if len(case) > 7 else None
        if only and fixture not in only:
            continue
        hps, net = build_reference_model(models, utils, cfg_name, n_vocab, overrides)
        my_hps = mutils.get_hparams_from_file(mutils.builtin_config(cfg_name))
        for k, v in (overrides or {}).items():
            my_hps.model[k] = v
        cfg = mspec.config_from_ctor(n_vocab, my_hps.data.filter_length // 2 + 1,
                                     my_hps.train.segment_size // my_hps.data.hop_length,
                                     n_speakers=my_hps.data.n_speakers, **my_hps.model)
        seed = wseed
        while True:
            sd = synth.make_state_dict(cfg, seed)
            full = net.state_dict()
            missing = [k for k in full if k not in sd]
            extra = [k for k in sd if k not in full]
            assert not missing and not extra, (missing[:5], extra[:5])
            for k, v in sd.items():
                assert tuple(full[k].shape) == v.shape, (k, full[k].shape, v.shape)
                full[k] = torch.from_numpy(v)
            net.load_state_dict(full)
            rs = np.random.RandomState(100 + len(fixture))
            x = rs.randint(1, n_vocab, size=(B, T)).astype(np.int64)
            xl = np.asarray(lens, np.int64)
            for b in range(B):
                x[b, xl[b]:] = 0
            sid = rs.randint(0, cfg.n_speakers, size=(B,)).astype(np.int64) if cfg.has_speaker else None
            taps = capture(net, torch.from_numpy(x), torch.from_numpy(xl),
                           torch.from_numpy(sid) if sid is not None else None,
                           noise_seed=(4321 + seed) if cfg.use_sdp else None)
            w = (torch.exp(taps["logw"]) * taps["x_mask"]).numpy()
            w = w[taps["x_mask"].numpy() > 0]
            margin = float(np.min(np.abs(w - np.round(w))))
            if margin >= 1e-3:
                break
            print("  seed %d rejected: ceil margin %.2e" % (seed, margin))
            seed += 1
        out = {"x": x, "x_lengths": xl, "weight_seed": np.int64(seed), "n_vocab": np.int64(n_vocab),
               "ceil_margin": np.float64(margin)}
        if sid is not None:
            out["sid"] = sid
        for k, v in taps.items():
            out[k] = thin(k, v).numpy()
        np.savez_compressed(os.path.join(HERE, fixture + ".npz"), **out)
        size = os.path.getsize(os.path.join(HERE, fixture + ".npz"))
        print("%-10s cfg=%s seed=%d T'=%d o=%s |o|rms=%.4f margin=%.3g  %.0f KB" % (
            fixture, cfg_name, seed, taps["z"].shape[-1], tuple(taps["o"].shape),
            float(taps["o"].pow(2).mean().sqrt()), margin, size / 1024))

    if not only or "vc_uudb_b2" in only:
        make_voice_conversion_golden(models, utils)
    if not only or "params_mb_b2" in only:
        make_params_and_dec_golden(models, utils)
    if only and "signal_ops" not in only:
        return
    # ---- stand-alone known-answer vectors for the signal ops -------------
    rs = np.random.RandomState(7)
    st = stft.TorchSTFT(filter_length=16, hop_length=4, win_length=16)
    mag = np.exp(rs.standard_normal((8, 9, 41)) * 0.5).astype(np.float32)
    ph = (np.pi * np.sin(rs.standard_normal((8, 9, 41)) * 2)).astype(np.float32)
    y = st.inverse(torch.from_numpy(mag), torch.from_numpy(ph))          # [8,1,160]
    pq = pqmf.PQMF("cpu")
    sub = rs.standard_normal((2, 4, 200)).astype(np.float32)
    full_band = pq.synthesis(torch.from_numpy(sub))                       # [2,1,800]
    np.savez_compressed(
        os.path.join(HERE, "signal_ops.npz"), istft_mag=mag, istft_phase=ph,
        istft_out=y.numpy(), hann16=st.window.numpy(), pqmf_in=sub, pqmf_out=full_band.numpy(),
        pqmf_synthesis_filter=pq.synthesis_filter.numpy()[0])
    print("signal_ops  istft", tuple(y.shape), "pqmf", tuple(full_band.shape))


if __name__ == "__main__":
    main()
