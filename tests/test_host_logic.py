"""CPU-only checks of the host side: drop-in API surface, hps handling, C-ABI library
symbols, error behaviour without a GPU.  No compute is launched here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

from mb_istft_vits_amd import _capi, models, spec, synth, utils
from helpers import config_for, FIXTURES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net(name="ljs_mini_mb_istft_vits"):
    hps = utils.get_hparams_from_file(utils.builtin_config(name))
    return hps, models.SynthesizerTrn(59, hps.data.filter_length // 2 + 1,
                                      hps.train.segment_size // hps.data.hop_length,
                                      n_speakers=hps.data.n_speakers, **hps.model)


def test_header_symbols_all_exported():
    """Every function `include/mbistft_vits.h` declares is exported by the built library."""
    hdr = open(os.path.join(ROOT, "include", "mbistft_vits.h")).read()
    declared = set(re.findall(r"\b(mbv_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mbv_model", "mbv_config", "mbv_outputs"}
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    L = _capi.lib()
    for sym in declared:
        assert getattr(L, sym) is not None
    assert L.mbv_abi_version() == _capi.ABI_VERSION == 3


def test_config_struct_layout_matches_header():
    # 10 scalars + 3 kernel sizes + 9 dilations + resblock_type + 4 trailing scalars, all int32
    assert C.sizeof(_capi.MbvConfig) == 4 * (10 + 3 + 9 + 1 + 4 + 1)
    assert C.sizeof(_capi.MbvOutputs) == 8 * 10


def test_create_fails_loudly_without_gpu_or_with_bad_config():
    L = _capi.lib()
    h = C.c_void_p()
    c = _capi.MbvConfig()
    c.struct_bytes = 7                         # ABI mismatch
    assert L.mbv_create(C.byref(c), C.byref(h)) != 0
    assert b"struct_bytes" in L.mbv_last_error(None)
    if not torch.cuda.is_available():
        c.struct_bytes = C.sizeof(_capi.MbvConfig)
        c.n_vocab, c.inter_channels, c.hidden_channels, c.filter_channels = 59, 192, 192, 768
        c.n_heads, c.n_layers, c.kernel_size, c.upsample_initial_channel = 2, 6, 3, 512
        c.spec_channels = 513
        for j, k in enumerate((3, 7, 11)):
            c.resblock_kernel_sizes[j] = k
            for q, d in enumerate((1, 3, 5)):
                c.resblock_dilations[j][q] = d
        c.resblock_type = 1
        assert L.mbv_create(C.byref(c), C.byref(h)) != 0
        assert b"no CPU fallback" in L.mbv_last_error(None)


def test_state_dict_keys_match_reference_contract():
    for name in set(FIXTURES.values()):
        _, net = _net(name)
        keys = list(net.state_dict().keys())
        want = spec.param_shapes(net.cfg)
        assert keys == list(want.keys())
        for k, t in net.state_dict().items():
            assert tuple(t.shape) == tuple(want[k]), k
    # counts probed from the reference (uudb has 383 keys outside enc_q, + 103 in enc_q)
    _, net = _net("uudb_ms_istft_vits_ms")
    assert len(net.state_dict()) == 383 + 103
    assert hasattr(net, "emb_g") and net.n_speakers == 12 and callable(net.dec)


def test_load_checkpoint_is_key_tolerant(tmp_path):
    hps, net = _net()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    dropped = "dec.conv_pre.bias"
    saved = {k: v + 1.0 for k, v in sd.items() if k != dropped}
    saved["dummy_discriminator.weight"] = torch.zeros(3)   # keys the model does not have are ignored
    path = tmp_path / "G_1.pth"
    torch.save({"model": saved, "iteration": 7, "optimizer": None, "learning_rate": 2e-4}, path)
    m, opt, lr, it = utils.load_checkpoint(str(path), net, None)
    assert (it, lr, opt) == (7, 2e-4, None) and m is net
    new = net.state_dict()
    assert torch.equal(new[dropped], sd[dropped])          # kept (utils.py:35-40)
    assert torch.equal(new["enc_p.emb.weight"], sd["enc_p.emb.weight"] + 1.0)


def test_hparams_behaves_like_reference():
    hps = utils.get_hparams_from_file(utils.builtin_config("ljs_mb_istft_vits"))
    assert hps.model.n_heads == 2 and hps["model"]["n_heads"] == 2
    assert "model" in hps and len(hps.model) == len(list(hps.model.keys()))
    kw = dict(**hps.model)
    assert kw["mb_istft_vits"] is True and kw["subbands"] == 4
    hps.train["seed"] = 5
    assert hps.train.seed == 5


def test_ctor_rejects_what_is_out_of_scope():
    hps = utils.get_hparams_from_file(utils.builtin_config("ljs_mb_istft_vits"))
    kw = dict(**hps.model)
    sdp = models.SynthesizerTrn(59, 513, 32, **{**kw, "use_sdp": True})   # models.py:649-650
    assert "dp.flows.7.proj.weight" in sdp.state_dict() and "dp.post_flows.0.m" in sdp.state_dict()
    assert "dp.conv_1.weight" not in sdp.state_dict()
    with pytest.raises(ValueError):                   # single-band family needs its own ups (8, 8)
        models.SynthesizerTrn(59, 513, 32, **{**kw, "mb_istft_vits": False, "istft_vits": True})
    with pytest.raises(ValueError):
        models.SynthesizerTrn(59, 513, 32, **{**kw, "mb_istft_vits": False})   # "Decoder Error"
    with pytest.raises(ValueError):                   # ResBlock2 takes 2 dilations per block
        models.SynthesizerTrn(59, 513, 32, **{**kw, "resblock": "2"})
    sb = dict(**utils.get_hparams_from_file(utils.builtin_config("ljs_mini_istft_vits")).model)
    net = models.SynthesizerTrn(59, 513, 32, **sb)
    assert "dec.conv_post.weight_v" in net.state_dict() and net.cfg.samples_per_frame == 256


def test_cpu_model_raises_instead_of_falling_back():
    _, net = _net()
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        net.infer(torch.zeros(1, 4, dtype=torch.long), torch.tensor([4]))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        net.dec(torch.zeros(1, 192, 4))
    with pytest.raises(AssertionError):                  # single-speaker model (models.py:791)
        net.voice_conversion(None, None, None, None)


def test_synthetic_checkpoint_is_deterministic_and_usable():
    _, cfg = config_for("ljs_mb_istft_vits")
    a, b = synth.make_state_dict(cfg, 1234), synth.make_state_dict(cfg, 1234)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c = synth.make_state_dict(cfg, 1235)
    assert not np.array_equal(a["dec.conv_pre.weight_v"], c["dec.conv_pre.weight_v"])
    assert np.abs(a["flow.flows.0.post.weight"]).max() > 0          # flows are not the identity
    assert np.allclose(a["dp.proj.bias"], np.log(2.5))
    x, xl, sid = synth.synthetic_batch(cfg, 8, 200, seed=0, ragged=True)
    assert x.shape == (8, 200) and xl.max() == 200 and sid is None
    assert all((x[i, xl[i]:] == 0).all() and (x[i, :xl[i]] > 0).all() for i in range(8))


def test_bench_flop_model_matches_survey():
    import bench
    _, cfg = config_for("ljs_mb_istft_vits")
    assert abs(bench.decoder_flops_per_frame(cfg) / 1e6 - 143.9) < 0.1
    _, mini = config_for("ljs_mini_mb_istft_vits")
    assert abs(bench.decoder_flops_per_frame(mini) / 1e6 - 36.8) < 0.1
    assert bench.ISTFT_BYTES_PER_FRAME == 5632


def test_graft_entry_build_runs_without_gpu():
    """`__graft_entry__.build()` is the driver's does-it-build check on the CPU container."""
    import __graft_entry__ as g
    g.build()


def test_timings_dict_behaves_like_the_reference_plain_dict():
    """`timings` (models.py:698-737) is a plain dict in the reference; ours fills itself lazily from
    HIP events, and EVERY way of reading it must see the resolved values."""
    import json
    import pickle
    from mb_istft_vits_amd.models import Timings, _STAGES

    class Owner:
        calls = 0

        def _stage_times(self, ticket):
            Owner.calls += 1
            return [0.001 * (i + 1) for i in range(5)]

    want = {k: 0.001 * (i + 1) for i, k in enumerate(_STAGES)}
    assert json.loads(json.dumps(Timings(Owner(), 1))) == want
    assert Timings(Owner(), 1).get("flow") == want["flow"]
    assert Timings(Owner(), 1).get("nope", 7) == 7
    assert Timings(Owner(), 1).copy() == want
    assert Timings(Owner(), 1) == want and not (Timings(Owner(), 1) != want)
    assert Timings(Owner(), 1) != {}
    assert pickle.loads(pickle.dumps(Timings(Owner(), 1))) == want
    assert dict(Timings(Owner(), 1)) == want and {**Timings(Owner(), 1)} == want
    assert list(Timings(Owner(), 1)) == list(_STAGES) and len(Timings(Owner(), 1)) == 5
    assert sorted(Timings(Owner(), 1).values()) == sorted(want.values())
    t = Timings(Owner(), 1)
    n = Owner.calls
    assert t["flow"] == want["flow"] and t["text_encoder"] == want["text_encoder"] and "flow" in t
    assert Owner.calls == n + 1                      # resolved once


def test_bench_self_launch_builds_a_child_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher: ranks are started as CHILD processes
    (torch.distributed.run), never by replacing this process; with fewer devices than ranks the
    launcher switches to the marked rehearsal mode."""
    import importlib
    import sys
    bench = importlib.import_module("bench")
    import torch
    calls = {}

    def fake_call(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.self_launch(bench.parse_args(["--gpus", "4", "--steps", "2"])) == 0
    cmd = calls["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and "4" in cmd
    assert "127.0.0.1" in cmd and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert calls["env"]["MBV_BENCH_SHARE_DEVICES"] == "1" and calls["env"]["MBV_BENCH_BACKEND"] == "gloo"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    bench.self_launch(bench.parse_args(["--gpus", "4"]))
    assert "MBV_BENCH_SHARE_DEVICES" not in calls["env"]
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    assert bench.self_launch(bench.parse_args(["--gpus", "2"])) == 1


def test_wire_framing_matches_the_service_wrapper():
    """20 ms int16 frames as base64 text (tts_vits.py:36-38, 219-226): against the oracle's loop and
    against known answers (frame count, last short frame, byte-exact round trip)."""
    import base64
    from mb_istft_vits_amd import wire
    from oracle import ref_infer as R
    rs = np.random.RandomState(0)
    pcm = rs.randint(-32768, 32767, size=22050 + 137).astype(np.int16)
    for rate, fl in ((24000, 0.02), (22050, 0.02), (16000, 0.0125)):
        got = wire.frame_pcm16(pcm, rate, fl)
        assert got == R.frame_pcm16(pcm, rate, fl)
        n = round(fl * rate)
        assert wire.chunk_size(rate, fl) == n and len(got) == -(-len(pcm) // n)
        back = np.frombuffer(b"".join(base64.b64decode(s) for s in got), dtype="<i2")
        assert np.array_equal(back, pcm)
        assert len(base64.b64decode(got[-1])) == 2 * (len(pcm) - n * (len(got) - 1))
    assert wire.frame_pcm16(pcm, 24000, valid_samples=1000) == R.frame_pcm16(pcm[:1000], 24000)
    assert wire.frame_pcm16(np.zeros(0, np.int16), 24000) == []
    # known answer: 480 samples at 24 kHz = one frame of 960 bytes -> 1280 base64 characters
    one = wire.frame_pcm16(np.arange(480, dtype=np.int16), 24000)
    assert len(one) == 1 and len(one[0]) == 1280 and one[0].startswith("AAABAAIAAwAEAAUA")
    import pytest as _pt
    with _pt.raises(ValueError):
        wire.frame_pcm16(pcm.astype(np.float32), 24000)
    assert wire.frame_pcm16(torch.from_numpy(pcm[:960]), 24000) == R.frame_pcm16(pcm[:960], 24000)


def test_reference_binding_binds_the_real_reference_class():
    """`reference_binding.bind` over the REAL `/root/reference/models.SynthesizerTrn` (build container only —
    the reference does not travel to the GPU box, where this test skips): construct through the reference's
    own ctor and hparams, strict `load_state_dict` of the synthetic checkpoint, every inference entry point of
    the class is the binding's, `model.dec` is routed, and with no GPU the calls raise instead of falling
    back to the reference's eager CPU modules.  Runs in a child process: importing the reference needs
    `sys.modules` stubs and a `.cuda()` no-op (tests/golden/make_golden.py) that must not leak into this one."""
    import subprocess
    import sys
    ref = os.environ.get("MBV_REFERENCE", "/root/reference")
    if not os.path.isfile(os.path.join(ref, "models.py")):
        pytest.skip("reference checkout not present (GPU box)")
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests", "golden"))
import torch
import make_golden
models, utils, stft, pqmf = make_golden.import_reference()
from mb_istft_vits_amd import synth, spec
from mb_istft_vits_amd.reference_binding import bind
Bound = bind(models.SynthesizerTrn)
assert Bound.__name__ == "SynthesizerTrn" and issubclass(Bound, models.SynthesizerTrn)
for name in ("infer", "infer_z_only", "voice_conversion", "decode"):
    assert name in Bound.__dict__, name
for cfg_name in ("ljs_mini_mb_istft_vits", "uudb_ms_istft_vits_ms"):
    hps = utils.get_hparams_from_file(os.path.join(make_golden.REF, "configs", cfg_name + ".json"))
    net = Bound(59, hps.data.filter_length // 2 + 1, hps.train.segment_size // hps.data.hop_length,
                n_speakers=hps.data.n_speakers, **hps.model).eval()
    cfg = spec.config_from_ctor(59, hps.data.filter_length // 2 + 1, hps.train.segment_size // hps.data.hop_length,
                                n_speakers=hps.data.n_speakers, **hps.model)
    sd = synth.make_state_dict(cfg, 1234)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not missing and not unexpected
    x, xl = torch.randint(1, 59, (2, 9)), torch.tensor([9, 7])
    sid = torch.tensor([0, 1]) if hps.data.n_speakers else None
    for call in (lambda: net.infer(x, xl, sid=sid, noise_scale=0), lambda: net.infer_z_only(x, xl, sid=sid),
                 lambda: net.dec(torch.zeros(1, hps.model["inter_channels"], 8)),
                 lambda: net.decode(torch.zeros(1, hps.model["inter_channels"], 8))):
        try:
            call()
        except RuntimeError as e:
            assert "no CPU path" in str(e), e
        else:
            raise AssertionError("a CPU call went through: the binding fell back to the reference's modules")
print("BOUND-OK")
""" % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "BOUND-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_integration_md_quotes_the_shipped_binding_verbatim():
    """INTEGRATION.md section B is the text of mb-istft-vits_amd/reference_binding.py (scripts/sync_integration_md.py)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sync_integration_md.py"), "--check"])
    assert r.returncode == 0, "INTEGRATION.md section B is stale: run python scripts/sync_integration_md.py"
