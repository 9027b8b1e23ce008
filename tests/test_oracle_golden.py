"""Pins the oracle (`oracle/ref_infer.py`) to vectors captured from the real
reference (`tests/golden/make_golden.py`): every stage boundary of
`SynthesizerTrn.infer`, for all four BASELINE configs + a 1..3-token edge case,
plus the stand-alone `TorchSTFT.inverse` / `PQMF.synthesis` known answers."""
import numpy as np
import pytest
import torch

from oracle import ref_infer as R
from helpers import FIXTURES, OVERRIDES, SDP_NOISE_SCALE_W, load_fixture, config_for, thin, rms
from mb_istft_vits_amd import synth

STAGES = ["x_enc", "m_text", "logs_text", "sdp_proj", "sdp_flow_7", "sdp_flow_5", "sdp_flow_3", "logw", "attn", "m_p", "logs_p", "z_p",
          "flow_after_3", "flow_after_2", "flow_after_1", "flow_after_0", "z",
          "dec_conv_pre", "dec_up_0", "dec_res_0", "dec_up_1", "dec_res_1", "x_post",
          "spec", "phase", "o_mb", "o"]


@pytest.mark.parametrize("fixture", list(FIXTURES))
def test_infer_matches_reference(fixture):
    gold = load_fixture(fixture)
    _, cfg = config_for(FIXTURES[fixture], int(gold["n_vocab"]), OVERRIDES.get(fixture))
    sd = synth.make_state_dict(cfg, int(gold["weight_seed"]))
    torch.set_num_threads(4)
    out = R.infer(sd, cfg, gold["x"], gold["x_lengths"], gold.get("sid"), want_taps=True,
                  noise_w=gold.get("noise_w"), noise_scale_w=SDP_NOISE_SCALE_W)
    assert np.array_equal(out["y_lengths"].numpy(), gold["y_mask"].sum((1, 2)).astype(np.int64))
    assert np.array_equal(thin("attn", out["attn"]).numpy(), gold["attn"])          # durations exact
    for name in STAGES:
        if name not in gold:                      # o_mb: iSTFT_Generator returns None (models.py:300)
            assert (name == "o_mb" or name.startswith("sdp_")) and name not in out
            continue
        got = thin(name, out[name]).numpy()
        ref = gold[name]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = rms(got - ref)
        scale = max(rms(ref), 1e-3)
        assert err <= 2e-5 * scale + 1e-6, "%s: rms err %.3e (scale %.3e)" % (name, err, scale)
    # the north-star bar, on the waveform: <= 1e-4 RMS
    assert rms(out["o"].numpy() - gold["o"]) < 1e-4


def test_signal_ops_known_answers():
    g = load_fixture("signal_ops")
    assert np.allclose(R.hann_periodic(16), g["hann16"], atol=1e-7)
    assert np.allclose(R.pqmf_synthesis_filter(), g["pqmf_synthesis_filter"], atol=2e-7)
    y = R.istft(torch.from_numpy(g["istft_mag"]), torch.from_numpy(g["istft_phase"]))
    assert y.shape == g["istft_out"][:, 0].shape
    assert np.abs(y.numpy() - g["istft_out"][:, 0]).max() < 5e-6
    sub = torch.from_numpy(g["pqmf_in"])
    full = R.synthesis_filter_apply(R.zero_stuff(sub), torch.from_numpy(R.pqmf_synthesis_filter()))
    assert np.abs(full.numpy() - g["pqmf_out"]).max() < 5e-6


def test_voice_conversion_matches_reference():
    """`SynthesizerTrn.voice_conversion` (models.py:790-798) with the posterior noise pinned."""
    gold = load_fixture("vc_uudb_b2")
    _, cfg = config_for("uudb_ms_istft_vits_ms", int(gold["n_vocab"]))
    sd = synth.make_state_dict(cfg, int(gold["weight_seed"]))
    torch.set_num_threads(4)
    out = R.voice_conversion(sd, cfg, gold["y"], gold["y_lengths"], gold["sid_src"], gold["sid_tgt"],
                             noise=gold["noise"])
    for name in ("y_mask", "z", "z_p", "z_hat", "o_mb", "o"):
        got, ref = out[name].numpy(), gold[name]
        assert got.shape == ref.shape, name
        assert rms(got - ref) <= 2e-5 * max(rms(ref), 1e-3) + 1e-6, name


def test_call_parameters_and_decoder_entry_match_reference():
    """noise_scale > 0 (noise pinned), length_scale != 1, max_len, and `net.dec(z_chunk)`
    (models.py:729-734, 344-377) against vectors from the real reference (`params_mb_b2`)."""
    gold = load_fixture("params_mb_b2")
    _, cfg = config_for("ljs_mb_istft_vits", int(gold["n_vocab"]))
    sd = synth.make_state_dict(cfg, int(gold["weight_seed"]))
    torch.set_num_threads(4)
    out = R.infer(sd, cfg, gold["x"], gold["x_lengths"], noise=gold["noise"],
                  noise_scale=float(gold["noise_scale"]), length_scale=float(gold["length_scale"]),
                  max_len=int(gold["max_len"]))
    assert np.array_equal(thin("attn", out["attn"]).numpy(), gold["attn"])
    for name in ("y_mask", "z_p", "z", "spec", "phase", "o_mb", "o"):
        got, ref = thin(name, out[name]).numpy(), gold[name]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        assert rms(got - ref) <= 2e-5 * max(rms(ref), 1e-3) + 1e-6, name
    lo, hi = int(gold["dec_chunk_lo"]), int(gold["dec_chunk_hi"])
    with torch.no_grad():
        o, o_mb, spec, phase = R.decode(sd, cfg, torch.from_numpy(gold["z"][:, :, lo:hi]))
    for name, got in (("dec_o", o), ("dec_o_mb", o_mb), ("dec_spec", thin("spec", spec)), ("dec_phase", thin("phase", phase))):
        ref = gold[name]
        assert tuple(got.shape) == ref.shape, name
        assert rms(got.numpy() - ref) <= 2e-5 * max(rms(ref), 1e-3) + 1e-6, name
