"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

from mb_istft_vits_amd import spec as mspec, synth, utils as mutils

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FIXTURES = {
    "mini_b1": "ljs_mini_mb_istft_vits",
    "mb_b3": "ljs_mb_istft_vits",
    "ms_b2": "ljs_ms_istft_vits",
    "uudb_b2": "uudb_ms_istft_vits_ms",
    "mb_short": "ljs_mb_istft_vits",
    "sb_mini_b2": "ljs_mini_istft_vits",
    "rb2_mini_b2": "ljs_mini_mb_istft_vits",
    "sdp_mini_b2": "ljs_mini_mb_istft_vits",
    "sdp_uudb_b2": "uudb_ms_istft_vits_ms",
}
# model-block overrides a fixture was generated with (no reference config ships resblock "2")
OVERRIDES = {"rb2_mini_b2": {"resblock": "2", "resblock_dilation_sizes": [[1, 3], [1, 3], [1, 3]]},
             # StochasticDurationPredictor: no reference config sets use_sdp
             "sdp_mini_b2": {"use_sdp": True}, "sdp_uudb_b2": {"use_sdp": True}}
SDP_NOISE_SCALE_W = 0.8      # noise_scale_w the sdp_* fixtures were captured with


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def config_for(cfg_name, n_vocab=59, overrides=None):
    hps = mutils.get_hparams_from_file(mutils.builtin_config(cfg_name))
    for k, v in (overrides or {}).items():
        hps.model[k] = v
    cfg = mspec.config_from_ctor(n_vocab, hps.data.filter_length // 2 + 1,
                                 hps.train.segment_size // hps.data.hop_length,
                                 n_speakers=hps.data.n_speakers, **hps.model)
    return hps, cfg


def thin(name, a):
    """Same subsampling `tests/golden/make_golden.py` applied before storing."""
    if name in ("dec_conv_pre", "dec_up_0", "dec_up_1", "dec_res_0", "dec_res_1"):
        return a[:, ::16, :]
    if name in ("spec", "phase"):
        return a[..., ::5]
    if name == "attn":
        return a.sum(2)
    return a


def rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0
