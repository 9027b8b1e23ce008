"""Deterministic synthetic checkpoint + synthetic LJSpeech-length batches.

No pretrained checkpoint ships with the reference (README links an external
drive), so the benchmark, the goldens and the parity tests all draw their
weights from this generator.  Each tensor is keyed by its
state-dict name, so both the reference model and this build load bit-identical
values without any weight file being committed.

Design notes (what a *usable* random checkpoint needs):
  * `flow.*.post` is zero-initialised in the reference (`modules.py:331-332`)
    which would make the flows the identity -> drawn non-zero here.
  * `dp.proj.bias = log 2.5` gives ~3 z-frames per token, i.e. LJSpeech-like
    utterance lengths (T_text=200 -> T' ~ 550-650); with `use_sdp` the same comes from
    `dp.flows.0.m`, and the zero-initialised `dp.flows.*.proj` (`modules.py:371-372`) are drawn
    non-zero so the splines are not the identity.
  * `weight_g` is ||v|| times a per-channel gain in [0.8, 1.2] so that the
    weight-norm fold w = g * v / ||v|| is actually exercised.
"""
import zlib
import numpy as np

from .spec import ModelConfig, param_shapes, DEC_MS


def _rs(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0xFFFFFFFF)


def _normal(name, seed, shape, std):
    return (_rs(name, seed).standard_normal(size=shape) * std).astype(np.float32)


def _v_std(name: str, shape, cfg: ModelConfig) -> float:
    """std of a conv weight so that activations stay O(1) through the stack."""
    if name.startswith("dec.ups."):
        cin, _, k = shape
        stride = cfg.upsample_rates[0]
        return float(1.0 / np.sqrt(cin * k / stride))
    if "multistream_conv_post" in name:
        return 0.1
    co, ci, k = shape
    fan_in = ci * k
    std = 1.0 / np.sqrt(fan_in)
    if name == "enc_q.pre.weight":
        std *= 0.3                                  # linear-spectrogram magnitudes are O(1..10)
    if name.endswith("post.weight") and name.startswith("flow."):
        std *= 0.5
    if ".cond" in name:
        std *= 0.3
    if name.startswith("dp.proj") and not cfg.use_sdp:
        std *= 0.3
    return float(std)


def make_state_dict(cfg: ModelConfig, seed: int = 1234):
    """name -> float32 ndarray for every key of `param_shapes(cfg)`."""
    shapes = param_shapes(cfg)
    sd = {}
    for name, shape in shapes.items():
        if name == "dec.updown_filter":
            f = np.zeros(shape, np.float32)
            for k in range(shape[0]):
                f[k, k, 0] = 1.0                     # models.py:421-423
            sd[name] = f
        elif name.endswith("weight_g"):
            continue                                  # after its weight_v
        elif name.endswith("weight_v"):
            v = _normal(name, seed, shape, _v_std(name, shape, cfg))
            sd[name] = v
            gname = name[:-1] + "g"
            norm = np.sqrt((v.astype(np.float64) ** 2).reshape(shape[0], -1).sum(1))
            gain = _rs(gname, seed).uniform(0.8, 1.2, size=shape[0])
            if "conv_post" in name and "multistream" not in name:
                gain = gain * 0.5                    # keep exp() of the magnitude head tame
            sd[gname] = (norm * gain).astype(np.float32).reshape(shapes[gname])
        elif name.endswith("emb.weight"):
            sd[name] = _normal(name, seed, shape, cfg.hidden_channels ** -0.5)  # models.py:161
        elif name == "emb_g.weight":
            sd[name] = _normal(name, seed, shape, 1.0)
        elif name.endswith("emb_rel_k") or name.endswith("emb_rel_v"):
            sd[name] = _normal(name, seed, shape, shape[-1] ** -0.5)            # attentions.py:123
        elif name.endswith(".gamma"):
            sd[name] = (1.0 + _normal(name, seed, shape, 0.1)).astype(np.float32)
        elif name.endswith(".beta"):
            sd[name] = _normal(name, seed, shape, 0.1)
        elif name == "dp.proj.bias" and not cfg.use_sdp:
            sd[name] = np.full(shape, np.log(2.5), np.float32)
        elif name.endswith("flows.0.m"):             # SDP: logw = (z - m) exp(-logs): ~3 frames / token
            sd[name] = np.asarray([[-0.9], [0.1]], np.float32)[:shape[0]]
        elif name.endswith("flows.0.logs"):
            sd[name] = _normal(name, seed, shape, 0.1)
        elif name.endswith(".bias"):
            sd[name] = _normal(name, seed, shape, 0.05)
        elif name.endswith(".weight"):
            w = _normal(name, seed, shape, _v_std(name, shape, cfg))
            if name in ("enc_p.proj.weight", "enc_q.proj.weight"):
                w[shape[0] // 2:] *= 0.2             # logs_p half: keep exp(logs_p) tame
            sd[name] = w
        else:  # pragma: no cover
            raise KeyError(name)
    return {k: sd[k] for k in shapes}                 # reference order


def synthetic_batch(cfg: ModelConfig, batch: int, t_text: int = 200, seed: int = 0,
                    ragged: bool = False):
    """Token ids / lengths (/ speaker ids) of SURVEY §8d: ids uniform in
    [1, n_vocab), T_text = 200 (fixed) or uniform [0.6 T, T] (ragged, 0-padded)."""
    rs = np.random.RandomState(seed)
    x = rs.randint(1, cfg.n_vocab, size=(batch, t_text)).astype(np.int64)
    if ragged:
        lens = rs.randint(int(0.6 * t_text), t_text + 1, size=(batch,)).astype(np.int64)
        lens[rs.randint(0, batch)] = t_text
        for b in range(batch):
            x[b, lens[b]:] = 0
    else:
        lens = np.full((batch,), t_text, np.int64)
    sid = rs.randint(0, max(cfg.n_speakers, 1), size=(batch,)).astype(np.int64) \
        if cfg.has_speaker else None
    return x, lens, sid
