"""MI355X-native inference path for MB-iSTFT-VITS `SynthesizerTrn.infer`.

Drop-in surface (mirrors the reference's `models.py` / `utils.py` for this path):
    from mb_istft_vits_amd import models, utils
    hps = utils.get_hparams_from_file(cfg)
    net = models.SynthesizerTrn(n_vocab, ..., **hps.model).to("cuda").eval()
    o, o_mb, spec, phase, attn, y_mask, (z, z_p, m_p, logs_p), timings = net.infer(x, x_lengths)

All arithmetic runs in hand-written gfx950 HIP kernels behind the C-ABI
library `csrc/libmbistft_vits.so` (`include/mbistft_vits.h`); there is no
CPU or eager-PyTorch fallback — a missing library or GPU raises.
"""
__version__ = "0.1.0"
