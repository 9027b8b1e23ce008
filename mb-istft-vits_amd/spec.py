"""Model hyper-parameters and the state-dict key/shape table of the infer path.

The key names and shapes are the checkpoint contract of the reference
(`models.py:573-655` builds the modules; old-style ``weight_norm`` gives the
``*.weight_g`` / ``*.weight_v`` pairs).  Only the modules
inference entry points touch are listed (`infer`, `dec`, `voice_conversion`): the
discriminators are training-only and have no entry.  With ``use_sdp`` the
``dp.*`` block is the StochasticDurationPredictor's (`models.py:20-52`), including its
training-only ``post_*`` half, so that a reference checkpoint loads strictly.
"""
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Tuple

# decoder families, `models.py:634-644`
DEC_MB = 0   # Multiband_iSTFT_Generator  (fixed PQMF synthesis)
DEC_MS = 1   # Multistream_iSTFT_Generator (trainable synthesis filter)
DEC_SB = 2   # iSTFT_Generator (single band, ups 8x8, no filter bank)

WINDOW_SIZE = 4          # attentions.py:14 (Encoder default, never overridden)
DP_FILTER = 256          # models.py:652
DP_KERNEL = 3            # models.py:652
FLOW_KERNEL = 5          # models.py:647
FLOW_WN_LAYERS = 4       # models.py:647
FLOW_N = 4               # models.py:191
ENC_Q_LAYERS = 16        # models.py:646
PQMF_TAPS = 62           # pqmf.py:53
SDP_KERNEL = 3           # models.py:650 (StochasticDurationPredictor(hidden, 192, 3, 0.5, 4))
SDP_FLOWS = 4            # models.py:650
SDP_DDS_LAYERS = 3       # models.py:33,47
SDP_BINS = 10            # modules.py:357


@dataclass
class ModelConfig:
    n_vocab: int
    spec_channels: int = 513         # filter_length // 2 + 1: input of enc_q (voice conversion)
    inter_channels: int = 192
    hidden_channels: int = 192
    filter_channels: int = 768
    n_heads: int = 2
    n_layers: int = 6
    kernel_size: int = 3
    resblock: str = "1"
    resblock_kernel_sizes: List[int] = field(default_factory=lambda: [3, 7, 11])
    resblock_dilation_sizes: List[List[int]] = field(
        default_factory=lambda: [[1, 3, 5], [1, 3, 5], [1, 3, 5]])
    upsample_rates: List[int] = field(default_factory=lambda: [4, 4])
    upsample_initial_channel: int = 512
    upsample_kernel_sizes: List[int] = field(default_factory=lambda: [16, 16])
    gen_istft_n_fft: int = 16
    gen_istft_hop_size: int = 4
    subbands: int = 4
    n_speakers: int = 0
    gin_channels: int = 0
    decoder: int = DEC_MB
    use_sdp: bool = False            # dp = StochasticDurationPredictor (models.py:649-652)

    # -- derived -----------------------------------------------------------
    @property
    def has_speaker(self) -> bool:
        # emb_g exists iff n_speakers > 1 (models.py:654)
        return self.n_speakers > 1

    @property
    def total_upsample(self) -> int:
        u = 1
        for r in self.upsample_rates:
            u *= r
        return u

    @property
    def samples_per_frame(self) -> int:
        # z-frame -> output samples: ups * istft hop * subbands (=256 for all BASELINE configs)
        return self.total_upsample * self.gen_istft_hop_size * self.subbands

    @property
    def post_channels(self) -> int:
        return self.subbands * (self.gen_istft_n_fft + 2)

    def validate(self):
        """Reject configurations the HIP path does not implement, loudly."""
        if self.resblock not in ("1", "2"):
            raise ValueError("resblock must be '1' (ResBlock1, modules.py:187) or '2' (ResBlock2, "
                             "modules.py:237)")
        want = 3 if self.resblock == "1" else 2
        if len(self.resblock_kernel_sizes) != 3 or any(len(d) != want for d in self.resblock_dilation_sizes):
            raise ValueError("resblock '%s' needs 3 kernel sizes with %d dilations each" %
                             (self.resblock, want))
        want_rates = [8, 8] if self.decoder == DEC_SB else [4, 4]
        if self.upsample_rates != want_rates or self.upsample_kernel_sizes != [16, 16]:
            raise ValueError("decoder upsampling must be rates %r kernels [16,16] for this decoder "
                             "(every reference config of the family); got %r %r" %
                             (want_rates, self.upsample_rates, self.upsample_kernel_sizes))
        if self.gen_istft_n_fft != 16 or self.gen_istft_hop_size != 4 or \
                self.subbands != (1 if self.decoder == DEC_SB else 4):
            raise ValueError("the iSTFT kernels are built for n_fft=16 hop=4 and 4 sub-bands "
                             "(mb/ms) or 1 (istft_vits)")
        if self.hidden_channels % self.n_heads:
            raise ValueError("hidden_channels must divide by n_heads (attentions.py:104)")
        if self.inter_channels % 2:
            raise ValueError("channels should be divisible by 2 (modules.py:318)")
        if self.decoder not in (DEC_MB, DEC_MS, DEC_SB):
            raise ValueError("decoder must be mb_istft_vits, ms_istft_vits or istft_vits")
        for c in (self.hidden_channels, self.inter_channels, self.filter_channels,
                  self.upsample_initial_channel // 4):
            if c % 32:
                raise ValueError("channel counts must be multiples of 32 (got %d)" % c)
        if self.n_speakers > 1 and self.gin_channels <= 0:
            raise ValueError("n_speakers > 1 needs gin_channels > 0")


def config_from_ctor(n_vocab, spec_channels, segment_size, inter_channels, hidden_channels,
                     filter_channels, n_heads, n_layers, kernel_size, p_dropout, resblock,
                     resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                     upsample_initial_channel, upsample_kernel_sizes, gen_istft_n_fft,
                     gen_istft_hop_size, n_speakers=0, gin_channels=0, use_sdp=False,
                     ms_istft_vits=False, mb_istft_vits=False, subbands=False,
                     istft_vits=False, **kwargs) -> ModelConfig:
    """Same positional/keyword surface as `models.py:573-599`."""
    if mb_istft_vits:
        dec = DEC_MB
    elif ms_istft_vits:
        dec = DEC_MS
    elif istft_vits:
        dec = DEC_SB
    else:
        raise ValueError("Decoder Error in json file")  # models.py:644 prints this
    cfg = ModelConfig(
        n_vocab=int(n_vocab), spec_channels=int(spec_channels), inter_channels=int(inter_channels),
        hidden_channels=int(hidden_channels), filter_channels=int(filter_channels),
        n_heads=int(n_heads), n_layers=int(n_layers), kernel_size=int(kernel_size),
        resblock=str(resblock), resblock_kernel_sizes=[int(k) for k in resblock_kernel_sizes],
        resblock_dilation_sizes=[[int(d) for d in ds] for ds in resblock_dilation_sizes],
        upsample_rates=[int(u) for u in upsample_rates],
        upsample_initial_channel=int(upsample_initial_channel),
        upsample_kernel_sizes=[int(k) for k in upsample_kernel_sizes],
        gen_istft_n_fft=int(gen_istft_n_fft), gen_istft_hop_size=int(gen_istft_hop_size),
        subbands=1 if dec == DEC_SB else (int(subbands) if subbands else 4), n_speakers=int(n_speakers),
        gin_channels=int(gin_channels), decoder=dec, use_sdp=bool(use_sdp))
    cfg.validate()
    return cfg


def _sdp_shapes(s, C, gin):
    """StochasticDurationPredictor (models.py:20-52; filter_channels := in_channels, :23).
    Registration order of the reference: flows, post_pre, post_proj, post_convs, post_flows,
    pre, proj, convs, cond."""
    def dds(p):                                      # modules.py:74-96
        for grp, shape in (("convs_sep", (C, 1, SDP_KERNEL)), ("convs_1x1", (C, C, 1))):
            for i in range(SDP_DDS_LAYERS):
                s[p + "%s.%d.weight" % (grp, i)] = shape
                s[p + "%s.%d.bias" % (grp, i)] = (C,)
        for grp in ("norms_1", "norms_2"):
            for i in range(SDP_DDS_LAYERS):
                s[p + "%s.%d.gamma" % (grp, i)] = (C,)
                s[p + "%s.%d.beta" % (grp, i)] = (C,)

    def flows(p):                                    # ElementwiseAffine(2) + 4 x (ConvFlow, Flip)
        s[p + "0.m"] = (2, 1)
        s[p + "0.logs"] = (2, 1)
        for f in range(SDP_FLOWS):
            q = p + "%d." % (2 * f + 1)
            s[q + "pre.weight"] = (C, 1, 1)
            s[q + "pre.bias"] = (C,)
            dds(q + "convs.")
            s[q + "proj.weight"] = (3 * SDP_BINS - 1, C, 1)
            s[q + "proj.bias"] = (3 * SDP_BINS - 1,)

    flows("dp.flows.")
    s["dp.post_pre.weight"] = (C, 1, 1)
    s["dp.post_pre.bias"] = (C,)
    s["dp.post_proj.weight"] = (C, C, 1)
    s["dp.post_proj.bias"] = (C,)
    dds("dp.post_convs.")
    flows("dp.post_flows.")
    s["dp.pre.weight"] = (C, C, 1)
    s["dp.pre.bias"] = (C,)
    s["dp.proj.weight"] = (C, C, 1)
    s["dp.proj.bias"] = (C,)
    dds("dp.convs.")
    if gin:
        s["dp.cond.weight"] = (C, gin, 1)
        s["dp.cond.bias"] = (C,)


def param_shapes(cfg: ModelConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """state-dict key -> shape for every tensor `infer` reads (reference order)."""
    H, I, Fc = cfg.hidden_channels, cfg.inter_channels, cfg.filter_channels
    dk = H // cfg.n_heads
    gin = cfg.gin_channels   # cond layers exist whenever gin_channels != 0 (modules.py:126,209)
    s = OrderedDict()
    # --- enc_p (models.py:140-181, attentions.py:13-47) -------------------
    s["enc_p.emb.weight"] = (cfg.n_vocab, H)
    for i in range(cfg.n_layers):
        p = "enc_p.encoder.attn_layers.%d." % i
        s[p + "emb_rel_k"] = (1, 2 * WINDOW_SIZE + 1, dk)
        s[p + "emb_rel_v"] = (1, 2 * WINDOW_SIZE + 1, dk)
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            s[p + n + ".weight"] = (H, H, 1)
            s[p + n + ".bias"] = (H,)
    for i in range(cfg.n_layers):
        s["enc_p.encoder.norm_layers_1.%d.gamma" % i] = (H,)
        s["enc_p.encoder.norm_layers_1.%d.beta" % i] = (H,)
    for i in range(cfg.n_layers):
        p = "enc_p.encoder.ffn_layers.%d." % i
        s[p + "conv_1.weight"] = (Fc, H, cfg.kernel_size)
        s[p + "conv_1.bias"] = (Fc,)
        s[p + "conv_2.weight"] = (H, Fc, cfg.kernel_size)
        s[p + "conv_2.bias"] = (H,)
    for i in range(cfg.n_layers):
        s["enc_p.encoder.norm_layers_2.%d.gamma" % i] = (H,)
        s["enc_p.encoder.norm_layers_2.%d.beta" % i] = (H,)
    s["enc_p.proj.weight"] = (2 * I, H, 1)
    s["enc_p.proj.bias"] = (2 * I,)
    # --- dec (models.py:309-342 / 387-426) --------------------------------
    if cfg.decoder == DEC_MS:
        s["dec.updown_filter"] = (cfg.subbands, cfg.subbands, cfg.subbands)  # buffer
    C0 = cfg.upsample_initial_channel
    s["dec.conv_pre.bias"] = (C0,)
    s["dec.conv_pre.weight_g"] = (C0, 1, 1)
    s["dec.conv_pre.weight_v"] = (C0, I, 7)
    for i, k in enumerate(cfg.upsample_kernel_sizes):
        cin, cout = C0 >> i, C0 >> (i + 1)
        s["dec.ups.%d.bias" % i] = (cout,)
        s["dec.ups.%d.weight_g" % i] = (cin, 1, 1)      # ConvTranspose1d: dim 0 is C_in
        s["dec.ups.%d.weight_v" % i] = (cin, cout, k)
    nk = len(cfg.resblock_kernel_sizes)
    for i in range(len(cfg.upsample_rates)):
        ch = C0 >> (i + 1)
        for j, k in enumerate(cfg.resblock_kernel_sizes):
            p = "dec.resblocks.%d." % (i * nk + j)
            groups = (("convs1", 3), ("convs2", 3)) if cfg.resblock == "1" else (("convs", 2),)
            for grp, n in groups:                      # modules.py:190-206 / 240-244
                for m in range(n):
                    s[p + "%s.%d.bias" % (grp, m)] = (ch,)
                    s[p + "%s.%d.weight_g" % (grp, m)] = (ch, 1, 1)
                    s[p + "%s.%d.weight_v" % (grp, m)] = (ch, ch, k)
            if gin:
                s[p + "cond.weight"] = (ch, gin, 1)
                s[p + "cond.bias"] = (ch,)
    ch = C0 >> len(cfg.upsample_rates)
    post = "dec.conv_post" if cfg.decoder == DEC_SB else "dec.subband_conv_post"   # models.py:272 / 336
    s[post + ".bias"] = (cfg.post_channels,)
    s[post + ".weight_g"] = (cfg.post_channels, 1, 1)
    s[post + ".weight_v"] = (cfg.post_channels, ch, 7)
    if cfg.decoder == DEC_MS:
        s["dec.multistream_conv_post.weight_g"] = (1, 1, 1)
        s["dec.multistream_conv_post.weight_v"] = (1, cfg.subbands, PQMF_TAPS + 1)
    # --- enc_q (PosteriorEncoder, models.py:217-246: 16 WN layers, k5) ----
    # read by voice_conversion only, but part of every reference checkpoint
    s["enc_q.pre.weight"] = (H, cfg.spec_channels, 1)
    s["enc_q.pre.bias"] = (H,)
    for l in range(ENC_Q_LAYERS):
        s["enc_q.enc.in_layers.%d.bias" % l] = (2 * H,)
        s["enc_q.enc.in_layers.%d.weight_g" % l] = (2 * H, 1, 1)
        s["enc_q.enc.in_layers.%d.weight_v" % l] = (2 * H, H, 5)
    for l in range(ENC_Q_LAYERS):
        rs = 2 * H if l < ENC_Q_LAYERS - 1 else H
        s["enc_q.enc.res_skip_layers.%d.bias" % l] = (rs,)
        s["enc_q.enc.res_skip_layers.%d.weight_g" % l] = (rs, 1, 1)
        s["enc_q.enc.res_skip_layers.%d.weight_v" % l] = (rs, H, 1)
    if gin:
        s["enc_q.enc.cond_layer.bias"] = (2 * H * ENC_Q_LAYERS,)
        s["enc_q.enc.cond_layer.weight_g"] = (2 * H * ENC_Q_LAYERS, 1, 1)
        s["enc_q.enc.cond_layer.weight_v"] = (2 * H * ENC_Q_LAYERS, gin, 1)
    s["enc_q.proj.weight"] = (2 * I, H, 1)
    s["enc_q.proj.bias"] = (2 * I,)
    # --- flow (models.py:184-214, modules.py:111-184, 308-353) ------------
    for f in range(FLOW_N):
        p = "flow.flows.%d." % (2 * f)       # odd indices are parameter-free Flip
        s[p + "pre.weight"] = (H, I // 2, 1)
        s[p + "pre.bias"] = (H,)
        for l in range(FLOW_WN_LAYERS):
            s[p + "enc.in_layers.%d.bias" % l] = (2 * H,)
            s[p + "enc.in_layers.%d.weight_g" % l] = (2 * H, 1, 1)
            s[p + "enc.in_layers.%d.weight_v" % l] = (2 * H, H, FLOW_KERNEL)
        for l in range(FLOW_WN_LAYERS):
            rs = 2 * H if l < FLOW_WN_LAYERS - 1 else H
            s[p + "enc.res_skip_layers.%d.bias" % l] = (rs,)
            s[p + "enc.res_skip_layers.%d.weight_g" % l] = (rs, 1, 1)
            s[p + "enc.res_skip_layers.%d.weight_v" % l] = (rs, H, 1)
        if gin:
            s[p + "enc.cond_layer.bias"] = (2 * H * FLOW_WN_LAYERS,)
            s[p + "enc.cond_layer.weight_g"] = (2 * H * FLOW_WN_LAYERS, 1, 1)
            s[p + "enc.cond_layer.weight_v"] = (2 * H * FLOW_WN_LAYERS, gin, 1)
        s[p + "post.weight"] = (I // 2, H, 1)
        s[p + "post.bias"] = (I // 2,)
    if cfg.use_sdp:
        _sdp_shapes(s, H, gin)
    else:
        # --- dp (models.py:103-137) ---------------------------------------
        s["dp.conv_1.weight"] = (DP_FILTER, H, DP_KERNEL)
        s["dp.conv_1.bias"] = (DP_FILTER,)
        s["dp.norm_1.gamma"] = (DP_FILTER,)
        s["dp.norm_1.beta"] = (DP_FILTER,)
        s["dp.conv_2.weight"] = (DP_FILTER, DP_FILTER, DP_KERNEL)
        s["dp.conv_2.bias"] = (DP_FILTER,)
        s["dp.norm_2.gamma"] = (DP_FILTER,)
        s["dp.norm_2.beta"] = (DP_FILTER,)
        s["dp.proj.weight"] = (1, DP_FILTER, 1)
        s["dp.proj.bias"] = (1,)
        if gin:
            s["dp.cond.weight"] = (H, gin, 1)
            s["dp.cond.bias"] = (H,)
    if cfg.has_speaker:
        s["emb_g.weight"] = (cfg.n_speakers, cfg.gin_channels)
    return s
