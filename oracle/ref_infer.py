"""ORACLE — test infrastructure only.

CPU restatement (PyTorch-CPU fp32 functional ops + NumPy filter design) of the
reference's `SynthesizerTrn.infer` hot path, written from the maths of
SURVEY §8a, each function citing the reference lines it follows.  It is the
checker for the HIP path and the `cpu_baseline` ("port") leg of `bench.py`;
nothing under `mb-istft-vits_amd/` may import it.

Pinned by: `tests/golden/*.npz`, captured by importing the real reference in the
build container (`tests/golden/make_golden.py`); `tests/test_oracle_golden.py`
checks every stage boundary of this file against those vectors.  The
reference itself has no tests/fixtures for this path.

Inputs are a flat ``state_dict`` (name -> tensor, the reference's checkpoint
keys) and a `ModelConfig`-like object with the hyper-parameters.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1            # modules.py:17
WINDOW = 4                   # attentions.py:14
FLOW_LAYERS = 4              # models.py:647
N_FLOWS = 4                  # models.py:191


# --------------------------------------------------------------------------
# weight handling
# --------------------------------------------------------------------------
def fold_weight_norm(v, g):
    """w = g * v / ||v||, norm over every dim but 0 (old-style
    torch.nn.utils.weight_norm, dim=0; sites SURVEY §2a).  For ConvTranspose1d
    dim 0 is C_in, so the norm is per input channel."""
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
    return v * (g / n)


class Weights:
    """state-dict view that folds weight-norm pairs on access (once)."""

    def __init__(self, sd):
        self.sd = {k: (torch.as_tensor(v).float()) for k, v in sd.items()}
        self._folded = {}

    def w(self, prefix):
        if prefix + ".weight" in self.sd:
            return self.sd[prefix + ".weight"]
        if prefix not in self._folded:
            self._folded[prefix] = fold_weight_norm(self.sd[prefix + ".weight_v"],
                                                    self.sd[prefix + ".weight_g"])
        return self._folded[prefix]

    def b(self, prefix):
        return self.sd.get(prefix + ".bias")

    def __getitem__(self, k):
        return self.sd[k]

    def __contains__(self, k):
        return k in self.sd


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def sequence_mask(lengths, max_len):
    """commons.py:121-125 -> float mask [B, 1, T]."""
    t = torch.arange(int(max_len), dtype=lengths.dtype)
    return (t[None, :] < lengths[:, None]).unsqueeze(1).float()


def channel_layer_norm(x, gamma, beta, eps=1e-5):
    """modules.py:29-32: LayerNorm over the channel axis of [B, C, T]."""
    mean = x.mean(dim=1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma[None, :, None] + beta[None, :, None]


def conv_same(x, w, b, dilation=1):
    k = w.shape[-1]
    return F.conv1d(x, w, b, padding=(k * dilation - dilation) // 2, dilation=dilation)


# --------------------------------------------------------------------------
# text encoder (models.py:172-181, attentions.py:35-47, 138-179, 277-285)
# --------------------------------------------------------------------------
def relative_attention(q, k, v, mask_bt, emb_k, emb_v, n_heads):
    """attentions.py:148-179 restated without the pad/reshape skew tricks:
        s[i,j] = (q_i/sqrt(d)) . k_j + [|j-i|<=4] (q_i/sqrt(d)) . Ek[j-i+4]
        s      = -1e4 where mask_i*mask_j == 0
        p      = softmax_j(s)
        o_i    = sum_j p[i,j] v_j + sum_{r=-4..4} p[i,i+r] Ev[r+4]
    q,k,v: [B, C, T]; mask_bt: [B, T] (1 = valid).  Ek/Ev: [9, d] (heads share)."""
    B, C, T = q.shape
    d = C // n_heads
    qh = q.view(B, n_heads, d, T).transpose(2, 3) / math.sqrt(d)      # [B,h,T,d]
    kh = k.view(B, n_heads, d, T).transpose(2, 3)
    vh = v.view(B, n_heads, d, T).transpose(2, 3)
    scores = qh @ kh.transpose(-1, -2)                                  # [B,h,T,T]
    rel = qh @ emb_k.t()                                                # [B,h,T,9]
    idx = torch.arange(T)
    for r in range(-WINDOW, WINDOW + 1):
        i = idx[(idx + r >= 0) & (idx + r < T)]
        scores[:, :, i, i + r] += rel[:, :, i, r + WINDOW]
    pair = mask_bt[:, None, :, None] * mask_bt[:, None, None, :]
    scores = scores.masked_fill(pair == 0, -1e4)
    p = torch.softmax(scores, dim=-1)
    out = p @ vh                                                        # [B,h,T,d]
    for r in range(-WINDOW, WINDOW + 1):
        i = idx[(idx + r >= 0) & (idx + r < T)]
        out[:, :, i, :] += p[:, :, i, i + r].unsqueeze(-1) * emb_v[r + WINDOW][None, None, None, :]
    return out.transpose(2, 3).reshape(B, C, T)


def text_encoder(W, cfg, ids, lengths):
    """models.py:172-181.  Returns x [B,H,T], m_p, logs_p [B,I,T], x_mask [B,1,T]."""
    H = cfg.hidden_channels
    T = ids.shape[1]
    x = W["enc_p.emb.weight"][ids] * math.sqrt(H)            # [B,T,H]
    x = x.transpose(1, 2)
    x_mask = sequence_mask(lengths, T)
    x = x * x_mask
    m_bt = x_mask[:, 0, :]
    for i in range(cfg.n_layers):
        a = "enc_p.encoder.attn_layers.%d" % i
        q = F.conv1d(x, W.w(a + ".conv_q"), W.b(a + ".conv_q"))
        k = F.conv1d(x, W.w(a + ".conv_k"), W.b(a + ".conv_k"))
        v = F.conv1d(x, W.w(a + ".conv_v"), W.b(a + ".conv_v"))
        y = relative_attention(q, k, v, m_bt, W[a + ".emb_rel_k"][0], W[a + ".emb_rel_v"][0],
                               cfg.n_heads)
        y = F.conv1d(y, W.w(a + ".conv_o"), W.b(a + ".conv_o"))
        n1 = "enc_p.encoder.norm_layers_1.%d" % i
        x = channel_layer_norm(x + y, W[n1 + ".gamma"], W[n1 + ".beta"])
        f = "enc_p.encoder.ffn_layers.%d" % i
        ks = cfg.kernel_size
        pad = ((ks - 1) // 2, ks // 2)                       # attentions.py:296-303
        h = F.conv1d(F.pad(x * x_mask, pad), W.w(f + ".conv_1"), W.b(f + ".conv_1"))
        h = torch.relu(h)
        h = F.conv1d(F.pad(h * x_mask, pad), W.w(f + ".conv_2"), W.b(f + ".conv_2"))
        y = h * x_mask
        n2 = "enc_p.encoder.norm_layers_2.%d" % i
        x = channel_layer_norm(x + y, W[n2 + ".gamma"], W[n2 + ".beta"])
    x = x * x_mask
    stats = F.conv1d(x, W.w("enc_p.proj"), W.b("enc_p.proj")) * x_mask
    m_p, logs_p = stats[:, :cfg.inter_channels], stats[:, cfg.inter_channels:]
    return x, m_p, logs_p, x_mask


# --------------------------------------------------------------------------
# duration predictor + length regulation (models.py:123-137, 717-725)
# --------------------------------------------------------------------------
def duration_predictor(W, cfg, x, x_mask, g=None):
    if g is not None:
        x = x + F.conv1d(g, W.w("dp.cond"), W.b("dp.cond"))
    h = conv_same(x * x_mask, W.w("dp.conv_1"), W.b("dp.conv_1"))
    h = channel_layer_norm(torch.relu(h), W["dp.norm_1.gamma"], W["dp.norm_1.beta"])
    h = conv_same(h * x_mask, W.w("dp.conv_2"), W.b("dp.conv_2"))
    h = channel_layer_norm(torch.relu(h), W["dp.norm_2.gamma"], W["dp.norm_2.beta"])
    return F.conv1d(h * x_mask, W.w("dp.proj"), W.b("dp.proj")) * x_mask


# --------------------------------------------------------------------------
# StochasticDurationPredictor, reverse direction (use_sdp checkpoints)
# --------------------------------------------------------------------------
SDP_BINS = 10                # modules.py:357 (num_bins)
SDP_TAIL = 5.0               # modules.py:357 (tail_bound)
SDP_DDS_LAYERS = 3           # models.py:33,47 (n_layers of every DDSConv in the SDP)
SDP_MIN = 1e-3               # transforms.py:7-9 (min bin width / height / derivative)


def dds_conv(W, prefix, x, x_mask, g=None):
    """modules.py:98-111: per layer depth-wise conv (k, dilation k**i) -> LN -> GELU -> 1x1 ->
    LN -> GELU -> residual; mask on the way in and out."""
    if g is not None:
        x = x + g
    C = x.shape[1]
    for i in range(SDP_DDS_LAYERS):
        w = W[prefix + ".convs_sep.%d.weight" % i]
        k = w.shape[-1]
        d = k ** i
        y = F.conv1d(x * x_mask, w, W[prefix + ".convs_sep.%d.bias" % i], padding=(k * d - d) // 2,
                     dilation=d, groups=C)
        y = channel_layer_norm(y, W[prefix + ".norms_1.%d.gamma" % i], W[prefix + ".norms_1.%d.beta" % i])
        y = F.gelu(y)
        y = F.conv1d(y, W[prefix + ".convs_1x1.%d.weight" % i], W[prefix + ".convs_1x1.%d.bias" % i])
        y = channel_layer_norm(y, W[prefix + ".norms_2.%d.gamma" % i], W[prefix + ".norms_2.%d.beta" % i])
        y = F.gelu(y)
        x = x + y
    return x * x_mask


def rq_spline_inverse(y, uw, uh, ud):
    """transforms.py:55-98 (linear tails) + 100-170 (inverse branch), element-wise.
    y [...], uw/uh [..., 10], ud [..., 9].  Outside [-5, 5] the transform is the identity."""
    nb = uw.shape[-1]
    inside = (y >= -SDP_TAIL) & (y <= SDP_TAIL)
    const = float(np.log(np.exp(1 - SDP_MIN) - 1))              # transforms.py:74
    ud = F.pad(ud, (1, 1))
    ud[..., 0] = const
    ud[..., -1] = const

    def knots(u):                                               # transforms.py:118-125 / 129-136
        p = F.softmax(u, dim=-1)
        p = SDP_MIN + (1 - SDP_MIN * nb) * p
        c = F.pad(torch.cumsum(p, dim=-1), (1, 0))
        c = 2 * SDP_TAIL * c - SDP_TAIL
        c[..., 0] = -SDP_TAIL
        c[..., -1] = SDP_TAIL
        return c, c[..., 1:] - c[..., :-1]

    cw, widths = knots(uw)
    ch, heights = knots(uh)
    deriv = SDP_MIN + F.softplus(ud)
    yc = torch.where(inside, y, torch.zeros_like(y))
    edges = ch.clone()
    edges[..., -1] += 1e-6                                      # transforms.py:46-51
    idx = (torch.sum(yc[..., None] >= edges, dim=-1) - 1).clamp(0, nb - 1)[..., None]
    take = lambda t: t.gather(-1, idx)[..., 0]
    in_cw, in_w, in_ch, in_h = take(cw), take(widths), take(ch), take(heights)
    in_delta = take(heights / widths)
    d0, d1 = take(deriv), take(deriv[..., 1:])
    t = yc - in_ch
    s2 = d0 + d1 - 2 * in_delta
    a = t * s2 + in_h * (in_delta - d0)
    b = in_h * d0 - t * s2
    c = -in_delta * t
    root = (2 * c) / (-b - torch.sqrt(b * b - 4 * a * c))
    return torch.where(inside, root * in_w + in_cw, y)


def stochastic_duration_predictor(W, cfg, x, x_mask, g=None, noise=None, noise_scale_w=1.0, taps=None):
    """models.py:53-60, 89-100 (reverse=True): conditioning trunk, then the flow list
    [Flip, ConvFlow_3, Flip, ConvFlow_2, Flip, ConvFlow_1, Flip, ElementwiseAffine] applied to
    z = noise * noise_scale_w ([B, 2, T]); logw = channel 0.  `noise` replaces torch.randn
    (models.py:94); None == zeros."""
    C = x.shape[1]
    h = F.conv1d(x, W["dp.pre.weight"], W["dp.pre.bias"])
    if g is not None:
        h = h + F.conv1d(g, W["dp.cond.weight"], W["dp.cond.bias"])
    h = dds_conv(W, "dp.convs", h, x_mask)
    cond = F.conv1d(h, W["dp.proj.weight"], W["dp.proj.bias"])
    if taps is not None:
        taps["sdp_proj"] = cond
    cond = cond * x_mask
    B, _, T = x.shape
    z = torch.zeros(B, 2, T) if noise is None else torch.as_tensor(noise).float() * noise_scale_w
    for f in (7, 5, 3):                                         # dp.flows.1 is dropped (models.py:93)
        z = torch.flip(z, [1])                                  # modules.py:282
        p = "dp.flows.%d" % f                                   # ConvFlow, modules.py:377-400
        x0, x1 = z[:, :1], z[:, 1:]
        hh = F.conv1d(x0, W[p + ".pre.weight"], W[p + ".pre.bias"])
        hh = dds_conv(W, p + ".convs", hh, x_mask, g=cond)
        hh = F.conv1d(hh, W[p + ".proj.weight"], W[p + ".proj.bias"]) * x_mask
        hh = hh.reshape(B, 1, -1, T).permute(0, 1, 3, 2)         # [B, 1, T, 29]
        uw = hh[..., :SDP_BINS] / math.sqrt(C)
        uh = hh[..., SDP_BINS:2 * SDP_BINS] / math.sqrt(C)
        ud = hh[..., 2 * SDP_BINS:]
        x1 = rq_spline_inverse(x1, uw, uh, ud)
        z = torch.cat([x0, x1], 1) * x_mask
        if taps is not None:
            taps["sdp_flow_%d" % f] = z
    z = torch.flip(z, [1])
    z = (z - W["dp.flows.0.m"]) * torch.exp(-W["dp.flows.0.logs"]) * x_mask   # modules.py:304
    return z[:, :1]


def length_regulate(logw, x_mask, m_p, logs_p, length_scale=1.0, t_frames=None):
    """models.py:717-725 + commons.generate_path (commons.py:128-143).
    The attn-matmul is restated as what it is: token t repeated w_ceil[t] times."""
    w = torch.exp(logw) * x_mask * length_scale
    w_ceil = torch.ceil(w)                                    # [B,1,T]
    y_lengths = torch.clamp_min(w_ceil.sum(dim=(1, 2)), 1).long()
    Tp = int(y_lengths.max()) if t_frames is None else int(t_frames)   # t_frames: pad to a larger batch's T'
    y_mask = sequence_mask(y_lengths, Tp)
    cum = torch.cumsum(w_ceil[:, 0, :], dim=-1)               # [B,T]
    frames = torch.arange(Tp, dtype=cum.dtype)[None, :, None]  # [1,T',1]
    below = (frames < cum[:, None, :]).float()                # [B,T',T]  frame < cum[t]
    path = below - F.pad(below, (1, 0))[:, :, :-1]            # one-hot token per frame
    attn = (path * x_mask[:, 0][:, None, :] * y_mask[:, 0][:, :, None]).unsqueeze(1)  # [B,1,T',T]
    m_e = torch.matmul(attn[:, 0], m_p.transpose(1, 2)).transpose(1, 2)
    logs_e = torch.matmul(attn[:, 0], logs_p.transpose(1, 2)).transpose(1, 2)
    return w_ceil, y_lengths, y_mask, attn, m_e, logs_e


# --------------------------------------------------------------------------
# reverse flow (models.py:207-214, modules.py:148-176, 334-353)
# --------------------------------------------------------------------------
def wn_stack(W, prefix, cfg, h, mask, g=None, n_layers=FLOW_LAYERS):
    H = cfg.hidden_channels
    out = torch.zeros_like(h)
    gc = None
    if g is not None:
        gc = F.conv1d(g, W.w(prefix + ".cond_layer"), W.b(prefix + ".cond_layer"))  # [B,2H*L,1]
    for l in range(n_layers):
        a = conv_same(h, W.w(prefix + ".in_layers.%d" % l), W.b(prefix + ".in_layers.%d" % l))
        if gc is not None:
            a = a + gc[:, 2 * H * l:2 * H * (l + 1)]
        acts = torch.tanh(a[:, :H]) * torch.sigmoid(a[:, H:])     # commons.py:100-107
        rs = F.conv1d(acts, W.w(prefix + ".res_skip_layers.%d" % l),
                      W.b(prefix + ".res_skip_layers.%d" % l))
        if l < n_layers - 1:
            h = (h + rs[:, :H]) * mask
            out = out + rs[:, H:]
        else:
            out = out + rs
    return out * mask


def flow_reverse(W, cfg, z_p, y_mask, g=None, taps=None):
    half = cfg.inter_channels // 2
    x = z_p
    for f in reversed(range(N_FLOWS)):
        x = torch.flip(x, [1])                                   # modules.py:282
        p = "flow.flows.%d" % (2 * f)
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, W.w(p + ".pre"), W.b(p + ".pre")) * y_mask
        h = wn_stack(W, p + ".enc", cfg, h, y_mask, g)
        m = F.conv1d(h, W.w(p + ".post"), W.b(p + ".post")) * y_mask
        x1 = (x1 - m) * y_mask                                   # mean_only: logs = 0
        x = torch.cat([x0, x1], 1)
        if taps is not None:
            taps["flow_after_%d" % f] = x
    return x


def flow_forward(W, cfg, z, y_mask, g=None):
    """models.py:209-210: coupling layer then Flip, flows 0..3 (mean_only: logs = 0)."""
    half = cfg.inter_channels // 2
    x = z
    for f in range(N_FLOWS):
        p = "flow.flows.%d" % (2 * f)
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, W.w(p + ".pre"), W.b(p + ".pre")) * y_mask
        h = wn_stack(W, p + ".enc", cfg, h, y_mask, g)
        m = F.conv1d(h, W.w(p + ".post"), W.b(p + ".post")) * y_mask
        x = torch.cat([x0, m + x1 * y_mask], 1)                  # modules.py:345
        x = torch.flip(x, [1])
    return x


def posterior_encoder(W, cfg, y, y_lengths, g, noise):
    """models.py:239-246.  `noise` replaces torch.randn_like(m)."""
    mask = sequence_mask(y_lengths, y.shape[2])
    h = F.conv1d(y, W.w("enc_q.pre"), W.b("enc_q.pre")) * mask
    h = wn_stack(W, "enc_q.enc", cfg, h, mask, g, n_layers=16)
    stats = F.conv1d(h, W.w("enc_q.proj"), W.b("enc_q.proj")) * mask
    m, logs = stats[:, :cfg.inter_channels], stats[:, cfg.inter_channels:]
    z = (m + (noise * torch.exp(logs) if noise is not None else 0.0)) * mask
    return z, m, logs, mask


def voice_conversion(sd, cfg, y, y_lengths, sid_src, sid_tgt, noise=None):
    """models.py:790-798 -> dict(o, o_mb, y_mask, z, z_p, z_hat)."""
    W = sd if isinstance(sd, Weights) else Weights(sd)
    y = torch.as_tensor(y).float()
    y_lengths = torch.as_tensor(y_lengths).long()
    with torch.no_grad():
        g_src = W["emb_g.weight"][torch.as_tensor(sid_src).long()].unsqueeze(-1)
        g_tgt = W["emb_g.weight"][torch.as_tensor(sid_tgt).long()].unsqueeze(-1)
        z, _, _, y_mask = posterior_encoder(W, cfg, y, y_lengths, g_src,
                                            torch.as_tensor(noise) if noise is not None else None)
        z_p = flow_forward(W, cfg, z, y_mask, g_src)
        z_hat = flow_reverse(W, cfg, z_p, y_mask, g_tgt)
        o, o_mb, _, _ = decode(W, cfg, z_hat * y_mask, g_tgt)
    out = dict(o=o, y_mask=y_mask, z=z, z_p=z_p, z_hat=z_hat)
    if o_mb is not None:
        out["o_mb"] = o_mb
    return out


# --------------------------------------------------------------------------
# decoder conv stack (models.py:348-366, modules.py:213-228)
# --------------------------------------------------------------------------
def resblock1(W, prefix, x, kernel, dilations, g=None):
    if g is not None and (prefix + ".cond.weight") in W:
        x = x + F.conv1d(g, W.w(prefix + ".cond"), W.b(prefix + ".cond"))
    for m, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = conv_same(xt, W.w(prefix + ".convs1.%d" % m), W.b(prefix + ".convs1.%d" % m), d)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = conv_same(xt, W.w(prefix + ".convs2.%d" % m), W.b(prefix + ".convs2.%d" % m), 1)
        x = xt + x
    return x


def resblock2(W, prefix, x, kernel, dilations, g=None):
    """modules.py:251-262: x = conv_d(lrelu(x)) + x per dilation."""
    if g is not None and (prefix + ".cond.weight") in W:
        x = x + F.conv1d(g, W.w(prefix + ".cond"), W.b(prefix + ".cond"))
    for m, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = conv_same(xt, W.w(prefix + ".convs.%d" % m), W.b(prefix + ".convs.%d" % m), d)
        x = xt + x
    return x


def decoder_convs(W, cfg, z, g=None, taps=None):
    """z [B,I,T'] -> x_post [B, 72, 16T'+1] (pre-activation output of subband_conv_post)."""
    x = conv_same(z, W.w("dec.conv_pre"), W.b("dec.conv_pre"))
    if taps is not None:
        taps["dec_conv_pre"] = x
    nk = len(cfg.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, W.w("dec.ups.%d" % i), W.b("dec.ups.%d" % i),
                               stride=u, padding=(k - u) // 2)
        if taps is not None:
            taps["dec_up_%d" % i] = x
        xs = None
        for j in range(nk):
            rb = resblock1 if str(getattr(cfg, "resblock", "1")) == "1" else resblock2   # models.py:317
            r = rb(W, "dec.resblocks.%d" % (i * nk + j), x,
                          cfg.resblock_kernel_sizes[j], cfg.resblock_dilation_sizes[j], g)
            xs = r if xs is None else xs + r
        x = xs / nk
        if taps is not None:
            taps["dec_res_%d" % i] = x
    x = F.leaky_relu(x)                                          # slope 0.01 (models.py:363)
    x = torch.cat([x[:, :, 1:2], x], dim=2)                      # ReflectionPad1d((1,0))
    post = "dec.conv_post" if "dec.conv_post.bias" in W else "dec.subband_conv_post"   # models.py:272 / 336
    return conv_same(x, W.w(post), W.b(post))


# --------------------------------------------------------------------------
# iSTFT (stft.py:197-202 == torch.istft n_fft=16 hop=4 hann, center=True)
# --------------------------------------------------------------------------
def hann_periodic(n):
    """scipy get_window('hann', n, fftbins=True) (stft.py:187) in closed form."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)


def _idft_basis(n_fft):
    """Real inverse DFT of a one-sided spectrum:
       x[n] = (1/N)[Re X0 + (-1)^n Re X_{N/2} + 2 sum_{k=1}^{N/2-1}(Re X_k cos - Im X_k sin)]
       (Im of DC/Nyquist is ignored, as c2r does)."""
    n = np.arange(n_fft)[:, None].astype(np.float64)
    k = np.arange(n_fft // 2 + 1)[None, :].astype(np.float64)
    ang = 2.0 * np.pi * n * k / n_fft
    scale = np.full((1, n_fft // 2 + 1), 2.0)
    scale[0, 0] = 1.0
    scale[0, -1] = 1.0
    C = scale * np.cos(ang) / n_fft
    S = -scale * np.sin(ang) / n_fft
    S[:, 0] = 0.0
    S[:, -1] = 0.0
    return torch.from_numpy(C.astype(np.float32)), torch.from_numpy(S.astype(np.float32))


def istft(mag, phase, n_fft=16, hop=4):
    """mag, phase: [N, n_fft/2+1, F] -> [N, hop*(F-1)].  Windowed overlap-add
    divided by the edge-aware sum of squared windows, n_fft/2 trimmed each side."""
    N, _, Fr = mag.shape
    C, S = _idft_basis(n_fft)
    re = mag * torch.cos(phase)
    im = mag * torch.sin(phase)
    frames = torch.einsum("nk,bkf->bnf", C, re) + torch.einsum("nk,bkf->bnf", S, im)  # [N,16,F]
    win = torch.from_numpy(hann_periodic(n_fft))
    frames = frames * win[None, :, None]
    total = n_fft + hop * (Fr - 1)
    ola = torch.zeros(N, total)
    env = torch.zeros(total)
    for n in range(n_fft):
        sl = slice(n, n + hop * (Fr - 1) + 1, hop)
        ola[:, sl] += frames[:, n, :]
        env[sl] += win[n] ** 2
    half = n_fft // 2
    return ola[:, half:total - half] / env[None, half:total - half]


# --------------------------------------------------------------------------
# PQMF synthesis (pqmf.py:15-43, 53-93, 105-116)
# --------------------------------------------------------------------------
def kaiser_window(n, beta):
    """scipy.signal.windows.kaiser (pqmf.py:40) in closed form."""
    a = (n - 1) / 2.0
    x = np.arange(n, dtype=np.float64)
    return np.i0(beta * np.sqrt(np.clip(1.0 - ((x - a) / a) ** 2, 0.0, None))) / np.i0(beta)


def pqmf_synthesis_filter(subbands=4, taps=62, cutoff_ratio=0.15, beta=9.0):
    """h_syn[k, n] float32 [subbands, taps+1]."""
    n = np.arange(taps + 1, dtype=np.float64)
    centre = n - 0.5 * taps
    with np.errstate(invalid="ignore", divide="ignore"):
        proto = np.sin(np.pi * cutoff_ratio * centre) / (np.pi * centre)
    proto[taps // 2] = cutoff_ratio
    proto = proto * kaiser_window(taps + 1, beta)
    h = np.zeros((subbands, taps + 1))
    for k in range(subbands):
        h[k] = 2.0 * proto * np.cos((2 * k + 1) * (np.pi / (2 * subbands)) * (n - (taps - 1) / 2.0)
                                    - (-1) ** k * np.pi / 4.0)
    return h.astype(np.float32)


def zero_stuff(y_mb, subbands=4):
    """conv_transpose1d with the one-hot updown filter * subbands
    (pqmf.py:115 / models.py:463): up[k, 4m] = 4 y[k, m]."""
    B, K, M = y_mb.shape
    up = torch.zeros(B, K, M * subbands)
    up[:, :, ::subbands] = y_mb * subbands
    return up


def synthesis_filter_apply(up, h):
    """F.conv1d(pad(up, 31), h[None]) — cross-correlation with 63 taps (pqmf.py:116)."""
    taps = h.shape[-1] - 1
    return F.conv1d(F.pad(up, (taps // 2, taps // 2)), torch.as_tensor(h).view(1, h.shape[0], -1))


def waveform_tail(W, cfg, x_post):
    """x_post [B,72,F] -> (o, o_mb, spec, phase), models.py:366-377 / 454-467."""
    B, _, Fr = x_post.shape
    K, nb = cfg.subbands, cfg.gen_istft_n_fft // 2 + 1
    if K == 1:                                               # iSTFT_Generator (models.py:296-300)
        spec = torch.exp(x_post[:, :nb])
        phase = math.pi * torch.sin(x_post[:, nb:])
        y = istft(spec, phase, cfg.gen_istft_n_fft, cfg.gen_istft_hop_size)
        return y.unsqueeze(1), None, spec, phase
    x4 = x_post.reshape(B, K, 2 * nb, Fr)
    spec = torch.exp(x4[:, :, :nb])
    phase = math.pi * torch.sin(x4[:, :, nb:])
    y_mb = istft(spec.reshape(B * K, nb, Fr), phase.reshape(B * K, nb, Fr),
                 cfg.gen_istft_n_fft, cfg.gen_istft_hop_size).reshape(B, K, -1)
    if "dec.multistream_conv_post.weight_v" in W:            # MS: trainable filter
        up = zero_stuff(y_mb, K)
        h = W.w("dec.multistream_conv_post")[0]               # [4,63]
        return synthesis_filter_apply(up, h), up, spec, phase
    h = torch.from_numpy(pqmf_synthesis_filter(K))
    return synthesis_filter_apply(zero_stuff(y_mb, K), h), y_mb, spec, phase


def decode(sd, cfg, z, g=None, taps=None):
    """`.dec(z, g)` of the reference (models.py:344-377 / 430-467)."""
    W = sd if isinstance(sd, Weights) else Weights(sd)
    x_post = decoder_convs(W, cfg, z, g, taps)
    if taps is not None:
        taps["x_post"] = x_post
    return waveform_tail(W, cfg, x_post)


# --------------------------------------------------------------------------
# the whole path (models.py:697-737)
# --------------------------------------------------------------------------
def infer(sd, cfg, ids, lengths, sid=None, noise=None, noise_scale=0.0, length_scale=1.0,
          max_len=None, want_taps=False, t_frames=None, noise_w=None, noise_scale_w=1.0):
    """Returns a dict with every stage boundary of `SynthesizerTrn.infer`.
    `noise` replaces torch.randn_like(m_p) (models.py:729); None == zeros.  `noise_w` [B, 2, T]
    replaces the SDP's torch.randn (models.py:94) when cfg.use_sdp."""
    W = sd if isinstance(sd, Weights) else Weights(sd)
    ids = torch.as_tensor(ids).long()
    lengths = torch.as_tensor(lengths).long()
    taps = {} if want_taps else None
    with torch.no_grad():
        x, m_t, logs_t, x_mask = text_encoder(W, cfg, ids, lengths)
        g = None
        if cfg.n_speakers > 0:
            g = W["emb_g.weight"][torch.as_tensor(sid).long()].unsqueeze(-1)
        if getattr(cfg, "use_sdp", False):                       # models.py:711-714
            logw = stochastic_duration_predictor(W, cfg, x, x_mask, g, noise_w, noise_scale_w, taps)
        else:
            logw = duration_predictor(W, cfg, x, x_mask, g)
        w_ceil, y_lengths, y_mask, attn, m_p, logs_p = length_regulate(
            logw, x_mask, m_t, logs_t, length_scale, t_frames)
        if noise is None or noise_scale == 0:
            z_p = m_p + torch.zeros_like(m_p) * torch.exp(logs_p) * noise_scale
        else:
            z_p = m_p + torch.as_tensor(noise) * torch.exp(logs_p) * noise_scale
        z = flow_reverse(W, cfg, z_p, y_mask, g, taps)
        zin = (z * y_mask)[:, :, :max_len]
        o, o_mb, spec, phase = decode(W, cfg, zin, g, taps)
    out = dict(x_enc=x, m_text=m_t, logs_text=logs_t, x_mask=x_mask, logw=logw, w_ceil=w_ceil,
               y_lengths=y_lengths, y_mask=y_mask, attn=attn, m_p=m_p, logs_p=logs_p,
               z_p=z_p, z=z, o=o, spec=spec, phase=phase)
    if o_mb is not None:
        out["o_mb"] = o_mb
    if taps:
        out.update(taps)
    return out


# --------------------------------------------------------------------------
# wire-format epilogue of the service wrapper (tts_vits.py:204-217), NumPy as there
# --------------------------------------------------------------------------
def to_pcm16(audio, auto_normalize=True):
    """audio: 1-D float32 -> int16 (normalise to 0.9 peak if peak > 0.01, clip, * 32767, truncate)."""
    audio = np.asarray(audio, np.float32)
    peak = np.abs(audio).max() if audio.size else 0.0
    if auto_normalize and peak > 0.01:
        audio = (audio / peak) * 0.9
    return (np.clip(audio, -1.0, 1.0) * 32767).astype(np.int16)


def frame_pcm16(audio_int16, rate, frame_length=0.02):
    """Chunk + base64 framing of tts_vits.py:36-38,219-226 (a loop over t as there)."""
    import base64
    chunk = round(frame_length * rate)
    out, t = [], 0
    while t < len(audio_int16):
        out.append(base64.b64encode(np.asarray(audio_int16[t:t + chunk], np.int16).tobytes()).decode("utf-8"))
        t += chunk
    return out
