"""Drop-in `models.SynthesizerTrn` for the infer path.

Same constructor signature, attribute names and return tuples as the
reference (`models.py:573-599`, `697-737`, `742-788`, `.dec` `344-377`), same
state-dict keys (so `utils.load_checkpoint` and reference checkpoints work),
but every forward computation happens in the gfx950 kernels behind
`libmbistft_vits.so`.  PyTorch is used for device memory, the current HIP
stream and the RNG only.  The training-side `forward` is out of scope and raises.
"""
import ctypes as C
import weakref

import numpy as np
import torch
from torch import nn

from . import _capi
from .spec import ModelConfig, config_from_ctor, param_shapes, DEC_MS, DEC_SB

_STAGES = ("text_encoder", "duration_predictor", "alignment_and_projection", "flow",
           "waveform_decoder")


class _Node(nn.Module):
    """Parameter container mirroring one reference sub-module (no compute)."""

    def forward(self, *a, **k):
        raise NotImplementedError(
            "this sub-module only holds parameters; the computation runs inside "
            "SynthesizerTrn.infer / .dec on the HIP path")


class _Decoder(_Node):
    """`net.dec(z, g=None)` -> (y_g_hat, y_mb_hat, spec, phase)  (models.py:344-377 / 430-467)."""

    def forward(self, x, g=None):
        return self._owner()._decode(x, g)


class _SpeakerEmbedding(_Node):
    """`net.emb_g(sid)` -> [B, gin]  (models.py:654-655, 705)."""

    def forward(self, sid):
        return self._owner()._speaker_embedding(sid)


class Timings(dict):
    """The reference's `timings` dict (seconds per stage, models.py:698-737), filled lazily from
    HIP events so that `infer` itself never blocks: the first READ of any kind (indexing, get,
    items/keys/values, iteration, copy, ==, repr, json.dumps, pickle, dict(t)) waits for the call's
    last kernel and fills in the five stage times.  Until then the stored values are NaN
    placeholders.  The handle keeps the events of its last 8 calls (`mbv_stage_times_ms_at`): a dict
    read after more than 7 later calls on the same model stays NaN."""

    def __init__(self, owner, ticket):
        super().__init__((k, float("nan")) for k in _STAGES)
        self._owner, self._ticket, self._done = owner, ticket, False

    def _resolve(self):
        if not self._done:
            self._done = True
            vals = self._owner._stage_times(self._ticket)
            self._owner = None
            for k, v in zip(_STAGES, vals):
                dict.__setitem__(self, k, v)
        return self

    def __getitem__(self, k):
        return dict.__getitem__(self._resolve(), k)

    def get(self, k, default=None):
        return dict.get(self._resolve(), k, default)

    def items(self):
        return dict.items(self._resolve())

    def keys(self):
        return dict.keys(self._resolve())

    def values(self):
        return dict.values(self._resolve())

    def __iter__(self):
        return dict.__iter__(self._resolve())

    def copy(self):
        return dict(dict.items(self._resolve()))

    def __eq__(self, other):
        if isinstance(other, Timings):
            other._resolve()
        return dict.__eq__(self._resolve(), other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __reduce__(self):
        return (dict, (self.copy(),))

    def __repr__(self):
        return dict.__repr__(self._resolve())


class SynthesizerTrn(nn.Module):
    """Synthesizer (inference path) — constructor surface of `models.py:573-599`."""

    def __init__(self, n_vocab, spec_channels, segment_size, inter_channels, hidden_channels,
                 filter_channels, n_heads, n_layers, kernel_size, p_dropout, resblock,
                 resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gen_istft_n_fft,
                 gen_istft_hop_size, n_speakers=0, gin_channels=0, use_sdp=False,
                 ms_istft_vits=False, mb_istft_vits=False, subbands=False, istft_vits=False,
                 **kwargs):
        super().__init__()
        self.cfg: ModelConfig = config_from_ctor(
            n_vocab, spec_channels, segment_size, inter_channels, hidden_channels,
            filter_channels, n_heads, n_layers, kernel_size, p_dropout, resblock,
            resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
            upsample_initial_channel, upsample_kernel_sizes, gen_istft_n_fft, gen_istft_hop_size,
            n_speakers=n_speakers, gin_channels=gin_channels, use_sdp=use_sdp,
            ms_istft_vits=ms_istft_vits, mb_istft_vits=mb_istft_vits, subbands=subbands,
            istft_vits=istft_vits)
        # attributes the reference exposes (models.py:602-624)
        self.n_vocab, self.spec_channels, self.segment_size = n_vocab, spec_channels, segment_size
        self.inter_channels, self.hidden_channels = inter_channels, hidden_channels
        self.filter_channels, self.n_heads, self.n_layers = filter_channels, n_heads, n_layers
        self.kernel_size, self.p_dropout, self.resblock = kernel_size, p_dropout, resblock
        self.resblock_kernel_sizes = resblock_kernel_sizes
        self.resblock_dilation_sizes = resblock_dilation_sizes
        self.upsample_rates, self.upsample_initial_channel = upsample_rates, upsample_initial_channel
        self.upsample_kernel_sizes = upsample_kernel_sizes
        self.n_speakers, self.gin_channels = n_speakers, gin_channels
        self.ms_istft_vits, self.mb_istft_vits, self.istft_vits = ms_istft_vits, mb_istft_vits, istft_vits
        self.use_sdp = use_sdp

        self._build_parameter_tree()
        self._handle = None
        self._handle_device = None
        self._synced_sig = None
        self._ticket = 0

    # ------------------------------------------------------------------ params
    def _build_parameter_tree(self):
        from . import synth
        init = synth.make_state_dict(self.cfg, seed=1234)    # deterministic "random init"
        roots = {"dec": _Decoder(), "emb_g": _SpeakerEmbedding()}
        for name, shape in param_shapes(self.cfg).items():
            parts = name.split(".")
            node = self
            for part in parts[:-1]:
                child = node._modules.get(part)
                if child is None:
                    if node is self and part in roots:
                        child = roots[part]
                        object.__setattr__(child, "_owner", weakref.ref(self))
                    else:
                        child = _Node()
                    node.add_module(part, child)
                node = child
            value = torch.from_numpy(np.ascontiguousarray(init[name]))
            if name == "dec.updown_filter":
                node.register_buffer(parts[-1], value)
            else:
                node.register_parameter(parts[-1], nn.Parameter(value, requires_grad=False))

    def _weights_signature(self):
        sig = []
        for t in list(self.parameters()) + list(self.buffers()):
            sig.append((t.data_ptr(), t._version))
        return tuple(sig)

    # ------------------------------------------------------------------ handle
    def _device(self):
        return next(self.parameters()).device

    def _ensure_handle(self):
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("SynthesizerTrn (MI355X path) has no CPU implementation: move the "
                               "model to a ROCm device with .to('cuda') before calling infer/dec")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        L = _capi.lib()
        if self._handle is not None and self._handle_device != idx:
            L.mbv_destroy(self._handle)
            self._handle, self._synced_sig = None, None
        if self._handle is None:
            c = _capi.MbvConfig()
            c.struct_bytes = C.sizeof(_capi.MbvConfig)
            cfg = self.cfg
            c.n_vocab, c.inter_channels, c.hidden_channels = cfg.n_vocab, cfg.inter_channels, cfg.hidden_channels
            c.filter_channels, c.n_heads, c.n_layers = cfg.filter_channels, cfg.n_heads, cfg.n_layers
            c.kernel_size, c.upsample_initial_channel = cfg.kernel_size, cfg.upsample_initial_channel
            c.spec_channels = cfg.spec_channels
            for j in range(3):
                c.resblock_kernel_sizes[j] = cfg.resblock_kernel_sizes[j]
                for q, d in enumerate(cfg.resblock_dilation_sizes[j]):
                    c.resblock_dilations[j][q] = d
            c.resblock_type = int(cfg.resblock)
            c.n_speakers, c.gin_channels = cfg.n_speakers, cfg.gin_channels
            c.decoder = int(cfg.decoder)
            c.device = idx
            c.use_sdp = int(bool(cfg.use_sdp))
            h = C.c_void_p()
            rc = L.mbv_create(C.byref(c), C.byref(h))
            if rc:
                raise _capi.MbvError("mbv_create failed: %s" % L.mbv_last_error(None).decode())
            self._handle, self._handle_device = h, idx
        sig = self._weights_signature()
        if sig != self._synced_sig:
            self._upload_weights()
            self._synced_sig = sig
        return self._handle

    def _upload_weights(self):
        L = _capi.lib()
        for name, t in self.state_dict().items():
            a = t.detach().to(device="cpu", dtype=torch.float32).contiguous().numpy()
            shape = (C.c_int64 * a.ndim)(*a.shape)
            _capi.check(self._handle, L.mbv_load_weight(self._handle, name.encode(),
                                                        a.ctypes.data_as(C.c_void_p), shape, a.ndim),
                        "mbv_load_weight(%s)" % name)
        _capi.check(self._handle, L.mbv_finalize_weights(self._handle, self._stream()),
                    "mbv_finalize_weights")

    def export_arena(self):
        """The folded, packed weight arena of this model as one flat fp32 device tensor (`mbv_export_arena`):
        what rank 0 broadcasts in a sharded run instead of the state dict (dist.broadcast_arena)."""
        h = self._ensure_handle()
        L = _capi.lib()
        n = int(L.mbv_arena_floats(h))
        if n <= 0:
            raise _capi.MbvError(L.mbv_last_error(h).decode())
        dev = self._device()
        flat = torch.empty(n, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _capi.check(h, L.mbv_export_arena(h, self._ptr(flat), n, self._stream()), "mbv_export_arena")
        return flat

    def import_arena(self, flat):
        """Take the weights from another process's `export_arena()` (same configuration, same library build):
        no state dict, no host-side weight-norm fold, no upload.  The module's own parameters keep whatever
        they held (the synthetic init) and are NOT what the kernels use afterwards; a later `load_state_dict`
        replaces the imported weights again."""
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("move the model to a ROCm device first (.to('cuda')): the arena lives on the GPU")
        flat = flat.to(device=dev, dtype=torch.float32).contiguous()
        self._synced_sig = self._weights_signature()      # _ensure_handle: nothing to upload
        h = self._ensure_handle()
        with torch.cuda.device(dev):
            try:
                _capi.check(h, _capi.lib().mbv_import_arena(h, self._ptr(flat), flat.numel(), self._stream()), "mbv_import_arena")
            except Exception:
                self._synced_sig = None
                raise

    def refresh_weights(self):
        """Force a re-fold/re-upload (only needed after in-place edits through `.data`)."""
        self._synced_sig = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self._device()).cuda_stream)

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _capi.lib().mbv_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def _check_inputs(self, x, x_lengths, sid):
        dev = self._device()
        if x.dim() != 2:
            raise ValueError("x must be [B, T] token ids")
        if x_lengths.dim() != 1 or x_lengths.shape[0] != x.shape[0]:
            raise ValueError("x_lengths must be [B]")
        x = x.to(device=dev, dtype=torch.int64).contiguous()
        x_lengths = x_lengths.to(device=dev, dtype=torch.int64).contiguous()
        if self.n_speakers > 0:
            if sid is None:
                raise ValueError("sid is required for a multi-speaker model (models.py:704-705)")
            sid = sid.to(device=dev, dtype=torch.int64).contiguous()
            if sid.shape != (x.shape[0],):
                raise ValueError("sid must be [B]")
        else:
            sid = None
        return x, x_lengths, sid

    def _stage_times(self, ticket):
        if self._handle is None or ticket[0] is not self._handle:
            return [float("nan")] * 5               # the handle was re-created (device move)
        buf = (C.c_float * 5)()
        with torch.cuda.device(self._device()):
            if _capi.lib().mbv_stage_times_ms_at(self._handle, ticket[1], C.byref(buf)):
                return [float("nan")] * 5           # more than 8 later calls: the events were reused
        return [v * 1e-3 for v in buf]              # reference reports seconds

    # ------------------------------------------------------------------ API
    _OUTPUT_NAMES = ("o", "o_mb", "spec", "phase", "attn", "y_mask", "z", "z_p", "m_p", "logs_p")

    @torch.no_grad()
    def _run(self, x, x_lengths, sid, noise_scale, length_scale, max_len, decode,
             frames_hook=None, noise_scale_w=1., noise_w=None, outputs=None, stat_reduce=None,
             prior_rows=None, trim=False):
        """One encode + synthesize pair.
          outputs      None = every tensor of the reference's 8-tuple; or a collection of names from
                       _OUTPUT_NAMES: only those are materialised (the others come back as None and
                       their stores never happen — `outputs=("o",)` is the waveform-only launch)
          stat_reduce  sharded runs: called with the device tensor [T'max, error flag] BEFORE the
                       one host read, reduces it in place over the ranks (all_reduce MAX), so that
                       every rank pads to the global T'max and every rank raises when any does
          frames_hook  host-side override of T' (tests: pad a sub-batch like its parent batch)
          prior_rows   (lo, hi, B_global): draw the prior noise for the whole global batch and use
                       rows lo:hi (ranks seeded alike then reproduce the single-process draw)
          trim         opt-in trimmed decode (see `infer`)"""
        h = self._ensure_handle()
        L = _capi.lib()
        x, x_lengths, sid = self._check_inputs(x, x_lengths, sid)
        dev, B, T = x.device, x.shape[0], x.shape[1]
        I = self.cfg.inter_channels
        if outputs is None:
            want = set(self._OUTPUT_NAMES)
        else:
            want = set(outputs)
            unknown = want - set(self._OUTPUT_NAMES)
            if unknown:
                raise ValueError("unknown output name(s) %s (known: %s)" % (sorted(unknown), ", ".join(self._OUTPUT_NAMES)))
        with torch.cuda.device(dev):
            stream = self._stream()
            y_lengths = torch.empty(B, dtype=torch.int64, device=dev)
            if self.cfg.use_sdp:
                # the reference draws on the default CPU generator and moves it over (models.py:94);
                # a sharded caller passes its block of the full-batch draw instead
                if noise_w is None:
                    noise_w = torch.randn(B, 2, T)
                if tuple(noise_w.shape) != (B, 2, T):
                    raise ValueError("noise_w must be [B, 2, T_text]")
                noise_w = noise_w.to(device=dev, dtype=torch.float32).contiguous()
            else:
                noise_w = None
            _capi.check(h, L.mbv_encode(h, self._ptr(x), self._ptr(x_lengths), self._ptr(sid), B, T,
                                        float(length_scale), self._ptr(noise_w), float(noise_scale_w),
                                        self._ptr(y_lengths), stream),
                        "mbv_encode")
            lo, hi = torch.aminmax(y_lengths)
            stat = torch.stack((hi, -lo))           # [T'max, > 0 iff an utterance was flagged -1]
            if stat_reduce is not None:
                stat_reduce(stat)
            Tp, flag = (int(v) for v in stat.tolist())      # the one host sync (commons.py:123)
            if flag > 0:                            # flagged by the kernels, no extra sync
                raise IndexError("index out of range in self (token id, x_lengths or sid outside the "
                                 "model's tables)")
            if frames_hook is not None:
                Tp = int(frames_hook(Tp))
            # the reference draws randn_like(m_p) even at noise_scale == 0 (models.py:729)
            if prior_rows is not None:
                # sharded run: the draw of the WHOLE batch, this shard's rows — also at noise_scale == 0, so that
                # the device generator advances exactly as in a single-process run of the full batch (a later
                # noisy call in the same process then still reproduces the single-process draws).  Large draws take
                # only the Philox calls that hold the shard's rows (rows_of_randn.py, verified against the full draw
                # on first use), so the cost does not grow with the world size.
                r_lo, r_hi, b_all = prior_rows
                from .rows_of_randn import randn_rows       # own rows of the full-batch draw, generator state included
                noise = randn_rows(r_lo, r_hi, b_all, (I, Tp), dev)
                noise = noise.contiguous() if float(noise_scale) != 0.0 else None
            else:
                noise = torch.randn(B, I, Tp, device=dev, dtype=torch.float32)
            f32 = dict(device=dev, dtype=torch.float32)
            out = _capi.MbvOutputs()
            t = {}
            if "attn" in want:
                t["attn"] = torch.empty(B, 1, Tp, T, **f32)
            if "y_mask" in want:
                t["y_mask"] = torch.empty(B, 1, Tp, **f32)
            for k in ("z", "z_p", "m_p", "logs_p"):
                if k in want:
                    t[k] = torch.empty(B, I, Tp, **f32)
            Td = Tp if max_len is None else max(0, min(Tp, int(max_len)))
            if trim:
                if want & {"o_mb", "spec", "phase"}:
                    raise ValueError("trim=True materialises the waveform only: pass outputs=('o',) (+ attn / y_mask / z ...)")
                if self.cfg.decoder == DEC_SB:
                    raise ValueError("trim=True is built for the multiband / multistream decoders")
            if decode and want & {"o", "o_mb", "spec", "phase"}:
                if Td <= 0:
                    raise ValueError("max_len leaves no frames to decode")
                o, o_mb, spec, phase = self._alloc_decoder_outputs(B, Td, dev, want)
                if trim and o is not None:
                    o.zero_()                         # tiles behind an utterance's end are never written
                t.update(o=o, o_mb=o_mb, spec=spec, phase=phase)
            for k, v in t.items():
                if v is not None:
                    setattr(out, k, v.data_ptr())
            self._ticket += 1
            if trim:
                _capi.check(h, L.mbv_set_option(h, b"trim", 1), "mbv_set_option")
            try:
                _capi.check(h, L.mbv_synthesize(h, Tp, self._ptr(noise), float(noise_scale),
                                                int(Td if max_len is not None else 0), C.byref(out), stream),
                            "mbv_synthesize")
            finally:
                if trim:
                    _capi.check(h, L.mbv_set_option(h, b"trim", 0), "mbv_set_option")
        timings = Timings(self, (h, int(L.mbv_ticket(h))))
        g = t.get
        return (g("o"), g("o_mb"), g("spec"), g("phase"), g("attn"), g("y_mask"),
                (g("z"), g("z_p"), g("m_p"), g("logs_p")), timings, y_lengths)

    def _alloc_decoder_outputs(self, B, Td, dev, want=None):
        f32 = dict(device=dev, dtype=torch.float32)
        spf = self.cfg.samples_per_frame
        w = (lambda k: True) if want is None else (lambda k: k in want)
        o = torch.empty(B, 1, spf * Td, **f32) if w("o") else None
        if self.cfg.decoder == DEC_SB:                         # models.py:300: (out, None, spec, phase)
            Fr = 64 * Td + 1
            return (o, None, torch.empty(B, 9, Fr, **f32) if w("spec") else None,
                    torch.empty(B, 9, Fr, **f32) if w("phase") else None)
        o_mb = None
        if w("o_mb"):
            if self.cfg.decoder == DEC_MS:
                o_mb = torch.empty(B, 4, spf * Td, **f32)     # zero-stuffed (models.py:463)
            else:
                o_mb = torch.empty(B, 4, (spf // 4) * Td, **f32)
        Fr = 16 * Td + 1
        spec = torch.empty(B, 4, 9, Fr, **f32) if w("spec") else None
        phase = torch.empty(B, 4, 9, Fr, **f32) if w("phase") else None
        return o, o_mb, spec, phase

    def infer(self, x, x_lengths, sid=None, noise_scale=1, length_scale=1, noise_scale_w=1.,
              max_len=None, outputs=None, trim=False):
        """-> (o, o_mb, spec, phase, attn, y_mask, (z, z_p, m_p, logs_p), timings)  (models.py:737)

        `outputs` (extension, default None = the reference's full tuple): names of the tensors to
        materialise; the rest of the tuple is None.  A caller that only takes `[0]`
        (tts_vits.py:134-137, synthesis_module.py:178-189) passes `outputs=("o",)` and gets the
        waveform-only launch of the fused iSTFT+PQMF stage (no spec / phase / o_mb / attn stores).

        `trim` (extension, default False; needs `outputs` without o_mb / spec / phase): opt-in trimmed decode for
        ragged batches — per utterance the decoder only computes what its valid 256 * y_lengths[b] samples depend
        on (frames below y_lengths[b] + 32).  Those samples are bitwise the default's; the padded region of `o`,
        which the reference's unmasked decoder fills with defined values, comes back as zeros."""
        r = self._run(x, x_lengths, sid, noise_scale, length_scale, max_len, decode=True,
                      noise_scale_w=noise_scale_w, outputs=outputs, trim=trim)
        return r[:8]

    def infer_z_only(self, x, x_lengths, sid=None, noise_scale=1, length_scale=1, noise_scale_w=1.,
                     max_len=None):
        """-> (attn, y_mask, (z, z_p, m_p, logs_p), timings)  (models.py:742-788)"""
        r = self._run(x, x_lengths, sid, noise_scale, length_scale, None, decode=False,
                      noise_scale_w=noise_scale_w)
        return r[4], r[5], r[6], r[7]

    def infer_with_lengths(self, x, x_lengths, sid=None, noise_scale=1, length_scale=1,
                           max_len=None, noise_scale_w=1., outputs=None, trim=False):
        """`infer` plus the per-utterance frame counts y_lengths [B] (int64) — what a batched
        caller needs to trim the padded waveforms (valid samples = 256 * y_lengths)."""
        r = self._run(x, x_lengths, sid, noise_scale, length_scale, max_len, decode=True,
                      noise_scale_w=noise_scale_w, outputs=outputs, trim=trim)
        return r[:8], r[8]

    @torch.no_grad()
    def _decode(self, z, g=None):
        h = self._ensure_handle()
        dev = self._device()
        if z.dim() != 3 or z.shape[1] != self.cfg.inter_channels:
            raise ValueError("z must be [B, %d, T']" % self.cfg.inter_channels)
        z = z.to(device=dev, dtype=torch.float32).contiguous()
        B, _, Tp = z.shape
        if g is not None:
            if self.cfg.gin_channels == 0:
                g = None
            else:
                g = g.to(device=dev, dtype=torch.float32).reshape(B, self.cfg.gin_channels).contiguous()
        with torch.cuda.device(dev):
            o, o_mb, spec, phase = self._alloc_decoder_outputs(B, Tp, dev)
            out = _capi.MbvOutputs()
            out.o, out.spec, out.phase = o.data_ptr(), spec.data_ptr(), phase.data_ptr()
            out.o_mb = o_mb.data_ptr() if o_mb is not None else None
            _capi.check(h, _capi.lib().mbv_decode(h, self._ptr(z), self._ptr(g), B, Tp, C.byref(out),
                                                  self._stream()), "mbv_decode")
        return o, o_mb, spec, phase

    @torch.no_grad()
    def _decode_into(self, z, g, outs):
        """`net.dec` on a block of rows, writing into caller-owned (o, o_mb, spec, phase) — any may be None.
        dist.sharded_infer(overlap="halves") decodes a shard in two pieces with it."""
        h = self._ensure_handle()
        dev = self._device()
        z = z.to(device=dev, dtype=torch.float32).contiguous()
        B, _, Tp = z.shape
        if g is not None:
            g = g.to(device=dev, dtype=torch.float32).reshape(B, self.cfg.gin_channels).contiguous()
        out = _capi.MbvOutputs()
        for name, t in zip(("o", "o_mb", "spec", "phase"), outs):
            if t is not None:
                if not t.is_contiguous() or t.shape[0] != B:
                    raise ValueError("%s: a contiguous block of %d rows expected" % (name, B))
                setattr(out, name, t.data_ptr())
        with torch.cuda.device(dev):
            _capi.check(h, _capi.lib().mbv_decode(h, self._ptr(z), self._ptr(g), B, Tp, C.byref(out),
                                                  self._stream()), "mbv_decode")

    @torch.no_grad()
    def istft_finalize(self, spec, phase):
        """(spec, phase) -> waveform [B, 1, samples] with the model's synthesis bank: the last step
        of the reference's chunked decoding (`istft_finalize` in inferz_test.ipynb cell 6; the
        cross-fade of the chunks' spectrograms stays in the caller).  Accepts the complex
        spectrogram too (`spec * exp(1j * phase)`), as the notebook passes it."""
        h = self._ensure_handle()
        dev = self._device()
        if phase is None:
            if not torch.is_complex(spec):
                raise ValueError("istft_finalize(spec, phase): phase missing and spec is not complex")
            spec, phase = torch.abs(spec), torch.angle(spec)
        spec = spec.to(device=dev, dtype=torch.float32).contiguous()
        phase = phase.to(device=dev, dtype=torch.float32).contiguous()
        if spec.shape != phase.shape:
            raise ValueError("spec and phase must have the same shape")
        sb = self.cfg.decoder == DEC_SB
        if (spec.dim() != (3 if sb else 4)) or spec.shape[-2] != 9 or (not sb and spec.shape[1] != 4):
            raise ValueError("spec must be [B, 9, F]" if sb else "spec must be [B, 4, 9, F]")
        B, Fr = spec.shape[0], spec.shape[-1]
        n = (4 if sb else 16) * (Fr - 1)
        o = torch.empty(B, 1, n, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _capi.check(h, _capi.lib().mbv_istft_finalize(h, self._ptr(spec), self._ptr(phase), B, Fr,
                                                          self._ptr(o), None, self._stream()),
                        "mbv_istft_finalize")
        return o

    @torch.no_grad()
    def to_pcm16(self, wave, y_lengths=None, auto_normalize=True):
        """Waveform [B, 1, n] -> int16 PCM [B, n] on the GPU: the normalise / clip / *32767 /
        astype(int16) sequence of the service wrapper (tts_vits.py:204-217), per utterance over
        its valid 256 * y_lengths samples (rest zero).  Bit-exact with the NumPy code."""
        h = self._ensure_handle()
        dev = self._device()
        wave = wave.to(device=dev, dtype=torch.float32).contiguous()
        B, n = wave.shape[0], wave.shape[-1]
        if y_lengths is not None:
            y_lengths = y_lengths.to(device=dev, dtype=torch.int64).contiguous()
        pcm = torch.empty(B, n, device=dev, dtype=torch.int16)
        with torch.cuda.device(dev):
            _capi.check(h, _capi.lib().mbv_pcm16(h, self._ptr(wave), self._ptr(y_lengths), B, n,
                                                 int(bool(auto_normalize)), self._ptr(pcm), self._stream()),
                        "mbv_pcm16")
        return pcm

    @torch.no_grad()
    def _speaker_embedding(self, sid):
        h = self._ensure_handle()
        dev = self._device()
        sid = sid.to(device=dev, dtype=torch.int64).contiguous()
        flat = sid.reshape(-1)
        if flat.numel() and (int(flat.min()) < 0 or int(flat.max()) >= self.n_speakers):
            raise IndexError("index out of range in self")       # nn.Embedding's message
        out = torch.empty(flat.numel(), self.cfg.gin_channels, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _capi.check(h, _capi.lib().mbv_speaker_embedding(h, self._ptr(flat), flat.numel(),
                                                             self._ptr(out), self._stream()),
                        "mbv_speaker_embedding")
        return out.reshape(*sid.shape, self.cfg.gin_channels)

    def kernel_times_ms(self):
        """(decoder conv stack ms, fused iSTFT+PQMF launch ms) of the last infer / dec call."""
        buf = (C.c_float * 2)()
        _capi.check(self._handle, _capi.lib().mbv_kernel_times_ms(self._handle, C.byref(buf)),
                    "mbv_kernel_times_ms")
        return float(buf[0]), float(buf[1])

    def set_option(self, name, value):
        """Run-time options of the library (`mbv_set_option`): "splitk" (low-latency split-K for
        small launches), "istft_exact", "xpost_chunk_bytes", "wn_fused", "dec_streams", "conv_bf16" (0 / 3: opt-in split-bf16
        arithmetic in the large conv launches, see include/mbistft_vits.h).  Kept across weight refreshes; a
        handle re-created on another device starts from the defaults again."""
        h = self._ensure_handle()
        with torch.cuda.device(self._device()):
            _capi.check(h, _capi.lib().mbv_set_option(h, name.encode(), int(value)), "mbv_set_option")

    def read_stage(self, name):
        """Internal stage tensor of the last call as a flat fp32 tensor (tests/debugging)."""
        h = self._ensure_handle()
        L = _capi.lib()
        n = L.mbv_read_stage(h, name.encode(), None, 0, self._stream())
        if n < 0:
            raise _capi.MbvError(L.mbv_last_error(h).decode())
        t = torch.empty(n, device=self._device(), dtype=torch.float32)
        if L.mbv_read_stage(h, name.encode(), self._ptr(t), n, self._stream()) < 0:
            raise _capi.MbvError(L.mbv_last_error(h).decode())
        return t

    # ------------------------------------------------------------------ out of scope
    def forward(self, *a, **k):
        raise NotImplementedError("training forward (models.py:657-695) is outside the inference "
                                  "hot path this package implements")

    @torch.no_grad()
    def voice_conversion(self, y, y_lengths, sid_src, sid_tgt):
        """-> (o_hat, o_hat_mb, y_mask, (z, z_p, z_hat))  (models.py:790-798)."""
        if not self.n_speakers > 0:
            raise AssertionError("n_speakers have to be larger than 0.")      # models.py:791
        h = self._ensure_handle()
        dev = self._device()
        cfg = self.cfg
        if y.dim() != 3 or y.shape[1] != cfg.spec_channels:
            raise ValueError("y must be [B, %d, T] (linear spectrogram)" % cfg.spec_channels)
        y = y.to(device=dev, dtype=torch.float32).contiguous()
        B, _, T = y.shape
        y_lengths = y_lengths.to(device=dev, dtype=torch.int64).contiguous()
        sid_src = sid_src.to(device=dev, dtype=torch.int64).contiguous()
        sid_tgt = sid_tgt.to(device=dev, dtype=torch.int64).contiguous()
        I = cfg.inter_channels
        f32 = dict(device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            noise = torch.randn(B, I, T, **f32)                  # randn_like(m) of models.py:245
            o, o_mb, spec, phase = self._alloc_decoder_outputs(B, T, dev)
            y_mask = torch.empty(B, 1, T, **f32)
            z, z_p, z_hat = (torch.empty(B, I, T, **f32) for _ in range(3))
            status = torch.empty(B, dtype=torch.int32, device=dev)
            out = _capi.MbvOutputs()
            out.o, out.spec, out.phase = o.data_ptr(), spec.data_ptr(), phase.data_ptr()
            out.o_mb = o_mb.data_ptr() if o_mb is not None else None
            out.y_mask, out.z, out.z_p, out.m_p = y_mask.data_ptr(), z.data_ptr(), z_p.data_ptr(), z_hat.data_ptr()
            _capi.check(h, _capi.lib().mbv_voice_conversion(
                h, self._ptr(y), self._ptr(y_lengths), self._ptr(sid_src), self._ptr(sid_tgt), B, T,
                self._ptr(noise), C.byref(out), self._ptr(status), self._stream()), "mbv_voice_conversion")
            if bool(status.any()):
                raise IndexError("index out of range in self (y_lengths or speaker id)")
        return o, o_mb, y_mask, (z, z_p, z_hat)
