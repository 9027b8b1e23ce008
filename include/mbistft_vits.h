/*
 * mbistft_vits.h — C-ABI of the MI355X-native MB-iSTFT-VITS inference path.
 *
 * The reference has no FFI/plugin layer: its boundary for this path is the
 * Python class surface `models.SynthesizerTrn` (+ `utils.HParams`).  This
 * header is what a reference-side binding (ctypes) binds
 * to replace the body of each method; every entry point names the reference
 * interface it replaces (file:line under the reference repo).
 *
 * Conventions
 *   - plain C, no torch types: raw device pointers + sizes; all tensors are
 *     fp32, contiguous, [B, C, time] with time fastest (ids/lengths int64).
 *   - every call returns 0 on success, non-zero on failure; the message is
 *     available from mbv_last_error().  No C++ exception crosses the ABI.
 *   - kernels are enqueued on the caller's `stream` (a hipStream_t passed as
 *     void*; NULL = default stream).  Calls never synchronise the device
 *     except where stated.
 *   - a handle is not re-entrant (reference callers are single-threaded
 *     w.r.t. the model: tts_vits.py:145,181); handles on different devices are
 *     independent.
 *   - the library owns only its folded-weight arena and scratch workspace;
 *     all inputs/outputs are borrowed for the duration of the call.
 */
#ifndef MBISTFT_VITS_H
#define MBISTFT_VITS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBV_ABI_VERSION 3   /* 3 (r03): + arena export / import, call tickets, option "trim", mbv_op_rel_attention; structs unchanged */

#define MBV_DEC_MULTIBAND   0   /* models.py:309 Multiband_iSTFT_Generator (fixed PQMF)        */
#define MBV_DEC_MULTISTREAM 1   /* models.py:387 Multistream_iSTFT_Generator (trainable filter)*/
#define MBV_DEC_SINGLEBAND  2   /* models.py:248 iSTFT_Generator (ups 8x8, no filter bank)      */

typedef struct mbv_model mbv_model;   /* opaque */

/* Hyper-parameters: the ctor arguments of models.py:573-599 that shape the
 * infer path (p_dropout, segment_size, n_layers_q … do not).  `struct_bytes` must be sizeof(mbv_config). */
typedef struct mbv_config {
  int32_t struct_bytes;
  int32_t n_vocab;
  int32_t inter_channels;            /* 192 */
  int32_t hidden_channels;           /* 192 (mini: 96) */
  int32_t filter_channels;           /* 768 */
  int32_t n_heads;                   /* 2 */
  int32_t n_layers;                  /* 6 (mini: 3) */
  int32_t kernel_size;               /* FFN kernel, 3 */
  int32_t upsample_initial_channel;  /* 512 (mini: 256) */
  int32_t spec_channels;             /* filter_length / 2 + 1 = 513: input of enc_q (voice conversion) */
  int32_t resblock_kernel_sizes[3];  /* 3,7,11 */
  int32_t resblock_dilations[3][3];  /* 1,3,5 each (ResBlock2: first two used) */
  int32_t resblock_type;             /* 1 = ResBlock1 (modules.py:187), 2 = ResBlock2 (modules.py:237) */
  int32_t n_speakers;                /* 0 = single speaker */
  int32_t gin_channels;              /* 0 or 256 */
  int32_t decoder;                   /* MBV_DEC_* */
  int32_t device;                    /* HIP device ordinal */
  int32_t use_sdp;                   /* 1: dp is the StochasticDurationPredictor (models.py:649-650) */
} mbv_config;

/* Output bundle of phase B / decode.  Any pointer may be NULL to skip
 * materialising that tensor (the reference always returns all of them:
 * models.py:737).  T' = frames, F = 16 T' + 1, all device pointers. */
typedef struct mbv_outputs {
  float *o;        /* [B, 1, 256 T']                         waveform           */
  float *o_mb;     /* MB: [B, 4, 64 T'];  MS: [B, 4, 256 T'] (zero-stuffed); SB: unused */
  float *spec;     /* [B, 4, 9, F]      (SB: [B, 9, F] with F = 64 T' + 1)      */
  float *phase;    /* same shape as spec                                         */
  float *attn;     /* [B, 1, T', T]                          (synthesize only)   */
  float *y_mask;   /* [B, 1, T']                             (synthesize only)   */
  float *z;        /* [B, 192, T']                           (synthesize only)   */
  float *z_p;      /* [B, 192, T']                                               */
  float *m_p;      /* [B, 192, T']                                               */
  float *logs_p;   /* [B, 192, T']                                               */
} mbv_outputs;

/* ---- life cycle ---------------------------------------------------------
 * replaces SynthesizerTrn.__init__ (models.py:573-655). */
int  mbv_abi_version(void);
int  mbv_create(const mbv_config *cfg, mbv_model **out);
void mbv_destroy(mbv_model *m);
/* Message of the last failed call on `m` (or of the last failed mbv_create
 * when m == NULL).  Valid until the next call. */
const char *mbv_last_error(const mbv_model *m);

/* ---- weights --------------------------------------------------------------
 * replaces nn.Module.load_state_dict as used by utils.load_checkpoint
 * (utils.py:22-47).  `name` is the reference state-dict key
 * ("dec.ups.0.weight_v", …); `data` is a HOST pointer to fp32 values of
 * `shape[0..ndim)`.  Keys no module of models.SynthesizerTrn owns (discriminators, optimizer
 * state) are rejected; enc_q.* (voice conversion) and, with use_sdp, the SDP's training-only
 * dp.post_* half are accepted so that a reference checkpoint loads strictly.
 * mbv_finalize_weights folds weight-norm (w = g v/||v||),
 * packs every conv for the kernels, uploads once, and may be called again
 * after further mbv_load_weight calls.  It synchronises `stream`. */
int mbv_load_weight(mbv_model *m, const char *name, const float *data,
                    const int64_t *shape, int ndim);
int mbv_finalize_weights(mbv_model *m, void *stream);

/* ---- the folded weight arena across processes (no reference counterpart; SURVEY §8e: "RCCL broadcast of the
 * folded weight arena from rank 0").  One rank loads the checkpoint and finalizes; the others receive the arena
 * over the collective of their choice (device buffers) and import it: its layout is a function of mbv_config
 * alone, so no state dict, no host-side weight-norm fold and no host->device upload happens on the receivers.
 *   mbv_arena_floats   size of the finalized arena in floats (-1: not finalized)
 *   mbv_export_arena   device-to-device copy of it into dst (capacity in floats)
 *   mbv_import_arena   lay the arena out and fill it from src (device); n_floats must equal the exporter's
 *                      mbv_arena_floats (same configuration, same library build), else the call fails */
int64_t mbv_arena_floats(mbv_model *m);
int mbv_export_arena(mbv_model *m, float *dst, int64_t capacity, void *stream);
int mbv_import_arena(mbv_model *m, const float *src, int64_t n_floats, void *stream);
/* Number of state-dict keys still missing before finalize can succeed;
 * writes up to `cap` bytes of a comma-separated list into `buf` if non-NULL. */
int mbv_missing_weights(mbv_model *m, char *buf, size_t cap);

/* ---- phase A: text encoder + duration predictor + durations ---------------
 * replaces models.py:701-719 (enc_p, emb_g, dp, exp/ceil/sum).
 *   ids      int64 [B, T]   device     token ids
 *   lengths  int64 [B]      device     valid tokens per utterance
 *   sid      int64 [B]      device     speaker ids, NULL iff n_speakers == 0
 *   y_lengths_out int64 [B] device     frames per utterance (clamped >= 1); -1 marks an utterance
 *                                      with a token id, length or speaker id out of range (the
 *                                      reference's nn.Embedding raises IndexError there)
 *   noise_w  fp32 [B, 2, T]  device     use_sdp only: the standard-normal draws of models.py:94
 *                                      (NULL == zeros); scaled by noise_scale_w inside.  Ignored
 *                                      by the deterministic DurationPredictor.
 * The caller reads max(y_lengths) back (the one host sync of the path,
 * mirroring commons.py:123) and passes it to mbv_synthesize. */
int mbv_encode(mbv_model *m, const int64_t *ids, const int64_t *lengths, const int64_t *sid,
               int B, int T, float length_scale, const float *noise_w, float noise_scale_w,
               int64_t *y_lengths_out, void *stream);

/* ---- phase B: length regulation + prior + reverse flow + decoder ----------
 * replaces models.py:720-734.
 *   t_frames  T' = max(y_lengths) as read back by the caller
 *   noise     fp32 [B, 192, T'] standard-normal draws (models.py:729), or NULL
 *             (== noise_scale 0)
 *   max_len   decoder input is truncated to this many frames (<=0: none)
 * Output shapes use T'_dec = min(T', max_len) for o/o_mb/spec/phase. */
int mbv_synthesize(mbv_model *m, int t_frames, const float *noise, float noise_scale,
                   int max_len, const mbv_outputs *outs, void *stream);

/* ---- decoder only ----------------------------------------------------------
 * replaces `net.dec(z, g)` (models.py:344-377 / 430-467; callers
 * synthesis_module.py:160, chunked decoding notebooks).
 *   z  fp32 [B, 192, T']   g  fp32 [B, gin, 1] or NULL */
int mbv_decode(mbv_model *m, const float *z, const float *g, int B, int t_frames,
               const mbv_outputs *outs, void *stream);

/* speaker embedding lookup: replaces `net.emb_g(sid)` (models.py:705).
 * out fp32 [B, gin] */
int mbv_speaker_embedding(mbv_model *m, const int64_t *sid, int B, float *out, void *stream);

/* ---- run-time options (no reference counterpart) ------------------------------
 *   "splitk"       1: split the input-channel loop of conv launches that leave most of the chip
 *                  idle over several workgroups (single-utterance latency: ljs_mb batch 1
 *                  10.3 -> 6.5 ms).  Deterministic, within fp32 rounding of the default; a row is
 *                  then no longer bitwise independent of the batch it is computed in.  Default 0
 *                  (or the MBV_CONV_SPLITK environment variable at mbv_create time).
 *   "istft_exact"  1: libm transcendentals in the fused iSTFT kernel (default 0 / MBV_ISTFT_EXACT).
 *   "xpost_chunk_bytes"  the fused iSTFT kernels address their input with 32-bit byte offsets, so a
 *                  batch whose x_post ([B, 72, F] fp32) would reach 2 GiB runs subband_conv_post +
 *                  iSTFT in sub-batches (same T', bitwise the unsplit result).  This option lowers
 *                  the cap (bytes; 0 = 2 GiB - 1) — tests use it to take the split path at small sizes.
 *   "dec_streams"  1 (default; MBV_DEC_STREAMS): when one ResBlock conv of a decoder stage cannot fill the
 *                  chip (single utterances, small batches) the stage's three ResBlocks (models.py:353-359)
 *                  run on three internal streams forked from / joined to `stream`; bitwise the result of
 *                  the one-stream schedule (0).
 *   "trim"         0 (default).  1: OPT-IN trimmed decode for ragged batches whose caller takes only the waveform
 *                  and cuts it by y_lengths (tts_vits.py:134-137 takes [0][0,0] of one utterance): in mbv_synthesize
 *                  the decoder computes, per utterance, only the tiles that hold frames below y_lengths[b] + 32
 *                  (its one-sided receptive field is 25 z-frames) and the fused iSTFT stops at 256 y_lengths[b]
 *                  samples.  Valid samples are bitwise those of the default; the padded region of `o` — defined
 *                  output of the reference, computed by the default — is left as the caller allocated it
 *                  (the Python shim zero-fills it).  Requires o_mb / spec / phase NULL; multiband / multistream
 *                  decoders, not the low-latency mode.  Never the headline configuration.
 *   "conv_bf16"    0 (default; MBV_CONV_BF16): every contraction in exact fp32.  3: OPT-IN split-bf16
 *                  arithmetic in the large conv launches (the decoder's ResBlock convs of a batch): each
 *                  fp32 operand is split into bf16(x) and bf16(x - bf16(x)), the three leading products
 *                  are accumulated in fp32 on the bf16 matrix instruction.  Not IEEE fp32 multiplication
 *                  (relative error of a product ~2^-16): ~1.9x faster decoder stage, waveform within 3e-6
 *                  RMS of the exact mode at batch 64 (bar 1e-4).  Any other value is refused. */
int mbv_set_option(mbv_model *m, const char *name, int value);

/* ---- stage timers -----------------------------------------------------------
 * replaces the `timings` dict (models.py:698-737): milliseconds of the five
 * stages of the last encode+synthesize pair, from HIP events on `stream`:
 * [text_encoder, duration_predictor, alignment_and_projection, flow,
 * waveform_decoder].  Synchronises on the recorded events. */
int mbv_stage_times_ms(mbv_model *m, float out[5]);
/* The same for an earlier call of this handle: mbv_ticket() after mbv_encode names the call (1, 2, ...); the stage
 * events of the last 8 calls are kept, so a caller that reads its timings late (the reference's `timings` dict is
 * often never read) still gets them after newer calls have started.  Fails for a call older than that. */
int64_t mbv_ticket(mbv_model *m);
int mbv_stage_times_ms_at(mbv_model *m, int64_t ticket, float out[5]);

/* Kernel-level timers of the last mbv_synthesize / mbv_decode (HIP events on the
 * launch stream, used by bench.py for the roofline lines):
 *   out[0] = decoder conv stack (conv_pre .. subband_conv_post), ms
 *   out[1] = the single fused iSTFT+PQMF launch, ms
 * Synchronises on the recorded events. */
int mbv_kernel_times_ms(mbv_model *m, float out[2]);

/* ---- stand-alone signal stage ------------------------------------------------
 * The fused iSTFT + PQMF kernel on its own: replaces TorchSTFT.inverse
 * (stft.py:197-202) + PQMF.synthesis (pqmf.py:105-116) or the MS tail
 * (models.py:463-465), including exp / pi*sin of models.py:368-369.
 *   x_post  fp32 [B, 72, F]  F = 16 T' + 1 (output of subband_conv_post)
 *   filter  fp32 [4, 63] device synthesis filter, NULL = the PQMF design
 *   multistream  bit 0: o_mb is the zero-stuffed [B,4,256T'] tensor (MS decoder);
 *                bit 1: x_post is in the library's internal units (log-magnitude rows times
 *                log2 e, phase rows divided by 2 pi), as the decoder stack produces it
 * Does not need a model handle's weights; `m` provides device + scratch. */
int mbv_istft_pqmf(mbv_model *m, const float *x_post, int B, int t_frames, const float *filter,
                   int multistream, float *o, float *o_mb, float *spec, float *phase,
                   void *stream);

/* ---- voice conversion -------------------------------------------------------------
 * replaces SynthesizerTrn.voice_conversion (models.py:790-798): posterior encoder on the source
 * spectrogram, forward flow with the source speaker, reverse flow + decoder with the target.
 *   y          fp32 [B, spec_channels, T] linear spectrogram     y_lengths int64 [B]
 *   sid_src, sid_tgt  int64 [B]
 *   noise      fp32 [B, 192, T] standard-normal draws of PosteriorEncoder (models.py:245), or
 *              NULL for the deterministic z = m_q
 *   outs       o, o_mb, spec, phase as in mbv_decode (T' = T); y_mask [B,1,T];
 *              z -> z (posterior sample), z_p -> z_p (source-normalised), m_p -> z_hat
 *   status     int32 [B] device, optional: non-zero where y_lengths / sid were out of range */
int mbv_voice_conversion(mbv_model *m, const float *y, const int64_t *y_lengths,
                         const int64_t *sid_src, const int64_t *sid_tgt, int B, int T,
                         const float *noise, const mbv_outputs *outs, int32_t *status, void *stream);

/* ---- spectrogram -> waveform ("istft_finalize") -------------------------------
 * The last step of the reference's chunked decoding (inferz_test.ipynb cells 6-7,
 * `istft_finalize`; intent of synthesis_module.py:306-353): chunks of z go through
 * mbv_decode, the caller cross-fades the returned (spec, phase) along time, and this
 * entry turns the stitched spectrogram into audio with the MODEL's synthesis bank
 * (PQMF / trained multistream filter / none for the single-band decoder).
 *   spec, phase  fp32 [B, 4, 9, F] (single band: [B, 9, F]), phase in radians
 *   frames       F; must be 16 n + 1 for mb / ms (F - 1 sub-band hops = whole z-frames)
 *   o            fp32 [B, 1, 16 (F - 1)]  (single band: [B, 1, 4 (F - 1)])
 *   o_mb         optional, as in mbv_outputs */
int mbv_istft_finalize(mbv_model *m, const float *spec, const float *phase, int B, int frames,
                       float *o, float *o_mb, void *stream);

/* ---- wire-format epilogue -----------------------------------------------------
 * replaces the NumPy post-processing of the service wrapper (tts_vits.py:204-217):
 * per-utterance peak normalisation to 0.9 (if auto_normalize and peak > 0.01), clip to
 * [-1, 1], * 32767, truncation to int16.  Bit-exact with the reference's fp32 NumPy.
 *   wave        fp32 [B, 1, stride] device
 *   y_lengths   int64 [B] device frames per utterance (valid samples = 256 * y_lengths),
 *               or NULL = every row is `stride` valid samples
 *   pcm         int16 [B, stride] device; samples past the valid length are 0 */
int mbv_pcm16(mbv_model *m, const float *wave, const int64_t *y_lengths, int B, int64_t stride,
              int auto_normalize, int16_t *pcm, void *stream);

/* ---- introspection (tests, debugging) ---------------------------------------
 * Copies an internal stage tensor of the last call into `dst` (device).
 * Names: "x_enc" [B,H,T], "m_text", "logs_text" [B,I,T], "logw", "w_ceil"
 * [B,1,T], "x_post" [B,72,F], "dec_conv_pre", "dec_up_0", "dec_res_0",
 * "dec_up_1", "dec_res_1".  Returns the element count, or < 0 on error;
 * dst == NULL only queries the count. */
int64_t mbv_read_stage(mbv_model *m, const char *name, float *dst, int64_t capacity,
                       void *stream);

/* The text encoder's windowed relative-position attention by itself (tests; attentions.py:148-243):
 * qkv DEVICE [B, 3H, T] (q | k | v as the fused projection leaves them), emb_k / emb_v DEVICE [9, H / n_heads]
 * (heads share), lengths DEVICE int64 [B], o DEVICE [B, H, T].  Synchronises the stream. */
int mbv_op_rel_attention(mbv_model *m, const float *qkv, const float *emb_k, const float *emb_v,
                         const int64_t *lengths, float *o, int B, int H, int n_heads, int T, void *stream);

/* Generic conv1d through the MFMA kernel (tests): y = conv(x, w) + bias,
 * 'same' padding.  w HOST [Cout, Cin, K], bias HOST [Cout] or NULL,
 * x/y DEVICE [B, Cin, T] / [B, Cout, T]; in_slope: leaky-relu slope applied
 * to x first (1 = none). */
int mbv_op_conv1d(mbv_model *m, const float *x, const float *w_host, const float *bias_host,
                  float *y, int B, int Cin, int Cout, int T, int K, int dilation,
                  float in_slope, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MBISTFT_VITS_H */
