// Internal launch interface of the gfx950 kernels (host side, no torch).
// Layout everywhere: fp32 [B, C, time], time fastest.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mbv {

// ---------------------------------------------------------------- conv1d (MFMA)
// Packed weight layout of the implicit-GEMM conv kernel (k-interleaved, see conv1d.hip):
//   Wp[K][Cin/8][ci & 1][Mpad][(ci % 8) / 2], Mpad a multiple of 128, rows >= M zero.
enum ConvEpilogue : int {
  EPI_STORE = 0,     // y = acc + bias  [relu] [*out_mask]
  EPI_RESID = 1,     // y = acc + bias + res (+chan_add)
  EPI_RESID_ACC = 2, // y = ((accum_in ? accum_in : 0) + acc + bias + res (+chan_add)) * out_scale
  EPI_GATE = 3,      // rows come in (tanh-tile, sigmoid-tile) pairs: y[c] = tanh(.)*sigmoid(.)
  EPI_RES_SKIP = 4,  // row < split: xio = (xio + v) * mask ; else skip[row-split] (+)= v
  EPI_COUPLE = 5,    // y = (y + couple_sign * (acc + bias) * mask) * mask   (sign -1: reverse flow)
  EPI_CONVT = 6,     // ConvTranspose1d(k16, stride U = 4 | 8, p (16-U)/2) as a (16/U + 1)-tap conv over its U
                     // output phases: rows come in groups of 64 = 64/U channels x (first U/2 phases | last
                     // U/2 phases); tap 0 is all-zero for the second half, the last tap for the first
                     // (their MFMAs are skipped); a lane stores y[co, U t .. U t + U - 1]
  EPI_LN = 7,        // conv1d_narrow.hip only (T <= 256, M <= 768): y = LN_c( relu?(acc + bias) * out_mask + res ) * gamma + beta
                     // [* ln_mask]: the conv that feeds a channel LayerNorm and that LayerNorm in one launch
                     // (attentions.py:40-46 conv_o / ffn -> norm(x + y); models.py:128-135 conv -> relu -> norm)
};

struct ConvArgs {
  // input
  const float* x;          // [B, Cin, Tin] (+ offsets folded into the pointer)
  int64_t x_bstride;       // elements between batches
  int Tin;                 // valid input length (positions >= Tin read as zero padding)
  int x_rstride;           // elements between channel rows of x (>= Tin)
  int Cin;
  // weights
  const float* w;          // packed, k-interleaved (above)
  const float* bias;       // [M] in packed-row order, or nullptr
  int M;                   // real output rows (packed order)
  int Mpad;
  int K;
  int dil;
  int pad_left;            // input index = t + tap*dil - pad_left
  // prologue
  float in_slope;          // leaky-relu slope on the input (1 = identity)
  const int* in_lens;      // if set: input *= (t < in_lens[b])
  const float* chan_add;   // if set: [B, Cin] added to the input before the activation
  int reflect1;            // ReflectionPad1d((1,0)) in front of the conv (models.py:364)
  // output
  float* y;                // [B, My, T]
  int64_t y_bstride;
  int T;                   // output length (row stride of y / res / accum)
  int epi;
  int relu;
  const int* out_lens;     // mask for STORE / RES_SKIP / COUPLE
  const float* res;        // RESID*: residual [B, M, T]
  int64_t res_bstride;
  const float* res_chan_add;  // RESID*: [B, M] added to the residual (ResBlock cond)
  const float* accum_in;   // RESID_ACC
  float out_scale;
  const float* gate_cond;  // GATE: [B, gate_cond_stride] conditioning, already offset to this layer
  int gate_cond_bstride;
  int gate_half;           // GATE: H (rows of the tanh half)
  float* skip;             // RES_SKIP: [B, M - split, T]
  int split;               // RES_SKIP
  int skip_accum;          // RES_SKIP: skip += v instead of skip = v
  float couple_sign;       // COUPLE: -1 reverse (x1 - m), +1 forward (x1 + m)
  int B;
  int debug;               // timing experiments only (MBV_CONV_DEBUG): 1 = no restaging, 3 = no MFMA
  // r03, opt-in trimmed decode (mbv_set_option "trim"): only the column tiles that hold frames below a per-utterance
  // limit exist as work.  trim_map (device, launch_trim_map): [0 .. B] prefix sums of ceil(limit_b / BN), then the
  // utterance of every column tile; trim_bn = the BN it was built for (checked by the launcher).  The kernel walks
  // (column tile, row tile) pairs of that compact list; nothing else changes.  nullptr: every tile (default).
  const int* trim_map;
  int trim_bn;
  // r03 virtual-sequence tiling (launch_conv1d, EPI_CONVT): > 0 = column tiles run through the batch laid end to end,
  // utterance b at virtual column b * vs_tv (vs_tv = T + halo); set by the launcher only
  int vs_tv;
  // split-K scratch (small launches): partial accumulators + one self-resetting ticket per tile
  float* ws;
  size_t ws_floats;
  unsigned* counters;
  int n_counters;
  int convt_u;             // EPI_CONVT: upsampling stride (4 or 8)
  int splitk;              // 1: split-K allowed for this launch (mbv_set_option "splitk" / MBV_CONV_SPLITK)
  int prec;                // 0: exact fp32 MFMA (default); 3: opt-in split-bf16, three products (mbv_set_option "conv_bf16")
  const float* w_split;    // prec == 3: the same packed weights as [bf16 hi x 4 | bf16 mid x 4] slots (launch_split_planes)
  // EPI_LN
  const float* ln_gamma;   // [M]
  const float* ln_beta;    // [M]
  const int* ln_out_lens;  // mask behind the LayerNorm (the encoder's last layer), or nullptr
};
void launch_conv1d(const ConvArgs& a, hipStream_t s);
bool conv1d_supported(int K, int dil);   // kernel sizes / dilations the MFMA kernel is built for
// conv1d_narrow.hip: the same contraction in 32-column x row-block units (launches with few columns)
bool conv1d_narrow_supported(const ConvArgs& a);
void launch_conv1d_narrow(const ConvArgs& a, bool by_launch_size, hipStream_t s);

// ---------------------------------------------------------------- fused WN layer (wn_fused.hip)
// One layer of modules.WN (k = 5, dilation 1) in one launch: gate conv + tanh*sigmoid + 1x1 res/skip +
// residual / skip update, over the 32-frame units that hold valid frames.
//   wg  gate conv packed like every conv (Wp[tap][H/8][ci & 1][Mg_pad][(ci % 8) / 2]) with the ROWS in
//       tiles of 32 = [tanh ch 16t..16t+7 | sigmoid 16t..16t+7 | tanh 16t+8..16t+15 | sigmoid 16t+8..16t+15]
//   wr  res/skip 1x1 packed with natural rows and the input channels permuted: packed channel ci
//       <- source channel 8 (ci / 8) + 4 (ci & 1) + ((ci & 7) >> 1)   (the order the gated tile leaves
//       the accumulators in)
//   bg / gcond in reference row order ([tanh rows | sigmoid rows]); br natural.
struct WnLayerArgs {
  const float* h_in;       // [B, H, T]  (read with the frame mask applied)
  float* h_out;            // [B, H, T]  != h_in; unused when last
  float* skip;             // [B, H, T]
  const int* lens;         // [B]
  const int* ustart;       // [B + 1]: launch_wn_units (prefix sums of ceil(len / 16))
  const int* hmap;         // [ustart[B] + 1]: utterance of every half-unit
  const float* wg; const float* bg;
  const float* gcond;      // [B, gcond_bstride] already offset to this layer, or nullptr
  int gcond_bstride;
  const float* wr; const float* br;
  int B, H, T;
  int Mg_pad;              // padded rows of wg
  int Mr, Mr_pad;          // rows of the res/skip conv (2H, or H for the last layer)
  int last;                // Mr == H: every row goes to skip
  int skip_accum;          // skip += (layers > 0) instead of skip =
  int debug;               // set by the launcher (MBV_WN_DEBUG_A): timing experiments, results wrong by design
  // r03: the coupling layer's 1x1 `post` conv (modules.py:346-350) folded into the res/skip convs: their skip rows are
  // W_post . W_rs[skip rows] (Cs = I/2 rows instead of H), so `skip` accumulates m = post(sum of skips) directly and
  // the LAST layer applies the coupling x1 = (x1 + couple_sign * m) on its valid frames instead of storing skip.
  // r03, first layer of a coupling only: the 1x1 `pre` conv (modules.py:339) folded in as well.  h = (W_pre x0 + b) mask
  // is linear in x0' = [x0 ; mask] (Cin' = I/2 + 1 channels, padded to 8 Gi), so the gate conv runs on x0' directly with
  // the composite weights W_in[tap] W_pre' (K-loop of Gi instead of H/8 groups: 13 instead of 24) and the residual rows
  // start from W_pre' x0'(t) computed as one more K-block of the res/skip GEMM (wpre: rows H, padded to >= 128 NRT).
  // h_in then is the x0 half of z: in_cb channels per utterance, the first 8 (Gi - 1) rows real, group Gi - 1 = the mask.
  int Gi;                  // input channel groups of the gate conv / the window; 0: H / 8
  int in_cb;               // channel rows per utterance of h_in; 0: H
  const float* wpre;       // W_pre' packed like every conv (Cin = 8 Gi, natural channel order), or nullptr
  int wpre_Mpad;
  int Cs;                  // channels of `skip` ([B, Cs, T]); 0: H (unfolded)
  float* x1;               // last layer only: the half of z the coupling updates, [B, x1_bstride / T ..] rows Cs; nullptr: store skip
  int64_t x1_bstride;      // elements between utterances of x1 (I * T)
  float couple_sign;       // -1 reverse (x1 - m), +1 forward (x1 + m)
};
bool wn_fused_supported(int H, int K);
bool wn_fused_fits(int B, int H, int T);      // h / skip small enough for the kernel's 32-bit offsets
// `ustart` points at wn_units_ints(B, T) ints: [B + 1] prefix sums, then the half-unit -> utterance map
size_t wn_units_ints(int B, int T);
void launch_wn_units(const int* lens, int B, int T, int* ustart, int* hmap, hipStream_t s);
void launch_wn_layer(const WnLayerArgs& a, hipStream_t s);

// ---------------------------------------------------------------- ConvTranspose1d k=16, stride 4 / 8 (MFMA)
// Packed: Wt[r][j][Cin][Mpad] with Wt[r][j][ci][co] = W[ci][co][(r + pad) % stride + stride * j],
// r < stride, j < 16 / stride, pad = (16 - stride) / 2
struct ConvTArgs {
  const float* x;     // [B, Cin, Tin]
  const float* w;     // packed
  const float* bias;  // [Cout]
  float* y;           // [B, Cout, stride*Tin]
  int B, Cin, Cout, Mpad, Tin;
  float in_slope;
  int stride;         // 4 or 8
};
void launch_convt(const ConvTArgs& a, hipStream_t s);

// ---------------------------------------------------------------- text encoder pieces
// bad[b] is set when an utterance has a token id / length outside the valid range
void launch_embed(const int64_t* ids, const int64_t* lens, const float* emb, float* x, int* lens32,
                  int* bad, int B, int T, int H, int n_vocab, hipStream_t s);
// y = LN_c( a (+ r) [relu] ) * gamma + beta  [* mask]
void launch_split_planes(const float* src, float* dst, size_t n_floats, hipStream_t s);   // fp32 slots -> [hi x 4 | mid x 4] bf16
void launch_layernorm(const float* a, const float* r, const float* gamma, const float* beta,
                      float* y, int B, int C, int T, int pre_relu, const int* out_lens,
                      hipStream_t s);
// windowed relative-position attention, qkv [B, 3H, T] -> o [B, H, T]
void launch_rel_attention(const float* qkv, const float* emb_k, const float* emb_v,
                          const int* lens, float* o, int B, int H, int n_heads, int T,
                          hipStream_t s);

// ---------------------------------------------------------------- durations / length regulation
// logw = (w . h*mask + b) * mask ; w_ceil = ceil(exp(logw)*mask*scale) ; cum = cumsum ; ylen
void launch_durations(const float* h, const float* w, const float* b, const int* lens,
                      float length_scale, float* logw, float* w_ceil, int* cum, int* ylen32,
                      int64_t* ylen64, const int* bad, int B, int C, int T, hipStream_t s);
// (w == nullptr: h is logw itself, [B, T] — the SDP path)

// ---------------------------------------------------------------- StochasticDurationPredictor (sdp.hip)
// DDSConv halves (modules.py:98-111), C <= 256
void launch_dds_sep(const float* x, const int* lens, const float* w, const float* bias,
                    const float* gamma, const float* beta, float* y, int B, int C, int T, int K,
                    int dil, hipStream_t s);
void launch_dds_res(const float* a, const float* xres, const float* gamma, const float* beta,
                    float* y, int B, int C, int T, const int* out_lens, hipStream_t s);
// h = pre_w * z[:, zc] + pre_b + cond                      (ConvFlow.pre + DDSConv's x + g)
void launch_sdp_pre(const float* z, int zc, const float* pre_w, const float* pre_b, const float* cond,
                    float* h, int B, int C, int T, hipStream_t s);
// Flip + inverse rational-quadratic spline + mask, in place on z [B, 2, T]; h [B, 29, T]
void launch_sdp_spline(const float* h, float* z, const int* lens, int B, int C, int T,
                       float edge_const, hipStream_t s);
void launch_sdp_logw(const float* z, const float* m, const float* logs, const int* lens, float* logw,
                     int B, int T, hipStream_t s);
void launch_sdp_noise(const float* noise, float scale, float* z, int64_t n, hipStream_t s);
// x[b, c, t] += v[b, c]
void launch_chan_add(float* x, const float* v, int B, int C, int T, hipStream_t s);
// m_t / logs_t: [B, C, T] views with batch stride src_bstride (halves of the enc_p.proj output)
void launch_expand(const float* m_t, const float* logs_t, int64_t src_bstride, const int* cum,
                   const int* ylen, const float* noise, float noise_scale, float* m_p,
                   float* logs_p, float* z_p, float* z, float* attn, float* y_mask, int B, int C,
                   int T, int Tp, hipStream_t s);

// ---------------------------------------------------------------- speaker conditioning
// out[b][co] = bias[co] + sum_ci W[co][ci] * g[b][ci]   (g = table[sid[b]] if sid)
void launch_cond_gemv(const float* g, const float* table, const int64_t* sid, const float* W,
                      const float* bias, float* out, int B, int Cin, int Cout, hipStream_t s);
void launch_gather_rows(const float* table, const int64_t* sid, float* out, int B, int C,
                        int n_rows, int* bad, hipStream_t s);

// ---------------------------------------------------------------- fused iSTFT + PQMF
struct IstftArgs {
  const float* x_post;   // [B, 72, F]
  const float* filt;     // 352-float device table, see istft_pqmf.hip (generic taps | c[k][q] | g[j])
  float* o;              // [B, 256 T']
  float* o_mb;           // MB: [B,4,64T'] ; MS: [B,4,256T'] zero-stuffed ; or null
  float* spec;           // [B,4,9,F] or null
  float* phase;          // [B,4,9,F] or null
  int B, Tp, multistream;
  int fixed_bank;        // 1: the PQMF design (cosine-modulated, factorised); 0: arbitrary 4x63 taps
  int exact_math;        // 1: libm expf/sinf/sincosf instead of the hardware transcendentals
  int prescaled;         // 1: x_post rows already carry log2(e) (magnitude) / 1/(2 pi) (phase)
  int polar_in;          // 1: x_post unused; spec / phase [B,4,9,F] are the INPUT (istft_finalize)
  int nt_stores;         // set by the launcher (MBV_ISTFT_NT, default 1): non-temporal stores for spec / phase / o_mb
  const int* trim_lens;  // opt-in trimmed decode: [B] valid z-frames; samples at and beyond 256 * trim_lens[b] are not computed
                         // (the caller zero-fills o; o_mb / spec / phase must be null)
};
void launch_istft_pqmf(const IstftArgs& a, hipStream_t s);

// single-band iSTFT (iSTFT_Generator, models.py:296-300): x_post [B, 18, F] -> o [B, 4 (F-1)]
struct IstftSbArgs {
  const float* x_post;   // [B, 18, F]
  float* o;              // [B, 4 (F - 1)]
  float* spec;           // [B, 9, F] or null
  float* phase;          // [B, 9, F] or null
  int B, F;
  int exact_math, prescaled;
  int polar_in;          // 1: spec / phase [B,9,F] are the input
};
void launch_istft_single(const IstftSbArgs& a, hipStream_t s);

// x_post rows back to the reference's units (stage introspection): inverse of the pre-scaling
void launch_unscale_xpost(const float* src, float* dst, int B, int rows, int F, hipStream_t s);

// float waveform -> int16 PCM (normalise / clip / scale), tts_vits.py:204-217
void launch_pcm16(const float* x, const int64_t* lens, int B, int64_t stride, int spf, int auto_normalize,
                  unsigned* peak_scratch, short* out, hipStream_t s);

// z = (m + noise * exp(logs)) * mask   (PosteriorEncoder, models.py:245); stats = [B, 2I, T]
void launch_posterior_sample(const float* stats, const float* noise, const int* lens, float* z, int B,
                             int I, int T, hipStream_t s);
void launch_sequence_mask(const int* lens, float* mask, int B, int T, hipStream_t s);   // commons.py:121
void launch_lens_to_i32(const int64_t* lens, int* out, int B, int T, int* bad, hipStream_t s);

// misc
void launch_fill(float* p, float v, int64_t n, hipStream_t s);
// trimmed decode: column-tile map of one conv geometry; limit_b = min(T, lens[b] * num + add) output frames.
// out: (B + 1) + B * ceil(T / BN) ints (launch_trim_map_ints)
size_t launch_trim_map_ints(int B, int T, int BN);
void launch_trim_map(const int* lens, int B, int num, int add, int T, int BN, int* out, hipStream_t s);
// which column-tile width launch_conv1d will use for this conv (128 or 384; 0: a kernel without trim support)
int conv1d_trim_bn(const ConvArgs& a);

}  // namespace mbv
