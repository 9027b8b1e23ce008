// C-ABI (include/mbistft_vits.h) + host orchestration of the infer path.
// No torch, no exceptions across the boundary; all device work is enqueued on
// the caller's stream.  Reference call stack being replaced: SURVEY §3.1.
#include "../../include/mbistft_vits.h"
#include "kernels.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace mbv;

namespace {

constexpr int kWindow = 4;        // attentions.py:14
constexpr int kDpFilter = 256;    // models.py:652
constexpr int kFlowLayers = 4;    // models.py:647
constexpr int kFlowK = 5;
constexpr int kNFlows = 4;
constexpr float kLrelu = 0.1f;    // modules.py:17

thread_local std::string g_create_error;

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
  int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct PConv {              // packed conv living in the weight arena (offsets in floats)
  size_t w = 0, bias = 0;
  bool has_bias = false;
  int M = 0, Mpad = 0, Cin = 0, K = 1;
};
struct PVec { size_t off = 0; int n = 0; bool present = false; };

struct StageRef { const float* ptr; int64_t numel; };

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct mbv_model {
  mbv_config cfg{};
  std::string err;
  std::map<std::string, HostTensor> raw;
  std::map<std::string, std::vector<int64_t>> expected;   // key -> shape
  bool finalized = false;
  size_t arena_used = 0;           // floats of the packed arena in use (darena may be larger after a reload)
  int64_t import_n = 0;
  const float* import_src = nullptr;   // do_finalize: lay the arena out from the config alone and fill it from this device buffer (mbv_import_arena)

  // weight arena
  std::vector<float> harena;
  float* darena = nullptr;
  size_t darena_floats = 0;
  float* darena_split = nullptr;   // conv_bf16 mode: the arena as [bf16 hi x 4 | bf16 mid x 4] slots (ensure_split_arena)
  size_t darena_split_floats = 0;
  bool split_valid = false;

  // packed weights
  struct Layer { PConv qkv, o, ffn1, ffn2; PVec ek, ev, g1, b1, g2, b2; };
  std::vector<Layer> enc;
  PVec emb;
  PConv enc_proj;
  PConv dp1, dp2;
  PVec dp_g1, dp_b1, dp_g2, dp_b2, dp_pw, dp_pb, dp_cw, dp_cb;
  // StochasticDurationPredictor (models.py:20-52), reverse direction only
  struct Dds { PVec sw[3], sb[3], g1[3], b1[3], g2[3], b2[3]; PConv c1[3]; };
  struct SdpFlow { PVec pre_w, pre_b; Dds dds; PConv proj; };
  struct Sdp { PConv pre, proj; Dds dds; SdpFlow flow[3]; PVec m, logs; float edge_const = 0.f; } sdp;
  struct Flow { PConv pre, post, in[kFlowLayers], rs[kFlowLayers], in16[kFlowLayers], rsp[kFlowLayers]; PVec cw, cb;
                PConv rspf[kFlowLayers];      // rspf: res/skip convs with `post` folded into their skip rows (wn_fused.hip, r03)
                PConv in16f0, pref; int Gi = 0; };   // `pre` folded too: layer 0's gate conv on [x0 ; mask] (composite weights), W_pre' for the residual rows
  Flow flow[kNFlows];
  PConv conv_pre, conv_post;
  static constexpr int kEncQLayers = 16;     // models.py:646
  struct EncQ { PConv pre, proj, in[kEncQLayers], rs[kEncQLayers], in16[kEncQLayers], rsp[kEncQLayers]; PVec cw, cb; int cin_pad = 0; } encq;
  struct Up { size_t w = 0, bias = 0; int Cin = 0, Cout = 0, Mpad = 0; } ups[2];
  PConv upc[2];              // stride-4 ups as 5-tap convs over the output phases (EPI_CONVT)
  struct RB { PConv c1[3], c2[3]; PVec cw, cb; } rb[6];
  PVec emb_g;
  PVec filt;                 // synthesis-bank table of the fused iSTFT+PQMF kernel (352 floats)

  // scratch
  float* conv_ws = nullptr; size_t conv_ws_floats = 0;   // split-K partials of small conv launches
  unsigned* conv_cnt = nullptr; int conv_ncnt = 0;        // one ticket counter per tile (zero between launches)
  int wn_fused = 1;           // MBV_WN_FUSED=0: the two-launch WN layer (gate conv, then res/skip conv)
  int splitk = 0;             // option "splitk": split-K for small conv launches (default: MBV_CONV_SPLITK or 0)
  char* scrA = nullptr; size_t scrA_bytes = 0;
  char* scrB = nullptr; size_t scrB_bytes = 0;
  float* user_tab = nullptr;   // polyphase table of the stand-alone mbv_istft_pqmf entry
  unsigned* peak_buf = nullptr; int peak_cap = 0;   // per-utterance peaks of mbv_pcm16
  bool user_tab_is_pqmf = false;
  int xpost_F = 1;             // frames per row of the last x_post stage tensor
  int xpost_rows = 72;         // 72 (4 bands x 18) or 18 (single band)
  int exact_math = 0;          // MBV_ISTFT_EXACT=1: libm transcendentals in the iSTFT kernel
  int trim = 0;                    // option "trim": opt-in trimmed decode (run_decoder)
  int64_t xpost_chunk_bytes = 0;   // option "xpost_chunk_bytes": sub-batch cap of conv_post + iSTFT (0: 2 GiB - 1)

  // state of the last encode
  int B = 0, T = 0;
  bool encoded = false, has_g = false;
  float *x_enc = nullptr, *stats = nullptr, *logw = nullptr, *w_ceil = nullptr, *gvec = nullptr;
  int *lens32 = nullptr, *cum = nullptr, *ylen32 = nullptr;
  std::map<std::string, StageRef> stages;

  static constexpr int kEvRing = 8;
  hipEvent_t evr[kEvRing][7]{};    // stage events of the last kEvRing encode (+ synthesize) calls
  hipEvent_t* ev = evr[0];         // ... of the current call
  int64_t ticket = 0;              // calls of mbv_encode so far; slot = ticket % kEvRing
  bool evr_a[kEvRing]{}, evr_b[kEvRing]{};
  hipEvent_t evk[3]{};          // decoder start / before istft / after istft
  // the three ResBlocks of a decoder stage on three streams when one of them cannot fill the chip (run_decoder)
  hipStream_t aux[2]{};
  hipEvent_t ev_fork{}, ev_rb[3]{};
  bool aux_ok = false;
  int dec_streams = 1;          // option "dec_streams" / MBV_DEC_STREAMS: 0 = always one stream
  int conv_bf16 = 0;            // option "conv_bf16" / MBV_CONV_BF16: 3 = opt-in split-bf16 arithmetic in the large conv launches
  bool ev_ok = false, ev_a = false, ev_b = false, evk_set = false;
  bool evk_split = false;          // the last decoder run split its batch: evk[1] does not separate conv stack and iSTFT

  int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    err = buf;
    return 1;
  }
  const float* W(size_t off) const { return darena + off; }
  const float* Wsplit(size_t off) const { return (conv_bf16 == 3 && split_valid) ? darena_split + off : nullptr; }
};

#define HIPCHK(m, call)                                                              \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) return (m)->fail("%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

namespace {

// conv_bf16 mode: (re)build the split copy of the weight arena (ops.hip launch_split_planes)
int ensure_split_arena(mbv_model* m, hipStream_t stream) {
  if (m->split_valid) return 0;
  if (!m->darena) return m->fail("conv_bf16: no weights on the device yet");
  if (m->darena_split && m->darena_split_floats < m->darena_floats) {
    HIPCHK(m, hipFree(m->darena_split));
    m->darena_split = nullptr;
  }
  if (!m->darena_split) {
    HIPCHK(m, hipMalloc((void**)&m->darena_split, m->darena_floats * sizeof(float)));
    m->darena_split_floats = m->darena_floats;
  }
  launch_split_planes(m->darena, m->darena_split, m->darena_floats, stream);
  HIPCHK(m, hipStreamSynchronize(stream));
  m->split_valid = true;
  return 0;
}

// Every entry point runs on the model's device and hands the caller's current device back on
// every exit path (a process may host models on several GPUs; hipSetDevice is per host thread).
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define DEVICE_GUARD(m)                                                          \
  DeviceGuard dev_guard_((m)->cfg.device);                                       \
  if (!dev_guard_.ok) return (m)->fail("hipSetDevice(%d) failed", (m)->cfg.device)

// ------------------------------------------------------------------ key table
void add_key(mbv_model* m, const std::string& k, std::initializer_list<int64_t> shape) {
  m->expected[k] = std::vector<int64_t>(shape);
}

void build_expected(mbv_model* m) {
  const mbv_config& c = m->cfg;
  const int H = c.hidden_channels, I = c.inter_channels, Fc = c.filter_channels;
  const int dk = H / c.n_heads, gin = c.gin_channels, C0 = c.upsample_initial_channel;
  char p[160];
  add_key(m, "enc_p.emb.weight", {c.n_vocab, H});
  for (int i = 0; i < c.n_layers; ++i) {
    snprintf(p, sizeof p, "enc_p.encoder.attn_layers.%d.", i);
    add_key(m, std::string(p) + "emb_rel_k", {1, 2 * kWindow + 1, dk});
    add_key(m, std::string(p) + "emb_rel_v", {1, 2 * kWindow + 1, dk});
    for (const char* n : {"conv_q", "conv_k", "conv_v", "conv_o"}) {
      add_key(m, std::string(p) + n + ".weight", {H, H, 1});
      add_key(m, std::string(p) + n + ".bias", {H});
    }
    for (int j = 1; j <= 2; ++j) {
      snprintf(p, sizeof p, "enc_p.encoder.norm_layers_%d.%d.", j, i);
      add_key(m, std::string(p) + "gamma", {H});
      add_key(m, std::string(p) + "beta", {H});
    }
    snprintf(p, sizeof p, "enc_p.encoder.ffn_layers.%d.", i);
    add_key(m, std::string(p) + "conv_1.weight", {Fc, H, c.kernel_size});
    add_key(m, std::string(p) + "conv_1.bias", {Fc});
    add_key(m, std::string(p) + "conv_2.weight", {H, Fc, c.kernel_size});
    add_key(m, std::string(p) + "conv_2.bias", {H});
  }
  add_key(m, "enc_p.proj.weight", {2 * I, H, 1});
  add_key(m, "enc_p.proj.bias", {2 * I});
  if (c.decoder == MBV_DEC_MULTISTREAM) add_key(m, "dec.updown_filter", {4, 4, 4});
  add_key(m, "dec.conv_pre.bias", {C0});
  add_key(m, "dec.conv_pre.weight_g", {C0, 1, 1});
  add_key(m, "dec.conv_pre.weight_v", {C0, I, 7});
  const bool sb = c.decoder == MBV_DEC_SINGLEBAND;
  const int post_rows = sb ? 18 : 72;
  const char* post_name = sb ? "dec.conv_post" : "dec.subband_conv_post";
  for (int i = 0; i < 2; ++i) {
    const int cin = C0 >> i, cout = C0 >> (i + 1);
    snprintf(p, sizeof p, "dec.ups.%d.", i);
    add_key(m, std::string(p) + "bias", {cout});
    add_key(m, std::string(p) + "weight_g", {cin, 1, 1});
    add_key(m, std::string(p) + "weight_v", {cin, cout, 16});
  }
  for (int i = 0; i < 2; ++i) {
    const int ch = C0 >> (i + 1);
    for (int j = 0; j < 3; ++j) {
      const int k = c.resblock_kernel_sizes[j];
      const bool rb1 = c.resblock_type == 1;
      for (const char* grp : {"convs1", "convs2", "convs"}) {
        const bool is2 = std::strcmp(grp, "convs") == 0;
        if (is2 == rb1) continue;                      // ResBlock1: convs1/convs2 x3, ResBlock2: convs x2
        for (int q = 0; q < (rb1 ? 3 : 2); ++q) {
          snprintf(p, sizeof p, "dec.resblocks.%d.%s.%d.", i * 3 + j, grp, q);
          add_key(m, std::string(p) + "bias", {ch});
          add_key(m, std::string(p) + "weight_g", {ch, 1, 1});
          add_key(m, std::string(p) + "weight_v", {ch, ch, k});
        }
      }
      if (gin) {
        snprintf(p, sizeof p, "dec.resblocks.%d.cond.", i * 3 + j);
        add_key(m, std::string(p) + "weight", {ch, gin, 1});
        add_key(m, std::string(p) + "bias", {ch});
      }
    }
  }
  add_key(m, std::string(post_name) + ".bias", {post_rows});
  add_key(m, std::string(post_name) + ".weight_g", {post_rows, 1, 1});
  add_key(m, std::string(post_name) + ".weight_v", {post_rows, C0 >> 2, 7});
  if (c.decoder == MBV_DEC_MULTISTREAM) {
    add_key(m, "dec.multistream_conv_post.weight_g", {1, 1, 1});
    add_key(m, "dec.multistream_conv_post.weight_v", {1, 4, 63});
  }
  // enc_q (PosteriorEncoder, models.py:217-246): only voice_conversion reads it, but it is part of
  // every reference checkpoint, so the keys are accepted and required like the rest
  add_key(m, "enc_q.pre.weight", {H, c.spec_channels, 1});
  add_key(m, "enc_q.pre.bias", {H});
  for (int l = 0; l < mbv_model::kEncQLayers; ++l) {
    const int rs = l < mbv_model::kEncQLayers - 1 ? 2 * H : H;
    char q[64];
    snprintf(q, sizeof q, "enc_q.enc.in_layers.%d.", l);
    add_key(m, std::string(q) + "bias", {2 * H});
    add_key(m, std::string(q) + "weight_g", {2 * H, 1, 1});
    add_key(m, std::string(q) + "weight_v", {2 * H, H, 5});
    snprintf(q, sizeof q, "enc_q.enc.res_skip_layers.%d.", l);
    add_key(m, std::string(q) + "bias", {rs});
    add_key(m, std::string(q) + "weight_g", {rs, 1, 1});
    add_key(m, std::string(q) + "weight_v", {rs, H, 1});
  }
  if (gin) {
    add_key(m, "enc_q.enc.cond_layer.bias", {2 * H * mbv_model::kEncQLayers});
    add_key(m, "enc_q.enc.cond_layer.weight_g", {2 * H * mbv_model::kEncQLayers, 1, 1});
    add_key(m, "enc_q.enc.cond_layer.weight_v", {2 * H * mbv_model::kEncQLayers, gin, 1});
  }
  add_key(m, "enc_q.proj.weight", {2 * I, H, 1});
  add_key(m, "enc_q.proj.bias", {2 * I});
  for (int f = 0; f < kNFlows; ++f) {
    snprintf(p, sizeof p, "flow.flows.%d.", 2 * f);
    const std::string s(p);
    add_key(m, s + "pre.weight", {H, I / 2, 1});
    add_key(m, s + "pre.bias", {H});
    for (int l = 0; l < kFlowLayers; ++l) {
      const int rs = l < kFlowLayers - 1 ? 2 * H : H;
      char q[64];
      snprintf(q, sizeof q, "enc.in_layers.%d.", l);
      add_key(m, s + q + "bias", {2 * H});
      add_key(m, s + q + "weight_g", {2 * H, 1, 1});
      add_key(m, s + q + "weight_v", {2 * H, H, kFlowK});
      snprintf(q, sizeof q, "enc.res_skip_layers.%d.", l);
      add_key(m, s + q + "bias", {rs});
      add_key(m, s + q + "weight_g", {rs, 1, 1});
      add_key(m, s + q + "weight_v", {rs, H, 1});
    }
    if (gin) {
      add_key(m, s + "enc.cond_layer.bias", {2 * H * kFlowLayers});
      add_key(m, s + "enc.cond_layer.weight_g", {2 * H * kFlowLayers, 1, 1});
      add_key(m, s + "enc.cond_layer.weight_v", {2 * H * kFlowLayers, gin, 1});
    }
    add_key(m, s + "post.weight", {I / 2, H, 1});
    add_key(m, s + "post.bias", {I / 2});
  }
  if (c.use_sdp) {
    // models.py:20-52 (filter_channels := in_channels); post_* is training-only but part of the checkpoint
    auto dds = [&](const std::string& q) {
      for (int i = 0; i < 3; ++i) {
        const std::string n = std::to_string(i);
        add_key(m, q + "convs_sep." + n + ".weight", {H, 1, 3});
        add_key(m, q + "convs_sep." + n + ".bias", {H});
        add_key(m, q + "convs_1x1." + n + ".weight", {H, H, 1});
        add_key(m, q + "convs_1x1." + n + ".bias", {H});
        for (const char* g : {"norms_1.", "norms_2."}) {
          add_key(m, q + g + n + ".gamma", {H});
          add_key(m, q + g + n + ".beta", {H});
        }
      }
    };
    auto flows = [&](const std::string& q) {
      add_key(m, q + "0.m", {2, 1});
      add_key(m, q + "0.logs", {2, 1});
      for (int f = 1; f <= 7; f += 2) {
        const std::string r = q + std::to_string(f) + ".";
        add_key(m, r + "pre.weight", {H, 1, 1});
        add_key(m, r + "pre.bias", {H});
        dds(r + "convs.");
        add_key(m, r + "proj.weight", {29, H, 1});
        add_key(m, r + "proj.bias", {29});
      }
    };
    flows("dp.flows.");
    add_key(m, "dp.post_pre.weight", {H, 1, 1});
    add_key(m, "dp.post_pre.bias", {H});
    add_key(m, "dp.post_proj.weight", {H, H, 1});
    add_key(m, "dp.post_proj.bias", {H});
    dds("dp.post_convs.");
    flows("dp.post_flows.");
    add_key(m, "dp.pre.weight", {H, H, 1});
    add_key(m, "dp.pre.bias", {H});
    add_key(m, "dp.proj.weight", {H, H, 1});
    add_key(m, "dp.proj.bias", {H});
    dds("dp.convs.");
  } else {
    add_key(m, "dp.conv_1.weight", {kDpFilter, H, 3});
    add_key(m, "dp.conv_1.bias", {kDpFilter});
    add_key(m, "dp.norm_1.gamma", {kDpFilter});
    add_key(m, "dp.norm_1.beta", {kDpFilter});
    add_key(m, "dp.conv_2.weight", {kDpFilter, kDpFilter, 3});
    add_key(m, "dp.conv_2.bias", {kDpFilter});
    add_key(m, "dp.norm_2.gamma", {kDpFilter});
    add_key(m, "dp.norm_2.beta", {kDpFilter});
    add_key(m, "dp.proj.weight", {1, kDpFilter, 1});
    add_key(m, "dp.proj.bias", {1});
  }
  if (gin) {
    add_key(m, "dp.cond.weight", {H, gin, 1});
    add_key(m, "dp.cond.bias", {H});
  }
  if (c.n_speakers > 1) add_key(m, "emb_g.weight", {c.n_speakers, gin});
}

// ------------------------------------------------------------------ packing
// k-interleaved weight order the conv kernel copies verbatim into LDS (conv1d.hip):
//   Wp[tap][Cin/8][h = ci & 1][Mpad][s = (ci % 8) / 2]
inline size_t conv_pack_index(int tap, int ci, int m, int Cin, int Mpad) {
  return ((((size_t)tap * (Cin / 8) + ci / 8) * 2 + (ci & 1)) * Mpad + m) * 4 + ((ci & 7) >> 1);
}

struct Packer {
  mbv_model* m;
  std::vector<float>& a;
  size_t alloc(size_t n) {
    const size_t off = align_up(a.size(), 64);
    a.resize(off + n, 0.f);
    return off;
  }
  const HostTensor& t(const std::string& k) const { return m->raw.at(k); }
  bool has(const std::string& k) const { return m->raw.count(k) != 0; }

  // conv weight [d0, d1, K] as stored; folds weight-norm over dim 0 if *_v/_g
  std::vector<float> dense(const std::string& prefix) const {
    if (has(prefix + ".weight")) return t(prefix + ".weight").data;
    const HostTensor& v = t(prefix + ".weight_v");
    const HostTensor& g = t(prefix + ".weight_g");
    const int64_t d0 = v.shape[0], inner = v.numel() / d0;
    std::vector<float> w(v.data.size());
    if (m->import_src) return w;                   // layout-only pass: the contents come from the imported arena
    for (int64_t i = 0; i < d0; ++i) {
      double n2 = 0;
      for (int64_t j = 0; j < inner; ++j) { const double x = v.data[i * inner + j]; n2 += x * x; }
      const float scale = g.data[i] / (float)std::sqrt(n2);
      for (int64_t j = 0; j < inner; ++j) w[i * inner + j] = v.data[i * inner + j] * scale;
    }
    return w;
  }
  PVec vec(const std::string& k) {
    PVec r;
    if (!has(k)) return r;
    const HostTensor& x = t(k);
    r.off = alloc(x.data.size());
    r.n = (int)x.data.size();
    r.present = true;
    std::memcpy(&a[r.off], x.data.data(), x.data.size() * sizeof(float));
    return r;
  }
  PVec vec_data(const std::vector<float>& d) {
    PVec r;
    r.off = alloc(d.size()); r.n = (int)d.size(); r.present = true;
    std::memcpy(&a[r.off], d.data(), d.size() * sizeof(float));
    return r;
  }
  // generic conv: rows[m] -> source output channel (or -1 = zero row), cin_map[ci] -> source ci
  PConv conv(const std::vector<float>& w, int Cout, int Cin, int K, const std::vector<int>& rows,
             const std::vector<int>& cin_map, const std::vector<float>* bias,
             const std::vector<int>* bias_rows) {
    PConv p;
    p.M = (int)rows.size(); p.Mpad = (int)align_up(p.M, 128); p.Cin = Cin; p.K = K;
    p.w = alloc((size_t)K * Cin * p.Mpad);
    for (int k = 0; k < K; ++k)
      for (int ci = 0; ci < Cin; ++ci) {
        const int sci = cin_map.empty() ? ci : cin_map[ci];
        for (int mrow = 0; mrow < p.M; ++mrow) {
          const int co = rows[mrow];
          a[p.w + conv_pack_index(k, ci, mrow, Cin, p.Mpad)] =
              co < 0 ? 0.f : w[((size_t)co * Cin + sci) * K + k];
        }
      }
    if (bias) {
      const std::vector<int>& br = bias_rows ? *bias_rows : rows;
      p.bias = alloc(br.size());
      p.has_bias = true;
      for (size_t i = 0; i < br.size(); ++i) a[p.bias + i] = br[i] < 0 ? 0.f : (*bias)[br[i]];
    }
    (void)Cout;
    return p;
  }
  PConv conv_plain(const std::string& prefix) {
    const std::vector<float> w = dense(prefix);
    const auto& sh = has(prefix + ".weight") ? t(prefix + ".weight").shape : t(prefix + ".weight_v").shape;
    const int Cout = (int)sh[0], Cin = (int)sh[1], K = (int)sh[2];
    std::vector<int> rows(Cout);
    for (int i = 0; i < Cout; ++i) rows[i] = i;
    const std::vector<float>* b = has(prefix + ".bias") ? &t(prefix + ".bias").data : nullptr;
    return conv(w, Cout, Cin, K, rows, {}, b, nullptr);
  }
};

// WN in_layer (modules.py:130-135): gated packing, 32-row tiles alternate tanh half / sigmoid half
PConv pack_gated(Packer& P, const std::string& prefix, int H, int K) {
  const std::vector<float> w = P.dense(prefix);
  std::vector<int> rows(2 * H), brows(2 * H);
  for (int ch = 0; ch < H; ++ch) {
    rows[(ch / 32) * 64 + (ch % 32)] = ch;
    rows[(ch / 32) * 64 + 32 + (ch % 32)] = H + ch;
  }
  for (int r = 0; r < 2 * H; ++r) brows[r] = r;        // bias stays in reference order
  return P.conv(w, 2 * H, H, K, rows, {}, &P.t(prefix + ".bias").data, &brows);
}

// The same in_layer for the fused WN kernel (wn_fused.hip): 32-row tiles of
// [tanh 16t..16t+7 | sigmoid 16t..16t+7 | tanh 16t+8..16t+15 | sigmoid 16t+8..16t+15], which puts the
// tanh and the sigmoid row of a channel into the same lane of the 32x32 accumulator.  The bias
// stays in reference order.
PConv pack_gated16(Packer& P, const std::string& prefix, int H, int K) {
  const std::vector<float> w = P.dense(prefix);
  std::vector<int> rows(2 * H), brows(2 * H);
  for (int r = 0; r < 2 * H; ++r) {
    const int tile = r / 32, rho = r % 32;
    const int ch = tile * 16 + (rho & 7) + 8 * (rho >> 4);
    rows[r] = (rho & 8) ? H + ch : ch;
    brows[r] = r;
  }
  return P.conv(w, 2 * H, H, K, rows, {}, &P.t(prefix + ".bias").data, &brows);
}
// res_skip 1x1 with the input channels in the order the gated tile leaves the accumulators
PConv pack_rs_permuted(Packer& P, const std::string& prefix) {
  const std::vector<float> w = P.dense(prefix);
  const auto& sh = P.t(prefix + ".weight_v").shape;
  const int Cout = (int)sh[0], Cin = (int)sh[1];
  std::vector<int> rows(Cout), cmap(Cin);
  for (int i = 0; i < Cout; ++i) rows[i] = i;
  for (int ci = 0; ci < Cin; ++ci) cmap[ci] = 8 * (ci / 8) + 4 * (ci & 1) + ((ci & 7) >> 1);
  return P.conv(w, Cout, Cin, 1, rows, cmap, &P.t(prefix + ".bias").data, nullptr);
}

// modified Bessel I0 (power series; converges fast for x <= 9)
double bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  const double q = x * x / 4.0;
  for (int k = 1; k < 200; ++k) {
    term *= q / ((double)k * k);
    sum += term;
    if (term < 1e-18 * sum) break;
  }
  return sum;
}

// pqmf.py:15-43: Kaiser-windowed sinc prototype (taps 62, cutoff 0.15, beta 9), float64
std::vector<double> pqmf_prototype() {
  const int taps = 62;
  const double cutoff = 0.15, beta = 9.0;
  std::vector<double> proto(taps + 1);
  for (int n = 0; n <= taps; ++n) {
    const double c = n - 0.5 * taps;
    const double h = n == taps / 2 ? cutoff : std::sin(M_PI * cutoff * c) / (M_PI * c);
    const double alpha = taps / 2.0;
    const double r = (n - alpha) / alpha;
    proto[n] = h * bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / bessel_i0(beta);
  }
  return proto;
}

// pqmf.py:53-75: cosine-modulated synthesis bank h_syn[k][n], float64 -> float32
std::vector<float> pqmf_synthesis_filter() {
  const int taps = 62, K = 4;
  const std::vector<double> proto = pqmf_prototype();
  std::vector<float> h(K * (taps + 1));
  for (int k = 0; k < K; ++k)
    for (int n = 0; n <= taps; ++n) {
      const double sign = (k % 2 == 0) ? 1.0 : -1.0;
      h[k * (taps + 1) + n] = (float)(2.0 * proto[n] *
          std::cos((2 * k + 1) * (M_PI / (2.0 * K)) * (n - (taps - 1) / 2.0) - sign * M_PI / 4.0));
    }
  return h;
}

// Device table of the fused iSTFT+PQMF kernel (352 floats, layout in istft_pqmf.hip):
//   generic taps t[band][p][i] = 4 h[band][3 - p + 4 i] (x4 up-sampling gain folded, exact),
//   and for the fixed PQMF design its factorisation 4 h_k[j] = g[j] * c[k][j mod 8].
constexpr int kFiltTable = 352;
std::vector<float> synthesis_table(const float* h63) {
  std::vector<float> t(kFiltTable, 0.f);
  for (int band = 0; band < 4; ++band)
    for (int p = 0; p < 4; ++p)
      for (int i = 0; i < 16; ++i) {
        const int j = 3 - p + 4 * i;
        t[band * 64 + p * 16 + i] = j <= 62 ? 4.f * h63[band * 63 + j] : 0.f;
      }
  // fixed-bank factors (pqmf.py:66-75): theta_k(j) = (2k+1)(pi/8)(j - 30.5) - (-1)^k pi/4
  for (int k = 0; k < 4; ++k)
    for (int q = 0; q < 8; ++q) {
      const double sign = (k % 2 == 0) ? 1.0 : -1.0;
      t[256 + k * 8 + q] = (float)std::cos((2 * k + 1) * (M_PI / 8.0) * (q - 30.5) - sign * M_PI / 4.0);
    }
  const std::vector<double> proto = pqmf_prototype();
  for (int j = 0; j <= 62; ++j) t[288 + j] = (float)(8.0 * proto[j] * (((j / 8) % 2) ? -1.0 : 1.0));
  return t;
}

int do_finalize(mbv_model* m, hipStream_t stream) {
  if (m->import_src) {                 // zero tensors of the expected shapes stand in for the checkpoint
    m->raw.clear();
    for (auto& kv : m->expected) {
      HostTensor t;
      t.shape = kv.second;
      t.data.assign((size_t)t.numel(), 0.f);
      m->raw[kv.first] = std::move(t);
    }
  }
  for (auto& kv : m->expected)
    if (!m->raw.count(kv.first)) return m->fail("missing weight '%s'", kv.first.c_str());
  const mbv_config& c = m->cfg;
  const int H = c.hidden_channels, I = c.inter_channels, gin = c.gin_channels;
  std::vector<float> arena;
  Packer P{m, arena};
  char p[160];

  m->emb = P.vec("enc_p.emb.weight");
  m->enc.assign(c.n_layers, mbv_model::Layer());
  for (int i = 0; i < c.n_layers; ++i) {
    auto& L = m->enc[i];
    snprintf(p, sizeof p, "enc_p.encoder.attn_layers.%d.", i);
    const std::string s(p);
    {   // fused q|k|v 1x1 conv: rows [q(0..H) | k | v]
      std::vector<float> w(3 * (size_t)H * H), b(3 * (size_t)H);
      const char* names[3] = {"conv_q", "conv_k", "conv_v"};
      for (int j = 0; j < 3; ++j) {
        std::memcpy(&w[(size_t)j * H * H], P.t(s + names[j] + ".weight").data.data(), (size_t)H * H * 4);
        std::memcpy(&b[(size_t)j * H], P.t(s + names[j] + ".bias").data.data(), (size_t)H * 4);
      }
      std::vector<int> rows(3 * H);
      for (int r = 0; r < 3 * H; ++r) rows[r] = r;
      L.qkv = P.conv(w, 3 * H, H, 1, rows, {}, &b, nullptr);
    }
    L.o = P.conv_plain(s + "conv_o");
    L.ek = P.vec(s + "emb_rel_k");
    L.ev = P.vec(s + "emb_rel_v");
    snprintf(p, sizeof p, "enc_p.encoder.norm_layers_1.%d.", i);
    L.g1 = P.vec(std::string(p) + "gamma"); L.b1 = P.vec(std::string(p) + "beta");
    snprintf(p, sizeof p, "enc_p.encoder.norm_layers_2.%d.", i);
    L.g2 = P.vec(std::string(p) + "gamma"); L.b2 = P.vec(std::string(p) + "beta");
    snprintf(p, sizeof p, "enc_p.encoder.ffn_layers.%d.", i);
    L.ffn1 = P.conv_plain(std::string(p) + "conv_1");
    L.ffn2 = P.conv_plain(std::string(p) + "conv_2");
  }
  m->enc_proj = P.conv_plain("enc_p.proj");
  if (c.use_sdp) {
    auto dds = [&](const std::string& q) {
      mbv_model::Dds d;
      for (int i = 0; i < 3; ++i) {
        const std::string n = std::to_string(i);
        d.sw[i] = P.vec(q + "convs_sep." + n + ".weight");     // [C, 1, 3] -> [C][3]
        d.sb[i] = P.vec(q + "convs_sep." + n + ".bias");
        d.c1[i] = P.conv_plain(q + "convs_1x1." + n);
        d.g1[i] = P.vec(q + "norms_1." + n + ".gamma"); d.b1[i] = P.vec(q + "norms_1." + n + ".beta");
        d.g2[i] = P.vec(q + "norms_2." + n + ".gamma"); d.b2[i] = P.vec(q + "norms_2." + n + ".beta");
      }
      return d;
    };
    m->sdp.pre = P.conv_plain("dp.pre");
    m->sdp.proj = P.conv_plain("dp.proj");
    m->sdp.dds = dds("dp.convs.");
    for (int k = 0; k < 3; ++k) {               // reverse order of use: flows 7, 5, 3 (models.py:92-93)
      const std::string q = "dp.flows." + std::to_string(7 - 2 * k) + ".";
      m->sdp.flow[k].pre_w = P.vec(q + "pre.weight");
      m->sdp.flow[k].pre_b = P.vec(q + "pre.bias");
      m->sdp.flow[k].dds = dds(q + "convs.");
      m->sdp.flow[k].proj = P.conv_plain(q + "proj");
    }
    m->sdp.m = P.vec("dp.flows.0.m");
    m->sdp.logs = P.vec("dp.flows.0.logs");
    m->sdp.edge_const = (float)std::log(std::exp(1.0 - 1e-3) - 1.0);   // transforms.py:74
  } else {
    m->dp1 = P.conv_plain("dp.conv_1");
    m->dp2 = P.conv_plain("dp.conv_2");
    m->dp_g1 = P.vec("dp.norm_1.gamma"); m->dp_b1 = P.vec("dp.norm_1.beta");
    m->dp_g2 = P.vec("dp.norm_2.gamma"); m->dp_b2 = P.vec("dp.norm_2.beta");
    m->dp_pw = P.vec("dp.proj.weight"); m->dp_pb = P.vec("dp.proj.bias");
  }
  m->dp_cw = P.vec("dp.cond.weight"); m->dp_cb = P.vec("dp.cond.bias");
  m->emb_g = P.vec("emb_g.weight");

  // ---- flows: the channel Flip (modules.py:280-287) is folded into the packing.
  // Reverse pass order is f = 3,2,1,0, each preceded by a flip, so layers 3 and 1
  // see the physical buffer flipped, 2 and 0 see it straight.
  const int half = I / 2;
  for (int f = 0; f < kNFlows; ++f) {
    auto& F = m->flow[f];
    const bool flipped = (f % 2) == 1;
    snprintf(p, sizeof p, "flow.flows.%d.", 2 * f);
    const std::string s(p);
    {
      const std::vector<float> w = P.dense(s + "pre");
      std::vector<int> rows(H), cmap(half);
      for (int r = 0; r < H; ++r) rows[r] = r;
      for (int ci = 0; ci < half; ++ci) cmap[ci] = flipped ? half - 1 - ci : ci;
      F.pre = P.conv(w, H, half, 1, rows, cmap, &P.t(s + "pre.bias").data, nullptr);
    }
    if (wn_fused_supported(H, kFlowK)) {
      // r03: `pre` folded into the first fused WN layer.  h = (W_pre x0 + b_pre) mask = W_pre' x0' with
      // x0' = [x0 ; mask ; 0 ..] (Cin' = half + 1 channels padded to a multiple of 8) and W_pre' = [W_pre | b_pre | 0 ..]
      // (input channels in the physical order of the half, the Flip folded as in F.pre).  The gate conv of layer 0
      // is linear in h, so it runs on x0' with W_in0[tap] W_pre' (composed in fp64): 13 channel groups per tap instead of 24.
      const int Cp = (int)align_up((size_t)half + 1, 8);
      F.Gi = Cp / 8;
      const std::vector<float> wpre = P.dense(s + "pre");                     // [H][half][1]
      const std::vector<float>& bpre = P.t(s + "pre.bias").data;
      std::vector<float> wp((size_t)H * Cp, 0.f);                             // W_pre' [H][Cp]
      for (int r = 0; r < H; ++r) {
        for (int ci = 0; ci < half; ++ci) wp[(size_t)r * Cp + ci] = wpre[(size_t)r * half + (flipped ? half - 1 - ci : ci)];
        wp[(size_t)r * Cp + half] = bpre[r];
      }
      {   // rows past H must exist and be zero: every wave runs all its tile slots over it (wn_fused.hip pre_loop)
        const int Mrows = (int)align_up((size_t)H, 128) < 384 ? 384 : (int)align_up((size_t)H, 128);
        std::vector<int> rws(Mrows);
        for (int i = 0; i < Mrows; ++i) rws[i] = i < H ? i : -1;
        F.pref = P.conv(wp, H, Cp, 1, rws, {}, nullptr, nullptr);
      }
      {
        const std::vector<float> win = P.dense(s + "enc.in_layers.0");        // [2H][H][K]
        std::vector<float> wc((size_t)2 * H * Cp * kFlowK);
        std::vector<double> acc(Cp);
        for (int r = 0; r < 2 * H; ++r)
          for (int tap = 0; tap < kFlowK; ++tap) {
            std::fill(acc.begin(), acc.end(), 0.0);
            for (int k = 0; k < H; ++k) {
              const double wv = win[((size_t)r * H + k) * kFlowK + tap];
              const float* src = &wp[(size_t)k * Cp];
              for (int ci = 0; ci <= half; ++ci) acc[ci] += wv * src[ci];
            }
            for (int ci = 0; ci < Cp; ++ci) wc[((size_t)r * Cp + ci) * kFlowK + tap] = (float)acc[ci];
          }
        std::vector<int> rows(2 * H), brows(2 * H);                            // row order of pack_gated16
        for (int r = 0; r < 2 * H; ++r) {
          const int tile = r / 32, rho = r % 32;
          const int ch = tile * 16 + (rho & 7) + 8 * (rho >> 4);
          rows[r] = (rho & 8) ? H + ch : ch;
          brows[r] = r;
        }
        F.in16f0 = P.conv(wc, 2 * H, Cp, kFlowK, rows, {}, &P.t(s + "enc.in_layers.0.bias").data, &brows);
      }
    }
    for (int l = 0; l < kFlowLayers; ++l) {
      char q[64];
      snprintf(q, sizeof q, "enc.in_layers.%d", l);
      F.in[l] = pack_gated(P, s + q, H, kFlowK);
      if (wn_fused_supported(H, kFlowK)) F.in16[l] = pack_gated16(P, s + q, H, kFlowK);
      snprintf(q, sizeof q, "enc.res_skip_layers.%d", l);
      F.rs[l] = P.conv_plain(s + q);
      if (wn_fused_supported(H, kFlowK)) F.rsp[l] = pack_rs_permuted(P, s + q);
    }
    if (gin) {
      F.cw = P.vec_data(P.dense(s + "enc.cond_layer"));
      F.cb = P.vec(s + "enc.cond_layer.bias");
    }
    {
      const std::vector<float> w = P.dense(s + "post");
      std::vector<int> rows(half);
      for (int r = 0; r < half; ++r) rows[r] = flipped ? half - 1 - r : r;
      F.post = P.conv(w, half, H, 1, rows, {}, &P.t(s + "post.bias").data, nullptr);
      // r03: `post` folded into the res/skip convs of the fused WN layers.  m = post(sum_l skip_l) is linear in the
      // layers' gated tiles: m = sum_l (W_post W_rs_l[skip rows]) acts_l + (W_post sum_l b_rs_l[skip] + b_post), so
      // layer l's res/skip conv gets `half` skip rows (W_post W_rs_l[skip rows], composed in fp64) instead of H,
      // `skip` accumulates m itself and the last layer applies the coupling (WnLayerArgs::x1).  Row r of m is the
      // physical channel r of the half being updated (the Flip folded as in F.post above).
      if (wn_fused_supported(H, kFlowK)) {
        const std::vector<float>& bpost = P.t(s + "post.bias").data;
        for (int l = 0; l < kFlowLayers; ++l) {
          const std::string q = s + "enc.res_skip_layers." + std::to_string(l);
          const std::vector<float> wrs = P.dense(q);                 // [2H or H][H][1]
          const std::vector<float>& brs = P.t(q + ".bias").data;
          const bool lastl = l == kFlowLayers - 1;
          const int skip0 = lastl ? 0 : H;                            // first skip row of W_rs_l
          const int Mrows = (lastl ? 0 : H) + half;
          std::vector<float> wc((size_t)Mrows * H), bc(Mrows);
          for (int r = 0; r < (lastl ? 0 : H); ++r) {
            std::memcpy(&wc[(size_t)r * H], &wrs[(size_t)r * H], (size_t)H * sizeof(float));
            bc[r] = brs[r];
          }
          for (int r = 0; r < half; ++r) {
            const int pr = rows[r];
            std::vector<double> acc(H, 0.0);
            double bacc = lastl ? (double)bpost[pr] : 0.0;
            for (int k = 0; k < H; ++k) {
              const double wp = w[(size_t)pr * H + k];
              const float* src = &wrs[(size_t)(skip0 + k) * H];
              for (int cch = 0; cch < H; ++cch) acc[cch] += wp * src[cch];
              bacc += wp * brs[skip0 + k];
            }
            float* dst = &wc[(size_t)((lastl ? 0 : H) + r) * H];
            for (int cch = 0; cch < H; ++cch) dst[cch] = (float)acc[cch];
            bc[(lastl ? 0 : H) + r] = (float)bacc;
          }
          std::vector<int> rws(Mrows), cmap(H);
          for (int i = 0; i < Mrows; ++i) rws[i] = i;
          for (int ci = 0; ci < H; ++ci) cmap[ci] = 8 * (ci / 8) + 4 * (ci & 1) + ((ci & 7) >> 1);     // as pack_rs_permuted
          F.rspf[l] = P.conv(wc, Mrows, H, 1, rws, cmap, &bc, nullptr);
        }
      }
    }
  }

  // ---- posterior encoder (voice conversion only)
  {
    auto& Q = m->encq;
    const int cin = c.spec_channels, cpad = (int)align_up(cin, 32);
    Q.cin_pad = cpad;
    const std::vector<float>& w0 = P.t("enc_q.pre.weight").data;       // [H][cin][1]
    std::vector<float> w((size_t)H * cpad, 0.f);
    for (int co = 0; co < H; ++co)
      std::memcpy(&w[(size_t)co * cpad], &w0[(size_t)co * cin], (size_t)cin * sizeof(float));
    std::vector<int> rows(H);
    for (int r = 0; r < H; ++r) rows[r] = r;
    Q.pre = P.conv(w, H, cpad, 1, rows, {}, &P.t("enc_q.pre.bias").data, nullptr);
    for (int l = 0; l < mbv_model::kEncQLayers; ++l) {
      char q[64];
      snprintf(q, sizeof q, "enc_q.enc.in_layers.%d", l);
      Q.in[l] = pack_gated(P, q, H, 5);
      if (wn_fused_supported(H, 5)) Q.in16[l] = pack_gated16(P, q, H, 5);
      snprintf(q, sizeof q, "enc_q.enc.res_skip_layers.%d", l);
      Q.rs[l] = P.conv_plain(q);
      if (wn_fused_supported(H, 5)) Q.rsp[l] = pack_rs_permuted(P, q);
    }
    if (gin) {
      Q.cw = P.vec_data(P.dense("enc_q.enc.cond_layer"));
      Q.cb = P.vec("enc_q.enc.cond_layer.bias");
    }
    Q.proj = P.conv_plain("enc_q.proj");
  }

  // ---- decoder
  m->conv_pre = P.conv_plain("dec.conv_pre");
  for (int i = 0; i < 2; ++i) {
    snprintf(p, sizeof p, "dec.ups.%d", i);
    const std::vector<float> w = P.dense(p);            // [Cin][Cout][16], norm per Cin
    const auto& sh = P.t(std::string(p) + ".weight_v").shape;
    auto& U = m->ups[i];
    U.Cin = (int)sh[0]; U.Cout = (int)sh[1]; U.Mpad = (int)align_up(U.Cout, 64);
    U.w = P.alloc((size_t)16 * U.Cin * U.Mpad);
    const int us = c.decoder == MBV_DEC_SINGLEBAND ? 8 : 4;        // stride; k = 16, pad = (16-us)/2
    const int tpp = 16 / us, pad = (16 - us) / 2;
    for (int r = 0; r < us; ++r)
      for (int j = 0; j < tpp; ++j) {
        const int k = (r + pad) % us + us * j;
        for (int ci = 0; ci < U.Cin; ++ci) {
          float* dst = &arena[U.w + ((size_t)(r * tpp + j) * U.Cin + ci) * U.Mpad];
          for (int co = 0; co < U.Cout; ++co) dst[co] = w[((size_t)ci * U.Cout + co) * 16 + k];
        }
      }
    U.bias = P.vec(std::string(p) + ".bias").off;
    if ((us == 4 || us == 8) && U.Cout % 32 == 0 && U.Cin % 16 == 0) {
      // The same ConvTranspose1d as ONE (16/us + 1)-tap conv on the conv1d kernel (EPI_CONVT):
      //   y[co, us m + r] = sum_ci sum_j W[ci][co][kr + us j] x[ci, m + sh_r - j],  kr = (r + pad) % us,
      //   sh_r = (r + pad - kr) / us;  tap tau reads x[m - pl + tau]  ->  j = sh_r + pl - tau, with
      //   pl = tpp - 1 - pad / us  (us 4: 5 taps, pl 2;  us 8: 3 taps, pl 1).
      // Packed rows: groups of 64 = 64/us channels x [first us/2 phases (32 rows) | last us/2 phases
      // (32 rows)], row k of a half = channel k / (us/2), phase k % (us/2).  The last tap is zero for
      // the first half, tap 0 for the second; the kernel skips those MFMAs.
      const int Mp = us * U.Cout, PH = us / 2, CG = 32 / PH, Kc = tpp + 1, pl = tpp - 1 - pad / us;
      PConv pc;
      pc.M = Mp; pc.Mpad = (int)align_up(Mp, 128); pc.Cin = U.Cin; pc.K = Kc;
      pc.w = P.alloc((size_t)Kc * U.Cin * pc.Mpad);
      pc.bias = P.alloc(Mp);
      pc.has_bias = true;
      const HostTensor& bt = P.t(std::string(p) + ".bias");
      for (int row = 0; row < Mp; ++row) {
        const int grp = row / 64, half = (row % 64) / 32, k = row % 32;
        const int co = grp * CG + k / PH, r = half * PH + k % PH;
        const int kr = (r + pad) % us, sh = (r + pad - kr) / us;
        arena[pc.bias + row] = bt.data[co];
        for (int tau = 0; tau < Kc; ++tau) {
          const int j = sh + pl - tau;
          for (int ci = 0; ci < U.Cin; ++ci)
            arena[pc.w + conv_pack_index(tau, ci, row, U.Cin, pc.Mpad)] =
                (j >= 0 && j < tpp) ? w[((size_t)ci * U.Cout + co) * 16 + kr + us * j] : 0.f;
        }
      }
      m->upc[i] = pc;
    } else {
      m->upc[i] = PConv{};
    }
  }
  for (int n = 0; n < 6; ++n) {
    if (c.resblock_type == 1) {
      for (int q = 0; q < 3; ++q) {
        snprintf(p, sizeof p, "dec.resblocks.%d.convs1.%d", n, q);
        m->rb[n].c1[q] = P.conv_plain(p);
        snprintf(p, sizeof p, "dec.resblocks.%d.convs2.%d", n, q);
        m->rb[n].c2[q] = P.conv_plain(p);
      }
    } else {
      for (int q = 0; q < 2; ++q) {
        snprintf(p, sizeof p, "dec.resblocks.%d.convs.%d", n, q);
        m->rb[n].c1[q] = P.conv_plain(p);
      }
    }
    if (gin) {
      snprintf(p, sizeof p, "dec.resblocks.%d.cond.", n);
      m->rb[n].cw = P.vec(std::string(p) + "weight");
      m->rb[n].cb = P.vec(std::string(p) + "bias");
    }
  }
  {   // subband_conv_post, rows pre-scaled for the fused iSTFT kernel: exp(x) = 2^(x log2 e),
      // sin(x) = sin_turns(x / 2 pi)  (istft_pqmf.hip, template PRE)
    const std::string pn = c.decoder == MBV_DEC_SINGLEBAND ? "dec.conv_post" : "dec.subband_conv_post";
    std::vector<float> w = P.dense(pn);
    std::vector<float> bsc = P.t(pn + ".bias").data;
    const auto& sh = P.t(pn + ".weight_v").shape;
    const int Cout = (int)sh[0], Cin = (int)sh[1], K = (int)sh[2];
    for (int co = 0; co < Cout; ++co) {
      const float sc = (co % 18) < 9 ? 1.44269504088896341f : 0.15915494309189535f;
      for (int j = 0; j < Cin * K; ++j) w[(size_t)co * Cin * K + j] *= sc;
      bsc[co] *= sc;
    }
    std::vector<int> rows(Cout);
    for (int i = 0; i < Cout; ++i) rows[i] = i;
    m->conv_post = P.conv(w, Cout, Cin, K, rows, {}, &bsc, nullptr);
  }
  if (c.decoder == MBV_DEC_MULTISTREAM) {
    const std::vector<float> h = P.dense("dec.multistream_conv_post");   // [1][4][63]
    m->filt = P.vec_data(synthesis_table(h.data()));
  } else if (c.decoder == MBV_DEC_MULTIBAND) {
    const std::vector<float> h = pqmf_synthesis_filter();
    m->filt = P.vec_data(synthesis_table(h.data()));
  }

  // ---- upload
  if (m->darena && m->darena_floats < arena.size()) { HIPCHK(m, hipFree(m->darena)); m->darena = nullptr; }
  if (!m->darena) {
    HIPCHK(m, hipMalloc((void**)&m->darena, arena.size() * sizeof(float)));
    m->darena_floats = arena.size();
  }
  if (m->import_src) {
    if ((int64_t)arena.size() != m->import_n)
      return m->fail("mbv_import_arena: %lld floats offered, this configuration's arena has %zu (exported by another configuration or library build?)",
                     (long long)m->import_n, arena.size());
    HIPCHK(m, hipMemcpyAsync(m->darena, m->import_src, arena.size() * sizeof(float), hipMemcpyDeviceToDevice, stream));
    m->raw.clear();
  } else {
    HIPCHK(m, hipMemcpyAsync(m->darena, arena.data(), arena.size() * sizeof(float),
                             hipMemcpyHostToDevice, stream));
  }
  HIPCHK(m, hipStreamSynchronize(stream));
  m->arena_used = arena.size();
  m->harena.swap(arena);
  m->finalized = true;
  m->split_valid = false;
  if (m->conv_bf16 == 3 && ensure_split_arena(m, stream)) return 1;
  return 0;
}

// ------------------------------------------------------------------ scratch
struct Bump {
  char* base; size_t cap, off = 0;
  template <typename Tp> Tp* take(size_t n) {
    off = align_up(off, 256);
    Tp* p = reinterpret_cast<Tp*>(base + off);
    off += n * sizeof(Tp);
    return p;
  }
};

int ensure(mbv_model* m, char** buf, size_t* cap, size_t need) {
  if (*cap >= need) return 0;
  if (*buf) { HIPCHK(m, hipDeviceSynchronize()); HIPCHK(m, hipFree(*buf)); *buf = nullptr; *cap = 0; }
  need = align_up(need + (need >> 3), 1 << 20);
  HIPCHK(m, hipMalloc((void**)buf, need));
  *cap = need;
  return 0;
}

ConvArgs conv_args(const mbv_model* m, const PConv& p, const float* x, int64_t x_bstride, int Tin,
                   float* y, int64_t y_bstride, int T, int B, int dil = 1) {
  ConvArgs a{};
  a.x = x; a.x_bstride = x_bstride; a.Tin = Tin; a.x_rstride = Tin; a.Cin = p.Cin;
  a.w = m->W(p.w); a.bias = p.has_bias ? m->W(p.bias) : nullptr;
  a.w_split = m->Wsplit(p.w);
  a.M = p.M; a.Mpad = p.Mpad; a.K = p.K; a.dil = dil;
  a.pad_left = (p.K - 1) * dil / 2;
  a.in_slope = 1.f;
  a.y = y; a.y_bstride = y_bstride; a.T = T; a.epi = EPI_STORE; a.out_scale = 1.f; a.B = B;
  a.ws = m->conv_ws; a.ws_floats = m->conv_ws_floats; a.counters = m->conv_cnt; a.n_counters = m->conv_ncnt;
  a.splitk = m->splitk;
  a.prec = a.w_split ? 3 : 0;
  return a;
}

// decoder + waveform tail on z [B, I, zstride] (first Td frames valid)
size_t decoder_scratch_bytes(const mbv_config& c, int B, int Td);

// The three ResBlocks of a decoder stage on three streams?  Only when one of their convs (ch channels, Lo
// frames) is at most 192 tiles of 128 x 384, i.e. cannot fill the chip by itself.
bool decoder_stage_concurrent(const mbv_model* m, int B, int ch, int Lo) {
  const long conv_tiles = (long)B * ((Lo + 383) / 384) * ((ch + 127) / 128);
  return m->dec_streams && m->aux_ok && m->cfg.resblock_type != 2 && conv_tiles <= 192 && !m->trim;   // (trim: its tile maps are built on the caller's stream)
}

int run_decoder(mbv_model* m, const float* z, int zstride, const int* zlens, const float* gvec,
                int B, int Td, const mbv_outputs* outs, hipStream_t s, Bump& sc) {
  const mbv_config& c = m->cfg;
  const int I = c.inter_channels, C0 = c.upsample_initial_channel, gin = c.gin_channels;
  hipStream_t const s_main = s;
  if (sc.off + decoder_scratch_bytes(c, B, Td) > sc.cap) return m->fail("internal error: decoder scratch arena undersized");
  float* x0 = sc.take<float>((size_t)B * C0 * Td);
  HIPCHK(m, hipEventRecord(m->evk[0], s));
  // Opt-in trimmed decode (option "trim"; the caller takes only `o` and trims by y_lengths): a conv computes the
  // column tiles that hold frames below (len_b + kTrimMargin) * rate of its utterance and nothing behind them.  The
  // decoder's one-sided receptive field is 25.2 z-frames (conv_pre 3 + ups 2 + 0.5, the k = 11 ResBlocks 15 + 3.75,
  // conv_post / iSTFT / PQMF < 1), so whatever sits behind an utterance's limit — stale scratch — stays more than
  // 6 frames away from its valid samples: those are bitwise the default's.  Column-tile maps are built on the device
  // from ylen32 (no extra host sync), one per (rate, tile width) geometry.
  constexpr int kTrimMargin = 32;
  const bool trim = m->trim && zlens != nullptr && !m->splitk && c.decoder != MBV_DEC_SINGLEBAND;
  struct TrimKey { int num, add, T, bn; const int* map; };
  std::vector<TrimKey> trim_maps;
  auto with_trim = [&](ConvArgs& a, int num, int add) {
    if (!trim) return;
    const int bn = conv1d_trim_bn(a);
    if (!bn) return;
    for (const auto& k : trim_maps)
      if (k.num == num && k.add == add && k.T == a.T && k.bn == bn) { a.trim_map = k.map; a.trim_bn = bn; return; }
    int* map = sc.take<int>(launch_trim_map_ints(B, a.T, bn));
    launch_trim_map(zlens, B, num, num * kTrimMargin + add, a.T, bn, map, s_main);
    trim_maps.push_back({num, add, a.T, bn, map});
    a.trim_map = map; a.trim_bn = bn;
  };
  {
    ConvArgs a = conv_args(m, m->conv_pre, z, (int64_t)I * zstride, Td, x0, (int64_t)C0 * Td, Td, B);
    a.x_rstride = zstride;
    a.in_lens = zlens;
    with_trim(a, 1, 0);
    launch_conv1d(a, s);
  }
  m->stages["dec_conv_pre"] = {x0, (int64_t)B * C0 * Td};
  const float* cur = x0;
  int L = Td;
  float* xs = nullptr;
  const bool sb = c.decoder == MBV_DEC_SINGLEBAND;
  const int us = sb ? 8 : 4;                       // upsample stride of both stages
  for (int i = 0; i < 2; ++i) {
    const int ch = C0 >> (i + 1);
    const int Lo = us * L;
    const size_t n = (size_t)B * ch * Lo;
    float* u = sc.take<float>(n);
    // The three ResBlocks of a stage read the same input and only meet in the running sum xs.  When one
    // of their convs cannot fill the chip (single utterances, small batches: decoder_stage_concurrent)
    // they run on three streams — own temporaries each, the three xs updates chained by events in the
    // order of the one-stream schedule, so the result is bitwise the same.
    const bool conc = decoder_stage_concurrent(m, B, ch, Lo);
    float *t1s[3], *rs[3];
    t1s[0] = sc.take<float>(n);
    rs[0] = sc.take<float>(n);
    for (int j = 1; j < 3; ++j) {
      t1s[j] = conc ? sc.take<float>(n) : t1s[0];
      rs[j] = conc ? sc.take<float>(n) : rs[0];
    }
    xs = sc.take<float>(n);
    static const int convt_as_conv = [] { const char* e = getenv("MBV_CONVT_AS_CONV"); return e ? atoi(e) : 1; }();
    if (m->upc[i].M && convt_as_conv) {
      ConvArgs a = conv_args(m, m->upc[i], cur, (int64_t)m->ups[i].Cin * L, L, u, (int64_t)ch * Lo, L, B);
      a.pad_left = us == 4 ? 2 : 1;
      a.in_slope = kLrelu;
      a.epi = EPI_CONVT;
      a.convt_u = us;
      with_trim(a, L / Td, 0);                      // tiles run over INPUT frames (rate of the stage below)
      launch_conv1d(a, s);
    } else {
      ConvTArgs a{};
      a.x = cur; a.w = m->W(m->ups[i].w); a.bias = m->W(m->ups[i].bias); a.y = u;
      a.B = B; a.Cin = m->ups[i].Cin; a.Cout = ch; a.Mpad = m->ups[i].Mpad; a.Tin = L;
      a.in_slope = kLrelu;
      a.stride = us;
      launch_convt(a, s);
    }
    m->stages[i == 0 ? "dec_up_0" : "dec_up_1"] = {u, (int64_t)n};
    if (conc) {
      HIPCHK(m, hipEventRecord(m->ev_fork, s_main));
      for (auto& st : m->aux) HIPCHK(m, hipStreamWaitEvent(st, m->ev_fork, 0));
    }
    // (an early return between the fork and the join below would leave work queued on the aux streams that
    // the caller's stream never waits for, while the next call reuses this scratch: the loop runs in a lambda
    // and a failure drains the aux streams first)
    auto resblocks = [&]() -> int {
    for (int j = 0; j < 3; ++j) {
      const auto& R = m->rb[i * 3 + j];
      hipStream_t s = (conc && j > 0) ? m->aux[j - 1] : s_main;   // this ResBlock's stream
      float* const t1 = t1s[j];
      float* const r = rs[j];
      // split-K scratch (low-latency mode): a third each, so that concurrent convs never share partials or tickets
      auto own_ws = [&](ConvArgs& a) {
        if (!conc) return;
        const size_t third = a.ws_floats / 3;
        const int cthird = a.n_counters / 3;
        a.ws += (size_t)j * third; a.ws_floats = third;
        a.counters += (size_t)j * cthird; a.n_counters = cthird;
      };
      const float* cadd = nullptr;
      if (gvec && gin && R.cw.present) {           // x = x + cond(g)   (modules.py:214-215)
        float* cb = sc.take<float>((size_t)B * ch);
        launch_cond_gemv(gvec, nullptr, nullptr, m->W(R.cw.off), m->W(R.cb.off), cb, B, gin, ch, s);
        cadd = cb;
      }
      const float* state = u;
      if (c.resblock_type == 2) {
        // ResBlock2 (modules.py:251-262): x = conv_d(lrelu(x)) + x for d in dilations
        for (int q = 0; q < 2; ++q) {
          const int d = c.resblock_dilations[j][q];
          ConvArgs a = conv_args(m, R.c1[q], state, (int64_t)ch * Lo, Lo, r, (int64_t)ch * Lo, Lo, B, d);
          a.in_slope = kLrelu;
          a.res = state; a.res_bstride = (int64_t)ch * Lo;
          if (q == 0) { a.chan_add = cadd; a.res_chan_add = cadd; }
          if (q == 0) {
            a.epi = EPI_RESID;
          } else {
            a.epi = EPI_RESID_ACC;
            a.y = xs;
            a.accum_in = j == 0 ? nullptr : xs;
            a.out_scale = j == 2 ? (1.f / 3.f) : 1.f;
          }
          with_trim(a, Lo / Td, 0);
          launch_conv1d(a, s);
          state = r;
        }
        continue;
      }
      for (int q = 0; q < 3; ++q) {
        const int d = c.resblock_dilations[j][q];
        {
          ConvArgs a = conv_args(m, R.c1[q], state, (int64_t)ch * Lo, Lo, t1, (int64_t)ch * Lo, Lo, B, d);
          a.in_slope = kLrelu;
          if (q == 0) a.chan_add = cadd;
          own_ws(a);
          with_trim(a, Lo / Td, 0);
          launch_conv1d(a, s);
        }
        {
          ConvArgs a = conv_args(m, R.c2[q], t1, (int64_t)ch * Lo, Lo, r, (int64_t)ch * Lo, Lo, B, 1);
          a.in_slope = kLrelu;
          a.res = state; a.res_bstride = (int64_t)ch * Lo;
          if (q == 0) a.res_chan_add = cadd;
          if (q < 2) {
            a.epi = EPI_RESID;
          } else {                                   // xs (+)= resblock output ; /3 on the last
            a.epi = EPI_RESID_ACC;
            a.y = xs;
            a.accum_in = j == 0 ? nullptr : xs;
            a.out_scale = j == 2 ? (1.f / 3.f) : 1.f;
            if (conc && j > 0) HIPCHK(m, hipStreamWaitEvent(s, m->ev_rb[j - 1], 0));   // xs of the ResBlock before
          }
          own_ws(a);
          with_trim(a, Lo / Td, 0);
          launch_conv1d(a, s);
          if (conc && q == 2) HIPCHK(m, hipEventRecord(m->ev_rb[j], s));
        }
        state = r;
      }
    }
    return 0;
    };
    if (resblocks()) {
      if (conc) for (auto& st : m->aux) (void)hipStreamSynchronize(st);
      return 1;
    }
    if (conc) HIPCHK(m, hipStreamWaitEvent(s_main, m->ev_rb[2], 0));
    m->stages[i == 0 ? "dec_res_0" : "dec_res_1"] = {xs, (int64_t)n};
    cur = xs;
    L = Lo;
  }
  const int Fr = L + 1;
  const int chl = C0 >> 2;
  const int prow = sb ? 18 : 72;
  // The fused iSTFT kernels address x_post with 32-bit byte offsets: a batch whose x_post would
  // reach 2 GiB runs conv_post + iSTFT in sub-batches (same T', same kernels, per-utterance
  // arithmetic unchanged: results are bitwise those of an unsplit launch).
  const int64_t utt_bytes = (int64_t)prow * Fr * 4;
  int64_t cap_bytes = (1LL << 31) - 1;
  if (m->xpost_chunk_bytes > 0 && m->xpost_chunk_bytes < cap_bytes) cap_bytes = m->xpost_chunk_bytes;
  if (utt_bytes > (1LL << 31) - 1)
    return m->fail("utterance too long for one launch: %d frames (x_post of ONE utterance must stay below 2 GiB)", Td);
  int Bc = (int)(cap_bytes / utt_bytes);
  if (Bc < 1) Bc = 1;
  if (Bc > B) Bc = B;
  float* xpost = sc.take<float>((size_t)Bc * prow * Fr);
  float* o = outs ? outs->o : nullptr;
  float* otmp = nullptr;
  if (!o) { otmp = sc.take<float>((size_t)B * 256 * Td); o = otmp; }
  const int64_t M4 = (int64_t)(sb ? 4 : 256) * (sb ? (Fr - 1) : Td);          // waveform samples per utterance
  for (int b0 = 0; b0 < B; b0 += Bc) {
    const int nb = B - b0 < Bc ? B - b0 : Bc;
    {
      ConvArgs a = conv_args(m, m->conv_post, cur + (size_t)b0 * chl * L, (int64_t)chl * L, L, xpost,
                             (int64_t)prow * Fr, Fr, nb);
      a.in_slope = 0.01f;                              // F.leaky_relu default slope (models.py:363)
      a.reflect1 = 1;                                  // ReflectionPad1d((1,0)) (models.py:364)
      if (nb == B) with_trim(a, L / Td, 1);            // (a split run keeps every tile: the maps are per full batch)
      launch_conv1d(a, s);
    }
    if (b0 == 0) HIPCHK(m, hipEventRecord(m->evk[1], s));   // (a split run interleaves conv_post and iSTFT launches: evk_split below)
    if (sb) {
      IstftSbArgs ia{};
      ia.x_post = xpost; ia.o = o + (size_t)b0 * M4;
      ia.spec = outs && outs->spec ? outs->spec + (size_t)b0 * 9 * Fr : nullptr;
      ia.phase = outs && outs->phase ? outs->phase + (size_t)b0 * 9 * Fr : nullptr;
      ia.B = nb; ia.F = Fr; ia.exact_math = m->exact_math; ia.prescaled = 1;
      launch_istft_single(ia, s);
    } else {
      IstftArgs ia{};
      const bool ms = c.decoder == MBV_DEC_MULTISTREAM;
      ia.x_post = xpost; ia.filt = m->W(m->filt.off); ia.o = o + (size_t)b0 * M4;
      ia.o_mb = outs && outs->o_mb ? outs->o_mb + (size_t)b0 * (ms ? 1024 : 256) * Td : nullptr;
      ia.spec = outs && outs->spec ? outs->spec + (size_t)b0 * 36 * Fr : nullptr;
      ia.phase = outs && outs->phase ? outs->phase + (size_t)b0 * 36 * Fr : nullptr;
      ia.B = nb; ia.Tp = Td; ia.multistream = ms;
      ia.fixed_bank = !ia.multistream; ia.exact_math = m->exact_math; ia.prescaled = 1;
      if (trim && nb == B && !ia.o_mb && !ia.spec && !ia.phase) ia.trim_lens = zlens;
      launch_istft_pqmf(ia, s);
    }
  }
  if (Bc == B) m->stages["x_post"] = {xpost, (int64_t)B * prow * Fr};     // (a split run keeps only its last chunk)
  m->xpost_F = Fr;
  m->xpost_rows = prow;
  HIPCHK(m, hipEventRecord(m->evk[2], s));
  m->evk_set = true;
  m->evk_split = Bc < B;
  return 0;
}


// WN stack (modules.py:148-176): h -> skip; `nl` layers, gate conditioning from gvec.
// Fused path (default): one launch per layer over the units that hold valid frames, h ping-pongs
// between hbuf and acts (the old path's gated-activation buffer); afterwards only `skip` is meaningful,
// and only where the frame mask is 1 (every reader masks on load).
// Two-launch path (MBV_WN_FUSED=0; experiments: MBV_WN_SMALL=<tiles> in the low-latency mode): gate conv,
// then res/skip conv.
struct WnFold { const PConv* rspf; int Cs; float* x1; int64_t x1_bstride; float sign;
                const PConv* in16f0; const PConv* pref; int Gi; const float* x0; int in_cb; };   // in16f0 != nullptr: `pre` folded as well
// can this WN stack take the fused one-launch-per-layer kernel (run_wn's own test; run_coupling asks before it folds `post`)
bool wn_takes_fused(const mbv_model* m, const PConv* in_l, const PConv* in16_l, int B, int T) {
  const int H = m->cfg.hidden_channels;
  static const int small_units = [] { const char* e = getenv("MBV_WN_SMALL"); return e ? atoi(e) : 0; }();
  const bool small = (long)B * ((T + 31) / 32) < small_units;
  static const int wn_bf16 = [] { const char* e = getenv("MBV_WN_BF16"); return e ? atoi(e) : 1; }();
  const bool two_launch_bf16 = wn_bf16 && m->Wsplit(0) != nullptr && (long)B * T >= 12288;
  return !two_launch_bf16 && m->wn_fused && in16_l[0].M && wn_fused_supported(H, in_l[0].K) && wn_fused_fits(B, H, T) && !(m->splitk && small);
}
int run_wn(mbv_model* m, const PConv* in_l, const PConv* rs_l, const PConv* in16_l, const PConv* rsp_l, int nl,
           const PVec& cw, const PVec& cb, const float* gvec, float* hbuf, float* acts, float* skip, float* gc,
           int* ustart, const int* lens, int B, int T, hipStream_t s, const WnFold* fold = nullptr) {
  const mbv_config& c = m->cfg;
  const int H = c.hidden_channels, gin = c.gin_channels;
  const int64_t bsH = (int64_t)H * T;
  const bool cond = gvec && gin && cw.present;
  if (cond) launch_cond_gemv(gvec, nullptr, nullptr, m->W(cw.off), m->W(cb.off), gc, B, gin, 2 * H * nl, s);
  // (r02f: the fused layer wins or ties at every size measured, down to one utterance = 9 workgroups —
  // ljs_mini B=1 3.55 -> 3.11 ms, B=8 4.33 -> 3.68, ljs_mb B=8 9.69 -> 9.04, B=1 5.10 -> 5.14 — so the
  // two-launch path is only taken when MBV_WN_SMALL asks for it: launches below that many 32-frame tiles)
  static const int small_units = [] { const char* e = getenv("MBV_WN_SMALL"); return e ? atoi(e) : 0; }();
  const bool small = (long)B * ((T + 31) / 32) < small_units;
  // (split-bf16 mode: the two-launch layer on the conv kernel, which has the mode — from ~12 k frames up the
  // gate conv + res/skip conv in split-bf16 beat the fused exact layer: ljs_mb B=64 23.6 -> 22.3 ms per infer,
  // uudb B=32 15.7 -> 14.9; B=16: 8.9 -> 9.0, so smaller launches keep the fused layer.  MBV_WN_BF16=0: never)
  static const int wn_bf16 = [] { const char* e = getenv("MBV_WN_BF16"); return e ? atoi(e) : 1; }();
  const bool two_launch_bf16 = wn_bf16 && m->Wsplit(0) != nullptr && (long)B * T >= 12288;
  if (!two_launch_bf16 && m->wn_fused && in16_l[0].M && wn_fused_supported(H, in_l[0].K) && wn_fused_fits(B, H, T) && !(m->splitk && small)) {
    int* hmap = ustart + B + 1;
    launch_wn_units(lens, B, T, ustart, hmap, s);
    float* hin = hbuf;
    float* hout = acts;
    for (int l = 0; l < nl; ++l) {
      WnLayerArgs a{};
      a.h_in = hin; a.h_out = hout; a.skip = skip; a.lens = lens; a.ustart = ustart; a.hmap = hmap;
      a.wg = m->W(in16_l[l].w); a.bg = m->W(in16_l[l].bias);
      int Mg_pad = in16_l[l].Mpad;
      if (fold && fold->in16f0 && l == 0) {          // the first layer reads the x0 half of z itself
        a.h_in = fold->x0; a.in_cb = fold->in_cb; a.Gi = fold->Gi;
        a.wg = m->W(fold->in16f0->w); a.bg = m->W(fold->in16f0->bias); Mg_pad = fold->in16f0->Mpad;
        a.wpre = m->W(fold->pref->w); a.wpre_Mpad = fold->pref->Mpad;
      }
      if (cond) { a.gcond = gc + (size_t)l * 2 * H; a.gcond_bstride = 2 * H * nl; }
      const PConv& R = fold ? fold->rspf[l] : rsp_l[l];
      a.wr = m->W(R.w); a.br = m->W(R.bias);
      a.B = B; a.H = H; a.T = T;
      a.Mg_pad = Mg_pad; a.Mr = R.M; a.Mr_pad = R.Mpad;
      a.last = l == nl - 1; a.skip_accum = l > 0;
      if (fold) {
        a.Cs = fold->Cs;
        if (a.last) { a.x1 = fold->x1; a.x1_bstride = fold->x1_bstride; a.couple_sign = fold->sign; }
      }
      launch_wn_layer(a, s);
      float* tmp = hin; hin = hout; hout = tmp;
    }
    return 0;
  }
  for (int l = 0; l < nl; ++l) {
    {
      ConvArgs a = conv_args(m, in_l[l], hbuf, bsH, T, acts, bsH, T, B);
      a.epi = EPI_GATE; a.gate_half = H;
      if (cond) { a.gate_cond = gc + (size_t)l * 2 * H; a.gate_cond_bstride = 2 * H * nl; }
      launch_conv1d(a, s);
    }
    {
      ConvArgs a = conv_args(m, rs_l[l], acts, bsH, T, hbuf, bsH, T, B);
      a.epi = EPI_RES_SKIP; a.out_lens = lens; a.skip = skip;
      a.split = l < nl - 1 ? H : 0;
      a.skip_accum = l > 0;
      launch_conv1d(a, s);
    }
  }
  return 0;
}

// One ResidualCouplingLayer (modules.py:334-353) in place on z [B, I, T]; the channel Flip that
// precedes (reverse) / follows (forward) it is folded into the packing, see do_finalize.
//   reverse: x1 = (x1 - m) * mask          forward: x1 = m + x1 * mask = (x1 + m) * mask
int run_coupling(mbv_model* m, int f, bool reverse, float* z, const float* gvec, float* hbuf, float* acts,
                 float* skip, float* gc, int* ustart, const int* lens, int B, int T, hipStream_t s) {
  const mbv_config& c = m->cfg;
  const int H = c.hidden_channels, I = c.inter_channels, half = I / 2;
  const int64_t bsI = (int64_t)I * T, bsH = (int64_t)H * T;
  const auto& F = m->flow[f];
  const bool flipped = (f % 2) == 1;
  float* x0 = flipped ? z + (size_t)half * T : z;
  float* x1 = flipped ? z : z + (size_t)half * T;
  static const int fold_env = [] { const char* e = getenv("MBV_FLOW_FOLD"); return e ? atoi(e) : 3; }();   // bit 0: post, bit 1: pre
  const bool fold_post = (fold_env & 1) && F.rspf[0].M && wn_takes_fused(m, F.in, F.in16, B, T);
  const bool fold_pre = fold_post && (fold_env & 2) && F.in16f0.M && F.pref.M && wn_fused_fits(B, I, T);   // (x0 is addressed through a whole-tensor view of z)
  if (!fold_pre) {
    ConvArgs a = conv_args(m, F.pre, x0, bsI, T, hbuf, bsH, T, B);
    a.out_lens = lens;
    launch_conv1d(a, s);
  }
  if (fold_post) {
    // `post` lives in the res/skip convs (do_finalize): the last WN layer applies the coupling on the valid frames.
    // Frames at and beyond lens[b] are not touched: z arrives masked (expand_kernel / posterior_sample) and stays so.
    const WnFold fold{F.rspf, half, x1, bsI, reverse ? -1.f : 1.f,
                      fold_pre ? &F.in16f0 : nullptr, fold_pre ? &F.pref : nullptr, F.Gi, x0, I};
    return run_wn(m, F.in, F.rs, F.in16, F.rsp, kFlowLayers, F.cw, F.cb, gvec, hbuf, acts, skip, gc, ustart, lens, B, T, s, &fold);
  }
  run_wn(m, F.in, F.rs, F.in16, F.rsp, kFlowLayers, F.cw, F.cb, gvec, hbuf, acts, skip, gc, ustart, lens, B, T, s);
  {
    ConvArgs a = conv_args(m, F.post, skip, bsH, T, x1, bsI, T, B);
    a.in_lens = lens; a.epi = EPI_COUPLE; a.out_lens = lens;
    a.couple_sign = reverse ? -1.f : 1.f;
    launch_conv1d(a, s);
  }
  return 0;
}

size_t decoder_scratch_bytes(const mbv_config& c, int B, int Td) {
  const size_t C0 = c.upsample_initial_channel;
  const size_t us = c.decoder == MBV_DEC_SINGLEBAND ? 8 : 4;
  size_t n = (size_t)B * C0 * Td;                              // conv_pre
  // per stage: u, t1, r, xs — and t1, r twice more when the ResBlocks may run on three streams (small
  // launches only; sized for it whenever the tile test can pass, whatever the option says)
  for (int i = 0; i < 2; ++i) {
    const size_t ch = C0 >> (i + 1), Lo = (i == 0 ? us : us * us) * (size_t)Td;
    const long conv_tiles = (long)B * (long)((Lo + 383) / 384) * (long)((ch + 127) / 128);
    n += (conv_tiles <= 192 ? 8 : 4) * (size_t)B * ch * Lo;
  }
  {   // x_post: run_decoder takes at most the sub-batch that stays below 2 GiB (its Bc; the option can only lower it)
    const size_t utt = 72 * (us * us * Td + 1);
    size_t bc = utt ? ((size_t)(1ULL << 31) - 1) / (utt * sizeof(float)) : (size_t)B;
    if (bc < 1) bc = 1;
    n += (bc < (size_t)B ? bc : (size_t)B) * utt + (size_t)B * 256 * Td;
  }
  n += 6 * (size_t)B * C0;                                     // cond vectors
  n += 12 * ((size_t)B + 2 + (size_t)B * ((us * us * Td + 1 + 127) / 128 + 1));   // trimmed decode: column-tile maps (ints)
  return n * sizeof(float) + 64 * 256 + 32 * 256;              // + the 256-byte alignment of every take
}

}  // namespace

// ======================================================================== C ABI
extern "C" {

int mbv_abi_version(void) { return MBV_ABI_VERSION; }

const char* mbv_last_error(const mbv_model* m) { return m ? m->err.c_str() : g_create_error.c_str(); }

int mbv_create(const mbv_config* cfg, mbv_model** out) {
  if (!out) { g_create_error = "out is NULL"; return 1; }
  *out = nullptr;
  if (!cfg || cfg->struct_bytes != (int32_t)sizeof(mbv_config)) {
    g_create_error = "mbv_config.struct_bytes does not match this library (ABI mismatch)";
    return 1;
  }
  auto bad = [&](const char* why) { g_create_error = why; return 1; };
  if (cfg->n_vocab <= 0) return bad("n_vocab must be > 0");
  if (cfg->hidden_channels % 32 || cfg->inter_channels % 64 || cfg->filter_channels % 32)
    return bad("hidden/filter channels must be multiples of 32, inter_channels of 64");
  if (cfg->hidden_channels > kDpFilter) return bad("hidden_channels > 256 not supported (LayerNorm tile)");
  if (cfg->n_heads <= 0 || cfg->hidden_channels % cfg->n_heads || (cfg->hidden_channels / cfg->n_heads) % 2 ||
      cfg->hidden_channels / cfg->n_heads > 128)
    return bad("hidden_channels / n_heads must be an even integer <= 128");
  if (cfg->upsample_initial_channel % 128) return bad("upsample_initial_channel must be a multiple of 128");
  if (cfg->decoder != MBV_DEC_MULTIBAND && cfg->decoder != MBV_DEC_MULTISTREAM &&
      cfg->decoder != MBV_DEC_SINGLEBAND)
    return bad("unknown decoder");
  if (cfg->n_speakers > 1 && cfg->gin_channels <= 0) return bad("n_speakers > 1 needs gin_channels > 0");
  for (int j = 0; j < 3; ++j)
    if (cfg->resblock_kernel_sizes[j] < 1 || cfg->resblock_kernel_sizes[j] % 2 == 0)
      return bad("resblock kernel sizes must be odd");
  if (cfg->resblock_type != 1 && cfg->resblock_type != 2) return bad("resblock_type must be 1 or 2");
  for (int j = 0; j < 3; ++j)
    for (int q = 0; q < (cfg->resblock_type == 1 ? 3 : 2); ++q)
      if (cfg->resblock_dilations[j][q] < 1 ||
          !conv1d_supported(cfg->resblock_kernel_sizes[j], cfg->resblock_dilations[j][q]))
        return bad("resblock kernel size / dilation outside the built range (k <= 11, (k-1)*d <= 72)");
  if (!conv1d_supported(cfg->kernel_size, 1)) return bad("FFN kernel_size outside the built range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return bad("no HIP device visible: this library has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return bad("device ordinal out of range");
  DeviceGuard dev_guard_(cfg->device);
  if (!dev_guard_.ok) return bad("hipSetDevice failed");
  mbv_model* m = new (std::nothrow) mbv_model();
  if (!m) return bad("out of host memory");
  m->cfg = *cfg;
  { const char* e = getenv("MBV_ISTFT_EXACT"); m->exact_math = (e && e[0] == '1') ? 1 : 0; }
  { const char* e = getenv("MBV_CONV_SPLITK"); m->splitk = (e && atoi(e) != 0) ? 1 : 0; }
  { const char* e = getenv("MBV_WN_FUSED"); m->wn_fused = e ? (atoi(e) != 0) : 1; }
  build_expected(m);
  for (auto& set : m->evr)
    for (auto& e : set)
      if (hipEventCreate(&e) != hipSuccess) { delete m; return bad("hipEventCreate failed"); }
  for (auto& e : m->evk)
    if (hipEventCreate(&e) != hipSuccess) { delete m; return bad("hipEventCreate failed"); }
  m->ev_ok = true;
  { const char* e = getenv("MBV_DEC_STREAMS"); m->dec_streams = e ? (atoi(e) != 0) : 1; }
  { const char* e = getenv("MBV_CONV_BF16"); m->conv_bf16 = (e && atoi(e) == 3) ? 3 : 0; }
  {
    bool ok = hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (auto& e : m->ev_rb) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    for (auto& st : m->aux) ok = ok && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    if (!ok) { delete m; return bad("creating the decoder's auxiliary streams failed"); }
    m->aux_ok = true;
  }
  {   // split-K scratch of small conv launches: 64 MB of partials, 8192 ticket counters (zeroed once;
      // the kernel resets a counter when its last split has arrived)
    constexpr size_t kWsFloats = (size_t)16 << 20;
    constexpr int kCounters = 8192;
    if (hipMalloc((void**)&m->conv_ws, kWsFloats * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&m->conv_cnt, kCounters * sizeof(unsigned)) != hipSuccess ||
        hipMemset(m->conv_cnt, 0, kCounters * sizeof(unsigned)) != hipSuccess) {
      mbv_destroy(m);
      return bad("hipMalloc of the split-K workspace failed");
    }
    m->conv_ws_floats = kWsFloats;
    m->conv_ncnt = kCounters;
  }
  *out = m;
  return 0;
}

int mbv_set_option(mbv_model* m, const char* name, int value) {
  if (!m) return 1;
  if (!name) return m->fail("mbv_set_option: name is NULL");
  DEVICE_GUARD(m);                 // conv_bf16 allocates and launches on the model's device, whatever the caller's current one
  if (!strcmp(name, "splitk")) { m->splitk = value != 0; return 0; }
  if (!strcmp(name, "istft_exact")) { m->exact_math = value != 0; return 0; }
  if (!strcmp(name, "wn_fused")) { m->wn_fused = value != 0; return 0; }
  if (!strcmp(name, "xpost_chunk_bytes")) { m->xpost_chunk_bytes = value > 0 ? value : 0; return 0; }
  if (!strcmp(name, "dec_streams")) { m->dec_streams = value != 0; return 0; }
  if (!strcmp(name, "trim")) { m->trim = value != 0; return 0; }
  if (!strcmp(name, "conv_bf16")) {
    if (value != 0 && value != 3) return m->fail("mbv_set_option: conv_bf16 takes 0 (exact fp32) or 3 (split-bf16, three products)");
    m->conv_bf16 = value;
    if (value == 3 && m->finalized && ensure_split_arena(m, nullptr)) return 1;
    return 0;
  }
  return m->fail("mbv_set_option: unknown option '%s' (known: splitk, istft_exact, wn_fused, xpost_chunk_bytes, dec_streams, trim, conv_bf16)", name);
}

void mbv_destroy(mbv_model* m) {
  if (!m) return;
  DeviceGuard dev_guard_(m->cfg.device);
  if (m->darena) (void)hipFree(m->darena);
  if (m->darena_split) (void)hipFree(m->darena_split);
  if (m->conv_ws) (void)hipFree(m->conv_ws);
  if (m->conv_cnt) (void)hipFree(m->conv_cnt);
  if (m->scrA) (void)hipFree(m->scrA);
  if (m->scrB) (void)hipFree(m->scrB);
  if (m->user_tab) (void)hipFree(m->user_tab);
  if (m->peak_buf) (void)hipFree(m->peak_buf);
  if (m->ev_ok) { for (auto& set : m->evr) for (auto& e : set) if (e) (void)hipEventDestroy(e); for (auto& e : m->evk) (void)hipEventDestroy(e); }
  if (m->aux_ok) {
    for (auto& st : m->aux) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    (void)hipEventDestroy(m->ev_fork);
    for (auto& e : m->ev_rb) (void)hipEventDestroy(e);
  }
  delete m;
}

int mbv_load_weight(mbv_model* m, const char* name, const float* data, const int64_t* shape, int ndim) {
  if (!m) return 1;
  if (!name || !data || !shape || ndim <= 0) return m->fail("mbv_load_weight: NULL argument");
  auto it = m->expected.find(name);
  if (it == m->expected.end())
    return m->fail("'%s' is not a weight of the infer path (enc_q.* and training-only keys are not accepted)", name);
  const auto& want = it->second;
  bool ok = (int)want.size() == ndim;
  for (int i = 0; ok && i < ndim; ++i) ok = want[i] == shape[i];
  if (!ok) {
    std::string w, g;
    for (auto v : want) w += std::to_string(v) + ",";
    for (int i = 0; i < ndim; ++i) g += std::to_string(shape[i]) + ",";
    return m->fail("shape mismatch for '%s': expected [%s] got [%s]", name, w.c_str(), g.c_str());
  }
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(data, data + t.numel());
  m->raw[name] = std::move(t);
  m->finalized = false;
  return 0;
}

int mbv_missing_weights(mbv_model* m, char* buf, size_t cap) {
  if (!m) return -1;
  int n = 0;
  std::string list;
  for (auto& kv : m->expected)
    if (!m->raw.count(kv.first)) { ++n; list += kv.first; list += ","; }
  if (buf && cap) { strncpy(buf, list.c_str(), cap - 1); buf[cap - 1] = 0; }
  return n;
}

int mbv_finalize_weights(mbv_model* m, void* stream) {
  if (!m) return 1;
  DEVICE_GUARD(m);
  return do_finalize(m, (hipStream_t)stream);
}

int64_t mbv_arena_floats(mbv_model* m) {
  if (!m) return -1;
  if (!m->finalized) { m->fail("weights not finalized"); return -1; }
  return (int64_t)m->arena_used;
}

int mbv_export_arena(mbv_model* m, float* dst, int64_t capacity, void* stream) {
  if (!m) return 1;
  if (!m->finalized) return m->fail("weights not finalized");
  if (!dst || capacity < (int64_t)m->arena_used) return m->fail("mbv_export_arena: destination holds %lld floats, the arena %zu", (long long)capacity, m->arena_used);
  DEVICE_GUARD(m);
  HIPCHK(m, hipMemcpyAsync(dst, m->darena, m->arena_used * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int mbv_import_arena(mbv_model* m, const float* src, int64_t n_floats, void* stream) {
  if (!m) return 1;
  if (!src || n_floats <= 0) return m->fail("mbv_import_arena: bad arguments");
  DEVICE_GUARD(m);
  // the arena's layout is a function of the configuration alone (every offset comes from a tensor SHAPE): the same
  // packing pass runs over zero tensors, skipping the weight-norm folds, and the contents arrive from `src`
  m->import_src = src;
  m->import_n = n_floats;
  m->finalized = false;
  const int rc = do_finalize(m, (hipStream_t)stream);
  m->import_src = nullptr;
  if (rc) m->raw.clear();
  return rc;
}

int mbv_speaker_embedding(mbv_model* m, const int64_t* sid, int B, float* out, void* stream) {
  if (!m) return 1;
  if (!m->finalized) return m->fail("weights not finalized");
  if (!m->emb_g.present) return m->fail("model has no speaker embedding (n_speakers <= 1)");
  DEVICE_GUARD(m);
  launch_gather_rows(m->W(m->emb_g.off), sid, out, B, m->cfg.gin_channels, m->cfg.n_speakers, nullptr,
                     (hipStream_t)stream);
  HIPCHK(m, hipGetLastError());
  return 0;
}

namespace {
// DDSConv (modules.py:98-111) on x [B, C, T] in place; t1, t2: scratch of the same size
void run_dds(mbv_model* m, const mbv_model::Dds& d, float* x, float* t1, float* t2, int B, int C, int T,
             hipStream_t s) {
  const int64_t bs = (int64_t)C * T;
  int dil = 1;
  for (int i = 0; i < 3; ++i, dil *= 3) {
    launch_dds_sep(x, m->lens32, m->W(d.sw[i].off), m->W(d.sb[i].off), m->W(d.g1[i].off),
                   m->W(d.b1[i].off), t1, B, C, T, 3, dil, s);
    launch_conv1d(conv_args(m, d.c1[i], t1, bs, T, t2, bs, T, B), s);
    launch_dds_res(t2, x, m->W(d.g2[i].off), m->W(d.b2[i].off), x, B, C, T,
                   i == 2 ? m->lens32 : nullptr, s);
  }
}
}  // namespace

int mbv_encode(mbv_model* m, const int64_t* ids, const int64_t* lengths, const int64_t* sid, int B,
               int T, float length_scale, const float* noise_w, float noise_scale_w,
               int64_t* y_lengths_out, void* stream) {
  if (!m) return 1;
  if (!m->finalized) return m->fail("weights not finalized (call mbv_finalize_weights)");
  if (!ids || !lengths || B <= 0 || T <= 0) return m->fail("mbv_encode: bad arguments");
  const mbv_config& c = m->cfg;
  if (c.n_speakers > 0 && !sid) return m->fail("sid is required when n_speakers > 0 (models.py:704-705)");
  if (c.n_speakers > 0 && !m->emb_g.present) return m->fail("n_speakers == 1: the reference has no emb_g either");
  DEVICE_GUARD(m);
  // The text encoder and the duration predictor ALWAYS run exact: the durations (ceil of an exponential) must
  // not depend on the opt-in split-bf16 mode, which is for the waveform path only.  (Long texts, T > 256,
  // use the conv kernels that have the mode; short ones the narrow kernel, which does not.)
  struct ExactScope {
    mbv_model* m; int saved;
    explicit ExactScope(mbv_model* mm) : m(mm), saved(mm->conv_bf16) { m->conv_bf16 = 0; }
    ~ExactScope() { m->conv_bf16 = saved; }
  } exact_scope(m);
  hipStream_t s = (hipStream_t)stream;
  const int H = c.hidden_channels, I = c.inter_channels, Fc = c.filter_channels, gin = c.gin_channels;
  const size_t BT = (size_t)B * T;
  size_t need = (BT * (H * 5 + 3 * H + Fc + 2 * I + 2 * kDpFilter + 4 + 5 * H + 32 + 2) + (size_t)B * (gin + H + 12)) * 4 + 96 * 256;
  if (ensure(m, &m->scrA, &m->scrA_bytes, need)) return 1;
  Bump sc{m->scrA, m->scrA_bytes};
  float* x = sc.take<float>(BT * H);
  float* x1 = sc.take<float>(BT * H);
  float* qkv = sc.take<float>(BT * 3 * H);
  float* att = sc.take<float>(BT * H);
  float* y = sc.take<float>(BT * H);
  float* ffn = sc.take<float>(BT * Fc);
  m->stats = sc.take<float>(BT * 2 * I);
  float* h1 = sc.take<float>(BT * kDpFilter);
  float* h2 = sc.take<float>(BT * kDpFilter);
  m->logw = sc.take<float>(BT);
  m->w_ceil = sc.take<float>(BT);
  m->cum = sc.take<int>(BT);
  m->lens32 = sc.take<int>(B);
  m->ylen32 = sc.take<int>(B);
  int* bad = sc.take<int>(B);
  m->gvec = sc.take<float>((size_t)B * (gin ? gin : 1));
  float* dpc = sc.take<float>((size_t)B * H);
  m->stages.clear();

  ++m->ticket;                                     // a new call: its own set of stage events (mbv_stage_times_ms_at)
  m->ev = m->evr[m->ticket % mbv_model::kEvRing];
  m->evr_a[m->ticket % mbv_model::kEvRing] = false;
  m->evr_b[m->ticket % mbv_model::kEvRing] = false;
  HIPCHK(m, hipEventRecord(m->ev[0], s));
  launch_embed(ids, lengths, m->W(m->emb.off), x, m->lens32, bad, B, T, H, c.n_vocab, s);
  const int64_t bsH = (int64_t)H * T;
  static const int fuse_ln_env = [] { const char* e = getenv("MBV_FUSE_LN"); return e ? atoi(e) : 1; }();
  // (a rule on T alone: rows stay batch-independent; the opt-in low-latency mode may look at the launch
  // size: fused, a conv + LayerNorm is one workgroup per 32 frames walking the whole K loop alone)
  const bool fuse_ln = fuse_ln_env && T <= 256 && !(m->splitk && (long)B * ((T + 15) / 16) < 128);
  for (int i = 0; i < c.n_layers; ++i) {
    const auto& L = m->enc[i];
    launch_conv1d(conv_args(m, L.qkv, x, bsH, T, qkv, 3 * bsH, T, B), s);
    launch_rel_attention(qkv, m->W(L.ek.off), m->W(L.ev.off), m->lens32, att, B, H, c.n_heads, T, s);
    // conv_o and the LayerNorm(x + y) behind it (attentions.py:40-41): one launch on the narrow kernel
    // for sequences it covers (a rule on T and H only), else conv + LayerNorm
    {
      ConvArgs a = conv_args(m, L.o, att, bsH, T, x1, bsH, T, B);
      a.epi = EPI_LN; a.res = x; a.res_bstride = bsH;
      a.ln_gamma = m->W(L.g1.off); a.ln_beta = m->W(L.b1.off);
      if (fuse_ln && conv1d_narrow_supported(a)) {
        launch_conv1d(a, s);
      } else {
        launch_conv1d(conv_args(m, L.o, att, bsH, T, y, bsH, T, B), s);
        launch_layernorm(x, y, m->W(L.g1.off), m->W(L.b1.off), x1, B, H, T, 0, nullptr, s);
      }
    }
    {
      ConvArgs a = conv_args(m, L.ffn1, x1, bsH, T, ffn, (int64_t)Fc * T, T, B);
      a.pad_left = (c.kernel_size - 1) / 2;            // attentions.py:296-303
      a.in_lens = m->lens32; a.relu = 1;
      launch_conv1d(a, s);
    }
    const bool last = i == c.n_layers - 1;
    {   // FFN's second conv (output masked, attentions.py:303) and LayerNorm(x1 + y) (+ the final mask, :45-46)
      ConvArgs a = conv_args(m, L.ffn2, ffn, (int64_t)Fc * T, T, x, bsH, T, B);
      a.pad_left = (c.kernel_size - 1) / 2;
      a.in_lens = m->lens32; a.out_lens = m->lens32;
      a.epi = EPI_LN; a.res = x1; a.res_bstride = bsH;
      a.ln_gamma = m->W(L.g2.off); a.ln_beta = m->W(L.b2.off);
      a.ln_out_lens = last ? m->lens32 : nullptr;
      if (fuse_ln && conv1d_narrow_supported(a)) {
        launch_conv1d(a, s);
      } else {
        a.epi = EPI_STORE; a.res = nullptr; a.y = y; a.ln_gamma = a.ln_beta = nullptr; a.ln_out_lens = nullptr;
        launch_conv1d(a, s);
        launch_layernorm(x1, y, m->W(L.g2.off), m->W(L.b2.off), x, B, H, T, 0, last ? m->lens32 : nullptr, s);
      }
    }
  }
  m->x_enc = x;
  {
    ConvArgs a = conv_args(m, m->enc_proj, x, bsH, T, m->stats, (int64_t)2 * I * T, T, B);
    a.out_lens = m->lens32;
    launch_conv1d(a, s);
  }
  HIPCHK(m, hipEventRecord(m->ev[1], s));

  // ---- speaker embedding + duration predictor (models.py:704-713)
  m->has_g = c.n_speakers > 0;
  const float* cadd = nullptr;
  if (m->has_g) {
    launch_gather_rows(m->W(m->emb_g.off), sid, m->gvec, B, gin, c.n_speakers, bad, s);
    if (m->dp_cw.present) {
      launch_cond_gemv(m->gvec, nullptr, nullptr, m->W(m->dp_cw.off), m->W(m->dp_cb.off), dpc, B, gin, H, s);
      cadd = dpc;
    }
  }
  if (c.use_sdp) {
    // models.py:53-60: conditioning trunk; :89-100: z through [Flip, ConvFlow] x 3, Flip, affine
    float* cond = sc.take<float>(BT * H);
    float* hh = sc.take<float>(BT * H);
    float* t1 = sc.take<float>(BT * H);
    float* t2 = sc.take<float>(BT * H);
    float* h29 = sc.take<float>(BT * 32);
    float* zf = sc.take<float>(BT * 2);
    launch_conv1d(conv_args(m, m->sdp.pre, x, bsH, T, hh, bsH, T, B), s);
    if (cadd) launch_chan_add(hh, cadd, B, H, T, s);
    run_dds(m, m->sdp.dds, hh, t1, t2, B, H, T, s);
    {
      ConvArgs a = conv_args(m, m->sdp.proj, hh, bsH, T, cond, bsH, T, B);
      a.out_lens = m->lens32;
      launch_conv1d(a, s);
    }
    launch_sdp_noise(noise_w, noise_scale_w, zf, (int64_t)BT * 2, s);
    for (int k = 0; k < 3; ++k) {
      const auto& f = m->sdp.flow[k];
      launch_sdp_pre(zf, 1, m->W(f.pre_w.off), m->W(f.pre_b.off), cond, hh, B, H, T, s);   // x0 = z[:, 1] after the Flip
      run_dds(m, f.dds, hh, t1, t2, B, H, T, s);
      {
        ConvArgs a = conv_args(m, f.proj, hh, bsH, T, h29, (int64_t)29 * T, T, B);
        a.out_lens = m->lens32;
        launch_conv1d(a, s);
      }
      launch_sdp_spline(h29, zf, m->lens32, B, H, T, m->sdp.edge_const, s);
    }
    launch_sdp_logw(zf, m->W(m->sdp.m.off), m->W(m->sdp.logs.off), m->lens32, h29, B, T, s);
    launch_durations(h29, nullptr, nullptr, m->lens32, length_scale, m->logw, m->w_ceil, m->cum,
                     m->ylen32, y_lengths_out, bad, B, 1, T, s);
    m->stages["sdp_cond"] = {cond, (int64_t)BT * H};
    m->stages["sdp_z"] = {zf, (int64_t)BT * 2};
  } else {
  // conv -> relu -> LayerNorm twice (models.py:128-135): each pair one launch where the narrow kernel applies
  const float* dp_out = h2;
  {
    ConvArgs a = conv_args(m, m->dp1, x, bsH, T, h2, (int64_t)kDpFilter * T, T, B);
    a.in_lens = m->lens32; a.chan_add = cadd;
    a.epi = EPI_LN; a.relu = 1; a.ln_gamma = m->W(m->dp_g1.off); a.ln_beta = m->W(m->dp_b1.off);
    if (fuse_ln && conv1d_narrow_supported(a)) {
      launch_conv1d(a, s);
    } else {
      a.epi = EPI_STORE; a.relu = 0; a.y = h1; a.ln_gamma = a.ln_beta = nullptr;
      launch_conv1d(a, s);
      launch_layernorm(h1, nullptr, m->W(m->dp_g1.off), m->W(m->dp_b1.off), h2, B, kDpFilter, T, 1, nullptr, s);
    }
  }
  {
    ConvArgs a = conv_args(m, m->dp2, h2, (int64_t)kDpFilter * T, T, h1, (int64_t)kDpFilter * T, T, B);
    a.in_lens = m->lens32;
    a.epi = EPI_LN; a.relu = 1; a.ln_gamma = m->W(m->dp_g2.off); a.ln_beta = m->W(m->dp_b2.off);
    if (fuse_ln && conv1d_narrow_supported(a)) {
      launch_conv1d(a, s);
      dp_out = h1;
    } else {
      a.epi = EPI_STORE; a.relu = 0; a.ln_gamma = a.ln_beta = nullptr;
      launch_conv1d(a, s);
      launch_layernorm(h1, nullptr, m->W(m->dp_g2.off), m->W(m->dp_b2.off), h2, B, kDpFilter, T, 1, nullptr, s);
    }
  }
  launch_durations(dp_out, m->W(m->dp_pw.off), m->W(m->dp_pb.off), m->lens32, length_scale, m->logw,
                   m->w_ceil, m->cum, m->ylen32, y_lengths_out, bad, B, kDpFilter, T, s);
  }
  HIPCHK(m, hipEventRecord(m->ev[2], s));
  HIPCHK(m, hipGetLastError());
  m->B = B; m->T = T; m->encoded = true; m->ev_a = true; m->ev_b = false;
  m->evr_a[m->ticket % mbv_model::kEvRing] = true;
  m->stages["x_enc"] = {x, (int64_t)BT * H};
  m->stages["stats"] = {m->stats, (int64_t)BT * 2 * I};
  m->stages["logw"] = {m->logw, (int64_t)BT};
  m->stages["w_ceil"] = {m->w_ceil, (int64_t)BT};
  return 0;
}

int mbv_synthesize(mbv_model* m, int t_frames, const float* noise, float noise_scale, int max_len,
                   const mbv_outputs* outs, void* stream) {
  if (!m) return 1;
  if (!m->encoded) return m->fail("mbv_synthesize without a preceding mbv_encode");
  if (t_frames <= 0) return m->fail("t_frames must be > 0");
  const mbv_config& c = m->cfg;
  DEVICE_GUARD(m);
  hipStream_t s = (hipStream_t)stream;
  const int B = m->B, T = m->T, Tp = t_frames, H = c.hidden_channels, I = c.inter_channels;
  const int gin = c.gin_channels;
  const int Td = (max_len > 0 && max_len < Tp) ? max_len : Tp;
  const size_t BTp = (size_t)B * Tp;
  const bool run_dec = outs && (outs->o || outs->o_mb || outs->spec || outs->phase);
  size_t need = (BTp * (4 * I + 3 * H) + (size_t)B * (2 * H * kFlowLayers + 2)) * 4 + wn_units_ints(B, Tp) * 4 + 64 * 256 +
                decoder_scratch_bytes(c, B, Td);
  if (ensure(m, &m->scrB, &m->scrB_bytes, need)) return 1;
  Bump sc{m->scrB, m->scrB_bytes};
  for (const char* k : {"dec_conv_pre", "dec_up_0", "dec_up_1", "dec_res_0", "dec_res_1", "x_post"})
    m->stages.erase(k);
  float* z = outs && outs->z ? outs->z : sc.take<float>(BTp * I);
  float* hbuf = sc.take<float>(BTp * H);
  float* acts = sc.take<float>(BTp * H);
  float* skip = sc.take<float>(BTp * H);
  float* gc = sc.take<float>((size_t)B * 2 * H * kFlowLayers);
  int* ustart = sc.take<int>(wn_units_ints(B, Tp));
  (void)gin;

  HIPCHK(m, hipEventRecord(m->ev[3], s));
  // m_text / logs_text are the two halves of enc_p.proj's output [B, 2I, T]
  launch_expand(m->stats, m->stats + (size_t)I * T, (int64_t)2 * I * T, m->cum, m->ylen32,
                noise_scale != 0.f ? noise : nullptr, noise_scale,
                outs ? outs->m_p : nullptr, outs ? outs->logs_p : nullptr, outs ? outs->z_p : nullptr, z,
                outs ? outs->attn : nullptr, outs ? outs->y_mask : nullptr, B, I, T, Tp, s);
  HIPCHK(m, hipEventRecord(m->ev[4], s));

  // ---- reverse flows, in place on z (models.py:207-214, modules.py:334-353)
  for (int f = kNFlows - 1; f >= 0; --f)
    run_coupling(m, f, true, z, m->has_g ? m->gvec : nullptr, hbuf, acts, skip, gc, ustart, m->ylen32, B, Tp, s);
  HIPCHK(m, hipEventRecord(m->ev[5], s));
  if (run_dec) {
    if (run_decoder(m, z, Tp, m->ylen32, m->has_g ? m->gvec : nullptr, B, Td, outs, s, sc)) return 1;
  }
  HIPCHK(m, hipEventRecord(m->ev[6], s));
  HIPCHK(m, hipGetLastError());
  m->ev_b = true;
  m->evr_b[m->ticket % mbv_model::kEvRing] = true;
  return 0;
}

int mbv_decode(mbv_model* m, const float* z, const float* g, int B, int t_frames,
               const mbv_outputs* outs, void* stream) {
  if (!m) return 1;
  if (!m->finalized) return m->fail("weights not finalized");
  if (!z || B <= 0 || t_frames <= 0 || !outs) return m->fail("mbv_decode: bad arguments");
  const mbv_config& c = m->cfg;
  DEVICE_GUARD(m);
  if (ensure(m, &m->scrB, &m->scrB_bytes, decoder_scratch_bytes(c, B, t_frames))) return 1;
  Bump sc{m->scrB, m->scrB_bytes};
  m->stages.clear();
  if (run_decoder(m, z, t_frames, nullptr, (g && c.gin_channels) ? g : nullptr, B, t_frames, outs,
                  (hipStream_t)stream, sc))
    return 1;
  HIPCHK(m, hipGetLastError());
  return 0;
}

int mbv_stage_times_ms(mbv_model* m, float out[5]) {
  if (!m || !out) return 1;
  DEVICE_GUARD(m);
  if (!m->ev_a || !m->ev_b) return m->fail("no completed encode+synthesize pair to time");
  HIPCHK(m, hipEventSynchronize(m->ev[6]));
  HIPCHK(m, hipEventElapsedTime(&out[0], m->ev[0], m->ev[1]));
  HIPCHK(m, hipEventElapsedTime(&out[1], m->ev[1], m->ev[2]));
  HIPCHK(m, hipEventElapsedTime(&out[2], m->ev[3], m->ev[4]));
  HIPCHK(m, hipEventElapsedTime(&out[3], m->ev[4], m->ev[5]));
  HIPCHK(m, hipEventElapsedTime(&out[4], m->ev[5], m->ev[6]));
  return 0;
}

int64_t mbv_ticket(mbv_model* m) { return m ? m->ticket : -1; }

int mbv_stage_times_ms_at(mbv_model* m, int64_t ticket, float out[5]) {
  if (!m || !out) return 1;
  DEVICE_GUARD(m);
  if (ticket <= 0 || ticket > m->ticket) return m->fail("mbv_stage_times_ms_at: ticket %lld was never issued", (long long)ticket);
  if (ticket <= m->ticket - mbv_model::kEvRing)
    return m->fail("mbv_stage_times_ms_at: the events of call %lld were reused (%d calls are kept)", (long long)ticket, mbv_model::kEvRing);
  const int slot = (int)(ticket % mbv_model::kEvRing);
  if (!m->evr_a[slot] || !m->evr_b[slot]) return m->fail("call %lld has no completed encode+synthesize pair to time", (long long)ticket);
  hipEvent_t* ev = m->evr[slot];
  HIPCHK(m, hipEventSynchronize(ev[6]));
  HIPCHK(m, hipEventElapsedTime(&out[0], ev[0], ev[1]));
  HIPCHK(m, hipEventElapsedTime(&out[1], ev[1], ev[2]));
  HIPCHK(m, hipEventElapsedTime(&out[2], ev[3], ev[4]));
  HIPCHK(m, hipEventElapsedTime(&out[3], ev[4], ev[5]));
  HIPCHK(m, hipEventElapsedTime(&out[4], ev[5], ev[6]));
  return 0;
}

int mbv_kernel_times_ms(mbv_model* m, float out[2]) {
  if (!m || !out) return 1;
  DEVICE_GUARD(m);
  if (!m->evk_set) return m->fail("no decoder run to time");
  if (m->evk_split)
    return m->fail("kernel times unavailable: the last decoder run split its batch into sub-batches (x_post >= 2 GiB or "
                   "xpost_chunk_bytes), so conv_post and iSTFT launches interleave");
  HIPCHK(m, hipEventSynchronize(m->evk[2]));
  HIPCHK(m, hipEventElapsedTime(&out[0], m->evk[0], m->evk[1]));
  HIPCHK(m, hipEventElapsedTime(&out[1], m->evk[1], m->evk[2]));
  return 0;
}

int mbv_istft_pqmf(mbv_model* m, const float* x_post, int B, int t_frames, const float* filter,
                   int multistream, float* o, float* o_mb, float* spec, float* phase, void* stream) {
  if (!m) return 1;
  if (!x_post || !o || B <= 0 || t_frames <= 0) return m->fail("mbv_istft_pqmf: bad arguments");
  if ((int64_t)B * 72 * (16 * (int64_t)t_frames + 1) * 4 >= (1LL << 31))
    return m->fail("mbv_istft_pqmf: B * T' too large for one launch (x_post must stay below 2 GiB)");
  DEVICE_GUARD(m);
  hipStream_t s = (hipStream_t)stream;
  float*& d_tab = m->user_tab;
  if (!d_tab) HIPCHK(m, hipMalloc((void**)&d_tab, kFiltTable * sizeof(float)));
  if (filter || !m->user_tab_is_pqmf) {     // the default PQMF table is uploaded once, then launch-only
    std::vector<float> h63(4 * 63);
    if (filter) {
      HIPCHK(m, hipMemcpyAsync(h63.data(), filter, h63.size() * 4, hipMemcpyDeviceToHost, s));
      HIPCHK(m, hipStreamSynchronize(s));
    } else {
      h63 = pqmf_synthesis_filter();
    }
    const std::vector<float> tab = synthesis_table(h63.data());
    HIPCHK(m, hipMemcpyAsync(d_tab, tab.data(), kFiltTable * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(m, hipStreamSynchronize(s));
    m->user_tab_is_pqmf = filter == nullptr;
  }
  IstftArgs a{};
  a.x_post = x_post; a.filt = d_tab; a.o = o; a.o_mb = o_mb; a.spec = spec; a.phase = phase;
  a.B = B; a.Tp = t_frames; a.multistream = multistream & 1;
  a.fixed_bank = filter == nullptr; a.exact_math = m->exact_math; a.prescaled = (multistream >> 1) & 1;
  launch_istft_pqmf(a, s);
  HIPCHK(m, hipGetLastError());
  return 0;
}

int mbv_voice_conversion(mbv_model* m, const float* y, const int64_t* y_lengths, const int64_t* sid_src,
                         const int64_t* sid_tgt, int B, int T, const float* noise, const mbv_outputs* outs,
                         int32_t* status, void* stream) {
  if (!m) return 1;
  if (!m->finalized) return m->fail("weights not finalized");
  const mbv_config& c = m->cfg;
  if (c.n_speakers <= 0 || !m->emb_g.present)
    return m->fail("n_speakers have to be larger than 0.");              // models.py:791 assert
  if (!y || !y_lengths || !sid_src || !sid_tgt || !outs || B <= 0 || T <= 0)
    return m->fail("mbv_voice_conversion: bad arguments");
  DEVICE_GUARD(m);
  hipStream_t s = (hipStream_t)stream;
  const int H = c.hidden_channels, I = c.inter_channels, gin = c.gin_channels, SC = c.spec_channels;
  const auto& Q = m->encq;
  const size_t BT = (size_t)B * T;
  size_t need = (BT * ((size_t)Q.cin_pad + 3 * H + 4 * I) + (size_t)B * (2 * gin + 2 * H * mbv_model::kEncQLayers + 18)) * 4 + wn_units_ints(B, T) * 4 +
                64 * 256 + decoder_scratch_bytes(c, B, T);
  if (ensure(m, &m->scrB, &m->scrB_bytes, need)) return 1;
  Bump sc{m->scrB, m->scrB_bytes};
  m->stages.clear();
  float* ypad = sc.take<float>(BT * Q.cin_pad);
  float* hbuf = sc.take<float>(BT * H);
  float* acts = sc.take<float>(BT * H);
  float* skip = sc.take<float>(BT * H);
  float* stats = sc.take<float>(BT * 2 * I);
  float* zbuf = outs->z ? outs->z : sc.take<float>(BT * I);
  float* zhat = outs->m_p ? outs->m_p : sc.take<float>(BT * I);
  float* g_src = sc.take<float>((size_t)B * gin);
  float* g_tgt = sc.take<float>((size_t)B * gin);
  float* gc = sc.take<float>((size_t)B * 2 * H * mbv_model::kEncQLayers);
  int* lens = sc.take<int>(B);
  int* bad = sc.take<int>(B);
  int* ustart = sc.take<int>(wn_units_ints(B, T));

  launch_lens_to_i32(y_lengths, lens, B, T, bad, s);
  launch_gather_rows(m->W(m->emb_g.off), sid_src, g_src, B, gin, c.n_speakers, bad, s);
  launch_gather_rows(m->W(m->emb_g.off), sid_tgt, g_tgt, B, gin, c.n_speakers, bad, s);
  if (status) HIPCHK(m, hipMemcpyAsync(status, bad, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s));
  // the 1x1 `pre` conv reads channel groups of 32: zero-pad spec_channels (513) to a multiple of 32
  launch_fill(ypad, 0.f, (int64_t)BT * Q.cin_pad, s);
  HIPCHK(m, hipMemcpy2DAsync(ypad, (size_t)Q.cin_pad * T * 4, y, (size_t)SC * T * 4, (size_t)SC * T * 4, B,
                             hipMemcpyDeviceToDevice, s));
  const int64_t bsH = (int64_t)H * T;
  {   // enc_q (models.py:239-246): pre * mask -> WN(g_src) -> proj * mask -> sample
    ConvArgs a = conv_args(m, Q.pre, ypad, (int64_t)Q.cin_pad * T, T, hbuf, bsH, T, B);
    a.out_lens = lens;
    launch_conv1d(a, s);
  }
  run_wn(m, Q.in, Q.rs, Q.in16, Q.rsp, mbv_model::kEncQLayers, Q.cw, Q.cb, g_src, hbuf, acts, skip, gc, ustart, lens, B, T, s);
  {
    ConvArgs a = conv_args(m, Q.proj, skip, bsH, T, stats, (int64_t)2 * I * T, T, B);
    a.in_lens = lens; a.out_lens = lens;
    launch_conv1d(a, s);
  }
  launch_posterior_sample(stats, noise, lens, zbuf, B, I, T, s);
  // forward flow with the source speaker (models.py:795), then reverse with the target (:796)
  HIPCHK(m, hipMemcpyAsync(zhat, zbuf, BT * I * 4, hipMemcpyDeviceToDevice, s));
  for (int f = 0; f < kNFlows; ++f)
    run_coupling(m, f, false, zhat, g_src, hbuf, acts, skip, gc, ustart, lens, B, T, s);
  if (outs->z_p) HIPCHK(m, hipMemcpyAsync(outs->z_p, zhat, BT * I * 4, hipMemcpyDeviceToDevice, s));
  for (int f = kNFlows - 1; f >= 0; --f)
    run_coupling(m, f, true, zhat, g_tgt, hbuf, acts, skip, gc, ustart, lens, B, T, s);
  if (outs->y_mask) launch_sequence_mask(lens, outs->y_mask, B, T, s);
  if (run_decoder(m, zhat, T, lens, g_tgt, B, T, outs, s, sc)) return 1;
  HIPCHK(m, hipGetLastError());
  return 0;
}

int mbv_istft_finalize(mbv_model* m, const float* spec, const float* phase, int B, int frames,
                       float* o, float* o_mb, void* stream) {
  if (!m) return 1;
  if (!m->finalized) return m->fail("weights not finalized");
  if (!spec || !phase || !o || B <= 0 || frames < 2) return m->fail("mbv_istft_finalize: bad arguments");
  const mbv_config& c = m->cfg;
  DEVICE_GUARD(m);
  hipStream_t s = (hipStream_t)stream;
  if (c.decoder == MBV_DEC_SINGLEBAND) {
    IstftSbArgs a{};
    a.o = o; a.spec = const_cast<float*>(spec); a.phase = const_cast<float*>(phase);
    a.B = B; a.F = frames; a.exact_math = m->exact_math; a.polar_in = 1;
    launch_istft_single(a, s);
  } else {
    if ((frames - 1) % 16) return m->fail("mbv_istft_finalize: frames must be 16 n + 1 for the 4-band decoders");
    if ((int64_t)B * 36 * frames * 4 >= (1LL << 31)) return m->fail("mbv_istft_finalize: tensor too large for one launch");
    IstftArgs a{};
    a.filt = m->W(m->filt.off); a.o = o; a.o_mb = o_mb;
    a.spec = const_cast<float*>(spec); a.phase = const_cast<float*>(phase);
    a.B = B; a.Tp = (frames - 1) / 16; a.multistream = c.decoder == MBV_DEC_MULTISTREAM;
    a.fixed_bank = !a.multistream; a.exact_math = m->exact_math; a.polar_in = 1;
    launch_istft_pqmf(a, s);
  }
  HIPCHK(m, hipGetLastError());
  return 0;
}

int mbv_pcm16(mbv_model* m, const float* wave, const int64_t* y_lengths, int B, int64_t stride,
              int auto_normalize, int16_t* pcm, void* stream) {
  if (!m) return 1;
  if (!wave || !pcm || B <= 0 || stride <= 0) return m->fail("mbv_pcm16: bad arguments");
  DEVICE_GUARD(m);
  if (m->peak_cap < B) {
    if (m->peak_buf) HIPCHK(m, hipFree(m->peak_buf));
    HIPCHK(m, hipMalloc((void**)&m->peak_buf, (size_t)B * sizeof(unsigned)));
    m->peak_cap = B;
  }
  launch_pcm16(wave, y_lengths, B, stride, 256, auto_normalize, m->peak_buf, reinterpret_cast<short*>(pcm),
               (hipStream_t)stream);
  HIPCHK(m, hipGetLastError());
  return 0;
}

int64_t mbv_read_stage(mbv_model* m, const char* name, float* dst, int64_t capacity, void* stream) {
  if (!m || !name) return -1;
  const mbv_config& c = m->cfg;
  DeviceGuard dev_guard_(c.device);
  if (!dev_guard_.ok) { m->fail("hipSetDevice(%d) failed", c.device); return -1; }
  std::string n(name);
  const float* src = nullptr;
  int64_t numel = 0;
  // m_text / logs_text are strided halves of `stats` [B, 2I, T]
  if (n == "m_text" || n == "logs_text") {
    auto it = m->stages.find("stats");
    if (it == m->stages.end()) { m->fail("stage '%s' not available", name); return -1; }
    const int I = c.inter_channels;
    numel = (int64_t)m->B * I * m->T;
    if (!dst) return numel;
    if (capacity < numel) { m->fail("capacity too small"); return -1; }
    const float* base = it->second.ptr + (n == "logs_text" ? (size_t)I * m->T : 0);
    if (hipMemcpy2DAsync(dst, (size_t)I * m->T * 4, base, (size_t)2 * I * m->T * 4, (size_t)I * m->T * 4,
                         m->B, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
      m->fail("hipMemcpy2DAsync failed"); return -1;
    }
    return numel;
  }
  auto it = m->stages.find(n);
  if (it == m->stages.end()) { m->fail("stage '%s' not available", name); return -1; }
  src = it->second.ptr; numel = it->second.numel;
  if (!dst) return numel;
  if (capacity < numel) { m->fail("capacity too small"); return -1; }
  if (n == "x_post") {     // stored pre-scaled for the iSTFT kernel: hand back the reference's units
    launch_unscale_xpost(src, dst, (int)(numel / (m->xpost_rows * (int64_t)m->xpost_F)), m->xpost_rows, m->xpost_F, (hipStream_t)stream);
    return numel;
  }
  if (hipMemcpyAsync(dst, src, numel * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
    m->fail("hipMemcpyAsync failed"); return -1;
  }
  return numel;
}

int mbv_op_rel_attention(mbv_model* m, const float* qkv, const float* emb_k, const float* emb_v, const int64_t* lengths,
                         float* o, int B, int H, int n_heads, int T, void* stream) {
  if (!m) return 1;
  if (!qkv || !emb_k || !emb_v || !lengths || !o || B <= 0 || T <= 0 || n_heads <= 0 || H % n_heads || (H / n_heads) % 2 || H / n_heads > 96)
    return m->fail("mbv_op_rel_attention: bad arguments (head dimension must be even and <= 96)");
  DEVICE_GUARD(m);
  hipStream_t s = (hipStream_t)stream;
  int* lens32 = nullptr;
  HIPCHK(m, hipMalloc((void**)&lens32, (size_t)B * sizeof(int) * 2));
  launch_lens_to_i32(lengths, lens32, B, T, lens32 + B, s);
  launch_rel_attention(qkv, emb_k, emb_v, lens32, o, B, H, n_heads, T, s);
  const hipError_t e = hipStreamSynchronize(s);
  (void)hipFree(lens32);
  HIPCHK(m, e);
  HIPCHK(m, hipGetLastError());
  return 0;
}

int mbv_op_conv1d(mbv_model* m, const float* x, const float* w_host, const float* bias_host, float* y,
                  int B, int Cin, int Cout, int T, int K, int dilation, float in_slope, void* stream) {
  if (!m) return 1;
  if (Cin % 32) return m->fail("mbv_op_conv1d: Cin must be a multiple of 32");
  if (!conv1d_supported(K, dilation)) return m->fail("mbv_op_conv1d: K <= 11 and (K-1)*dilation <= 72 required");
  DEVICE_GUARD(m);
  hipStream_t s = (hipStream_t)stream;
  const int Mpad = (int)align_up(Cout, 128);
  std::vector<float> packed((size_t)K * Cin * Mpad, 0.f);
  for (int k = 0; k < K; ++k)
    for (int ci = 0; ci < Cin; ++ci)
      for (int co = 0; co < Cout; ++co)
        packed[conv_pack_index(k, ci, co, Cin, Mpad)] = w_host[((size_t)co * Cin + ci) * K + k];
  float *dw = nullptr, *db = nullptr;
  HIPCHK(m, hipMalloc((void**)&dw, packed.size() * 4));
  HIPCHK(m, hipMemcpy(dw, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  if (bias_host) {
    HIPCHK(m, hipMalloc((void**)&db, (size_t)Cout * 4));
    HIPCHK(m, hipMemcpy(db, bias_host, (size_t)Cout * 4, hipMemcpyHostToDevice));
  }
  ConvArgs a{};
  a.x = x; a.x_bstride = (int64_t)Cin * T; a.Tin = T; a.x_rstride = T; a.Cin = Cin;
  a.w = dw; a.bias = db; a.M = Cout; a.Mpad = Mpad; a.K = K; a.dil = dilation;
  a.pad_left = (K - 1) * dilation / 2; a.in_slope = in_slope;
  a.y = y; a.y_bstride = (int64_t)Cout * T; a.T = T; a.epi = EPI_STORE; a.out_scale = 1.f; a.B = B;
  a.ws = m->conv_ws; a.ws_floats = m->conv_ws_floats; a.counters = m->conv_cnt; a.n_counters = m->conv_ncnt;
  a.splitk = m->splitk;
  float* dws = nullptr;
  if (m->conv_bf16 == 3) {                           // the split copy of this call's weights
    HIPCHK(m, hipMalloc((void**)&dws, packed.size() * 4));
    launch_split_planes(dw, dws, packed.size(), s);
    a.w_split = dws;
  }
  a.prec = a.w_split ? 3 : 0;
  launch_conv1d(a, s);
  HIPCHK(m, hipStreamSynchronize(s));
  HIPCHK(m, hipFree(dw));
  if (dws) HIPCHK(m, hipFree(dws));
  if (db) HIPCHK(m, hipFree(db));
  return 0;
}

}  // extern "C"
