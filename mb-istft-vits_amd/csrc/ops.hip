// Small memory-bound pieces of the path: embedding, channel LayerNorm,
// durations (proj + exp/ceil/cumsum), length regulation, speaker conditioning.
#include "kernels.h"

namespace mbv {

// ---------------------------------------------------------------------------
// x[b, c, t] = emb[ids[b,t]][c] * sqrt(H) * (t < len[b])       models.py:173-177
// Also narrows the int64 lengths to the int32 copy the other kernels use.
// ---------------------------------------------------------------------------
__global__ void embed_kernel(const int64_t* ids, const int64_t* lens, const float* emb, float* x,
                             int* lens32, int* bad, int B, int T, int H, int n_vocab, float scale) {
  const int b = blockIdx.z;
  const int c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0 && c == 0) {
    const long long l = lens[b];
    lens32[b] = l < 0 ? 0 : (l > T ? T : (int)l);
    if (l < 0 || l > T) bad[b] = 1;               // x_lengths outside [0, T]
  }
  if (t >= T) return;
  const int len = (int)lens[b];
  long long id = ids[(int64_t)b * T + t];
  if (id < 0 || id >= n_vocab) {                    // nn.Embedding would raise IndexError: report it
    if (c == 0) bad[b] = 1;                         // (mbv_encode returns y_lengths = -1 for the utterance)
    id = 0;
  }
  float v = emb[id * H + c] * scale;
  x[((int64_t)b * H + c) * T + t] = t < len ? v : 0.f;
}

void launch_embed(const int64_t* ids, const int64_t* lens, const float* emb, float* x, int* lens32,
                  int* bad, int B, int T, int H, int n_vocab, hipStream_t s) {
  (void)hipMemsetAsync(bad, 0, (size_t)B * sizeof(int), s);
  dim3 grid((T + 63) / 64, H, B);
  hipLaunchKernelGGL(embed_kernel, grid, dim3(64), 0, s, ids, lens, emb, x, lens32, bad, B, T, H, n_vocab,
                     sqrtf((float)H));
}

// ---------------------------------------------------------------------------
// Split-bf16 copy of the packed conv weights (opt-in conv_bf16 mode, conv1d.hip): every 16-byte slot of
// four fp32 K-values becomes [bf16 hi x 4 | bf16 mid x 4], hi = bf16(x), mid = bf16(x - hi) — the layout
// the conv kernel also gives its input window in that mode, so an MFMA operand is two slots' halves.
// ---------------------------------------------------------------------------
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void split_planes_kernel(const float4* src, float4* dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = src[i];
    const float x[4] = {v.x, v.y, v.z, v.w};
    bf16x4_t hi, mid;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const __bf16 h = (__bf16)x[k];
      hi[k] = h;
      mid[k] = (__bf16)(x[k] - (float)h);
    }
    struct { bf16x4_t h, m; } o{hi, mid};
    dst[i] = __builtin_bit_cast(float4, o);
  }
}
void launch_split_planes(const float* src, float* dst, size_t n_floats, hipStream_t s) {
  const size_t n4 = n_floats / 4;
  const int grid = (int)(n4 / 256 + 1 < 2048 ? n4 / 256 + 1 : 2048);
  hipLaunchKernelGGL(split_planes_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), n4);
}

// ---------------------------------------------------------------------------
// Channel LayerNorm (modules.py:29-32), eps 1e-5, two-pass statistics like
// F.layer_norm.  Fused: residual add (attentions.py:41,45), ReLU in front
// (models.py:129-130), mask behind (attentions.py:46).
// Block = 32 time steps x 8 channel groups; each thread keeps its <= 32
// channel values in registers, the 8 partials are reduced through LDS.
// ---------------------------------------------------------------------------
constexpr int LN_TX = 32, LN_CY = 8, LN_MAXPER = 32;   // C <= 256 (H and the dp filter width)

__global__ __launch_bounds__(256) void layernorm_kernel(const float* a, const float* r,
                                                        const float* gamma, const float* beta,
                                                        float* y, int C, int T, int pre_relu,
                                                        const int* out_lens) {
  __shared__ float red[LN_CY][LN_TX];
  const int tx = threadIdx.x & 31, cy = threadIdx.x >> 5;
  const int b = blockIdx.y;
  const int t = blockIdx.x * LN_TX + tx;
  const bool ok = t < T;
  const int64_t base = (int64_t)b * C * T + t;
  float v[LN_MAXPER];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXPER; ++i) {
    const int c = cy + i * LN_CY;
    float x = 0.f;
    if (ok && c < C) {
      x = a[base + (int64_t)c * T];
      if (r) x += r[base + (int64_t)c * T];
      if (pre_relu) x = fmaxf(x, 0.f);
    }
    v[i] = x;
    sum += x;
  }
  red[cy][tx] = sum;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int k = 0; k < LN_CY; ++k) mean += red[k][tx];
  mean /= (float)C;
  __syncthreads();
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXPER; ++i) {
    const int c = cy + i * LN_CY;
    if (c < C) { const float d = v[i] - mean; sq += d * d; }
  }
  red[cy][tx] = sq;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < LN_CY; ++k) var += red[k][tx];
  const float rstd = rsqrtf(var / (float)C + 1e-5f);
  const float m = (out_lens && t >= out_lens[b]) ? 0.f : 1.f;
  if (!ok) return;
#pragma unroll
  for (int i = 0; i < LN_MAXPER; ++i) {
    const int c = cy + i * LN_CY;
    if (c < C) y[base + (int64_t)c * T] = ((v[i] - mean) * rstd * gamma[c] + beta[c]) * m;
  }
}

void launch_layernorm(const float* a, const float* r, const float* gamma, const float* beta,
                      float* y, int B, int C, int T, int pre_relu, const int* out_lens,
                      hipStream_t s) {
  dim3 grid((T + LN_TX - 1) / LN_TX, B);
  hipLaunchKernelGGL(layernorm_kernel, grid, dim3(256), 0, s, a, r, gamma, beta, y, C, T, pre_relu,
                     out_lens);
}

// ---------------------------------------------------------------------------
// Durations: dp.proj (C -> 1, models.py:136-137) fused with models.py:717-719:
//   logw = (w . (h * mask) + b) * mask ; w = exp(logw) * mask * length_scale
//   w_ceil = ceil(w) ; cum = inclusive cumsum ; y_len = max(sum, 1)
// One block per utterance; T is scanned in chunks of 256.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void durations_kernel(const float* h, const float* w,
                                                        const float* bias, const int* lens,
                                                        float length_scale, float* logw,
                                                        float* w_ceil, int* cum, int* ylen32,
                                                        int64_t* ylen64, const int* bad, int C, int T) {
  __shared__ int scan[256];
  __shared__ int carry_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int len = lens[b];
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + tid;
    int d = 0;
    if (t < T) {
      float lw = 0.f, wc = 0.f;
      if (t < len) {
        if (w) {
          // four interleaved partial sums: one dependent fmaf chain over 256 channels behind 256 loads was
          // 63 us per launch, as long as a text-encoder conv
          float acc4[4] = {0.f, 0.f, 0.f, 0.f};
          const float* hp = h + (int64_t)b * C * T + t;
          int c = 0;
          for (; c + 8 <= C; c += 8) {
            float hv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) hv[q] = hp[(int64_t)(c + q) * T];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc4[q & 3] = fmaf(w[c + q], hv[q], acc4[q & 3]);
          }
          for (; c < C; ++c) acc4[c & 3] = fmaf(w[c], hp[(int64_t)c * T], acc4[c & 3]);
          lw = ((acc4[0] + acc4[1]) + (acc4[2] + acc4[3])) + bias[0];
        } else {
          lw = h[(int64_t)b * T + t];            // SDP: logw computed by the flows (sdp.hip)
        }
        wc = ceilf(expf(lw) * length_scale);
      }
      logw[(int64_t)b * T + t] = lw;
      w_ceil[(int64_t)b * T + t] = wc;
      d = (int)wc;
    }
    scan[tid] = d;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {          // Hillis-Steele inclusive scan
      int v = tid >= off ? scan[tid - off] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    const int carry = carry_s;
    if (t < T) cum[(int64_t)b * T + t] = carry + scan[tid];
    __syncthreads();
    if (tid == 255) carry_s = carry + scan[255];
    __syncthreads();
  }
  if (tid == 0) {
    const int total = carry_s < 1 ? 1 : carry_s;
    ylen32[b] = total;
    if (ylen64) ylen64[b] = (bad && bad[b]) ? -1 : total;     // -1: invalid token id / length / sid
  }
}

void launch_durations(const float* h, const float* w, const float* b, const int* lens,
                      float length_scale, float* logw, float* w_ceil, int* cum, int* ylen32,
                      int64_t* ylen64, const int* bad, int B, int C, int T, hipStream_t s) {
  hipLaunchKernelGGL(durations_kernel, dim3(B), dim3(256), 0, s, h, w, b, lens, length_scale, logw,
                     w_ceil, cum, ylen32, ylen64, bad, C, T);
}

// ---------------------------------------------------------------------------
// Length regulation (models.py:720-729, commons.py:128-143).  The reference
// builds a one-hot path [B,1,T',T] and matmuls; it is a gather: frame t' takes
// token j = first index with cum[j] > t'.  Writes m_p, logs_p, z_p (= m_p +
// noise * exp(logs_p) * noise_scale), z (copy of z_p: the flows run in place),
// and optionally the dense attn path and y_mask.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_kernel(const float* m_t, const float* logs_t,
                                                     int64_t src_bstride, const int* cum,
                                                     const int* ylen,
                                                     const float* noise, float noise_scale,
                                                     float* m_p, float* logs_p, float* z_p, float* z,
                                                     float* y_mask, int C, int T, int Tp) {
  const int b = blockIdx.y;
  const int tp = blockIdx.x * blockDim.x + threadIdx.x;
  if (tp >= Tp) return;
  const int yl = ylen[b];
  const int* cb = cum + (int64_t)b * T;
  int j = -1;
  if (tp < yl) {
    int lo = 0, hi = T - 1;                  // first j with cum[j] > tp
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cb[mid] > tp) hi = mid; else lo = mid + 1;
    }
    j = cb[lo] > tp ? lo : -1;               // -1: all-zero durations (y_len clamped to 1)
  }
  if (y_mask && blockIdx.z == 0) y_mask[(int64_t)b * Tp + tp] = tp < yl ? 1.f : 0.f;
  const int c_lo = blockIdx.z * 16, c_hi = c_lo + 16 < C ? c_lo + 16 : C;   // 16 channels per block
  for (int c = c_lo; c < c_hi; ++c) {
    const int64_t o = ((int64_t)b * C + c) * Tp + tp;
    float m = 0.f, lg = 0.f;
    if (j >= 0) {
      m = m_t[(int64_t)b * src_bstride + (int64_t)c * T + j];
      lg = logs_t[(int64_t)b * src_bstride + (int64_t)c * T + j];
    }
    float zp = m;
    if (noise) zp = m + noise[o] * expf(lg) * noise_scale;
    if (m_p) m_p[o] = m;
    if (logs_p) logs_p[o] = lg;
    if (z_p) z_p[o] = zp;
    z[o] = tp < yl ? zp : 0.f;       // z enters the flows masked (the reference masks it in the first coupling that updates a half;
                                     // the fused WN layers only touch valid frames, wn_fused.hip); z_p keeps the reference's unmasked draw
  }
}

__global__ void attn_path_kernel(const int* cum, const int* ylen, float* attn, int T, int Tp) {
  // attn[b, 0, tp, t] = 1 iff cum[t-1] <= tp < cum[t] and tp < y_len
  const int b = blockIdx.z, tp = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int* cb = cum + (int64_t)b * T;
  const int hi = cb[t], lo = t > 0 ? cb[t - 1] : 0;
  attn[((int64_t)b * Tp + tp) * T + t] = (tp < ylen[b] && tp >= lo && tp < hi) ? 1.f : 0.f;
}

void launch_expand(const float* m_t, const float* logs_t, int64_t src_bstride, const int* cum,
                   const int* ylen, const float* noise, float noise_scale, float* m_p,
                   float* logs_p, float* z_p, float* z, float* attn, float* y_mask, int B, int C,
                   int T, int Tp, hipStream_t s) {
  dim3 grid((Tp + 255) / 256, B, (C + 15) / 16);
  hipLaunchKernelGGL(expand_kernel, grid, dim3(256), 0, s, m_t, logs_t, src_bstride, cum, ylen, noise,
                     noise_scale, m_p, logs_p, z_p, z, y_mask, C, T, Tp);
  if (attn) {
    dim3 g2((T + 255) / 256, Tp, B);
    hipLaunchKernelGGL(attn_path_kernel, g2, dim3(256), 0, s, cum, ylen, attn, T, Tp);
  }
}

// ---------------------------------------------------------------------------
// Speaker conditioning: 1x1 convs on g [B, gin, 1] are GEMVs
// (models.py:127 dp.cond, modules.py:152 WN cond_layer, modules.py:215 ResBlock cond).
// ---------------------------------------------------------------------------
__global__ void cond_gemv_kernel(const float* g, const float* table, const int64_t* sid,
                                 const float* W, const float* bias, float* out, int Cin, int Cout) {
  const int b = blockIdx.y;
  const int co = blockIdx.x * blockDim.x + threadIdx.x;
  if (co >= Cout) return;
  const float* gv = table ? table + sid[b] * Cin : g + (int64_t)b * Cin;
  const float* wr = W + (int64_t)co * Cin;
  float acc = 0.f;
  for (int ci = 0; ci < Cin; ++ci) acc = fmaf(wr[ci], gv[ci], acc);
  out[(int64_t)b * Cout + co] = acc + (bias ? bias[co] : 0.f);
}

void launch_cond_gemv(const float* g, const float* table, const int64_t* sid, const float* W,
                      const float* bias, float* out, int B, int Cin, int Cout, hipStream_t s) {
  dim3 grid((Cout + 127) / 128, B);
  hipLaunchKernelGGL(cond_gemv_kernel, grid, dim3(128), 0, s, g, table, sid, W, bias, out, Cin, Cout);
}

__global__ void gather_rows_kernel(const float* table, const int64_t* sid, float* out, int C,
                                   int n_rows, int* bad) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  long long r = sid[b];
  if (r < 0 || r >= n_rows) {
    if (bad && c == 0) bad[b] = 1;
    r = 0;
  }
  out[(int64_t)b * C + c] = table[r * C + c];
}

void launch_gather_rows(const float* table, const int64_t* sid, float* out, int B, int C,
                        int n_rows, int* bad, hipStream_t s) {
  dim3 grid((C + 127) / 128, B);
  hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(128), 0, s, table, sid, out, C, n_rows, bad);
}

__global__ void unscale_xpost_kernel(const float* src, float* dst, int rows, int F, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int ch = (int)((i / F) % rows);
    dst[i] = src[i] * ((ch % 18) < 9 ? 0.69314718055994531f : 6.28318530717958648f);
  }
}

void launch_unscale_xpost(const float* src, float* dst, int B, int rows, int F, hipStream_t s) {
  const int64_t n = (int64_t)B * rows * F;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(unscale_xpost_kernel, dim3(blocks), dim3(256), 0, s, src, dst, rows, F, n);
}

// ---------------------------------------------------------------------------
// Wire-format epilogue of the service wrapper (tts_vits.py:204-217): per utterance
//   peak = max |x| over the valid samples; if auto_normalize and peak > 0.01: x = x / peak * 0.9
//   x = clip(x, -1, 1); pcm = int16(x * 32767)   (truncation toward zero, as ndarray.astype)
// Same fp32 operation order as the NumPy code, so the int16 stream is bit-exact.
// ---------------------------------------------------------------------------
// valid samples of row b, clamped to the row: after infer(max_len=k) the waveform rows hold only
// spf * k samples while y_lengths still counts the untruncated frames (models.py:733-734)
__device__ __forceinline__ int64_t valid_samples(const int64_t* lens, int b, int spf, int64_t stride) {
  if (!lens) return stride;
  const int64_t n = lens[b] * (int64_t)spf;
  return n < 0 ? 0 : (n > stride ? stride : n);
}

__global__ void absmax_kernel(const float* x, const int64_t* lens, int64_t stride, int spf,
                              unsigned* peak_bits) {
  const int b = blockIdx.y;
  const int64_t n = valid_samples(lens, b, spf, stride);
  const float* xb = x + (int64_t)b * stride;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(xb[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) atomicMax(&peak_bits[b], __float_as_uint(m));   // m >= 0: bit order == value order
}

__global__ void pcm16_kernel(const float* x, const int64_t* lens, int64_t stride, int spf,
                             const unsigned* peak_bits, int auto_normalize, short* out) {
  const int b = blockIdx.y;
  const int64_t n = valid_samples(lens, b, spf, stride);
  const float peak = __uint_as_float(peak_bits[b]);
  const bool norm = auto_normalize && peak > 0.01f;
  const float* xb = x + (int64_t)b * stride;
  short* ob = out + (int64_t)b * stride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < stride; i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < n) {
      v = xb[i];
      if (norm) v = (v / peak) * 0.9f;
      v = fminf(fmaxf(v, -1.f), 1.f);
      v = v * 32767.f;
    }
    ob[i] = (short)(int)v;
  }
}

void launch_pcm16(const float* x, const int64_t* lens, int B, int64_t stride, int spf, int auto_normalize,
                  unsigned* peak_scratch, short* out, hipStream_t s) {
  (void)hipMemsetAsync(peak_scratch, 0, (size_t)B * sizeof(unsigned), s);
  int bx = (int)((stride + 255) / 256);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(absmax_kernel, dim3(bx, B), dim3(256), 0, s, x, lens, stride, spf, peak_scratch);
  hipLaunchKernelGGL(pcm16_kernel, dim3(bx, B), dim3(256), 0, s, x, lens, stride, spf, peak_scratch,
                     auto_normalize, out);
}

__global__ void posterior_sample_kernel(const float* stats, const float* noise, const int* lens,
                                        float* z, int I, int T) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float m = stats[((int64_t)b * 2 * I + c) * T + t];
  const float lg = stats[((int64_t)b * 2 * I + I + c) * T + t];
  const int64_t o = ((int64_t)b * I + c) * T + t;
  const float v = noise ? m + noise[o] * expf(lg) : m;
  z[o] = t < lens[b] ? v : 0.f;
}

void launch_posterior_sample(const float* stats, const float* noise, const int* lens, float* z, int B,
                             int I, int T, hipStream_t s) {
  dim3 grid((T + 127) / 128, I, B);
  hipLaunchKernelGGL(posterior_sample_kernel, grid, dim3(128), 0, s, stats, noise, lens, z, I, T);
}

__global__ void sequence_mask_kernel(const int* lens, float* mask, int T) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < T) mask[(int64_t)b * T + t] = t < lens[b] ? 1.f : 0.f;
}

void launch_sequence_mask(const int* lens, float* mask, int B, int T, hipStream_t s) {
  hipLaunchKernelGGL(sequence_mask_kernel, dim3((T + 255) / 256, B), dim3(256), 0, s, lens, mask, T);
}

__global__ void lens_to_i32_kernel(const int64_t* lens, int* out, int B, int T, int* bad) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const long long l = lens[b];
  out[b] = l < 0 ? 0 : (l > T ? T : (int)l);
  bad[b] = (l < 0 || l > T) ? 1 : 0;
}

void launch_lens_to_i32(const int64_t* lens, int* out, int B, int T, int* bad, hipStream_t s) {
  hipLaunchKernelGGL(lens_to_i32_kernel, dim3((B + 63) / 64), dim3(64), 0, s, lens, out, B, T, bad);
}

__global__ void fill_kernel(float* p, float v, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// trimmed decode (ConvArgs::trim_map): per-utterance column-tile counts, their prefix sums and the tile -> utterance map
__global__ void trim_map_kernel(const int* __restrict__ lens, int B, int num, int add, int T, int BN, int* __restrict__ out) {
  __shared__ int part[256];
  const int tid = threadIdx.x;
  const int per = (B + 255) / 256;
  auto tiles_of = [&](int b) {
    long lim = (long)lens[b] * num + add;
    lim = lim < 0 ? 0 : (lim > T ? T : lim);
    return (int)((lim + BN - 1) / BN);
  };
  int s = 0;
  for (int i = 0; i < per; ++i) { const int b = tid * per + i; if (b < B) s += tiles_of(b); }
  part[tid] = s;
  __syncthreads();
  if (tid == 0) { int run = 0; for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = run; run += v; } }
  __syncthreads();
  int run = part[tid];
  for (int i = 0; i < per; ++i) {
    const int b = tid * per + i;
    if (b < B) {
      out[b] = run;
      const int n = tiles_of(b);
      for (int k = 0; k < n; ++k) out[B + 1 + run + k] = b;
      run += n;
      if (b == B - 1) out[B] = run;
    }
  }
}
size_t launch_trim_map_ints(int B, int T, int BN) { return (size_t)B + 1 + (size_t)B * ((T + BN - 1) / BN); }
void launch_trim_map(const int* lens, int B, int num, int add, int T, int BN, int* out, hipStream_t s) {
  hipLaunchKernelGGL(trim_map_kernel, dim3(1), dim3(256), 0, s, lens, B, num, add, T, BN, out);
}

void launch_fill(float* p, float v, int64_t n, hipStream_t s) {
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, p, v, n);
}

}  // namespace mbv
