// StochasticDurationPredictor, reverse direction (models.py:53-60, 89-100): the small
// memory-bound pieces around the 1x1 convolutions (which run on conv1d_mfma):
//   * DDSConv halves (modules.py:98-111): depth-wise dilated conv -> LayerNorm -> GELU, and
//     LayerNorm -> GELU -> residual behind the 1x1 conv
//   * ConvFlow.pre (a 1 -> C outer product) fused with the `x + g` of DDSConv.forward
//   * the inverse rational-quadratic spline with linear tails (transforms.py:55-170), fused
//     with the Flip in front of it and the mask behind it
//   * the last Flip + ElementwiseAffine.reverse (modules.py:282, 304) -> logw
// Everything here is per token ([B, C <= 256, T_text]); a batch of 64 x 200 tokens is 12 800 columns.
#include "kernels.h"

namespace mbv {

namespace {
constexpr int TX = 32, CY = 8, MAXPER = 32;    // block = 32 time steps x 8 channel groups, C <= 256

__device__ __forceinline__ float gelu_erf(float x) {          // F.gelu default (exact)
  return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
}

// statistics over the channel axis for the TX columns of a block (two passes, like F.layer_norm)
__device__ __forceinline__ void channel_stats(const float (&v)[MAXPER], int C, int cy, int tx,
                                              float (&red)[CY][TX], float& mean, float& rstd) {
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPER; ++i) sum += v[i];            // entries past C are 0
  red[cy][tx] = sum;
  __syncthreads();
  mean = 0.f;
#pragma unroll
  for (int k = 0; k < CY; ++k) mean += red[k][tx];
  mean /= (float)C;
  __syncthreads();
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPER; ++i) {
    const int c = cy + i * CY;
    if (c < C) { const float d = v[i] - mean; sq += d * d; }
  }
  red[cy][tx] = sq;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < CY; ++k) var += red[k][tx];
  rstd = rsqrtf(var / (float)C + 1e-5f);
}
}  // namespace

// y = gelu(LN(dwconv_k,dil(x * mask) + bias))                       modules.py:102-104
__global__ __launch_bounds__(256) void dds_sep_kernel(const float* x, const int* lens, const float* w,
                                                      const float* bias, const float* gamma,
                                                      const float* beta, float* y, int C, int T, int K,
                                                      int dil) {
  __shared__ float red[CY][TX];
  const int tx = threadIdx.x & 31, cy = threadIdx.x >> 5;
  const int b = blockIdx.y, t = blockIdx.x * TX + tx;
  const int len = lens[b];
  const int64_t base = (int64_t)b * C * T;
  const int half = (K * dil - dil) / 2;
  float v[MAXPER];
#pragma unroll
  for (int i = 0; i < MAXPER; ++i) {
    const int c = cy + i * CY;
    float acc = 0.f;
    if (t < T && c < C) {
      acc = bias[c];
      for (int k = 0; k < K; ++k) {
        const int tt = t - half + k * dil;
        if (tt >= 0 && tt < len) acc = fmaf(w[c * K + k], x[base + (int64_t)c * T + tt], acc);   // len <= T
      }
    }
    v[i] = acc;
  }
  float mean, rstd;
  channel_stats(v, C, cy, tx, red, mean, rstd);
  if (t >= T) return;
#pragma unroll
  for (int i = 0; i < MAXPER; ++i) {
    const int c = cy + i * CY;
    if (c < C) y[base + (int64_t)c * T + t] = gelu_erf((v[i] - mean) * rstd * gamma[c] + beta[c]);
  }
}

// y = (xres + gelu(LN(a))) [* mask]                                  modules.py:106-110
__global__ __launch_bounds__(256) void dds_res_kernel(const float* a, const float* xres,
                                                      const float* gamma, const float* beta, float* y,
                                                      int C, int T, const int* out_lens) {
  __shared__ float red[CY][TX];
  const int tx = threadIdx.x & 31, cy = threadIdx.x >> 5;
  const int b = blockIdx.y, t = blockIdx.x * TX + tx;
  const int64_t base = (int64_t)b * C * T + t;
  float v[MAXPER];
#pragma unroll
  for (int i = 0; i < MAXPER; ++i) {
    const int c = cy + i * CY;
    v[i] = (t < T && c < C) ? a[base + (int64_t)c * T] : 0.f;
  }
  float mean, rstd;
  channel_stats(v, C, cy, tx, red, mean, rstd);
  if (t >= T) return;
  const float m = (out_lens && t >= out_lens[b]) ? 0.f : 1.f;
#pragma unroll
  for (int i = 0; i < MAXPER; ++i) {
    const int c = cy + i * CY;
    if (c < C) {
      const int64_t o = base + (int64_t)c * T;
      y[o] = (xres[o] + gelu_erf((v[i] - mean) * rstd * gamma[c] + beta[c])) * m;
    }
  }
}

// h[b, c, t] = pre_w[c] * z[b, zc, t] + pre_b[c] + cond[b, c, t]       modules.py:379-380, 99-100
__global__ void sdp_pre_kernel(const float* z, int zc, const float* pre_w, const float* pre_b,
                               const float* cond, float* h, int C, int T) {
  const int b = blockIdx.z, c = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int64_t o = ((int64_t)b * C + c) * T + t;
  h[o] = fmaf(pre_w[c], z[((int64_t)b * 2 + zc) * T + t], pre_b[c]) + cond[o];
}

// Flip + ConvFlow tail (modules.py:282, 386-400, reverse): with (a, b) = z[:, 0], z[:, 1] on entry,
//   z[:, 0] <- b * mask ; z[:, 1] <- spline^-1(a; h) * mask
// h [B, 29, T]: 10 widths, 10 heights (both / sqrt(C)), 9 inner derivatives; the two outer
// derivatives are the constant of transforms.py:74.  fp32, operation order of transforms.py.
__global__ __launch_bounds__(128) void sdp_spline_kernel(const float* h, float* z, const int* lens,
                                                         int T, float inv_sqrt_c, float edge_const) {
  constexpr int NB = 10;
  constexpr float TAIL = 5.f, MINV = 1e-3f;
  const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  float* z0 = z + ((int64_t)b * 2) * T + t;
  float* z1 = z0 + T;
  const float mask = t < lens[b] ? 1.f : 0.f;
  const float x0 = *z1;                     // after the Flip
  const float yv = *z0;
  float out = yv;
  if (yv >= -TAIL && yv <= TAIL) {
    const float* hp = h + (int64_t)b * (3 * NB - 1) * T + t;
    float cw[NB + 1], ch[NB + 1], d[NB + 1];
    // knots: softmax -> min + (1 - min nb) p -> cumsum -> affine to [-5, 5], ends pinned
    auto knots = [&](int row0, float (&c)[NB + 1]) {
      float u[NB], mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < NB; ++k) { u[k] = hp[(int64_t)(row0 + k) * T] * inv_sqrt_c; mx = fmaxf(mx, u[k]); }
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NB; ++k) { u[k] = expf(u[k] - mx); sum += u[k]; }
      float run = 0.f;
      c[0] = -TAIL;
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        run += MINV + (1.f - MINV * NB) * (u[k] / sum);
        c[k + 1] = 2.f * TAIL * run - TAIL;
      }
      c[NB] = TAIL;
    };
    knots(0, cw);
    knots(NB, ch);
    auto softplus = [](float v) { return v > 20.f ? v : log1pf(expf(v)); };   // F.softplus, threshold 20
    d[0] = MINV + softplus(edge_const);
    d[NB] = d[0];
#pragma unroll
    for (int k = 1; k < NB; ++k) d[k] = MINV + softplus(hp[(int64_t)(2 * NB + k - 1) * T]);
    // bin: number of knots <= y, minus one (the last knot is nudged up by 1e-6, transforms.py:47)
    int idx = -1;
#pragma unroll
    for (int k = 0; k < NB; ++k) idx += yv >= ch[k] ? 1 : 0;
    idx += yv >= ch[NB] + 1e-6f ? 1 : 0;
    idx = idx < 0 ? 0 : (idx > NB - 1 ? NB - 1 : idx);
    float in_cw = 0.f, in_w = 0.f, in_ch = 0.f, in_h = 0.f, d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (k == idx) {
        in_cw = cw[k]; in_w = cw[k + 1] - cw[k];
        in_ch = ch[k]; in_h = ch[k + 1] - ch[k];
        d0 = d[k]; d1 = d[k + 1];
      }
    const float delta = in_h / in_w;
    const float tt = yv - in_ch;
    const float s2 = d0 + d1 - 2.f * delta;
    const float qa = tt * s2 + in_h * (delta - d0);
    const float qb = in_h * d0 - tt * s2;
    const float qc = -delta * tt;
    const float root = (2.f * qc) / (-qb - sqrtf(qb * qb - 4.f * qa * qc));
    out = root * in_w + in_cw;
  }
  *z0 = x0 * mask;
  *z1 = out * mask;
}

// last Flip + ElementwiseAffine.reverse: logw = (z[:, 1] - m[0]) * exp(-logs[0]) * mask
__global__ void sdp_logw_kernel(const float* z, const float* m, const float* logs, const int* lens,
                                float* logw, int T) {
  const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float mask = t < lens[b] ? 1.f : 0.f;
  logw[(int64_t)b * T + t] = (z[((int64_t)b * 2 + 1) * T + t] - m[0]) * expf(-logs[0]) * mask;
}

// z = noise * noise_scale_w (models.py:94), or zeros
__global__ void sdp_noise_kernel(const float* noise, float scale, float* z, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) z[i] = noise ? noise[i] * scale : 0.f;
}

__global__ void chan_add_kernel(float* x, const float* v, int C, int T) {
  const int b = blockIdx.z, c = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < T) x[((int64_t)b * C + c) * T + t] += v[b * C + c];
}

void launch_chan_add(float* x, const float* v, int B, int C, int T, hipStream_t s) {
  hipLaunchKernelGGL(chan_add_kernel, dim3((T + 63) / 64, C, B), dim3(64), 0, s, x, v, C, T);
}

void launch_dds_sep(const float* x, const int* lens, const float* w, const float* bias,
                    const float* gamma, const float* beta, float* y, int B, int C, int T, int K,
                    int dil, hipStream_t s) {
  hipLaunchKernelGGL(dds_sep_kernel, dim3((T + TX - 1) / TX, B), dim3(256), 0, s, x, lens, w, bias,
                     gamma, beta, y, C, T, K, dil);
}

void launch_dds_res(const float* a, const float* xres, const float* gamma, const float* beta,
                    float* y, int B, int C, int T, const int* out_lens, hipStream_t s) {
  hipLaunchKernelGGL(dds_res_kernel, dim3((T + TX - 1) / TX, B), dim3(256), 0, s, a, xres, gamma, beta,
                     y, C, T, out_lens);
}

void launch_sdp_pre(const float* z, int zc, const float* pre_w, const float* pre_b, const float* cond,
                    float* h, int B, int C, int T, hipStream_t s) {
  hipLaunchKernelGGL(sdp_pre_kernel, dim3((T + 63) / 64, C, B), dim3(64), 0, s, z, zc, pre_w, pre_b,
                     cond, h, C, T);
}

void launch_sdp_spline(const float* h, float* z, const int* lens, int B, int C, int T,
                       float edge_const, hipStream_t s) {
  hipLaunchKernelGGL(sdp_spline_kernel, dim3((T + 127) / 128, B), dim3(128), 0, s, h, z, lens, T,
                     1.f / sqrtf((float)C), edge_const);
}

void launch_sdp_logw(const float* z, const float* m, const float* logs, const int* lens, float* logw,
                     int B, int T, hipStream_t s) {
  hipLaunchKernelGGL(sdp_logw_kernel, dim3((T + 127) / 128, B), dim3(128), 0, s, z, m, logs, lens,
                     logw, T);
}

void launch_sdp_noise(const float* noise, float scale, float* z, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(sdp_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, noise,
                     scale, z, n);
}

}  // namespace mbv
