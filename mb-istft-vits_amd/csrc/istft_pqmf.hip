// Fused waveform tail: subband_conv_post output -> 22.05 kHz waveform in ONE
// launch (HBM-bound: 4608 B read + 1024 B written per z-frame).
//
//   spec  = exp(x[:, :, :9]) ; phase = pi * sin(x[:, :, 9:])          models.py:368-369
//   y_mb  = istft(spec * e^{j phase}, n_fft 16, hop 4, hann, center)   stft.py:197-202
//   o     = conv1d(pad31(zero_stuff4(y_mb) * 4), h_syn)                pqmf.py:115-116
//           (MS: h_syn = weight-normed multistream_conv_post)         models.py:463-465
//
// One workgroup owns TM sub-band samples per band (4*TM output samples) of
// one utterance and runs three phases separated by two barriers:
//   A  one lane per (band, frame): 18 coalesced loads down the frame axis,
//      exp / pi*sin / sincos, 16-point real inverse DFT (even/odd-bin split),
//      hann window -> LDS  fr[band][n][frame]
//   B  one lane per (band, 4 consecutive samples): overlap-add of the 4 frames
//      that cover them, divide by the edge-aware sum of squared windows
//      (what torch.istft does), zero outside the signal -> LDS ys[band][.]
//   C  one lane per sub-band sample m: the 4 polyphase outputs o[4m..4m+3]
//      (<= 16 taps x 4 bands each; the zero-stuffed x4 upsampling never
//      materialises), one 16-byte store per lane.
// Frames / samples in the 8/7-sample halos are recomputed, not exchanged;
// consecutive tiles are mapped to the same XCD so halo rows hit in its L2.
#include "kernels.h"

namespace mbv {

namespace {

constexpr float kPi = 3.14159265358979323846f;

// cos(2*pi*j/16), sin(2*pi*j/16)
__device__ constexpr float COS16[16] = {
    1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
    0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
    -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f,
    0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
__device__ constexpr float SIN16[16] = {
    0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
    1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
    0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
    -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};
// periodic hann(16): 0.5 - 0.5 cos(2 pi n / 16)
__device__ constexpr float HANN16[16] = {
    0.0f, 0.03806023374435663f, 0.14644660940672624f, 0.30865828381745514f,
    0.5f, 0.69134171618254486f, 0.85355339059327376f, 0.96193976625564337f,
    1.0f, 0.96193976625564337f, 0.85355339059327376f, 0.69134171618254486f,
    0.5f, 0.30865828381745514f, 0.14644660940672624f, 0.03806023374435663f};

}  // namespace

// filt: [band][p][16] with filt[band][p][i] = 4 * h[band][3 - p + 4 i]  (0 where the tap is > 62)
template <int TM, int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void istft_pqmf_kernel(const IstftArgs a, int tiles_per_utt,
                                                              int total_tiles) {
  constexpr int NF = TM / 4 + 7;          // frames a tile touches per band
  constexpr int NFS = TM / 4 + 8;         // padded frame stride in LDS
  constexpr int YL = TM + 16;             // sub-band samples incl. PQMF halo
  static_assert(4 * NF <= NTHREADS, "one lane per (band, frame)");
  static_assert(TM <= NTHREADS, "one lane per sub-band sample");
  __shared__ __attribute__((aligned(16))) float fr[4 * 16 * NFS];
  __shared__ __attribute__((aligned(16))) float ys[4 * YL];

  // XCD-aware tile order: workgroups with equal (id % 8) share an L2; give each
  // of the 8 groups a contiguous run of tiles so halo rows are re-read on-die.
  int tile;
  {
    const int bid = blockIdx.x, q = total_tiles / 8, r = total_tiles % 8, x = bid % 8;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
  }
  const int b = tile / tiles_per_utt;
  const int m0 = (tile % tiles_per_utt) * TM;
  const int Tp = a.Tp;
  const int F = 16 * Tp + 1;
  const int M = 64 * Tp;                  // sub-band samples per band
  const int tid = threadIdx.x;
  const int f_lo = m0 / 4 - 3;

  // ---------------- phase A: frames --------------------------------------
  if (tid < 4 * NF) {
    const int band = tid / NF, fl = tid % NF;
    const int f = f_lo + fl;
    float out[16];
    if (f >= 0 && f < F) {
      const float* xp = a.x_post + ((int64_t)b * 72 + band * 18) * F + f;
      float xin[18];
#pragma unroll
      for (int k = 0; k < 18; ++k) xin[k] = xp[(int64_t)k * F];
      float re[9], im[9];
      // frame f is owned (for the spec/phase outputs) by the tile holding sample 4f
      const bool own = (4 * f >= m0 && 4 * f < m0 + TM) || (f == F - 1 && m0 + TM >= M);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float mag = expf(xin[k]);
        const float ph = kPi * sinf(xin[9 + k]);
        float sn, cs;
        sincosf(ph, &sn, &cs);
        re[k] = mag * cs;
        im[k] = mag * sn;
        if (own) {
          if (a.spec) a.spec[(((int64_t)b * 4 + band) * 9 + k) * F + f] = mag;
          if (a.phase) a.phase[(((int64_t)b * 4 + band) * 9 + k) * F + f] = ph;
        }
      }
      // x[n] = E[n] + O[n], x[n+8] = E[n] - O[n]  (even / odd bins)
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        float e = re[0] + ((n & 1) ? -re[8] : re[8]);
        float o = 0.f;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
          const float term = 2.f * (re[k] * COS16[(k * n) & 15] - im[k] * SIN16[(k * n) & 15]);
          if (k & 1) o += term; else e += term;
        }
        out[n] = (e + o) * (1.f / 16.f) * HANN16[n];
        out[n + 8] = (e - o) * (1.f / 16.f) * HANN16[n + 8];
      }
    } else {
#pragma unroll
      for (int n = 0; n < 16; ++n) out[n] = 0.f;
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) fr[(band * 16 + n) * NFS + fl] = out[n];
  }
  __syncthreads();

  // ---------------- phase B: overlap-add + envelope -----------------------
  if (tid < YL) {                         // 4 bands x YL/4 quads == YL work items
    constexpr int QB = YL / 4;            // quads per band
    const int band = tid / QB, q = tid % QB;
    // quad q covers m = m0 - 8 + 4q + r ; f' = m0/4 - 2 + q ; frames f'-1 .. f'+2
    const int fbase = q;                  // local index of frame f'-1  (f_lo = m0/4 - 3)
    const int fp = m0 / 4 - 2 + q;
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = 0.f, env = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {       // frame f'-1+g contributes its sample n = 12 - 4g + r
        const int n = 12 - 4 * g + r;
        const int f = fp - 1 + g;
        // q <= YL/4 - 1 = TM/4 + 3 and g <= 3 -> local frame index <= NF - 1
        s += fr[(band * 16 + n) * NFS + fbase + g];
        env += (f >= 0 && f < F) ? HANN16[n] * HANN16[n] : 0.f;
      }
      const int m = m0 - 8 + 4 * q + r;
      y[r] = (m >= 0 && m < M) ? s / env : 0.f;
    }
    *reinterpret_cast<float4*>(&ys[band * YL + 4 * q]) = make_float4(y[0], y[1], y[2], y[3]);
    if (a.o_mb && q >= 2 && q < QB - 2) {          // owned samples m0 .. m0+TM-1
      const int m = m0 - 8 + 4 * q;
      if (m < M) {
        if (!a.multistream) {
          *reinterpret_cast<float4*>(a.o_mb + ((int64_t)b * 4 + band) * M + m) =
              make_float4(y[0], y[1], y[2], y[3]);
        } else {                                    // zero-stuffed x4, gain 4 (models.py:463)
          float* dst = a.o_mb + ((int64_t)b * 4 + band) * 4 * M + 4 * (int64_t)m;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            reinterpret_cast<float4*>(dst)[r] = make_float4(4.f * y[r], 0.f, 0.f, 0.f);
        }
      }
    }
  }
  __syncthreads();

  // ---------------- phase C: polyphase synthesis filter --------------------
  if (tid < TM) {
    const int m = m0 + tid;
    if (m < M) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int band = 0; band < 4; ++band) {
        const float* yb = &ys[band * YL + tid + 1];        // y[m - 7 + i]
        const float* hb = a.filt + band * 64;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float yv = yb[i];
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[p] = fmaf(hb[p * 16 + i], yv, acc[p]);
        }
      }
      *reinterpret_cast<float4*>(a.o + (int64_t)b * 4 * M + 4 * (int64_t)m) =
          make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

void launch_istft_pqmf(const IstftArgs& a, hipStream_t s) {
  constexpr int TM = 480, NT = 512;
  const int M = 64 * a.Tp;
  const int tiles_per_utt = (M + TM - 1) / TM;
  const int total = tiles_per_utt * a.B;
  hipLaunchKernelGGL((istft_pqmf_kernel<TM, NT>), dim3(total), dim3(NT), 0, s, a, tiles_per_utt,
                     total);
}

}  // namespace mbv
