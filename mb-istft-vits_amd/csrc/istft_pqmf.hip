// Fused waveform tail: subband_conv_post output -> 22.05 kHz waveform in ONE
// launch (HBM-bound: 4608 B read + 1024 B written per z-frame).
//
//   spec  = exp(x[:, :, :9]) ; phase = pi * sin(x[:, :, 9:])          models.py:368-369
//   y_mb  = istft(spec * e^{j phase}, n_fft 16, hop 4, hann, center)   stft.py:197-202
//   o     = conv1d(pad31(zero_stuff4(y_mb) * 4), h_syn)                pqmf.py:115-116
//           (MS: h_syn = weight-normed multistream_conv_post)         models.py:463-465
//
// One workgroup owns TM sub-band samples per band (4*TM output samples) of
// one utterance and runs three phases separated by barriers:
//   A  one lane per (band, frame): 18 coalesced loads down the frame axis,
//      exp / sin / sincos, 16-point real inverse DFT (even/odd-bin split),
//      hann window -> LDS  fr[band][n][frame]
//   B  one lane per sub-band time index (all 4 bands): overlap-add of the 4
//      frames that cover it, divide by the edge-aware sum of squared windows
//      (what torch.istft does), zero outside the signal.
//      Fixed PQMF bank: the four band samples are immediately rotated into the
//      8 cosine-modulation phases U_q = sum_k cos(theta_k(q)) y_k  (the bank is
//      h_k[j] = 2 p[j] cos(theta_k(j)) with theta_k(j+8) = theta_k(j) + (2k+1)pi,
//      so cos(theta_k(j)) = (-1)^(j/8) cos(theta_k(j mod 8))) -> LDS Us[q][.]
//      Trainable bank (MS): the band samples go to LDS ys[band][.] as they are.
//   C  one lane per sub-band sample m: the 4 polyphase outputs o[4m..4m+3];
//      fixed bank: 16 prototype taps each (64 FMA per lane instead of 252),
//      trainable bank: <= 16 taps x 4 bands each.  The zero-stuffed x4
//      upsampling never materialises; one 16-byte store per lane.
// Frames / samples in the halos are recomputed, not exchanged; consecutive
// tiles are mapped to the same XCD so halo rows hit in its L2.
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
// squared window, for per-lane (runtime-indexed) envelope sums at the signal edges
__device__ const float WSQ16[16] = {
    0.0f, 0.0014485813926750633f, 0.021446609406726238f, 0.095269936190567076f,
    0.25f, 0.47795336526437190f, 0.72855339059327373f, 0.92533011387037270f,
    1.0f, 0.92533011387037270f, 0.72855339059327373f, 0.47795336526437190f,
    0.25f, 0.095269936190567076f, 0.021446609406726238f, 0.0014485813926750633f};

template <bool FAST>
__device__ __forceinline__ void polar(float xm, float xp, float& mag, float& ph, float& re,
                                      float& im, bool need_im) {
  if constexpr (FAST) {
    // hardware transcendentals: v_exp_f32 (2^x), v_sin_f32 / v_cos_f32 (argument in turns).
    // pi*sin(x) in turns is 0.5*sin(x): no multiply by pi on the path to cos/sin.
    mag = __builtin_amdgcn_exp2f(xm * 1.44269504088896341f);
    const float t = xp * 0.15915494309189535f;
    const float s = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(t));
    ph = kPi * s;
    re = mag * __builtin_amdgcn_cosf(0.5f * s);
    im = need_im ? mag * __builtin_amdgcn_sinf(0.5f * s) : 0.f;
  } else {
    mag = expf(xm);
    ph = kPi * sinf(xp);
    float sn, cs;
    sincosf(ph, &sn, &cs);
    re = mag * cs;
    im = mag * sn;
  }
}

}  // namespace

// Table `filt` (device, 320 floats):
//   [0, 256)    trainable-bank polyphase taps  t[band][p][i] = 4 h[band][3 - p + 4 i]
//   [256, 288)  fixed bank: c[k][q] = cos(theta_k(q)), k < 4, q < 8
//   [288, 352)  fixed bank: g[j] = 8 p[j] (-1)^(j/8), j < 63 (g[63] = 0)
template <int TM, int NTHREADS, bool FIXED, bool FAST>
__global__ __launch_bounds__(NTHREADS, (2048 / NTHREADS) * (NTHREADS / 256)) void istft_pqmf_kernel(const IstftArgs a, int tiles_per_utt,
                                                              int total_tiles) {
  constexpr int NF = TM / 4 + 7;          // frames a tile touches per band
  constexpr int NFS = ((NF + 31) / 32) * 32 + 8;   // LDS frame stride, == 8 (mod 32): conflict-free phase B
  constexpr int YL = TM + 16;             // sub-band samples incl. PQMF halo
  constexpr int NROW = FIXED ? 8 : 4;     // rows of the phase-B product (U_q or y_band)
  static_assert(4 * NF <= NTHREADS, "one lane per (band, frame)");
  static_assert(YL <= NTHREADS, "one lane per sub-band time index");
  static_assert(NROW * YL <= 4 * 16 * NFS, "phase-B product aliases the frame buffer");
  __shared__ __attribute__((aligned(16))) float fr[4 * 16 * NFS];
  float* const prod = fr;                 // reused after the frames are consumed

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
        float mag, ph;
        polar<FAST>(xin[k], xin[9 + k], mag, ph, re[k], im[k], k != 0 && k != 8);
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

  // ---------------- phase B: overlap-add + envelope (+ modulation) ---------
  float rowv[NROW];
  {
    const int u = tid;                    // m = m0 - 8 + u
    const int q = u >> 2, r = u & 3;      // quad f' = m0/4 - 2 + q ; frames f'-1 .. f'+2
    const int m = m0 - 8 + u;
    const int fp = m0 / 4 - 2 + q;
    float y[4] = {0.f, 0.f, 0.f, 0.f};
    if (u < YL && m >= 0 && m < M) {
      float env;
      if (fp - 1 >= 0 && fp + 2 < F) {
        env = 1.5f;                        // sum of squared hann over 4 overlapping frames
      } else {
        env = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f = fp - 1 + g;
          env += (f >= 0 && f < F) ? WSQ16[12 - 4 * g + r] : 0.f;
        }
      }
#pragma unroll
      for (int band = 0; band < 4; ++band) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)      // frame f'-1+g contributes its sample n = 12 - 4g + r
          s += fr[(band * 16 + 12 - 4 * g + r) * NFS + q + g];
        y[band] = s / env;
      }
      if (a.o_mb && u >= 8 && u < TM + 8) {          // owned samples m0 .. m0+TM-1
        if (!a.multistream) {
#pragma unroll
          for (int band = 0; band < 4; ++band) a.o_mb[((int64_t)b * 4 + band) * M + m] = y[band];
        } else {                                       // zero-stuffed x4, gain 4 (models.py:463)
#pragma unroll
          for (int band = 0; band < 4; ++band)
            *reinterpret_cast<float4*>(a.o_mb + ((int64_t)b * 4 + band) * 4 * M + 4 * (int64_t)m) =
                make_float4(4.f * y[band], 0.f, 0.f, 0.f);
        }
      }
    }
    if constexpr (FIXED) {
      const float* c = a.filt + 256;
#pragma unroll
      for (int qq = 0; qq < 8; ++qq)
        rowv[qq] = c[qq] * y[0] + c[8 + qq] * y[1] + c[16 + qq] * y[2] + c[24 + qq] * y[3];
    } else {
#pragma unroll
      for (int band = 0; band < 4; ++band) rowv[band] = y[band];
    }
  }
  __syncthreads();                        // every lane has consumed its frames: reuse the buffer
  if (tid < YL) {
#pragma unroll
    for (int k = 0; k < NROW; ++k) prod[k * YL + tid] = rowv[k];
  }
  __syncthreads();

  // ---------------- phase C: polyphase synthesis filter --------------------
  if (tid < TM) {
    const int m = m0 + tid;
    if (m < M) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (FIXED) {
        const float* g = a.filt + 288;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int j = 3 - p + 4 * i;           // tap index; y index m - 7 + i
            if (j <= 62) acc[p] = fmaf(g[j], prod[(j & 7) * YL + tid + 1 + i], acc[p]);
          }
        }
      } else {
#pragma unroll
        for (int band = 0; band < 4; ++band) {
          const float* yb = &prod[band * YL + tid + 1];        // y[m - 7 + i]
          const float* hb = a.filt + band * 64;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float yv = yb[i];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[p] = fmaf(hb[p * 16 + i], yv, acc[p]);
          }
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
  const dim3 grid(total), block(NT);
  if (a.fixed_bank) {
    if (a.exact_math) hipLaunchKernelGGL((istft_pqmf_kernel<TM, NT, true, false>), grid, block, 0, s, a, tiles_per_utt, total);
    else hipLaunchKernelGGL((istft_pqmf_kernel<TM, NT, true, true>), grid, block, 0, s, a, tiles_per_utt, total);
  } else {
    if (a.exact_math) hipLaunchKernelGGL((istft_pqmf_kernel<TM, NT, false, false>), grid, block, 0, s, a, tiles_per_utt, total);
    else hipLaunchKernelGGL((istft_pqmf_kernel<TM, NT, false, true>), grid, block, 0, s, a, tiles_per_utt, total);
  }
}

}  // namespace mbv
