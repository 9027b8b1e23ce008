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
//      fixed bank: 16 prototype taps each (63 FMA per lane instead of 252),
//      trainable bank: <= 16 taps x 4 bands each.  The zero-stuffed x4
//      upsampling never materialises; one 16-byte store per lane.
// Frames / samples in the halos are recomputed, not exchanged; consecutive
// tiles are mapped to the same XCD so halo rows hit in its L2.
#include "kernels.h"
#include <cstdlib>

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

// Fixed PQMF bank (pqmf.py:15-75: taps 62, cutoff 0.15, Kaiser beta 9), factorised:
//   4 h_k[j] = PQMF_G[j] * PQMF_C[k][j mod 8],  PQMF_C[k][q] = cos((2k+1)(pi/8)(q - 30.5) - (-1)^k pi/4),
//   PQMF_G[j] = 8 p[j] (-1)^(j/8).  Generated in float64 by scripts/gen_pqmf_tables.py, rounded to fp32.
__device__ constexpr float PQMF_C[32] = {
    9.807852507e-01f, 9.807852507e-01f, 8.314695954e-01f, 5.555702448e-01f,
    1.950903237e-01f, -1.950903237e-01f, -5.555702448e-01f, -8.314695954e-01f,
    -8.314695954e-01f, -8.314695954e-01f, 1.950903237e-01f, 9.807852507e-01f,
    5.555702448e-01f, -5.555702448e-01f, -9.807852507e-01f, -1.950903237e-01f,
    -5.555702448e-01f, -5.555702448e-01f, 9.807852507e-01f, -1.950903237e-01f,
    -8.314695954e-01f, 8.314695954e-01f, 1.950903237e-01f, -9.807852507e-01f,
    1.950903237e-01f, 1.950903237e-01f, -5.555702448e-01f, 8.314695954e-01f,
    -9.807852507e-01f, 9.807852507e-01f, -8.314695954e-01f, 5.555702448e-01f,
};
__device__ constexpr float PQMF_G[64] = {
    6.692762690e-05f, 2.144142782e-04f, 4.045689129e-04f, 4.907859839e-04f,
    2.202252799e-04f, -6.902719615e-04f, -2.394147683e-03f, -4.707115702e-03f,
    6.936517078e-03f, 7.863246836e-03f, 5.977601744e-03f, -6.432701142e-18f,
    -1.040009875e-02f, -2.390390635e-02f, -3.716831654e-02f, -4.507908970e-02f,
    4.180690646e-02f, 2.259947546e-02f, -1.405207906e-02f, -6.448587775e-02f,
    -1.188977659e-01f, -1.619237214e-01f, -1.750242710e-01f, -1.404098570e-01f,
    4.571796954e-02f, -1.117221490e-01f, -3.222790956e-01f, -5.640172958e-01f,
    -8.056510091e-01f, -1.012026548e+00f, -1.150984049e+00f, -1.200000048e+00f,
    1.150984049e+00f, 1.012026548e+00f, 8.056510091e-01f, 5.640172958e-01f,
    3.222790956e-01f, 1.117221490e-01f, -4.571796954e-02f, -1.404098570e-01f,
    1.750242710e-01f, 1.619237214e-01f, 1.188977659e-01f, 6.448587775e-02f,
    1.405207906e-02f, -2.259947546e-02f, -4.180690646e-02f, -4.507908970e-02f,
    3.716831654e-02f, 2.390390635e-02f, 1.040009875e-02f, 6.432701142e-18f,
    -5.977601744e-03f, -7.863246836e-03f, -6.936517078e-03f, -4.707115702e-03f,
    2.394147683e-03f, 6.902719615e-04f, -2.202252799e-04f, -4.907859839e-04f,
    -4.045689129e-04f, -2.144142782e-04f, -6.692762690e-05f, 0.000000000e+00f,
};

// PRE: the producing conv already scaled the log-magnitude rows by log2(e) and the phase rows by
// 1/(2 pi) (folded into subband_conv_post's packed weights), so exp2 / sin-in-turns apply directly.
template <bool FAST, bool PRE>
__device__ __forceinline__ void polar(float xm, float xp, float& mag, float& ph, float& re,
                                      float& im, bool need_im) {
  if constexpr (FAST) {
    // hardware transcendentals: v_exp_f32 (2^x), v_sin_f32 / v_cos_f32 (argument in turns, valid
    // for |turns| <= 256, i.e. |x| <= 1608 rad).  pi*sin(x) in turns is 0.5*sin(x): no multiply
    // by pi on the path to cos/sin.
    mag = __builtin_amdgcn_exp2f(PRE ? xm : xm * 1.44269504088896341f);
    const float s = __builtin_amdgcn_sinf(PRE ? xp : xp * 0.15915494309189535f);
    ph = kPi * s;
    re = mag * __builtin_amdgcn_cosf(0.5f * s);
    im = need_im ? mag * __builtin_amdgcn_sinf(0.5f * s) : 0.f;
  } else {
    if constexpr (PRE) { xm *= 0.69314718055994531f; xp *= 6.28318530717958648f; }
    mag = expf(xm);
    ph = kPi * sinf(xp);
    float sn, cs;
    sincosf(ph, &sn, &cs);
    re = mag * cs;
    im = mag * sn;
  }
}

// (magnitude, phase [rad]) given directly: the `istft_finalize` entry of the chunked-decode flow
template <bool FAST>
__device__ __forceinline__ void polar_in(float mag, float ph, float& re, float& im, bool need_im) {
  if constexpr (FAST) {
    const float t = ph * 0.15915494309189535f;          // radians -> turns
    re = mag * __builtin_amdgcn_cosf(t);
    im = need_im ? mag * __builtin_amdgcn_sinf(t) : 0.f;
  } else {
    float sn, cs;
    sincosf(ph, &sn, &cs);
    re = mag * cs;
    im = mag * sn;
  }
}

// 16-point real inverse DFT of a one-sided spectrum (Im of DC / Nyquist ignored, as c2r
// does), times hann(16)/16.  Packed real-IFFT: with A_k = X_k + conj(X_{8-k}),
// D_k = X_k - conj(X_{8-k}), Z_k = A_k + j W^k D_k (W = e^{j 2 pi/16}, k < 8) one has
// x[2n] + j x[2n+1] = (1/16) sum_k Z_k e^{j 2 pi k n / 8}; Z_{8-k} = conj(A_k - j W^k D_k).
// The 8-point complex inverse transform is two radix-4 butterflies + one radix-2 stage:
// ~110 flops instead of the 16 x 16 matrix form.
struct cpx { float r, i; };
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.r - b.r, a.i - b.i}; }
__device__ __forceinline__ void idft4(cpx y0, cpx y1, cpx y2, cpx y3, cpx o[4]) {
  const cpx t0 = cadd(y0, y2), t1 = csub(y0, y2), t2 = cadd(y1, y3), t3 = csub(y1, y3);
  o[0] = cadd(t0, t2);
  o[2] = csub(t0, t2);
  o[1] = {t1.r - t3.i, t1.i + t3.r};      // t1 + j t3
  o[3] = {t1.r + t3.i, t1.i - t3.r};      // t1 - j t3
}
__device__ __forceinline__ void irfft16_hann(const float* re, const float* im, float* out) {
  cpx Z[8];
  Z[0] = {re[0] + re[8], re[0] - re[8]};
  Z[4] = {2.f * re[4], -2.f * im[4]};
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const float Ar = re[k] + re[8 - k], Ai = im[k] - im[8 - k];
    const float Dr = re[k] - re[8 - k], Di = im[k] + im[8 - k];
    const float Br = -SIN16[k] * Dr - COS16[k] * Di;
    const float Bi = COS16[k] * Dr - SIN16[k] * Di;
    Z[k] = {Ar + Br, Ai + Bi};
    Z[8 - k] = {Ar - Br, Bi - Ai};
  }
  cpx E[4], O[4];
  idft4(Z[0], Z[2], Z[4], Z[6], E);
  idft4(Z[1], Z[3], Z[5], Z[7], O);
  constexpr float r = 0.70710678118654752f;
  const cpx T0 = O[0];
  const cpx T1 = {(O[1].r - O[1].i) * r, (O[1].r + O[1].i) * r};
  const cpx T2 = {-O[2].i, O[2].r};
  const cpx T3 = {(-O[3].r - O[3].i) * r, (O[3].r - O[3].i) * r};
  const cpx T[4] = {T0, T1, T2, T3};
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const cpx a = cadd(E[n], T[n]), b = csub(E[n], T[n]);
    out[2 * n] = a.r * (HANN16[2 * n] * (1.f / 16.f));
    out[2 * n + 1] = a.i * (HANN16[2 * n + 1] * (1.f / 16.f));
    out[2 * n + 8] = b.r * (HANN16[2 * n + 8] * (1.f / 16.f));
    out[2 * n + 9] = b.i * (HANN16[2 * n + 9] * (1.f / 16.f));
  }
}

}  // namespace

// `taps` (device, 256 floats, only read by the trainable-bank variant):
//   t[band][p][i] = 4 h[band][3 - p + 4 i]   (x4 up-sampling gain folded in; 0 where the tap is > 62)
template <int TM, int NTHREADS, bool FIXED, bool FAST, bool PRE, bool POLAR>
__global__ __launch_bounds__(NTHREADS, (2048 / NTHREADS) * (NTHREADS / 256)) void istft_pqmf_kernel(const IstftArgs a, const float* __restrict__ taps,
                                                              int tiles_per_utt, int total_tiles) {
  constexpr int NF = TM / 4 + 7;          // frames a tile touches per band
  constexpr int NFS = ((NF + 31) / 32) * 32 + 8;   // LDS frame stride, == 8 (mod 32): conflict-free phase B
  constexpr int YL = TM + 16;             // sub-band samples incl. PQMF halo
  constexpr int NROW = 4;                 // rows of the phase-B product (U_1..U_4 or y_band)
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
  // opt-in trimmed decode: sub-band samples at and beyond 64 * trim_lens[b] belong to no valid frame; a tile
  // wholly beyond is not computed (the caller zero-filled o), the tile across the boundary stores zeros there
  const int m_valid = a.trim_lens ? 64 * a.trim_lens[b] : 0x7fffffff;
  if (m0 >= m_valid) return;
  const int Tp = a.Tp;
  const int F = 16 * Tp + 1;
  const int M = 64 * Tp;                  // sub-band samples per band
  const int tid = threadIdx.x;
  const int f_lo = m0 / 4 - 3;
  // raw buffer descriptors (stride 0, byte range of the whole tensor; launcher checks < 4 GiB)
  constexpr int kRsrcFlags = 0x00020000;
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x_post), 0, POLAR ? 0 : a.B * 72 * F * 4, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t srsrc =
      __builtin_amdgcn_make_buffer_rsrc(a.spec, 0, a.spec ? a.B * 36 * F * 4 : 0, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t prsrc =
      __builtin_amdgcn_make_buffer_rsrc(a.phase, 0, a.phase ? a.B * 36 * F * 4 : 0, kRsrcFlags);

  // ---------------- phase A: frames --------------------------------------
  if (tid < 4 * NF) {
    const int band = tid / NF, fl = tid % NF;
    const int f = f_lo + fl;
    float out[16];
    if (f >= 0 && f < F) {
      float re[9], im[9];
      if constexpr (POLAR) {
        // input = (spec, phase) tensors [B, 4, 9, F] (chunked decode: cross-faded spectrograms)
        const int so = ((b * 4 + band) * 9 * F + f) * 4;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const float mag = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srsrc, so, k * F * 4, 0));
          const float ph = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, so, k * F * 4, 0));
          polar_in<FAST>(mag, ph, re[k], im[k], k != 0 && k != 8);
        }
      } else {
        // buffer loads: one 32-bit lane offset, the 18 channel strides ride in scalar registers
        const int voff = ((b * 72 + band * 18) * F + f) * 4;
        float xin[18];
#pragma unroll
        for (int k = 0; k < 18; ++k)
          xin[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, voff, k * F * 4, 0));
        // frame f is owned (for the spec/phase outputs) by the tile holding sample 4f
        const bool own = (4 * f >= m0 && 4 * f < m0 + TM) || (f == F - 1 && m0 + TM >= M);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          float mag, ph;
          polar<FAST, PRE>(xin[k], xin[9 + k], mag, ph, re[k], im[k], k != 0 && k != 8);
          if (own) {
            const int so = ((b * 4 + band) * 9 * F + f) * 4;
            if (a.nt_stores) {                         // (experiment, off by default: see the launcher)
              if (a.spec) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, mag), srsrc, so, k * F * 4, 2);
              if (a.phase) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ph), prsrc, so, k * F * 4, 2);
            } else {
              if (a.spec) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, mag), srsrc, so, k * F * 4, 0);
              if (a.phase) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ph), prsrc, so, k * F * 4, 0);
            }
          }
        }
      }
      irfft16_hann(re, im, out);
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
      const float renv = 1.f / env;          // one division per lane (torch.istft divides per sample)
#pragma unroll
      for (int band = 0; band < 4; ++band) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)      // frame f'-1+g contributes its sample n = 12 - 4g + r
          s += fr[(band * 16 + 12 - 4 * g + r) * NFS + q + g];
        y[band] = s * renv;
      }
      if (a.o_mb && u >= 8 && u < TM + 8) {          // owned samples m0 .. m0+TM-1
        if (!a.multistream) {
          if (a.nt_stores) {
#pragma unroll
            for (int band = 0; band < 4; ++band)
              __builtin_nontemporal_store(y[band], &a.o_mb[((int64_t)b * 4 + band) * M + m]);
          } else {
#pragma unroll
            for (int band = 0; band < 4; ++band) a.o_mb[((int64_t)b * 4 + band) * M + m] = y[band];
          }
        } else {                                       // zero-stuffed x4, gain 4 (models.py:463)
          typedef float f4v __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int band = 0; band < 4; ++band) {
            f4v v = {4.f * y[band], 0.f, 0.f, 0.f};
            f4v* dst = reinterpret_cast<f4v*>(a.o_mb + ((int64_t)b * 4 + band) * 4 * M + 4 * (int64_t)m);
            if (a.nt_stores) __builtin_nontemporal_store(v, dst); else *dst = v;
          }
        }
      }
    }
    if constexpr (FIXED) {
      // Of the 8 modulation phases only 4 are distinct: U_0 = U_1, U_5 = -U_4, U_6 = -U_3,
      // U_7 = -U_2 (theta_k(q) is symmetric about q = 0.5 and anti-symmetric about q = 4.5).
      // Row r holds U_{r+1}; the signs are folded into the tap constants of phase C.
#pragma unroll
      for (int r = 0; r < 4; ++r)
        rowv[r] = PQMF_C[r + 1] * y[0] + PQMF_C[8 + r + 1] * y[1] + PQMF_C[16 + r + 1] * y[2] +
                  PQMF_C[24 + r + 1] * y[3];
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
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int j = 3 - p + 4 * i;           // tap index; y index m - 7 + i
            if (j <= 62) {
              constexpr int ROW[8] = {0, 0, 1, 2, 3, 3, 2, 1};          // U_q -> stored row
              constexpr float SGN[8] = {1.f, 1.f, 1.f, 1.f, 1.f, -1.f, -1.f, -1.f};
              acc[p] = fmaf(PQMF_G[j] * SGN[j & 7], prod[ROW[j & 7] * YL + tid + 1 + i], acc[p]);
            }
          }
        }
      } else {
#pragma unroll
        for (int band = 0; band < 4; ++band) {
          const float* yb = &prod[band * YL + tid + 1];        // y[m - 7 + i]
          const float* hb = taps + band * 64;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float yv = yb[i];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[p] = fmaf(hb[p * 16 + i], yv, acc[p]);
          }
        }
      }
      *reinterpret_cast<float4*>(a.o + (int64_t)b * 4 * M + 4 * (int64_t)m) =
          m < m_valid ? make_float4(acc[0], acc[1], acc[2], acc[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <int TM, int NT>
static void launch_istft_pqmf_t(const IstftArgs& a, hipStream_t s) {
  const int M = 64 * a.Tp;
  const int tiles_per_utt = (M + TM - 1) / TM;
  const int total = tiles_per_utt * a.B;
  const dim3 grid(total), block(NT);
#define MBV_ISTFT_LAUNCH(FIXED, FAST, PRE, POLAR) \
  hipLaunchKernelGGL((istft_pqmf_kernel<TM, NT, FIXED, FAST, PRE, POLAR>), grid, block, 0, s, a, a.filt, tiles_per_utt, total)
  if (a.polar_in) {          // (spec, phase) input: a.spec / a.phase are read, not written
    const int v = (a.fixed_bank ? 2 : 0) | (a.exact_math ? 0 : 1);
    switch (v) {
      case 0: MBV_ISTFT_LAUNCH(false, false, false, true); break;
      case 1: MBV_ISTFT_LAUNCH(false, true, false, true); break;
      case 2: MBV_ISTFT_LAUNCH(true, false, false, true); break;
      default: MBV_ISTFT_LAUNCH(true, true, false, true); break;
    }
    return;
  }
  const int variant = (a.fixed_bank ? 4 : 0) | (a.exact_math ? 0 : 2) | (a.prescaled ? 1 : 0);
  switch (variant) {
    case 0: MBV_ISTFT_LAUNCH(false, false, false, false); break;
    case 1: MBV_ISTFT_LAUNCH(false, false, true, false); break;
    case 2: MBV_ISTFT_LAUNCH(false, true, false, false); break;
    case 3: MBV_ISTFT_LAUNCH(false, true, true, false); break;
    case 4: MBV_ISTFT_LAUNCH(true, false, false, false); break;
    case 5: MBV_ISTFT_LAUNCH(true, false, true, false); break;
    case 6: MBV_ISTFT_LAUNCH(true, true, false, false); break;
    default: MBV_ISTFT_LAUNCH(true, true, true, false); break;
  }
#undef MBV_ISTFT_LAUNCH
}

void launch_istft_pqmf(const IstftArgs& a_in, hipStream_t s) {
  IstftArgs a = a_in;
  // Measured and rejected (r02): (a) non-temporal stores for spec / phase / o_mb — 91 vs 84 us for the
  // all-outputs launch (MBV_ISTFT_NT=1 keeps the A/B); (b) a persistent grid with the next tile's 18
  // inputs prefetched into registers: needs 80 registers per lane = 3 instead of 4 workgroups per CU,
  // and loses more to the lower occupancy than the prefetch gains (35.9 / 39.3 vs 32.2 us);
  // (c) 16-byte accesses through in-register 4 x 4 transposes across lane quads (DPP): x_post loads
  // 36.6 vs 31.6 us, spec / phase stores 84.8 vs 80.0 us in the same run (an apparent 84 -> 75 us gain
  // was box-to-box variation).  All three are in the history of this file (r02).  What the launch is
  // bound by: bytes in flight per CU at full occupancy (4 x 512 threads, 18 loads per lane) against
  // the latency of the level that serves them — 0.81 of the HBM peak from the Infinity Cache, 0.63
  // from HBM itself (bench.py roofline.past_cache), 0.61-0.65 with all outputs written.
  static const int nt = [] { const char* e = getenv("MBV_ISTFT_NT"); return e ? atoi(e) : 0; }();
  a.nt_stores = nt;
  // 480 sub-band samples x 512 threads (4 workgroups / CU) by default; MBV_ISTFT_TILE=224 selects
  // the 224 x 256-thread shape (8 workgroups / CU) for A/B runs
  static const int tile = [] { const char* e = getenv("MBV_ISTFT_TILE"); return e ? atoi(e) : 480; }();
  if (tile == 224) launch_istft_pqmf_t<224, 256>(a, s);
  else if (tile == 960) launch_istft_pqmf_t<960, 1024>(a, s);
  else launch_istft_pqmf_t<480, 512>(a, s);
}

// ============================================================================
// Single-band tail of iSTFT_Generator (models.py:296-300): exp / pi*sin, TorchSTFT.inverse
// (n_fft 16, hop 4), no filter bank.  Phase A: one lane per frame (255 frames per workgroup);
// phase B: one lane per quad of output samples (252 quads): overlap-add of the 4 covering
// frames, edge-aware envelope, one 16-byte store.
// ============================================================================
template <bool FAST, bool PRE, bool POLAR>
__global__ __launch_bounds__(256) void istft_single_kernel(const IstftSbArgs a, int tiles_per_utt) {
  constexpr int QPB = 252;                 // output quads per workgroup
  constexpr int NFS = 256;
  __shared__ float fr[16 * NFS];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / tiles_per_utt;
  const int q0 = (blockIdx.x % tiles_per_utt) * QPB;
  const int F = a.F;
  const int nquads = F - 1;                // 4 (F-1) output samples

  {
    const int f = q0 - 1 + tid;
    float out[16];
    if (tid < QPB + 3 && f >= 0 && f < F) {
      float re[9], im[9];
      if constexpr (POLAR) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
          polar_in<FAST>(a.spec[((int64_t)b * 9 + k) * F + f], a.phase[((int64_t)b * 9 + k) * F + f],
                         re[k], im[k], k != 0 && k != 8);
      } else {
        const float* xp = a.x_post + (int64_t)b * 18 * F + f;
        float xin[18];
#pragma unroll
        for (int k = 0; k < 18; ++k) xin[k] = xp[(int64_t)k * F];
        const bool own = f >= q0 && (f < q0 + QPB || f == F - 1);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          float mag, ph;
          polar<FAST, PRE>(xin[k], xin[9 + k], mag, ph, re[k], im[k], k != 0 && k != 8);
          if (own) {
            if (a.spec) a.spec[((int64_t)b * 9 + k) * F + f] = mag;
            if (a.phase) a.phase[((int64_t)b * 9 + k) * F + f] = ph;
          }
        }
      }
      irfft16_hann(re, im, out);
    } else {
#pragma unroll
      for (int n = 0; n < 16; ++n) out[n] = 0.f;
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) fr[n * NFS + tid] = out[n];
  }
  __syncthreads();
  const int fp = q0 + tid;                 // quad f': samples 4 f' + r from frames f'-1 .. f'+2
  if (tid < QPB && fp < nquads) {
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sacc = 0.f, env = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f = fp - 1 + g;
        sacc += fr[(12 - 4 * g + r) * NFS + tid + g];
        env += (f >= 0 && f < F) ? HANN16[12 - 4 * g + r] * HANN16[12 - 4 * g + r] : 0.f;
      }
      y[r] = sacc / env;
    }
    *reinterpret_cast<float4*>(a.o + (int64_t)b * 4 * nquads + 4 * (int64_t)fp) =
        make_float4(y[0], y[1], y[2], y[3]);
  }
}

void launch_istft_single(const IstftSbArgs& a, hipStream_t s) {
  const int tiles = (a.F - 1 + 251) / 252;
  const dim3 grid(tiles * a.B), block(256);
  if (a.polar_in) {
    if (a.exact_math) hipLaunchKernelGGL((istft_single_kernel<false, false, true>), grid, block, 0, s, a, tiles);
    else hipLaunchKernelGGL((istft_single_kernel<true, false, true>), grid, block, 0, s, a, tiles);
    return;
  }
  const int variant = (a.exact_math ? 0 : 2) | (a.prescaled ? 1 : 0);
  switch (variant) {
    case 0: hipLaunchKernelGGL((istft_single_kernel<false, false, false>), grid, block, 0, s, a, tiles); break;
    case 1: hipLaunchKernelGGL((istft_single_kernel<false, true, false>), grid, block, 0, s, a, tiles); break;
    case 2: hipLaunchKernelGGL((istft_single_kernel<true, false, false>), grid, block, 0, s, a, tiles); break;
    default: hipLaunchKernelGGL((istft_single_kernel<true, true, false>), grid, block, 0, s, a, tiles); break;
  }
}

}  // namespace mbv
