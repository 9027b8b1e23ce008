// Windowed relative-position self-attention of the text encoder
// (attentions.py:148-179, helpers :181-243) on the fp32 matrix cores.
//
//   s[i,j] = (q_i/sqrt(d)) . k_j + [|j-i|<=4] (q_i/sqrt(d)) . Ek[j-i+4]
//   s[i,j] = -1e4 where query i or key j is padding           (attentions.py:166)
//   p      = softmax_j s
//   o_i    = sum_j p[i,j] v_j + sum_{r=-4..4} p[i,i+r] Ev[r+4]
//
// One workgroup of NW = 2 waves per (32-query tile, head, utterance); wave w owns key tiles
// w, w + NW, ...  At T ~ 200 (7 key tiles, 896 workgroups at batch 64) a tile's work is a
// latency-bound chain and there are fewer workgroups than SIMDs, so the chain is split across
// waves; the waves meet once through LDS at the end: running max / sum, partial O^T / band weights.  (Measured: 1 wave 205 us, 2 waves see profiles/README.md;
// 4 waves need 2 rounds of workgroups at 213 registers per lane and are slower.)  The score tile is computed
// TRANSPOSED (S^T = K^T Q, keys on the accumulator rows, queries on the lanes)
// so that (a) both MFMA operands are read time-contiguous straight from the
// [B, C, T] activations, (b) the softmax reductions run down a lane's own
// registers (one cross-half shuffle at the end), and (c) the probability tile
// is already the B operand of the P.V product (O^T = V P^T) with no data
// movement: k-step s of that product takes accumulator register s.
// One pass over the key tiles with running softmax statistics (r02; r01 made two passes and computed
// every score tile twice).  The relative-key logits are one extra MFMA tile
// (rows = the 9 embeddings), the relative-value term is accumulated as 9 band
// weights per query and applied to O^T at the end.
// r02h: Q lives in LDS as a k-interleaved image (one 16-byte read = four k-steps; it used to take 48 of the
// 254 registers), the score tile and the relative-key logits accumulate in 2 / 4 interleaved MFMA chains
// (a dependent fp32 32x32x2 MFMA issues ~250 cycles after its predecessor), k-steps past the head dimension
// multiply zeros instead of being tested for, and V never touches LDS: the four k-steps 4q .. 4q + 3 of the
// P.V product are four consecutive keys of a lane's V row, i.e. one 16-byte load straight into the operand.
#include "kernels.h"

namespace mbv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int acc_row(int reg, int hl) { return (reg & 3) + 8 * (reg >> 2) + 4 * hl; }

constexpr int ATT_NW = 2;          // waves per workgroup (key tiles are dealt round-robin)

template <int DT>
__global__ __launch_bounds__(64 * ATT_NW, 2) void rel_attention_kernel(const float* __restrict__ qkv,
                                                           const float* __restrict__ emb_k,
                                                           const float* __restrict__ emb_v,
                                                           const int* __restrict__ lens,
                                                           float* __restrict__ o, int H, int n_heads,
                                                           int T) {
  constexpr int DMAX = DT * 32;
  constexpr int NW = ATT_NW;
  constexpr int PER = DT * 16 + 9;                     // floats per lane in the final reduction
  constexpr int VALL = (NW - 1) * PER * 64;
  __shared__ float Vall[VALL];                         // the final reduction's hand-over
  __shared__ float stat[NW][2][64];                    // per-wave (max, sum) of pass 1

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, head = blockIdx.y;
  const int d = H / n_heads;
  const int tq0 = blockIdx.x * 32;
  const int tq = tq0 + l31;
  const int len = lens[b];
  const float inv = sqrtf((float)d);

  const float* qb = qkv + ((int64_t)b * 3 * H + head * d) * T;
  const float* kb = qb + (int64_t)H * T;
  const float* vb = kb + (int64_t)H * T;
  const int nsteps = d / 2;                           // MFMA k-steps over the head dim
  // K / V rows are read through raw buffer descriptors of this (utterance, head) slice: one 32-bit
  // lane offset per tile + a scalar row offset per k-step.  With 64-bit pointers hipcc hoists 2 x 48
  // row addresses per operand out of the key-tile loop and the kernel needs > 400 registers
  // (1 wave / SIMD, every latency exposed).  Out-of-range lanes (keys >= T) use an offset past the
  // slice and read 0.
  constexpr int kRsrcFlags = 0x00020000;
  constexpr int kOob = 0x7ffffff0;
  const __amdgpu_buffer_rsrc_t krsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kb), 0, d * T * 4, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t vrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vb), 0, d * T * 4, kRsrcFlags);
  const int row2 = 2 * T * 4;                         // bytes between k-steps (two rows)

  // Q fragments (B operand): B[k = 2s+hl][j = l31] = q[2s+hl][tq] / sqrt(d), in LDS, k-interleaved like the
  // conv kernels' images: one 16-byte read = the four k-steps 4g .. 4g + 3 of a lane.  (In registers they
  // were 48 of the kernel's 254; that room now holds a second score accumulator and V.)
  constexpr int QG = DMAX / 8;
  __shared__ f32x4 Qs[QG * 2 * 32];
  for (int e = threadIdx.x; e < QG * 2 * 32; e += 64 * NW) {
    const int g = e >> 6, h = (e >> 5) & 1, t = tq0 + (e & 31);
    f32x4 v;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int k = 4 * g + s4;
      v[s4] = (k < nsteps && t < T) ? qb[(int64_t)(2 * k + h) * T + t] / inv : 0.f;
    }
    Qs[e] = v;
  }
  // relative-key embeddings as the A operand of the logits below, in the same k-interleaved form: [g][h][row], rows
  // 9 .. 15 zero.  (r03: they used to be read from global memory inside the MFMA chain, `l31 < 9 ? emb_k[..] : 0` —
  // which hipcc turns into branch, load, s_waitcnt vmcnt(0), MFMA, 48 times in a row: 48 exposed load latencies in the
  // prologue of every workgroup.)
  __shared__ f32x4 Eks[QG * 2 * 16];
  for (int e = threadIdx.x; e < QG * 2 * 16; e += 64 * NW) {
    const int g = e >> 5, h = (e >> 4) & 1, row = e & 15;
    f32x4 v;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int kk = 2 * (4 * g + s4) + h;
      v[s4] = (row < 9 && kk < d) ? emb_k[row * d + kk] : 0.f;
    }
    Eks[e] = v;
  }
  __syncthreads();
  const f32x4* const ql = Qs + hl * 32 + l31;          // + 64 g
  const f32x4* const el = Eks + hl * 16 + min(l31, 15);   // + 32 g (lanes 16 .. 31 read the zero row 15)

  // ---- relative-key logits: R^T[r][tq] = sum_d Ek[r][d] q[d][tq] ------------
  // (four interleaved MFMA chains: an fp32 32x32x2 MFMA feeds the next one on the same accumulator only
  // ~250 cycles later, see conv1d_narrow.hip, and nothing else runs in this prologue)
  float rel[9];
  {
    f32x16 acc, accb[3];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accb[0][r] = 0.f; accb[1][r] = 0.f; accb[2][r] = 0.f; }
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      const f32x4 qv = ql[64 * g];
      const f32x4 ev = el[32 * g];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        // (k-steps past the head dimension multiply zeros by zeros: no test between the MFMAs)
        if (s4 == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[s4], qv[s4], acc, 0, 0, 0);
        else accb[s4 - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[s4], qv[s4], accb[s4 - 1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 5; ++r) acc[r] = (acc[r] + accb[0][r]) + (accb[1][r] + accb[2][r]);   // (only rows 0-8 are read)
    // rows 0-3 / 8 live in half 0 (regs 0-3 / 4), rows 4-7 in half 1 (regs 0-3)
    float mine[5], other[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) { mine[k] = acc[k]; other[k] = __shfl_xor(acc[k], 32); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rel[k] = hl == 0 ? mine[k] : other[k];
      rel[4 + k] = hl == 0 ? other[k] : mine[k];
    }
    rel[8] = hl == 0 ? mine[4] : other[4];
  }

  const int ntiles = (T + 31) / 32;
  const bool q_valid = tq < len;

  // K fragments of key tile kt (A operand of S^T = K^T Q): one latency round, requested early
  float kf[DMAX / 2];
  auto load_k = [&](int kt) {
    const int tkl = kt * 32 + l31;
    const int koff = (kt < ntiles && tkl < T) ? (hl * T + tkl) * 4 : kOob;
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s)
      kf[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(krsrc, koff, s * row2, 0));   // rows past d: past the slice, read 0
  };
  // score tile for key tile kt from kf, masked; rows beyond T get -inf (absent keys)
  auto score_tile = [&](int kt, f32x16& S) {
    const int tk0 = kt * 32;
    // two interleaved chains (with the wave sharing the SIMD: four)
    f32x16 S2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S[r] = 0.f; S2[r] = 0.f; }
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      const f32x4 qv = ql[64 * g];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (s4 & 1) S2 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[4 * g + s4], qv[s4], S2, 0, 0, 0);
        else S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[4 * g + s4], qv[s4], S, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] += S2[r];
    const bool near = (tk0 - tq0) <= 35 && (tq0 - tk0) <= 35;   // wave-uniform
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tk = tk0 + acc_row(r, hl);
      float sv = S[r];
      if (near) {
        const int rr = tk - tq + 4;
#pragma unroll
        for (int q = 0; q < 9; ++q) sv += (rr == q) ? rel[q] : 0.f;
      }
      if (!(q_valid && tk < len)) sv = -1e4f;
      if (tk >= T) sv = -INFINITY;
      S[r] = sv;
    }
  };

  // ---- ONE pass over this wave's key tiles, running (max, sum) per query ("online" softmax):
  //   m' = max(m, max_tile S);  P = exp(S - m');  l = l e^(m - m') + sum P;  O = O e^(m - m') + V P^T
  // and the same rescaling for the band weights; the division by l happens once at the end.  The
  // two-pass form (statistics first, then normalised P) computed every score tile twice; the results
  // differ only in rounding (P is normalised after the contraction instead of before).
  // Memory latency: V of this tile is requested before the score MFMAs, K of the wave's next tile right
  // after them (kf is dead by then), so both travel under the arithmetic of the current tile.
  f32x16 O[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
  float wb[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) wb[q] = 0.f;
  float mx = -INFINITY, sum = 0.f;         // sum: this lane half's share (its 16 keys per tile)

  load_k(wave);
  for (int kt = wave; kt < ntiles; kt += NW) {
    const int tk0 = kt * 32;
    // V of this tile straight into the A operands of O^T = V P^T: k-step s of that product is key
    // (s & 3) + 8 (s >> 2) + 4 hl of the tile (the accumulator row of probability register s), so the four
    // k-steps 4 q .. 4 q + 3 of a lane are FOUR CONSECUTIVE keys of its V row — one 16-byte load, no LDS.
    // (Keys past T read the next row's first values or, past the slice, zeros: finite, and their
    // probabilities are exactly 0.)  Requested before the score MFMAs, consumed after the softmax.
    f32x4 vf[DT][4];
    {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const int voff = ((t * 32 + l31) * T + tk0 + 4 * hl) * 4;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          vf[t][q4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, voff, q4 * 32, 0));
      }
    }
    f32x16 S;
    score_tile(kt, S);
    float tmax = S[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, S[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));          // both halves of a query's 32 keys
    const float mnew = fmaxf(mx, tmax);                // finite: every tile holds at least one key < T
    const float alpha = expf(mx - mnew);               // exp(-inf) = 0 on the first tile
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S[r] = expf(S[r] - mnew); part += S[r]; }
    sum = sum * alpha + part;
    mx = mnew;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
#pragma unroll
    for (int q = 0; q < 9; ++q) wb[q] *= alpha;
    const bool near = (tk0 - tq0) <= 35 && (tq0 - tk0) <= 35;
    if (near) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = tk0 + acc_row(r, hl) - tq + 4;
#pragma unroll
        for (int q = 0; q < 9; ++q) wb[q] += (rr == q) ? S[r] : 0.f;
      }
    }
    load_k(kt + NW);                       // next tile's K under the P.V MFMAs (past the last tile: nothing is read)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int t = 0; t < DT; ++t)
        O[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[t][s >> 2][s & 3], S[s], O[t], 0, 0, 0);
    }
  }
  sum += __shfl_xor(sum, 32);
#pragma unroll
  for (int q = 0; q < 9; ++q) wb[q] += __shfl_xor(wb[q], 32);
  // Relative values: O^T += Ev^T W, one more contraction over the 9 (padded to 10) relative positions on the matrix
  // pipe: A[dd][q] = emb_v[q][dd] (15 coalesced loads per lane, requested here, used after the waves are folded),
  // B[q][tq] = the band weights every lane holds for its own query.  (r03: this used to be, per accumulator
  // register, `if (dd < d && tq < T) { 9 loads of emb_v; 9 fma; store }` — 48 rounds of loads behind the previous
  // round's store, each waited for in full: most of the kernel's tail.)
  float evf[5][DT];
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int q = 2 * s + hl, dd = t * 32 + l31;
      evf[s][t] = (q < 9 && dd < d) ? emb_v[q * d + dd] : 0.f;
    }

  // ---- fold the waves' partial (m, l, O^T, band weights) into wave 0 -------------------------
  __syncthreads();                                    // every wave is done with its V tile
  if (wave > 0) {
    float* dst = Vall + ((wave - 1) * PER) * 64 + lane;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(t * 16 + r) * 64] = O[t][r];
#pragma unroll
    for (int q = 0; q < 9; ++q) dst[(DT * 16 + q) * 64] = wb[q];
    stat[wave][0][lane] = mx;
    stat[wave][1][lane] = sum;
  }
  __syncthreads();
  if (wave > 0) return;
  {
    float mall = mx;
#pragma unroll
    for (int w = 1; w < NW; ++w) mall = fmaxf(mall, stat[w][0][lane]);
    const float a0 = expf(mx - mall);
    sum *= a0;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[t][r] *= a0;
#pragma unroll
    for (int q = 0; q < 9; ++q) wb[q] *= a0;
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float mw = stat[w][0][lane];
      const float aw = mw > -INFINITY ? expf(mw - mall) : 0.f;      // a wave without key tiles: m = -inf, everything 0
      sum += stat[w][1][lane] * aw;
      const float* src = Vall + ((w - 1) * PER) * 64 + lane;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] = fmaf(src[(t * 16 + r) * 64], aw, O[t][r]);
#pragma unroll
      for (int q = 0; q < 9; ++q) wb[q] = fmaf(src[(DT * 16 + q) * 64], aw, wb[q]);
    }
    const float rsum = 1.f / sum;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[t][r] *= rsum;
#pragma unroll
    for (int q = 0; q < 9; ++q) wb[q] *= rsum;
  }

  // ---- relative values + store ------------------------------------------------
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const float w_lo = wb[2 * s], w_hi = 2 * s + 1 < 9 ? wb[2 * s + 1 < 9 ? 2 * s + 1 : 8] : 0.f;
    const float bw = hl ? w_hi : w_lo;                 // B[k = hl][j = l31]: band weight 2 s + hl of query l31
#pragma unroll
    for (int t = 0; t < DT; ++t) O[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(evf[s][t], bw, O[t], 0, 0, 0);
  }
  float* ob = o + ((int64_t)b * H + head * d) * T;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = t * 32 + acc_row(r, hl);
      if (dd < d && tq < T) ob[(int64_t)dd * T + tq] = O[t][r];
    }
  }
}

void launch_rel_attention(const float* qkv, const float* emb_k, const float* emb_v,
                          const int* lens, float* o, int B, int H, int n_heads, int T,
                          hipStream_t s) {
  const int d = H / n_heads;
  dim3 grid((T + 31) / 32, n_heads, B);
  if (d <= 32) hipLaunchKernelGGL((rel_attention_kernel<1>), grid, dim3(64 * ATT_NW), 0, s, qkv, emb_k, emb_v, lens, o, H, n_heads, T);
  else if (d <= 64) hipLaunchKernelGGL((rel_attention_kernel<2>), grid, dim3(64 * ATT_NW), 0, s, qkv, emb_k, emb_v, lens, o, H, n_heads, T);
  else if (d <= 96) hipLaunchKernelGGL((rel_attention_kernel<3>), grid, dim3(64 * ATT_NW), 0, s, qkv, emb_k, emb_v, lens, o, H, n_heads, T);
  else hipLaunchKernelGGL((rel_attention_kernel<4>), grid, dim3(64 * ATT_NW), 0, s, qkv, emb_k, emb_v, lens, o, H, n_heads, T);
}

}  // namespace mbv
