// conv1d for launches with few columns (single utterances, short sequences): the same contraction as
// conv1d.hip, decomposed the way wn_fused.hip is.
//
// conv1d.hip tiles the output 128 rows x 128 / 384 columns per workgroup; one utterance of 3 s is 34
// such tiles of a 128-channel decoder conv on a 256-CU chip, and the split-K detour that spreads them
// (partials through memory, ticket, ordered reduce) costs more than the contraction (74-95 us for
// ~1 GFLOP).  Here a work unit is 32 COLUMNS x one ROW BLOCK:
//   * columns: two half-units of 16 consecutive frames, each inside one utterance, numbered through the
//     batch (hu -> utterance hu / hpu, first frame 16 (hu % hpu)); tile c takes half-units 2c, 2c + 1,
//     so no column is padded beyond a multiple of 16;
//   * rows: the 4 waves of a 256-thread workgroup own row tiles wave, wave + 4, ... (NRT each) of a row
//     block of 128 NRT rows; row blocks of one column tile are separate workgroups (no reduction:
//     rows are independent), so a 128-channel conv on 4 240 frames is 133 workgroups of 4 x 1 tile;
//   * the A operand comes straight from L2 (a lane's four K-steps = one 16-byte load of the packed
//     weights conv1d.hip uses too), through a register ring 11 / 7 / 3 steps ahead (1 / 2 / >= 3 row tiles
//     per wave), after a prologue in which the workgroups of an XCD touch their row block's weight lines
//     once; the input window (all channels of a block of CB channels x 2 x (16 + halo) frames, activation /
//     mask / conditioning applied on the way in) is the only thing staged in LDS; two barriers per channel
//     block;
//   * a step (one tap of one 8-channel group = 4 K-steps) is 4 NRT MFMAs and a handful of scalar additions:
//     the cursors advance additively and the bookkeeping sits between the MFMAs (see the kernel);
//   * 192-row outputs use K-split wave pairs (template parameter KS).
// (Measured and dropped, r02: two column tiles per workgroup with the rows over two waves each — text
// encoder 3.03 -> 2.99 ms at batch 64 but 2.15 -> 2.53 ms at batch 32; the next block's window requested
// before the MFMA loop of the current one into a second LDS buffer: 2.78 -> 2.75 ms at batch 64, slower
// for a single utterance.)
// Epilogues: STORE (+ relu, + output mask), RESID, RESID_ACC, LN (conv -> (relu) -> + residual -> channel
// LayerNorm, one row block holding every channel) — what the text encoder, the duration predictor and the
// decoder's ResBlocks need; everything else stays on conv1d.hip.
#include "kernels.h"
#include <cstdio>
#include <cstdlib>

namespace mbv {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHalf = 16;
constexpr int kRsrcFlags = 0x00020000;
constexpr unsigned kOob = 0x7fffffffu;

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bstore1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ float lrelu_(float v, float slope) { return v > 0.f ? v : v * slope; }

// In-kernel timeline (scripts/narrow_stamps.hip builds this file with -DMBV_NARROW_STAMPS): wave 0 of every
// workgroup writes the 100 MHz wall clock at the phase boundaries of its first unit into a.ws.
#ifdef MBV_NARROW_STAMPS
#define MBV_STAMP(I)                                                                          \
  if (tid == 0 && u == (int)blockIdx.x) reinterpret_cast<unsigned long long*>(a.ws)[blockIdx.x * 8 + (I)] = __builtin_amdgcn_s_memrealtime();
#define MBV_CYCLES(I)                                                                         \
  if (tid == 0 && u == (int)blockIdx.x) reinterpret_cast<unsigned long long*>(a.ws)[blockIdx.x * 8 + (I)] = __builtin_readcyclecounter();
#else
#define MBV_STAMP(I)
#define MBV_CYCLES(I)
#endif

struct NarrowGeom {
  int hpu;        // half-units per utterance: ceil(T / 16)
  int n_ctiles;   // column tiles: ceil(B hpu / 2)
  int n_rblk;     // row blocks of 128 NRT rows
  int CB;         // channels per staged block (multiple of 8, divides Cin)
  int XS;         // column stride of a half-unit's window in the LDS image (multiple of 16, >= 16 + halo)
  int pf_period;  // lcm(8, n_rblk): workgroups blockIdx = i (mod pf_period) share an XCD and a row block
};

// KS ("K-split pairs"): for heights that four waves split badly (6 row tiles = 2 / 2 / 1 / 1 with a quarter of
// the MFMAs spent on tiles that do not exist), waves 0 / 1 own row tiles 0 .. NRT-1 / NRT .. 2 NRT-1 of a
// 64 NRT-row block and waves 2 / 3 the same tiles again: each pair splits the K loop (even / odd steps) and
// waves 2 / 3 hand their partial sums over through LDS before the epilogue, which waves 0 / 1 run alone.
template <int NRT, int EPI, bool KS = false>
__global__ __launch_bounds__(256, 2) void conv1d_narrow_kernel(const ConvArgs a, const NarrowGeom gm) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4* const Xs = reinterpret_cast<f32x4*>(lds);                     // [CB / 8][2][2 XS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = lane >> 5, l31 = lane & 31;
  const int half = l31 >> 4, jl = l31 & 15;
  const int K = a.K, T = a.T, M = a.M;
  const int G = a.Cin / 8, GB = gm.CB / 8, XL = 2 * gm.XS;
  const int W = kHalf + (K - 1) * a.dil;                               // window of a half-unit
  const int nblk = a.Cin / gm.CB;
  const int steps_blk = GB * K;
  const int n_hu = a.B * gm.hpu;
  constexpr int kRowsBlk = (KS ? 64 : 128) * NRT;                      // rows of a row block
  constexpr unsigned kJS = KS ? 512u : 2048u;                          // byte distance of a wave's consecutive row tiles in the packed weights
  const int tile0 = KS ? (wave & 1) * NRT : wave;                      // a wave's row tiles: tile0 + j * kTS
  constexpr int kTS = KS ? 1 : 4;
  const int kh = KS ? wave >> 1 : 0;                                   // KS: which half of the steps
  const bool idle = KS && kh;                                          // KS: no start values, no epilogue

  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, K * a.Cin * a.Mpad * 4, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias ? M * 4 : 0, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)(a.B * a.x_bstride) * 4u, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)(a.B * a.y_bstride) * 4u, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res), 0, a.res ? (unsigned)(a.B * a.res_bstride) * 4u : 0u, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t ac_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.accum_in), 0, a.accum_in ? (unsigned)(a.B * a.y_bstride) * 4u : 0u, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t rc_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res_chan_add), 0, a.res_chan_add ? a.B * M * 4 : 0, kRsrcFlags);
  const unsigned w_step = (unsigned)(2 * a.Mpad * 16);                 // bytes per (tap, group) step
  const unsigned rowT = (unsigned)T * 4u;

  // Weight prefetch into this XCD's L2.  A single utterance is ~130 workgroups that all stream the SAME
  // weights (0.2 - 1.8 MB per conv, read once per infer, so they come from the Infinity Cache / HBM) through
  // a ring that looks ~1 us ahead: the whole launch advances at one memory latency per ring depth (measured:
  // 46 - 50 us for 12 us of MFMA work).  So before anything else the workgroups of one XCD that share a row
  // block touch that row block's weight lines once, spread over their threads (one 128-byte line per load,
  // kPF loads per thread): the K loop then streams from L2.  The values are never used (the empty asm at
  // the end only keeps the loads alive; they return in order, long before anything waits for them).
  constexpr int kPF = 4;
  float pfv[kPF];
  {
    const int lpr = kRowsBlk / 8;                                      // 128-byte lines per (tap, group, h) run of a row block
    const int n_lines = K * G * 2 * lpr;
    const int first_rb = blockIdx.x % gm.n_rblk;
    const int jj = blockIdx.x / gm.pf_period;
    const int n_same = (gridDim.x - blockIdx.x % gm.pf_period + gm.pf_period - 1) / gm.pf_period;
#pragma unroll
    for (int i = 0; i < kPF; ++i) {
      const int idx = (jj + i * n_same) * 256 + tid;
      const int run = idx / lpr, within = idx - run * lpr;
      const unsigned vo = idx < n_lines ? (unsigned)((run * a.Mpad + first_rb * kRowsBlk) * 16 + within * 128) : kOob;
      pfv[i] = bload1(w_rs, vo, 0);
    }
  }

  // A ring: kD slots, loads run kD - 1 steps ahead.  A step is 4 NRT MFMAs (256 NRT cycles), the ring has
  // to cover an L2 round trip (~1-2 k cycles under load): deep for one row tile per wave, shallow for six.
  constexpr int kD = NRT == 1 ? 12 : (NRT == 2 ? 8 : 4);
  const int n_units = gm.n_ctiles * gm.n_rblk;
  for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
    MBV_STAMP(0)
    const int ct = u / gm.n_rblk, rb = u - ct * gm.n_rblk;
    // the two half-units: utterance, first frame, "exists"
    int hb[2], ht0[2], hok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int hu = 2 * ct + k;
      hok[k] = hu < n_hu;
      const int huc = hok[k] ? hu : n_hu - 1;
      hb[k] = huc / gm.hpu;
      ht0[k] = (huc - hb[k] * gm.hpu) * kHalf;
    }
    // input lengths of the two half-units' utterances, looked up ONCE per unit.  (Inside the staging loop —
    // `a.in_lens ? min(a.in_lens[bb], a.Tin) : a.Tin` per item, r02 — every item's lookup was a load followed by
    // s_waitcnt vmcnt(0), which also drained the window loads of the item before it: the staging of a channel block
    // was 2 NXI dependent memory round trips instead of one batch.  r03 audit)
    int lim_h[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)          // (readfirstlane: a scalar load — it must not sit on the vector-memory counter in front of the staging loop)
      lim_h[k] = a.in_lens ? min(a.in_lens[__builtin_amdgcn_readfirstlane(hb[k])], a.Tin) : a.Tin;
    const int b = half ? hb[1] : hb[0];
    const int t = (half ? ht0[1] : ht0[0]) + jl;                       // this lane's output frame
    const bool tv = (half ? hok[1] : hok[0]) && t < T;
    const int row_blk0 = rb * kRowsBlk;
    const unsigned w_voff = (unsigned)((hl * a.Mpad + row_blk0 + tile0 * 32 + l31) * 16);
    // first row of the wave's j-th tile as the start values / epilogue see it: past M for a wave without an
    // epilogue, so that every `row0 < M` test below skips it (and its bias loads read zeros)
#define MBV_ROW0(J) (idle ? M + 32 : row_blk0 + (tile0 + (J) * kTS) * 32)
    const unsigned y_voff = tv ? (unsigned)(b * (int)a.y_bstride + 4 * hl * T + t) * 4u : kOob;
    const int xoff = half * gm.XS + jl;

    // ---- accumulators start from bias (+ what the epilogue would have to read) -----------------
    f32x16 acc[NRT];
    {
      const unsigned bvo = (unsigned)(4 * hl) * 4u;
      const unsigned r_voff = tv ? (unsigned)(b * (int)a.res_bstride + 4 * hl * T + t) * 4u : kOob;
      const unsigned rc_voff = (unsigned)(b * M + 4 * hl) * 4u;
#pragma unroll
      for (int j = 0; j < NRT; ++j) {
        const int row0 = MBV_ROW0(j);
#pragma unroll
        // Every load below is UNCONDITIONAL: an absent operand is a zero-sized buffer (its loads return 0), a tile
        // slot past M reads values nobody stores.  With `if (row0 < M) x += load` in this nest (r02) hipcc emitted
        // branch, load, s_waitcnt vmcnt(0) per element — 16 NRT (running sum: 32 NRT) memory round trips one after the
        // other at the start of every unit (r03, scripts/asm_serial_loads.py); now they are one batch.
        for (int q = 0; q < 4; ++q) {
          f32x4 v = bload4(b_rs, bvo, (unsigned)(row0 + 8 * q) * 4u);  // past M / no bias: reads 0
          if constexpr (EPI == EPI_RESID || EPI == EPI_RESID_ACC) v += bload4(rc_rs, rc_voff, (unsigned)(row0 + 8 * q) * 4u);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            float x = v[s];
            if constexpr (EPI == EPI_RESID || EPI == EPI_RESID_ACC) x += bload1(r_rs, r_voff, (unsigned)(row0 + 8 * q + s) * rowT);
            if constexpr (EPI == EPI_RESID_ACC) x += bload1(ac_rs, y_voff, (unsigned)(row0 + 8 * q + s) * rowT);
            acc[j][4 * q + s] = x;
          }
        }
      }
    }

    MBV_STAMP(1)
    // Independent MFMA chains per wave.  A 32x32x2 fp32 MFMA occupies the pipe for 64 cycles but its result
    // feeds the next MFMA on the same accumulator only ~250 cycles later (scripts/narrow_stamps.hip: with one
    // workgroup per CU, two chains ran at 132 cycles per MFMA, four at 60), so a wave needs four accumulators
    // in flight: the K-steps of a group go round-robin over kExtra + 1 accumulator sets per row tile, summed
    // before the epilogue.
    constexpr int kExtra = NRT == 1 ? 3 : (NRT == 2 ? 1 : 0);
    constexpr bool kTwoAcc = kExtra > 0;
    f32x16 acc2[kTwoAcc ? kExtra * NRT : 1];
    if constexpr (kTwoAcc) {
#pragma unroll
      for (int j = 0; j < kExtra * NRT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
    }
    for (int blk = 0; blk < nblk; ++blk) {
      const int g0 = blk * GB;
      // ---- A ring: the first kD - 1 steps of this block (step = g * K + tap inside the block).  The step
      // cursors advance by additions only (a wave has 4 NRT MFMAs per step to hide its scalar work behind,
      // and with one wave per SIMD nothing else hides it: the multiply / clamp form of this bookkeeping
      // cost ~500 cycles per step, twice the MFMA time of one row tile).  Loads that run past the block
      // read the next block's weights or, past the tensor, zeros (buffer range check): never used.
      f32x4 ra[kD][NRT];
      const unsigned dWt = (unsigned)G * w_step, dWw = w_step - (unsigned)K * dWt;   // (the wrap comes on top of a tap step)
      unsigned so = (unsigned)g0 * w_step;
      int ltap = 0;
#define MBV_ADV_W1() { so += dWt; if (++ltap == K) { ltap = 0; so += dWw; } }
#define MBV_ADV_W() { MBV_ADV_W1() if constexpr (KS) MBV_ADV_W1() }    /* KS: a wave takes every second step */
      if (KS && kh) MBV_ADV_W1()
#pragma unroll
      for (int d = 0; d < kD - 1; ++d) {
#pragma unroll
        for (int j = 0; j < NRT; ++j) ra[d][j] = bload4(w_rs, w_voff, so + j * kJS);
        MBV_ADV_W()
      }
      __syncthreads();                                                 // the previous block's / unit's readers of Xs are done
      if (blk == 0) { MBV_STAMP(2) }
      // ---- input window of this channel block: [GB][2][2 XS], activated, masked.  The loads of NXI
      // items per thread are all issued before the first is used (one memory latency per batch
      // instead of one per item: at batch 1 a workgroup has nothing else to hide them behind).
      {
        constexpr int NXI = NRT <= 4 ? 8 : 4;
        const int items = GB * 2 * XL;
        const unsigned rs2 = 2u * (unsigned)a.x_rstride * 4u;
        for (int e0 = 0; e0 < items; e0 += 256 * NXI) {
          f32x4 xw[NXI];
          int chs[NXI];
#pragma unroll
          for (int i = 0; i < NXI; ++i) {
            const int e = e0 + tid + 256 * i;
            const int P = e / XL, cc = e - P * XL;
            const int k = cc >= gm.XS, c2 = cc - k * gm.XS;
            const int bb = k ? hb[1] : hb[0];
            const int ti = (k ? ht0[1] : ht0[0]) - a.pad_left + c2;
            const int lim = k ? lim_h[1] : lim_h[0];
            const int ch = (g0 + (P >> 1)) * 8 + (P & 1);
            const bool ok = e < items && c2 < W && (k ? hok[1] : hok[0]) && ti >= 0 && ti < lim;
            const unsigned vo = ok ? (unsigned)(bb * (int)a.x_bstride + ch * a.x_rstride + ti) * 4u : kOob;
            xw[i][0] = bload1(x_rs, vo, 0); xw[i][1] = bload1(x_rs, vo, rs2);
            xw[i][2] = bload1(x_rs, vo, 2 * rs2); xw[i][3] = bload1(x_rs, vo, 3 * rs2);
            chs[i] = ok ? bb * a.Cin + ch : -1;
          }
#pragma unroll
          for (int i = 0; i < NXI; ++i) {
            const int e = e0 + tid + 256 * i;
            if (e < items) {
              f32x4 v = xw[i];
              if (chs[i] >= 0) {
                if (a.chan_add) {
                  const float* ca = a.chan_add + chs[i];
                  v[0] += ca[0]; v[1] += ca[2]; v[2] += ca[4]; v[3] += ca[6];
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) v[s4] = lrelu_(v[s4], a.in_slope);
              }
              Xs[e] = v;
            }
          }
        }
      }
      __syncthreads();
      if (blk == 0) { MBV_STAMP(3) MBV_CYCLES(6) }

      // ---- MFMA over the block's (group, tap) steps ----------------------------------------------
      {
        static_assert(kD % 2 == 0, "the B operand ping-pongs between two registers by step parity");
        const f32x4* xl = Xs + hl * XL + xoff;
        const int dXw = 2 * XL - K * a.dil, xo_max = (GB - 1) * 2 * XL + (K - 1) * a.dil;
        int xo = 0, ntap = 0;             // window offset of the step after the one being multiplied
#define MBV_ADV_X1() { xo += a.dil; if (++ntap == K) { ntap = 0; xo += dXw; } }
#define MBV_ADV_X() { MBV_ADV_X1() if constexpr (KS) MBV_ADV_X1() }
        if (KS && kh) MBV_ADV_X1()
        f32x4 bvv[2];
        bvv[0] = xl[xo];
        MBV_ADV_X()
        const int my_steps = KS ? steps_blk / 2 : steps_blk;           // (KS: the host makes steps_blk even)
        for (int s0 = 0; s0 < my_steps; s0 += kD) {
#pragma unroll
          for (int d = 0; d < kD; ++d) {
            if (s0 + d < my_steps) {
              // One step = 4 K-steps x NRT row tiles.  The bookkeeping (next ring load, next window read,
              // cursor arithmetic) is issued BETWEEN the MFMAs: a wave stalls at every MFMA until the pipe
              // takes it, so scalar work placed there is free, while in front of the first MFMA of a step it
              // leaves the pipe idle (one wave per SIMD at batch 1).
#define MBV_MMA(S4)                                                                                         \
              _Pragma("unroll") for (int j = 0; j < NRT; ++j) {                                             \
                constexpr int set = (S4) % (kExtra + 1);                                                    \
                if constexpr (set != 0)                                                                     \
                  acc2[(set - 1) * NRT + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[d][j][S4], bvv[d & 1][S4], acc2[(set - 1) * NRT + j], 0, 0, 0); \
                else                                                                                        \
                  acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[d][j][S4], bvv[d & 1][S4], acc[j], 0, 0, 0);                                        \
              }
              __builtin_amdgcn_sched_barrier(0);
              MBV_MMA(0)
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < NRT; ++j) {
#if defined(MBV_NARROW_STAMPS) && MBV_NARROW_EXP == 1      // timeline experiment: no weight loads inside the loop
                asm volatile("" : "+v"(ra[(d + kD - 1) % kD][j]) : "s"(so));
#else
                ra[(d + kD - 1) % kD][j] = bload4(w_rs, w_voff, so + j * kJS);
#endif
              }
              MBV_ADV_W()
              __builtin_amdgcn_sched_barrier(0);
              MBV_MMA(1)
              __builtin_amdgcn_sched_barrier(0);
              bvv[(d + 1) & 1] = xl[min(xo, xo_max)];
              MBV_ADV_X()
              __builtin_amdgcn_sched_barrier(0);
              MBV_MMA(2)
              MBV_MMA(3)
              __builtin_amdgcn_sched_barrier(0);
#undef MBV_MMA
            }
          }
        }
#undef MBV_ADV_X
#undef MBV_ADV_X1
#undef MBV_ADV_W
#undef MBV_ADV_W1
      }
    }
    if constexpr (KS) {
      // waves 2 / 3 hand their halves over: [pair][tile][register][lane], lane-contiguous
      __syncthreads();                                                 // every wave is done with the input window
      float* const xch = lds + (size_t)((wave & 1) * NRT) * 16 * 64 + lane;
      if (kh) {
#pragma unroll
        for (int j = 0; j < NRT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) xch[(j * 16 + r) * 64] = acc[j][r];
      }
      __syncthreads();
      if (!kh) {
#pragma unroll
        for (int j = 0; j < NRT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] += xch[(j * 16 + r) * 64];
      }
    }

    MBV_STAMP(4) MBV_CYCLES(7)
    if constexpr (EPI == EPI_LN) {
      // ---- conv -> (relu) -> (mask) -> + residual -> channel LayerNorm -> affine -> (mask): the workgroup
      // holds every channel of its 32 frames (one row block), the statistics of a frame are reduced over
      // the lane's registers, its two lane halves and the four waves (LDS); two passes like F.layer_norm.
      const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ln_gamma), 0, M * 4, kRsrcFlags);
      const __amdgpu_buffer_rsrc_t be_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ln_beta), 0, M * 4, kRsrcFlags);
      const unsigned r_voff = tv ? (unsigned)(b * (int)a.res_bstride + 4 * hl * T + t) * 4u : kOob;
      const bool keep = !a.out_lens || t < a.out_lens[b];
      float part = 0.f;
      // the residual of row tile j + 1 is requested (16 loads in one batch, unconditional: no residual = a zero-sized
      // buffer) before row tile j is worked on; nothing in the arithmetic below branches.  (r02 had `if (a.res && ..) v +=
      // load` between an `if (a.relu)` and an `if (row0 >= M)`: hipcc kept every load in its own basic block and waited
      // for it there — 16 NRT memory round trips in a row in every unit's epilogue.)
      const float relu_floor = a.relu ? 0.f : -INFINITY;
      float rv[2][16];
#define MBV_LN_RES(J, BUF)                                                                      \
      _Pragma("unroll") for (int r = 0; r < 16; ++r)                                            \
        rv[BUF][r] = bload1(r_rs, r_voff, (unsigned)(MBV_ROW0(J) + (r & 3) + 8 * (r >> 2)) * rowT);
      MBV_LN_RES(0, 0)
#pragma unroll
      for (int j = 0; j < NRT; ++j) {
        const int row0 = MBV_ROW0(j);
        if (j + 1 < NRT) { MBV_LN_RES(j + 1, (j + 1) & 1) }
        __builtin_amdgcn_sched_barrier(0);
        const float live = row0 >= M ? 0.f : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[j][r];
          if constexpr (kExtra == 1) v += acc2[j][r];
          if constexpr (kExtra == 3) v = (v + acc2[j][r]) + (acc2[NRT + j][r] + acc2[2 * NRT + j][r]);
          v = fmaxf(v, relu_floor);
          v = keep ? v : 0.f;
          v += rv[j & 1][r];
          v = live != 0.f ? v : 0.f;
          acc[j][r] = v;
          part += v;
        }
      }
#undef MBV_LN_RES
      part += __shfl_xor(part, 32);
      float* red = lds;                                                // [2][4 waves][32 frames]
      // gamma / beta through LDS ([M] each, behind the statistics): read back below by ds_read_b128 between the stores.
      // (As buffer loads inside the store loop — r02 — every (tile, q) pair of loads queued behind the four stores before
      // it on the one in-order memory counter and was waited for in full: 4 NRT store round trips per unit.)
      float* const gam_s = lds + 256;
      float* const bet_s = gam_s + M;
      __syncthreads();                                                 // every wave is done with the input window
      if (hl == 0) red[wave * 32 + l31] = part;
      for (int e = tid; e < M; e += 256) {
        gam_s[e] = bload1(g_rs, (unsigned)e * 4u, 0);
        bet_s[e] = bload1(be_rs, (unsigned)e * 4u, 0);
      }
      __syncthreads();
      const float mean = (red[l31] + red[32 + l31] + red[64 + l31] + red[96 + l31]) / (float)M;
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < NRT; ++j) {
        const int row0 = MBV_ROW0(j);
        if (row0 < M) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float d = acc[j][r] - mean; sq += d * d; }
        }
      }
      sq += __shfl_xor(sq, 32);
      if (hl == 0) red[128 + wave * 32 + l31] = sq;
      __syncthreads();
      const float var = (red[128 + l31] + red[160 + l31] + red[192 + l31] + red[224 + l31]) / (float)M;
      const float rstd = rsqrtf(var + 1e-5f);
      const float lm = (a.ln_out_lens && t >= a.ln_out_lens[b]) ? 0.f : 1.f;
      const unsigned bvo = (unsigned)(4 * hl) * 4u;
#pragma unroll
      for (int j = 0; j < NRT; ++j) {
        const int row0 = MBV_ROW0(j);
        if (row0 >= M) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 gq = *reinterpret_cast<const f32x4*>(gam_s + row0 + 8 * q + 4 * hl);
          const f32x4 bq = *reinterpret_cast<const f32x4*>(bet_s + row0 + 8 * q + 4 * hl);
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2)
            bstore1(((acc[j][4 * q + s2] - mean) * rstd * gq[s2] + bq[s2]) * lm, y_rs, y_voff, (unsigned)(row0 + 8 * q + s2) * rowT);
        }
      }
    } else
    // ---- stores (masked lanes' stores are dropped by the range check) ------------------------------
    {
      const bool keep = !a.out_lens || t < a.out_lens[b];
#pragma unroll
      for (int j = 0; j < NRT; ++j) {
        const int row0 = MBV_ROW0(j);
        if (row0 >= M) continue;                                       // idle tile slot (rows past M would land in the next utterance)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = (r & 3) + 8 * (r >> 2);
          float v = acc[j][r];
          if constexpr (kExtra == 1) v += acc2[j][r];
          if constexpr (kExtra == 3) v = (v + acc2[j][r]) + (acc2[NRT + j][r] + acc2[2 * NRT + j][r]);
          if constexpr (EPI == EPI_STORE) {
            if (a.relu) v = fmaxf(v, 0.f);
            if (!keep) v = 0.f;
          } else if constexpr (EPI == EPI_RESID_ACC) {
            v *= a.out_scale;
          }
          if (row0 + k + 4 * hl < M) bstore1(v, y_rs, y_voff, (unsigned)(row0 + k) * rowT);
        }
      }
    }
    MBV_STAMP(5)
#undef MBV_ROW0
  }
#pragma unroll
  for (int i = 0; i < kPF; ++i) asm volatile("" ::"v"(pfv[i]));
}

template <int NRT, int EPI, bool KS = false>
void launch_narrow_t(const ConvArgs& a, const NarrowGeom& gm, hipStream_t s) {
  size_t lds_bytes = (size_t)(gm.CB / 8) * 2 * 2 * gm.XS * 16;
  if (EPI == EPI_LN && lds_bytes < (size_t)(256 + 2 * a.M) * 4) lds_bytes = (size_t)(256 + 2 * a.M) * 4;   // statistics + gamma / beta
  if (KS && lds_bytes < (size_t)2 * NRT * 16 * 64 * 4) lds_bytes = (size_t)2 * NRT * 16 * 64 * 4;   // the pairs' hand-over
  const long units = (long)gm.n_ctiles * gm.n_rblk;
  const int grid = (int)(units < 512 ? (units < 1 ? 1 : units) : 512);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_narrow_kernel<NRT, EPI, KS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((conv1d_narrow_kernel<NRT, EPI, KS>), dim3(grid), dim3(256), lds_bytes, s, a, gm);
}

template <int NRT, bool KS = false>
void launch_narrow_epi(const ConvArgs& a, const NarrowGeom& gm, hipStream_t s) {
  switch (a.epi) {
    case EPI_STORE: launch_narrow_t<NRT, EPI_STORE, KS>(a, gm, s); break;
    case EPI_RESID: launch_narrow_t<NRT, EPI_RESID, KS>(a, gm, s); break;
    case EPI_LN: launch_narrow_t<NRT, EPI_LN, KS>(a, gm, s); break;
    default: launch_narrow_t<NRT, EPI_RESID_ACC, KS>(a, gm, s); break;
  }
}

}  // namespace

// What the narrow kernel covers: the three plain epilogues, rows in whole 32-row tiles, no reflect
// padding, tensors addressable with 32-bit byte offsets.
bool conv1d_narrow_supported(const ConvArgs& a) {
  if (a.epi != EPI_STORE && a.epi != EPI_RESID && a.epi != EPI_RESID_ACC && a.epi != EPI_LN) return false;
  if (a.epi == EPI_LN && (a.M > 768 || !a.ln_gamma || !a.ln_beta)) return false;      // one row block holds every channel
  if (a.reflect1 || a.M % 32 || a.Cin % 8 || a.K < 1 || a.K > 11) return false;
  if (a.x_rstride != a.Tin && a.x_rstride < a.Tin) return false;
  const unsigned long long lim = 1ull << 31;
  if ((unsigned long long)a.B * a.x_bstride * 4 >= lim || (unsigned long long)a.B * a.y_bstride * 4 >= lim) return false;
  if (a.res && (unsigned long long)a.B * a.res_bstride * 4 >= lim) return false;
  const int W = kHalf + (a.K - 1) * a.dil;
  return W <= 96;
}

// by_launch_size: pick the row-block height from the number of column tiles (the low-latency mode: more,
// smaller workgroups for single utterances).  false: from M alone, so that the summation order of a
// row never depends on what else is in the batch (the default mode's bitwise batch independence).
void launch_conv1d_narrow(const ConvArgs& a, bool by_launch_size, hipStream_t s) {
  if (a.epi == EPI_LN) by_launch_size = false;       // the LayerNorm needs the whole height in one workgroup
  NarrowGeom gm;
  gm.hpu = (a.T + kHalf - 1) / kHalf;
  gm.n_ctiles = (int)(((long)a.B * gm.hpu + 1) / 2);
  const int W = kHalf + (a.K - 1) * a.dil;
  gm.XS = (W + 15) / 16 * 16;
  // channel block: the largest divisor of Cin (in groups of 8) whose window image fits 64 KB (two
  // workgroups per CU) — or 128 KB when there is at most one workgroup per CU anyway
  const size_t lds_cap = (by_launch_size && (long)gm.n_ctiles * ((a.M + 127) / 128) <= 256) ? 128 * 1024 : 64 * 1024;
  int cb = a.Cin;
  while (cb > 8 && ((size_t)cb * gm.XS * 8 > lds_cap || a.Cin % cb)) cb -= 8;
  gm.CB = cb;
  // row tiles per wave: few columns -> small row blocks (more workgroups); many columns -> the whole
  // height in one workgroup (the window is staged once per row block)
  const int tiles_m = (a.M + 31) / 32;
  int nrt = (tiles_m + 3) / 4;
  if (nrt > 6) nrt = 6;               // (fewer row tiles per wave = more row blocks per column tile: measured slower at B = 64)
  static const int nrt_cap = [] { const char* e = getenv("MBV_NARROW_NRT_CAP"); return e ? atoi(e) : 6; }();
  if (nrt > nrt_cap) nrt = nrt_cap;
  while (by_launch_size && nrt > 1 && (long)gm.n_ctiles * ((tiles_m + 4 * nrt - 1) / (4 * nrt)) < 384) --nrt;
  // six row tiles (192 channels: the text encoder's conv_o and conv_2) do not divide over four waves: K-split
  // wave pairs of three tiles each (a rule on M alone).  The pairs alternate steps, so a block has an even number.
  static const int ks_env = [] { const char* e = getenv("MBV_NARROW_KS"); return e ? atoi(e) : 1; }();
  const bool ks = ks_env && !by_launch_size && tiles_m == 6 && a.M == 192;
  if (ks) {
    int cbk = cb;                                                    // largest block that divides Cin with an even step count
    while (cbk >= 8 && (a.Cin % cbk || ((cbk / 8) * a.K) % 2)) cbk -= 8;
    if (cbk >= 8) {
      gm.CB = cbk;
      gm.n_rblk = 1;
      gm.pf_period = 8;
      launch_narrow_epi<3, true>(a, gm, s);
      return;
    }
  }
  gm.n_rblk = (tiles_m + 4 * nrt - 1) / (4 * nrt);
  gm.pf_period = 8;
  while (gm.pf_period % gm.n_rblk) gm.pf_period += 8;               // lcm(8, n_rblk)
  switch (nrt) {
    case 1: launch_narrow_epi<1>(a, gm, s); break;
    case 2: launch_narrow_epi<2>(a, gm, s); break;
    case 3: launch_narrow_epi<3>(a, gm, s); break;
    case 4: launch_narrow_epi<4>(a, gm, s); break;
    case 5: launch_narrow_epi<5>(a, gm, s); break;
    default: launch_narrow_epi<6>(a, gm, s); break;
  }
}

}  // namespace mbv
