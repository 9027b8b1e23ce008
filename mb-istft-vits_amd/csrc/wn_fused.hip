// One WN layer (modules.py:148-176 + commons.py:100-107) in ONE launch:
//
//   x_in = conv_k5(h) + b (+ g_l)            in_layers[l]        192 -> 384, weight-normed
//   acts = tanh(x_in[:H]) * sigmoid(x_in[H:])                    fused_add_tanh_sigmoid_multiply
//   rs   = W_rs . acts + b_rs                res_skip_layers[l]  192 -> 384 (last layer: -> 192)
//   h    = (h + rs[:H]) * mask ; skip += rs[H:]                  (last layer: skip += rs)
//
// Work unit = one 32-column MFMA tile = TWO half-units of 16 consecutive frames, each of one
// utterance; only half-units that hold valid frames (t0 < len[b]) exist — everything the flows / the
// posterior encoder compute is masked, and every reader of h / skip masks on load, so padded frames
// are neither computed nor written.  Half-units are numbered through the batch (prefix sums of
// ceil(len / 16), built once per WN stack by wn_units_kernel) and tile u takes numbers 2u and 2u + 1:
// usually the two halves of 32 consecutive frames, at the end of an utterance its last 16 frames and
// the first 16 of the next one (a lane's operand addresses are its own anyway).  The batch then
// costs sum_b ceil(len_b / 16) / 2 tiles instead of sum_b ceil(len_b / 32).
//
// A 256-thread workgroup owns one unit at a time; its 4 waves (one per SIMD) split the ROWS:
//   gate GEMM   [2H x 5H] . [5H x 32]   row tile t (32 packed rows = 16 tanh + 16 sigmoid rows of
//                                       channels 16t..16t+15) -> wave t % 4
//   gating      in registers: the packing puts the tanh and the sigmoid row of a channel into the
//               same lane (registers r and r + 4 of the 32x32 accumulator)
//   rs GEMM     [Mr x H] . [H x 32]     the gated tile goes through LDS once, written as the
//               k-interleaved B image the MFMA loop reads (the k order of that product is
//               whatever the accumulator layout yields; W_rs is packed to match on the host)
// Weights are NOT staged in LDS: every wave needs different rows, so a lane's A operand (four
// K-steps = one 16-byte load, 512 contiguous bytes per half wave in the packed layout) comes
// straight from L2 through a register ring filled five (gate) / four (rs) steps ahead.  The only
// LDS traffic is the input window (all H channels x 36 frames, staged once per unit) and the gated
// tile; two barriers per unit, none inside the MFMA loops.  Two workgroups per CU cover each
// other's prologue / gating / epilogue.
#include "kernels.h"
#include <cstdio>
#include <cstdlib>

namespace mbv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kK = 5;            // WN kernel size of the flows and of enc_q (models.py:646-647)
constexpr int kUnit = 32;        // columns of a tile
constexpr int kHalf = 16;        // frames per half-unit
constexpr int kXW = kHalf + kK - 1;   // input window of a half-unit (20 frames)
constexpr int kXS = 32;               // its column stride in the LDS image: the second half's lanes then read
                                      // 16 slots of 16 B behind the first half's, i.e. the same banks a
                                      // contiguous wave would (a stride of 20 is a 2-way ds_read_b128 conflict)
constexpr int kXL = 2 * kXS;          // columns per (group, parity) row of the LDS image
// tanh(x) * sigmoid(y) on the hardware transcendentals (v_exp_f32 / v_rcp_f32, 1 ulp each):
//   sigmoid(y) = 1 / (1 + 2^(-y log2 e)),  tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e))
// absolute error ~1e-7 (the libm forms cost ~100 instructions per gated value; 24 values per lane and tile)
__device__ __forceinline__ float gate_fast(float x, float y) {
  const float e2x = __builtin_amdgcn_exp2f(x * 2.88539008177792681f);
  const float emy = __builtin_amdgcn_exp2f(y * -1.44269504088896341f);
  const float th = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e2x);
  return th * __builtin_amdgcn_rcpf(1.f + emy);
}
__device__ __forceinline__ float sigmoid_(float v) { return 1.f / (1.f + expf(-v)); }
}  // namespace

// ustart[b] = sum_{b' < b} ceil(len[b'] / 16), ustart[B] = number of half-units;
// hmap[hu] = utterance of half-unit hu (so that a tile finds its two utterances with one dependent
// load each instead of a binary search of ~log2 B dependent loads: 12 L2 round trips per tile at B = 64)
__global__ void wn_units_kernel(const int* lens, int B, int T, int* ustart, int* hmap) {
  __shared__ int part[256];
  const int tid = threadIdx.x;
  const int per = (B + 255) / 256;
  int s = 0;
  for (int i = 0; i < per; ++i) {
    const int b = tid * per + i;
    if (b < B) { int l = lens[b]; l = l < 0 ? 0 : (l > T ? T : l); s += (l + kHalf - 1) / kHalf; }
  }
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = run; run += v; }
  }
  __syncthreads();
  int run = part[tid];
  for (int i = 0; i < per; ++i) {
    const int b = tid * per + i;
    if (b < B) {
      ustart[b] = run;
      int l = lens[b]; l = l < 0 ? 0 : (l > T ? T : l);
      const int n = (l + kHalf - 1) / kHalf;
      for (int k = 0; k < n; ++k) hmap[run + k] = b;
      run += n;
      if (b == B - 1) { ustart[B] = run; hmap[run] = b; }     // one entry past the end: the odd tile's second half
    }
  }
}

void launch_wn_units(const int* lens, int B, int T, int* ustart, int* hmap, hipStream_t s) {
  hipLaunchKernelGGL(wn_units_kernel, dim3(1), dim3(256), 0, s, lens, B, T, ustart, hmap);
}

size_t wn_units_ints(int B, int T) { return (size_t)B + 1 + (size_t)B * ((T + kHalf - 1) / kHalf) + 1; }

// raw buffer access: one 32-bit lane offset per tensor, every row / step stride rides in a scalar
// register (64-bit per-access pointers made the unrolled loads below spill); out-of-range lanes read 0
// and their stores are dropped, which doubles as the frame mask.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRsrcFlags = 0x00020000;
constexpr unsigned kOob = 0x7fffffffu;          // lane offset of a masked lane (beyond every range)
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bstore1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)voff, (int)soff, 0);
}

// ---- the per-wave pieces.  Every wave always runs its NRT tile slots, with no tests in the MFMA loops
// (`if (j < nact)` around every MFMA made hipcc emit a branch per instruction).  Slots past the real
// tile count (only the last layer's 6-tile rs GEMM at H = 192; the small configurations) cost idle
// MFMAs and nothing else: their weight / bias reads fall outside the buffer ranges or into padding
// (garbage in, never used), and their loads / stores of h / skip are skipped by a wave-uniform test.
struct WnCtx {
  __amdgpu_buffer_rsrc_t wg_rs, wr_rs, bg_rs, br_rs, gc_rs, hin_rs, hout_rs, skip_rs, x1_rs;
  unsigned wg_voff, wr_voff, wg_step, wr_step, rowT, io_voff, gc_voff;
  unsigned sk_voff, x1_voff;     // lane offsets into skip ([B, Cs, T]) and into the coupled half of z (last layer, folded post)
  float couple_sign;
  int couple;                    // last layer applies x1 += sign * (skip + rs) instead of storing skip
  int xoff;                      // lane's column in the input-window image: half * kXS + (l31 & 15)
  int G, H, Mr, wave, hl, l31, last, skip_accum, exact_gate;
  int Gi;                        // input groups of the gate conv (== G unless `pre` is folded in)
  int prefold;                   // the window holds x0' = [x0 ; mask], not h: residual rows start from the wpre K-block
  __amdgpu_buffer_rsrc_t wp_rs; unsigned wp_voff, wp_step;
};

// A ring of the gate GEMM: slot = tap; the load for step s + kDG (s = g * 5 + tap) is issued at step s
// into the slot step s + kDG - 5 vacated (already consumed), so no value is ever copied and a load has
// kDG steps (12 MFMAs each) to arrive.  (With the obvious "reload the slot just read" form hipcc sank
// every load behind the MFMAs that read the old value and then waited for it at the end of the
// iteration: the whole L2 latency exposed once per channel group.)
constexpr int kDG = 4;
template <int NRT>
__device__ __forceinline__ void gate_ring_init(f32x4 (&ra)[kK][NRT], const WnCtx& c) {
#pragma unroll
  for (int tap = 0; tap < kDG; ++tap)
#pragma unroll
    for (int j = 0; j < NRT; ++j) ra[tap][j] = bload4(c.wg_rs, c.wg_voff, (unsigned)(tap * c.Gi) * c.wg_step + j * 2048u);
}

// gate accumulators start from bias (+ speaker conditioning).  Packed row (r, hl) of tile
// `wave + 4 j`: registers 0-3 / 8-11 tanh, 4-7 / 12-15 sigmoid of channel 16 tile + (r & 3) + 8 (r >> 3) + 4 hl:
// every group of four registers is four consecutive channels = one 16-byte load
template <int NRT>
__device__ __forceinline__ void gate_acc_init(f32x16 (&acc)[NRT], const WnCtx& c, bool cond) {
  const unsigned vo = (unsigned)(4 * c.hl) * 4u;
#pragma unroll
  for (int j = 0; j < NRT; ++j) {
    const unsigned so = (unsigned)((c.wave + 4 * j) * 16) * 4u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned sr = so + (unsigned)(8 * (q >> 1)) * 4u + ((q & 1) ? (unsigned)c.H * 4u : 0u);
      f32x4 v = bload4(c.bg_rs, vo, sr);
      if (cond) v += bload4(c.gc_rs, c.gc_voff, sr);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[j][4 * q + s] = v[s];
    }
  }
}

template <int NRT>
__device__ __forceinline__ void gate_loop(f32x16 (&acc)[NRT], f32x4 (&ra)[kK][NRT], const f32x4* Xs, const WnCtx& c) {
  const f32x4* xl = Xs + c.hl * kXL + c.xoff;
  const int G = c.Gi;
  f32x4 bv = xl[0];
  for (int g = 0; g < G; ++g) {
    const int gn = g + 1 < G ? g + 1 : g;      // past the end: re-load the last group (unused)
#pragma unroll
    for (int tap = 0; tap < kK; ++tap) {
      // A operands of step + kDG into the slot that step vacated kK - kDG steps ago
      {
        constexpr int dummy = 0; (void)dummy;
        const int tl = (tap + kDG) % kK;
        const int gl = tap + kDG < kK ? g : gn;
#pragma unroll
        for (int j = 0; j < NRT; ++j) ra[tl][j] = bload4(c.wg_rs, c.wg_voff, (unsigned)(tl * G + gl) * c.wg_step + j * 2048u);
      }
      // B operand of the NEXT step, read while this step's MFMAs run
      const f32x4 bn = tap + 1 < kK ? xl[g * 2 * kXL + tap + 1] : xl[gn * 2 * kXL];
      __builtin_amdgcn_sched_barrier(0);       // keep the prefetches HERE (hipcc sinks them to their first use otherwise)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int j = 0; j < NRT; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[tap][j][s4], bv[s4], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      bv = bn;
    }
  }
}

// gating; the gated tile goes to LDS as the B image of the rs GEMM
template <int NRT>
__device__ __forceinline__ void gate_act(const f32x16 (&acc)[NRT], f32x4* As, const WnCtx& c) {
#pragma unroll
  for (int j = 0; j < NRT; ++j) {
    const int tile = c.wave + 4 * j;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 v;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        v[s] = c.exact_gate ? tanhf(acc[j][8 * q + s]) * sigmoid_(acc[j][8 * q + s + 4]) : gate_fast(acc[j][8 * q + s], acc[j][8 * q + s + 4]);
      if (2 * tile + q < c.G) As[((2 * tile + q) * 2 + c.hl) * 32 + c.l31] = v;   // channels 16 tile + 8 q + 4 hl + s
    }
  }
}

constexpr int kDR = 4;           // slots of the rs GEMM's A ring; loads run kDR - 1 steps ahead
template <int NRT>
__device__ __forceinline__ void rs_ring_init(f32x4 (&rb)[kDR][NRT], const WnCtx& c) {
#pragma unroll
  for (int d = 0; d < kDR - 1; ++d)
#pragma unroll
    for (int j = 0; j < NRT; ++j) rb[d][j] = bload4(c.wr_rs, c.wr_voff, (unsigned)d * c.wr_step + j * 2048u);
}

// rs accumulators start from bias + what the layer updates in place; rows of tile `wave + 4 j`:
// (r & 3) + 8 (r >> 2) + 4 hl.  The residual rows' start values (h of the tile's own frames) are
// already in LDS: the centre of the input window (channel ch -> image row (ch / 8) * 2 + (ch & 1),
// component (ch & 7) >> 1); masked frames hold 0 there.  skip comes from memory (masked lanes read 0).
// Two parts, nothing in either is conditional (r03, after the assembly audit of scripts/asm_serial_loads.py: with
// `if (row0 < Mr) { if (is_res) .. else if (skip_accum) v += load }` in one nest hipcc emitted branch, load,
// s_waitcnt vmcnt(0) per element — up to 20 memory round trips per tile, one after the other, in every unit):
//   rs_acc_request: the skip start values, loaded straight into the accumulators (no temporaries: the kernel has no
//     registers to spare here) — a tile that takes none loads through an out-of-range lane offset and gets zeros;
//     issued BEFORE the gating arithmetic, which hides the one batch;
//   rs_acc_finish (after the gating, when its accumulators are dead): + bias (from LDS, staged once per workgroup)
//     + the residual rows' own input (the window centre in LDS; times 0 for tiles that take none).
template <int NRT>
__device__ __forceinline__ void rs_acc_request(f32x16 (&acr)[NRT], const WnCtx& c) {
#pragma unroll
  for (int j = 0; j < NRT; ++j) {
    const int row0 = (c.wave + 4 * j) * 32;
    const bool is_res = !c.last && row0 < c.H;                  // wave-uniform (H % 32 == 0)
    const unsigned srow0 = (unsigned)(c.last ? row0 : row0 - c.H);
    const unsigned sko = (row0 < c.Mr && !is_res && c.skip_accum) ? c.sk_voff : kOob;   // (row0 >= Mr: an idle tile slot, see rs_store)
#pragma unroll
    for (int r = 0; r < 16; ++r) acr[j][r] = bload1(c.skip_rs, sko, (srow0 + 8 * (r >> 2) + (r & 3)) * c.rowT);
  }
}

template <int NRT>
__device__ __forceinline__ void rs_acc_finish(f32x16 (&acr)[NRT], const f32x4* Xs, const float* Bs, const WnCtx& c) {
  const float* xc = reinterpret_cast<const float*>(Xs + c.xoff + (kK - 1) / 2);
#pragma unroll
  for (int j = 0; j < NRT; ++j) {
    const int row0 = (c.wave + 4 * j) * 32;
    const bool use_x = row0 < c.Mr && !c.last && row0 < c.H && !c.prefold;
    const float xw = use_x ? 1.f : 0.f;
    const int rowx = use_x ? row0 : 0;                           // (keeps the LDS reads inside the window image)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(Bs + row0 + 8 * q + 4 * c.hl);   // rows past Mr: zeros
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int r = 4 * q + s;
        const int ch = rowx + 8 * q + 4 * c.hl + s;             // (ch & 7) = 4 hl + s
        const float xv = xc[(((ch >> 3) * 2 + (s & 1)) * kXL) * 4 + 2 * c.hl + (s >> 1)];
        acr[j][r] = (acr[j][r] + bq[s]) + xw * xv;
      }
    }
  }
}

template <int NRT>
__device__ __forceinline__ void rs_loop(f32x16 (&acr)[NRT], f32x4 (&rb)[kDR][NRT], const f32x4* As, const WnCtx& c) {
  const f32x4* al = As + c.hl * 32 + c.l31;
  const int G = c.G;
  f32x4 bv = al[0];
  for (int g0 = 0; g0 < G; g0 += kDR) {
#pragma unroll
    for (int d = 0; d < kDR; ++d) {
      const int g = g0 + d;
      {
        const int gl = g + kDR - 1 < G ? g + kDR - 1 : G - 1;
#pragma unroll
        for (int j = 0; j < NRT; ++j) rb[(d + kDR - 1) % kDR][j] = bload4(c.wr_rs, c.wr_voff, (unsigned)gl * c.wr_step + j * 2048u);
      }
      const int gb = g + 1 < G ? g + 1 : g;
      const f32x4 bn = al[gb * 2 * 32];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int j = 0; j < NRT; ++j)
          acr[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(rb[d][j][s4], bv[s4], acr[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      bv = bn;
    }
  }
}

// folded `pre`: acr += W_pre' . x0'(t) — the residual rows' start values h(t) as one more K-block (Gi groups) of the
// res/skip GEMM.  B operand = the centre column of the input window (tap (K - 1) / 2 of the gate loop's reads); A =
// wpre, whose rows past H are zero padding (>= 128 NRT rows are allocated), so the skip tiles just add zeros.
template <int NRT>
__device__ __forceinline__ void pre_loop(f32x16 (&acr)[NRT], f32x4 (&rb)[kDR][NRT], const f32x4* Xs, const WnCtx& c) {
  const f32x4* xl = Xs + c.hl * kXL + c.xoff + (kK - 1) / 2;
  const int G = c.Gi;
#pragma unroll
  for (int d = 0; d < kDR - 1; ++d)
#pragma unroll
    for (int j = 0; j < NRT; ++j) rb[d][j] = bload4(c.wp_rs, c.wp_voff, (unsigned)(d < G ? d : G - 1) * c.wp_step + j * 2048u);
  f32x4 bv = xl[0];
  for (int g0 = 0; g0 < G; g0 += kDR) {
#pragma unroll
    for (int d = 0; d < kDR; ++d) {
      const int g = g0 + d;
      if (g < G) {                              // (G = 13 is not a multiple of the ring depth; wave-uniform)
        {
          const int gl = g + kDR - 1 < G ? g + kDR - 1 : G - 1;
#pragma unroll
          for (int j = 0; j < NRT; ++j) rb[(d + kDR - 1) % kDR][j] = bload4(c.wp_rs, c.wp_voff, (unsigned)gl * c.wp_step + j * 2048u);
        }
        const int gb = g + 1 < G ? g + 1 : g;
        const f32x4 bn = xl[gb * 2 * kXL];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int j = 0; j < NRT; ++j)
            acr[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(rb[d][j][s4], bv[s4], acr[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bv = bn;
      }
    }
  }
}

// h_out = h + rs[:H] (the frame is valid, mask = 1) ; skip (+)= rs[H:]; masked lanes' stores are dropped
template <int NRT>
__device__ __forceinline__ void rs_store(const f32x16 (&acr)[NRT], const WnCtx& c) {
#pragma unroll
  for (int j = 0; j < NRT; ++j) {
    const int row0 = (c.wave + 4 * j) * 32;
    // an idle tile slot (rows past Mr: the last layer's 6-tile rs GEMM, the small configurations) must
    // not store: h / skip are whole-tensor views, rows past H would land in the NEXT utterance
    if (row0 >= c.Mr) continue;
    const bool is_res = !c.last && row0 < c.H;
    const unsigned srow0 = (unsigned)(c.last ? row0 : row0 - c.H);
    if (!is_res && c.couple) {
      // folded post, last layer: acr = m (bias + the skips of the layers before + this layer's): the coupling itself.
      // All 16 loads of x1 first, then the stores (one in-order counter for both on gfx9)
      float xv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) xv[r] = bload1(c.x1_rs, c.x1_voff, (srow0 + (unsigned)((r & 3) + 8 * (r >> 2))) * c.rowT);
#pragma unroll
      for (int r = 0; r < 16; ++r)
        bstore1(xv[r] + c.couple_sign * acr[j][r], c.x1_rs, c.x1_voff, (srow0 + (unsigned)((r & 3) + 8 * (r >> 2))) * c.rowT);
      continue;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned k = (unsigned)((r & 3) + 8 * (r >> 2));
      if (is_res) bstore1(acr[j][r], c.hout_rs, c.io_voff, (unsigned)(row0 + k) * c.rowT);
      else bstore1(acr[j][r], c.skip_rs, c.sk_voff, (srow0 + k) * c.rowT);
    }
  }
}

// NRT = row tiles per wave (ceil(2H / 32 / 4)): 3 for H = 192 / 160, 2 for H <= 128
template <int NRT>
__global__ __launch_bounds__(256, 2) void wn_layer_kernel(const WnLayerArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  WnCtx c;
  c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.hl = lane >> 5; c.l31 = lane & 31;
  const int H = a.H, T = a.T;
  c.H = H; c.G = H / 8; c.Mr = a.Mr; c.last = a.last; c.skip_accum = a.skip_accum; c.exact_gate = a.debug == 2;
  const int G = c.G;                           // 8-channel groups (4 K-steps each) of the gated tile
  c.Gi = a.Gi ? a.Gi : G;
  c.prefold = a.wpre != nullptr;
  const int Gi = c.Gi;                         // ... of the input window
  const int in_cb = a.in_cb ? a.in_cb : H;
  f32x4* const Xs = reinterpret_cast<f32x4*>(lds);          // [Gi][2][kXL]
  f32x4* const As = Xs + Gi * 2 * kXL;                       // [G][2][32]
  float* const Bs = reinterpret_cast<float*>(As + G * 2 * 32);   // [128 NRT]: bias of the res / skip conv, zeros past Mr
  for (int e = tid; e < 128 * NRT; e += 256) Bs[e] = e < a.Mr && a.br ? a.br[e] : 0.f;   // (read after the unit loop's barriers)
  const int Hn = a.ustart[a.B];                // half-units of the batch
  const int U = (Hn + 1) / 2;                  // tiles

  // packed weights W[step][h][Mpad][4 floats], step = tap * G + g: lane offset + scalar step offset
  c.wg_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wg), 0, kK * 8 * Gi * a.Mg_pad * 4, kRsrcFlags);
  c.wp_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpre), 0, a.wpre ? 8 * Gi * a.wpre_Mpad * 4 : 0, kRsrcFlags);
  c.wp_voff = (unsigned)((c.hl * a.wpre_Mpad + c.wave * 32 + c.l31) * 16);
  c.wp_step = (unsigned)(2 * a.wpre_Mpad * 16);
  c.wr_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wr), 0, H * a.Mr_pad * 4, kRsrcFlags);
  c.bg_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bg), 0, 2 * H * 4, kRsrcFlags);
  c.br_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.br), 0, a.Mr * 4, kRsrcFlags);
  c.wg_voff = (unsigned)((c.hl * a.Mg_pad + c.wave * 32 + c.l31) * 16);
  c.wr_voff = (unsigned)((c.hl * a.Mr_pad + c.wave * 32 + c.l31) * 16);
  c.wg_step = a.debug == 1 ? 0u : (unsigned)(2 * a.Mg_pad * 16);
  c.wr_step = a.debug == 1 ? 0u : (unsigned)(2 * a.Mr_pad * 16);
  c.rowT = (unsigned)T * 4u;                   // bytes between channel rows of h / skip
  // whole-tensor views (the launcher checks B H T 4 < 2^32): the two halves of a tile may belong to
  // different utterances, so the utterance goes into the lane offset
  const unsigned all_bytes = (unsigned)a.B * (unsigned)H * c.rowT;
  c.hin_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.h_in), 0, (unsigned)a.B * (unsigned)in_cb * c.rowT, kRsrcFlags);
  c.hout_rs = __builtin_amdgcn_make_buffer_rsrc(a.h_out, 0, a.last ? 0 : all_bytes, kRsrcFlags);
  const int Cs = a.Cs ? a.Cs : H;
  c.skip_rs = __builtin_amdgcn_make_buffer_rsrc(a.skip, 0, (unsigned)a.B * (unsigned)Cs * c.rowT, kRsrcFlags);
  c.couple = a.last && a.x1 != nullptr;
  c.couple_sign = a.couple_sign;
  c.x1_rs = __builtin_amdgcn_make_buffer_rsrc(a.x1, 0, c.couple ? (unsigned)(a.B * a.x1_bstride * 4) : 0, kRsrcFlags);
  c.gc_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gcond), 0, a.gcond ? a.B * a.gcond_bstride * 4 : 0, kRsrcFlags);
  const int half = c.l31 >> 4, jl = c.l31 & 15;
  c.xoff = half * kXS + jl;

  for (int u = blockIdx.x; u < U; u += gridDim.x) {
    // the two half-units of this tile: utterance, first frame, utterance length (0: no such half-unit)
    int hb[2], ht0[2], hlen[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int hu = 2 * u + k;
      const int lo = a.hmap[hu];               // (hu <= Hn: the map has one entry past the end)
      hb[k] = lo;
      ht0[k] = (hu - a.ustart[lo]) * kHalf;
      int len = a.lens[lo];
      len = len > T ? T : len;
      hlen[k] = hu < Hn ? len : 0;
    }
    {
      const int b = half ? hb[1] : hb[0];
      const int t = (half ? ht0[1] : ht0[0]) + jl;          // this lane's frame
      const bool tv = t < (half ? hlen[1] : hlen[0]);
      c.io_voff = tv ? (unsigned)((b * H + 4 * c.hl) * T + t) * 4u : kOob;
      c.sk_voff = tv ? (unsigned)((b * Cs + 4 * c.hl) * T + t) * 4u : kOob;
      c.x1_voff = tv ? (unsigned)(b * a.x1_bstride + (int64_t)(4 * c.hl) * T + t) * 4u : kOob;
      c.gc_voff = (unsigned)(b * a.gcond_bstride + 4 * c.hl) * 4u;
    }

    f32x4 ra[kK][NRT];
    gate_ring_init<NRT>(ra, c);      // the first kDG steps of the A ring

    // ---- input windows: all H channels x 2 x 20 frames, masked, k-interleaved.  Every load of the
    // tile's prologue (ring, windows, bias) is issued before the first wait: one memory latency
    // instead of one per staging pass.
    constexpr int NXI = NRT == 3 ? 12 : 8;     // window items per thread (G * 2 * kXL / 256, G <= 8 NRT)
    f32x4 xw[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int e = tid + 256 * i;
      const int P = e / kXL, cc = e - P * kXL;
      const int k = cc >= kXS, c2 = cc - k * kXS;
      const int ti = (k ? ht0[1] : ht0[0]) - (kK - 1) / 2 + c2;
      const int b = k ? hb[1] : hb[0];
      const bool ok = P < 2 * Gi && c2 < kXW && ti >= 0 && ti < (k ? hlen[1] : hlen[0]);
      const bool mrow = c.prefold && (P >> 1) == Gi - 1;                 // the mask group of x0' (no memory behind it)
      const unsigned vo = ok && !mrow ? (unsigned)((b * in_cb + (P >> 1) * 8 + (P & 1)) * T + ti) * 4u : kOob;
      xw[i][0] = bload1(c.hin_rs, vo, 0); xw[i][1] = bload1(c.hin_rs, vo, 2 * c.rowT);
      xw[i][2] = bload1(c.hin_rs, vo, 4 * c.rowT); xw[i][3] = bload1(c.hin_rs, vo, 6 * c.rowT);
      if (mrow && ok && (P & 1) == 0) xw[i][0] = 1.f;                    // channel 8 (Gi - 1) = the frame mask
    }
    f32x16 acc[NRT];
    gate_acc_init<NRT>(acc, c, a.gcond != nullptr);
    __syncthreads();                           // the previous tile's readers of Xs / As are done
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int e = tid + 256 * i;
      if (e < Gi * 2 * kXL) Xs[e] = xw[i];
    }
    __syncthreads();

    gate_loop<NRT>(acc, ra, Xs, c);

    f32x4 rb[kDR][NRT];
    rs_ring_init<NRT>(rb, c);        // both issued before the gating arithmetic, which hides their latency
    f32x16 acr[NRT];
    rs_acc_request<NRT>(acr, c);
    gate_act<NRT>(acc, As, c);
    rs_acc_finish<NRT>(acr, Xs, Bs, c);
    __syncthreads();

    rs_loop<NRT>(acr, rb, As, c);
    if (c.prefold) pre_loop<NRT>(acr, rb, Xs, c);
    rs_store<NRT>(acr, c);
  }
}

// (the kernel addresses h / skip through whole-tensor buffer views: B * H * T * 4 must stay below 2^32)
bool wn_fused_fits(int B, int H, int T) { return (unsigned long long)B * H * T * 4ull < (1ull << 32); }

bool wn_fused_supported(int H, int K) {
  return K == kK && H % 32 == 0 && H >= 32 && H <= 192;      // <= 3 row tiles per wave (4 spill)
}

void launch_wn_layer(const WnLayerArgs& a, hipStream_t s) {
  const int G = a.H / 8, Gi = a.Gi ? a.Gi : G;
  const int nrt = (2 * a.H / 32 + 3) / 4;
  const size_t lds_bytes = (size_t)(Gi * 2 * kXL + G * 2 * 32) * 16 + (size_t)128 * (nrt <= 2 ? 2 : 3) * 4;   // + the res / skip bias
  WnLayerArgs a2 = a;
  static const int dbg = [] { const char* e = getenv("MBV_WN_DEBUG_A"); return e ? atoi(e) : 0; }();
  a2.debug = dbg;                    // experiments: 1 = every step reads the weights of step 0 (timing only, results wrong);
                                     // 2 = libm tanhf / expf in the gating instead of the hardware transcendentals
  // at most two resident workgroups per CU (register / LDS budget); a workgroup walks units
  // blockIdx.x, blockIdx.x + grid, ...; the unit count itself lives on the device
  long max_units = ((long)a.B * ((a.T + kHalf - 1) / kHalf) + 1) / 2;
  static const int grid_cap = [] { const char* e = getenv("MBV_WN_GRID"); return e ? atoi(e) : 512; }();
  const int grid = (int)(max_units < grid_cap ? (max_units < 1 ? 1 : max_units) : grid_cap);
#define MBV_WN_LAUNCH(N)                                                                          \
  {                                                                                               \
    static bool attr = false;                                                                     \
    if (!attr) {                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wn_layer_kernel<N>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);          \
      attr = true;                                                                                \
    }                                                                                             \
    hipLaunchKernelGGL((wn_layer_kernel<N>), dim3(grid), dim3(256), lds_bytes, s, a2);            \
  }
  if (nrt <= 2) MBV_WN_LAUNCH(2)
  else MBV_WN_LAUNCH(3)
#undef MBV_WN_LAUNCH
}

}  // namespace mbv
