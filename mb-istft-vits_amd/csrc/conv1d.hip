// conv1d as an implicit GEMM on the gfx950 matrix cores, exact fp32
// (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain, same peak as the
// fp32 VALU, but one VGPR per operand and the VALU left free for the fused
// prologue / epilogue).
//
//   y[b, m, t] = bias[m] + sum_{ci, tap} W[m][ci][tap] * act(x[b, ci, t + tap*dil - pad_left])
//
// covers every dense contraction of the path (reference call sites):
//   attentions.py:139-146 (q/k/v/o 1x1), attentions.py:278-285 (FFN k3),
//   models.py:128-136 (duration predictor), models.py:178 (enc_p.proj),
//   modules.py:151-169 (WN in_layers k5 + gate, res_skip 1x1),
//   modules.py:336-350 (coupling pre/post), models.py:348 (conv_pre k7),
//   modules.py:216-226 (ResBlock1 convs, leaky-relu fused on the input,
//   residual add fused on the output), models.py:363-365 (lrelu 0.01 +
//   ReflectionPad1d((1,0)) + subband_conv_post k7), models.py:321-323 (stride-4
//   ConvTranspose1d as a 5-tap conv over its output phases, EPI_CONVT),
//   models.py:222-231 / modules.py:98-111 (posterior encoder, SDP 1x1 convs).
//
// Tiling (wave64): block = 2(M) x NWN(N) waves, NWN = 2 (256 threads) or 4 (512 threads,
// double-buffered LDS); each wave owns WM x WN MFMA
// tiles of 32x32 (rows = output channels, columns = time).  K loop: Cin in
// chunks of CK channels (CK/8 groups of 8); per chunk the activated input window
// and the weight slab are staged in LDS once and reused by all taps.
//
// k-interleaved LDS images ("read wide"): a lane's MFMA operand for K-step s of a
// group is channel 2s + (lane>>5); both images keep the four K-steps of a group
// adjacent,
//     Ws[tap][group][h][m][4]        Xs[group][h][col][4]        (h = channel & 1)
// so ONE ds_read_b128 per operand tile feeds four MFMAs per accumulator (16 / 32
// MFMAs per 4 / 6 LDS reads instead of one read per MFMA), conflict-free (lanes
// are 16 B apart).  The weights are packed in exactly this order on the host
// (capi.hip pack_conv), so the slab is a linear copy.
//
// Staging is an async split (global -> registers -> LDS): all loads of chunk c+1
// are issued before the MFMA loop of chunk c and committed to LDS after it; the
// operand reads inside the loop are double-buffered in registers one group ahead.
// Work distribution: persistent workgroups walk (tile[, K split]) units as one staging stream;
// epilogue kind is a template parameter; accumulators start from whatever the epilogue would
// have to read (see conv_acc_init and the notes at the kernel).
#include "kernels.h"
#include <cstdio>
#include <cstdlib>

#ifndef MBV_CONV_GLDS
#define MBV_CONV_GLDS 8      // widest channel chunk CK whose 512-thread kernels take their weights by LDS-DMA (0: none; A/B builds)
#endif
namespace mbv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// In-kernel timeline (scripts/conv_stamps.hip builds this file with -DMBV_CONV_STAMPS): threads 0 and 256 of
// every workgroup (wave 0 and its SIMD partner wave 4 in the 512-thread shape) append the 100 MHz wall clock at
// the phase boundaries named at the call sites; 512 slots per sampled thread, slot 0 = count.  Kept in LDS until
// the workgroup ends: a global store per stamp would sit in front of the kernel's own s_waitcnt vmcnt(0)s and
// stretch exactly the phases being measured.  Without the macro nothing of this exists in the kernels.
#ifdef MBV_CONV_STAMPS
#define MBV_CSTAMP(ID)                                                                        \
  if ((tid & 255) == 0 && stamp_n < 511) stamp_lds[tid >> 8][++stamp_n] = (__builtin_amdgcn_s_memrealtime() << 4) | (unsigned)(ID);
#define MBV_CSTAMP_FLUSH()                                                                    \
  {                                                                                           \
    if ((tid & 255) == 0) stamp_lds[tid >> 8][0] = stamp_n;                                   \
    __syncthreads();                                                                          \
    unsigned long long* sb_ = reinterpret_cast<unsigned long long*>(a.ws) + (size_t)blockIdx.x * 1024; \
    for (int e = tid; e < 1024; e += NT) sb_[e] = stamp_lds[e >> 9][e & 511];                 \
  }
#else
#define MBV_CSTAMP(ID)
#define MBV_CSTAMP_FLUSH()
#endif

// Opt-in split-bf16 arithmetic (ConvArgs::prec == 3, mbv_set_option "conv_bf16"): x = hi + mid + O(2^-17 x)
// with hi = bf16(x), mid = bf16(x - hi); a product keeps hi*hi + hi*mid + mid*hi (relative error ~2^-16),
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16 — 3 MFMAs of 32 cycles per 16 K-values instead of 8 of
// 64.  The LDS images keep their shape: a 16-byte slot of four K-values holds [hi x 4 | mid x 4] instead of
// four floats.  The weights arrive that way (ConvArgs::w_split, split once per checkpoint by
// launch_split_planes), the input window is split when it is committed to LDS (once per element — it is
// read 2 K times), and an MFMA operand (eight K-values of a lane) is the matching halves of two slots:
// no arithmetic between the LDS read and the MFMA.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 split_slot(const f32x4& v) {
  bf16x4 hi, mid;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const __bf16 h = (__bf16)v[k];
    hi[k] = h;
    mid[k] = (__bf16)(v[k] - (float)h);
  }
  struct { bf16x4 h, m; } o{hi, mid};
  return __builtin_bit_cast(f32x4, o);
}
// the hi (PLANE = 0) or mid (PLANE = 1) halves of the slots of two consecutive steps as one MFMA operand
template <int PLANE>
__device__ __forceinline__ bf16x8 plane_of(const f32x4& s0, const f32x4& s1) {
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  const f32x4_ o = {s0[2 * PLANE], s0[2 * PLANE + 1], s1[2 * PLANE], s1[2 * PLANE + 1]};
  return __builtin_bit_cast(bf16x8, o);
}

// Start values of a wave's accumulators: everything the epilogue would otherwise have to READ
// after the MFMA loop (residual, running ResBlock sum, the tensor a flow layer updates in place);
// zero for the epilogues that read nothing.  `full`: the wave's patch lies inside [M, T].
template <int WM, int WN, int EPI>
__device__ __forceinline__ void conv_acc_init(f32x16 (&acc)[WM][WN], const ConvArgs& a, int b, int wrow0,
                                              int t0, int wn, int hl, int l31, bool full,
                                              int64_t lane_off) {
  constexpr bool kInit = EPI == EPI_RESID || EPI == EPI_RESID_ACC || EPI == EPI_RES_SKIP || EPI == EPI_COUPLE;
    if ((EPI == EPI_RESID || EPI == EPI_RESID_ACC) && wrow0 + 32 * WM <= a.M) {
      // Every row real (any patch of the decoder's convs).  Loads through a buffer view of the utterance: one 32-bit
      // lane offset per column tile — out of range for a lane whose column lies past the end of the sequence, which
      // then reads 0 — plus a scalar row offset: no 64-bit address per element, and the edge patches (the last column
      // tile of an utterance) take the same batched loads as interior ones.  In the per-element form at the bottom
      // (`if (row < M && t < T) { v0 = res[..]; if (accum_in) v0 += accum_in[..]; }`) hipcc emits branch, load,
      // s_waitcnt vmcnt(0) per element for the running-sum variant: 96 memory round trips one after the other in a wave
      // of every edge tile, and the workgroups holding one end the launch that much later (r03 audit,
      // scripts/asm_serial_loads.py).
      const unsigned nrec = (unsigned)a.M * (unsigned)a.T * 4u;   // < 4 GiB: checked by launch_conv1d
      const __amdgpu_buffer_rsrc_t rr =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res + (int64_t)b * a.res_bstride), 0, nrec, 0x00020000);
      unsigned vo[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int t = t0 + wn * 32 * WN + j * 32 + l31;
        vo[j] = t < a.T ? (unsigned)((wrow0 + 4 * hl) * a.T + t) * 4u : 0x7fffffffu;
      }
      const unsigned rowT = (unsigned)a.T * 4u;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned so = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * rowT;
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)vo[j], (int)so, 0));
        }
      if (EPI == EPI_RESID_ACC && a.accum_in) {
        // second operand through temporaries, half a patch at a time: two waits instead of one
        // per element (a load feeding an add right away serialises the whole initialisation)
        const __amdgpu_buffer_rsrc_t ra =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.accum_in + (int64_t)b * a.y_bstride), 0, nrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          float tmp[16][WN];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned so = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * rowT;
#pragma unroll
            for (int j = 0; j < WN; ++j)
              tmp[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (int)vo[j], (int)so, 0));
          }
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j][r] += tmp[r][j];
        }
      }
    } else if (EPI == EPI_RES_SKIP && full && a.split % 32 == 0) {
      // a 32-row tile lies wholly on one side of the res / skip split (the split is a channel count)
      const int64_t col_off = (int64_t)(4 * hl) * a.T + t0 + wn * 32 * WN + l31;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int trow = wrow0 + i * 32;                               // wave-uniform
        const bool is_x = trow < a.split;
        const float* base = is_x ? a.y + (int64_t)b * a.y_bstride + (int64_t)trow * a.T
                                 : a.skip + ((int64_t)b * (a.M - a.split) + (trow - a.split)) * a.T;
        if (is_x || a.skip_accum) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float* q = base + col_off + (int64_t)((r & 3) + 8 * (r >> 2)) * a.T;
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j][r] = q[j * 32];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j][r] = 0.f;
        }
      }
    } else if (EPI == EPI_COUPLE && full) {
      const float* py = a.y + (int64_t)b * a.y_bstride + lane_off;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* q = py + (int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * a.T;
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j][r] = a.couple_sign * q[j * 32];
        }
    } else {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          float v0 = 0.f;
          if constexpr (kInit) {
            const int t = t0 + wn * 32 * WN + j * 32 + l31;
            if (row < a.M && t < a.T) {
              if constexpr (EPI == EPI_RESID) {
                v0 = a.res[(int64_t)b * a.res_bstride + (int64_t)row * a.T + t];
              } else if constexpr (EPI == EPI_RESID_ACC) {
                v0 = a.res[(int64_t)b * a.res_bstride + (int64_t)row * a.T + t];
                if (a.accum_in) v0 += a.accum_in[(int64_t)b * a.y_bstride + (int64_t)row * a.T + t];
              } else if constexpr (EPI == EPI_RES_SKIP) {
                if (row < a.split) v0 = a.y[(int64_t)b * a.y_bstride + (int64_t)row * a.T + t];
                else if (a.skip_accum) v0 = a.skip[((int64_t)b * (a.M - a.split) + (row - a.split)) * a.T + t];
              } else {   // EPI_COUPLE: y' = mask * sign * (sign * y + conv + bias)
                v0 = a.couple_sign * a.y[(int64_t)b * a.y_bstride + (int64_t)row * a.T + t];
              }
            }
          }
          acc[i][j][r] = v0;
        }
      }
    }
}

// NWN = waves along time.  NWN == 2: 256 threads, single LDS buffer, 2 barriers per chunk,
// two workgroups per CU (short sequences / small launches).  NWN == 4: 512 threads = one
// workgroup per CU with 2 waves per SIMD, the weight slab shared by twice as many columns,
// DOUBLE-buffered LDS: chunk c+1 is committed into the other buffer right after the MFMA loop of
// chunk c, one barrier per chunk, waves drift apart instead of staging in lockstep.
//
// Persistent tiles: the grid is at most the number of workgroups resident on the chip; each
// workgroup walks output tiles blockIdx.x, blockIdx.x + gridDim.x, ... and treats the chunks of all
// its tiles as ONE staging stream: chunk 0 of the next tile is loaded and committed to LDS while the
// last chunk of the current tile is in the MFMA loop.  The epilogue (residual reads + output stores;
// with one workgroup per CU it used to run with the matrix pipe idle: measured 25 % of a k=3 conv)
// is then followed immediately by MFMA work on data already in LDS, so its stores drain underneath.
// No staging registers are live across the epilogue.
// NWM (r03) = waves along the rows: 2 everywhere except the 64 x 384 shape (NWM = 1, NWN = 4: 256 threads, the same
// 2 x 3 MFMA tiles per wave and double-buffered staging as the 128 x 384 shape, two workgroups per CU) that serves
// convs with <= 64 output rows — the mini configurations' 64-channel stage — instead of the 64 x 128 shape's two
// accumulators per wave.
template <int WM, int WN, int CK, int NWN, int EPI, int PREC = 0, int NWM = 2, int VS = 0>
__global__ __launch_bounds__(64 * NWM * NWN, 2) void conv1d_mfma_kernel(const ConvArgs a, int tiles_x,
                                                                        int tiles_y, int total_tiles,
                                                                        int ksplit) {
  constexpr int NT = 64 * NWM * NWN;     // threads
  constexpr bool DB = NWN == 4;          // double-buffered LDS
  constexpr int BM = 32 * WM * NWM;      // NWM waves x WM tiles of 32 rows
  constexpr int BN = 32 * WN * NWN;      // NWN waves x WN tiles of 32 columns
  constexpr int G = CK / 8;              // channel groups (4 K-steps each) per chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef MBV_CONV_STAMPS
  int stamp_n = 0;
  __shared__ unsigned long long stamp_lds[2][512];
#endif
  // 256-thread shapes: wave index made provably uniform, so the nact / nj tests become scalar
  // branches (in the 512-thread shape the extra SGPR pressure makes the kernel spill instead)
  const int wave = NWN == 2 ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int hl = lane >> 5, l31 = lane & 31;

  const int halo = (a.K - 1) * a.dil;
  const int XL = BN + halo;                            // staged columns
  // one LDS buffer: input image [G][2][XL] + weight slab [K][G][2][BM] (+ 2*BM prefetch overrun)
  const int buf_f4 = G * 2 * XL + a.K * G * 2 * BM + 2 * BM;
  f32x4* const lds4 = reinterpret_cast<f32x4*>(lds);
  const int tin_eff = a.reflect1 ? a.Tin + 1 : a.Tin;  // length of the (virtually padded) input

  // ---- async-stage bookkeeping ------------------------------------------------
  constexpr int HALO_MAX = CK == 8 ? 72 : (CK == 16 ? 24 : 0);
  constexpr int KMAX = CK == 8 ? 11 : (CK == 16 ? 5 : 1);
  constexpr int NXP = (G * 2 * (BN + HALO_MAX) + NT - 1) / NT;   // (group, h, col) items per thread
  constexpr int NW = (KMAX * G * 2 * BM + NT - 1) / NT;        // weight float4 per thread
  constexpr int RPI = NT / BM;                                 // weight rows [BM x f32x4] per pass
  constexpr int ROWS_PER_TAP = 2 * G;
  static_assert(RPI % ROWS_PER_TAP == 0 || ROWS_PER_TAP % RPI == 0, "weight-row decomposition");
  // 512-thread shape: the weight slab goes global -> LDS by LDS-DMA (no registers: the kernel sits at the
  // 256-register cap), see MBV_GLDS_W; the 256-thread shapes stage it through wreg
  // Only the CK = 8 kernels (k = 7 / 11): their MFMA loop (9 - 14 us) covers the DMA issued at its top.
  // With CK = 16 (k = 3, 4 us per chunk) the same change measured +5 % (489 -> 516 us): register staging
  // issues a chunk's weights one barrier EARLIER and so has two loops to land them.
  constexpr bool GLDS = DB && (CK <= MBV_CONV_GLDS || NWM == 1);   // (NWM = 1: twice the window items per thread, no room for wreg)
  f32x4 wreg[GLDS ? 1 : NW];
  f32x4 xreg[NXP];
  int xoff[NXP];
  const int totalW = a.K * ROWS_PER_TAP * BM;                  // float4 in the slab
  const int wq = tid % BM, wR0 = tid / BM;
  const int wtap0 = RPI >= ROWS_PER_TAP ? wR0 / ROWS_PER_TAP : 0;
  const int wrem0 = RPI >= ROWS_PER_TAP ? wR0 % ROWS_PER_TAP : wR0;
  const int64_t tap_stride = (int64_t)(a.Cin / 8) * 2 * a.Mpad * 4;       // floats per tap
  const int nchunks = (a.debug == 1 || a.debug == 4) ? 1 : a.Cin / CK;   // debug: stage chunk 0 only (timing)

  // per-tile load descriptors (of the tile whose chunks are being staged)
  int lb = 0, lm0 = 0, lt0 = 0;
  const float* wlane = nullptr;          // weight slab: lane base, pass u adds a wave-uniform offset
  const float* xb = nullptr;

  // Weight slab: global order == LDS order ([tap][Cin/8][2][Mpad][4] vs [tap][G][2][BM][4]);
  // thread copies float4 (row R_0 + u*RPI, column wq).
  // Input window: item e = tid + NT u -> (P = group*2 + h, col); its four K-steps are the
  // channels 8 g + h + 2 s.  xoff = offset of s = 0 inside a chunk, -1 = zero padding / masked.
#define MBV_SETUP_TILE(TILE)                                                                 \
  {                                                                                          \
    const int tile_ = (TILE);                                                                \
    if (a.trim_map) {                 /* compact list: (column tile of the map, row tile) */    \
      const int ux_ = tile_ / tiles_y;                                                       \
      lb = a.trim_map[a.B + 1 + ux_]; lm0 = (tile_ % tiles_y) * BM; lt0 = (ux_ - a.trim_map[lb]) * BN; \
    } else {                                                                                 \
      const int tx_ = tile_ % tiles_x, rest_ = tile_ / tiles_x;                              \
      lb = rest_ / tiles_y; lm0 = (rest_ % tiles_y) * BM; lt0 = tx_ * BN;                    \
    }                                                                                        \
    wlane = (PREC == 3 ? a.w_split : a.w) + (int64_t)wtap0 * tap_stride + ((int64_t)wrem0 * a.Mpad + lm0 + wq) * 4; \
    xb = a.x + (int64_t)lb * a.x_bstride;                                                    \
    const int len_in_ = a.in_lens ? a.in_lens[lb] : 0x7fffffff;                              \
    int P = tid / XL, col = tid - P * XL;                                                    \
    _Pragma("unroll") for (int u = 0; u < NXP; ++u) {                                        \
      int off = -1;                                                                          \
      if (P < 2 * G) {                                                                       \
        int gi = lt0 - a.pad_left + col;                                                     \
        if constexpr (VS) {     /* virtual column -> (utterance, frame); the gaps read as padding */ \
          if (gi >= 0) {                                                                     \
            const int bq_ = gi / a.vs_tv, tq_ = gi - bq_ * a.vs_tv;                          \
            if (bq_ < a.B && tq_ < a.Tin && (!a.in_lens || tq_ < a.in_lens[bq_]))            \
              off = bq_ * (int)a.x_bstride + ((P >> 1) * 8 + (P & 1)) * a.x_rstride + tq_;   \
          }                                                                                  \
        } else                                                                               \
        if (gi >= 0 && gi < tin_eff) {                                                       \
          if (a.reflect1) gi = gi == 0 ? 1 : gi - 1;                                         \
          if (gi < len_in_) off = ((P >> 1) * 8 + (P & 1)) * a.x_rstride + gi;               \
        }                                                                                    \
      }                                                                                      \
      xoff[u] = off;                                                                         \
      col += NT;                                                                             \
      _Pragma("unroll") for (int w = 0; w < NT / 128; ++w)                                   \
        if (col >= XL) { col -= XL; ++P; }                                                   \
    }                                                                                        \
  }
#define MBV_ISSUE(CN)                                                                        \
  {                                                                                          \
    const int cn_ = (CN);                                                                    \
    const float* wchunk = wlane + (int64_t)(cn_ / 8) * 2 * a.Mpad * 4;                       \
    if constexpr (!GLDS) {                                                                   \
    if (a.debug != 5 || cn_ == 0)     /* (ablation 5: weights staged for a tile's first chunk only; timing) */ \
    _Pragma("unroll") for (int u = 0; u < NW; ++u) {                                         \
      int64_t off;                                                                           \
      if constexpr (RPI >= ROWS_PER_TAP) {                                                   \
        int tap = wtap0 + u * (RPI / ROWS_PER_TAP); /* per-lane, clamped into the slab */    \
        tap = tap < a.K ? tap : a.K - 1;                                                     \
        off = (int64_t)(tap - wtap0) * tap_stride;                                           \
      } else {                                                                               \
        constexpr int PPT = ROWS_PER_TAP / RPI;      /* passes per tap */                    \
        int tap = u / PPT;                           /* wave-uniform */                      \
        tap = tap < a.K ? tap : a.K - 1;                                                     \
        off = tap * tap_stride + (int64_t)((u % PPT) * RPI) * a.Mpad * 4;                    \
      }                                                                                      \
      wreg[u] = *reinterpret_cast<const f32x4*>(wchunk + off);                               \
    }                                                                                        \
    }                                                                                        \
    const float* xchunk = xb + (int64_t)cn_ * a.x_rstride;                                   \
    const unsigned rs2 = 2u * (unsigned)a.x_rstride;                                         \
    _Pragma("unroll") for (int u = 0; u < NXP; ++u) {                                        \
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                        \
      if (xoff[u] >= 0) {                                                                    \
        const unsigned o = (unsigned)xoff[u];                                                \
        v[0] = xchunk[o]; v[1] = xchunk[o + rs2]; v[2] = xchunk[o + 2 * rs2]; v[3] = xchunk[o + 3 * rs2]; \
      }                                                                                      \
      xreg[u] = v;                                                                           \
    }                                                                                        \
  }
#define MBV_COMMIT(CN, XS, WS)                                                               \
  {                                                                                          \
    const int cn_ = (CN);                                                                    \
    if constexpr (!GLDS) {                                                                   \
    if (a.debug != 5 || cn_ == 0)                                                            \
    _Pragma("unroll") for (int u = 0; u < NW; ++u) {                                         \
      const int e = tid + NT * u;                                                            \
      if (e < totalW) (WS)[e] = wreg[u];           /* LDS order == copy order */             \
    }                                                                                        \
    }                                                                                        \
    int P = tid / XL, col = tid - P * XL;                                                    \
    _Pragma("unroll") for (int u = 0; u < NXP; ++u) {                                        \
      if (P < 2 * G) {                                                                       \
        f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                      \
        if (xoff[u] >= 0) {                                                                  \
          v = xreg[u];                                                                       \
          if (a.chan_add) {                                                                  \
            const float* ca = a.chan_add + lb * a.Cin + cn_ + (P >> 1) * 8 + (P & 1);        \
            v[0] += ca[0]; v[1] += ca[2]; v[2] += ca[4]; v[3] += ca[6];                      \
          }                                                                                  \
          _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) v[s4] = lrelu(v[s4], a.in_slope); \
        }                                                                                    \
        (XS)[P * XL + col] = PREC == 3 ? split_slot(v) : v;                                  \
      }                                                                                      \
      col += NT;                                                                             \
      _Pragma("unroll") for (int w = 0; w < NT / 128; ++w)                                   \
        if (col >= XL) { col -= XL; ++P; }                                                   \
    }                                                                                        \
  }

  // Weight slab of chunk CN by LDS-DMA: pass u of a wave is 64 consecutive float4 of the slab, in LDS at
  // float4 index (tid & ~63) + NT u — a wave-uniform base + lane * 16 B, which is what
  // global_load_lds_dwordx4 writes.  Issued right after the barrier that retired the last readers of WS,
  // lands under the MFMA loop of the current chunk.  Written as asm on purpose: with the builtin hipcc
  // (ROCm 7.2) drains the whole vector-memory queue at every barrier of the loop — including the input
  // window loads of the chunk after, issued just before it — and the kernel was 2 % SLOWER than register
  // staging.  hipcc does not count asm memory operations, so the ordering is explicit: MBV_GLDS_DRAIN
  // (s_waitcnt vmcnt(0), after the MFMA loop, when the DMA has long landed) before the chunk's barrier,
  // the readers start after that barrier.
#define MBV_GLDS_W(CN, WS)                                                                   \
  {                                                                                          \
    const float* wchunk = wlane + (int64_t)((CN) / 8) * 2 * a.Mpad * 4;                      \
    const int wave_e0 = __builtin_amdgcn_readfirstlane(tid & ~63);                           \
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(                                    \
        (unsigned)(size_t)(__attribute__((address_space(3))) void*)((WS) + wave_e0));        \
    _Pragma("unroll") for (int u = 0; u < NW; ++u) {                                         \
      int64_t off;                                                                           \
      if constexpr (RPI >= ROWS_PER_TAP) {                                                   \
        off = (int64_t)(u * (RPI / ROWS_PER_TAP)) * tap_stride;                              \
      } else {                                                                               \
        constexpr int PPT = ROWS_PER_TAP / RPI;                                              \
        off = (u / PPT) * tap_stride + (int64_t)((u % PPT) * RPI) * a.Mpad * 4;              \
      }                                                                                      \
      if (wave_e0 + NT * u < totalW) {           /* wave-uniform: rows past the slab */      \
        unsigned keep_m0;                                                                    \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"                  \
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"                   \
                     : "=&s"(keep_m0) : "v"(wchunk + off), "s"(lds0 + (unsigned)(NT * u * 16)) : "memory"); \
      }                                                                                      \
    }                                                                                        \
  }
#define MBV_GLDS_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

  // staging cursor: the next chunk to load (runs ahead of the tile being multiplied)
  const int nck = a.Cin / CK;
  const bool stage0_only = a.debug == 1 || a.debug == 4;      // timing experiments: stage chunk 0 of a tile only
  // Split-K (small launches only, S = ksplit > 1): a work unit is (tile, split); split sp owns chunks
  // [sp nck / S, (sp + 1) nck / S).  Units of a tile leave their partial accumulators in a
  // workspace; the last one to arrive (ticket counter) sums them in split order on top of the
  // start values and runs the epilogue.  S == 1: a unit is a tile.
  const int S = NWN == 2 ? ksplit : 1;             // the 512-thread shape is only picked for launches that fill the chip
  if (a.trim_map) total_tiles = a.trim_map[a.B] * tiles_y;    // trimmed decode: the tile count lives on the device
  const int total_units = total_tiles * S;
  __shared__ int s_ticket;
  int s_unit = blockIdx.x;
  int s_c = (s_unit % S) * nck / S, s_end = (s_unit % S + 1) * nck / S;
  int pend_cn = -1;                      // channel offset of the chunk held in wreg/xreg (-1: none)
#define MBV_ISSUE_NEXT()                                                                     \
  if (s_unit < total_units) {                                                                \
    const bool first_ = s_c == (s_unit % S) * nck / S;                                       \
    if (first_) MBV_SETUP_TILE(s_unit / S);                                                  \
    if (first_ || !stage0_only) {                                                            \
      MBV_ISSUE(s_c * CK);                                                                   \
      pend_cn = s_c * CK;                                                                    \
    }                                                                                        \
    if (++s_c == s_end) {                                                                    \
      s_unit += gridDim.x;                                                                   \
      s_c = (s_unit % S) * nck / S;                                                          \
      s_end = (s_unit % S + 1) * nck / S;                                                    \
    }                                                                                        \
  }

  // (experiment, MBV_CONV_START_STAGGER via a.debug >> 8: workgroups start (blockIdx & 7) x that many s_sleep(127)s apart,
  // so that their tile rounds — the start-value load bursts and the epilogue store bursts — stop coinciding)
  if constexpr (DB) {
    const int stg = (a.debug >> 8) & 0xff;
    for (int i = (int)(blockIdx.x & 7) * stg; i > 0; --i) __builtin_amdgcn_s_sleep(127);
  }
  // ---- prime: chunk 0 of the first tile -> LDS buffer 0 -------------------------
  MBV_ISSUE_NEXT();
  {
    f32x4* const X0 = lds4;
    f32x4* const W0 = X0 + G * 2 * XL;
    if (pend_cn >= 0) {
      if constexpr (GLDS) { MBV_GLDS_W(pend_cn, W0); }
      MBV_COMMIT(pend_cn, X0, W0);
    }
    pend_cn = -1;
  }
  if constexpr (GLDS) { MBV_GLDS_DRAIN(); }
  __syncthreads();

  int q = 0;                             // chunks multiplied so far (LDS buffer parity)
  for (int unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
    const int tile = unit / S, sp = unit % S;
    const int c_lo = sp * nck / S, c_hi = (sp + 1) * nck / S;
    int b, m0, t0;
    if (a.trim_map) {
      const int ux = tile / tiles_y;
      b = a.trim_map[a.B + 1 + ux]; m0 = (tile % tiles_y) * BM; t0 = (ux - a.trim_map[b]) * BN;
    } else {
      const int tx = tile % tiles_x, rest = tile / tiles_x;
      b = rest / tiles_y; m0 = (rest % tiles_y) * BM; t0 = tx * BN;
    }
    const int wrow0 = m0 + wm * 32 * WM;
    int nact = (a.M - wrow0 + 31) / 32;              // 32-row tiles of this wave that hold real rows
    nact = nact < 0 ? 0 : (nact > WM ? WM : nact);
    int nj = ((VS ? a.B * a.vs_tv : a.T) - (t0 + wn * 32 * WN) + 31) / 32;  // ... and 32-column tiles inside the sequence
    nj = nj < 0 ? 0 : (nj > WN ? WN : nj);

    // Accumulators start from everything the epilogue would otherwise have to READ after the MFMA
    // loop (residual, running ResBlock sum, the tensor a flow layer updates in place).  gfx9 counts
    // loads and stores on one in-order counter, so a load issued after a store waits for that store
    // to reach L2: an epilogue that interleaves the two is a chain of round trips (measured: a
    // quarter of a k=3 conv).  With the reads up here the epilogue is stores only.
    f32x16 acc[WM][WN];
    MBV_CSTAMP(1)                        // tile start
    // bias (+ per-utterance row terms) of row wrow0 + lane, handed out by cross-lane reads below
    float rowc = 0.f;
    {
      const int row = wrow0 + lane;
      if constexpr (EPI == EPI_GATE) {
        // packed (tanh tile, sigmoid tile) pairs: lane < 32 -> tanh row of channel cbase + lane
        const int ch = (wrow0 >> 6) * 32 + l31;
        if (lane < 32 * WM && ch < a.gate_half) {
          const int idx = hl ? a.gate_half + ch : ch;
          rowc = a.bias ? a.bias[idx] : 0.f;
          if (a.gate_cond) rowc += a.gate_cond[(int64_t)b * a.gate_cond_bstride + idx];
        }
      } else if (lane < 32 * WM && row < a.M) {
        rowc = a.bias ? a.bias[row] : 0.f;
        if constexpr (EPI == EPI_RESID || EPI == EPI_RESID_ACC)
          if (a.res_chan_add) rowc += a.res_chan_add[b * a.M + row];
      }
    }
    // a wave whose 32*WM x 32*WN patch lies wholly inside [M, T] takes the unguarded paths in
    // conv_acc_init and in the epilogue (no per-element exec masking, loads issued back to back)
    const bool full = !VS && wrow0 + 32 * WM <= a.M && t0 + wn * 32 * WN + 32 * WN <= a.T;
    const int64_t lane_off = (int64_t)(wrow0 + 4 * hl) * a.T + t0 + wn * 32 * WN + l31;   // element (k = 0, j = 0)
    if (S == 1) {
      conv_acc_init<WM, WN, EPI>(acc, a, b, wrow0, t0, wn, hl, l31, full, lane_off);
    } else {                                         // split-K: partial sums start from zero
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    if constexpr (DB) { MBV_ISSUE_NEXT(); }   // second chunk of this tile (its first is in LDS)
    // Start values that were LOADED: make them land HERE, by the count of loads issued since (straight-line code, so
    // hipcc emits the exact s_waitcnt vmcnt(n) and the chunk requested above stays in flight).  Without this the
    // accumulators are "pending vector-memory results" at the head of the step loop, where the counter cannot be
    // tracked through the chunk loop's control flow: hipcc then puts s_waitcnt vmcnt(0) in front of the first MFMA
    // of EVERY chunk, which drains the window loads of the chunk after next issued just before it — one exposed global
    // load latency per chunk (r03, found in the assembly: decoder 40.99 -> 40.77 ms at batch 64, profiles/r03x_acc_pin_ab.txt).
    if constexpr (EPI == EPI_RESID || EPI == EPI_RESID_ACC || EPI == EPI_RES_SKIP || EPI == EPI_COUPLE) {
#ifndef MBV_CONV_NO_ACC_PIN               // (the A/B build of profiles/r03x_acc_pin_ab.txt)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(acc[i][j]));
#endif
    }
    MBV_CSTAMP(2)                        // start values requested, second chunk requested

    for (int c = c_lo; c < c_hi; ++c, ++q) {
      f32x4* const Xs = lds4 + (DB ? ((q & 1) ? buf_f4 : 0) : 0);           // buffer holding this chunk
      f32x4* const Ws = Xs + G * 2 * XL;
      f32x4* const Xn = lds4 + (DB ? ((q & 1) ? 0 : buf_f4) : 0);           // buffer for the next one
      f32x4* const Wn = Xn + G * 2 * XL;
      if constexpr (!DB) { MBV_ISSUE_NEXT(); }       // next chunk (possibly the next tile's first)
      if constexpr (GLDS) {
        if (pend_cn >= 0) { MBV_GLDS_W(pend_cn, Wn); }   // its input window is already in flight (xreg)
      }

      MBV_CSTAMP(3)                      // chunk: MFMA loop starts
      if (a.debug != 3) {
        // ---- MFMA over (tap, group) steps; each step = 4 K-steps from one b128 per operand tile
        const int nsteps = a.K * G;
        const f32x4* wbase = Ws + hl * BM + wm * 32 * WM + l31;       // + step * 2 * BM
        const f32x4* xbase = Xs + hl * XL + wn * 32 * WN + l31;       // + g * 2 * XL + tap * dil
        // (split-bf16: a wave with one real row tile takes the same loop — its second tile's weights are the
        // zero rows of the padded slab — because the half-height loop below reads the slots as fp32)
        if (nact == WM || (PREC == 3 && nact > 0)) {
          f32x4 a0[WM], b0[WN], a1[WM], b1[WN];
#define MBV_LOAD_AB(ST, AV, BV)                                                   \
          {                                                                        \
            const int st_ = (ST);                                                  \
            const int tap_ = st_ / G, g_ = st_ % G;                                \
            const f32x4* wp_ = wbase + st_ * (2 * BM);                             \
            const f32x4* xp_ = xbase + g_ * (2 * XL) + tap_ * a.dil;               \
            _Pragma("unroll") for (int i = 0; i < WM; ++i) AV[i] = wp_[i * 32];    \
            _Pragma("unroll") for (int j = 0; j < WN; ++j) BV[j] = xp_[j * 32];    \
          }
          /* EPI_CONVT: the i-tile whose weights are structurally zero at this tap (none: -1) */ \
#define MBV_SKIP(ST) (EPI == EPI_CONVT ? ((ST) / G == 0 ? 1 : ((ST) / G == a.K - 1 ? 0 : -1)) : -1)
#define MBV_MMA(AV, BV, SK, ALLJ)                                                  \
          {                                                                        \
            const int sk_ = (SK);                                                  \
            _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4)                       \
              _Pragma("unroll") for (int i = 0; i < WM; ++i)                       \
                _Pragma("unroll") for (int j = 0; j < WN; ++j)                     \
                  if (((ALLJ) || j < nj) && (EPI != EPI_CONVT || i != sk_))        \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[i][s4], BV[j][s4], acc[i][j], 0, 0, 0); \
          }
          // operands double-buffered one step ahead; sched_barrier keeps hipcc from sinking the
          // prefetch back behind the MFMAs
          // interior patches (every column tile inside the sequence) run a loop without any test
          // between the MFMAs; edge patches the predicated one
#define MBV_STEP_LOOP(ALLJ)                                                        \
          {                                                                        \
            MBV_LOAD_AB(0, a0, b0);                                                \
            int st = 0;                                                            \
            for (; st + 1 < nsteps; st += 2) {                                     \
              MBV_LOAD_AB(st + 1, a1, b1);                                         \
              __builtin_amdgcn_sched_barrier(0);                                   \
              MBV_MMA(a0, b0, MBV_SKIP(st), ALLJ);                                 \
              __builtin_amdgcn_sched_barrier(0);                                   \
              MBV_LOAD_AB(st + 2, a0, b0); /* last pass: one step past the slab (padded, unused) */ \
              __builtin_amdgcn_sched_barrier(0);                                   \
              MBV_MMA(a1, b1, MBV_SKIP(st + 1), ALLJ);                             \
              __builtin_amdgcn_sched_barrier(0);                                   \
            }                                                                      \
            if (st < nsteps) { MBV_MMA(a0, b0, MBV_SKIP(st), ALLJ); } /* odd step count */ \
          }
          // (256-thread shapes only: in the 512-thread shape the second loop body costs ~100 registers
          // of live ranges and the kernel spills)
          // split-bf16: two steps (16 K-values per lane half) per MFMA; the slots are already
          // [hi x 4 | mid x 4] (see split_slot): an operand is two slots' halves, nothing to compute.
          // No column test in either shape: columns past the sequence end are staged as zeros.
#define MBV_BF16_LOOP()                                                                                   \
          for (int st = 0; st < nsteps; st += 2) {                                                         \
            f32x4 ra0[WM], rb0[WN], ra1[WM], rb1[WN];                                                      \
            MBV_LOAD_AB(st, ra0, rb0);                                                                     \
            if (st + 1 < nsteps) {                                                                         \
              MBV_LOAD_AB(st + 1, ra1, rb1);                                                               \
            } else {                                   /* odd step count: the second half of the K-values is absent */ \
              _Pragma("unroll") for (int i = 0; i < WM; ++i) ra1[i] = f32x4{0.f, 0.f, 0.f, 0.f};           \
              _Pragma("unroll") for (int j = 0; j < WN; ++j) rb1[j] = f32x4{0.f, 0.f, 0.f, 0.f};           \
            }                                                                                              \
            bf16x8 ah[WM], am[WM], bh[WN], bm[WN];                                                         \
            _Pragma("unroll") for (int i = 0; i < WM; ++i) { ah[i] = plane_of<0>(ra0[i], ra1[i]); am[i] = plane_of<1>(ra0[i], ra1[i]); } \
            _Pragma("unroll") for (int j = 0; j < WN; ++j) { bh[j] = plane_of<0>(rb0[j], rb1[j]); bm[j] = plane_of<1>(rb0[j], rb1[j]); } \
            /* three rounds over the accumulators: consecutive MFMAs never share one */                   \
            _Pragma("unroll") for (int i = 0; i < WM; ++i)                                                 \
              _Pragma("unroll") for (int j = 0; j < WN; ++j)                                               \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);     \
            _Pragma("unroll") for (int i = 0; i < WM; ++i)                                                 \
              _Pragma("unroll") for (int j = 0; j < WN; ++j)                                               \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);     \
            _Pragma("unroll") for (int i = 0; i < WM; ++i)                                                 \
              _Pragma("unroll") for (int j = 0; j < WN; ++j)                                               \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);     \
          }
          if constexpr (EPI == EPI_CONVT && G == 2 && PREC == 0 && NWN == 4) {   // (256-thread shape: 31 registers spilled)
            // The polyphase ConvTranspose skips the row tile whose weights are structurally zero at the first / last
            // tap.  With the skip decided per MFMA at run time (`i != sk_` inside MBV_MMA, sk_ a function of the
            // step) hipcc put a scalar branch around EVERY MFMA of the loop (r03 assembly audit: 24 branches per
            // step; the two upsampling convs ran at 0.77 of peak against 0.85 for the k = 7 convs).  The stride-4
            // form has exactly ten steps (5 taps x 2 groups): written out, every skip is a compile-time fact.
            if (nj > 0 && a.K == 5) {
              MBV_LOAD_AB(0, a0, b0);
#pragma unroll
              for (int st = 0; st < 10; st += 2) {
                MBV_LOAD_AB(st + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                MBV_MMA(a0, b0, (st / 2 == 0 ? 1 : (st / 2 == 4 ? 0 : -1)), true);
                __builtin_amdgcn_sched_barrier(0);
                MBV_LOAD_AB(st + 2, a0, b0); /* last pass: one step past the slab (padded, unused) */
                __builtin_amdgcn_sched_barrier(0);
                MBV_MMA(a1, b1, (st / 2 == 0 ? 1 : (st / 2 == 4 ? 0 : -1)), true);
                __builtin_amdgcn_sched_barrier(0);
              }
            } else if (nj > 0 && a.K == 3) {     // stride 8 (single-band decoder): six steps
              MBV_LOAD_AB(0, a0, b0);
#pragma unroll
              for (int st = 0; st < 6; st += 2) {
                MBV_LOAD_AB(st + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                MBV_MMA(a0, b0, (st / 2 == 0 ? 1 : (st / 2 == 2 ? 0 : -1)), true);
                __builtin_amdgcn_sched_barrier(0);
                MBV_LOAD_AB(st + 2, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                MBV_MMA(a1, b1, (st / 2 == 0 ? 1 : (st / 2 == 2 ? 0 : -1)), true);
                __builtin_amdgcn_sched_barrier(0);
              }
            } else if (nj > 0) MBV_STEP_LOOP(true)
          } else if constexpr (PREC == 3) {
            if (nj > 0) { MBV_BF16_LOOP() }
          } else if constexpr (NWN == 2) {
            // (r03: one loop for every wave that holds a real column, as in the 512-thread shape — columns past the end of
            // the sequence are staged as zeros and never stored.  The per-MFMA `j < nj` form that edge patches used to
            // take compiled to a scalar branch around every MFMA, and carrying a second loop body cost registers.)
            if (nj > 0) MBV_STEP_LOOP(true)
          } else {
            // 512-thread shape: no column test at all — columns past the end of the sequence are
            // staged as zeros and never stored, so the few MFMAs they cost on the last tile of a row
            // buy a loop without exec-mask juggling around every MFMA
            static_assert(NWN == 4, "");
            if (nj > 0) MBV_STEP_LOOP(true)
          }
#undef MBV_BF16_LOOP
#undef MBV_STEP_LOOP
#undef MBV_LOAD_AB
#undef MBV_MMA
#undef MBV_SKIP
        } else if (nact == 1) {                                // only reachable with WM == 2
          for (int st = 0; st < nsteps; ++st) {
            const int tap = st / G, g = st % G;
            const f32x4 av = wbase[st * (2 * BM)];
            const f32x4* xp = xbase + g * (2 * XL) + tap * a.dil;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              const f32x4 bv = xp[j * 32];
#pragma unroll
              for (int s4 = 0; s4 < 4; ++s4)
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s4], bv[s4], acc[0][j], 0, 0, 0);
            }
          }
        }
      }

      // pend_cn and the cursor depend on blockIdx only: the barriers below are workgroup-uniform
      MBV_CSTAMP(4)                      // chunk: MFMA loop done
      if constexpr (DB) {
        if (pend_cn >= 0) { MBV_COMMIT(pend_cn, Xn, Wn); }   // other buffer: last read one barrier ago
        pend_cn = -1;
        MBV_CSTAMP(9)                    // chunk: input window committed
        if constexpr (GLDS) { MBV_GLDS_DRAIN(); }            // this wave's share of the next weight slab has landed
        MBV_CSTAMP(10)                   // chunk: weight DMA drained
        if (c + 1 < c_hi) { MBV_ISSUE_NEXT(); }              // the last chunk issues AFTER the epilogue
        MBV_CSTAMP(5)                    // chunk: next chunk committed, the one after requested
        __syncthreads();
        MBV_CSTAMP(6)                    // chunk: barrier passed
      } else {
        if (pend_cn >= 0) {
          __syncthreads();                          // every wave is done reading the buffer
          MBV_COMMIT(pend_cn, Xn, Wn);
          pend_cn = -1;
          __syncthreads();
        }
      }
    }

    // ---- epilogue ----------------------------------------------------------
    // accumulator layout (32x32 tile): column = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    // stores only (see the accumulator initialisation above)
    MBV_CSTAMP(7)                        // epilogue starts
    bool skip_epi = a.debug == 4 && acc[0][0][0] != 12345.678f;         // timing experiment: no epilogue traffic
    if (S > 1) {
      const size_t tile_floats = (size_t)BM * BN;                      // == 16 WM WN NT
      float* wsu = a.ws + ((size_t)tile * S + sp) * tile_floats + tid;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) wsu[(size_t)((i * WN + j) * 16 + r) * NT] = acc[i][j][r];
      __threadfence();                                                 // partials visible device-wide ...
      __syncthreads();
      if (tid == 0) {
        const unsigned t = atomicAdd(a.counters + tile, 1u);           // ... before the ticket is taken
        if (t == (unsigned)S - 1) a.counters[tile] = 0;                // self-resetting for the next launch
        s_ticket = (int)t;
      }
      __syncthreads();
      if (s_ticket != S - 1) {
        skip_epi = true;
      } else {
        __threadfence();                                               // drop stale lines before reading the others' partials
        conv_acc_init<WM, WN, EPI>(acc, a, b, wrow0, t0, wn, hl, l31, full, lane_off);
        const float* w0 = a.ws + (size_t)tile * S * tile_floats + tid;
        for (int sq = 0; sq < S; ++sq) {                               // fixed order: deterministic
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                acc[i][j][r] += w0[(size_t)sq * tile_floats + (size_t)((i * WN + j) * 16 + r) * NT];
        }
      }
    }
    const int T = a.T;
    const int len_out = a.out_lens ? a.out_lens[b] : 0x7fffffff;
    if (skip_epi) {
    } else if constexpr (EPI == EPI_GATE) {
      if constexpr (WM == 2) {
        if (nact == 2) {
          const int cbase = (wrow0 >> 6) * 32;            // channel of packed tile pair
          float* yb = a.y + (int64_t)b * a.y_bstride;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * hl;
            const float bt = __shfl(rowc, k), bs = __shfl(rowc, 32 + k);
            const int c = cbase + k;
            if (c >= a.gate_half) continue;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              const int t = t0 + wn * 32 * WN + j * 32 + l31;
              if (t < T) {
                const float vt = tanhf(acc[0][j][r] + bt);
                const float vs = sigmoidf_(acc[1][j][r] + bs);
                yb[(int64_t)c * T + t] = vt * vs;
              }
            }
          }
        }
      }
    } else if constexpr (EPI == EPI_CONVT) {
      if constexpr (WM == 2) {
        float* yb = a.y + (int64_t)b * a.y_bstride;
        if (a.convt_u == 4) {
          // lane: column t, rows k (even) and k + 1 of both i-tiles = the 4 output phases of one channel
          const int Tout = 4 * T;
          int64_t vs_o[VS ? WN : 1];                   // virtual column -> offset of (utterance, 4 x frame) in y
          bool vs_ok[VS ? WN : 1];
          if constexpr (VS) {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              const int tv = t0 + wn * 32 * WN + j * 32 + l31;
              const int bq = tv / a.vs_tv, tq = tv - bq * a.vs_tv;
              vs_ok[j] = bq < a.B && tq < T;
              vs_o[j] = (int64_t)bq * a.y_bstride + 4 * (int64_t)tq;
            }
          }
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * hl;
            const float bias = __shfl(rowc, k);
            const int co = (wrow0 >> 6) * 16 + (k >> 1);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              const int t = t0 + wn * 32 * WN + j * 32 + l31;
              if constexpr (VS) {
                if (vs_ok[j]) {
                  f32x4 o = {acc[0][j][r] + bias, acc[0][j][r + 1] + bias, acc[1][j][r] + bias,
                             acc[1][j][r + 1] + bias};
                  *reinterpret_cast<f32x4*>(a.y + vs_o[j] + (int64_t)co * Tout) = o;
                }
              } else
              if (t < T) {
                f32x4 o = {acc[0][j][r] + bias, acc[0][j][r + 1] + bias, acc[1][j][r] + bias,
                           acc[1][j][r + 1] + bias};
                *reinterpret_cast<f32x4*>(yb + (int64_t)co * Tout + 4 * (int64_t)t) = o;
              }
            }
          }
        } else {
          // stride 8: registers 4 q .. 4 q + 3 are rows k .. k + 3 = phases 0-3 (i = 0) / 4-7 (i = 1) of one channel
          const int Tout = 8 * T;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int k = 8 * rg + 4 * hl;
            const float bias = __shfl(rowc, k);
            const int co = (wrow0 >> 6) * 8 + (k >> 2);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              const int t = t0 + wn * 32 * WN + j * 32 + l31;
              if (t < T) {
                float* dst = yb + (int64_t)co * Tout + 8 * (int64_t)t;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                  f32x4 o = {acc[i][j][4 * rg] + bias, acc[i][j][4 * rg + 1] + bias,
                             acc[i][j][4 * rg + 2] + bias, acc[i][j][4 * rg + 3] + bias};
                  *reinterpret_cast<f32x4*>(dst + 4 * i) = o;
                }
              }
            }
          }
        }
      }
    } else if ((EPI == EPI_STORE || EPI == EPI_RESID || EPI == EPI_RESID_ACC) && full && !a.out_lens) {
      float* py = a.y + (int64_t)b * a.y_bstride + lane_off;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k0 = i * 32 + (r & 3) + 8 * (r >> 2);
          const float bias = __shfl(rowc, k0 + 4 * hl);
          float* q = py + (int64_t)k0 * T;
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            float v = acc[i][j][r] + bias;
            if constexpr (EPI == EPI_STORE) { if (a.relu) v = fmaxf(v, 0.f); }
            if constexpr (EPI == EPI_RESID_ACC) v *= a.out_scale;
            q[j * 32] = v;
          }
        }
    } else if (EPI == EPI_RES_SKIP && full && a.split % 32 == 0) {
      const int64_t col_off = (int64_t)(4 * hl) * T + t0 + wn * 32 * WN + l31;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int trow = wrow0 + i * 32;
        const bool is_x = trow < a.split;
        float* base = is_x ? a.y + (int64_t)b * a.y_bstride + (int64_t)trow * T
                           : a.skip + ((int64_t)b * (a.M - a.split) + (trow - a.split)) * T;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k0 = (r & 3) + 8 * (r >> 2);
          const float bias = __shfl(rowc, i * 32 + k0 + 4 * hl);
          float* q = base + col_off + (int64_t)k0 * T;
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            const int t = t0 + wn * 32 * WN + j * 32 + l31;
            float v = acc[i][j][r] + bias;
            if (is_x) v = t < len_out ? v : 0.f;                   // xio = (xio + rs) * mask
            q[j * 32] = v;
          }
        }
      }
    } else if (EPI == EPI_COUPLE && full) {
      float* py = a.y + (int64_t)b * a.y_bstride + lane_off;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k0 = i * 32 + (r & 3) + 8 * (r >> 2);
          const float bias = __shfl(rowc, k0 + 4 * hl);
          float* q = py + (int64_t)k0 * T;
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            const int t = t0 + wn * 32 * WN + j * 32 + l31;
            const float v = a.couple_sign * (acc[i][j][r] + bias);
            q[j * 32] = t < len_out ? v : 0.f;
          }
        }
    } else {
#pragma unroll
      for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
          const float bias = __shfl(rowc, k);              // every lane takes part (no early exit above)
          const int row = wrow0 + k;
          if (i >= nact || row >= a.M) continue;
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            const int t = t0 + wn * 32 * WN + j * 32 + l31;
            if (t >= T) continue;
            float v = acc[i][j][r] + bias;
            const float mask = t < len_out ? 1.f : 0.f;
            const int64_t o = (int64_t)b * a.y_bstride + (int64_t)row * T + t;
            if constexpr (EPI == EPI_STORE) {
              if (a.relu) v = fmaxf(v, 0.f);
              if (a.out_lens) v *= mask;
              a.y[o] = v;
            } else if constexpr (EPI == EPI_RESID) {
              a.y[o] = v;
            } else if constexpr (EPI == EPI_RESID_ACC) {
              a.y[o] = v * a.out_scale;
            } else if constexpr (EPI == EPI_RES_SKIP) {
              if (row < a.split) a.y[o] = v * mask;
              else a.skip[((int64_t)b * (a.M - a.split) + (row - a.split)) * T + t] = v;
            } else if constexpr (EPI == EPI_COUPLE) {
              a.y[o] = a.couple_sign * v * mask;
            }
          }
        }
      }
    }
    MBV_CSTAMP(8)                        // epilogue issued (stores in flight)
  }
  MBV_CSTAMP_FLUSH()
#undef MBV_ISSUE_NEXT
#undef MBV_SETUP_TILE
#undef MBV_ISSUE
#undef MBV_COMMIT
#undef MBV_GLDS_W
#undef MBV_GLDS_DRAIN
}

template <int WM, int WN, int CK, int NWN, int EPI, int PREC = 0, int NWM = 2, int VS = 0>
static void launch_epi(const ConvArgs& a, hipStream_t s) {
  constexpr int BM = 32 * WM * NWM, BN = 32 * WN * NWN, G = CK / 8, NT = 64 * NWM * NWN;
  const int XL = BN + (a.K - 1) * a.dil;
  // float4 units per buffer: input image + weight slab + one step of padding for the operand
  // prefetch overrun; two buffers for the 512-thread variant
  const size_t buf_f4 = (size_t)G * 2 * XL + (size_t)a.K * G * 2 * BM + 2 * BM;
  const size_t lds_bytes = buf_f4 * 16 * (NWN == 4 ? 2 : 1);
  // VS: column tiles run through the batch laid end to end (utterance b at virtual column b * vs_tv); the kernel's
  // tile decode then yields utterance 0 for every tile and the addressing above does the rest
  const int tiles_x = VS ? (int)(((long)a.B * a.vs_tv + BN - 1) / BN) : (a.T + BN - 1) / BN, tiles_y = (a.M + BM - 1) / BM;
  const long total = VS ? (long)tiles_x * tiles_y : (long)tiles_x * tiles_y * a.B;
  // persistent tiles: at most the workgroups that are resident at once (1 per CU for the
  // 512-thread shape, 2 for the 256-thread one: both are register-limited to 2 waves / SIMD)
  static const int persist = [] { const char* e = getenv("MBV_CONV_PERSIST"); return e ? atoi(e) : 1; }();
  const long slots = 256L * (NT == 512 ? 1 : 2);
  // Split-K for launches that fill less than a quarter of the chip (single utterances: 34 tiles of a
  // 128-channel decoder conv at batch 1): at least two chunks per split, workspace and ticket
  // counters permitting.  OPT-IN (mbv_set_option("splitk", 1) or MBV_CONV_SPLITK=1: the low-latency service setting): the order of
  // summation then depends on the launch size, so a row computed inside a large batch is no longer
  // bitwise equal to the same row computed alone (it is within fp32 rounding); the default keeps
  // that property.  Every parity test passes in either mode.
  const int splitk = a.splitk;
  int S = 1;
  const int nck = a.Cin / CK;
  // an almost empty chip (<= 32 tiles: one utterance) repays splits of two chunks; up to a quarter
  // full, K loops of >= 16 chunks cut into >= 4-chunk pieces (measured at batch 1 and 8)
  static const int tiny_div = [] { const char* e = getenv("MBV_SPLITK_TINY"); return e ? atoi(e) : 16; }();
  const bool tiny = total * (long)tiny_div <= slots;
  const int min_chunks = tiny ? 2 : 4;
  if (a.trim_map && a.trim_bn != BN) {
    fprintf(stderr, "mbv: trimmed decode: tile map built for %d-column tiles, the launch uses %d\n", a.trim_bn, BN);
    abort();
  }
  if (NWN == 2 && splitk && a.ws && a.counters && total * 4 <= slots && nck >= (tiny ? 4 : 16) &&
      total <= a.n_counters && !a.trim_map) {
    S = (int)(slots / total);
    if (S > nck / min_chunks) S = nck / min_chunks;
    if (S > 16) S = 16;
    while (S > 1 && (size_t)total * S * BM * BN > a.ws_floats) --S;
    if (S < 1) S = 1;
  }
  const long units = total * S;
  const int grid = (int)((persist && units > slots) ? slots : units);
  ConvArgs a2 = a;
  static const int dbg = [] { const char* e = getenv("MBV_CONV_DEBUG"); return e ? atoi(e) : 0; }();
  static const int stg = [] { const char* e = getenv("MBV_CONV_START_STAGGER"); return e ? atoi(e) : 0; }();
  a2.debug = dbg | ((stg & 0xff) << 8);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_mfma_kernel<WM, WN, CK, NWN, EPI, PREC, NWM, VS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
#ifdef MBV_CONV_STAMPS
                              150 * 1024);                        // static LDS: the ticket + 8 KB of stamps
#else
                              160 * 1024 - 256);                  // static LDS: the split-K ticket
#endif
    attr = true;
  }
  hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN, CK, NWN, EPI, PREC, NWM, VS>), dim3(grid), dim3(NT), lds_bytes, s, a2,
                     tiles_x, tiles_y, (int)total, S);
}

// The epilogue is a compile-time parameter: with a run-time switch inside the unrolled
// (i, r, j) nest every kernel carried all six epilogues (21 k instructions after the last MFMA,
// SGPRs spilled to VGPR lanes) and a k=3 conv spent a quarter of its time walking that code.
template <int WM, int WN, int CK, int NWN>
static void launch_one(const ConvArgs& a, hipStream_t s) {
  // opt-in split-bf16 (ConvArgs::prec): built for the 128-row shapes, k > 1, and the epilogues of the
  // decoder's convs, where the time is; everything else stays exact fp32
  if constexpr (WM == 2) {
    if (a.prec == 3) {
      if constexpr (CK <= 16) {
        switch (a.epi) {
          case EPI_STORE: launch_epi<WM, WN, CK, NWN, EPI_STORE, 3>(a, s); return;
          case EPI_RESID: launch_epi<WM, WN, CK, NWN, EPI_RESID, 3>(a, s); return;
          case EPI_RESID_ACC: launch_epi<WM, WN, CK, NWN, EPI_RESID_ACC, 3>(a, s); return;
          case EPI_CONVT:
            if constexpr (CK == 16) { launch_epi<WM, WN, CK, NWN, EPI_CONVT, 3>(a, s); return; }
            break;
          case EPI_GATE:                             // the two-launch WN layer of the flows in this mode (capi.hip run_wn)
            if constexpr (CK == 16) { launch_epi<WM, WN, CK, NWN, EPI_GATE, 3>(a, s); return; }
            break;
          default: break;
        }
      } else {
        if (a.epi == EPI_RES_SKIP) { launch_epi<WM, WN, CK, NWN, EPI_RES_SKIP, 3>(a, s); return; }
      }
    }
  }
  switch (a.epi) {
    case EPI_STORE: launch_epi<WM, WN, CK, NWN, EPI_STORE>(a, s); break;
    case EPI_RESID: launch_epi<WM, WN, CK, NWN, EPI_RESID>(a, s); break;
    case EPI_RESID_ACC: launch_epi<WM, WN, CK, NWN, EPI_RESID_ACC>(a, s); break;
    case EPI_RES_SKIP: launch_epi<WM, WN, CK, NWN, EPI_RES_SKIP>(a, s); break;
    case EPI_COUPLE: launch_epi<WM, WN, CK, NWN, EPI_COUPLE>(a, s); break;
    case EPI_GATE:
      if constexpr (WM == 2) { launch_epi<WM, WN, CK, NWN, EPI_GATE>(a, s); break; }
    case EPI_CONVT:
      if constexpr (WM == 2 && CK == 16) {
        if (a.epi == EPI_CONVT) { launch_epi<WM, WN, CK, NWN, EPI_CONVT>(a, s); break; }
      }
    default:
      fprintf(stderr, "mbv: conv1d epilogue %d not built for this tile shape\n", a.epi);
      abort();
  }
}

bool conv1d_supported(int K, int dil) {
  return K >= 1 && K <= 11 && (K - 1) * dil <= 72;
}

template <int WM, int WN, int NWN>
static void launch_ck(const ConvArgs& a, hipStream_t s) {
  // chunk of input channels staged per LDS pass (register budget of the async stage:
  // CK=8: K <= 11, halo <= 72; CK=16: K <= 5, halo <= 24; CK=32: K == 1)
  const int halo = (a.K - 1) * a.dil;
  if (!conv1d_supported(a.K, a.dil)) {
    fprintf(stderr, "mbv: conv1d kernel size %d / dilation %d outside the built range\n", a.K, a.dil);
    abort();
  }
  if (a.K == 1) launch_one<WM, WN, 32, NWN>(a, s);
  else if (a.K <= 5 && halo <= 24) launch_one<WM, WN, 16, NWN>(a, s);
  else launch_one<WM, WN, 8, NWN>(a, s);
}

// trimmed decode: the column-tile width launch_conv1d will pick for this conv (the tile map is built for it)
int conv1d_trim_bn(const ConvArgs& a) {
  if (a.epi == EPI_LN || a.splitk) return 0;
  {   // a conv that launch_conv1d sends to the narrow kernel (<= 256 frames) keeps that route: trimming must never
      // change which kernel — i.e. which summation order — computes a sample
    static const int narrow = [] { const char* e = getenv("MBV_CONV_NARROW"); return e ? atoi(e) : 1; }();
    if (narrow && conv1d_narrow_supported(a) && (a.T <= 256 || narrow == 2)) return 0;
  }
  const bool wide_m = a.M > 64 || a.epi == EPI_GATE || a.epi == EPI_CONVT;
  static const int mode = [] { const char* e = getenv("MBV_CONV_WIDE"); return e ? atoi(e) : 3; }();
  if (wide_m && mode >= 3 && a.T >= 384) {
    const double eff2 = 0.90 * a.T / (double)(((a.T + 127) / 128) * 128);
    const double eff3 = a.T / (double)(((a.T + 383) / 384) * 384);
    const long blocks3 = (long)((a.T + 383) / 384) * ((a.M + 127) / 128) * a.B;
    if (eff3 >= eff2 && blocks3 >= 512) return 384;
  }
  return 128;
}

void launch_conv1d(const ConvArgs& a, hipStream_t s) {
  // (the residual / running-sum start values are read through a 32-bit buffer view of one utterance, conv_acc_init)
  if ((a.epi == EPI_RESID || a.epi == EPI_RESID_ACC) && (unsigned long long)a.M * a.T * 4ull >= (1ull << 32)) {
    fprintf(stderr, "mbv: conv1d: one utterance's [%d x %d] output exceeds 4 GiB\n", a.M, a.T);
    abort();
  }
  // conv1d_narrow.hip (32-column units, rows split over waves, weights from L2) takes over where the
  // 128-column tiles below fit badly:
  //   (a) every sequence of <= 256 frames (the text encoder, the duration predictor): T = 200 fills
  //       1.56 tiles of 128 columns; a rule on T alone, so a row's arithmetic never depends on the batch;
  //   (b) in the opt-in low-latency mode, launches that would leave most of the chip idle (single
  //       utterances) instead of 128-column tiles + split-K: like split-K this makes the summation
  //       order a function of the launch size.
  // MBV_CONV_NARROW: 0 = never, 1 = these rules, 2 = whenever supported (experiments).
  {
    static const int narrow = [] { const char* e = getenv("MBV_CONV_NARROW"); return e ? atoi(e) : 1; }();
    if (a.epi == EPI_LN) {
      if (!conv1d_narrow_supported(a)) { fprintf(stderr, "mbv: EPI_LN outside the narrow kernel's range\n"); abort(); }
      launch_conv1d_narrow(a, false, s);
      return;
    }
    if (narrow && !a.trim_map && conv1d_narrow_supported(a)) {
      const long tiles128 = (long)((a.T + 127) / 128) * ((a.M + 127) / 128) * a.B;
      // (fewer than 16 units — conv_o / conv_2 of the text encoder of one short utterance: 8 — are better
      // served by split-K over the long Cin loop than by 8 workgroups walking it alone; measured on ljs_mb,
      // one utterance: threshold 48 / 16 / 4 -> 4.43 / 4.15 / 4.34 ms per infer)
      const long units32 = (((long)a.B * ((a.T + 15) / 16) + 1) / 2) * ((a.M + 127) / 128);
      if (a.splitk && tiles128 <= 128) {
        static const int min_units = [] { const char* e = getenv("MBV_NARROW_MIN_UNITS"); return e ? atoi(e) : 16; }();
        if (units32 >= min_units || narrow == 2) { launch_conv1d_narrow(a, true, s); return; }
      } else if (a.T <= 256 || narrow == 2) {
        launch_conv1d_narrow(a, false, s);
        return;
      }
    }
  }
  const bool wide_m = a.M > 64 || a.epi == EPI_GATE || a.epi == EPI_CONVT;
  // Long sequences with enough blocks to fill the chip: 512-thread workgroups, 128 x 384 tile
  // (6 accumulators per wave), double-buffered LDS.  Otherwise 256-thread, 128 x 128 tile.
  static const int mode = [] { const char* e = getenv("MBV_CONV_WIDE"); return e ? atoi(e) : 3; }();
  bool big = false;
  int nb_big = 0;                          // r03: > 0: the first nb_big utterances on the 512-thread shape, the rest on the 256-thread one
  if (wide_m && mode >= 3 && a.T >= 384) {
    const double eff2 = 0.90 * a.T / (double)(((a.T + 127) / 128) * 128);
    const double eff3 = a.T / (double)(((a.T + 383) / 384) * 384);
    const long tpb3 = (long)((a.T + 383) / 384) * ((a.M + 127) / 128);          // tiles per utterance, 128 x 384
    const long tpb2 = (long)((a.T + 127) / 128) * ((a.M + 127) / 128);          // ... 128 x 128
    const long blocks3 = tpb3 * a.B;
    big = eff3 >= eff2 && blocks3 >= 512;
    // Launches between one and two rounds of 128 x 384 tiles (the per-GPU shares of the sharded configurations:
    // uudb B = 32 has 384 tiles for its 256-channel convs) quantise badly on either shape: 1.5 rounds of big
    // tiles run as 2, 2.25 rounds of small ones as 3.  Both shapes compute an output element with the same
    // chain of operations, so the batch can be cut between them: whole rounds of big tiles (one per CU) for the
    // first utterances, the rest as 128 x 128 tiles on the 512 half-CU slots.  In units of one big tile's time
    // (a small tile on half a CU: 1/3 of the work on 1/2 of the waves, measured 0.63 - 0.65):
    static const int split_on = [] { const char* e = getenv("MBV_CONV_BATCH_SPLIT"); return e ? atoi(e) : 1; }();
    if (split_on && !a.splitk && !a.trim_map && eff3 >= eff2 && blocks3 > 256 && a.B > 1) {
      const double c2 = 0.65;
      auto rounds = [](long n, long slots) { return (double)((n + slots - 1) / slots); };
      double best = big ? rounds(blocks3, 256) : rounds(tpb2 * a.B, 512) * c2;
      for (long k = blocks3 / 256; k >= 1; --k) {
        const long nb = 256 * k / tpb3;                                          // utterances that fill k rounds (or just under)
        if (nb < 1 || nb >= a.B) continue;
        const double cost = rounds(nb * tpb3, 256) + rounds((a.B - nb) * tpb2, 512) * c2;
        if (cost < 0.9 * best) { best = cost / 0.9; nb_big = (int)nb; }      // (predicted gains below ~10 % did not materialise: B = 96 measured +4 %)
      }
    }
  }
  // r03: the stride-4 upsampling conv on few, long-ish sequences (T' = 566 input frames: 1.47 tiles of 384 columns, 4.4 of
  // 128) tiled over the VIRTUAL sequence of the whole batch — utterance b at column b (T + halo), the gaps reading as
  // zero padding — so that only the last tile of the launch is ragged.  Same chain of operations per output element.
  {
    static const int vs_on = [] { const char* e = getenv("MBV_CONV_VS"); return e ? atoi(e) : 1; }();
    const int halo = (a.K - 1) * a.dil;
    if (vs_on && a.epi == EPI_CONVT && a.convt_u == 4 && a.prec != 3 && !a.splitk && !a.trim_map && !a.chan_add &&
        !a.reflect1 && a.B > 1 && a.K > 1 && a.K <= 5 && halo <= 24 && a.T >= 128 &&
        (int64_t)a.B * a.x_bstride < (1ll << 31)) {
      auto rounds = [](long n, long slots) { return (double)((n + slots - 1) / slots); };
      const long tiles_y = (a.M + 127) / 128;
      const long tv = a.T + halo;
      const double cost_vs = rounds(((long)a.B * tv + 383) / 384 * tiles_y, 256);
      double cur;
      if (nb_big > 0) {
        const long tpb3 = (long)((a.T + 383) / 384) * tiles_y, tpb2 = (long)((a.T + 127) / 128) * tiles_y;
        cur = rounds(nb_big * tpb3, 256) + rounds((a.B - nb_big) * tpb2, 512) * 0.65;
      } else if (big) {
        cur = rounds((long)((a.T + 383) / 384) * tiles_y * a.B, 256);
      } else {
        cur = rounds((long)((a.T + 127) / 128) * tiles_y * a.B, 512) * 0.65;
      }
      if (cost_vs >= 2.0 && cost_vs < 0.95 * cur) {
        ConvArgs av = a;
        av.vs_tv = (int)tv;
        launch_epi<2, 3, 16, 4, EPI_CONVT, 0, 2, 1>(av, s);
        return;
      }
    }
  }
  if (nb_big > 0) {
    ConvArgs a1 = a, a2 = a;
    const int nb = nb_big;
    a1.B = nb;
    a2.B = a.B - nb;
    a2.x += (int64_t)nb * a.x_bstride;
    a2.y += (int64_t)nb * a.y_bstride;
    if (a.in_lens) a2.in_lens += nb;
    if (a.out_lens) a2.out_lens += nb;
    if (a.chan_add) a2.chan_add += (int64_t)nb * a.Cin;
    if (a.res) a2.res += (int64_t)nb * a.res_bstride;
    if (a.res_chan_add) a2.res_chan_add += (int64_t)nb * a.M;
    if (a.accum_in) a2.accum_in += (int64_t)nb * a.y_bstride;
    if (a.gate_cond) a2.gate_cond += (int64_t)nb * a.gate_cond_bstride;
    if (a.skip) a2.skip += (int64_t)nb * (a.M - a.split) * a.T;
    launch_ck<2, 3, 4>(a1, s);
    launch_ck<2, 2, 2>(a2, s);
    return;
  }
  if (wide_m) {
    if (big) launch_ck<2, 3, 4>(a, s);
    else launch_ck<2, 2, 2>(a, s);
  } else {
    // <= 64 output rows: the 64 x 384 shape when the launch fills its 512 half-CU slots (exact fp32, K > 1, the
    // decoder's epilogues); else 64 x 128.  Same chain of operations per output element either way.
    static const int half_on = [] { const char* e = getenv("MBV_CONV_HALF"); return e ? atoi(e) : 1; }();
    if (half_on && mode >= 3 && a.T >= 384 && a.K > 1 && a.prec != 3 && !a.splitk && !a.trim_map &&
        (long)((a.T + 383) / 384) * a.B >= 512 && a.T / (double)(((a.T + 383) / 384) * 384) >= 0.90 * a.T / (double)(((a.T + 127) / 128) * 128)) {
      const bool ck16 = a.K <= 5 && (a.K - 1) * a.dil <= 24;
      bool done = true;
      if (ck16) {
        switch (a.epi) {
          case EPI_STORE: launch_epi<2, 3, 16, 4, EPI_STORE, 0, 1>(a, s); break;
          case EPI_RESID: launch_epi<2, 3, 16, 4, EPI_RESID, 0, 1>(a, s); break;
          case EPI_RESID_ACC: launch_epi<2, 3, 16, 4, EPI_RESID_ACC, 0, 1>(a, s); break;
          default: done = false;
        }
      } else {
        switch (a.epi) {
          case EPI_STORE: launch_epi<2, 3, 8, 4, EPI_STORE, 0, 1>(a, s); break;
          case EPI_RESID: launch_epi<2, 3, 8, 4, EPI_RESID, 0, 1>(a, s); break;
          case EPI_RESID_ACC: launch_epi<2, 3, 8, 4, EPI_RESID_ACC, 0, 1>(a, s); break;
          default: done = false;
        }
      }
      if (done) return;
    }
    launch_ck<1, 2, 2>(a, s);
  }
}

// ============================================================================
// ConvTranspose1d(k=16, stride=U, padding=(16-U)/2), U in {4, 8} — models.py:321-323 (mb/ms: 4),
// models.py:264-267 (single-band iSTFT_Generator: 8).
// Polyphase form: output phase r = t mod U of frame m = t / U is a (16/U)-tap conv
//   y[co, U m + r] = bias + sum_ci sum_j W[ci][co][kr + U j] * act(x[ci, m + sh_r - j])
//   kr = (r + p) % U, sh_r = (r + p - kr) / U   (p = padding)
// All phases share the input window x[m + sh_0 - (TPP-1) .. m + sh_0 + 1]; one wave accumulates
// the U phase tiles of a (32 co x 32 m) patch, so each lane ends up owning U consecutive output
// samples -> 16-byte stores.
// ============================================================================
template <int U, int CK>
__global__ __launch_bounds__(256) void convt_mfma_kernel(const ConvTArgs a) {
  constexpr int TPP = 16 / U;                 // taps per phase
  constexpr int PAD = (16 - U) / 2;
  constexpr int SH0 = PAD / U;                // sh_r of phase 0 (phases with kr wrapped get SH0 + 1)
  constexpr int BMc = 64;                     // output channels per block
  constexpr int BNm = 64;                     // input frames per block
  constexpr int XS = BNm + TPP;
  __shared__ __attribute__((aligned(16))) float Xs[CK * XS];
  __shared__ __attribute__((aligned(16))) float Ws[16 * CK * BMc];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hl = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, co0 = blockIdx.y * BMc, mb0 = blockIdx.x * BNm;

  f32x16 acc[U];
#pragma unroll
  for (int p = 0; p < U; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

  const float* xb = a.x + (int64_t)b * a.Cin * a.Tin;
  for (int ci0 = 0; ci0 < a.Cin; ci0 += CK) {
    __syncthreads();
    for (int e = tid; e < CK * XS; e += 256) {
      const int r = e / XS, c = e % XS;
      const int n = mb0 + SH0 - (TPP - 1) + c;
      float v = 0.f;
      if (n >= 0 && n < a.Tin) v = lrelu(xb[(int64_t)(ci0 + r) * a.Tin + n], a.in_slope);
      Xs[e] = v;
    }
    {
      constexpr int Q = BMc / 4;
      for (int e = tid; e < 16 * CK * Q; e += 256) {
        const int row = e / Q, q = e % Q;      // row = (r*TPP + j)*CK + c
        const int rj = row / CK, c = row % CK;
        const float4* src = reinterpret_cast<const float4*>(
            a.w + ((int64_t)rj * a.Cin + ci0 + c) * a.Mpad + co0) + q;
        reinterpret_cast<float4*>(Ws + row * BMc)[q] = *src;
      }
    }
    __syncthreads();
#pragma unroll
    for (int c2 = 0; c2 < CK / 2; ++c2) {
      const float* xrow = Xs + (2 * c2 + hl) * XS + wn * 32 + l31;
      const float* wbase = Ws + (2 * c2 + hl) * BMc + wm * 32 + l31;
#pragma unroll
      for (int tau = 0; tau <= TPP; ++tau) {
        const float bv = xrow[tau];
#pragma unroll
        for (int r = 0; r < U; ++r) {
          constexpr int dummy = 0; (void)dummy;
          const int sh = ((r + PAD) - ((r + PAD) % U)) / U - SH0;   // 0 or 1 (compile-time after unroll)
          const int j = sh + TPP - 1 - tau;
          if (j >= 0 && j < TPP) {
            const float av = wbase[((r * TPP + j) * CK) * BMc];
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[r], 0, 0, 0);
          }
        }
      }
    }
  }
  const int m = mb0 + wn * 32 + l31;
  if (m < a.Tin) {
    const int Tout = U * a.Tin;
    float* yb = a.y + (int64_t)b * a.Cout * Tout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
      if (co < a.Cout) {
        const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int q = 0; q < U / 4; ++q) {
          float4 o;
          o.x = acc[4 * q + 0][r] + bias; o.y = acc[4 * q + 1][r] + bias;
          o.z = acc[4 * q + 2][r] + bias; o.w = acc[4 * q + 3][r] + bias;
          *reinterpret_cast<float4*>(yb + (int64_t)co * Tout + U * m + 4 * q) = o;
        }
      }
    }
  }
}

void launch_convt(const ConvTArgs& a, hipStream_t s) {
  dim3 grid((a.Tin + 63) / 64, (a.Cout + 63) / 64, a.B);
  if (a.stride == 8) hipLaunchKernelGGL((convt_mfma_kernel<8, 8>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((convt_mfma_kernel<4, 8>), grid, dim3(256), 0, s, a);
}

}  // namespace mbv
