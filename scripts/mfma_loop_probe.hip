// What does the conv kernel's MFMA step loop reach by itself?  The loop of conv1d.hip (per step: 2 + 3 ds_read_b128
// prefetched one step ahead, 24 v_mfma_f32_32x32x2_f32 on 6 accumulators) without any staging, as variants:
//   0  512 threads (two waves per SIMD), one s_barrier per chunk of STEPS steps     (the shipped structure)
//   1  512 threads, no barrier
//   2  as 1, waves 4-7 half a step (12 MFMAs) out of phase
//   3  as 1, the five reads spread over the step (one read behind every fourth MFMA) instead of one burst
//   4  as 1, s_setprio 1 on waves 4-7
//   5  256 threads (one wave per SIMD), no barrier
//   6  as 0, waves 4-7 half a step out of phase (their extra half step sits in front of the first chunk)
//   7  as 0 with 2 x 256-thread workgroups per CU (independent barriers)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma_loop_probe.hip -o scripts/mfma_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LOAD_AB(ST, AV, BV)                                                        \
  {                                                                                \
    const f32x4* wp_ = wbase + (ST) * 256;                                         \
    const f32x4* xp_ = xbase + ((ST) % 7) * 3;                                     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) AV[i] = wp_[i * 32];             \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) BV[j] = xp_[j * 32];             \
  }
#define MMA(AV, BV)                                                                \
  _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4)                                 \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                  \
      _Pragma("unroll") for (int j = 0; j < 3; ++j)                                \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[i][s4], BV[j][s4], acc[i][j], 0, 0, 0);

template <int VAR, int NT>
__global__ __launch_bounds__(NT, 2) void k(float* out, int chunks, int steps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4* const lds4 = reinterpret_cast<f32x4*>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4096; i += NT) lds4[i] = f32x4{0.001f * i, 1.f, -0.5f, 0.25f};
  __syncthreads();
  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int hl = lane >> 5, l31 = lane & 31, wm = wave / 4, wn = wave % 4;
  const f32x4* wbase = lds4 + hl * 128 + wm * 64 + l31;         // + step * 256 (<= 11 steps -> < 3072)
  const f32x4* xbase = lds4 + 3072 + hl * 448 + wn * 96 + l31;   // 3072 .. 4095
  f32x4 a0[2], b0[3], a1[2], b1[3];
  if constexpr (VAR == 4) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
  if constexpr (VAR == 2 || VAR == 6) {
    if (wave >= 4) {                                              // half a step ahead of the partner wave
      LOAD_AB(0, a0, b0);
#pragma unroll
      for (int s4 = 0; s4 < 2; ++s4)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][s4], b0[j][s4], acc[i][j], 0, 0, 0);
    }
  }
  for (int c = 0; c < chunks; ++c) {
    LOAD_AB(0, a0, b0);
    int st = 0;
    for (; st + 1 < steps; st += 2) {
      if constexpr (VAR == 3) {
        // reads of step st + 1 spread behind the MFMAs of step st
        const f32x4* wp_ = wbase + (st + 1) * 256;
        const f32x4* xp_ = xbase + ((st + 1) % 7) * 3;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][s4], b0[j][s4], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (s4 == 0) { a1[0] = wp_[0]; a1[1] = wp_[32]; }
          if (s4 == 1) { b1[0] = xp_[0]; }
          if (s4 == 2) { b1[1] = xp_[32]; }
          if (s4 == 3) { b1[2] = xp_[64]; }
          __builtin_amdgcn_sched_barrier(0);
        }
        const f32x4* wq_ = wbase + (st + 2) * 256;
        const f32x4* xq_ = xbase + ((st + 2) % 7) * 3;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][s4], b1[j][s4], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (s4 == 0) { a0[0] = wq_[0]; a0[1] = wq_[32]; }
          if (s4 == 1) { b0[0] = xq_[0]; }
          if (s4 == 2) { b0[1] = xq_[32]; }
          if (s4 == 3) { b0[2] = xq_[64]; }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        LOAD_AB(st + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        MMA(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        LOAD_AB(st + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        MMA(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (st < steps) { MMA(a0, b0); }
    if constexpr (VAR == 0 || VAR == 6 || VAR == 7) __syncthreads();
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(size_t)blockIdx.x * NT + tid] = s;
}

template <int VAR, int NT>
static void run(const char* name, int steps) {
  const int blocks = VAR == 5 ? 256 : 256 * (512 / NT);
  float* out; (void)hipMalloc(&out, (size_t)blocks * NT * 4);
  const int chunks = 1600 / steps * 8;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<VAR, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int it = 0; it < 4; ++it) {
    (void)hipEventRecord(e0);
    k<VAR, NT><<<blocks, NT, 65536>>>(out, chunks, steps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  const double flop = (double)blocks * (NT / 64) * chunks * steps * 24 * 4096.0;
  printf("var %d %-58s steps/chunk %2d: %.3f ms  %.1f TFLOP/s = %.3f of 157.3\n", VAR, name, steps, best, flop / best / 1e9, flop / best / 1e9 / 157.3);
  (void)hipFree(out);
}
int main() {
  for (int steps : {7, 11, 6}) {
    run<0, 512>("512 thr, barrier per chunk (shipped structure)", steps);
    run<1, 512>("512 thr, no barrier", steps);
    run<2, 512>("512 thr, no barrier, waves 4-7 half a step ahead", steps);
    run<3, 512>("512 thr, no barrier, reads spread over the step", steps);
    run<4, 512>("512 thr, no barrier, setprio 1 on waves 4-7", steps);
    run<5, 256>("256 thr = one wave per SIMD, no barrier", steps);
    run<6, 512>("512 thr, barrier per chunk, waves 4-7 half a step ahead", steps);
    run<7, 256>("2 x 256 thr per CU, barrier per chunk", steps);
  }
  return 0;
}
