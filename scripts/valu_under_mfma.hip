// How fast does one wave issue plain vector / LDS / memory instructions while its SIMD partner streams
// v_mfma_f32_32x32x2_f32 back to back?  512-thread workgroup: one half of the waves (roles by wave >= 4 or < 4)
// runs a dense MFMA loop, the other half a chain-free block of N instructions of one kind, timed with s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/valu_under_mfma.hip -o scripts/valu_under_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// KIND 0: v_fma (independent), 1: ds_write_b128, 2: global_load_dword (then one wait), 3: v_cmp + v_cndmask
template <int KIND, int READS>
__global__ __launch_bounds__(512, 2) void k(float* out, const float* in, unsigned long long* cyc, int mfma_iters, int worker_hi, int prio, int mfma_on) {
  __shared__ f32x4 sm[2048];
  const int tid = threadIdx.x, wave = tid >> 6;
  const bool worker = worker_hi ? wave >= 4 : wave < 4;
  sm[tid] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const unsigned long long t_start = __builtin_readcyclecounter();
  if (!worker) {
    if (!mfma_on) return;
    f32x16 acc[6];
    for (int n = 0; n < 6; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = 1.f + tid, b = 2.f - tid;
    if constexpr (READS == 2) {
      float ar[8], br[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { ar[q] = a + q; br[q] = b - q; }
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int n = 0; n < 6; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[(u * 2 + n) & 7], br[(u + n * 3) & 7], acc[n], 0, 0, 0);
      }
    } else if constexpr (READS == 3) {
      const f32x4* base = sm + (tid & 63) + (wave & 3) * 64;
      f32x4 r0[5];
      float sink = 0.f;
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int q = 0; q < 5; ++q) r0[q] = base[q * 256 + (it & 3) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int n = 0; n < 6; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 5; ++q) sink += r0[q][0];
      }
      acc[0][0] += sink;
    } else if constexpr (READS == 4 || READS == 5 || READS == 6) {
      // 4: as 1, but every LDS result is copied (v_mov) before an MFMA reads it
      // 5: as 1 with ONE ds_read_b128 per step (its four values feed all 24 MFMAs)
      // 6: as 1, reads issued TWO steps ahead (three register sets)
      const f32x4* base = sm + (tid & 63) + (wave & 3) * 64;
      constexpr int NR = READS == 5 ? 1 : 5;
      f32x4 r0[5], r1[5], r2[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) { r0[q] = base[(q % NR) * 256]; r1[q] = base[(q % NR) * 256 + 64]; r2[q] = r0[q]; }
      for (int it = 0; it < mfma_iters; ++it) {
        f32x4 use[5];
        if constexpr (READS == 6) {
#pragma unroll
          for (int q = 0; q < NR; ++q) r2[q] = base[q * 256 + ((it + 2) & 3) * 64];
#pragma unroll
          for (int q = 0; q < 5; ++q) use[q] = r0[q];
        } else {
#pragma unroll
          for (int q = 0; q < NR; ++q) r1[q] = base[q * 256 + ((it + 1) & 3) * 64];
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            use[q] = r0[q % NR];
            if constexpr (READS == 4) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(use[q][0]), "+v"(use[q][1]), "+v"(use[q][2]), "+v"(use[q][3]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int n = 0; n < 6; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(use[n % 2][u], use[2 + n % 3][u], acc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (READS == 6) {
#pragma unroll
          for (int q = 0; q < 5; ++q) { r0[q] = r1[q]; r1[q] = r2[q]; }
        } else {
#pragma unroll
          for (int q = 0; q < NR; ++q) r0[q] = r1[q];
        }
      }
    } else if constexpr (READS == 1) {
      // the conv kernel's step: 5 ds_read_b128 prefetched one step ahead + 24 MFMAs
      const f32x4* base = sm + (tid & 63) + (wave & 3) * 64;
      f32x4 r0[5], r1[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) r0[q] = base[q * 256];
      for (int it = 0; it < mfma_iters; it += 2) {
#pragma unroll
        for (int q = 0; q < 5; ++q) r1[q] = base[q * 256 + ((it + 1) & 3) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int n = 0; n < 6; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0[n % 2][u], r0[2 + n % 3][u], acc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 5; ++q) r0[q] = base[q * 256 + ((it + 2) & 3) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int n = 0; n < 6; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(r1[n % 2][u], r1[2 + n % 3][u], acc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int n = 0; n < 6; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
    }
    float s = 0;
    for (int n = 0; n < 6; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = __builtin_readcyclecounter() - t_start;
    return;
  }
  // worker: let the partner get going, then time N instructions
  __builtin_amdgcn_s_sleep(100);
  if (prio) __builtin_amdgcn_s_setprio(3);
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = in[tid + i * 512];
  f32x4 w = {v[0], v[1], v[2], v[3]};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_readcyclecounter();
  constexpr int REP = 16;
  if constexpr (KIND == 0) {
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(w[0]));
  } else if constexpr (KIND == 1) {
#pragma unroll
    for (int r = 0; r < REP * 4; ++r) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(16384 + (tid & 255) * 16 + (r & 3) * 4096)), "v"(w) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else if constexpr (KIND == 2) {
#pragma unroll
    for (int r = 0; r < REP * 4; ++r) asm volatile("global_load_dword %0, %1, off" : "=v"(v[r & 15]) : "v"(in + tid + (r & 15) * 512) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
#pragma unroll
    for (int r = 0; r < REP; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cmp_lt_f32 vcc, 0, %0\n\ts_nop 1\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(v[i]) : "v"(w[1]) : "vcc");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (prio) __builtin_amdgcn_s_setprio(0);
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 512 + tid] = s;
  if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int KIND, int READS>
static void run(const char* what, int n_inst) {
  float *out, *in; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&in, 16 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
  (void)hipMemset(in, 0, 16 * 512 * 4);
  for (int mfma_on : {0, 1})
    for (int worker_hi : {0, 1})
      for (int prio : {0, 1}) {
        if (!mfma_on && (worker_hi || prio)) continue;
        (void)hipMemset(cyc, 0, 256 * 8 * 8);
        k<KIND, READS><<<256, 512>>>(out, in, cyc, 3000, worker_hi, prio, mfma_on);
        (void)hipDeviceSynchronize();
        unsigned long long h[256 * 8]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        double sum = 0, psum = 0; int n = 0, pn = 0;
        for (int i = 0; i < 256 * 8; ++i) {
          const bool is_worker = worker_hi ? (i & 7) >= 4 : (i & 7) < 4;
          if (!h[i]) continue;
          if (is_worker) { sum += h[i]; ++n; } else { psum += h[i]; ++pn; }
        }
        printf("%-34s partner %-12s worker = %-26s prio %d: %7.0f cycles for %d instructions = %6.1f per instruction", what,
               mfma_on ? "MFMA stream" : "absent", worker_hi ? "waves 4-7 (younger)" : "waves 0-3 (older)", prio * 3, sum / n, n_inst, sum / n / n_inst);
        if (pn) printf("   | partner: %.1f cycles per MFMA", psum / pn / (3000.0 * 24));
        printf("\n");
      }
}
int main() {
  run<0, 0>("v_fma_f32 (independent)", 256);
  run<3, 0>("v_cmp + s_nop + v_cndmask", 256 * 3);
  run<1, 0>("ds_write_b128 (+ final wait)", 64);
  run<2, 0>("global_load_dword (+ final wait)", 64);
  printf("---- partner = the conv step loop (5 ds_read_b128 prefetched + 24 MFMA per step)\n");
  run<0, 1>("v_fma_f32 (independent)", 256);
  run<1, 1>("ds_write_b128 (+ final wait)", 64);
  run<2, 1>("global_load_dword (+ final wait)", 64);
  printf("---- partner = conv step loop, every LDS result copied by v_mov before its MFMAs\n");
  run<0, 4>("v_fma_f32 (independent)", 256);
  run<2, 4>("global_load_dword (+ final wait)", 64);
  printf("---- partner = conv step loop with ONE ds_read_b128 per step\n");
  run<0, 5>("v_fma_f32 (independent)", 256);
  run<2, 5>("global_load_dword (+ final wait)", 64);
  printf("---- partner = conv step loop, reads issued two steps ahead\n");
  run<0, 6>("v_fma_f32 (independent)", 256);
  run<2, 6>("global_load_dword (+ final wait)", 64);
  printf("---- partner = MFMA only, source registers rotating over 8 + 8\n");
  run<0, 2>("v_fma_f32 (independent)", 256);
  run<2, 2>("global_load_dword (+ final wait)", 64);
  printf("---- partner = 5 ds_read_b128 per 24 MFMAs, MFMA operands constant (reads consumed by VALU adds)\n");
  run<0, 3>("v_fma_f32 (independent)", 256);
  run<2, 3>("global_load_dword (+ final wait)", 64);
  return 0;
}
