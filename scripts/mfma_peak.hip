// Calibration: achievable fp32 MFMA rate (v_mfma_f32_32x32x2_f32) on this device,
// operands in registers, NACC independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  __shared__ float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = a0 * i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (LDS) { a = sm[(threadIdx.x + it * 64) & 4095]; b = sm[(threadIdx.x * 2 + it * 32) & 4095]; }
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, bool LDS>
void run(const char* name, int blocks) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, LDS><<<blocks, 256>>>(out, 100, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, LDS><<<blocks, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * NACC * 4096.0;
  printf("%-28s blocks=%5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4, false>("4 acc, regs, 1 wave/SIMD", 256);
  run<4, false>("4 acc, regs, 2 waves/SIMD", 512);
  run<4, false>("4 acc, regs, 3 waves/SIMD", 768);
  run<8, false>("8 acc, regs, 2 waves/SIMD", 512);
  run<4, true>("4 acc, +2 ds_read/4 mfma, 2w", 512);
  run<4, true>("4 acc, +2 ds_read/4 mfma, 3w", 768);
  return 0;
}
