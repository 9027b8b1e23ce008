// Calibration: how fast can the conv epilogue's store pattern drain?  One 512-thread workgroup per
// CU writes a [128 x 384] fp32 tile of a [B][128][T] tensor per iteration, exactly like
// conv1d_mfma_kernel<2,3,*,4>'s epilogue, in four instruction shapes:
//   0: dword, a wave store = 2 rows x 128 B (the MFMA accumulator layout as is)
//   1: dwordx3 per lane (columns interleaved 3 l + j): a wave store = 2 rows x 384 B
//   2: dwordx4 per lane, a wave store = 1 row x 1 KB (layout not reachable from MFMA; reference)
//   3: like 0 but only every 4th CU active (per-CU vs chip-wide limit)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct f3 { float a, b, c; };
template <int MODE>
__global__ __launch_bounds__(512) void k(float* y, int T, int tiles_x, int total, float v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3, hl = lane >> 5, l31 = lane & 31;
  if (MODE == 3 && (blockIdx.x & 3)) return;
  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int tx = tile % tiles_x, b = tile / tiles_x, t0 = tx * 384;
    float* yb = y + (size_t)b * 128 * T;
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int t = t0 + wn * 96 + j * 32 + l31;
            if (t < T) yb[(size_t)row * T + t] = v + r;
          }
        }
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
          const int t = t0 + wn * 96 + 3 * l31;
          if (t + 2 < T) { f3 o{v + r, v, v - r}; *reinterpret_cast<f3*>(&yb[(size_t)row * T + t]) = o; }
        }
    } else {
      // 128 rows x 96 float4 per row = 12288 float4 / 512 threads = 24 per thread
#pragma unroll
      for (int u = 0; u < 24; ++u) {
        const int e = u * 512 + threadIdx.x, row = e / 96, c4 = e % 96;
        const int t = t0 + c4 * 4;
        if (t + 3 < T) { f32x4 o = {v, v + u, v, v}; *reinterpret_cast<f32x4*>(&yb[(size_t)row * T + t]) = o; }
      }
    }
  }
}
template <int MODE>
void run(const char* name, float* y, int B, int T) {
  const int tiles_x = (T + 383) / 384, total = tiles_x * B;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 512>>>(y, T, tiles_x, total, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) k<MODE><<<256, 512>>>(y, T, tiles_x, total, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  double bytes = (double)B * 128 * T * 4 * (MODE == 3 ? 0.25 : 1.0);
  printf("%-44s %.1f us  %.2f TB/s  (%.1f GB/s per active CU)\n", name, ms * 1e3, bytes / ms / 1e9,
         bytes / ms / 1e6 / (MODE == 3 ? 64 : 256));
}
int main() {
  const int B = 64, T = 9056;
  float* y; hipMalloc(&y, (size_t)B * 128 * T * 4);
  run<0>("dword, 2 rows x 128 B per wave store", y, B, T);
  run<1>("dwordx3, 2 rows x 384 B per wave store", y, B, T);
  run<2>("dwordx4, 1 row x 1 KB per wave store", y, B, T);
  run<3>("dword, 1 of 4 CUs active", y, B, T);
  hipFree(y);
  return 0;
}
