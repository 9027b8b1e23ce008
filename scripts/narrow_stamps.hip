// Timeline of conv1d_narrow_kernel on a single-utterance decoder conv (C = 128, k = 7, 16 T' = 4240 frames):
// wall-clock stamps (100 MHz) at the phase boundaries of every workgroup's first unit, plus HIP-event time
// of the launch.  Checks the result against a CPU loop as well.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMBV_NARROW_STAMPS -Imb-istft-vits_amd/csrc -Iinclude scripts/narrow_stamps.hip -o /tmp/narrow_stamps
#include "../mb-istft-vits_amd/csrc/conv1d_narrow.hip"
#include <algorithm>
#include <cmath>
#include <random>
#include <vector>
using namespace mbv;
static size_t pack_idx(int tap, int ci, int m, int Cin, int Mpad) {
  return ((((size_t)tap * (Cin / 8) + ci / 8) * 2 + (ci & 1)) * Mpad + m) * 4 + ((ci & 7) >> 1);
}
int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 128, K = argc > 2 ? atoi(argv[2]) : 7, T = argc > 3 ? atoi(argv[3]) : 4240;
  const int B = argc > 4 ? atoi(argv[4]) : 1, iters = 20;
  const bool by_size = argc > 5 ? atoi(argv[5]) != 0 : true;      // 0: the default mode's geometry (rules on M / T only)
  const int Cin = argc > 6 ? atoi(argv[6]) : C;
  const int Mpad = (C + 127) / 128 * 128;
  std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> W((size_t)C * Cin * K), bias(C), x((size_t)B * Cin * T), wp((size_t)K * Cin * Mpad, 0.f), yv((size_t)B * C * T);
  for (auto& v : W) v = nd(rng) / std::sqrt((float)Cin * K);
  for (auto& v : bias) v = nd(rng) * 0.1f;
  for (auto& v : x) v = nd(rng);
  for (int m = 0; m < C; ++m) for (int ci = 0; ci < Cin; ++ci) for (int k = 0; k < K; ++k) wp[pack_idx(k, ci, m, Cin, Mpad)] = W[((size_t)m * Cin + ci) * K + k];
  float *d_x, *d_y, *d_w, *d_b, *d_ws;
  hipMalloc(&d_x, x.size() * 4); hipMalloc(&d_y, yv.size() * 4); hipMalloc(&d_w, wp.size() * 4); hipMalloc(&d_b, C * 4); hipMalloc(&d_ws, 512 * 8 * 8);
  hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_w, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_b, bias.data(), C * 4, hipMemcpyHostToDevice); hipMemset(d_ws, 0, 512 * 64);
  ConvArgs a{};
  a.x = d_x; a.x_bstride = (int64_t)Cin * T; a.Tin = T; a.x_rstride = T; a.Cin = Cin; a.w = d_w; a.bias = d_b; a.M = C; a.Mpad = Mpad; a.K = K; a.dil = 1;
  a.pad_left = (K - 1) / 2; a.in_slope = 0.1f; a.y = d_y; a.y_bstride = (int64_t)C * T; a.T = T; a.epi = EPI_STORE; a.B = B; a.ws = d_ws; a.out_scale = 1.f;
  if (!conv1d_narrow_supported(a)) { printf("not supported\n"); return 1; }
  // a scratch kernel between launches so that every timed launch starts from the state a pipeline leaves (x rewritten by another kernel)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ms(iters);
  for (int it = 0; it < iters; ++it) {
    hipMemcpyAsync(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice, 0);   // x arrives from elsewhere, as in the pipeline
    hipEventRecord(e0, 0);
    launch_conv1d_narrow(a, by_size, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[it], e0, e1);
  }
  std::sort(ms.begin(), ms.end());
  printf("M=%d K=%d T=%d B=%d: launch median %.1f us (min %.1f)\n", C, K, T, B, ms[iters / 2] * 1e3, ms[0] * 1e3);
  std::vector<unsigned long long> st(512 * 8);
  hipMemcpy(st.data(), d_ws, 512 * 64, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull; int nb = 0;
  for (int b = 0; b < 512; ++b) if (st[b * 8]) { t0 = std::min(t0, st[b * 8]); ++nb; }
  printf("%d workgroups; phase ends in us after the first workgroup's start: start | acc init | ring+sync | window staged | MFMA loop | stores issued\n", nb);
  double mean[6] = {0}, mx[6] = {0};
  for (int b = 0; b < 512; ++b) if (st[b * 8]) for (int i = 0; i < 6; ++i) { const double v = (st[b * 8 + i] - t0) * 0.01; mean[i] += v / nb; mx[i] = std::max(mx[i], v); }
  printf("mean "); for (int i = 0; i < 6; ++i) printf(" %7.2f", mean[i]); printf("\nmax  "); for (int i = 0; i < 6; ++i) printf(" %7.2f", mx[i]); printf("\n");
  printf("shader clock during the MFMA loop of workgroup 0: %.0f cycles in %.2f us = %.2f GHz\n", (double)(st[7] - st[6]), (st[4] - st[3]) * 0.01,
         (double)(st[7] - st[6]) / ((st[4] - st[3]) * 10.0));
  for (int b : {0, 1, 8, 64, nb - 1}) { printf("wg %3d", b); for (int i = 0; i < 6; ++i) printf(" %7.2f", (st[b * 8 + i] - t0) * 0.01); printf("\n"); }
  // parity
  std::vector<float>& y = yv;
  hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost);
  double err = 0;
  for (int m = 0; m < C; m += 17) for (int t = 0; t < T; t += 97) {
    double v = bias[m];
    for (int ci = 0; ci < Cin; ++ci) for (int k = 0; k < K; ++k) { const int ti = t + k - (K - 1) / 2; if (ti >= 0 && ti < T) { float xv = x[(size_t)ci * T + ti]; xv = xv > 0 ? xv : 0.1f * xv; v += (double)W[((size_t)m * Cin + ci) * K + k] * xv; } }
    err = std::max(err, std::fabs(v - y[(size_t)m * T + t]));
  }
  printf("max |err| vs CPU loop (utterance 0, sampled): %.2e %s\n", err, err < 1e-4 ? "OK" : "MISMATCH");
  return err < 1e-4 ? 0 : 1;
}
