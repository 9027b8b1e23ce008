// Timeline of conv1d_mfma_kernel (the decoder's 128 x 384 / 64 x 384 shapes) on one decoder conv of the bench batch:
// threads 0 and 256 of every workgroup stamp the 100 MHz wall clock at phase boundaries (MBV_CSTAMP in conv1d.hip);
// this prints where a workgroup's time goes: start-value / tile set-up, MFMA loops, commit + requests, barrier
// waits, gaps between barrier and the next loop, epilogue.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMBV_CONV_STAMPS -Imb-istft-vits_amd/csrc -Iinclude scripts/conv_stamps.hip -o /tmp/conv_stamps
// usage: conv_stamps [C=128] [K=7] [T=9056] [B=64] [dil=1] [epi: 0 store | 1 resid | 2 resid_acc]
#include "../mb-istft-vits_amd/csrc/conv1d.hip"
#include "../mb-istft-vits_amd/csrc/conv1d_narrow.hip"
#include <algorithm>
#include <cmath>
#include <map>
#include <random>
#include <vector>
using namespace mbv;
int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 128, K = argc > 2 ? atoi(argv[2]) : 7, T = argc > 3 ? atoi(argv[3]) : 9056;
  const int B = argc > 4 ? atoi(argv[4]) : 64, dil = argc > 5 ? atoi(argv[5]) : 1, epi = argc > 6 ? atoi(argv[6]) : 0;
  const int Mpad = (C + 127) / 128 * 128;
  const size_t nx = (size_t)B * C * T;
  float *d_x, *d_y, *d_r, *d_acc, *d_w, *d_b; unsigned long long* d_ws;
  const int NWG = 512;
  hipMalloc(&d_x, nx * 4); hipMalloc(&d_y, nx * 4); hipMalloc(&d_r, nx * 4); hipMalloc(&d_acc, nx * 4);
  hipMalloc(&d_w, (size_t)K * C * Mpad * 4); hipMalloc(&d_b, C * 4); hipMalloc(&d_ws, (size_t)NWG * 1024 * 8);
  {
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> h(1 << 22); for (auto& v : h) v = nd(rng);
    for (size_t o = 0; o < nx; o += h.size()) {
      const size_t n = std::min(h.size(), nx - o);
      hipMemcpy(d_x + o, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(d_r + o, h.data(), n * 4, hipMemcpyHostToDevice);
      hipMemcpy(d_acc + o, h.data(), n * 4, hipMemcpyHostToDevice);
    }
    std::vector<float> w((size_t)K * C * Mpad); for (auto& v : w) v = nd(rng) * 0.03f;
    hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice); hipMemset(d_b, 0, C * 4);
  }
  ConvArgs a{};
  a.x = d_x; a.x_bstride = (int64_t)C * T; a.Tin = T; a.x_rstride = T; a.Cin = C; a.w = d_w; a.bias = d_b; a.M = C; a.Mpad = Mpad; a.K = K; a.dil = dil;
  a.pad_left = (K - 1) * dil / 2; a.in_slope = 0.1f; a.y = d_y; a.y_bstride = (int64_t)C * T; a.T = T; a.epi = epi; a.B = B; a.out_scale = 1.f;
  a.ws = reinterpret_cast<float*>(d_ws);
  if (epi) { a.res = d_r; a.res_bstride = (int64_t)C * T; }
  if (epi == 2) { a.accum_in = d_acc; a.y = d_acc; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0, best = 1e9;
  for (int it = 0; it < 6; ++it) {
    hipMemsetAsync(d_ws, 0, (size_t)NWG * 2 * 512 * 8, 0);
    hipEventRecord(e0, 0); launch_conv1d(a, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
  }
  const double flop = 2.0 * C * C * K * (double)T * B;
  printf("C=%d K=%d dil=%d T=%d B=%d epi=%d: launch %.1f us (best %.1f) = %.1f TFLOP/s (%.3f of 157.3)\n", C, K, dil, T, B, epi, ms * 1e3, best * 1e3,
         flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 157.3e12);
  std::vector<unsigned long long> st((size_t)NWG * 2 * 512);
  hipMemcpy(st.data(), d_ws, st.size() * 8, hipMemcpyDeviceToHost);
  // phases: id pairs (from -> to)
  const char* names[] = {"", "tile set-up + start values (1->2)", "", "MFMA loop (3->4)", "commit + drain + next requests (4->5)", "barrier wait (5->6)",
                         "barrier -> next loop / epilogue (6->3|7)", "epilogue (7->8)", "epilogue end -> next tile start (8->1)", "set-up -> first loop (2->3)",
                         "  late: loop end -> window committed (4->9)", "  late: -> weight DMA drained (9->10)", "  late: -> next requests issued (10->5)",
                         "  early: barrier -> DMA issued + window committed (6->11)", "  early: -> next requests issued, loop starts (11->3)"};
  for (int smp = 0; smp < 2; ++smp) {
    std::map<int, double> sum; std::map<int, long> cnt; double span = 0; int nwg = 0; unsigned long long first = ~0ull, last = 0;
    for (int wg = 0; wg < NWG; ++wg) {
      const unsigned long long* s = &st[((size_t)wg * 2 + smp) * 512];
      const int n = (int)s[0];
      if (n < 2) continue;
      ++nwg;
      first = std::min(first, s[1] >> 4); last = std::max(last, s[n] >> 4);
      span += ((s[n] >> 4) - (s[1] >> 4)) * 0.01;
      for (int i = 1; i < n; ++i) {
        const int id0 = s[i] & 15, id1 = s[i + 1] & 15;
        const double dt = ((s[i + 1] >> 4) - (s[i] >> 4)) * 0.01;
        int ph = -1;
        if (id0 == 4 && id1 == 9) ph = 10; else if (id0 == 9 && id1 == 10) ph = 11; else if (id0 == 10 && id1 == 5) ph = 12;
        else if (id0 == 6 && id1 == 11) ph = 13; else if (id0 == 11 && id1 == 3) ph = 14;
        else if (id0 == 1 && id1 == 2) ph = 1; else if (id0 == 2 && id1 == 3) ph = 9; else if (id0 == 3 && id1 == 4) ph = 3;
        else if (id0 == 4 && id1 == 5) ph = 4; else if (id0 == 5 && id1 == 6) ph = 5; else if (id0 == 6) ph = 6;
        else if (id0 == 7 && id1 == 8) ph = 7; else if (id0 == 8 && id1 == 1) ph = 8;
        if (ph >= 0) { sum[ph] += dt; ++cnt[ph]; }
      }
    }
    if (!nwg) continue;
    {   // per tile of a workgroup (in the order it walks them): tile start -> epilogue issued, and the MFMA loops inside
      double tt[32] = {0}, tl[32] = {0}, t_first = 0; long tn[32] = {0};
      for (int wg = 0; wg < NWG; ++wg) {
        const unsigned long long* s = &st[((size_t)wg * 2 + smp) * 512];
        const int n = (int)s[0];
        if (n < 2) continue;
        int tile = -1; unsigned long long t1 = 0, t3 = 0;
        t_first += ((s[1] >> 4) - first) * 0.01;
        for (int i = 1; i <= n; ++i) {
          const int id = s[i] & 15; const unsigned long long t = s[i] >> 4;
          if (id == 1) { ++tile; t1 = t; }
          if (tile < 0 || tile >= 32) continue;
          if (id == 3) t3 = t;
          if (id == 4) tl[tile] += (t - t3) * 0.01;
          if (id == 8) { tt[tile] += (t - t1) * 0.01; ++tn[tile]; }
        }
      }
      printf("-- sampled thread %d, per tile in walking order (us: whole tile | its MFMA loops):", smp * 256);
      for (int k = 0; k < 32 && tn[k]; ++k) printf("  [%d] %.1f | %.1f", k, tt[k] / tn[k], tl[k] / tn[k]);
      printf("\n   first stamp of a workgroup after the launch's first stamp: mean %.2f us\n", t_first / nwg);
    }
    printf("-- sampled thread %d: %d workgroups, mean first->last stamp %.1f us, launch-wide first->last %.1f us\n", smp * 256, nwg, span / nwg, (last - first) * 0.01);
    double tot = 0;
    for (auto& kv : sum) tot += kv.second / nwg;
    for (auto& kv : sum)
      printf("   %-48s %8.2f us per workgroup (%5.1f %%), %5ld x %6.3f us\n", names[kv.first], kv.second / nwg, 100.0 * kv.second / nwg / tot, cnt[kv.first] / nwg,
             kv.second / cnt[kv.first]);
  }
  return 0;
}
