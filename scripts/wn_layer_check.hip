// Stand-alone check of the fused WN layer kernel (wn_fused.hip) against a plain CPU loop.
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -I../mb-istft-vits_amd/csrc scripts/wn_layer_check.hip -o scripts/wn_layer_check
#include "../mb-istft-vits_amd/csrc/wn_fused.hip"
#include <cmath>
#include <random>
#include <vector>
using namespace mbv;
static constexpr double kTol = 1e-4;      // fp32 sums of <= 960 products of O(1) values, fast-math gate
static size_t pack_idx(int tap, int ci, int m, int Cin, int Mpad) {
  return ((((size_t)tap * (Cin / 8) + ci / 8) * 2 + (ci & 1)) * Mpad + m) * 4 + ((ci & 7) >> 1);
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 3, H = argc > 2 ? atoi(argv[2]) : 96, T = argc > 3 ? atoi(argv[3]) : 45;
  const int last = argc > 4 ? atoi(argv[4]) : 0;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<int> lens(B);
  for (int b = 0; b < B; ++b) lens[b] = b == 0 ? T : 1 + (int)(rng() % T);
  for (int b = 0; b < B && 5 + b < argc; ++b) lens[b] = atoi(argv[5 + b]);      // explicit lengths
  const int Mr = last ? H : 2 * H, Mg_pad = (2 * H + 127) / 128 * 128, Mr_pad = (Mr + 127) / 128 * 128;
  std::vector<float> Wg((size_t)2 * H * H * 5), bg(2 * H), Wr((size_t)Mr * H), br(Mr), h((size_t)B * H * T), skip0((size_t)B * H * T);
  for (auto& v : Wg) v = nd(rng) / std::sqrt(5.f * H);
  for (auto& v : Wr) v = nd(rng) / std::sqrt((float)H);
  for (auto& v : bg) v = nd(rng) * 0.1f;
  for (auto& v : br) v = nd(rng) * 0.1f;
  for (auto& v : h) v = nd(rng);
  for (auto& v : skip0) v = nd(rng);
  // CPU reference
  std::vector<float> hout_ref(h.size(), 0.f), skip_ref = skip0, acts((size_t)H * T);
  for (int b = 0; b < B; ++b) {
    const int len = lens[b];
    for (int c = 0; c < H; ++c)
      for (int t = 0; t < len; ++t) {
        double at = bg[c], as = bg[H + c];
        for (int ci = 0; ci < H; ++ci)
          for (int k = 0; k < 5; ++k) {
            const int ti = t + k - 2;
            if (ti < 0 || ti >= len) continue;
            const float x = h[((size_t)b * H + ci) * T + ti];
            at += (double)Wg[((size_t)c * H + ci) * 5 + k] * x;
            as += (double)Wg[((size_t)(H + c) * H + ci) * 5 + k] * x;
          }
        acts[(size_t)c * T + t] = (float)(std::tanh(at) / (1.0 + std::exp(-as)));
      }
    for (int r = 0; r < Mr; ++r)
      for (int t = 0; t < len; ++t) {
        double v = br[r];
        for (int c = 0; c < H; ++c) v += (double)Wr[(size_t)r * H + c] * acts[(size_t)c * T + t];
        if (!last && r < H) hout_ref[((size_t)b * H + r) * T + t] = h[((size_t)b * H + r) * T + t] + (float)v;
        else skip_ref[((size_t)b * H + (last ? r : r - H)) * T + t] += (float)v;
      }
  }
  // pack
  std::vector<float> wgp((size_t)5 * H * Mg_pad, 0.f), wrp((size_t)H * Mr_pad, 0.f);
  for (int r = 0; r < 2 * H; ++r) {
    const int tile = r / 32, rho = r % 32;
    const int ch = tile * 16 + (rho & 7) + 8 * (rho >> 4);
    const int src = (rho & 8) ? H + ch : ch;
    for (int ci = 0; ci < H; ++ci)
      for (int k = 0; k < 5; ++k) wgp[pack_idx(k, ci, r, H, Mg_pad)] = Wg[((size_t)src * H + ci) * 5 + k];
  }
  for (int r = 0; r < Mr; ++r)
    for (int ci = 0; ci < H; ++ci) {
      const int src = 8 * (ci / 8) + 4 * (ci & 1) + ((ci & 7) >> 1);
      wrp[pack_idx(0, ci, r, H, Mr_pad)] = Wr[(size_t)r * H + src];
    }
  float *d_h, *d_ho, *d_skip, *d_wg, *d_bg, *d_wr, *d_br; int *d_lens, *d_us;
  hipMalloc(&d_h, h.size() * 4); hipMalloc(&d_ho, h.size() * 4); hipMalloc(&d_skip, h.size() * 4);
  hipMalloc(&d_wg, wgp.size() * 4); hipMalloc(&d_bg, bg.size() * 4); hipMalloc(&d_wr, wrp.size() * 4); hipMalloc(&d_br, br.size() * 4);
  hipMalloc(&d_lens, B * 4); hipMalloc(&d_us, wn_units_ints(B, T) * 4);
  hipMemcpy(d_h, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(d_ho, 0, h.size() * 4);
  hipMemcpy(d_skip, skip0.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_wg, wgp.data(), wgp.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_bg, bg.data(), bg.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_wr, wrp.data(), wrp.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_br, br.data(), br.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_lens, lens.data(), B * 4, hipMemcpyHostToDevice);
  launch_wn_units(d_lens, B, T, d_us, d_us + B + 1, 0);
  WnLayerArgs a{};
  a.h_in = d_h; a.h_out = d_ho; a.skip = d_skip; a.lens = d_lens; a.ustart = d_us; a.hmap = d_us + B + 1; a.wg = d_wg; a.bg = d_bg; a.wr = d_wr; a.br = d_br;
  a.B = B; a.H = H; a.T = T; a.Mg_pad = Mg_pad; a.Mr = Mr; a.Mr_pad = Mr_pad; a.last = last; a.skip_accum = 1;
  launch_wn_layer(a, 0);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<float> ho(h.size()), sk(h.size());
  hipMemcpy(ho.data(), d_ho, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(sk.data(), d_skip, h.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < B; ++b) {
    double eh = 0, es = 0; int first = -1;
    for (int c = 0; c < H; ++c)
      for (int t = 0; t < lens[b]; ++t) {
        const size_t i = ((size_t)b * H + c) * T + t;
        const double dh = last ? 0.0 : std::fabs(ho[i] - hout_ref[i]), ds = std::fabs(sk[i] - skip_ref[i]);
        if ((dh > kTol || ds > kTol) && first < 0) first = t * 1000 + c;
        eh = std::max(eh, dh); es = std::max(es, ds);
      }
    printf("utt %d len %d: max |dh| %.2e max |dskip| %.2e first bad (t,c)=(%d,%d)\n", b, lens[b], eh, es, first / 1000, first % 1000);
    if (eh > kTol || es > kTol) {
      printf("   bad frames (skip):");
      for (int t = 0; t < lens[b]; ++t) {
        double m = 0;
        for (int c = 0; c < H; ++c) m = std::max(m, (double)std::fabs(sk[((size_t)b * H + c) * T + t] - skip_ref[((size_t)b * H + c) * T + t]));
        if (m > kTol) printf(" %d", t);
      }
      printf("\n   bad channels (skip):");
      for (int c = 0; c < H; ++c) {
        double m = 0;
        for (int t = 0; t < lens[b]; ++t) m = std::max(m, (double)std::fabs(sk[((size_t)b * H + c) * T + t] - skip_ref[((size_t)b * H + c) * T + t]));
        if (m > kTol) printf(" %d", c);
      }
      printf("\n");
    }
    bad += eh > kTol || es > kTol;
    // frames past the utterance's last 16-frame half-unit belong to nobody: the kernel must not touch them
    int stray = 0;
    for (int c = 0; c < H; ++c)
      for (int t = (lens[b] + 15) / 16 * 16; t < T; ++t) {
        const size_t i = ((size_t)b * H + c) * T + t;
        stray += sk[i] != skip0[i] || ho[i] != 0.f;
      }
    if (stray) printf("   %d stray stores past the last half-unit\n", stray);
    bad += stray != 0;
  }
  printf(bad ? "MISMATCH\n" : "OK\n");
  return bad != 0;
}
