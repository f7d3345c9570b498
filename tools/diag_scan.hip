// Diagnostic ablation harness for the MFMA scan kernels (not part of the product build).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/diag_scan tools/diag_scan.hip && /tmp/diag_scan
// Includes the kernel TU directly so that the anonymous-namespace templates can be
// instantiated with DIAG ablation bits.  Ablated variants produce wrong results; only
// their run time is read.
#define FASTGRNN_DIAG_STAMPS 1
#include "../kws_amd/csrc/kernels_mfma.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace fastgrnn;

static float* dev_rand(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX * 2.f - 1.f);
  float* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}

template <int DIAG>
void run_fwd(const char* tag, int T, int B, bool gates) {
  constexpr int H = 128, F = 32;
  static float *x, *h0, *w, *u, *bz, *bh, *zeta, *nu, *hs, *zs, *cs;
  static bool init = false;
  if (!init) {
    x = dev_rand((size_t)T * B * F, 1.f); h0 = dev_rand((size_t)B * H, 0.f); w = dev_rand(H * F, 0.17f);
    u = dev_rand(H * H, 0.17f); bz = dev_rand(H, 1.f); bh = dev_rand(H, 1.f); zeta = dev_rand(1, 1.f); nu = dev_rand(1, 1.f);
    (void)hipMalloc(&hs, (size_t)T * B * H * 4); (void)hipMalloc(&zs, (size_t)T * B * H * 4); (void)hipMalloc(&cs, (size_t)T * B * H * 4);
    init = true;
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rep = 0; rep < 12; ++rep) {
    (void)hipEventRecord(e0);
    if (gates) hipLaunchKernelGGL((fwd_scan_mfma<H, F, 0, true, false, DIAG>), dim3((B + 15) / 16), dim3(256), 0, 0, T, B, x, h0, w, u, bz, bh, zeta, nu, hs, zs, cs);
    else hipLaunchKernelGGL((fwd_scan_mfma<H, F, 0, false, false, DIAG>), dim3((B + 15) / 16), dim3(256), 0, 0, T, B, x, h0, w, u, bz, bh, zeta, nu, hs, nullptr, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("fwd %-34s gates=%d: %.1f us  (%.2f us/step)\n", tag, (int)gates, ts[ts.size() / 2] * 1e3, ts[ts.size() / 2] * 1e3 / T);
  if (DIAG & 16) {
    unsigned long long h[4][16];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_diag), sizeof(h));
    const char* names[5] = {"x req + LDS read + W.x MFMA", "stores(t-1) + chain MFMA", "epilogue + LDS write", "barrier wait", "-"};
    for (int wv = 0; wv < 4; ++wv) {
      unsigned long long tot = 0;
      for (int k = 0; k < 5; ++k) tot += h[wv][k];
      printf("   wave %d: total %.0f cycles/step:", wv, (double)tot / T);
      for (int k = 0; k < 5; ++k) printf("  [%s] %.0f", names[k], (double)h[wv][k] / T);
      printf("\n");
    }
  }
}

template <int DIAG>
void run_bwd(const char* tag, int T, int B) {
  constexpr int H = 128, F = 32;
  static float *x, *h0, *w, *u, *zeta, *nu, *hs, *zs, *cs, *ghs, *dx, *dh0, *part;
  static bool init = false;
  if (!init) {
    x = dev_rand((size_t)T * B * F, 1.f); h0 = dev_rand((size_t)B * H, 0.f); w = dev_rand(H * F, 0.17f);
    u = dev_rand(H * H, 0.17f); zeta = dev_rand(1, 1.f); nu = dev_rand(1, 1.f);
    hs = dev_rand((size_t)T * B * H, 1.f); zs = dev_rand((size_t)T * B * H, 0.5f); cs = dev_rand((size_t)T * B * H, 0.9f);
    ghs = dev_rand((size_t)T * B * H, 1.f);
    (void)hipMalloc(&dx, (size_t)T * B * F * 4); (void)hipMalloc(&dh0, (size_t)B * H * 4);
    (void)hipMalloc(&part, (size_t)((B + 15) / 16) * slab_stride(H, F) * 4);
    init = true;
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rep = 0; rep < 12; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((bwd_scan_mfma<H, F, 0, false, DIAG>), dim3((B + 15) / 16), dim3(256), 0, 0, T, B, ghs, x, hs, zs, cs,
                       h0, w, u, zeta, nu, dx, dh0, part);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("bwd %-34s: %.1f us  (%.2f us/step)\n", tag, ts[ts.size() / 2] * 1e3, ts[ts.size() / 2] * 1e3 / T);
  if (DIAG & 16) {
    unsigned long long h[4][16];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_diag), sizeof(h));
    const char* names[5] = {"LDS reads + d_x MFMA", "requests + chain MFMA", "4 x (dW/dU MFMA || EW chunk)", "LDS publish", "barrier wait"};
    for (int wv = 0; wv < 4; ++wv) {
      unsigned long long tot = 0;
      for (int k = 0; k < 5; ++k) tot += h[wv][k];
      printf("   wave %d: total %.0f cycles/step:", wv, (double)tot / T);
      for (int k = 0; k < 5; ++k) printf("  [%s] %.0f", names[k], (double)h[wv][k] / T);
      printf("\n");
    }
  }
}

int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 4096, T = 99;
  run_bwd<16>("stamped", T, B);
  run_bwd<0>("production (pins + spread)", T, B);
  run_bwd<32>("no pins", T, B);
  run_bwd<64>("pins, no load spreading", T, B);
  run_bwd<128>("pins, no MFMA/VALU interleave", T, B);
  run_fwd<16>("stamped", T, B, true);
  run_fwd<16>("stamped", T, B, false);
  run_fwd<0>("production (pins + spread)", T, B, true);
  run_fwd<0>("production (pins + spread)", T, B, false);
  run_fwd<32>("no pins", T, B, true);
  run_fwd<64>("pins, no store spreading", T, B, true);
  run_fwd<1>("no stores", T, B, true);
  run_fwd<4>("cheap epilogue", T, B, false);
  return 0;
}
