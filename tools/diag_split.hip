// Diagnostic harness for the split-precision scan kernels (not part of the product build).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/diag_split tools/diag_split.hip && /tmp/diag_split
#define FASTGRNN_DIAG_STAMPS 1
#include "../kws_amd/csrc/kernels_split.hip"
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
template <bool PREACT>
void run_bwd_w8(int T, int B) {
  constexpr int H = 128, F = 32;
  float *x = dev_rand((size_t)T * B * F, 1.f), *h0 = dev_rand((size_t)B * H, 0.f), *w = dev_rand(H * F, 0.17f);
  float *u = dev_rand(H * H, 0.17f), *zeta = dev_rand(1, 1.f), *nu = dev_rand(1, 1.f), *bz = dev_rand(H, 1.f), *bh = dev_rand(H, 1.f);
  float *hs = dev_rand((size_t)T * B * H, 1.f), *a0 = dev_rand((size_t)T * B * H, 0.5f), *a1 = dev_rand((size_t)T * B * H, 0.9f);
  float *ghs = dev_rand((size_t)T * B * H, 1.f), *dx, *dh0, *part;
  (void)hipMalloc(&dx, (size_t)T * B * F * 4); (void)hipMalloc(&dh0, (size_t)B * H * 4);
  (void)hipMalloc(&part, (size_t)((B + 15) / 16) * SLAB * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rep = 0; rep < 10; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((bwd_scan_split_w8<0, PREACT, false>), dim3((B + 15) / 16), dim3(512), 0, 0, T, B, B, 1, 0, ghs, x, hs, a0, a1,
                       h0, w, u, bz, bh, zeta, nu, dx, dh0, part);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("bwd_scan_split_w8 PREACT=%d: %.1f us (%.2f us/step)\n", (int)PREACT, ts[ts.size() / 2] * 1e3, ts[ts.size() / 2] * 1e3 / T);
  unsigned long long h[8][8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sdiag), sizeof(h));
  const char* names[5] = {"(glue)", "dB reads + chain + ew_pre", "ew_post + publish", "weight_grads (every other iteration)", "barrier"};
  for (int wv = 0; wv < 8; wv += 4) {
    unsigned long long tot = 0;
    for (int k = 0; k < 5; ++k) tot += h[wv][k];
    printf("   wave %d: %.0f cycles/step:", wv, (double)tot / T);
    for (int k = 0; k < 5; ++k) printf("  [%s] %.0f", names[k], (double)h[wv][k] / T);
    printf("\n");
  }
}
template <int AUX>
void run_fwd(int T, int B) {
  constexpr int H = 128, F = 32;
  float *x = dev_rand((size_t)T * B * F, 1.f), *h0 = dev_rand((size_t)B * H, 0.f), *w = dev_rand(H * F, 0.17f);
  float *u = dev_rand(H * H, 0.17f), *zeta = dev_rand(1, 1.f), *nu = dev_rand(1, 1.f), *bz = dev_rand(H, 1.f), *bh = dev_rand(H, 1.f);
  float *hs, *zs, *cs;
  (void)hipMalloc(&hs, (size_t)T * B * H * 4); (void)hipMalloc(&zs, (size_t)T * B * H * 4); (void)hipMalloc(&cs, (size_t)T * B * H * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rep = 0; rep < 10; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((fwd_scan_split<0, AUX, false>), dim3((B + 15) / 16), dim3(256), 0, 0, T, B, B, 1, x, h0, w, u, bz, bh,
                       zeta, nu, hs, zs, cs);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("fwd_scan_split AUX=%d: %.1f us (%.2f us/step)\n", AUX, ts[ts.size() / 2] * 1e3, ts[ts.size() / 2] * 1e3 / T);
  unsigned long long h[8][8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sdiag), sizeof(h));
  const char* names[5] = {"x req + LDS reads + x split", "tile0 chain (+stores)", "tile1 chain || epilogue0", "epilogue1 + h split + LDS write", "barrier"};
  for (int wv = 0; wv < 4; wv += 3) {
    unsigned long long tot = 0;
    for (int k = 0; k < 5; ++k) tot += h[wv][k];
    printf("   wave %d: %.0f cycles/step:", wv, (double)tot / T);
    for (int k = 0; k < 5; ++k) printf("  [%s] %.0f", names[k], (double)h[wv][k] / T);
    printf("\n");
  }
}
template <int AUX>
void run_fwd_w8(int T, int B) {
  constexpr int H = 128, F = 32;
  float *x = dev_rand((size_t)T * B * F, 1.f), *h0 = dev_rand((size_t)B * H, 0.f), *w = dev_rand(H * F, 0.17f);
  float *u = dev_rand(H * H, 0.17f), *zeta = dev_rand(1, 1.f), *nu = dev_rand(1, 1.f), *bz = dev_rand(H, 1.f), *bh = dev_rand(H, 1.f);
  float *hs, *zs, *cs;
  (void)hipMalloc(&hs, (size_t)T * B * H * 4); (void)hipMalloc(&zs, (size_t)T * B * H * 4); (void)hipMalloc(&cs, (size_t)T * B * H * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rep = 0; rep < 10; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((fwd_scan_split_w8<0, AUX, false>), dim3((B + 15) / 16), dim3(512), 0, 0, T, B, B, 1, 0, x, h0, w, u, bz, bh,
                       zeta, nu, hs, zs, cs);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("fwd_scan_split_w8 AUX=%d: %.1f us (%.2f us/step)\n", AUX, ts[ts.size() / 2] * 1e3, ts[ts.size() / 2] * 1e3 / T);
  unsigned long long h[8][8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sdiag), sizeof(h));
  const char* names[5] = {"(glue)", "frag reads + stores + x plane + landing wait", "chain (30 MFMAs)", "epilogue + h planes", "barrier"};
  for (int wv = 0; wv < 8; wv += 4) {
    unsigned long long tot = 0;
    for (int k = 0; k < 5; ++k) tot += h[wv][k];
    printf("   wave %d: %.0f cycles/step:", wv, (double)tot / T);
    for (int k = 0; k < 5; ++k) printf("  [%s] %.0f", names[k], (double)h[wv][k] / T);
    printf("\n");
  }
}
int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 4096;
  run_fwd<0>(99, B);
  run_fwd<2>(99, B);
  run_fwd_w8<2>(99, B);
  run_fwd_w8<0>(99, B);
  run_fwd<1>(99, B);
  run_bwd_w8<true>(99, B);
  run_bwd_w8<false>(99, B);
  return 0;
}
