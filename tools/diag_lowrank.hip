// Diagnostic harness for the low-rank backward scan (not part of the product build): per-segment cycle sums of block 7.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/_diag_lowrank.bin tools/diag_lowrank.hip && tools/_diag_lowrank.bin
#define FASTGRNN_DIAG_STAMPS 1
#include "../kws_amd/csrc/kernels_lowrank.hip"
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
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, T = 99;
  constexpr int H = 256, F = 32, R = 16;
  float *x = dev_rand((size_t)T * B * F, 1.f), *h0 = dev_rand((size_t)B * H, 0.f);
  float *w1 = dev_rand(R * F, 0.17f), *w2 = dev_rand(H * R, 0.17f), *u1 = dev_rand(R * H, 0.17f), *u2 = dev_rand(H * R, 0.17f);
  float *zeta = dev_rand(1, 1.f), *nu = dev_rand(1, 1.f), *bz = dev_rand(H, 1.f), *bh = dev_rand(H, 1.f);
  float *hs = dev_rand((size_t)T * B * H, 1.f), *pre = dev_rand((size_t)T * B * H, 1.f), *ms = dev_rand((size_t)T * B * 32, 1.f);
  float *ghs = dev_rand((size_t)T * B * H, 1.f);
  float *dx, *dh0, *sink, *part;
  const int nwg = (B + 15) / 16;
  (void)hipMalloc(&dx, (size_t)T * B * F * 4); (void)hipMalloc(&dh0, (size_t)B * H * 4); (void)hipMalloc(&sink, 16 * 32 * 4);
  (void)hipMalloc(&part, (size_t)nwg * SLAB_LR * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rep = 0; rep < 10; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((bwd_scan_lowrank_split<0, false, false>), dim3(nwg), dim3(512), 0, 0, T, B, B, 1, B, 1, R, R, 0,
                       (const float*)ghs, (const float*)x, (const float*)hs, (const float*)pre, (const float*)ms, (const float*)h0,
                       (const float*)w1, (const float*)w2, (const float*)u1, (const float*)u2, (const float*)bz, (const float*)bh,
                       (const float*)zeta, (const float*)nu, dx, dh0, sink, part);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms_; (void)hipEventElapsedTime(&ms_, e0, e1);
    if (rep >= 2) ts.push_back(ms_);
  }
  std::sort(ts.begin(), ts.end());
  printf("bwd_scan_lowrank_split (stamped): %.1f us (%.2f us/step)\n", ts[ts.size() / 2] * 1e3, ts[ts.size() / 2] * 1e3 / T);
  unsigned long long h[8][8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sdiag), sizeof(h));
  const char* names[8] = {"EW+completion", "images", "refill", "partial", "barrier", "sum+mB+d_h", "trailing issue", "-"};
  for (int wv = 0; wv < 8; ++wv) {
    unsigned long long tot = 0;
    for (int k = 0; k < 8; ++k) tot += h[wv][k];
    printf("   wave %d: %.0f cycles/step:", wv, (double)tot / T);
    for (int k = 0; k < 8; ++k) printf("  [%s] %.0f", names[k], (double)h[wv][k] / T);
    printf("\n");
  }
  return 0;
}
