// Diagnostic harness for the H = 256 scans (not part of the product build): per-segment cycle sums of block 7.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/diag_h256 tools/diag_h256.hip && /tmp/diag_h256
#define FASTGRNN_DIAG_STAMPS 1
#include "../kws_amd/csrc/kernels_h256.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace fastgrnn;
namespace fastgrnn {   // (the GEMM launchers the file's host code refers to; not used here)
size_t tn_gemm_big_ws(size_t, int, int) { return 0; }
int tn_gemm_big_run(size_t, int, int, const float*, int, const float*, const float*, size_t, int, float*, float*, int, hipStream_t) { return 0; }
int rows_gemm(size_t, int, int, bool, const void*, const float*, void*, bool, bool, hipStream_t) { return 0; }
}
static float* dev_rand(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX * 2.f - 1.f);
  float* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}
template <typename K, typename... A>
void timeit(const char* name, int T, int nwg, int threads, const char* const* names, int nseg, K kern, A... args) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rep = 0; rep < 10; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), 0, 0, args...);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("%s: %.1f us (%.2f us/step)\n", name, ts[ts.size() / 2] * 1e3, ts[ts.size() / 2] * 1e3 / T);
  unsigned long long h[8][8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sdiag), sizeof(h));
  for (int wv = 0; wv < 8; wv += 4) {
    unsigned long long tot = 0;
    for (int k = 0; k < nseg; ++k) tot += h[wv][k];
    printf("   wave %d: %.0f cycles/step:", wv, (double)tot / T);
    for (int k = 0; k < nseg; ++k) printf("  [%s] %.0f", names[k], (double)h[wv][k] / T);
    printf("\n");
  }
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, T = 99;
  constexpr int H = 256, F = 32;
  float *x = dev_rand((size_t)T * B * F, 1.f), *h0 = dev_rand((size_t)B * H, 0.f), *w = dev_rand(H * F, 0.17f);
  float *u = dev_rand(H * H, 0.1f), *zeta = dev_rand(1, 1.f), *nu = dev_rand(1, 1.f), *bz = dev_rand(H, 1.f), *bh = dev_rand(H, 1.f);
  float *hs, *zs, *cs, *dh0, *dpre, *part;
  (void)hipMalloc(&hs, (size_t)T * B * H * 4); (void)hipMalloc(&zs, (size_t)T * B * H * 4); (void)hipMalloc(&cs, (size_t)T * B * H * 4);
  (void)hipMalloc(&dh0, (size_t)B * H * 4); (void)hipMalloc(&dpre, (size_t)T * B * H * 4); (void)hipMalloc(&part, (size_t)(B / 16 + 1) * 576 * 4);
  unsigned* flags; (void)hipMalloc(&flags, (size_t)(B / 16 + 1) * 4);
  float* ghs = dev_rand((size_t)T * B * H, 1.f);
  const int nwg = (B + 15) / 16;
  const char* fn[6] = {"top", "W batch", "state batches", "epilogue", "publish", "barrier"};
  timeit("fwd_scan_h256 fp16 path, AUX=2", T, nwg, 512, fn, 6, fwd_scan_h256<0, 2, false, 1>, T, B, (const float*)x, (const float*)h0,
         (const float*)w, (const float*)u, (const float*)bz, (const float*)bh, (const float*)zeta, (const float*)nu, hs, zs, cs, flags);
  timeit("fwd_scan_h256 bf16 path, AUX=2", T, nwg, 512, fn, 6, fwd_scan_h256<0, 2, false, 0>, T, B, (const float*)x, (const float*)h0,
         (const float*)w, (const float*)u, (const float*)bz, (const float*)bh, (const float*)zeta, (const float*)nu, hs, zs, cs, flags);
  const char* bn[6] = {"chain (prev step) + glue", "EW + scale", "barrier A", "split + publish", "barrier B", "tail"};
  timeit("bwd_scan_h256 PREACT", T, nwg, 512, bn, 6, bwd_scan_h256<0, true, false>, T, B, 0, (const float*)ghs, (const float*)hs,
         (const float*)zs, (const float*)cs, (const float*)h0, (const float*)u, (const float*)bz, (const float*)bh, (const float*)zeta,
         (const float*)nu, dh0, dpre, part);
  return 0;
}
