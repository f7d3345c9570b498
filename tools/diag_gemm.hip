// Diagnostic harness for tn_gemm_big (not part of the product build): per-segment cycle sums of block 7's waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/diag_gemm tools/diag_gemm.hip && /tmp/diag_gemm
#define FASTGRNN_DIAG_STAMPS 1
#include "../kws_amd/csrc/kernels_gemm.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace fastgrnn;
static float* dev_rand(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / (float)RAND_MAX * 2.f - 1.f);
  float* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}
int main() {
  const size_t R = 99 * 4096;
  const int M = 256, N = 256;
  float *A = dev_rand(R * M, 1.f), *B = dev_rand(R * N, 1.f), *C, *part;
  (void)hipMalloc(&C, M * N * 4);
  (void)hipMalloc(&part, tn_gemm_big_ws(R, M, N));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rep = 0; rep < 10; ++rep) {
    (void)hipEventRecord(e0);
    tn_gemm_big_run(R, M, N, A, M, B, B, 0, N, part, C, N, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  const size_t nstages = (R + 31) / 32;
  const double per_wg = (double)nstages / 128.0;          // stages per workgroup at M = 256 (two column blocks)
  printf("tn_gemm_big M=N=256: %.1f us (with stamps), %.0f stages per workgroup\n", ts[ts.size() / 2] * 1e3, per_wg);
  unsigned long long h[8][8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sdiag), sizeof(h));
  const char* names[5] = {"top (load issue of the previous stage)", "A + B fragment reads and waits", "products + drain", "publication", "barrier"};
  for (int wv = 0; wv < 8; wv += 4) {
    unsigned long long tot = 0;
    for (int k = 0; k < 5; ++k) tot += h[wv][k];
    printf("   wave %d: %.0f cycles/stage:", wv, (double)tot / per_wg);
    for (int k = 0; k < 5; ++k) printf("  [%s] %.0f", names[k], (double)h[wv][k] / per_wg);
    printf("\n");
  }
  return 0;
}
