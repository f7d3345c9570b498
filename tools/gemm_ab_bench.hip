// A/B harness for the M = N = 256 weight-gradient GEMM (not part of the product build): tn_gemm_big<16> (8 waves, each
// does everything, phase after phase) against tn_gemm_w4<16, PRE> (4 waves, split between the MFMAs), same operands,
// results compared bit for bit and against an fp64 sum for a sample of outputs.  -DFASTGRNN_DIAG_STAMPS adds
// tn_gemm_w4's per-segment cycle sums.  (Uniform random operands draw more power than a training step's: absolute
// times are 20-25 % above the stack's, the ratio is what to read.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_gemm_ab.bin tools/gemm_ab_bench.hip && tools/_gemm_ab.bin
#include "../kws_amd/csrc/kernels_gemm.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace fastgrnn;
static std::vector<float> host_rand(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / (float)RAND_MAX * 2.f - 1.f);
  return h;
}
template <typename K>
static float run(K kern, size_t R, int M, int N, const float* A, const float* B, size_t shift, float* part, float* C, int reps, int threads = 512) {
  int spw;
  const int nblk = M / 128, nch = tnb_chunks(R, nblk, &spw);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rep = 0; rep < reps; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(((nch + 7) / 8) * 8 * nblk), dim3(threads), 0, 0, R, spw, nblk, nch, A, M, B, B, shift, N, part);
    (void)hipEventRecord(e1);
    hipLaunchKernelGGL(tn_big_reduce, dim3((M * N + 63) / 64), dim3(1024), 0, 0, nch, nblk, N, (const float*)part, C, N);
    (void)hipEventSynchronize(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2 || reps < 3) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2] * 1e3f;
}
int main(int argc, char** argv) {
  const size_t R = argc > 1 ? (size_t)atol(argv[1]) : (size_t)99 * 4096;
  const int M = 256, N = 256;
  std::vector<float> hA = host_rand(R * M, 1.f), hB = host_rand(R * N, 1.f);
  float *A, *B, *C0, *C1, *part;
  (void)hipMalloc(&A, R * M * 4); (void)hipMalloc(&B, R * N * 4);
  (void)hipMemcpy(A, hA.data(), R * M * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hB.data(), R * N * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&C0, M * N * 4); (void)hipMalloc(&C1, M * N * 4);
  (void)hipMalloc(&part, tn_gemm_big_ws(R, M, N));
  for (int w = 0; w < 150; ++w) run(tn_gemm_w4<16, TNW4_PRE>, R, M, N, A, B, 0, part, C1, 1, 256);   // clocks up
  for (int round = 0; round < 3; ++round) {
    const float t0 = run(tn_gemm_big<16>, R, M, N, A, B, 0, part, C0, 8);
    const float t1 = run(tn_gemm_w4<16, TNW4_PRE>, R, M, N, A, B, 0, part, C1, 8, 256);
    printf("R=%zu  tn_gemm_big<16> %.1f us   tn_gemm_w4<16,%d> %.1f us\n", R, t0, TNW4_PRE, t1);
  }
#ifdef FASTGRNN_DIAG_STAMPS
  {
    unsigned long long h[8][8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sdiag), sizeof(h));
    const double per_wg = (double)((R + 31) / 32) / 128.0;
    const char* cn[4] = {"barrier wait", "fragment reads", "products + split", "requests"};
    for (int wv = 0; wv < 4; ++wv) {
      printf("   wave %d:", wv);
      for (int k = 0; k < 4; ++k) printf("  [%s] %.0f", cn[k], (double)h[wv][k] / per_wg);
      printf("  ticks per stage\n");
    }
  }
#endif
  std::vector<float> h0(M * N), h1(M * N);
  (void)hipMemcpy(h0.data(), C0, M * N * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(h1.data(), C1, M * N * 4, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (int i = 0; i < M * N; ++i) diff += std::memcmp(&h0[i], &h1[i], 4) != 0;
  double worst = 0, scale = 0;
  for (int s = 0; s < 64; ++s) {
    const int m = (s * 37) % M, nn = (s * 101 + 7) % N;
    double ref = 0;
    for (size_t r = 0; r < R; ++r) ref += (double)hA[r * M + m] * (double)hB[r * N + nn];
    worst = std::max(worst, std::fabs(ref - (double)h1[m * N + nn]));
    scale = std::max(scale, std::fabs(ref));
  }
  printf("elements differing between the two kernels: %zu of %d;  w4 vs fp64 on 64 samples: max abs err %.3e (max |ref| %.3e)\n",
         diff, M * N, worst, scale);
  // ragged tail + shifted B (h0 rows)
  {
    const size_t R2 = R - 37, shift = 512;   // (a multiple of the stage: tn_gemm_w4's contract)
    run(tn_gemm_big<16>, R2, M, N, A, B, shift, part, C0, 1);
    run(tn_gemm_w4<16, TNW4_PRE>, R2, M, N, A, B, shift, part, C1, 1, 256);
    (void)hipMemcpy(h0.data(), C0, M * N * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h1.data(), C1, M * N * 4, hipMemcpyDeviceToHost);
    size_t d2 = 0;
    for (int i = 0; i < M * N; ++i) d2 += std::memcmp(&h0[i], &h1[i], 4) != 0;
    printf("ragged R=%zu, shiftB=%zu: elements differing: %zu\n", R2, shift, d2);
  }
  return 0;
}
