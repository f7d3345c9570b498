// Diagnostic: in-kernel clock and v_mfma_f32_16x16x4_f32 issue rate at a given launch length.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/clockprobe tools/clockprobe.hip && /tmp/clockprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(int iters, float* out, unsigned long long* stamps) {
  f32x4 acc[NACC];
  for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  float av = threadIdx.x * 1e-3f, bv = 1.0f + threadIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 32; ++k)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[a], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run(int iters, int blocks, const char* tag) {
  float* out; unsigned long long* st;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, iters, out, st);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
  double cyc = (double)h[0], real = (double)h[1];
  double nm = (double)iters * 32 * NACC;
  printf("%-10s NACC=%d blocks=%d iters=%d: %.1f us, in-kernel clock %.2f GHz, %.1f cycles/MFMA, %.1f ns/MFMA\n", tag, NACC,
         blocks, iters, ms * 1e3, cyc / real * 0.1, cyc / nm, ms * 1e6 / nm);
  hipFree(out); hipFree(st);
}

int main() {
  run<2>(40, 256, "250us");      // ~2560 MFMA/wave
  run<2>(100, 256, "short");
  run<2>(2000, 256, "long");
  run<1>(100, 256, "1acc");
  run<4>(100, 256, "4acc");
  run<2>(100, 512, "2wg/cu");
  run<2>(100, 64, "64 CUs");
  return 0;
}
