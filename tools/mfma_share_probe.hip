// Diagnostic: two waves per SIMD sharing the bf16 matrix pipe on gfx950.
//   roleA / roleB per wave group (waves 0-3 / 4-7):  0 = idle, 1 = independent MFMAs (4 accumulators),
//   2 = one dependent MFMA chain, 3 = VALU only (v_fma), 4 = dependent chain + 3 v_fma per MFMA
// Reports cycles per MFMA (or per 3 v_fma) seen by wave 0 and wave 4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int ROLE>
__device__ __forceinline__ void body(int iters, float& sink) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  u32x4 a8 = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 1u}, b8 = a8;
  float v[6];
  for (int k = 0; k < 6; ++k) v[k] = threadIdx.x * 0.01f + k;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (ROLE == 1) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc[k & 3], 0, 0, 0);
      if (ROLE == 2 || ROLE == 4) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc[0], 0, 0, 0);
      if (ROLE == 3 || ROLE == 4) {
#pragma unroll
        for (int j = 0; j < 3; ++j) v[(k + j) % 6] = __builtin_fmaf(v[(k + j) % 6], 1.0001f, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  sink = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + v[0] + v[1] + v[2] + v[3] + v[4] + v[5];
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void probe(int iters, float* out, unsigned long long* st) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float sink = 0.f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) body<RA>(iters, sink); else body<RB>(iters, sink);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = sink;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int RA, int RB> void run(const char* tag) {
  const int blocks = 256, iters = 100;
  float* out; unsigned long long* st;
  (void)hipMalloc(&out, blocks * 512 * 4); (void)hipMalloc(&st, blocks * 64);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<RA, RB>), dim3(blocks), dim3(512), 0, 0, iters, out, st);
  (void)hipDeviceSynchronize();
  unsigned long long h[8];
  (void)hipMemcpy(h, st + 8 * 5, 64, hipMemcpyDeviceToHost);
  printf("%-70s wave0 %.1f  wave4 %.1f  cycles per slot\n", tag, (double)h[0] / (iters * 32.0), (double)h[4] / (iters * 32.0));
  (void)hipFree(out); (void)hipFree(st);
}

int main() {
  run<1, 0>("A: independent MFMAs                 B: idle");
  run<2, 0>("A: dependent MFMA chain              B: idle");
  run<1, 1>("A: independent MFMAs                 B: independent MFMAs");
  run<2, 2>("A: dependent chain                   B: dependent chain");
  run<2, 1>("A: dependent chain                   B: independent MFMAs");
  run<1, 3>("A: independent MFMAs                 B: 3 v_fma per slot");
  run<2, 3>("A: dependent chain                   B: 3 v_fma per slot");
  run<4, 0>("A: dependent chain + 3 v_fma/MFMA    B: idle");
  run<4, 4>("A: dependent chain + 3 v_fma/MFMA    B: same");
  run<4, 1>("A: dependent chain + 3 v_fma/MFMA    B: independent MFMAs");
  run<3, 3>("A: 3 v_fma per slot                  B: 3 v_fma per slot");
  return 0;
}
