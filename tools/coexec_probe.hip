// Diagnostic: does v_mfma_f32_16x16x4_f32 overlap with independent VALU work of the SAME wave?
// Loop of {1 MFMA + K independent v_fma_f32}; reports cycles per MFMA for K = 0..12, and the same
// with the VALU work in a second wave on the same SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int TRANS>
__global__ __launch_bounds__(256) void probe(int iters, float* out, unsigned long long* stamps) {
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  float av = threadIdx.x * 1e-3f, bv = 1.0f + threadIdx.x * 1e-4f;
  float v[12];
  for (int k = 0; k < 12; ++k) v[k] = threadIdx.x * 0.01f + k;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16; ++rep) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (TRANS) v[k] = __builtin_amdgcn_exp2f(v[k]); else v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc1, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (TRANS) v[k] = __builtin_amdgcn_exp2f(v[k]); else v[k] = __builtin_fmaf(v[k], 0.9999f, 0.25f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc0[0] + acc0[1] + acc1[2] + acc1[3];
  for (int k = 0; k < 12; ++k) s += v[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int K, int TRANS>
void run(const char* tag) {
  const int blocks = 256, iters = 200;
  float* out; unsigned long long* st;
  (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&st, blocks * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<K, TRANS>), dim3(blocks), dim3(256), 0, 0, iters, out, st);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  (void)hipMemcpy(h.data(), st, blocks * 8, hipMemcpyDeviceToHost);
  printf("%s K=%2d %s: %.1f cycles per MFMA (MFMA alone = 32)\n", tag, K, TRANS ? "v_exp_f32" : "v_fma_f32", (double)h[3] / (iters * 32.0));
  (void)hipFree(out); (void)hipFree(st);
}

// two waves per SIMD: waves 0-3 MFMA only, waves 4-7 VALU only (ROLE split) or both do both
template <int K>
__global__ __launch_bounds__(512) void probe2(int iters, float* out, unsigned long long* stamps, int split) {
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  float av = threadIdx.x * 1e-3f, bv = 1.0f + threadIdx.x * 1e-4f;
  float v[12];
  for (int k = 0; k < 12; ++k) v[k] = threadIdx.x * 0.01f + k;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_mfma = !split || wave < 4, do_valu = !split || wave >= 4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16; ++rep) {
      if (do_mfma) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc1, 0, 0, 0);
      }
      if (do_valu) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = __builtin_fmaf(v[k], 0.9999f, 0.25f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc0[0] + acc0[1] + acc1[2] + acc1[3];
  for (int k = 0; k < 12; ++k) s += v[k];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int K>
void run2(int split) {
  const int blocks = 256, iters = 200;
  float* out; unsigned long long* st;
  (void)hipMalloc(&out, blocks * 512 * 4); (void)hipMalloc(&st, blocks * 64);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe2<K>), dim3(blocks), dim3(512), 0, 0, iters, out, st, split);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  (void)hipMemcpy(h.data(), st, blocks * 64, hipMemcpyDeviceToHost);
  printf("2 waves/SIMD, %s, K=%2d v_fma per MFMA: wave0 %.1f cycles per MFMA-slot, wave4 %.1f\n",
         split ? "roles split (w0-3 MFMA, w4-7 VALU)" : "both waves do both", K, (double)h[24] / (iters * 32.0), (double)h[28] / (iters * 32.0));
  (void)hipFree(out); (void)hipFree(st);
}

// bf16 MFMA (16x16x32) + K independent v_fma in the same wave
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int K>
__global__ __launch_bounds__(256) void probe_bf16(int iters, float* out, unsigned long long* stamps) {
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  uint4 a8 = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 1u}, b8 = a8;
  float v[12];
  for (int k = 0; k < 12; ++k) v[k] = threadIdx.x * 0.01f + k;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16; ++rep) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a8), __builtin_bit_cast(bf16x8_t, b8), acc0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a8), __builtin_bit_cast(bf16x8_t, b8), acc1, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = __builtin_fmaf(v[k], 0.9999f, 0.25f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc0[0] + acc0[1] + acc1[2] + acc1[3];
  for (int k = 0; k < 12; ++k) s += v[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}
template <int K>
void run_bf16() {
  const int blocks = 256, iters = 200;
  float* out; unsigned long long* st;
  (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&st, blocks * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe_bf16<K>), dim3(blocks), dim3(256), 0, 0, iters, out, st);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  (void)hipMemcpy(h.data(), st, blocks * 8, hipMemcpyDeviceToHost);
  printf("bf16 16x16x32 MFMA + K=%2d v_fma (same wave): %.1f cycles per MFMA\n", K, (double)h[3] / (iters * 32.0));
  (void)hipFree(out); (void)hipFree(st);
}

int main() {
  run_bf16<0>(); run_bf16<1>(); run_bf16<2>(); run_bf16<3>(); run_bf16<4>(); run_bf16<6>(); run_bf16<8>(); run_bf16<12>();
  run2<0>(0); run2<8>(0); run2<8>(1); run2<12>(1); run2<4>(1);
  run<0, 0>("same wave"); run<2, 0>("same wave"); run<4, 0>("same wave"); run<6, 0>("same wave");
  run<8, 0>("same wave"); run<12, 0>("same wave");
  run<2, 1>("same wave"); run<4, 1>("same wave");
  return 0;
}
