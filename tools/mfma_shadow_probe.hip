// Diagnostic: how many VALU instructions of the SAME wave fit between back-to-back v_mfma_f32_16x16x32_bf16 for free?
// One wave per SIMD.  Loop body = 24 MFMAs on four independent accumulators (the tn_gemm_w4 tile), K vector
// instructions of one kind behind each; reports s_memtime ticks per MFMA for K = 0..4.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_shadow.bin tools/mfma_shadow_probe.hip && tools/_shadow.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int K, int KIND>
__global__ __launch_bounds__(256) void probe(int iters, float* out, unsigned long long* stamps) {
  f32x4 acc[4];
  for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 A[4], B;
  for (int a = 0; a < 4; ++a) A[a] = u32x4{threadIdx.x + a, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u};
  B = u32x4{0x3f803f80u, threadIdx.x, 0x3f003f00u, 0x3e803e80u};
  float v[8]; unsigned w[8]; f32x2 vv[4];
  for (int k = 0; k < 4; ++k) vv[k] = f32x2{threadIdx.x * 0.5f + k, 1.f + k};
  for (int k = 0; k < 8; ++k) { v[k] = threadIdx.x * 0.01f + k; w[k] = threadIdx.x * 77u + k; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[a]), __builtin_bit_cast(bf16x8, B), acc[a], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int i = (t * 4 + a + k) & 7;
          if (KIND == 0) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
          else if (KIND == 1) w[i] = (w[i] & 0xFFFF0000u) + 3u;               // v_and_or / v_and + v_add (integer)
          else if (KIND == 2) w[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[i], v[(i + 1) & 7]}, bf16x2)) ^ w[i];
          else if (KIND == 3) { f32x2 r = vv[i & 3] - f32x2{1.f, 2.f}; vv[i & 3] = r; }
          else if (KIND == 4) v[i] = v[i] - 1.5f;
          else if (KIND == 5) v[i] = v[i] * 1.0001f;
          else if (KIND == 8) w[i] = __builtin_amdgcn_perm(w[i], w[(i + 1) & 7], 0x07060302u);
          else if (KIND == 9) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); h2 r = __builtin_bit_cast(h2, w[i]) + h2{(_Float16)1.f, (_Float16)2.f}; w[i] = __builtin_bit_cast(unsigned, r); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][3];
  for (int k = 0; k < 8; ++k) s += v[k] + (float)w[k];
  for (int k = 0; k < 4; ++k) s += vv[k][0] + vv[k][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int K, int KIND>
void run() {
  const int blocks = 256, iters = 400;
  float* out; unsigned long long* st;
  (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&st, blocks * 8);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<K, KIND>), dim3(blocks), dim3(256), 0, 0, iters, out, st);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  (void)hipMemcpy(h.data(), st, blocks * 8, hipMemcpyDeviceToHost);
  const char* kinds[10] = {"v_fma_f32", "integer and/add", "v_cvt_pk_bf16_f32 + xor", "v_pk_add_f32", "v_sub_f32", "v_mul_f32", "-", "-", "v_perm_b32", "v_pk_add_f16"};
  printf("K=%d %-26s %.1f ticks per MFMA\n", K, kinds[KIND], (double)h[3] / (iters * 24.0));
  (void)hipFree(out); (void)hipFree(st);
}
int main() {
  run<0, 0>();
  run<1, 0>(); run<2, 0>(); run<3, 0>(); run<4, 0>();
  run<1, 1>(); run<2, 1>(); run<3, 1>();
  run<1, 2>(); run<2, 2>();
  run<1, 3>(); run<2, 3>();
  run<1, 4>(); run<2, 4>(); run<1, 5>(); run<2, 5>();
  run<1, 8>(); run<2, 8>(); run<1, 9>(); run<2, 9>();
  return 0;
}
