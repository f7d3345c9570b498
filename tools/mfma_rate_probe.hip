// Diagnostic: issue rate of bf16 MFMA chains on gfx950.
//   (1) 16x16x32 bf16, 2 independent accumulators      (2) one accumulator (dependent chain)
//   (3) one accumulator, A operand copied out of an AGPR before every MFMA (what hipcc emits when the
//       resident operands live in the accumulation file)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(int iters, float* out, unsigned long long* st) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
  u32x4 a8 = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 1u}, b8 = a8;
  u32x4 areg[8];
  for (int k = 0; k < 8; ++k) { areg[k] = a8 + (unsigned)k; asm volatile("" : "+a"(areg[k])); }   // park in AGPRs
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (MODE == 0) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc1, 0, 0, 0);
      } else if (MODE == 1) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b8), __builtin_bit_cast(bf16x8, a8), acc0, 0, 0, 0);
      } else {
        u32x4 av = areg[k & 7];
        asm volatile("" : "+a"(areg[k & 7]));           // keep it an AGPR value
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b8), acc0, 0, 0, 0);
        u32x4 aw = areg[(k + 1) & 7];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aw), __builtin_bit_cast(bf16x8, b8), acc0, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1];
  if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* tag) {
  float* out; unsigned long long* st; (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&st, 256 * 8);
  unsigned long long h[256];
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, 200, out, st);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s %.1f cycles/MFMA\n", tag, h[5] / (200.0 * 64));
}
int main() {
  run<0>("16x16x32 bf16, two independent accumulators:");
  run<1>("16x16x32 bf16, one accumulator (dependent chain):");
  run<2>("one accumulator, A operand read out of AGPRs each time:");
  return 0;
}
