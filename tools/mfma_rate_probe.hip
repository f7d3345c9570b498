#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(int iters, float* out, unsigned long long* st) {
  f32x4 acc0 = {0,0,0,0}, acc1 = acc0;
  unsigned long long a = 0x3f803f803f803f80ull + threadIdx.x, b = 0x3f803f803f803f80ull;
  uint4 a8 = {(unsigned)a, (unsigned)(a>>32), (unsigned)a, 1u}, b8 = a8;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (MODE == 0) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc1, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x*256+threadIdx.x] = acc0[0]+acc1[1];
  if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; unsigned long long* st; hipMalloc(&out, 256*256*4); hipMalloc(&st, 256*8);
  unsigned long long h[256];
  hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, 200, out, st); hipDeviceSynchronize();
  hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost); printf("16x16x16 bf16_1k: %.1f cycles/MFMA\n", h[5] / (200.0 * 32));
  hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, 200, out, st); hipDeviceSynchronize();
  hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost); printf("16x16x32 bf16:    %.1f cycles/MFMA\n", h[5] / (200.0 * 32));
  return 0;
}
