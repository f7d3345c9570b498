// Diagnostic: v_mfma_f32_16x16x16_bf16 (K = 16, two registers per operand) on gfx950 -- issue rate beside the
// 16x16x32 form, alone and with VALU fillers, and the K mapping of its operands (lane group g holds k = 4g..4g+3).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(int iters, float* out, unsigned long long* st) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
  u32x4 a8 = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 1u}, b8 = a8;
  s16x4 a4 = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)threadIdx.x}, b4 = a4;
  float f0 = threadIdx.x, f1 = 1.0f, f2 = 2.0f, f3 = 0.5f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (MODE == 0) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc1, 0, 0, 0);
      } else if (MODE == 1) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc1, 0, 0, 0);
      } else if (MODE == 2) {                        // K = 16 form + 4 VALU fillers per pair
        acc0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc0, 0, 0, 0);
        f0 = __builtin_fmaf(f0, f1, f2); f3 = __builtin_fmaf(f3, f1, f2);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc1, 0, 0, 0);
        f1 = __builtin_fmaf(f1, f3, f0); f2 = __builtin_fmaf(f2, f0, f3);
      } else {                                       // K = 32 form + 4 VALU fillers per pair
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc0, 0, 0, 0);
        f0 = __builtin_fmaf(f0, f1, f2); f3 = __builtin_fmaf(f3, f1, f2);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc1, 0, 0, 0);
        f1 = __builtin_fmaf(f1, f3, f0); f2 = __builtin_fmaf(f2, f0, f3);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + f0 + f1 + f2 + f3;
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
// layout: C[m][n] = sum_k A[m][k] B[k][n] with A[m][k] = (m + 1) if k == m else 0 ... checked against a host loop
__global__ void layout(float* out) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  s16x4 a, b;
  for (int e = 0; e < 4; ++e) {
    const int k = 4 * g + e;
    const float av = (float)((i * 3 + k * 5) % 7 - 3), bv = (float)((k * 2 + i * 7) % 5 - 2);   // A[i][k], B[k][i]
    a[e] = (short)(__builtin_bit_cast(unsigned, av) >> 16); b[e] = (short)(__builtin_bit_cast(unsigned, bv) >> 16);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + i] = c[r];     // row 4g + r, column i
}
int main() {
  run<0>("16x16x32 bf16, two independent accumulators:");
  run<1>("16x16x16 bf16, two independent accumulators:");
  run<2>("16x16x16 bf16 + 2 v_fma per MFMA:");
  run<3>("16x16x32 bf16 + 2 v_fma per MFMA:");
  float* d; (void)hipMalloc(&d, 256 * 4); float h[256];
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    float ref = 0;
    for (int k = 0; k < 16; ++k) ref += (float)((m * 3 + k * 5) % 7 - 3) * (float)((k * 2 + n * 7) % 5 - 2);
    if (h[m * 16 + n] != ref) ++bad;
  }
  printf("16x16x16 layout (lane group g holds k = 4g..4g+3; D row 4g+r, column l&15): %d mismatches\n", bad);
  return 0;
}
