// Diagnostic: how does v_mfma_f32_16x16x32_bf16 round its internal sum?  D[0][0] = sum_k A[0][k] * B[k][0] + C
// with one big product and one sub-ulp product (both exact in fp32), every sign combination, C = 0 or the big
// addend in C.  Prints the result as a multiple of 2^-24 away from the big addend.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(const float* ab, float* out) {   // ab: a0, b0, a1, b1, c
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // A operand: lane (i, g) holds A[i][8g .. 8g+7]; B operand: lane (i, g) holds B[8g .. 8g+7][i]
  bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
  if (i == 0 && g == 0) { a[0] = (__bf16)ab[0]; a[1] = (__bf16)ab[2]; b[0] = (__bf16)ab[1]; b[1] = (__bf16)ab[3]; }
  f32x4 c = {0, 0, 0, 0};
  if (l == 0) c[0] = ab[4];
  f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  if (l == 0) out[0] = d[0];
}
int main() {
  float *dab, *dout; (void)hipMalloc(&dab, 20); (void)hipMalloc(&dout, 4);
  const float e = ldexpf(1.0f, -24);
  struct Case { const char* tag; float a0, b0, a1, b1, c; } cases[] = {
    {"+1 + 1.5e (two products)", 1, 1, 1.5f, e, 0},     {"-1 - 1.5e", 1, -1, 1.5f, -e, 0},
    {"+1 - 1.25e", 1, 1, 1.25f, -e, 0},                  {"-1 + 1.25e", 1, -1, 1.25f, e, 0},
    {"+1 + 0.75e", 1, 1, 0.75f, e, 0},                   {"-1 - 0.75e", 1, -1, 0.75f, -e, 0},
    {"+1 - 0.75e", 1, 1, 0.75f, -e, 0},                  {"-1 + 0.75e", 1, -1, 0.75f, e, 0},
    {"C=+1, product +1.5e", 0, 0, 1.5f, e, 1},           {"C=-1, product -1.5e", 0, 0, 1.5f, -e, -1},
    {"C=+1, product -1.25e", 0, 0, 1.25f, -e, 1},        {"C=-1, product +1.25e", 0, 0, 1.25f, e, -1},
    {"C=+1, product -0.75e", 0, 0, 0.75f, -e, 1},        {"C=-1, product +0.75e", 0, 0, 0.75f, e, -1},
  };
  for (auto& cs : cases) {
    float h[5] = {cs.a0, cs.b0, cs.a1, cs.b1, cs.c};
    (void)hipMemcpy(dab, h, 20, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dab, dout);
    float r; (void)hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
    const double exact = (double)cs.a0 * cs.b0 + (double)cs.a1 * cs.b1 + cs.c;
    const float rne = (float)exact;
    printf("%-26s exact %+.10f  mfma %+.10f  fp32-RNE %+.10f  %s\n", cs.tag, exact, r, rne, r == rne ? "= RNE" : (fabs(r) < fabs(rne) ? "toward zero" : "away"));
  }
  return 0;
}
