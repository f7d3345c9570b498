// Diagnostic: cost of the scans' operand loads per lane mapping.  512 threads = 16 utterances x 256 units, three
// [T][B][256] fp32 tensors, each wave reads its 32 units of every utterance per step (2 KB per tensor and wave):
//   MODE 0: lane (utterance l & 15, group l >> 4) reads units 8g..8g+7      (the MFMA operand layout: a quarter-wave
//           touches 16 different 128-byte lines, 16 bytes of each)
//   MODE 1: lane (utterance l >> 2, piece l & 3) reads units 8q..8q+7       (a quarter-wave = 4 full lines)
//   MODE 2: lane (utterance l >> 3, piece l & 7): ONE 16-byte piece of utterances u and u + 8 (a quarter-wave = 2 full lines)
// with a fixed amount of dependent VALU work per step in between.  Prints microseconds per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void probe(int T, int B, const float* a, const float* b, const float* c, float* out, int work) {
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
  const int b0 = blockIdx.x * 16;
  size_t off0, off1;
  if (MODE == 0) { const int u = l & 15, g = l >> 4; off0 = (size_t)(b0 + u) * 256 + wv * 32 + g * 8; off1 = off0 + 4; }
  else if (MODE == 1) { const int u = l >> 2, q = l & 3; off0 = (size_t)(b0 + u) * 256 + wv * 32 + q * 8; off1 = off0 + 4; }
  else { const int u = l >> 3, q = l & 7; off0 = (size_t)(b0 + u) * 256 + wv * 32 + q * 4; off1 = off0 + 8 * 256; }
  f32x4 acc = {0, 0, 0, 0};
  f32x4 v[6];
  auto ld = [&](int t) {
    const size_t s = (size_t)t * B * 256;
    v[0] = *(const f32x4*)(a + s + off0); v[1] = *(const f32x4*)(a + s + off1);
    v[2] = *(const f32x4*)(b + s + off0); v[3] = *(const f32x4*)(b + s + off1);
    v[4] = *(const f32x4*)(c + s + off0); v[5] = *(const f32x4*)(c + s + off1);
  };
  ld(T - 1);
  for (int t = T - 1; t >= 0; --t) {
    f32x4 s = (v[0] + v[1]) * (v[2] + v[3]) + (v[4] + v[5]);
    __builtin_amdgcn_sched_barrier(0);
    if (t > 0) ld(t - 1);
    __builtin_amdgcn_sched_barrier(0);
    for (int k = 0; k < work; ++k) s = s * s + acc;       // dependent VALU chain
    acc += s;
    __syncthreads();
  }
  out[(size_t)blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int MODE> void run(const char* tag, int T, int B, const float* a, const float* b, const float* c, float* out, int work) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 6; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(B / 16), dim3(512), 0, 0, T, B, a, b, c, out, work);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (r >= 1 && ms < best) best = ms;
  }
  printf("%-34s work %4d: %.1f us  (%.2f us/step, %.2f TB/s)\n", tag, work, best * 1e3, best * 1e3 / T, 3.0 * T * B * 1024 / (best * 1e-3) / 1e12);
}
int main() {
  const int T = 99, B = 4096;
  float *a, *b, *c, *out;
  const size_t n = (size_t)T * B * 256;
  (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4); (void)hipMalloc(&c, n * 4); (void)hipMalloc(&out, (size_t)B / 16 * 512 * 4);
  (void)hipMemset(a, 0, n * 4); (void)hipMemset(b, 0, n * 4); (void)hipMemset(c, 0, n * 4);
  for (int work : {200, 600, 200}) {
    run<2>("8 lanes per utterance", T, B, a, b, c, out, work);
    run<1>("4 lanes per utterance", T, B, a, b, c, out, work);
    run<0>("MFMA layout (16 lines / quarter)", T, B, a, b, c, out, work);
    run<1>("4 lanes per utterance", T, B, a, b, c, out, work);
    run<0>("MFMA layout (16 lines / quarter)", T, B, a, b, c, out, work);
  }
  return 0;
}
