// Diagnostic: semantics of ds_read_b64_tr_b16 on gfx950 (which element lands where).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[16][72];   // row stride 144 B
  for (int idx = threadIdx.x; idx < 16 * 72; idx += 64) img[idx / 72][idx % 72] = (unsigned short)((idx / 72) * 256 + (idx % 72));
  __syncthreads();
  const int l = threadIdx.x, G = l >> 4, li = l & 15, q = li >> 2, p = li & 3;
  // group G reads rows 4G..4G+3 (q), columns 16..31 (4p + 16)
  unsigned addr = (unsigned)(size_t)&img[4 * G + q][16 + 4 * p];
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)(v >> (16 * e));
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) {
    if (l % 16 < 3 || l % 16 == 15) {
      printf("lane %2d:", l);
      for (int e = 0; e < 4; ++e) printf("  (row %d, col %d)", h[l * 4 + e] >> 8, h[l * 4 + e] & 255);
      printf("\n");
    }
  }
  return 0;
}
