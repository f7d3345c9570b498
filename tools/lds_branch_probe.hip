// Probe for the hazard suspected in DESIGN.md 4.0:  a pending LDS write's dependency on its DATA register is lost
// across a conditional branch.  Every wave repeats
//     ds_write2st64_b64 addr, d0, d1            (two 8-byte values into its own LDS slot)
//     s_and_saveexec / s_cbranch_execz           (an exec-masked block that every lane enters)
//       v_lshl_add_u64 d1, ptr, 0, soff          (VALU overwrite of the write's second data register)
//       global_store_dwordx2 d1, d0, off         (... which is the store's address)
//     s_or_b64 exec
// then waits, reads the slot back and counts the words that are not what was written.  MODE 1 puts
// `s_waitcnt lgkmcnt(0)` in front of the branch, MODE 2 leaves the branch out.  8 waves per workgroup, one workgroup
// per CU, all hammering the LDS at once (the failing kernel ran two waves per SIMD next to heavy LDS traffic).
//   hipcc -O3 --offload-arch=gfx950 -o lds_branch_probe tools/lds_branch_probe.hip && ./lds_branch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long* sink, unsigned* bad) {
  __shared__ unsigned long long slots[512 * 20];                      // 80 KB: one workgroup per CU
  const int tid = threadIdx.x;
  const unsigned lds_addr = (unsigned)(size_t)&slots[0] + tid * 8;     // st64 offsets: +0 and +9*512 bytes
  unsigned long long gptr = (unsigned long long)(sink + (size_t)blockIdx.x * 512 + tid);
  unsigned errors = 0;
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(true);
  for (int it = 0; it < iters; ++it) {
    unsigned long long d0 = 0x1111000000000000ull + ((unsigned long long)it << 16) + tid;
    unsigned long long d1 = 0x2222000000000000ull + ((unsigned long long)it << 16) + tid;
    const unsigned long long want1 = d1;
    unsigned long long sv;
    const unsigned long long soff = (unsigned long long)((it & 7) * 0);   // wave-uniform zero the compiler cannot fold
    if (MODE == 2) {
      asm volatile("ds_write2st64_b64 %[addr], %[d0], %[d1] offset0:0 offset1:9\n\t"
                   "v_lshl_add_u64 %[d1], %[gp], 0, %[soff]\n\t"
                   "global_store_dwordx2 %[d1], %[d0], off\n\t"
                   : [d1] "+v"(d1)
                   : [addr] "v"(lds_addr), [d0] "v"(d0), [gp] "v"(gptr), [soff] "s"(soff)
                   : "memory");
    } else if (MODE == 1) {
      asm volatile("ds_write2st64_b64 %[addr], %[d0], %[d1] offset0:0 offset1:9\n\t"
                   "s_waitcnt lgkmcnt(0)\n\t"
                   "s_and_saveexec_b64 %[sv], %[mask]\n\t"
                   "s_cbranch_execz .Lskipw%=\n\t"
                   "v_lshl_add_u64 %[d1], %[gp], 0, %[soff]\n\t"
                   "global_store_dwordx2 %[d1], %[d0], off\n\t"
                   ".Lskipw%=:\n\t"
                   "s_or_b64 exec, exec, %[sv]\n\t"
                   : [d1] "+v"(d1), [sv] "=&s"(sv)
                   : [addr] "v"(lds_addr), [d0] "v"(d0), [gp] "v"(gptr), [soff] "s"(soff), [mask] "s"(mask)
                   : "memory");
    } else {
      asm volatile("ds_write2st64_b64 %[addr], %[d0], %[d1] offset0:0 offset1:9\n\t"
                   "s_and_saveexec_b64 %[sv], %[mask]\n\t"
                   "s_cbranch_execz .Lskip%=\n\t"
                   "v_lshl_add_u64 %[d1], %[gp], 0, %[soff]\n\t"
                   "global_store_dwordx2 %[d1], %[d0], off\n\t"
                   ".Lskip%=:\n\t"
                   "s_or_b64 exec, exec, %[sv]\n\t"
                   : [d1] "+v"(d1), [sv] "=&s"(sv)
                   : [addr] "v"(lds_addr), [d0] "v"(d0), [gp] "v"(gptr), [soff] "s"(soff), [mask] "s"(mask)
                   : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long got0 = slots[tid], got1 = slots[tid + 9 * 64];
    if (got0 != d0) ++errors;
    if (got1 != want1) ++errors;
    asm volatile("" :: "v"(d1));
  }
  if (errors) atomicAdd(bad, errors);
}

template <int MODE> void run(const char* tag, int iters) {
  unsigned long long* sink; unsigned* bad;
  (void)hipMalloc(&sink, 256 * 512 * 8); (void)hipMalloc(&bad, 4); (void)hipMemset(bad, 0, 4);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, iters, sink, bad);
  (void)hipDeviceSynchronize();
  unsigned h = 0; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("%-58s %u wrong words in %d x 131072 writes\n", tag, h, iters);
  (void)hipFree(sink); (void)hipFree(bad);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  run<0>("LDS write, branch, VALU overwrite of its data, store:", iters);
  run<2>("same without the branch:", iters);
  run<1>("with s_waitcnt lgkmcnt(0) in front of the branch:", iters);
  run<0>("LDS write, branch, VALU overwrite of its data, store:", iters);
  return 0;
}
