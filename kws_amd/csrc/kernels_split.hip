// Split-precision FastGRNN scan for gfx950: fp32 results from the bf16 matrix pipe.
//
// Measured on MI355X (tools/coexec_probe.hip): v_mfma_f32_16x16x4_f32 runs on the SIMD's fp32
// FMA lanes -- it does NOT overlap with VALU work of the same (or another) wave, so an
// fp32-MFMA scan pays MFMA cycles + VALU cycles.  The bf16 matrix pipe is separate, overlaps
// with the VALU and is 16x faster per flop.  Every fp32 operand is therefore split exactly
// into three bf16 planes,  a = a0 + a1 + a2  (a0 = top 16 bits of a, a1 = top 16 bits of
// a - a0, a2 = a - a0 - a1: exact, 8+8+8 mantissa bits), and each product is evaluated as the
// six terms  a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0)  on v_mfma_f32_16x16x32_bf16 with
// fp32 accumulation; the dropped terms are O(2^-24) -- the size of an fp32 rounding.  On the
// cell's shapes the result is closer to an fp64 evaluation than an fp32 fma chain
// (tests/test_hip_parity.py), at 6/16 of the fp32-MFMA cycles.
//
// Geometry as kernels_mfma.hip: workgroup = 4 waves = 16 utterances for all T; wave w owns
// hidden units 32w..32w+31 (two 16-row tiles); utterances on the MFMA N axis.  A operands
// (the three planes of the wave's U / W rows) stay in VGPRs for the whole scan.  A lane's 8
// results n0..n0+7 are exactly one B fragment (8 consecutive k of K-step w), so the producer
// splits them and publishes three 16-byte fragments per step with ds_write_b128; consumers
// read 4 K-steps x 3 planes with conflict-free ds_read_b128.
//
// Reference semantics: forward .cu:42-60 + .cu:367-413; backward .cu:91-119 + .cu:473-545.
#include "common.h"
#include <type_traits>

namespace fastgrnn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ float fsigmoid(float a) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * a));
}
__device__ __forceinline__ float ftanh(float a) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((2.0f * LOG2E) * a));
}
template <int GATE> __device__ __forceinline__ float gate_act(float a) {
  if (GATE == FASTGRNN_NL_SIGMOID) return fsigmoid(a);
  if (GATE == FASTGRNN_NL_RELU) return a > 0.0f ? a : 0.0f;
  return ftanh(a);
}
template <int GATE> __device__ __forceinline__ float gate_dact(float y) {
  if (GATE == FASTGRNN_NL_SIGMOID) return (1.0f - y) * y;
  if (GATE == FASTGRNN_NL_RELU) return y > 0.0f ? 1.0f : 0.0f;
  return 1.0f - y * y;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                 0, 0);
}

// Three bf16 planes of 8 fp32 values (one MFMA fragment each).  Exact: p0+p1+p2 == v.
struct Frag3 { u32x4 p[3]; };

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned x) { return __builtin_bit_cast(float, x); }
// {hi16(b), hi16(a)} -> one dword of two bf16 (element order a, b)
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__device__ __forceinline__ Frag3 split3(const f32x4 lo, const f32x4 hi) {
  float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  unsigned b0[8], b1[8], b2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    b0[j] = fbits(v[j]);
    const float r1 = v[j] - bitsf(b0[j] & 0xFFFF0000u);
    b1[j] = fbits(r1);
    const float r2 = r1 - bitsf(b1[j] & 0xFFFF0000u);
    b2[j] = fbits(r2);                       // <= 8 significant bits left: its low half is zero
  }
  Frag3 f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f.p[0][q] = pack_hi(b0[2 * q], b0[2 * q + 1]);
    f.p[1][q] = pack_hi(b1[2 * q], b1[2 * q + 1]);
    f.p[2][q] = pack_hi(b2[2 * q], b2[2 * q + 1]);
  }
  return f;
}

// acc += sum over the six retained plane pairs of A[pa] . B[pb]   (small terms first)
__device__ __forceinline__ f32x4 mfma6(const Frag3& a, const Frag3& b, f32x4 acc) {
  acc = mfma_bf16(a.p[2], b.p[0], acc);
  acc = mfma_bf16(a.p[1], b.p[1], acc);
  acc = mfma_bf16(a.p[0], b.p[2], acc);
  acc = mfma_bf16(a.p[1], b.p[0], acc);
  acc = mfma_bf16(a.p[0], b.p[1], acc);
  acc = mfma_bf16(a.p[0], b.p[0], acc);
  return acc;
}

// ------------------------------------------------------------------------------------------
// forward  (H = 128, F = 32)
// ------------------------------------------------------------------------------------------
template <int GATE, bool GATES_OUT, bool RAGGED>
__global__ __launch_bounds__(256) void fwd_scan_split(
    int Tn, int B, const float* __restrict__ x, const float* __restrict__ h0,
    const float* __restrict__ w, const float* __restrict__ u,
    const float* __restrict__ bz, const float* __restrict__ bh,
    const float* __restrict__ zeta, const float* __restrict__ nu,
    float* __restrict__ hs, float* __restrict__ zs, float* __restrict__ cs) {
  constexpr int H = 128, F = 32, MT = 2, KS = H / 32;
  // state tile: [buffer][plane][(K-step*4 + lane group)*16 + utterance] fragments of 8 bf16
  __shared__ u32x4 hl[2][3][KS * 4 * 16];

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, i = l & 15, g = l >> 4;
  const int b = blockIdx.x * 16 + i;
  const bool valid = !RAGGED || b < B;
  const int bc = valid ? b : B - 1;
  const int n0 = wv * 32 + g * 8;                  // this lane's 8 hidden units (= K-step wv, group g)
  const int myfrag = (wv * 4 + g) * 16 + i;

  // ---- resident A operands: three planes of U (4 K-steps) and W (1 K-step) per tile -------------
  Frag3 Uf[MT][KS], Wf[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int nA = wv * 32 + (i >> 2) * 8 + mt * 4 + (i & 3);     // A row i of tile mt
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float* up = u + (size_t)nA * H + 32 * s + 8 * g;
      Uf[mt][s] = split3(ld4(up), ld4(up + 4));
    }
    const float* wp = w + (size_t)nA * F + 8 * g;
    Wf[mt] = split3(ld4(wp), ld4(wp + 4));
  }
  f32x4 bzv[MT], bhv[MT], hown[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    bzv[mt] = ld4(bz + n0 + 4 * mt);
    bhv[mt] = ld4(bh + n0 + 4 * mt);
    hown[mt] = ld4(h0 + (size_t)bc * H + n0 + 4 * mt);
  }
  {
    const Frag3 f = split3(hown[0], hown[1]);
#pragma unroll
    for (int p = 0; p < 3; ++p) hl[0][p][myfrag] = f.p[p];
  }
  const float sz = fsigmoid(zeta[0]), sn = fsigmoid(nu[0]);

  struct Feat { f32x4 lo, hi; };          // x[t][b][8g .. 8g+7]
  struct Gates { f32x4 z[MT], c[MT]; };
  auto load_x = [&](int t, Feat& q) __attribute__((always_inline)) {
    const float* xp = x + ((size_t)t * B + bc) * F + 8 * g;
    q.lo = ld4(xp); q.hi = ld4(xp + 4);
  };
  auto store_step = [&](int t, const Gates& gt) __attribute__((always_inline)) {   // hown still holds h_t
    if (valid) {
      const size_t o = ((size_t)t * B + b) * H + n0;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) st4(hs + o + 4 * mt, hown[mt]);
      if (GATES_OUT) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { st4(zs + o + 4 * mt, gt.z[mt]); st4(cs + o + 4 * mt, gt.c[mt]); }
      }
    }
  };

  // One step; xuse = features of frame t (requested a step ago), xload <- frame t+1.
  auto step = [&](auto first_tag, int t, int cur, Feat& xuse, Feat& xload, Gates& gprev,
                  Gates& gout) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    load_x(t + 1 < Tn ? t + 1 : t, xload);
    // B operands: the whole state tile h_{t-1}, three planes x four K-steps
    Frag3 hB[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int p = 0; p < 3; ++p) hB[s].p[p] = hl[cur][p][(s * 4 + g) * 16 + i];
    // W.x_t: independent of h, covers the LDS round trip
    const Frag3 xB = split3(xuse.lo, xuse.hi);
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma6(Wf[mt], xB, f32x4{0.f, 0.f, 0.f, 0.f});
    __builtin_amdgcn_sched_barrier(0);
    if (!FIRST) store_step(t - 1, gprev);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma6(Uf[mt][s], hB[s], acc[mt]);          // .cu:368
    if (!FIRST) {
      // spread the store instructions of step t-1 evenly under the chain's 48 MFMAs
      constexpr int NST = GATES_OUT ? 3 * MT : MT, NM = KS * MT * 6, PER = NM / (NST + 1);
#pragma unroll
      for (int j = 0; j < NST; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);     // VMEM write
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM - NST * PER, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // epilogue (.cu:55-58)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = acc[mt][r];
        const float z = gate_act<GATE>(pre + bzv[mt][r]);
        const float c = ftanh(pre + bhv[mt][r]);
        hown[mt][r] = (sz * (1.0f - z) + sn) * c + hown[mt][r] * z;
        gout.z[mt][r] = z; gout.c[mt][r] = c;
      }
    }
    const Frag3 f = split3(hown[0], hown[1]);
#pragma unroll
    for (int p = 0; p < 3; ++p) hl[cur ^ 1][p][myfrag] = f.p[p];
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
  };

  Feat xa, xb;
  Gates ga, gb;
  load_x(0, xa);
  __syncthreads();
  step(std::true_type{}, 0, 0, xa, xb, gb, ga);
  int t = 1;
  for (; t + 1 < Tn; t += 2) {
    step(std::false_type{}, t, 1, xb, xa, ga, gb);
    step(std::false_type{}, t + 1, 0, xa, xb, gb, ga);
  }
  if (t < Tn) {
    step(std::false_type{}, t, 1, xb, xa, ga, gb);
    store_step(Tn - 1, gb);
  } else {
    store_step(Tn - 1, ga);
  }
}

template <int GATE>
void launch_fwd_gate(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs,
                     void* zs, void* cs, hipStream_t s) {
  dim3 grid((d.B + 15) / 16), block(256);
  const bool ragged = (d.B % 16) != 0, gates = zs != nullptr;
  auto go = [&](auto kern) __attribute__((always_inline)) {
    hipLaunchKernelGGL(kern, grid, block, 0, s, d.T, d.B, (const float*)x, (const float*)h0, (const float*)p.w,
                       (const float*)p.u, (const float*)p.bias_gate, (const float*)p.bias_update,
                       (const float*)p.zeta, (const float*)p.nu, (float*)hs, (float*)zs, (float*)cs);
  };
  if (gates) { if (ragged) go(fwd_scan_split<GATE, true, true>); else go(fwd_scan_split<GATE, true, false>); }
  else       { if (ragged) go(fwd_scan_split<GATE, false, true>); else go(fwd_scan_split<GATE, false, false>); }
}

}  // namespace

bool split_supported(const fastgrnn_desc& d, int direction) {
  return direction == 0 && d.dtype == FASTGRNN_F32 && d.w_rank == 0 && d.u_rank == 0 &&
         d.update_nl == FASTGRNN_NL_TANH && d.gate_nl >= FASTGRNN_NL_SIGMOID && d.gate_nl <= FASTGRNN_NL_TANH &&
         d.H == 128 && d.F == 32;
}

size_t split_backward_ws(const fastgrnn_desc&) { return 0; }
int split_backward(const fastgrnn_desc&, const fastgrnn_params&, const void*, const void*, const void*, const void*,
                   const void*, const void*, const fastgrnn_grads&, void*, hipStream_t) {
  return FASTGRNN_ERR_UNSUPPORTED;
}

int split_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs,
                  void* zs, void* cs, void*, hipStream_t s) {
  if ((zs == nullptr) != (cs == nullptr)) return FASTGRNN_ERR_NULL_POINTER;
  switch (d.gate_nl) {
    case FASTGRNN_NL_SIGMOID: launch_fwd_gate<FASTGRNN_NL_SIGMOID>(d, p, x, h0, hs, zs, cs, s); break;
    case FASTGRNN_NL_RELU: launch_fwd_gate<FASTGRNN_NL_RELU>(d, p, x, h0, hs, zs, cs, s); break;
    default: launch_fwd_gate<FASTGRNN_NL_TANH>(d, p, x, h0, hs, zs, cs, s); break;
  }
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

}  // namespace fastgrnn
