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
#include "split_common.h"

namespace fastgrnn {
namespace {

// ------------------------------------------------------------------------------------------
// forward  (H = 128, F = 32)
// ------------------------------------------------------------------------------------------
// AUX: 0 = hs only; 1 = also z_s, h_prime_s (the reference operator's outputs); 2 = also the
// pre-activation W.x+U.h into zs (FASTGRNN_FLAG_SAVE_PREACT, consumed by bwd_scan_split_w8<PREACT>).
template <int GATE, int AUX, bool RAGGED>
__global__ __launch_bounds__(256) ONE_WAVE_PER_SIMD void fwd_scan_split(
    int Tn, int B, int rsT, int rsB, const float* __restrict__ x, const float* __restrict__ h0,
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
    const float* xp = x + ((size_t)t * rsT + (size_t)bc * rsB) * F + 8 * g;
    q.lo = ld4(xp); q.hi = ld4(xp + 4);
  };
  auto store_step = [&](int t, const Gates& gt) __attribute__((always_inline)) {   // hown still holds h_t
    if (valid) {
      const size_t o = ((size_t)t * rsT + (size_t)b * rsB) * H + n0;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) st4(hs + o + 4 * mt, hown[mt]);
      if (AUX == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { st4(zs + o + 4 * mt, gt.z[mt]); st4(cs + o + 4 * mt, gt.c[mt]); }
      } else if (AUX == 2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) st4(zs + o + 4 * mt, gt.z[mt]);      // gt.z carries the pre-activation
      }
    }
  };

  // One step; xuse = features of frame t (requested a step ago), xload <- frame t+1.
#ifdef FASTGRNN_DIAG_STAMPS
  unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dlast)::"memory");
#endif
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
    __builtin_amdgcn_sched_barrier(0);       // every state-plane read is issued before the first MFMA (operand rule)
    // The two row tiles run one after the other: tile 0's VALU epilogue then sits under tile 1's MFMA
    // chain (the bf16 matrix pipe overlaps with the VALU).  W.x_t goes first in each tile: it does not
    // depend on h, so it (and the x split) covers the LDS round trip of the state planes.
    const Frag3 xB = split3(xuse.lo, xuse.hi);
    auto tile_chain = [&](int mt) __attribute__((always_inline)) {
      // (one accumulator: this A/B kernel keeps the plain six-term chain; mfma6_hl is in the 8-wave kernels)
      f32x4 a = mfma6(Wf[mt], xB, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int s = 0; s < KS; ++s) a = mfma6(Uf[mt][s], hB[s], a);                        // .cu:368
      return a;
    };
    auto tile_epilogue = [&](int mt, const f32x4 a) __attribute__((always_inline)) {     // .cu:55-58
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = a[r];
        const float z = gate_act<GATE>(pre + bzv[mt][r]);
        const float c = ftanh(pre + bhv[mt][r]);
        hown[mt][r] = (sz * (1.0f - z) + sn) * c + hown[mt][r] * z;
        gout.z[mt][r] = (AUX == 2) ? pre : z; gout.c[mt][r] = c;
      }
    };
    constexpr int NM = (KS + 1) * 6;                 // MFMAs per tile chain
    SPLIT_STAMP(0)
    // ---- tile 0 chain; the global stores of step t-1 (they still read h_{t-1}) spread beneath it
    if (!FIRST) store_step(t - 1, gprev);
    const f32x4 acc0 = tile_chain(0);
    if (!FIRST) {
      constexpr int NST = (AUX == 1 ? 3 : AUX == 2 ? 2 : 1) * MT, PER = NM / (NST + 1);
#pragma unroll
      for (int j = 0; j < NST; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);     // VMEM write
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STAMP(1)
    // ---- tile 1 chain || tile 0 epilogue
    const f32x4 acc1 = tile_chain(1);
    tile_epilogue(0, acc0);
#pragma unroll
    for (int k = 0; k < NM; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);       // VALU
    }
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STAMP(2)
    // ---- tile 1 epilogue, planes of h_t
    {
      // the operand registers stay allocated until the youngest accumulator has been read (= the chains have
      // retired): nothing may be loaded into a fragment behind an MFMA that still has to fetch it
      float probe = acc1[0];
      keep_alive(probe, xB);
#pragma unroll
      for (int s = 0; s < KS; ++s) keep_alive(probe, hB[s]);
      completion_read(probe);
      __builtin_amdgcn_sched_barrier(0);
    }
    tile_epilogue(1, acc1);
    const Frag3 f = split3(hown[0], hown[1]);
#pragma unroll
    for (int p = 0; p < 3; ++p) hl[cur ^ 1][p][myfrag] = f.p[p];
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STAMP(3)
    lds_barrier();
    SPLIT_STAMP(4)
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
  {                                           // ... and the weight fragments through the last step
    float probe = hown[0][0];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      keep_alive(probe, Wf[mt]);
#pragma unroll
      for (int s = 0; s < KS; ++s) keep_alive(probe, Uf[mt][s]);
    }
  }
#ifdef FASTGRNN_DIAG_STAMPS
  if (blockIdx.x == 7 && l == 0) { for (int k = 0; k < 8; ++k) g_sdiag[wv][k] = dsum[k]; }
#endif
}

// ------------------------------------------------------------------------------------------
// forward, 8 waves  (H = 128, F = 32)
// ------------------------------------------------------------------------------------------
// Same arithmetic as fwd_scan_split, other shape: workgroup = 8 waves = 16 utterances, wave w owns
// ONE 16-unit row tile (units 16w..16w+15), two waves per SIMD.  A wave's critical path per step is
// one 30-MFMA chain + a 4-element epilogue; while it waits (chain latency, LDS round trip, barrier)
// the SIMD's other wave issues, so matrix pipe and VALU overlap across waves without hand-placed
// schedules.  State AND feature planes go through LDS in natural [utterance][unit] order: every lane
// splits one feature value of the next frame, so the x split is not replicated per wave.
constexpr int W8_ROWH = 272;            // bytes per utterance row of a 128-wide bf16 plane (256 + 16: conflict-free b128 reads)
constexpr int W8_ROWX = 80;             // bytes per utterance row of a 32-wide bf16 plane (64 + 16)

// PREIN: the frame product W.x_t is NOT computed here: it arrives as a [T,B,H] fp32 tensor P = X.W^T from the batched
// GEMM of kernels_gemm.hip (input widths other than 32: the reference's second layer has F = 256, model.py:196-203)
// and enters each step as the C-in of the state product.  P is read through the pointer the step's fp32 auxiliary
// output goes to -- zs (AUX 0 / 2 / 3: the pre-activation overwrites it in place under AUX == 2) or cs (AUX == 1: c
// overwrites it) -- so no two restrict-qualified pointers alias.  x, w are unused.
// UQ: the update nonlinearity is quantTanh = clip(a, -1, 1) instead of tanh (the CPU cell allows it, rnn.py:57-58,
// 292-293; the reference's CUDA classes fix tanh).  Instantiated for fp32 time- or batch-major sequences without
// PREIN; hs only or the one-saved-tensor contract.
template <int GATE, int AUX, bool RAGGED, bool BF = false, bool F16H = true, bool PREIN = false, bool UQ = false>
__global__ __launch_bounds__(512) void fwd_scan_split_w8(
    int Tn, int B, int rsT, int rsB, int mode, const float* __restrict__ x, const float* __restrict__ h0,
    const float* __restrict__ w, const float* __restrict__ u,
    const float* __restrict__ bz, const float* __restrict__ bh,
    const float* __restrict__ zeta, const float* __restrict__ nu,
    float* __restrict__ hs, float* __restrict__ zs, float* __restrict__ cs) {
  constexpr int H = 128, F = 32, KS = H / 32;
  __shared__ __attribute__((aligned(16))) unsigned char hpl[2][3][16 * W8_ROWH];
  __shared__ __attribute__((aligned(16))) unsigned char xpl[PREIN ? 1 : 2][3][PREIN ? 16 : 16 * W8_ROWX];
  float* const pbuf = (AUX == 1) ? cs : zs;        // PREIN: where P lives

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, i = l & 15, g = l >> 4;
  const int b = blockIdx.x * 16 + i;
  const bool valid = !RAGGED || b < B;
  const int bc = valid ? b : B - 1;
  const int n0 = wv * 16 + g * 4;                  // this lane's 4 hidden units
  // mode bit 0: x is the trainer's [B,F,T].  AUX == 3 (FASTGRNN_FLAG_HS_LAST): hs is [B,H] and receives h_T alone
  // (a compile-time variant: as a run-time test in store_step it cost the plain AUX == 0 forward 8 %)
  const bool xbft = (mode & 1) != 0;
  constexpr bool hs_last = AUX == 3;
  // feature element this lane converts each step: utterance xu, feature xf
  const int xu = tid >> 5, xf = tid & 31;
  const int xb = blockIdx.x * 16 + xu;
  const int xbc = (!RAGGED || xb < B) ? xb : B - 1;

  const f32x4 bzv = ld4(bz + n0), bhv = ld4(bh + n0);
  f32x4 hown = ld4(h0 + (size_t)bc * H + n0);
  const float sz = fsigmoid(zeta[0]), sn = fsigmoid(nu[0]);

  // F16H: the state product U.h_{t-1} MAY run on fp16 two-plane operands (3 MFMAs per K-step instead of 6, two
  // state planes through LDS instead of three).  That needs |h| inside fp16's range for the whole scan.  The
  // host only instantiates F16H for gates with z in [0,1] (sigmoid, quantSigm, quantSigm4; a relu or tanh gate
  // does not bound h at all).  For those,  |h_t| <= z|h_{t-1}| + sigma(zeta)(1-z) + sigma(nu)
  //                                              <= max(|h_{t-1}|, sigma(zeta)) + sigma(nu),
  // i.e. |h_t| <= max(|h0|, 1) + t: the workgroup checks its own 16 rows of h0 against that bound here and
  // falls back to the three-bf16-plane product (no range limit) when a user-supplied h0 is too large (or NaN).
  // Two planes resolve 2^-22 of the value anywhere in range (lo subnormal below 2^-14: absolute 2^-24), and U is
  // pre-scaled per wave by an exact power of two that puts its largest element in [2^12, 2^13), undone by one
  // fma in the epilogue.  W.x keeps the three bf16 planes: x is unbounded user data.
  bool use_h16 = false;
  if constexpr (F16H) {
    __shared__ float hmax_s[8];
    float hm = fmaxf(fmaxf(fabsf(hown[0]), fabsf(hown[1])), fmaxf(fabsf(hown[2]), fabsf(hown[3])));
    if (!(hm == hm)) hm = 3.0e38f;                  // NaN in h0: take the path without a range assumption
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) hm = fmaxf(hm, __shfl_xor(hm, m));
    if (l == 0) hmax_s[wv] = hm;
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) hm = fmaxf(hm, hmax_s[k2]);
    use_h16 = __builtin_amdgcn_readfirstlane((int)(hm + (float)Tn + 2.0f < 3.0e4f)) != 0;
  }

  auto scan = [&](auto h16_tag) __attribute__((always_inline)) {
  constexpr bool H16 = decltype(h16_tag)::value;
  Frag3 Uf[KS], Wf;
  Frag2h Uh[KS];
  float u_unscale = 1.0f;
  {
    const int nA = wv * 16 + i;                    // A row i = unit 16w + i, K in natural unit order
    f32x4 ulo[KS], uhi[KS];
    float umax = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
      const float* up = u + (size_t)nA * H + 32 * s2 + 8 * g;
      ulo[s2] = ld4(up); uhi[s2] = ld4(up + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) umax = fmaxf(umax, fmaxf(fabsf(ulo[s2][j]), fabsf(uhi[s2][j])));
    }
    if (H16) {
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) umax = fmaxf(umax, __shfl_xor(umax, m));
      int e = 0;
      if (umax > 0.f && umax < 3.0e38f) (void)frexpf(umax, &e);      // umax = f * 2^e, f in [0.5, 1)
      e = e < -100 ? -100 : (e > 100 ? 100 : e);
      const float u_scale = ldexpf(1.0f, 13 - e);
      u_unscale = ldexpf(1.0f, e - 13);
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) Uh[s2] = split2h8(ulo[s2] * u_scale, uhi[s2] * u_scale);
    } else {
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) Uf[s2] = split3(ulo[s2], uhi[s2]);
    }
    if (!PREIN) {
      const float* wp = w + (size_t)nA * F + 8 * g;
      Wf = split3(ld4(wp), ld4(wp + 4));
    }
  }

  // planes of this lane's 4 state values -> 8 bytes per plane at [utterance i][unit n0]
  auto publish_h = [&](int buf) __attribute__((always_inline)) {
    const unsigned off = i * W8_ROWH + n0 * 2;
    if (H16) {
      uint2 hi, lo;
      split2h(hown[0], hown[1], hi.x, lo.x); split2h(hown[2], hown[3], hi.y, lo.y);
      *reinterpret_cast<uint2*>(&hpl[buf][0][off]) = hi;
      *reinterpret_cast<uint2*>(&hpl[buf][1][off]) = lo;
      return;
    }
    uint2 q0, q1, q2;
    split_quad(hown, q0, q1, q2);
    *reinterpret_cast<uint2*>(&hpl[buf][0][off]) = q0;
    *reinterpret_cast<uint2*>(&hpl[buf][1][off]) = q1;
    *reinterpret_cast<uint2*>(&hpl[buf][2][off]) = q2;
  };
  auto publish_x = [&](int buf, float v) __attribute__((always_inline)) {
    unsigned short s0, s1, s2;
    split_one(v, s0, s1, s2);
    const unsigned off = xu * W8_ROWX + xf * 2;
    *reinterpret_cast<unsigned short*>(&xpl[buf][0][off]) = s0;
    *reinterpret_cast<unsigned short*>(&xpl[buf][1][off]) = s1;
    *reinterpret_cast<unsigned short*>(&xpl[buf][2][off]) = s2;
  };
  // x is [T,B,F] / [B,T,F] (rsT, rsB) or, with FASTGRNN_FLAG_X_BFT, the trainer's [B,F,T]: this lane's value of
  // frame t is x[xbase + t * xstep]
  const size_t xbase = xbft ? ((size_t)xbc * F + xf) * Tn : (size_t)xbc * rsB * F + xf;
  const size_t xstep = xbft ? 1 : (size_t)rsT * F;
  auto load_x = [&](int t) __attribute__((always_inline)) {
    const size_t e = xbase + (size_t)t * xstep;
    return BF ? bf16_to_f32(reinterpret_cast<const unsigned short*>(x)[e]) : x[e];
  };
  auto load_p = [&](int t) __attribute__((always_inline)) {   // PREIN: this lane's four values of P_t
    return ld4(pbuf + ((size_t)t * rsT + (size_t)bc * rsB) * H + n0);
  };
  auto store_step = [&](int t, const f32x4 aux) __attribute__((always_inline)) {   // hown holds h_t
    if (hs_last && t != Tn - 1) return;               // (wave-uniform) the classifier reads h_T only: model.py:227
    if (RAGGED) lds_writes_landed();                  // (an exec-masked store block: no LDS write pending in front of it)
    if (valid) {
      const size_t o = hs_last ? (size_t)b * H + n0 : ((size_t)t * rsT + (size_t)b * rsB) * H + n0;
      if (BF) st4_bf16(reinterpret_cast<unsigned short*>(hs) + o, hown); else st4(hs + o, hown);
      if (AUX == 2) st4(zs + o, aux);               // the saved pre-activation stays fp32
    }
  };

  publish_h(0);
  float xnext = 0.f;
  f32x4 pnext = f32x4{0.f, 0.f, 0.f, 0.f};
  if (PREIN) {
    pnext = load_p(0);
  } else {
    publish_x(0, load_x(0));
    xnext = load_x(Tn > 1 ? 1 : 0);                // frame t+1, published during step t
  }
  __syncthreads();

  f32x4 aux_prev = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef FASTGRNN_DIAG_STAMPS
  unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dlast)::"memory");
#endif
  for (int t = 0; t < Tn; ++t) {
    const int cur = t & 1;
    SPLIT_STAMP(0)
    const float xpub = xnext;
    const f32x4 pcur = pnext;
    if (PREIN) pnext = load_p(t + 1 < Tn ? t + 1 : Tn - 1);
    else xnext = load_x(t + 2 < Tn ? t + 2 : Tn - 1);
    Frag3 xB, hB[KS];
    Frag2h hH[KS];
    if (!PREIN) {
#pragma unroll
      for (int p = 0; p < 3; ++p) xB.p[p] = *reinterpret_cast<const u32x4*>(&xpl[cur][p][i * W8_ROWX + 16 * g]);
    }
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
      if (H16) {
        hH[s2].hi = *reinterpret_cast<const u32x4*>(&hpl[cur][0][i * W8_ROWH + 64 * s2 + 16 * g]);
        hH[s2].lo = *reinterpret_cast<const u32x4*>(&hpl[cur][1][i * W8_ROWH + 64 * s2 + 16 * g]);
      } else {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          hB[s2].p[p] = *reinterpret_cast<const u32x4*>(&hpl[cur][p][i * W8_ROWH + 64 * s2 + 16 * g]);
      }
    }
    if (t > 0) store_step(t - 1, aux_prev);          // h_{t-1} and its pre-activation: issued during the LDS round trip
    if (!PREIN) publish_x(cur ^ 1, xpub);
    // every fragment read is ISSUED -- so each has registers of its own -- before the first MFMA that reads
    // one (two waves per SIMD: see bwd_scan_split_w8::weight_grads for why the compiler must not stream
    // them through fewer registers); the waits for them stay progressive
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STAMP(1)
    // big and small terms in accumulators of their own (see mfma6_hl), summed once at the end
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, alo = a;
    if (PREIN) {
      a = pcur;
    } else if (BF) {                                 // a bf16 frame is its own first plane: three of the six terms, same bits
      alo = mfma_bf16(Wf.p[2], xB.p[0], alo);
      alo = mfma_bf16(Wf.p[1], xB.p[0], alo);
      a = mfma_bf16(Wf.p[0], xB.p[0], a);
    } else {
      mfma6_hl(Wf, xB, a, alo);
    }
    if (H16) {
      f32x4 ah = f32x4{0.f, 0.f, 0.f, 0.f}, ahlo = ah;
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) mfma3h_hl(Uh[s2], hH[s2], ah, ahlo);               // .cu:368, scaled by 2^k
      a = (a + alo) + (ah + ahlo) * u_unscale;
    } else {
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) mfma6_hl(Uf[s2], hB[s2], a, alo);                  // .cu:368
      a += alo;
    }
    f32x4 zq, cq;
    SPLIT_STAMP(2)
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                                        // .cu:55-58
      const float z = gate_act<GATE>(a[r] + bzv[r]);
      const float c = UQ ? fminf(fmaxf(a[r] + bhv[r], -1.0f), 1.0f) : ftanh(a[r] + bhv[r]);
      hown[r] = (sz * (1.0f - z) + sn) * c + hown[r] * z;
      zq[r] = z; cq[r] = c;
    }
    if (AUX == 1 && RAGGED) lds_writes_landed();
    if (AUX == 1 && valid) {                          // reference operator outputs: stored at once
      const size_t o = ((size_t)t * rsT + (size_t)b * rsB) * H + n0;
      st4(zs + o, zq); st4(cs + o, cq);
    }
    aux_prev = a;
    publish_h(cur ^ 1);
    SPLIT_STAMP(3)
    lds_barrier();
    SPLIT_STAMP(4)
  }
#ifdef FASTGRNN_DIAG_STAMPS
  if (blockIdx.x == 7 && l == 0) { for (int k2 = 0; k2 < 8; ++k2) g_sdiag[wv][k2] = dsum[k2]; }
#endif
  store_step(Tn - 1, aux_prev);
  };
  if constexpr (F16H) {
    if (use_h16) scan(std::true_type{}); else scan(std::false_type{});
  } else {
    scan(std::false_type{});
  }
}


// ------------------------------------------------------------------------------------------
// backward  (H = 128, F = 32)
// ------------------------------------------------------------------------------------------

// floats per workgroup slab of backward partial sums, padded to 64: dU | dW | d_bz | d_bh | (zeta, nu)
constexpr int SLAB = (128 * 128 + 128 * 32 + 2 * 128 + 2 + 63) & ~63;

// row (t, b) of a [T,B,*] / [B,T,*] tensor = t * row_stride_t + b * row_stride_b
static inline int row_stride_t(const fastgrnn_desc& d) { return (d.flags & FASTGRNN_FLAG_BATCH_MAJOR) ? 1 : d.B; }
static inline int row_stride_b(const fastgrnn_desc& d) { return (d.flags & FASTGRNN_FLAG_BATCH_MAJOR) ? d.T : 1; }

constexpr int ROW_H = 288;             // bytes per utterance row of a 128-wide bf16 plane (256 + 32 pad)
constexpr int ROW_X = 96;              // bytes per utterance row of a 32-wide bf16 plane (64 + 32 pad)
constexpr int PLANE_H = 16 * ROW_H;    // 4608
constexpr int PLANE_X = 16 * ROW_X;    // 1536
constexpr int IMG = 2 * 3 * PLANE_H + 3 * PLANE_X;   // one step's images: d_pre | h_prev | x  (32256 B)

constexpr int OFF_DP = 0, OFF_HP = 3 * PLANE_H, OFF_XP = 6 * PLANE_H;

// ------------------------------------------------------------------------------------------
// backward, 8 waves  (H = 128, F = 32)
// ------------------------------------------------------------------------------------------
// Reverse scan on a workgroup of 8 waves (two per SIMD), so that a wave stalled on a dependent MFMA, an LDS
// round trip or a transcendental leaves its SIMD to the other wave, and no wave needs more than 256
// registers (no AGPR<->VGPR traffic).  PREACT: aux0 holds the pre-activation W x + U h saved by the forward
// (one tensor) and z, c are recomputed here; otherwise aux0 = z_s, aux1 = h_prime_s (the reference
// operator's tensors).  (An earlier 4-wave shape, one wave per SIMD with every fragment streamed through a
// two-deep register pipeline, was 9-11 % slower and is gone: it could not be written to the operand rule of
// DESIGN.md section 4.0.)
//   recurrence   wave w owns ONE 16-unit row tile (units 16w..16w+15, 4 per lane): chain of 24 MFMAs,
//                EW of 4 elements, planes of its d_pre / h_prev slice and of one feature value per lane
//   d_x          wave w: feature tile w&1, K-step w>>1 (6 MFMAs); the four partials of a tile meet in
//                LDS an iteration later (waves 0, 1 store)
//   dW / dU      each step pair's 80 tiles = 4 row-tile pairs x 2 column halves: wave w owns row tiles
//                2(w&3), 2(w&3)+1 and column tiles 5(w>>2)..+4.  Waves 0-3 contract pair (t+1, t) on even
//                t, waves 4-7 pair (t+2, t+1) on odd t: every iteration each SIMD has one wave with 60
//                independent MFMAs in flight and one with only the recurrence.
struct BwdW8Lds {
  // Per step s, buffer s&3: natural [utterance][unit] images of the exact bf16 planes of d_pre_s,
  // h_{s-1} and x_s.  Each is read two ways: as MFMA B fragments (ds_read_b128: 8 consecutive units
  // of one utterance, the chain) and transposed (ds_read_b64_tr_b16: 8 consecutive utterances of
  // one unit) for the K = utterance products dW, dU.  The MFMA's K is 32 and a tile has 16
  // utterances, so dW/dU contract TWO consecutive steps per MFMA (lane groups 0,1 = step s+1,
  // groups 2,3 = step s); four buffers keep a pair readable for the two iterations that share it.
  unsigned char img[4][IMG];
  f32x4 DX[2][8][64];                  // d_x partial of wave w: feature tile w&1, K-step w>>1
  float bias[2][128];                  // bias_gate | bias_update (PREACT: the gates are recomputed)
  float red[16];
};

// NOX: the layer's input is not 32 wide (the reference's second layer: F = 256, model.py:196-203).  Everything that
// involves W or x leaves the scan: it writes d_pre[T,B,H] (fp32, through the d_x pointer) and the batched GEMMs of
// kernels_gemm.hip produce  dW = d_pre^T X  and  d_x = d_pre W  afterwards (.cu:538-539 does both per step).  The
// recurrence, dU (eight column tiles, four per column half) and the bias / zeta / nu sums stay as they are.
// UQ: update nonlinearity quantTanh (see fwd_scan_split_w8); PREACT contract only.
template <int GATE, bool PREACT, bool RAGGED, bool BF = false, bool NOX = false, bool UQ = false>
__global__ __launch_bounds__(512) void bwd_scan_split_w8(
    int Tn, int B, int rsT, int rsB, int mode, const float* __restrict__ ghs, const float* __restrict__ x,
    const float* __restrict__ hs, const float* __restrict__ aux0, const float* __restrict__ aux1,
    const float* __restrict__ h0, const float* __restrict__ w, const float* __restrict__ u,
    const float* __restrict__ bz, const float* __restrict__ bh,
    const float* __restrict__ zeta, const float* __restrict__ nu,
    float* __restrict__ d_x, float* __restrict__ d_h0, float* __restrict__ part) {
  constexpr int H = 128, F = 32, KS = 4, NFT = NOX ? 0 : 2, NC = NOX ? 4 : 5;   // NC column tiles per column half
  __shared__ BwdW8Lds S;
#ifdef FASTGRNN_DIAG_STAMPS
  unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dlast)::"memory");
#endif

  // mode bit 0: x / d_x are the trainer's [B,F,T]; bit 1 (FASTGRNN_FLAG_GRAD_LAST): ghs is [B,H], the gradient of
  // the last state alone (the classifier head reads hs[T-1] only, model.py:227) -- every other step's is zero
  const bool xbft = (mode & 1) != 0, g_last = (mode & 2) != 0;
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, i = l & 15, g = l >> 4;
  const int b = blockIdx.x * 16 + i;
  const bool valid = !RAGGED || b < B;
  const int bc = valid ? b : B - 1;
  const int n0 = wv * 16 + g * 4;                    // this lane's 4 hidden units
  const float sz = fsigmoid(zeta[0]), sn = fsigmoid(nu[0]);
  // Odd T: step T-1 has no partner; it is paired with a virtual step T whose images are zero.
  if (Tn & 1) {
    for (int idx = tid; idx < IMG / 4; idx += 512) reinterpret_cast<unsigned*>(&S.img[Tn & 3][0])[idx] = 0u;
  }
  if (PREACT && tid < 256) S.bias[tid >> 7][tid & 127] = (tid < 128 ? bz : bh)[tid & 127];

  // ---- resident A operands ------------------------------------------------------------------
  // chain: d_h[k][b] = sum_n U[n][k] d_pre[b][n]; A row i is k = 16w + i, K in natural unit order
  Frag3 UTf[KS];
#pragma unroll
  for (int s2 = 0; s2 < KS; ++s2) {
    f32x4 lo, hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo[j] = u[(size_t)(32 * s2 + 8 * g + j) * H + wv * 16 + i];
      hi[j] = u[(size_t)(32 * s2 + 8 * g + 4 + j) * H + wv * 16 + i];
    }
    UTf[s2] = split3(lo, hi);
  }
  // d_x[f][b] = sum_n W[n][f] d_pre[b][n]: feature tile xf2, K-step xks
  const int xf2 = wv & 1, xks = wv >> 1;
  Frag3 WTf;
  if (!NOX) {
    f32x4 lo, hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo[j] = w[(size_t)(32 * xks + 8 * g + j) * F + xf2 * 16 + i];
      hi[j] = w[(size_t)(32 * xks + 8 * g + 4 + j) * F + xf2 * 16 + i];
    }
    WTf = split3(lo, hi);
  }
  f32x4 sbz = f32x4{0.f, 0.f, 0.f, 0.f}, sbh = sbz, dh = sbz;
  float pz = 0.f, pn = 0.f, pz_c = 0.f, pn_c = 0.f;     // d_zeta / d_nu partial sums, compensated
  auto role_body = [&](auto role_tag) __attribute__((always_inline)) {
  // dW / dU accumulators: row tiles 2rp + a, column tiles 5ch + c  (column tile 0,1 = dW, 2..9 = dU)
  const int rp = wv & 3;
  constexpr int ch = decltype(role_tag)::value;   // column half = wv >> 2, a compile-time constant per code path
  f32x4 acc[2][NC];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane-constant LDS byte offsets (within one step's image block)
  const unsigned lds_img = (unsigned)(size_t)&S.img[0][0];
  const unsigned my_row_h = (unsigned)(i * ROW_H + n0 * 2);          // 4 units of utterance i: 8 bytes per plane
  const int xu = tid >> 5, xf = tid & 31;                            // the feature value this lane converts
  const unsigned my_x = (unsigned)(xu * ROW_X + xf * 2);
  const int xbb = blockIdx.x * 16 + xu;
  const int xbc = (!RAGGED || xbb < B) ? xbb : B - 1;
  const int q = (l & 15) >> 2, pp = l & 3;
  const unsigned trA_off = OFF_DP + (8 * (g & 1) + q) * ROW_H + (rp * 32 + 4 * pp) * 2;   // d_pre^T rows 32rp + 16a + i
  const unsigned trH_off = OFF_HP + (8 * (g & 1) + q) * ROW_H + (4 * pp) * 2;             // h_prev^T column 16c + i
  const unsigned trX_off = OFF_XP + (8 * (g & 1) + q) * ROW_X + (4 * pp) * 2;             // x^T column 16f + i

  // all fragment reads of a phase are issued (hence: each in registers of its own) before its first MFMA
  auto fragments_landed = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  // grad_hs, aux0 (z or pre), aux1 (c), h_prev; one x value.  bf16 sequences stay RAW (packed) until they are
  // used an iteration later: unpacking at load time would make every request synchronous.
  struct EwOps { f32x4 g, a0, a1, h; float xv; uint2 graw, hraw; unsigned short xraw; };
  // Addresses: a wave-uniform step base (scalar registers) + a 32-bit BYTE lane offset (global_load ... v_off, s[base]):
  // one VGPR per stream instead of a 64-bit pointer pair and its per-step arithmetic.  split_supported() keeps the
  // whole sequence tensor below 2^32 bytes for this.
  constexpr unsigned ESZ = BF ? 2u : 4u;             // bytes per sequence element (x, hs, grad_hs)
  const unsigned lane_e = (unsigned)bc * (unsigned)rsB * H + n0;          // element offset of this lane inside a step's rows
  const unsigned lane_bh = (unsigned)bc * H + n0;                         // ... inside a [B,H] tensor
  // NOX: byte offset of this lane's d_pre values inside a step's rows and the byte stride between steps (0 for the sink)
  const unsigned dp_step = (NOX && valid) ? (unsigned)rsT * H * 4u : 0u;
  const unsigned dp_off = !NOX ? 0u : (valid ? ((unsigned)b * (unsigned)rsB * H + n0) * 4u
                                             : (((unsigned)Tn * (unsigned)B + i) * H + n0) * 4u);
  const unsigned lane_x = NOX ? 0u : (xbft ? ((unsigned)xbc * F + xf) * (unsigned)Tn : (unsigned)xbc * (unsigned)rsB * F + xf);
  auto ldg4 = [](const void* base, unsigned byte_off) __attribute__((always_inline)) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  auto ldg2 = [](const void* base, unsigned byte_off) __attribute__((always_inline)) {
    return *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  auto load_ew = [&](int t, EwOps& e) __attribute__((always_inline)) {
    const size_t step_h = (size_t)t * rsT * H;       // uniform: element offset of step t's rows
    e.a0 = ldg4(aux0 + step_h, lane_e * 4u);
    if (!PREACT) e.a1 = ldg4(aux1 + step_h, lane_e * 4u);
    const bool g_zero = g_last && t != Tn - 1;       // wave-uniform
    const char* gbase = g_last ? reinterpret_cast<const char*>(ghs) : reinterpret_cast<const char*>(ghs) + step_h * ESZ;
    const unsigned goff = (g_last ? lane_bh : lane_e) * ESZ;
    const char* hbase = (t == 0) ? reinterpret_cast<const char*>(hs)
                                 : reinterpret_cast<const char*>(hs) + (step_h - (size_t)rsT * H) * ESZ;   // .cu:478-481
    const char* xbase = xbft ? reinterpret_cast<const char*>(x) + (size_t)t * ESZ
                             : reinterpret_cast<const char*>(x) + (size_t)t * rsT * F * ESZ;
    // (the gradient is requested unconditionally and replaced by zeros where it does not apply: a load inside the
    // conditional would put a wave-uniform branch between the step's LDS plane writes and these requests -- DESIGN.md
    // 4.0, second rule)
    if (BF) {                                        // h0 and the saved tensor are fp32; EW(0) fetches h0 itself
      const uint2 graw = ldg2(gbase, goff);
      e.graw = g_zero ? uint2{0u, 0u} : graw;
      if (RAGGED && !valid) e.graw = uint2{0u, 0u};
      e.hraw = ldg2(hbase, lane_e * ESZ);            // (t == 0: a dummy row of hs; h0 below is what EW(0) uses)
      // h_prev of step 0 is the fp32 h0: requested HERE, with the step's other operands (load_ew runs behind a
      // completion read), not in unpack_ew, which sits between the chain's MFMAs and the first read of their result
      if (t == 0) e.h = ldg4(h0, lane_bh * 4u);
      if (!NOX) e.xraw = *reinterpret_cast<const unsigned short*>(xbase + lane_x * ESZ);
    } else {
      const f32x4 gv = ldg4(gbase, goff);
      e.g = g_zero ? f32x4{0.f, 0.f, 0.f, 0.f} : gv;
      // Lanes beyond a ragged batch re-read the last utterance's (finite) rows with a ZERO gradient: with dh = 0
      // at the start, gg, d_pre and every sum they enter stay exactly zero for them -- nothing else to mask.
      if (RAGGED && !valid) e.g = f32x4{0.f, 0.f, 0.f, 0.f};
      e.h = (t == 0) ? ldg4(h0, lane_bh * 4u) : ldg4(hbase, lane_e * 4u);
      if (!NOX) e.xv = *reinterpret_cast<const float*>(xbase + lane_x * 4u);
    }
  };
  auto unpack4 = [](const uint2 v) __attribute__((always_inline)) {
    return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xFFFF0000u),
                 __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xFFFF0000u)};
  };
  // the operands of EW(t) as fp32 (a no-op for fp32 sequences)
  auto unpack_ew = [&](int t, EwOps& e) __attribute__((always_inline)) {
    if (BF) {
      e.g = unpack4(e.graw);
      if (t != 0) e.h = unpack4(e.hraw);                 // (t == 0: load_ew fetched the fp32 h0)
      if (!NOX) e.xv = bf16_to_f32(e.xraw);
    }
  };
  struct EwPre { f32x4 kc, kz, z, c; };
  auto ew_pre = [&](const EwOps& e, EwPre& f) __attribute__((always_inline)) {
    f32x4 bzq = f32x4{0.f, 0.f, 0.f, 0.f}, bhq = bzq;
    if (PREACT) {
      bzq = *reinterpret_cast<const f32x4*>(&S.bias[0][n0]);
      bhq = *reinterpret_cast<const f32x4*>(&S.bias[1][n0]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float z, c;
      if (PREACT) {
        z = gate_act<GATE>(e.a0[r] + bzq[r]);
        c = UQ ? fminf(fmaxf(e.a0[r] + bhq[r], -1.0f), 1.0f) : ftanh(e.a0[r] + bhq[r]);
      } else {
        z = e.a0[r]; c = e.a1[r];
      }
      const float dc = UQ ? ((c < 1.0f && c > -1.0f) ? 1.0f : 0.0f) : 1.0f - c * c;      // update nonlinearity's derivative
      f.kc[r] = (sz * (1.0f - z) + sn) * dc;                    // d_pre_c = kc * gg   (.cu:109)
      f.kz[r] = (e.h[r] - sz * c) * gate_dact<GATE>(z);         // d_pre_z = kz * gg   (.cu:110)
      f.z[r] = z; f.c[r] = c;
    }
  };
  // planes of 4 fp32 values -> 8 bytes per plane at byte offset off of the three planes of one image part
  auto put4 = [&](unsigned char* base, int plane_stride, unsigned off, const f32x4 v) __attribute__((always_inline)) {
    uint2 q0, q1, q2;
    split_quad(v, q0, q1, q2);
    *reinterpret_cast<uint2*>(base + off) = q0;
    *reinterpret_cast<uint2*>(base + plane_stride + off) = q1;
    *reinterpret_cast<uint2*>(base + 2 * plane_stride + off) = q2;
  };
  // EW(t) second half + everything step t publishes: planes of d_pre_t, h_{t-1} (own 4 units) and of one
  // value of x_t.  Leaves dh = z*g, the C-in of chain(t).
  auto ew_post = [&](int t, const EwOps& e, const EwPre& f, const f32x4 ggv) __attribute__((always_inline)) {
    f32x4 dpv;
    float sn4 = 0.f, sz4 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gg = ggv[r];                                         // grad + d_old_h  (.cu:474)
      const float dcp = f.kc[r] * gg, dzp = f.kz[r] * gg;
      sbz[r] += dzp; sbh[r] += dcp;
      const float cg = f.c[r] * gg;
      if constexpr (NOX) { sn4 += cg; sz4 += cg - f.z[r] * cg; }       // .cu:114-115
      else { pn += cg; pz += cg - f.z[r] * cg; }
      dpv[r] = dzp + dcp;                                              // .cu:113
      dh[r] = f.z[r] * gg;                                             // .cu:108
    }
    // compensated only where registers are to spare (the F = 32 build of this kernel has none: there the plain sum,
    // 1.2e-5 of the result at B = 4096, stays)
    if constexpr (NOX) { kahan_add(pn, pn_c, sn4); kahan_add(pz, pz_c, sz4); }
    unsigned char* im = &S.img[t & 3][0];
    put4(im + OFF_DP, PLANE_H, my_row_h, dpv);
    put4(im + OFF_HP, PLANE_H, my_row_h, e.h);
    if (NOX) {
      // d_pre_t for the weight-gradient / d_x GEMMs (fp32, [T,B,H] layout of hs).  EVERY lane stores, without a
      // branch: lanes beyond a ragged batch write to the 16 sink rows behind the T*B rows (step stride 0).  The
      // first form, "if (valid) store" -- an exec-masked block with its own address arithmetic between the plane
      // writes above and the next step's loads -- gave rare wrong results in whole workgroups (all of them full
      // ones) that went away with idle cycles on either side of the block; the cause is not understood, so the
      // steady state of this scan has no conditional code around memory instructions at all.
      *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(d_x) + (size_t)t * dp_step + dp_off) = dpv;
    } else {
      unsigned short s0, s1, s2;
      split_one(e.xv, s0, s1, s2);
      *reinterpret_cast<unsigned short*>(im + OFF_XP + my_x) = s0;
      *reinterpret_cast<unsigned short*>(im + OFF_XP + PLANE_X + my_x) = s1;
      *reinterpret_cast<unsigned short*>(im + OFF_XP + 2 * PLANE_X + my_x) = s2;
    }
  };
  auto finish_dx = [&](int t) __attribute__((always_inline)) {
    if (!NOX && wv < NFT) {                         // wave-uniform: feature tile wv = sum over the four K-steps
      const f32x4 sacc = (S.DX[t & 1][wv][l] + S.DX[t & 1][wv + 2][l]) + (S.DX[t & 1][wv + 4][l] + S.DX[t & 1][wv + 6][l]);
      if (xbft) {                                    // d_x in the trainer's [B,F,T]: features are T apart
        if (valid) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const size_t o = ((size_t)b * F + wv * 16 + 4 * g + r) * Tn + t;
            if (BF) reinterpret_cast<unsigned short*>(d_x)[o] = (unsigned short)f32_to_bf16_rne(sacc[r]);
            else d_x[o] = sacc[r];
          }
        }
      } else {
        char* xo = reinterpret_cast<char*>(d_x) + (size_t)t * rsT * F * ESZ;                 // uniform step base
        const unsigned lo = ((unsigned)b * (unsigned)rsB * F + wv * 16 + 4 * g) * ESZ;
        if (valid) { if (BF) st4_bf16(xo + lo, sacc); else *reinterpret_cast<f32x4*>(xo + lo) = sacc; }
      }
    }
  };
  // dW += d_pre_s^T x_s, dU += d_pre_s^T h_{s-1} (.cu:539-540) for the step pair (sU, sU-1): this wave's
  // 2 row tiles x 5 column tiles, 6 terms each.  Fragments come out of the plane images through the
  // hardware transpose read.
  // PURE3 (compile time): bf16 sequences and a pair beyond step 0 -- x and h_prev are their own first planes, the other two
  // are zero: three of mfma6's six terms, in its order (the same bits).  The pair (1, 0) reads the fp32 h0 and keeps six.
  // (Decided per call site, not by a branch on sU: a wave-uniform branch between the MFMAs of a scan step brought back
  // intermittent wrong d_x rows -- 2 of 12 test processes -- that the static scans do not see: DESIGN.md 4.0.)
  auto weight_grads = [&](auto pure_tag, int sU) __attribute__((always_inline)) {
    constexpr bool PURE3 = decltype(pure_tag)::value;
    const unsigned im = lds_img + (unsigned)(((g < 2) ? sU : sU - 1) & 3) * IMG;
    const unsigned trA = im + trA_off, trH = im + trH_off, trX = im + trX_off;
    Frag3 Af[2];
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) Af[a2].p[pl] = tr_frag(trA + pl * PLANE_H + a2 * 32, ROW_H);
    auto load_b = [&](int c, Frag3& bf) __attribute__((always_inline)) {
      const int ct = NC * ch + c;                      // wave-uniform
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        bf.p[pl] = (ct < NFT) ? tr_frag(trX + pl * PLANE_X + ct * 32, ROW_X)
                              : tr_frag(trH + pl * PLANE_H + (ct - NFT) * 32, ROW_H);
    };
    // All fragment reads of a batch are issued (each into registers of its own) BEFORE the first MFMA that
    // reads one, and a batch's registers are reloaded only after its last MFMA has RETIRED (a VALU read of the
    // youngest accumulators waits for it).  With two waves per SIMD an issued MFMA can sit behind the other
    // wave's MFMAs and fetches its operands when it starts: a ds_read into a fragment right after the MFMAs
    // that read it -- what the compiler emits when left alone -- gave rare run-to-run differences
    // (tools/check_determinism.py; tests/test_hip_parity.py checks repeatability).
    auto batch = [&](auto c0_tag, auto n_tag) __attribute__((always_inline)) {
      constexpr int C0 = decltype(c0_tag)::value, NB = decltype(n_tag)::value;
      Frag3 Bf[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) load_b(C0 + c, Bf[c]);
      fragments_landed();
      if constexpr (PURE3) {
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2) {
            f32x4 v = mfma_bf16(Af[a2].p[2], Bf[c].p[0], acc[a2][C0 + c]);
            v = mfma_bf16(Af[a2].p[1], Bf[c].p[0], v);
            acc[a2][C0 + c] = mfma_bf16(Af[a2].p[0], Bf[c].p[0], v);
          }
      } else {
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2) acc[a2][C0 + c] = mfma6(Af[a2], Bf[c], acc[a2][C0 + c]);
      }
      __builtin_amdgcn_sched_barrier(0);       // (the scheduler otherwise sinks MFMAs below the read)
      float touch = 0.f;                       // one element of every accumulator of the batch: all have retired
#pragma unroll
      for (int c = 0; c < NB; ++c) touch += acc[0][C0 + c][0] + acc[1][C0 + c][0];
      completion_read(touch);
      __builtin_amdgcn_sched_barrier(0);
    };
    // batches of 2 + 2 + 1 column tiles: with 3 + 2 the kernel spilled nine registers inside the loop
    // (the tanh gate recomputed from the pre-activation needs a few registers more around this phase: with
    // two-tile batches it spilled the next step's operands inside the loop; one tile per batch there)
    constexpr bool ONE_TILE_BATCHES = PREACT && GATE == FASTGRNN_NL_TANH && !NOX;
    if constexpr (ONE_TILE_BATCHES) {
      static_for<NC>([&](auto c_tag) __attribute__((always_inline)) { batch(c_tag, std::integral_constant<int, 1>{}); });
    } else {
      batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
      batch(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
      if constexpr (NC == 5) batch(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
    }
  };

  const int top = (Tn & 1) ? Tn : Tn - 1;          // highest (possibly virtual) step: pairs are (odd, even)
  // One iteration.  eo = operands of EW(t-1) (requested an iteration ago); once EW(t-1) has consumed them the same
  // registers receive the requests for EW(t-2) (ONE operand set: two sets alternating over the unrolled loop cost
  // the compiler ~40 registers more, i.e. spills inside the loop).
  auto iter = [&](auto last_tag, auto even_tag, int t, EwOps& eo) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value, EVEN = decltype(even_tag)::value;
    SPLIT_STAMP(0)
    constexpr bool MY_TURN = EVEN ? (ch == 0) : (ch == 1);
    const bool heavy = MY_TURN && (EVEN || t + 2 <= top);
    // (tried: s_setprio 1 / 3 from here to the weight gradients, 0 there -- no change in three interleaved runs)
    const unsigned char* im = &S.img[t & 3][0];
    Frag3 dB[KS];
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        dB[s2].p[p] = *reinterpret_cast<const u32x4*>(im + OFF_DP + p * PLANE_H + i * ROW_H + (32 * s2 + 8 * g) * 2);
    if (t + 1 < Tn) finish_dx(t + 1);               // partials published in the previous iteration
    // All twelve fragment reads are issued, each into registers of its own, before the first MFMA that reads
    // one: the compiler would otherwise stream them through fewer registers (a ds_read into a fragment
    // right after the MFMAs that read it), which is not safe here -- see weight_grads.
    fragments_landed();
    EwPre f;
    auto chain = [&]() __attribute__((always_inline)) {
      // d_h chain (.cu:537), C-in = z*g; d_x partial of step t (.cu:538) from the same fragments
      const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 dlo = z4;
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) mfma6_hl(UTf[s2], dB[s2], dh, dlo);
      dh += dlo;
      // this sum is the read that proves the chain has retired: pinned here (in the last iteration nothing else
      // needs dh before the flush, and the optimiser would sink the add below the next fragment loads)
      asm volatile("" : "+v"(dh));
      // (a runtime index would put the fragments in scratch.)  Every arm issues the product -- the last one
      // unconditionally: xks < KS -- and the partial is stored OUTSIDE the selection: that store is the read which
      // proves these MFMAs, the youngest of the step, have retired (tools/war_scan.py follows every feasible path).
      if constexpr (!NOX) {
        f32x4 dxp;
        static_assert(KS == 4, "four K-steps");
        if (xks == 0) dxp = mfma6(WTf, dB[0], z4);
        else if (xks == 1) dxp = mfma6(WTf, dB[1], z4);
        else if (xks == 2) dxp = mfma6(WTf, dB[2], z4);
        else dxp = mfma6(WTf, dB[3], z4);
        S.DX[t & 1][wv][l] = dxp;
      }
    };
    // Same order in both waves of a SIMD.  (Measured, tools/mfma_share_probe.hip: two waves with MFMAs ready
    // do not interleave on the matrix pipe -- one streams at 16.7 cycles per MFMA, the other waits -- and a
    // complementary order, EW first in the wave whose turn it is not, was 6 % slower in the same run.)
    chain();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LAST) { unpack_ew(t - 1, eo); ew_pre(eo, f); }
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STAMP(1)
    // reading dh (.cu:474) also means every chain MFMA has retired before a later load reuses a fragment register
    if constexpr (!LAST) {
      ew_post(t - 1, eo, f, eo.g + dh);
      __builtin_amdgcn_sched_barrier(0);
      // requests for EW(t-2), after the chain has retired (they may land in registers its fragments used).  bf16
      // sequences: the request for the fp32 h0 sits in a wave-uniform branch -- the plane writes above have landed first
      if (BF) lds_writes_landed();
      load_ew(t >= 2 ? t - 2 : 0, eo);
    }
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STAMP(2)
    // this wave's share of a step pair's dW / dU: 60 independent MFMAs and no VALU work
    if (heavy) weight_grads(std::bool_constant<BF && !LAST>{}, EVEN ? t + 1 : t + 2);
    SPLIT_STAMP(3)
    lds_barrier();
    SPLIT_STAMP(4)
  };

  EwOps E;
  __syncthreads();                                  // bias staged, virtual-step images zeroed
  {
    EwPre f;
    load_ew(Tn - 1, E);
    unpack_ew(Tn - 1, E); ew_pre(E, f); ew_post(Tn - 1, E, f, E.g);
    __builtin_amdgcn_sched_barrier(0);
    if (Tn >= 2) load_ew(Tn - 2, E);
  }
  __syncthreads();
  {
    int t = Tn - 1;
    if (t & 1) { iter(std::false_type{}, std::false_type{}, t, E); --t; }
    for (; t >= 2; t -= 2) {
      iter(std::false_type{}, std::true_type{}, t, E);
      iter(std::false_type{}, std::false_type{}, t - 1, E);
    }
    iter(std::true_type{}, std::true_type{}, 0, E);
  }
#ifdef FASTGRNN_DIAG_STAMPS
  if (blockIdx.x == 7 && l == 0) { for (int k2 = 0; k2 < 8; ++k2) g_sdiag[wv][k2] = dsum[k2]; }
#endif
  if constexpr (ch == 1) weight_grads(std::false_type{}, 1);   // pair (1, 0), column tiles NC..2NC-1
  finish_dx(0);
  // ---- flush ---------------------------------------------------------------------------------
  if (valid) st4(d_h0 + (size_t)b * H + n0, dh);
  {
    float* pu = part + (size_t)blockIdx.x * SLAB;
    float* pw = pu + H * H;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = rp * 32 + a * 16 + 4 * g + r;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int ct = NC * ch + c;                // wave-uniform
          if (ct < NFT) pw[(size_t)n * F + ct * 16 + i] = acc[a][c][r];
          else pu[(size_t)n * H + (ct - NFT) * 16 + i] = acc[a][c][r];
        }
      }
  }
  };
  if (wv < 4) role_body(std::integral_constant<int, 0>{}); else role_body(std::integral_constant<int, 1>{});
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float a = sbz[r], c = sbh[r];
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); c += __shfl_xor(c, m); }
    if (i == 0) {
      float* pb = part + (size_t)blockIdx.x * SLAB + H * H + H * F;
      pb[n0 + r] = a;
      pb[H + n0 + r] = c;
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { pz += __shfl_xor(pz, m); pn += __shfl_xor(pn, m); }
  if (l == 0) { S.red[wv] = pz; S.red[8 + wv] = pn; }
  __syncthreads();
  if (tid == 0) {
    float* pzn = part + (size_t)blockIdx.x * SLAB + H * H + H * F + 2 * H;
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a += S.red[k]; c += S.red[8 + k]; }
    pzn[0] = a; pzn[1] = c;
  }
}

// Deterministic reduction of the per-workgroup slabs in ONE launch (same scheme as kernels_mfma.hip).
__global__ __launch_bounds__(1024) void reduce_slabs_split(int nwg, const float* __restrict__ part,
                                                           const float* __restrict__ zeta, const float* __restrict__ nu,
                                                           float* __restrict__ d_u, float* __restrict__ d_w,
                                                           float* __restrict__ d_bz, float* __restrict__ d_bh,
                                                           float* __restrict__ d_zeta, float* __restrict__ d_nu) {
  constexpr int H = 128, F = 32;
  __shared__ float sm[16][64];
  const int o = threadIdx.x & 63, part_id = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;
  const int ntot = H * H + H * F + 2 * H + 2;
  float a = 0.f;
  if (idx < ntot) {
    float v[8];
    for (int wg0 = part_id; wg0 < nwg; wg0 += 16 * 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int wg = wg0 + 16 * j;
        v[j] = wg < nwg ? part[(size_t)wg * SLAB + idx] : 0.f;
      }
      a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
  }
  sm[part_id][o] = a;
  __syncthreads();
  if (part_id == 0 && idx < ntot) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += sm[j][o];
    const int oW = H * H, oBz = oW + H * F, oBh = oBz + H, oZ = oBh + H;
    if (idx < oW) d_u[idx] = t;
    else if (idx < oBz) { if (d_w) d_w[idx - oW] = t; }     // (NOX scans leave dW to the TN GEMM: d_w == nullptr)
    else if (idx < oBh) d_bz[idx - oBz] = t;
    else if (idx < oZ) d_bh[idx - oBh] = t;
    else if (idx == oZ) { const float sz = 1.0f / (1.0f + expf(-zeta[0])); d_zeta[0] = t * sz * (1.0f - sz); }   // .cu:116,544
    else { const float sn = 1.0f / (1.0f + expf(-nu[0])); d_nu[0] = t * sn * (1.0f - sn); }                      // .cu:117,545
  }
}

// Dense H = 128 layers whose input is not 32 wide (F = 64 / 128 / 256: the reference's second layer, model.py:196-203):
// the scans keep the recurrence only, the frame products are batched GEMMs (kernels_gemm.hip).
bool dense_wide_shape(const fastgrnn_desc& d) {
  return d.w_rank == 0 && d.u_rank == 0 && d.H == 128 && (d.F == 64 || d.F == 128 || d.F == 256) &&
         // 32-bit byte offsets inside a tensor (see split_supported), the d_pre workspace's 16 sink rows included
         ((double)d.T * d.B + 16.0) * (d.F > 128 ? d.F : 128) * 4.0 < 4294967296.0;
}
struct WideBwdWs { size_t slabs, dpre, tn, total; };
WideBwdWs wide_bwd_layout(const fastgrnn_desc& d) {
  const size_t TB = (size_t)d.T * d.B, nwg = (d.B + 15) / 16;
  WideBwdWs L; size_t o = 0;
  L.slabs = o; o += align256(nwg * SLAB * 4);
  L.dpre = o; o += align256((TB + 16) * 128 * 4);    // + 16 sink rows for the lanes beyond a ragged batch
  L.tn = o; o += tn_gemm_big_ws(TB, 128, d.F);
  L.total = o;
  return L;
}

template <int GATE>
void launch_bwd_gate(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x, const void* hs,
                     const void* a0, const void* a1, const void* h0, const fastgrnn_grads& g, void* ws, hipStream_t s) {
  const int nwg = (d.B + 15) / 16;
  dim3 grid(nwg);
  const bool ragged = (d.B % 16) != 0, preact = (d.flags & FASTGRNN_FLAG_SAVE_PREACT) != 0;
  const bool wide = dense_wide_shape(d);
  const WideBwdWs L = wide ? wide_bwd_layout(d) : WideBwdWs{0, 0, 0, 0};
  float* part = reinterpret_cast<float*>(ws);
  float* dpre = wide ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + L.dpre) : nullptr;
  auto go8 = [&](auto kern) __attribute__((always_inline)) {     // 8-wave kernels also take the x layout
    hipLaunchKernelGGL(kern, grid, dim3(512), 0, s, d.T, d.B, row_stride_t(d), row_stride_b(d),
                       ((d.flags & FASTGRNN_FLAG_X_BFT) ? 1 : 0) | ((d.flags & FASTGRNN_FLAG_GRAD_LAST) ? 2 : 0), (const float*)ghs,
                       (const float*)x, (const float*)hs,
                       (const float*)a0, (const float*)a1, (const float*)h0, (const float*)p.w, (const float*)p.u,
                       (const float*)p.bias_gate, (const float*)p.bias_update, (const float*)p.zeta,
                       (const float*)p.nu, wide ? dpre : (float*)g.d_x, (float*)g.d_h0, part);
  };
  if (d.update_nl == FASTGRNN_NL_QUANT_TANH) {       // fp32, SAVE_PREACT, F = 32 (split_supported)
    if (ragged) go8(bwd_scan_split_w8<GATE, true, true, false, false, true>); else go8(bwd_scan_split_w8<GATE, true, false, false, false, true>);
  } else if (wide && d.dtype == FASTGRNN_BF16_IO) {  // SAVE_PREACT, gates sigmoid / relu / tanh (split_supported)
    if constexpr (GATE <= FASTGRNN_NL_TANH) {
      if (ragged) go8(bwd_scan_split_w8<GATE, true, true, true, true>); else go8(bwd_scan_split_w8<GATE, true, false, true, true>);
    }
  } else if (wide) {                                 // fp32 sequences; both saved-tensor contracts
    if (preact) { if (ragged) go8(bwd_scan_split_w8<GATE, true, true, false, true>); else go8(bwd_scan_split_w8<GATE, true, false, false, true>); }
    else        { if (ragged) go8(bwd_scan_split_w8<GATE, false, true, false, true>); else go8(bwd_scan_split_w8<GATE, false, false, false, true>); }
  } else if (d.dtype == FASTGRNN_BF16_IO) {
    if (ragged) go8(bwd_scan_split_w8<GATE, true, true, true>); else go8(bwd_scan_split_w8<GATE, true, false, true>);
  } else if (preact) {
    if (ragged) go8(bwd_scan_split_w8<GATE, true, true>); else go8(bwd_scan_split_w8<GATE, true, false>);
  } else {
    if (ragged) go8(bwd_scan_split_w8<GATE, false, true>); else go8(bwd_scan_split_w8<GATE, false, false>);
  }
  const int ntot = 128 * 128 + 128 * 32 + 2 * 128 + 2;
  hipLaunchKernelGGL(reduce_slabs_split, dim3((ntot + 63) / 64), dim3(1024), 0, s, nwg, part, (const float*)p.zeta,
                     (const float*)p.nu, (float*)g.d_u, wide ? (float*)nullptr : (float*)g.d_w, (float*)g.d_bias_gate,
                     (float*)g.d_bias_update, (float*)g.d_zeta, (float*)g.d_nu);
  if (wide) {
    const size_t TB = (size_t)d.T * d.B;
    // dW[H,F] = d_pre^T . X   (.cu:539 summed over the steps)
    const bool bfw = d.dtype == FASTGRNN_BF16_IO;    // (x and d_x are bf16 then; d_pre is fp32 always)
    tn_gemm_big_run(TB, 128, d.F, dpre, 128, (const float*)x, x, (size_t)0, d.F,
                    reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + L.tn), (float*)g.d_w, d.F, s, bfw);
    // d_x[T*B,F] = d_pre . W   (.cu:538 for every step at once; W is [H,F] = [K,N])
    if (g.d_x) rows_gemm(TB, d.F, 128, true, dpre, (const float*)p.w, g.d_x, false, bfw, s);   // (NULL: not wanted)
  }
}


// pws != nullptr: PREIN -- the frame product P = X.W^T has been written by rows_gemm to zs (SAVE_PREACT), cs (the
// reference's outputs) or, when the caller wants no auxiliary tensor, to the workspace pws
template <int GATE>
void launch_fwd_gate(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs,
                     void* zs, void* cs, hipStream_t s, void* pws = nullptr) {
  dim3 grid((d.B + 15) / 16), block(256);
  const bool ragged = (d.B % 16) != 0;
  const int aux = zs == nullptr ? 0 : ((d.flags & FASTGRNN_FLAG_SAVE_PREACT) ? 2 : 1);
  const bool prein = pws != nullptr;
  if (prein && aux == 0) zs = pws;
  auto go = [&](auto kern) __attribute__((always_inline)) {
    hipLaunchKernelGGL(kern, grid, block, 0, s, d.T, d.B, row_stride_t(d), row_stride_b(d), (const float*)x,
                       (const float*)h0, (const float*)p.w,
                       (const float*)p.u, (const float*)p.bias_gate, (const float*)p.bias_update,
                       (const float*)p.zeta, (const float*)p.nu, (float*)hs, (float*)zs, (float*)cs);
  };
  auto go8 = [&](auto kern) __attribute__((always_inline)) {     // 8-wave kernels also take the x layout
    hipLaunchKernelGGL(kern, grid, dim3(512), 0, s, d.T, d.B, row_stride_t(d), row_stride_b(d),
                       (d.flags & FASTGRNN_FLAG_X_BFT) ? 1 : 0, (const float*)x, (const float*)h0, (const float*)p.w,
                       (const float*)p.u, (const float*)p.bias_gate, (const float*)p.bias_update,
                       (const float*)p.zeta, (const float*)p.nu, (float*)hs, (float*)zs, (float*)cs);
  };
  // fp16 two-plane state product (F16H) only for gates that keep z in [0,1] -- they bound the growth of h to
  // sigma(nu) per step, and the kernel itself checks h0 (see fwd_scan_split_w8); relu / tanh / quantTanh gates
  // always run the three-bf16-plane product, as does FASTGRNN_FLAG_FWD_BF16X3 (A/B)
  constexpr bool BOUNDED = GATE == FASTGRNN_NL_SIGMOID || GATE == FASTGRNN_NL_QUANT_SIGM || GATE == FASTGRNN_NL_QUANT_SIGM4;
  const bool h16 = BOUNDED && !(d.flags & FASTGRNN_FLAG_FWD_BF16X3);
  const bool bf = d.dtype == FASTGRNN_BF16_IO;
  auto pick8 = [&](auto aux_tag) __attribute__((always_inline)) {
    constexpr int A = decltype(aux_tag)::value;
    if constexpr (A == 0 || A == 2) {
      if (d.update_nl == FASTGRNN_NL_QUANT_TANH) {   // fp32, no PREIN (split_supported)
        if constexpr (BOUNDED) {
          if (h16) { if (ragged) go8(fwd_scan_split_w8<GATE, A, true, false, true, false, true>); else go8(fwd_scan_split_w8<GATE, A, false, false, true, false, true>); return; }
        }
        if (ragged) go8(fwd_scan_split_w8<GATE, A, true, false, false, false, true>); else go8(fwd_scan_split_w8<GATE, A, false, false, false, false, true>);
        return;
      }
    }
    if constexpr (GATE <= FASTGRNN_NL_TANH && (A == 0 || A == 2)) {
      if (prein && bf) {                             // wide layer, bf16 hs (P is fp32 whatever the frames are)
        if constexpr (BOUNDED) {
          if (h16) {
            if (ragged) go8(fwd_scan_split_w8<GATE, A, true, true, true, true>); else go8(fwd_scan_split_w8<GATE, A, false, true, true, true>);
            return;
          }
        }
        if (ragged) go8(fwd_scan_split_w8<GATE, A, true, true, false, true>); else go8(fwd_scan_split_w8<GATE, A, false, true, false, true>);
        return;
      }
    }
    if (prein) {                                     // fp32 sequences
      if constexpr (BOUNDED) {
        if (h16) {
          if (ragged) go8(fwd_scan_split_w8<GATE, A, true, false, true, true>); else go8(fwd_scan_split_w8<GATE, A, false, false, true, true>);
          return;
        }
      }
      if (ragged) go8(fwd_scan_split_w8<GATE, A, true, false, false, true>); else go8(fwd_scan_split_w8<GATE, A, false, false, false, true>);
      return;
    }
    if constexpr (BOUNDED) {
      if (h16) {
        if (bf) { if (ragged) go8(fwd_scan_split_w8<GATE, A, true, true, true>); else go8(fwd_scan_split_w8<GATE, A, false, true, true>); }
        else    { if (ragged) go8(fwd_scan_split_w8<GATE, A, true, false, true>); else go8(fwd_scan_split_w8<GATE, A, false, false, true>); }
        return;
      }
    }
    if (bf) { if (ragged) go8(fwd_scan_split_w8<GATE, A, true, true, false>); else go8(fwd_scan_split_w8<GATE, A, false, true, false>); }
    else    { if (ragged) go8(fwd_scan_split_w8<GATE, A, true, false, false>); else go8(fwd_scan_split_w8<GATE, A, false, false, false>); }
  };
  if (d.flags & FASTGRNN_FLAG_HS_LAST) {             // inference: h_T alone (split_supported() admits aux == 0 only)
    pick8(std::integral_constant<int, 3>{});
    return;
  }
  if (bf) {                                          // bf16 sequences: hs only or hs + pre-activation
    if (aux == 2) pick8(std::integral_constant<int, 2>{}); else pick8(std::integral_constant<int, 0>{});
    return;
  }
  if (GATE > FASTGRNN_NL_TANH || (d.flags & (FASTGRNN_FLAG_X_BFT | FASTGRNN_FLAG_HS_LAST | FASTGRNN_FLAG_FWD_BF16X3)) ||
      !(d.flags & FASTGRNN_FLAG_FWD_4WAVE)) {        // default: the 8-wave shape
    if (aux == 1)      pick8(std::integral_constant<int, 1>{});
    else if (aux == 2) pick8(std::integral_constant<int, 2>{});
    else               pick8(std::integral_constant<int, 0>{});
    return;
  }
  if constexpr (GATE <= FASTGRNN_NL_TANH) {          // the 4-wave kernel knows the reference's three gates only
    if (aux == 1)      { if (ragged) go(fwd_scan_split<GATE, 1, true>); else go(fwd_scan_split<GATE, 1, false>); }
    else if (aux == 2) { if (ragged) go(fwd_scan_split<GATE, 2, true>); else go(fwd_scan_split<GATE, 2, false>); }
    else               { if (ragged) go(fwd_scan_split<GATE, 0, true>); else go(fwd_scan_split<GATE, 0, false>); }
  }
}


}  // namespace

bool split_supported(const fastgrnn_desc& d, int direction) {
  if ((d.dtype != FASTGRNN_F32 && d.dtype != FASTGRNN_BF16_IO) ||
      (d.update_nl != FASTGRNN_NL_TANH && d.update_nl != FASTGRNN_NL_QUANT_TANH) ||
      d.gate_nl < FASTGRNN_NL_SIGMOID || d.gate_nl > FASTGRNN_NL_QUANT_SIGM4)
    return false;
  if (d.update_nl == FASTGRNN_NL_QUANT_TANH) {
    // quantTanh update (rnn.py:57-58,292-293): dense H = 128 / F = 32, fp32, hs only or the one-saved-tensor contract
    const bool shape = d.w_rank == 0 && d.u_rank == 0 && d.H == 128 && d.F == 32 && d.dtype == FASTGRNN_F32 &&
                       (double)d.T * d.B * 128 * 4.0 < 4294967296.0;
    if (!shape || (d.flags & (FASTGRNN_FLAG_X_BFT | FASTGRNN_FLAG_HS_LAST | FASTGRNN_FLAG_FWD_4WAVE))) return false;
    return direction == 0 || (d.flags & FASTGRNN_FLAG_SAVE_PREACT) != 0;
  }
  // the 8-wave scans address a step's rows with 32-bit byte offsets from a scalar base: whole sequence tensors
  // below 2^32 bytes (B = 4096, T = 99, H = 128 is 2e8; anything larger goes to the other paths)
  const bool fits32 = (double)d.T * d.B * (d.H > d.F ? d.H : d.F) * 4.0 < 4294967296.0;
  const bool dense = d.w_rank == 0 && d.u_rank == 0 && d.H == 128 && d.F == 32 && fits32;
  const bool preact = (d.flags & FASTGRNN_FLAG_SAVE_PREACT) != 0;
  if (h256_shape(d)) return h256_supported(d, direction);      // dense H = 256 / F = 32: kernels_h256.hip
  if (lowrank_shape(d)) return lowrank_supported(d, direction); // H = 256 / F = 32, ranks <= 16: kernels_lowrank.hip
  if (densified_shape(d)) return densified_supported(d, direction);   // every other factorised cell on a dense shape: kernels_densify.hip
  // dense H = 128 with a wider input (F = 64 / 128 / 256; the reference's second layer): recurrence-only scans +
  // batched GEMMs.  fp32 sequences, time- or batch-major, every gate, full or last-state outputs / gradients.
  if (dense_wide_shape(d)) {
    if (d.flags & FASTGRNN_FLAG_X_BFT) return false;
    if (d.dtype == FASTGRNN_BF16_IO)                 // bf16 sequences (round 3): the reference's three gates, no last-state
      return d.gate_nl <= FASTGRNN_NL_TANH &&        // flags, the backward under the one-saved-tensor contract
             !(d.flags & (FASTGRNN_FLAG_HS_LAST | FASTGRNN_FLAG_GRAD_LAST)) && (direction == 0 || preact);
    if (direction == 0 && preact && (d.flags & FASTGRNN_FLAG_HS_LAST)) return false;
    return d.gate_nl <= FASTGRNN_NL_TANH || direction == 0 || preact;
  }
  // quantised gates (rnn.py:53-60) and the [B,F,T] input layout: 8-wave dense kernels only, i.e. the backward
  // under the SAVE_PREACT contract
  if (d.gate_nl > FASTGRNN_NL_TANH || (d.flags & FASTGRNN_FLAG_X_BFT)) return dense && (direction == 0 || preact);
  // last-state-only outputs / gradients (the classifier's view of the last layer, model.py:227): 8-wave dense kernels
  if (d.flags & (FASTGRNN_FLAG_GRAD_LAST | FASTGRNN_FLAG_HS_LAST)) {
    if (!dense) return false;
    if (direction == 0 && (d.flags & FASTGRNN_FLAG_SAVE_PREACT) && (d.flags & FASTGRNN_FLAG_HS_LAST)) return false;
  }
  // bf16 sequences: dense shape only; the backward only under the SAVE_PREACT contract (8-wave kernel)
  if (d.dtype == FASTGRNN_BF16_IO) return dense && (direction == 0 || preact);
  return dense;
}

size_t split_forward_ws(const fastgrnn_desc& d) {
  // wide layers: the frame product P = X.W^T goes to the auxiliary output the caller passes (z_s under SAVE_PREACT,
  // c_s otherwise); a forward without auxiliary outputs needs room for it.  The query cannot see the pointers, so
  // it answers for that case; forward-only callers (HS_LAST or no gates) are the ones that pay.
  if (dense_wide_shape(d)) return align256((size_t)d.T * d.B * 128 * 4);
  if (h256_shape(d)) return h256_forward_ws(d);
  if (lowrank_shape(d)) return lowrank_forward_ws(d);
  if (densified_shape(d)) return densified_forward_ws(d);
  return 0;
}

bool split_forward_ws_optional(const fastgrnn_desc& d) { return dense_wide_shape(d); }
// shapes whose d_x is a GEMM of its own behind the scan: the caller may pass d_x = NULL to skip it
bool split_dx_optional(const fastgrnn_desc& d) { return dense_wide_shape(d) || h256_shape(d); }

size_t split_backward_ws(const fastgrnn_desc& d) {
  if (h256_shape(d)) return h256_backward_ws(d);
  if (dense_wide_shape(d)) return wide_bwd_layout(d).total;
  if (lowrank_shape(d)) return lowrank_backward_ws(d);
  if (densified_shape(d)) return densified_backward_ws(d);
  return align256((size_t)((d.B + 15) / 16) * SLAB * 4);
}

int split_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x, const void* hs,
                   const void* zs, const void* cs, const void* h0, const fastgrnn_grads& g, void* ws,
                   hipStream_t s) {
  if (h256_shape(d)) return h256_backward(d, p, ghs, x, hs, zs, cs, h0, g, ws, s);
  if (lowrank_shape(d)) return lowrank_backward(d, p, ghs, x, hs, zs, cs, h0, g, ws, s);
  if (densified_shape(d)) return densified_backward(d, p, ghs, x, hs, zs, cs, h0, g, ws, s);
  switch (d.gate_nl) {
    case FASTGRNN_NL_SIGMOID: launch_bwd_gate<FASTGRNN_NL_SIGMOID>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    case FASTGRNN_NL_RELU: launch_bwd_gate<FASTGRNN_NL_RELU>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    case FASTGRNN_NL_TANH: launch_bwd_gate<FASTGRNN_NL_TANH>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    case FASTGRNN_NL_QUANT_TANH: launch_bwd_gate<FASTGRNN_NL_QUANT_TANH>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    case FASTGRNN_NL_QUANT_SIGM: launch_bwd_gate<FASTGRNN_NL_QUANT_SIGM>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    default: launch_bwd_gate<FASTGRNN_NL_QUANT_SIGM4>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
  }
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

int split_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs,
                  void* zs, void* cs, void* ws, hipStream_t s) {
  if (!(d.flags & FASTGRNN_FLAG_SAVE_PREACT) && (zs == nullptr) != (cs == nullptr)) return FASTGRNN_ERR_NULL_POINTER;
  if (d.dtype == FASTGRNN_BF16_IO && zs && !(d.flags & FASTGRNN_FLAG_SAVE_PREACT)) return FASTGRNN_ERR_UNSUPPORTED;
  if (h256_shape(d)) return h256_forward(d, p, x, h0, hs, zs, cs, ws, s);
  if (lowrank_shape(d)) return lowrank_forward(d, p, x, h0, hs, zs, cs, ws, s);
  if (densified_shape(d)) return densified_forward(d, p, x, h0, hs, zs, cs, ws, s);
  void* pws = nullptr;
  if (dense_wide_shape(d)) {
    // P[T*B,H] = X . W^T, the one genuinely dense contraction of the layer (.cu:356 per step), into the buffer the
    // scan reads it from: z_s (SAVE_PREACT: overwritten by the pre-activation), c_s (overwritten by h_prime), or
    // the workspace when the caller wants neither
    pws = zs == nullptr ? ws : ((d.flags & FASTGRNN_FLAG_SAVE_PREACT) ? zs : cs);
    if (!pws) return FASTGRNN_ERR_WORKSPACE;
    const int st = rows_gemm((size_t)d.T * d.B, 128, d.F, false, x, (const float*)p.w, pws, d.dtype == FASTGRNN_BF16_IO, false, s);
    if (st != FASTGRNN_OK) return st;
  }
  switch (d.gate_nl) {
    case FASTGRNN_NL_SIGMOID: launch_fwd_gate<FASTGRNN_NL_SIGMOID>(d, p, x, h0, hs, zs, cs, s, pws); break;
    case FASTGRNN_NL_RELU: launch_fwd_gate<FASTGRNN_NL_RELU>(d, p, x, h0, hs, zs, cs, s, pws); break;
    case FASTGRNN_NL_TANH: launch_fwd_gate<FASTGRNN_NL_TANH>(d, p, x, h0, hs, zs, cs, s, pws); break;
    case FASTGRNN_NL_QUANT_TANH: launch_fwd_gate<FASTGRNN_NL_QUANT_TANH>(d, p, x, h0, hs, zs, cs, s, pws); break;
    case FASTGRNN_NL_QUANT_SIGM: launch_fwd_gate<FASTGRNN_NL_QUANT_SIGM>(d, p, x, h0, hs, zs, cs, s, pws); break;
    default: launch_fwd_gate<FASTGRNN_NL_QUANT_SIGM4>(d, p, x, h0, hs, zs, cs, s, pws); break;
  }
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

}  // namespace fastgrnn
