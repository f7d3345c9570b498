// Low-rank FastGRNN scans for gfx950 (H = 256, F = 32, wRank, uRank <= 16: BASELINE config 4 is 16 / 16),
// split-precision like kernels_split.hip: fp32 results from three exact bf16 planes on the bf16 matrix pipe.
// Reference semantics: rnn.py:280-295 (factorised cell), .cu:546-555 (factor gradients).
#include "split_common.h"

namespace fastgrnn {
namespace {

// ------------------------------------------------------------------------------------------
// forward, low-rank  (H = 256, F = 32, wRank, uRank <= 16; BASELINE config 4 is 16 / 16)
// ------------------------------------------------------------------------------------------
// pre = W2 (W1 x) + U2 (U1 h), evaluated factorised like the CPU cell (rnn.py:280-287).  Workgroup =
// NW waves = 16 utterances; wave w owns hidden units UPW*w .. (NW = 8: 32 units, two 16-row tiles; NW = 4: 64
// units, four tiles) and a lane its UPW/4 consecutive units.  All factor planes are resident in registers.
//   A  m_h partial: U1 contracted over the wave's OWN 64 units -- the B operand is the lane's own
//      two fragments of h, straight from registers (h never goes through LDS); m_x = W1 x.
//   -  the four partials (and m_x, from wave 0) meet in a 16 KB LDS buffer: ONE barrier per step
//   B  pre tile = [U2 | W2] . [m_h ; m_x]: K = 16 + 16 = one K-step of 32, four row tiles per wave,
//      run one after the other so that each tile's epilogue sits under the next tile's MFMAs.
// Ranks below 16 are zero-extended to 16 when the factors are loaded (rows of W1 / U1, columns of W2 / U2 beyond
// the rank are exact zeros, so are the matching entries of the rank-space vector): same arithmetic, same code.
// Row (t, b) of hs / zs is t*rsT + b*rsB (time- or batch-major), of x t*xsT + b*xsB; the saved rank-space vector cs
// is always [T*B, 32] time-major (it only travels from this kernel to the backward).  BF: x and hs are bf16 (the
// state itself stays fp32 in registers, as in the dense scans).  LAST (AUX == 0): hs is [B,H] and receives h_T only.
template <int GATE, int AUX, bool RAGGED, bool BF = false, bool LAST = false, int NW = 8>
__global__ __launch_bounds__(NW * 64) void fwd_scan_lowrank_split(
    int Tn, int B, int rsT, int rsB, int xsT, int xsB, int rw, int ru, const float* __restrict__ x,
    const float* __restrict__ h0,
    const float* __restrict__ w1, const float* __restrict__ w2,
    const float* __restrict__ u1, const float* __restrict__ u2,
    const float* __restrict__ bz, const float* __restrict__ bh,
    const float* __restrict__ zeta, const float* __restrict__ nu,
    float* __restrict__ hs, float* __restrict__ zs, float* __restrict__ cs) {
  // NW waves per workgroup (8 = two per SIMD: one wave's VALU epilogue runs beside the other's MFMAs / LDS round
  // trip; 4 = the first shape, kept for A/B).  UPW units per wave, NT row tiles, KU K-steps of U1 over own units.
  constexpr int H = 256, F = 32, UPW = H / NW, NT = UPW / 16, KU = UPW / 32, UPL = UPW / 4;
  constexpr int MROW = 36;   // padded floats per (wave, utterance) row of m
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  __shared__ __attribute__((aligned(16))) float mp[2][NW][16][MROW];
  __shared__ __attribute__((aligned(16))) float msum[2][16][MROW];   // NW == 8: the summed rank-space vector

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, i = l & 15, g = l >> 4;
  const int b = blockIdx.x * 16 + i;
  const bool valid = !RAGGED || b < B;
  const int bc = valid ? b : B - 1;
  const int n0 = wv * UPW + g * UPL;               // this lane's UPL consecutive hidden units

  // ---- resident A operands -----------------------------------------------------------------
  Frag3 U1f[KU], W1f, UW2f[NT];
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int v = 0; v < KU; ++v) {                   // rows = rank index i; K-step v = units n0 + 8v + j of lane group g
    const float* p = u1 + (size_t)(i < ru ? i : 0) * H + n0 + 8 * v;
    U1f[v] = i < ru ? split3(ld4(p), ld4(p + 4)) : split3(zero4, zero4);
  }
  {
    const float* p = w1 + (size_t)(i < rw ? i : 0) * F + 8 * g;
    W1f = i < rw ? split3(ld4(p), ld4(p + 4)) : split3(zero4, zero4);
  }
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {                // rows = units; K = [m_h rows 8g.. | m_x rows 8(g-2)..]
    const int nA = wv * UPW + (i >> 2) * UPL + mt * 4 + (i & 3);
    const int rk = g < 2 ? ru : rw, j0 = g < 2 ? 8 * g : 8 * (g - 2);
    const float* p = (g < 2 ? u2 : w2) + (size_t)nA * rk;
    f32x4 lo, hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo[j] = j0 + j < rk ? p[j0 + j] : 0.f;
      hi[j] = j0 + 4 + j < rk ? p[j0 + 4 + j] : 0.f;
    }
    UW2f[mt] = split3(lo, hi);
  }
  f32x4 bzv[NT], bhv[NT], hown[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
    bzv[mt] = ld4(bz + n0 + 4 * mt);
    bhv[mt] = ld4(bh + n0 + 4 * mt);
    hown[mt] = ld4(h0 + (size_t)bc * H + n0 + 4 * mt);
  }
  Frag3 hfrag[KU];
#pragma unroll
  for (int v = 0; v < KU; ++v) hfrag[v] = split3(hown[2 * v], hown[2 * v + 1]);
  const float sz = fsigmoid(zeta[0]), sn = fsigmoid(nu[0]);

  struct Feat { f32x4 lo, hi; };
  struct Gates { f32x4 z[NT], c[NT], mlo, mhi; };
  auto load_x = [&](int t, Feat& q) __attribute__((always_inline)) {
    const size_t e = ((size_t)t * xsT + (size_t)bc * xsB) * F + 8 * g;
    if (BF) {                                       // 8 bf16 = 16 bytes
      const u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(x) + e);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        q.lo[2 * j] = __uint_as_float(r[j] << 16); q.lo[2 * j + 1] = __uint_as_float(r[j] & 0xffff0000u);
        q.hi[2 * j] = __uint_as_float(r[2 + j] << 16); q.hi[2 * j + 1] = __uint_as_float(r[2 + j] & 0xffff0000u);
      }
    } else {
      q.lo = ld4(x + e); q.hi = ld4(x + e + 4);
    }
  };
  auto store_step = [&](int t, const Gates& gt) __attribute__((always_inline)) {   // hown still holds h_t
    if (LAST) return;
    if (valid) {
      const size_t o = ((size_t)t * rsT + (size_t)b * rsB) * H + n0;
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) {
        if (BF) st4_bf16(reinterpret_cast<unsigned short*>(hs) + o + 4 * mt, hown[mt]); else st4(hs + o + 4 * mt, hown[mt]);
      }
      if (AUX == 1) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) { st4(zs + o + 4 * mt, gt.z[mt]); st4(cs + o + 4 * mt, gt.c[mt]); }
      } else if (AUX == 2 || AUX == 3) {             // (3: the rank-space vector alone -- the backward recomputes the
        if (AUX == 2) {                              //  pre-activation from it, bwd_scan_lowrank_split<.., RECOMP>)
#pragma unroll
          for (int mt = 0; mt < NT; ++mt) st4(zs + o + 4 * mt, gt.z[mt]);    // gt.z carries the pre-activation
        }
        // [m_h | m_x] of the step: cs is [T,B,32] in this mode; every wave holds the same sum, wave w
        // stores lane groups g == w (8 floats each)
        if (g == wv) {
          float* mo = cs + ((size_t)t * B + b) * 32 + 8 * g;
          st4(mo, gt.mlo); st4(mo + 4, gt.mhi);
        }
      }
    }
  };

  auto step = [&](auto first_tag, int t, int cur, Feat& xuse, Feat& xload, Gates& gprev,
                  Gates& gout) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    // ---- A: rank-space partials ------------------------------------------------------------
    const Frag3 xB = split3(xuse.lo, xuse.hi);
    // xB is needed inside a wave-uniform branch only; left alone the compiler sinks the split AND the load of
    // x_0 into that branch, i.e. behind the MFMAs below and into their dead operand registers (the first step
    // of the 4-wave variant came out wrong that way).  The empty asm pins the planes here (operand rule, 4.0).
    asm volatile("" :: "v"(xB.p[0]), "v"(xB.p[1]), "v"(xB.p[2]));
    __builtin_amdgcn_sched_barrier(0);
    f32x4 mh = mfma6(U1f[0], hfrag[0], f32x4{0.f, 0.f, 0.f, 0.f});                      // rnn.py:286 (partial over own units)
    if constexpr (KU == 2) mh = mfma6(U1f[1], hfrag[1], mh);
    f32x4 mx = f32x4{0.f, 0.f, 0.f, 0.f};                                               // m_x enters the sum once:
    if (wv == 0) mx = mfma6(W1f, xB, mx);                                               // wave 0 (uniform branch); rnn.py:280
    if (!FIRST) store_step(t - 1, gprev);
    // lane (b=i, g) holds rows 4g..4g+3 of both 16-row results
    *reinterpret_cast<f32x4*>(&mp[cur][wv][i][4 * g]) = mh;
    *reinterpret_cast<f32x4*>(&mp[cur][wv][i][16 + 4 * g]) = mx;
    lds_barrier();
    // The request for x_{t+1} goes out here: every MFMA issued so far has retired (its result went through LDS
    // above), so the load cannot land in an operand register that the matrix pipe still has to fetch.
    load_x(t + 1 < Tn ? t + 1 : t, xload);
    __builtin_amdgcn_sched_barrier(0);
    // ---- m = sum of the partials; this lane's B fragment is rows 8g..8g+7 of [m_h ; m_x] -------
    f32x4 mlo = f32x4{0.f, 0.f, 0.f, 0.f}, mhi = mlo;
    if constexpr (NW == 8) {
      // Two stages: every wave needs the whole sum, and eight waves each reading all eight partials is 131 KB of
      // LDS reads per step.  512 threads = 16 utterances x 32 values: each adds ONE value's eight partials (same
      // order as below: identical bits), the sums go through a 2 KB buffer and one more barrier.
      const int u = tid & 15, j = tid >> 4;
      float sj = 0.f;
#pragma unroll
      for (int w2i = 0; w2i < NW; ++w2i) sj += mp[cur][w2i][u][j];
      msum[cur][u][j] = sj;
      lds_barrier();
      mlo = *reinterpret_cast<const f32x4*>(&msum[cur][i][8 * g]);
      mhi = *reinterpret_cast<const f32x4*>(&msum[cur][i][8 * g + 4]);
    } else {
#pragma unroll
      for (int w2i = 0; w2i < NW; ++w2i) {
        mlo += *reinterpret_cast<const f32x4*>(&mp[cur][w2i][i][8 * g]);
        mhi += *reinterpret_cast<const f32x4*>(&mp[cur][w2i][i][8 * g + 4]);
      }
    }
    if (AUX == 2 || AUX == 3) { gout.mlo = mlo; gout.mhi = mhi; }     // stored with the step's other outputs
    const Frag3 mB = split3(mlo, mhi);
    // ---- B: pre-activation tiles, epilogue of tile k under the MFMAs of tile k+1 -----------------
    f32x4 acc[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      acc[mt] = mfma6(UW2f[mt], mB, f32x4{0.f, 0.f, 0.f, 0.f});                          // rnn.py:281,287,289
      if (mt > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                                    // .cu:55-58, tile mt-1
          const float pre = acc[mt - 1][r];
          const float z = gate_act<GATE>(pre + bzv[mt - 1][r]);
          const float c = ftanh(pre + bhv[mt - 1][r]);
          hown[mt - 1][r] = (sz * (1.0f - z) + sn) * c + hown[mt - 1][r] * z;
          gout.z[mt - 1][r] = (AUX == 2) ? pre : z; gout.c[mt - 1][r] = c;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pre = acc[NT - 1][r];
      const float z = gate_act<GATE>(pre + bzv[NT - 1][r]);
      const float c = ftanh(pre + bhv[NT - 1][r]);
      hown[NT - 1][r] = (sz * (1.0f - z) + sn) * c + hown[NT - 1][r] * z;
      gout.z[NT - 1][r] = (AUX == 2) ? pre : z; gout.c[NT - 1][r] = c;
    }
#pragma unroll
    for (int v = 0; v < KU; ++v) hfrag[v] = split3(hown[2 * v], hown[2 * v + 1]);
  };

  Feat xa, xb;
  Gates ga, gb;
  load_x(0, xa);
  __builtin_amdgcn_sched_barrier(0);         // every prologue request is out before the first MFMA
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
  if (LAST && valid) {
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      const size_t o = (size_t)b * H + n0 + 4 * mt;
      if (BF) st4_bf16(reinterpret_cast<unsigned short*>(hs) + o, hown[mt]); else st4(hs + o, hown[mt]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, low-rank  (H = 256, F = 32, wRank, uRank <= 16), FASTGRNN_FLAG_SAVE_PREACT contract only
// ------------------------------------------------------------------------------------------
// Mirror of fwd_scan_lowrank_split.  Per step: EW on the VALU (z, c recomputed from the saved
// pre-activation) -> d_pre; rank-space partial [U2|W2]^T d_pre over the wave's OWN units with the B
// operand straight from registers; the four partials meet in LDS (one barrier); d_h = z*g + U1^T d_m_h
// for the wave's own units and d_x = W1^T d_m_x.  d_pre[T,B,H] and d_m[T,B,32] go to the workspace:
// the weight gradients (K = T*B) are contracted afterwards by split-K GEMMs (.cu:546-555 evaluated
// factorised), because neither their accumulators nor the images they would need fit on chip beside
// the factors.
// Ranks below 16: the factors are zero-extended when they are loaded, as in the forward.  Row (t, b) of grad_hs / hs /
// pre_s is t*rsT + b*rsB, of d_x t*xsT + b*xsB; the two workspace tensors are time-major.  glast: grad_hs is [B,H], the
// gradient of the last state only (FASTGRNN_FLAG_GRAD_LAST).  BF: grad_hs, hs and d_x are bf16 (h0, pre_s fp32).
constexpr int SLAB_LR = 576;   // floats per workgroup: d_bz[256] | d_bh[256] | (zeta, nu) sums, padded to 64

// RECOMP: pre_s is NULL and the pre-activation of step t is recomputed from the rank-space vector the forward saved,
// pre = [U2|W2] . [m_h ; m_x] -- the forward's own second product (one K-step, six MFMA terms per row tile, same
// operands in the same order: the same bits) with the [U2|W2] fragments parked in LDS.  Config 4 moves 3.8 GB per step
// at 4.6 TB/s; the saved pre-activation is 0.83 GB of that (written by the forward, read here).
template <int GATE, bool RAGGED, bool BF = false, bool RECOMP = false, int NW = 8>
__global__ __launch_bounds__(NW * 64) void bwd_scan_lowrank_split(
    int Tn, int B, int rsT, int rsB, int xsT, int xsB, int rw, int ru, int glast,
    const float* __restrict__ ghs, const float* __restrict__ hs, const float* __restrict__ pre_s,
    const float* __restrict__ m_s,
    const float* __restrict__ h0, const float* __restrict__ w1, const float* __restrict__ w2,
    const float* __restrict__ u1, const float* __restrict__ u2,
    const float* __restrict__ bz, const float* __restrict__ bh,
    const float* __restrict__ zeta, const float* __restrict__ nu,
    float* __restrict__ d_x, float* __restrict__ d_h0,
    float* __restrict__ dpre_ws, float* __restrict__ dm_ws, float* __restrict__ part) {
  // NW waves (8 = two per SIMD, 32 units each; 4 = the first shape): see fwd_scan_lowrank_split
  constexpr int H = 256, F = 32, UPW = H / NW, NT = UPW / 16, KU = UPW / 32, UPL = UPW / 4, MROW = 36;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  __shared__ __attribute__((aligned(16))) float mp[2][NW][16][MROW];
  __shared__ __attribute__((aligned(16))) float sbias[2][H];
  __shared__ float red[2 * NW];
  __shared__ __attribute__((aligned(16))) u32x4 uw2l[RECOMP ? NW * NT * 3 * 64 : 1];   // [U2|W2] fragments, per wave

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, i = l & 15, g = l >> 4;
  const int b = blockIdx.x * 16 + i;
  const bool valid = !RAGGED || b < B;
  const int bc = valid ? b : B - 1;
  const int n0 = wv * UPW + g * UPL;
  const float sz = fsigmoid(zeta[0]), sn = fsigmoid(nu[0]);
  if (tid < H) { sbias[0][tid] = bz[tid]; sbias[1][tid] = bh[tid]; }

  // ---- resident A operands -----------------------------------------------------------------
  // d_m[j][b] = sum_n [U2|W2][n][j] d_pre[b][n] over own units: tile 0 rows = U2 columns, tile 1 = W2 columns
  Frag3 UW2Tf[2][KU];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl) {
    const float* src = tl == 0 ? u2 : w2;
    const int rk = tl == 0 ? ru : rw;
    const int ic = i < rk ? i : 0;
#pragma unroll
    for (int v = 0; v < KU; ++v) {
      f32x4 lo, hi;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo[j] = i < rk ? src[(size_t)(n0 + 8 * v + j) * rk + ic] : 0.f;
        hi[j] = i < rk ? src[(size_t)(n0 + 8 * v + 4 + j) * rk + ic] : 0.f;
      }
      UW2Tf[tl][v] = split3(lo, hi);
    }
  }
  // d_h[k][b] = z*g + sum_j U1[j][k] d_m_h[j][b]: rows = own units, K = [d_m_h rows 8g.. | nothing]
  Frag3 U1Tf[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
    const int kA = wv * UPW + (i >> 2) * UPL + mt * 4 + (i & 3);
    f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (g < 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo[j] = 8 * g + j < ru ? u1[(size_t)(8 * g + j) * H + kA] : 0.f;
        hi[j] = 8 * g + 4 + j < ru ? u1[(size_t)(8 * g + 4 + j) * H + kA] : 0.f;
      }
    }
    U1Tf[mt] = split3(lo, hi);
  }
  // d_x[f][b] = sum_j W1[j][f] d_m_x[j][b]: feature tile wv&1 (stored by waves 0,1), K = [nothing | d_m_x rows 8(g-2)..]
  Frag3 W1Tf;
  {
    const int f = (wv & 1) * 16 + i;
    f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (g >= 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo[j] = 8 * (g - 2) + j < rw ? w1[(size_t)(8 * (g - 2) + j) * F + f] : 0.f;
        hi[j] = 8 * (g - 2) + 4 + j < rw ? w1[(size_t)(8 * (g - 2) + 4 + j) * F + f] : 0.f;
      }
    }
    W1Tf = split3(lo, hi);
  }

  if (RECOMP) {                                    // rows = own units; K = [m_h rows 8g.. | m_x rows 8(g-2)..]: as the forward
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      const int nA = wv * UPW + (i >> 2) * UPL + mt * 4 + (i & 3);
      const int rk = g < 2 ? ru : rw, j0 = g < 2 ? 8 * g : 8 * (g - 2);
      const float* pf = (g < 2 ? u2 : w2) + (size_t)nA * rk;
      f32x4 lo, hi;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo[j] = j0 + j < rk ? pf[j0 + j] : 0.f;
        hi[j] = j0 + 4 + j < rk ? pf[j0 + 4 + j] : 0.f;
      }
      const Frag3 f = split3(lo, hi);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) uw2l[((wv * NT + mt) * 3 + pl) * 64 + l] = f.p[pl];
    }
  }

  f32x4 sbz[NT], sbh[NT], dh[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) { sbz[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; sbh[mt] = sbz[mt]; dh[mt] = sbz[mt]; }
  float pz = 0.f, pn = 0.f, pz_c = 0.f, pn_c = 0.f;     // d_zeta / d_nu partial sums, compensated

  // grad_hs, pre-activation (RECOMP: a0[0], a0[1] carry the 8 rank-space values of this lane's K rows), h_prev
  struct EwOps { f32x4 g[NT], a0[NT], h[NT]; };
  static_assert(!RECOMP || NT == 2, "the recomputation parks the rank-space fragment in a0[0..1]");
  // Addresses are a wave-uniform step base (scalar registers) + a 32-bit lane offset: one VGPR per stream instead
  // of a 64-bit pointer pair each (the host rejects B*H*4 >= 2^31 for this path).
  const unsigned lane_h = (unsigned)bc * rsB * H + n0, lane_0 = (unsigned)bc * H + n0;
  // stores: rows of the two workspace tensors advance by B per step for the lanes of the batch and not at all for
  // the others, whose rows are the 16 sink rows behind row T*B; d_x likewise, its sink behind d_m's
  const unsigned sk_step = valid ? (unsigned)B : 0u;
  const unsigned sk_row = valid ? (unsigned)b : (unsigned)Tn * (unsigned)B + i;
  const unsigned dp_off = (sk_row * H + n0) * 4u, dm_off = (sk_row * 32 + 8 * g) * 4u;
  constexpr unsigned XSZ = BF ? 2u : 4u;
  char* const dx_base = valid ? reinterpret_cast<char*>(d_x) + ((size_t)b * xsB * F + (wv & 1) * 16 + 4 * g) * XSZ
                              : reinterpret_cast<char*>(dm_ws) + (((size_t)Tn * B + 16) * 32 + i * 32 + (wv & 1) * 16 + 4 * g) * 4u;
  const unsigned dx_step = valid ? (unsigned)xsT * F * XSZ : 0u;
  auto ld4s = [&](const float* base, unsigned e) __attribute__((always_inline)) -> f32x4 {   // 4 sequence elements
    if (BF) {
      const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + e);
      return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                   __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
    }
    return ld4(base + e);
  };
  auto load_ew = [&](int t, EwOps& e) __attribute__((always_inline)) {
    const size_t step = (size_t)t * rsT * H;                                 // uniform; in elements
    const float* pt = pre_s + step;
    // sequence tensors are addressed in ELEMENTS from their typed base (bf16: half the bytes)
    const size_t gstep = glast ? 0 : step, hstep = step - (size_t)rsT * H;
    const unsigned lane_g = glast ? lane_0 : lane_h;
    const bool gzero = (RAGGED && !valid) || (glast && t != Tn - 1);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      // lanes beyond a ragged batch: the last utterance's rows with a ZERO gradient (dh starts at zero, so gg,
      // d_pre and every sum they enter stay exactly zero for them; see bwd_scan_split_w8)
      e.g[mt] = gzero ? f32x4{0.f, 0.f, 0.f, 0.f}
                      : ld4s(BF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(ghs) + gstep) : ghs + gstep,
                             lane_g + 4 * mt);
      if (RECOMP) e.a0[mt] = ld4(m_s + ((size_t)t * B + bc) * 32 + 8 * g + 4 * mt);   // rows 8g.. of [m_h ; m_x], time-major
      else e.a0[mt] = ld4(pt + lane_h + 4 * mt);
      if (t == 0) e.h[mt] = ld4(h0 + lane_0 + 4 * mt);                       // .cu:478-481
      else e.h[mt] = ld4s(BF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(hs) + hstep) : hs + hstep,
                          lane_h + 4 * mt);
    }
  };

  // RECOMP: e.a0 arrives holding this lane's 8 rank-space values and leaves holding the pre-activation of the own units.
  // Two parts, because the fragments must be in registers of their own BEFORE any MFMA that is still in flight when
  // they are requested (operand rule): recomp_prep reads them (and splits the rank-space values), recomp_issue runs
  // the products and ends with their completion read.
  struct Recomp { Frag3 Uf[NT], mBf; };
  auto recomp_prep = [&](const EwOps& e, Recomp& rc) __attribute__((always_inline)) {
    rc.mBf = split3(e.a0[0], e.a0[1]);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) rc.Uf[mt].p[pl] = uw2l[((wv * NT + mt) * 3 + pl) * 64 + l];
    __builtin_amdgcn_sched_barrier(0);
  };
  auto recomp_issue = [&](EwOps& e, const Recomp& rc) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) e.a0[mt] = mfma6(rc.Uf[mt], rc.mBf, f32x4{0.f, 0.f, 0.f, 0.f});  // rnn.py:281,287,289
    __builtin_amdgcn_sched_barrier(0);
    // completion read HERE: the next step's first instructions are LDS reads (the biases) that the compiler would
    // otherwise land in the fragments' dead registers right behind the MFMAs (operand rule; the scanner found it)
    float touch = 0.f;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) touch += e.a0[mt][0];
    if (touch == 1.2345678e38f) red[1] = 1.f;
    __builtin_amdgcn_sched_barrier(0);
  };
  auto step = [&](int t, EwOps& e) __attribute__((always_inline)) {
    const int cur = t & 1;
    {
      // dh read = the previous step's MFMAs have retired: the requests below may land in registers they read
      // (operand rule)
      float touch = 0.f;
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) touch += dh[mt][0];
      if (touch == 1.2345678e38f) red[0] = 1.f;
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- EW(t): .cu:107-117 ------------------------------------------------------------------
    f32x4 dpv[NT];
    float sz8 = 0.f, sn8 = 0.f;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      const f32x4 bzq = *reinterpret_cast<const f32x4*>(&sbias[0][n0 + 4 * mt]);
      const f32x4 bhq = *reinterpret_cast<const f32x4*>(&sbias[1][n0 + 4 * mt]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = gate_act<GATE>(e.a0[mt][r] + bzq[r]);
        const float c = ftanh(e.a0[mt][r] + bhq[r]);
        const float gg = e.g[mt][r] + dh[mt][r];                                 // .cu:474
        const float dcp = (sz * (1.0f - z) + sn) * (1.0f - c * c) * gg;          // .cu:109
        const float dzp = (e.h[mt][r] - sz * c) * gate_dact<GATE>(z) * gg;       // .cu:110
        const float zg = z * gg;                                                  // .cu:108
        const float tz = (1.0f - z) * c * gg, tn = c * gg;                        // .cu:114-115
        sbz[mt][r] += dzp; sbh[mt][r] += dcp; sz8 += tz; sn8 += tn;
        dpv[mt][r] = dzp + dcp;                                                   // .cu:113
        dh[mt][r] = zg;
      }
    }
    kahan_add(pz, pz_c, sz8); kahan_add(pn, pn_c, sn8);
    // EW(t) has consumed the operand set (and read dh: the previous step's MFMAs have retired): refill it for t-1
    __builtin_amdgcn_sched_barrier(0);
    if (t > 0) load_ew(t - 1, e);
    __builtin_amdgcn_sched_barrier(0);
    {
      // every lane stores, without a branch: lanes beyond a ragged batch write to sink rows behind the T*B rows
      // (step stride 0) -- no exec-masked block around memory instructions in the steady state
      char* o = reinterpret_cast<char*>(dpre_ws) + (size_t)t * sk_step * (H * 4u) + dp_off;
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) *reinterpret_cast<f32x4*>(o + 16 * mt) = dpv[mt];
    }
    // ---- rank-space partial over own units: B operand = this lane's two fragments of d_pre ----
    Frag3 dfr[KU];
#pragma unroll
    for (int v = 0; v < KU; ++v) dfr[v] = split3(dpv[2 * v], dpv[2 * v + 1]);
    // big and small terms in accumulators of their own, as in the dense scans (mfma6_hl: inside one MFMA the
    // addends are chopped at the largest one, a one-signed loss that showed in d_zeta / d_nu at B = 4096)
    f32x4 mh = f32x4{0.f, 0.f, 0.f, 0.f}, mx = mh, mhl = mh, mxl = mh;
#pragma unroll
    for (int v = 0; v < KU; ++v) mfma6_hl(UW2Tf[0][v], dfr[v], mh, mhl);
#pragma unroll
    for (int v = 0; v < KU; ++v) mfma6_hl(UW2Tf[1][v], dfr[v], mx, mxl);
    *reinterpret_cast<f32x4*>(&mp[cur][wv][i][4 * g]) = mh + mhl;
    *reinterpret_cast<f32x4*>(&mp[cur][wv][i][16 + 4 * g]) = mx + mxl;
    lds_barrier();
    f32x4 mlo = f32x4{0.f, 0.f, 0.f, 0.f}, mhi = mlo;
    // (the forward's two-stage sum was tried here too: 1 % at most, and its extra registers made the kernel spill)
#pragma unroll
    for (int w2i = 0; w2i < NW; ++w2i) {
      mlo += *reinterpret_cast<const f32x4*>(&mp[cur][w2i][i][8 * g]);
      mhi += *reinterpret_cast<const f32x4*>(&mp[cur][w2i][i][8 * g + 4]);
    }
    if (wv == 0) {                                   // (wave-uniform) [d_m_h | d_m_x] of this step for the weight-gradient GEMMs
      char* mo = reinterpret_cast<char*>(dm_ws) + (size_t)t * sk_step * (32 * 4u) + dm_off;
      *reinterpret_cast<f32x4*>(mo) = mlo; *reinterpret_cast<f32x4*>(mo + 16) = mhi;
    }
    const Frag3 mB = split3(mlo, mhi);
    Recomp rc;
    if (RECOMP && t > 0) recomp_prep(e, rc);         // (fragments of the NEXT step's pre-activation: requested before the products below)
    // ---- d_old_h for the own units (C-in = z*g) and d_x ---------------------------------------------
    f32x4 dlo[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) { dlo[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; mfma6_hl(U1Tf[mt], mB, dh[mt], dlo[mt]); }
    if (wv < 2) {                                    // (wave-uniform) feature tile wv
      f32x4 dxv = f32x4{0.f, 0.f, 0.f, 0.f}, dxl = dxv;
      mfma6_hl(W1Tf, mB, dxv, dxl);
      dxv += dxl;
      {                                              // lanes beyond a ragged batch: the sink behind d_m (dx_base)
        char* o = dx_base + (size_t)t * dx_step;
        if (BF) st4_bf16(o, dxv); else *reinterpret_cast<f32x4*>(o) = dxv;
      }
    }
    // RECOMP: the next step's pre-activation now, behind this step's last products in the matrix pipe (at the top of
    // the next step it is a dependent phase of ~350 cycles in front of EW: +34 us per launch, measured)
    if (RECOMP && t > 0) recomp_issue(e, rc);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) dh[mt] += dlo[mt];
  };

  EwOps ea;                                          // ONE operand set, refilled right behind its use (see step)
  __syncthreads();                                   // sbias
  load_ew(Tn - 1, ea);
  if (RECOMP) { Recomp rc0; recomp_prep(ea, rc0); recomp_issue(ea, rc0); }
  for (int t = Tn - 1; t >= 0; --t) step(t, ea);
  {
    // the last step's MFMAs have retired before anything below (stores masked by `valid`, the reductions' LDS
    // traffic) may reuse their operand registers: an unconditional read of every d_h accumulator
    float touch = 0.f;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) touch += dh[mt][0];
    if (touch == 1.2345678e38f) red[0] = 1.f;
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- flush ---------------------------------------------------------------------------------
  if (valid) {
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) st4(d_h0 + (size_t)b * H + n0 + 4 * mt, dh[mt]);
  }
#pragma unroll
  for (int mt = 0; mt < NT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = sbz[mt][r], c = sbh[mt][r];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); c += __shfl_xor(c, m); }
      if (i == 0) {
        float* pb = part + (size_t)blockIdx.x * SLAB_LR;
        pb[n0 + 4 * mt + r] = a;
        pb[H + n0 + 4 * mt + r] = c;
      }
    }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { pz += __shfl_xor(pz, m); pn += __shfl_xor(pn, m); }
  if (l == 0) { red[wv] = pz; red[NW + wv] = pn; }
  __syncthreads();
  if (tid == 0) {
    float* pzn = part + (size_t)blockIdx.x * SLAB_LR + 2 * H;
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int w2i = 0; w2i < NW; ++w2i) { a += red[w2i]; c += red[NW + w2i]; }
    pzn[0] = a; pzn[1] = c;
  }
}

// ------------------------------------------------------------------------------------------
// C[M,N] = A[:, :M]^T . B[:, :N] over R rows (R = T*B, huge; M, N small): split-precision, matrix pipe
// ------------------------------------------------------------------------------------------
// The weight gradients of the low-rank backward (.cu:546-555, factorised).  Workgroup = 4 waves = one
// chunk of TN_CHUNK rows, staged 32 rows at a time: global fp32 -> three exact bf16 planes in LDS in
// natural [row][column] order -> hardware-transposed fragment reads (K = rows) -> 6-term MFMAs into
// register accumulators.  Each workgroup leaves its partial C in the workspace; tn_reduce sums them
// in a fixed order.  MT x NT = 16x16 tiles of C; wave w owns tiles w, w+4, ...
// MAPB: B's rows are rows of a sequence tensor in the caller's layout: row r = (t, b) (time-major numbering, as A's)
// lives at B1 + ((t - shift)*rsT + b*rsB)*ldb, and rows with t < shift come from B0 + b*ldb (h0).  Without MAPB the
// tensor is time-major and that is r - shiftB (shiftB = shift*Bn).  BFB: B1 is bf16 (B0 stays fp32).
constexpr int TN_CHUNK = 800, TN_STAGE = 32;

template <int MT, int NT, bool MAPB = false, bool BFB = false>
__global__ __launch_bounds__(256) void tn_gemm_split(size_t R, const float* __restrict__ A, int lda,
                                                     const float* __restrict__ B0, const float* __restrict__ B1,
                                                     size_t shiftB, int ldb, float* __restrict__ part,
                                                     int Bn = 1, int rsT = 0, int rsB = 0) {
  constexpr int M = MT * 16, N = NT * 16, ROWA = M * 2 + 32, ROWB = N * 2 + 32;
  constexpr int NTILE = MT * NT, TPW = (NTILE + 3) / 4;
  constexpr int VA = (TN_STAGE * M / 4 + 255) / 256, VB = (TN_STAGE * N / 4 + 255) / 256;   // float4 per thread per stage
  __shared__ __attribute__((aligned(16))) unsigned char la[3][TN_STAGE * ROWA];
  __shared__ __attribute__((aligned(16))) unsigned char lb[3][TN_STAGE * ROWB];

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, g = l >> 4, q = (l & 15) >> 2, pp = l & 3;
  const size_t r_begin = (size_t)blockIdx.x * TN_CHUNK;
  const size_t r_end = (r_begin + TN_CHUNK < R) ? r_begin + TN_CHUNK : R;

  f32x4 va[VA], vb[VB];
  auto ld4b = [&](const float* base, size_t e) __attribute__((always_inline)) -> f32x4 {   // 4 elements of B1
    if (BFB) {
      const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + e);
      return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                   __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
    }
    return ld4(base + e);
  };
  auto load_stage = [&](size_t r0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < VA; ++j) {
      const int idx = tid + 256 * j, row = idx / (M / 4), c4 = idx % (M / 4);
      const size_t r = r0 + row;
      va[j] = (idx < TN_STAGE * M / 4 && r < r_end) ? ld4(A + r * lda + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    unsigned t0 = 0, b0 = 0;
    if (MAPB) { t0 = (unsigned)(r0 / (unsigned)Bn); b0 = (unsigned)(r0 - (size_t)t0 * Bn); }   // uniform
#pragma unroll
    for (int j = 0; j < VB; ++j) {
      const int idx = tid + 256 * j, row = idx / (N / 4), c4 = idx % (N / 4);
      const size_t r = r0 + row;
      const bool ok = idx < TN_STAGE * N / 4 && r < r_end;
      if (MAPB) {
        unsigned t = t0, b = b0 + row;
        while (b >= (unsigned)Bn) { b -= Bn; ++t; }
        const unsigned sh = shiftB ? 1u : 0u;
        if (!ok) vb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        else if (t < sh) vb[j] = ld4(B0 + (size_t)b * ldb + 4 * c4);
        else vb[j] = ld4b(B1, ((size_t)(t - sh) * rsT + (size_t)b * rsB) * ldb + 4 * c4);
      } else if (BFB) {
        if (!ok) vb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        else if (r < shiftB) vb[j] = ld4(B0 + r * ldb + 4 * c4);
        else vb[j] = ld4b(B1, (r - shiftB) * ldb + 4 * c4);
      } else {
        const float* src = r < shiftB ? B0 + r * ldb : B1 + (r - shiftB) * ldb;   // H_prev: rows of t = 0 are h0
        vb[j] = ok ? ld4(src + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto split4 = [&](const f32x4 v, unsigned char* p0, unsigned char* p1, unsigned char* p2, unsigned off)
      __attribute__((always_inline)) {
    uint2 q0, q1, q2;
    split_quad(v, q0, q1, q2);
    *reinterpret_cast<uint2*>(p0 + off) = q0;
    *reinterpret_cast<uint2*>(p1 + off) = q1;
    *reinterpret_cast<uint2*>(p2 + off) = q2;
  };
  auto publish = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < VA; ++j) {
      const int idx = tid + 256 * j, row = idx / (M / 4), c4 = idx % (M / 4);
      if (idx < TN_STAGE * M / 4) split4(va[j], la[0], la[1], la[2], (unsigned)(row * ROWA + c4 * 8));
    }
#pragma unroll
    for (int j = 0; j < VB; ++j) {
      const int idx = tid + 256 * j, row = idx / (N / 4), c4 = idx % (N / 4);
      if (idx < TN_STAGE * N / 4) split4(vb[j], lb[0], lb[1], lb[2], (unsigned)(row * ROWB + c4 * 8));
    }
  };

  f32x4 acc[TPW];
#pragma unroll
  for (int k = 0; k < TPW; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned la0 = (unsigned)(size_t)&la[0][0], lb0 = (unsigned)(size_t)&lb[0][0];
  // transposed fragment of this lane: rows 8g + q (+4) of the stage, 4 columns at 4*pp of a 16-column tile
  const unsigned trA = la0 + (8 * g + q) * ROWA + 4 * pp * 2;
  const unsigned trB = lb0 + (8 * g + q) * ROWB + 4 * pp * 2;

  load_stage(r_begin);
  for (size_t r0 = r_begin; r0 < r_end; r0 += TN_STAGE) {
    __syncthreads();                                 // the previous stage's fragment reads are done
    publish();
    __syncthreads();
    if (r0 + TN_STAGE < r_end) load_stage(r0 + TN_STAGE);
    // Several waves share a SIMD here (2 workgroups per CU): all of a batch's fragment reads are issued,
    // into registers of their own, before its first MFMA, and the MFMAs have retired before the next batch or
    // stage reloads them (operand rule).  A wave's tiles wv, wv+4, ... share their B fragment
    // when NT divides 4 (nt = wv % NT) and their A fragment when MT == 1: those are read once.  Batches of at
    // most four tiles keep the kernel under 128 registers' worth of fragments (two workgroups per CU).
    constexpr bool A_CONST = (MT == 1), B_CONST = (4 % NT == 0);
    constexpr int BATCH = TPW < 4 ? TPW : 4;
    Frag3 fa1, fb1;
    if (A_CONST) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fa1.p[pl] = tr_frag(trA + pl * (TN_STAGE * ROWA), ROWA);
    }
    if (B_CONST) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fb1.p[pl] = tr_frag(trB + pl * (TN_STAGE * ROWB) + (wv % NT) * 32, ROWB);
    }
#pragma unroll
    for (int k0 = 0; k0 < TPW; k0 += BATCH) {
      Frag3 fa[BATCH], fb[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int tile = wv + 4 * (k0 + k);
        if (k0 + k < TPW && tile < NTILE) {            // wave-uniform
          const int mt = tile / NT, nt = tile % NT;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            if (!A_CONST) fa[k].p[pl] = tr_frag(trA + pl * (TN_STAGE * ROWA) + mt * 32, ROWA);
            if (!B_CONST) fb[k].p[pl] = tr_frag(trB + pl * (TN_STAGE * ROWB) + nt * 32, ROWB);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      float touch = 0.f;
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int tile = wv + 4 * (k0 + k);
        if (k0 + k < TPW && tile < NTILE) {
          acc[k0 + k] = mfma6(A_CONST ? fa1 : fa[k], B_CONST ? fb1 : fb[k], acc[k0 + k]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);               // (the scheduler otherwise sinks MFMAs below the read)
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int tile = wv + 4 * (k0 + k);
        if (k0 + k < TPW && tile < NTILE) touch += acc[k0 + k][0];
      }
      if (touch == 1.2345678e38f) part[0] = 1.f;       // VALU read of every accumulator: the MFMAs have retired
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // D row 4g + r of tile (mt, nt) is m = 16mt + 4g + r, column n = 16nt + (l & 15)
  float* pc = part + (size_t)blockIdx.x * M * N;
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    const int tile = wv + 4 * k;
    if (tile < NTILE) {
      const int mt = tile / NT, nt = tile % NT;
#pragma unroll
      for (int r = 0; r < 4; ++r) pc[(size_t)(mt * 16 + 4 * g + r) * N + nt * 16 + (l & 15)] = acc[k][r];
    }
  }
}

// C[idx] = sum over workgroups, fixed order.  split16: a [M, 32] result becomes two [M, r0] / [M, r1] matrices
// (columns 0..r0-1 and 16..16+r1-1: the rest is rank padding).  Otherwise the first `keep` elements are written.
__global__ __launch_bounds__(1024) void tn_reduce(int nwg, int MN, const float* __restrict__ part,
                                                  float* __restrict__ C0, float* __restrict__ C1, int split16,
                                                  int r0, int r1, int keep) {
  __shared__ float sm[16][64];
  const int o = threadIdx.x & 63, pid = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;
  float a = 0.f;
  if (idx < MN) {
    for (int wg0 = pid; wg0 < nwg; wg0 += 64) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int wg = wg0 + 16 * j; v[j] = wg < nwg ? part[(size_t)wg * MN + idx] : 0.f; }
      a += (v[0] + v[1]) + (v[2] + v[3]);
    }
  }
  sm[pid][o] = a;
  __syncthreads();
  if (pid == 0 && idx < MN) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += sm[j][o];
    if (split16) {
      const int n = idx >> 5, j = idx & 31;
      if (j < 16) { if (j < r0) C0[n * r0 + j] = t; } else if (j - 16 < r1) C1[n * r1 + (j - 16)] = t;
    } else if (idx < keep) {
      C0[idx] = t;
    }
  }
}

static inline int tn_nwg(size_t R) { return (int)((R + TN_CHUNK - 1) / TN_CHUNK); }

// bias / zeta / nu gradients of the low-rank backward: fixed-order sum over workgroups
__global__ __launch_bounds__(1024) void reduce_lowrank_small(int nwg, const float* __restrict__ part,
                                                             const float* __restrict__ zeta, const float* __restrict__ nu,
                                                             float* __restrict__ d_bz, float* __restrict__ d_bh,
                                                             float* __restrict__ d_zeta, float* __restrict__ d_nu) {
  __shared__ float sm[16][64];
  const int o = threadIdx.x & 63, pid = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;               // 0 .. 2*256+1
  float a = 0.f;
  if (idx < 2 * 256 + 2) {
    for (int wg0 = pid; wg0 < nwg; wg0 += 64) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int wg = wg0 + 16 * j; v[j] = wg < nwg ? part[(size_t)wg * SLAB_LR + idx] : 0.f; }
      a += (v[0] + v[1]) + (v[2] + v[3]);
    }
  }
  sm[pid][o] = a;
  __syncthreads();
  if (pid == 0 && idx < 2 * 256 + 2) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += sm[j][o];
    if (idx < 256) d_bz[idx] = t;
    else if (idx < 512) d_bh[idx - 256] = t;
    else if (idx == 512) { const float sz = 1.0f / (1.0f + expf(-zeta[0])); d_zeta[0] = t * sz * (1.0f - sz); }   // .cu:116,544
    else { const float sn = 1.0f / (1.0f + expf(-nu[0])); d_nu[0] = t * sn * (1.0f - sn); }                      // .cu:117,545
  }
}

// x[B][F][T] (FASTGRNN_FLAG_X_BFT, the data loader's layout, trainClassifier.py:204) <-> time-major [T][B][F].  The
// low-rank scans take the time-major copy from the workspace (the reference makes the same copy with .contiguous(),
// rnn.py:910); the dense H = 128 scans read [B,F,T] in place.  One workgroup per utterance, 64 frames at a time.
template <typename E, bool TO_TBF>
__global__ __launch_bounds__(256) void bft_transpose(int B, int T, const E* __restrict__ src, E* __restrict__ dst) {
  constexpr int F = 32;
  __shared__ E tile[F][65];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int tc = 0; tc < T; tc += 64) {
    __syncthreads();
    // all (predicated) loads first, then unconditional LDS writes: no LDS write sits in front of a branch with memory
    // instructions behind it (tools/lds_branch_vmem_scan.py)
    E v[8];
    if (TO_TBF) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int f = (tid >> 6) + 4 * k, tt = tid & 63;
        v[k] = (tc + tt < T) ? src[((size_t)b * F + f) * T + tc + tt] : E(0);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) tile[(tid >> 6) + 4 * k][tid & 63] = v[k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int tt = (tid >> 5) + 8 * k, f = tid & 31;
        v[k] = (tc + tt < T) ? src[((size_t)(tc + tt) * B + b) * F + f] : E(0);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) tile[tid & 31][(tid >> 5) + 8 * k] = v[k];
    }
    __syncthreads();
    if (TO_TBF) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int tt = (tid >> 5) + 8 * k, f = tid & 31;
        if (tc + tt < T) dst[((size_t)(tc + tt) * B + b) * F + f] = tile[f][tt];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int f = (tid >> 6) + 4 * k, tt = tid & 63;
        if (tc + tt < T) dst[((size_t)b * F + f) * T + tc + tt] = tile[f][tt];
      }
    }
  }
}

static inline int row_stride_t(const fastgrnn_desc& d) { return (d.flags & FASTGRNN_FLAG_BATCH_MAJOR) ? 1 : d.B; }
static inline int row_stride_b(const fastgrnn_desc& d) { return (d.flags & FASTGRNN_FLAG_BATCH_MAJOR) ? d.T : 1; }
static inline size_t esz(const fastgrnn_desc& d) { return d.dtype == FASTGRNN_BF16_IO ? 2 : 4; }

struct LowrankBwdWs { size_t dpre, dm, part, splitk, xt, dxt, total; };
LowrankBwdWs lowrank_bwd_layout(const fastgrnn_desc& d) {
  const size_t TB = (size_t)d.T * d.B, nwg = (d.B + 15) / 16;
  LowrankBwdWs L; size_t o = 0;
  L.dpre = o; o += align256((TB + 16) * 256 * 4);    // + 16 sink rows (lanes beyond a ragged batch)
  L.dm = o; o += align256((TB + 32) * 32 * 4);       // + 16 sink rows, + 16 more as the sink of d_x
  L.part = o; o += align256(nwg * SLAB_LR * 4);
  L.splitk = o; o += align256((size_t)tn_nwg(TB) * 256 * 32 * 4);   // partial C of the largest product, per workgroup
  L.xt = L.dxt = o;
  if (d.flags & FASTGRNN_FLAG_X_BFT) {                               // time-major copies of x and d_x
    L.xt = o; o += align256(TB * 32 * esz(d));
    L.dxt = o; o += align256(TB * 32 * esz(d));
  }
  L.total = o;
  return L;
}

template <int GATE>
void launch_bwd_lowrank_gate(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x,
                             const void* hs, const void* pre_s, const void* m_s, const void* h0,
                             const fastgrnn_grads& g, void* ws, hipStream_t s) {
  const LowrankBwdWs L = lowrank_bwd_layout(d);
  char* base = reinterpret_cast<char*>(ws);
  float* dpre = (float*)(base + L.dpre); float* dm = (float*)(base + L.dm); float* part = (float*)(base + L.part);
  float* splitk = (float*)(base + L.splitk);
  const int nwg = (d.B + 15) / 16;
  const size_t TB = (size_t)d.T * d.B;
  const bool bf = d.dtype == FASTGRNN_BF16_IO, bft = (d.flags & FASTGRNN_FLAG_X_BFT) != 0;
  const bool bm = (d.flags & FASTGRNN_FLAG_BATCH_MAJOR) != 0;
  const int rsT = row_stride_t(d), rsB = row_stride_b(d);
  const int xsT = bft ? d.B : rsT, xsB = bft ? 1 : rsB;               // x / d_x rows (time-major copies under X_BFT)
  const void* xs = x;
  void* dxs = g.d_x;
  if (bft) {
    xs = base + L.xt; dxs = base + L.dxt;
    if (bf) hipLaunchKernelGGL((bft_transpose<unsigned short, true>), dim3(d.B), dim3(256), 0, s, d.B, d.T,
                               (const unsigned short*)x, (unsigned short*)(base + L.xt));
    else hipLaunchKernelGGL((bft_transpose<float, true>), dim3(d.B), dim3(256), 0, s, d.B, d.T, (const float*)x,
                            (float*)(base + L.xt));
  }
  auto go = [&](auto kern) __attribute__((always_inline)) {
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 0, s, d.T, d.B, rsT, rsB, xsT, xsB, d.w_rank, d.u_rank,
                       (d.flags & FASTGRNN_FLAG_GRAD_LAST) ? 1 : 0, (const float*)ghs, (const float*)hs,
                       (const float*)pre_s, (const float*)m_s, (const float*)h0, (const float*)p.w1, (const float*)p.w2,
                       (const float*)p.u1, (const float*)p.u2, (const float*)p.bias_gate, (const float*)p.bias_update,
                       (const float*)p.zeta, (const float*)p.nu, (float*)dxs, (float*)g.d_h0, dpre, dm, part);
  };
  // 8 waves (two per SIMD) for full and ragged batches alike (lanes beyond a ragged batch only get a zero gradient,
  // which needs no extra registers; the first ragged variant masked five values per element and spilled)
  // pre_s == NULL: the pre-activation is recomputed from the rank-space vector (the forward then did not store it)
  if (pre_s) {
    if (bf) { if (d.B % 16) go(bwd_scan_lowrank_split<GATE, true, true>); else go(bwd_scan_lowrank_split<GATE, false, true>); }
    else    { if (d.B % 16) go(bwd_scan_lowrank_split<GATE, true, false>); else go(bwd_scan_lowrank_split<GATE, false, false>); }
  } else {
    if (bf) { if (d.B % 16) go(bwd_scan_lowrank_split<GATE, true, true, true>); else go(bwd_scan_lowrank_split<GATE, false, true, true>); }
    else    { if (d.B % 16) go(bwd_scan_lowrank_split<GATE, true, false, true>); else go(bwd_scan_lowrank_split<GATE, false, false, true>); }
  }
  if (bft) {
    if (bf) hipLaunchKernelGGL((bft_transpose<unsigned short, false>), dim3(d.B), dim3(256), 0, s, d.B, d.T,
                               (const unsigned short*)(base + L.dxt), (unsigned short*)g.d_x);
    else hipLaunchKernelGGL((bft_transpose<float, false>), dim3(d.B), dim3(256), 0, s, d.B, d.T,
                            (const float*)(base + L.dxt), (float*)g.d_x);
  }
  hipLaunchKernelGGL(reduce_lowrank_small, dim3((2 * 256 + 2 + 63) / 64), dim3(1024), 0, s, nwg, part, (const float*)p.zeta,
                     (const float*)p.nu, (float*)g.d_bias_gate, (float*)g.d_bias_update, (float*)g.d_zeta,
                     (float*)g.d_nu);
  // d_u2 | d_w2 = d_pre^T . [m_h | m_x]     (.cu:546-555, factorised); both operands are time-major fp32
  const int ng = tn_nwg(TB);
  hipLaunchKernelGGL((tn_gemm_split<16, 2>), dim3(ng), dim3(256), 0, s, TB, dpre, 256, (const float*)m_s,
                     (const float*)m_s, (size_t)0, 32, splitk, 1, 0, 0);
  hipLaunchKernelGGL(tn_reduce, dim3(256 * 32 / 64), dim3(1024), 0, s, ng, 256 * 32, splitk, (float*)g.d_u2,
                     (float*)g.d_w2, 1, d.u_rank, d.w_rank, 0);
  // d_u1 = d_m_h^T . H_prev  (rows of t = 0 are h0, the rest hs[t-1]);  d_w1 = d_m_x^T . X
  auto tn = [&](auto kern, const float* A, const void* B0, const void* B1, size_t shiftB, int ldb, int rT, int rB)
      __attribute__((always_inline)) {
    hipLaunchKernelGGL(kern, dim3(ng), dim3(256), 0, s, TB, A, 32, (const float*)B0, (const float*)B1, shiftB, ldb,
                       splitk, d.B, rT, rB);
  };
  if (bm) { if (bf) tn(tn_gemm_split<1, 16, true, true>, dm, h0, hs, (size_t)d.B, 256, rsT, rsB);
            else    tn(tn_gemm_split<1, 16, true, false>, dm, h0, hs, (size_t)d.B, 256, rsT, rsB); }
  else    { if (bf) tn(tn_gemm_split<1, 16, false, true>, dm, h0, hs, (size_t)d.B, 256, rsT, rsB);
            else    tn(tn_gemm_split<1, 16, false, false>, dm, h0, hs, (size_t)d.B, 256, rsT, rsB); }
  hipLaunchKernelGGL(tn_reduce, dim3(16 * 256 / 64), dim3(1024), 0, s, ng, 16 * 256, splitk, (float*)g.d_u1,
                     (float*)nullptr, 0, 0, 0, d.u_rank * 256);
  const bool xmap = bm && !bft;
  if (xmap) { if (bf) tn(tn_gemm_split<1, 2, true, true>, dm + 16, xs, xs, (size_t)0, 32, xsT, xsB);
              else    tn(tn_gemm_split<1, 2, true, false>, dm + 16, xs, xs, (size_t)0, 32, xsT, xsB); }
  else      { if (bf) tn(tn_gemm_split<1, 2, false, true>, dm + 16, xs, xs, (size_t)0, 32, xsT, xsB);
              else    tn(tn_gemm_split<1, 2, false, false>, dm + 16, xs, xs, (size_t)0, 32, xsT, xsB); }
  hipLaunchKernelGGL(tn_reduce, dim3(16 * 32 / 64), dim3(1024), 0, s, ng, 16 * 32, splitk, (float*)g.d_w1,
                     (float*)nullptr, 0, 0, 0, d.w_rank * 32);
}

template <int GATE>
void launch_fwd_lowrank_gate(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs,
                             void* zs, void* cs, void* ws, hipStream_t s) {
  dim3 grid((d.B + 15) / 16), block(512);
  const bool ragged = (d.B % 16) != 0, bf = d.dtype == FASTGRNN_BF16_IO, bft = (d.flags & FASTGRNN_FLAG_X_BFT) != 0;
  const int rsT = row_stride_t(d), rsB = row_stride_b(d);
  const int xsT = bft ? d.B : rsT, xsB = bft ? 1 : rsB;
  if (bft) {
    if (bf) hipLaunchKernelGGL((bft_transpose<unsigned short, true>), dim3(d.B), dim3(256), 0, s, d.B, d.T,
                               (const unsigned short*)x, (unsigned short*)ws);
    else hipLaunchKernelGGL((bft_transpose<float, true>), dim3(d.B), dim3(256), 0, s, d.B, d.T, (const float*)x, (float*)ws);
    x = ws;
  }
  auto go = [&](auto kern) __attribute__((always_inline)) {
    hipLaunchKernelGGL(kern, grid, block, 0, s, d.T, d.B, rsT, rsB, xsT, xsB, d.w_rank, d.u_rank, (const float*)x,
                       (const float*)h0, (const float*)p.w1, (const float*)p.w2, (const float*)p.u1, (const float*)p.u2,
                       (const float*)p.bias_gate, (const float*)p.bias_update, (const float*)p.zeta, (const float*)p.nu,
                       (float*)hs, (float*)zs, (float*)cs);
  };
  // SAVE_PREACT with z_s == NULL: only the rank-space vector is saved (the backward recomputes the pre-activation)
  const bool preact = (d.flags & FASTGRNN_FLAG_SAVE_PREACT) != 0;
  const int aux = zs == nullptr ? ((preact && cs) ? 3 : 0) : (preact ? 2 : 1);
  auto pick = [&](auto bf_tag) __attribute__((always_inline)) {
    constexpr bool BFv = decltype(bf_tag)::value;
    if (d.flags & FASTGRNN_FLAG_HS_LAST) {           // inference: aux == 0 (lowrank_forward)
      if (ragged) go(fwd_scan_lowrank_split<GATE, 0, true, BFv, true>); else go(fwd_scan_lowrank_split<GATE, 0, false, BFv, true>);
    } else if (aux == 2) {
      if (ragged) go(fwd_scan_lowrank_split<GATE, 2, true, BFv>); else go(fwd_scan_lowrank_split<GATE, 2, false, BFv>);
    } else if (aux == 3) {
      if (ragged) go(fwd_scan_lowrank_split<GATE, 3, true, BFv>); else go(fwd_scan_lowrank_split<GATE, 3, false, BFv>);
    } else if (aux == 0) {
      if (ragged) go(fwd_scan_lowrank_split<GATE, 0, true, BFv>); else go(fwd_scan_lowrank_split<GATE, 0, false, BFv>);
    } else if constexpr (!BFv) {                     // the reference's (z_s, h_prime_s) outputs: fp32 sequences
      if (ragged) go(fwd_scan_lowrank_split<GATE, 1, true, false>); else go(fwd_scan_lowrank_split<GATE, 1, false, false>);
    }
  };
  if (bf) pick(std::true_type{}); else pick(std::false_type{});
}

}  // namespace

// H = 256, F = 32, both factorisations with rank 1..16 (zero-extended to 16 in the kernels).  Larger ranks and
// half-factorised cells (only W or only U low-rank, rnn.py:783-798) stay on the generic scan.
bool lowrank_shape(const fastgrnn_desc& d) {
  return d.H == 256 && d.F == 32 && d.w_rank >= 1 && d.w_rank <= 16 && d.u_rank >= 1 && d.u_rank <= 16;
}

bool lowrank_supported(const fastgrnn_desc& d, int direction) {
  if (d.gate_nl > FASTGRNN_NL_TANH || d.update_nl != FASTGRNN_NL_TANH) return false;
  // the backward scan addresses a step's rows with 32-bit offsets: whole sequence tensors below 2^32 bytes
  if (((double)d.T * d.B + 16.0) * 256 * 4.0 >= 4294967296.0) return false;   // (the workspace tensors carry 16 sink rows)
  const bool preact = (d.flags & FASTGRNN_FLAG_SAVE_PREACT) != 0;
  if (direction == 0) {
    if ((d.flags & FASTGRNN_FLAG_HS_LAST) && preact) return false;
    return true;
  }
  // backward: one-saved-tensor contract only (the reference-style backward with z_s / h_prime_s is the generic scan)
  return preact;
}

size_t lowrank_forward_ws(const fastgrnn_desc& d) {
  return (d.flags & FASTGRNN_FLAG_X_BFT) ? align256((size_t)d.T * d.B * 32 * esz(d)) : 0;
}

size_t lowrank_backward_ws(const fastgrnn_desc& d) { return lowrank_bwd_layout(d).total; }

int lowrank_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs, void* zs,
                    void* cs, void* ws, hipStream_t s) {
  const bool preact = (d.flags & FASTGRNN_FLAG_SAVE_PREACT) != 0;
  if (preact && !cs) return FASTGRNN_ERR_NULL_POINTER;   // (z_s may be NULL: the rank-space vector alone is saved)
  if ((d.flags & FASTGRNN_FLAG_HS_LAST) && (zs || (preact && cs))) return FASTGRNN_ERR_UNSUPPORTED;
  if (d.dtype == FASTGRNN_BF16_IO && zs && !preact) return FASTGRNN_ERR_UNSUPPORTED;
  if ((d.flags & FASTGRNN_FLAG_X_BFT) && !ws) return FASTGRNN_ERR_WORKSPACE;
  switch (d.gate_nl) {
    case FASTGRNN_NL_SIGMOID: launch_fwd_lowrank_gate<FASTGRNN_NL_SIGMOID>(d, p, x, h0, hs, zs, cs, ws, s); break;
    case FASTGRNN_NL_RELU: launch_fwd_lowrank_gate<FASTGRNN_NL_RELU>(d, p, x, h0, hs, zs, cs, ws, s); break;
    default: launch_fwd_lowrank_gate<FASTGRNN_NL_TANH>(d, p, x, h0, hs, zs, cs, ws, s); break;
  }
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

int lowrank_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x, const void* hs,
                     const void* zs, const void* cs, const void* h0, const fastgrnn_grads& g, void* ws, hipStream_t s) {
  if (!cs) return FASTGRNN_ERR_NULL_POINTER;         // the rank-space vector saved by the forward
  switch (d.gate_nl) {
    case FASTGRNN_NL_SIGMOID: launch_bwd_lowrank_gate<FASTGRNN_NL_SIGMOID>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    case FASTGRNN_NL_RELU: launch_bwd_lowrank_gate<FASTGRNN_NL_RELU>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    default: launch_bwd_lowrank_gate<FASTGRNN_NL_TANH>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
  }
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

// fp32 [B,32,T] <-> [T,B,32] for the dense H = 256 layer's backward (kernels_h256.hip): its dW GEMM and d_x GEMM work on
// time-major rows
void bft_transpose_f32(int B, int T, const float* src, float* dst, bool to_time_major, hipStream_t s) {
  if (to_time_major) hipLaunchKernelGGL((bft_transpose<float, true>), dim3(B), dim3(256), 0, s, B, T, src, dst);
  else hipLaunchKernelGGL((bft_transpose<float, false>), dim3(B), dim3(256), 0, s, B, T, src, dst);
}

}  // namespace fastgrnn
