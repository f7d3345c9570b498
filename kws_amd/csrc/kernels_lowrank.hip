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
      } else if (AUX == 2) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) st4(zs + o + 4 * mt, gt.z[mt]);      // gt.z carries the pre-activation
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
    if (AUX == 2) { gout.mlo = mlo; gout.mhi = mhi; }                 // stored with the step's other outputs
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
// operand straight from registers; the eight partials meet in LDS; d_h = z*g + U1^T d_m_h for the wave's own
// units and d_x = W1^T d_m_x.
// The factor gradients (.cu:546-555, evaluated factorised) are contracted INSIDE the scan since round 3 (they were
// three split-K GEMMs over a d_pre[T,B,H] / d_m[T,B,32] round trip through HBM: 1.9 GB of the 2.8 GB the backward
// moved).  Every product sums over the 16 utterances of the workgroup, which sit on the lanes, so both operands go
// through the hardware-transposed LDS read -- but each wave only ever needs ITS OWN 32 units:
//     d_u2|d_w2 [own unit][j] += d_pre_t[b][own unit] . m_t[b][j]         (2 x 2 tiles)
//     d_u1^T    [own unit][j] += h_{t-1}[b][own unit] . d_m_h,t[b][j]     (2 tiles)
//     d_w1^T    [f][j]        += x_t[b][f]            . d_m_x,t[b][j]     (2 tiles: waves 2 and 3, one each)
// so the images of d_pre, h_prev and d_m are wave-PRIVATE (one 3 KB scratch per wave, used one after the other: LDS
// instructions of a wave execute in order); the images of m_t and x_t -- the same for every wave -- are filled by
// all 512 threads, one value each, in front of the step's first barrier.  K = 16 utterances is exactly one
// v_mfma_f32_16x16x16_bf16 (lane group g holds k = 4g..4g+3 -- what ONE transposed read returns -- in two registers
// per plane; same issue rate as the K = 32 form, tools/mfma_k16_probe.hip); the rank-space products d_h, d_x (K = 16
// rank rows) use the same form.  Accumulators (28 registers) live for all T, per-workgroup slabs, one deterministic
// reduce (reduce_lowrank_slabs).
// Ranks below 16: the factors are zero-extended when they are loaded, as in the forward.  Row (t, b) of grad_hs / hs /
// pre_s is t*rsT + b*rsB, of x / d_x t*xsT + b*xsB; the rank-space vector m_s is time-major.  glast: grad_hs is [B,H], the
// gradient of the last state only (FASTGRNN_FLAG_GRAD_LAST).  BF: grad_hs, hs, x and d_x are bf16 (h0, pre_s, m_s fp32).
// slab of one workgroup (floats): d_u2|d_w2 [256][32] | d_u1^T [256][16] | d_w1^T [32][16] | d_bz[256] | d_bh[256] | (zeta, nu)
constexpr int SL_UW2 = 0, SL_U1T = 256 * 32, SL_W1T = SL_U1T + 256 * 16, SL_BZ = SL_W1T + 32 * 16, SL_ZN = SL_BZ + 512;
constexpr int SLAB_LR = ((SL_ZN + 2 + 63) / 64) * 64;

template <int GATE, bool RAGGED, bool BF = false>
__global__ __launch_bounds__(512) void bwd_scan_lowrank_split(
    int Tn, int B, int rsT, int rsB, int xsT, int xsB, int rw, int ru, int glast,
    const float* __restrict__ ghs, const float* __restrict__ x, const float* __restrict__ hs,
    const float* __restrict__ pre_s, const float* __restrict__ m_s,
    const float* __restrict__ h0, const float* __restrict__ w1, const float* __restrict__ w2,
    const float* __restrict__ u1, const float* __restrict__ u2,
    const float* __restrict__ bz, const float* __restrict__ bh,
    const float* __restrict__ zeta, const float* __restrict__ nu,
    float* __restrict__ d_x, float* __restrict__ d_h0, float* __restrict__ dx_sink, float* __restrict__ part) {
  // 8 waves (two per SIMD), 32 units each (two 16-row tiles, one K-step over the own units): see fwd_scan_lowrank_split
  constexpr int H = 256, F = 32, NW = 8, UPW = H / NW, NT = UPW / 16, UPL = UPW / 4, MROW = 36;
  constexpr int IMG_PLANE = 1024;                    // one plane of an image: 16 utterances x 64 bytes
  // rank-space partials of the eight waves, by step parity (ONE barrier per step: a wave may write the next step's
  // partial while another still sums this step's)
  __shared__ __attribute__((aligned(16))) float mp[2][NW][16][MROW];
  __shared__ __attribute__((aligned(16))) float sbias[2][H];
  __shared__ float red[2 * NW];
  __shared__ __attribute__((aligned(16))) unsigned char wimg[NW][3 * IMG_PLANE];   // wave-private: d_pre_t, h_{t-1}, d_m_t
  __shared__ __attribute__((aligned(16))) unsigned char simg[2][2][3 * IMG_PLANE]; // shared, by step parity: m_t [b][32 j], x_t [b][32 f]
#ifdef FASTGRNN_DIAG_STAMPS
  unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dlast)::"memory");
#endif
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, i = l & 15, g = l >> 4;
  const int b = blockIdx.x * 16 + i;
  const bool valid = !RAGGED || b < B;
  const int bc = valid ? b : B - 1;
  const int n0 = wv * UPW + g * UPL;
  const bool xwave = (wv >> 1) == 1;                 // (wave-uniform) waves 2, 3: d_w1, feature tile wv & 1
  // (wave-uniform: kept in scalar registers)
  const float sz = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fsigmoid(zeta[0]))));
  const float sn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fsigmoid(nu[0]))));
  if (tid < H) { sbias[0][tid] = bz[tid]; sbias[1][tid] = bh[tid]; }

  // ---- resident A operands -------------------------------------------------------------------
  // (Round 3's first in-scan build parked all three in LDS and re-read them every step: with the operand set refilled
  // by buffer loads and the rank-space products on the K = 16 MFMA form they fit beside everything else again.)
  // d_m[j][b] = sum_n [U2|W2][n][j] d_pre[b][n] over own units (K = 32): tile 0 rows = U2 columns, tile 1 = W2 columns
  Frag3 UW2Tf[2];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl) {
    const float* src = tl == 0 ? u2 : w2;
    const int rk = tl == 0 ? ru : rw;
    const int ic = i < rk ? i : 0;
    f32x4 lo, hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo[j] = i < rk ? src[(size_t)(n0 + j) * rk + ic] : 0.f;
      hi[j] = i < rk ? src[(size_t)(n0 + 4 + j) * rk + ic] : 0.f;
    }
    UW2Tf[tl] = split3(lo, hi);
  }
  // d_h[k][b] = z*g + sum_j U1[j][k] d_m_h[j][b]: rows = own units (row 4c + r of tile mt = unit 8c + 4mt + r of the
  // wave, the unit lane (b, c) holds as element r of tile mt), K = the 16 rank rows: lane group g holds j = 4g..4g+3
  Half3 U1Tf[NT], W1Tf;
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
    const int kA = wv * UPW + (i >> 2) * UPL + mt * 4 + (i & 3);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = 4 * g + j < ru ? u1[(size_t)(4 * g + j) * H + kA] : 0.f;
    uint2 q0, q1, q2;
    split_quad(v, q0, q1, q2);
    U1Tf[mt].p[0] = __builtin_bit_cast(s16x4, q0); U1Tf[mt].p[1] = __builtin_bit_cast(s16x4, q1); U1Tf[mt].p[2] = __builtin_bit_cast(s16x4, q2);
  }
  // d_x[f][b] = sum_j W1[j][f] d_m_x[j][b]: feature tile wv & 1 (used by waves 0, 1), K = the 16 rank rows
  {
    const int f = (wv & 1) * 16 + i;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = 4 * g + j < rw ? w1[(size_t)(4 * g + j) * F + f] : 0.f;
    uint2 q0, q1, q2;
    split_quad(v, q0, q1, q2);
    W1Tf.p[0] = __builtin_bit_cast(s16x4, q0); W1Tf.p[1] = __builtin_bit_cast(s16x4, q1); W1Tf.p[2] = __builtin_bit_cast(s16x4, q2);
  }

  f32x4 sbz[NT], sbh[NT], dh[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) { sbz[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; sbh[mt] = sbz[mt]; dh[mt] = sbz[mt]; }
  float pz = 0.f, pn = 0.f, pz_c = 0.f, pn_c = 0.f;     // d_zeta / d_nu partial sums, compensated
  // factor-gradient accumulators, resident for all T: [row tile of own units][U2 | W2 columns], d_u1^T, d_w1^T
  f32x4 accUW[NT][2], accU1[NT], accW1;
#pragma unroll
  for (int a = 0; a < NT; ++a) { accUW[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accUW[a][1] = accUW[a][0]; accU1[a] = accUW[a][0]; }
  accW1 = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- plane images ---------------------------------------------------------------------------
  // natural order [utterance][32 columns] bf16, 64-byte rows, one plane per KB.  Private image: a lane writes its 8
  // values of utterance i (16 bytes per plane) and reads back, transposed, utterances 4g..4g+3 of column 16*tile + i.
  unsigned char* const my_img = &wimg[wv][0];
  const unsigned wr_off = (unsigned)(i * 64 + g * 16);
  const unsigned tr_lane = (unsigned)((4 * g + (i >> 2)) * 64 + (i & 3) * 8);
  const unsigned tr_off = (unsigned)(size_t)my_img + tr_lane;
  const unsigned trs_off = (unsigned)(size_t)&simg[0][0][0] + tr_lane;
  auto img_fence = []() __attribute__((always_inline)) { asm volatile("" ::: "memory"); };   // (compiler ordering only)
  auto put_img = [&](const Frag3& f) __attribute__((always_inline)) {
    img_fence();
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(my_img + pl * IMG_PLANE + wr_off) = f.p[pl];
    img_fence();
  };
  auto tr_half_at = [&](unsigned addr, Half3& f) __attribute__((always_inline)) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      f.p[pl] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>((size_t)(addr + pl * IMG_PLANE)));
  };
  auto tr_half = [&](int tile, Half3& f) __attribute__((always_inline)) { tr_half_at(tr_off + tile * 32, f); };
  // Shared images: thread (su, sc) = (tid >> 5, tid & 31) owns value sc of utterance su of m_t and of x_t
  const int su = tid >> 5, sc = tid & 31;
  const int sb = blockIdx.x * 16 + su, sbc = (!RAGGED || sb < B) ? sb : B - 1;
  unsigned char* const my_s = &simg[0][0][0] + su * 64 + sc * 2;
  auto fresh_lane = [&]() __attribute__((always_inline)) { unsigned v = (unsigned)l; asm volatile("" : "+v"(v)); return v; };

  // grad_hs, pre-activation, h_prev of the lane's 8 units; one value of the step's rank-space vector and frame
  struct EwOps { f32x4 g[NT], a0[NT], h[NT]; float m1, x1; };
  // Addresses: buffer loads -- a resource descriptor (four scalar registers per tensor), a wave-uniform step offset in
  // a scalar register and ONE loop-invariant 32-bit byte offset per lane, shared by every stream with the same row
  // geometry.  (Left to itself the compiler keeps a 64-bit pointer pair per stream in vector registers and advances
  // each per step: sixteen registers this kernel does not have.  The host keeps whole tensors below 2^32 bytes.)
  constexpr unsigned ESZ = BF ? 2u : 4u;             // bytes per sequence element (grad_hs, hs, x, d_x)
  auto rsrc_of = [](const void* p) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t r_g = rsrc_of(ghs), r_hs = rsrc_of(hs), r_pre = rsrc_of(pre_s), r_m = rsrc_of(m_s),
                               r_x = rsrc_of(x), r_h0 = rsrc_of(h0);
  const unsigned lane_h = (unsigned)bc * rsB * H + n0, lane_0 = (unsigned)bc * H + n0;   // elements
  const unsigned lane_g = glast ? lane_0 : lane_h;
  const unsigned lane_m1 = ((unsigned)sbc * 32 + sc) * 4u, lane_x1 = ((unsigned)sbc * xsB * F + sc) * ESZ;   // bytes
  // d_x: every lane stores, lanes beyond a ragged batch into 16 sink rows of the workspace (step stride 0)
  char* const dx_base = valid ? reinterpret_cast<char*>(d_x) + ((size_t)b * xsB * F + (wv & 1) * 16 + 4 * g) * ESZ
                              : reinterpret_cast<char*>(dx_sink) + (i * 32 + (wv & 1) * 16 + 4 * g) * 4u;
  const unsigned dx_step = valid ? (unsigned)xsT * F * ESZ : 0u;
  auto bld4 = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) __attribute__((always_inline)) -> f32x4 {   // 4 floats
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
  };
  auto bld4s = [&](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) __attribute__((always_inline)) -> f32x4 {   // 4 sequence elements
    if (BF) {
      const uint2 q = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
      return f32x4{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u),
                   __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
    }
    return bld4(r, voff, soff);
  };
  // The refill of the operand set is issued in three parts, each right behind the last use of what it overwrites
  // (requests spread over the step queue less in the CU's one address unit than a burst of all eight waves' requests)
  auto load_ew_ga = [&](int t, EwOps& e, int mt) __attribute__((always_inline)) {    // grad_hs, pre-activation of row tile mt
    // wave-uniform offsets of step t's rows (scalar registers)
    const unsigned step = (unsigned)t * (unsigned)rsT * H;                   // elements
    const unsigned gstep = glast ? 0u : step;
    const bool gzero = (RAGGED && !valid) || (glast && t != Tn - 1);
    // lanes beyond a ragged batch: the last utterance's rows with a ZERO gradient (dh starts at zero, so gg,
    // d_pre and every sum or product they enter stay exactly zero for them; see bwd_scan_split_w8)
    const f32x4 gv = bld4s(r_g, lane_g * ESZ, (gstep + 4 * mt) * ESZ);
    e.g[mt] = gzero ? f32x4{0.f, 0.f, 0.f, 0.f} : gv;
    e.a0[mt] = bld4(r_pre, lane_h * 4u, (step + 4 * mt) * 4u);
  };
  auto load_ew_h = [&](int t, EwOps& e) __attribute__((always_inline)) {     // h_prev
    const unsigned hstep = (unsigned)t * (unsigned)rsT * H - (unsigned)rsT * H;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      // h_prev of step 0 is h0 (.cu:478-481).  fp32: one load whose descriptor and offsets are selected in scalar
      // registers (no branch); bf16 sequences: h0 is fp32, a different load
      if (BF) {
        if (t == 0) e.h[mt] = bld4(r_h0, lane_0 * 4u, 16u * mt);
        else e.h[mt] = bld4s(r_hs, lane_h * ESZ, (hstep + 4 * mt) * ESZ);
      } else {
        e.h[mt] = bld4(t == 0 ? r_h0 : r_hs, (t == 0 ? lane_0 : lane_h) * 4u, (t == 0 ? 4u * mt : hstep + 4 * mt) * 4u);
      }
    }
  };
  auto load_ew_mx = [&](int t, EwOps& e) __attribute__((always_inline)) {    // one value of m_t and of x_t
    e.m1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_m, (int)lane_m1, (int)((unsigned)t * (unsigned)B * 128u), 0));
    const unsigned xstep = (unsigned)t * (unsigned)xsT * F * ESZ;
    if (BF) e.x1 = bf16_to_f32(__builtin_amdgcn_raw_buffer_load_b16(r_x, (int)lane_x1, (int)xstep, 0));
    else e.x1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_x, (int)lane_x1, (int)xstep, 0));
  };
  auto load_ew = [&](int t, EwOps& e) __attribute__((always_inline)) { load_ew_ga(t, e, 0); load_ew_ga(t, e, 1); load_ew_h(t, e); load_ew_mx(t, e); };

  // Transposed operands of the step's factor-gradient products.  They are loop-carried on purpose: those products
  // trail behind the recurrence's (nothing of it depends on them) and run in the matrix pipe UNDER the next step's
  // element-wise work; their completion read sits behind that work, and until then these registers stay allocated
  // (operand rule)
  Half3 Adp[NT], Bm[2], Ahp[NT], Bdm, Ax, Bdmx;
  {
    const s16x4 z4 = s16x4{0, 0, 0, 0};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      Adp[0].p[pl] = z4; Adp[1].p[pl] = z4; Bm[0].p[pl] = z4; Bm[1].p[pl] = z4;
      Ahp[0].p[pl] = z4; Ahp[1].p[pl] = z4; Bdm.p[pl] = z4; Ax.p[pl] = z4; Bdmx.p[pl] = z4;
    }
  }
  auto trailing_done = [&]() __attribute__((always_inline)) {
    // read of the youngest factor-gradient accumulator = every MFMA issued so far has retired (they retire in order)
    float tie = accU1[NT - 1][0];
    asm volatile("" : "+v"(tie)
                 : "v"(Adp[0].p[0]), "v"(Adp[0].p[1]), "v"(Adp[0].p[2]), "v"(Adp[1].p[0]), "v"(Adp[1].p[1]), "v"(Adp[1].p[2]),
                   "v"(Bm[0].p[0]), "v"(Bm[0].p[1]), "v"(Bm[0].p[2]), "v"(Bm[1].p[0]), "v"(Bm[1].p[1]), "v"(Bm[1].p[2]),
                   "v"(Ahp[0].p[0]), "v"(Ahp[0].p[1]), "v"(Ahp[0].p[2]), "v"(Ahp[1].p[0]), "v"(Ahp[1].p[1]), "v"(Ahp[1].p[2]),
                   "v"(Bdm.p[0]), "v"(Bdm.p[1]), "v"(Bdm.p[2]), "v"(Ax.p[0]), "v"(Ax.p[1]), "v"(Ax.p[2]),
                   "v"(Bdmx.p[0]), "v"(Bdmx.p[1]), "v"(Bdmx.p[2]));
    completion_read(tie);
    __builtin_amdgcn_sched_barrier(0);
  };
  // The factor-gradient products of a step (.cu:546-555): d_u2 | d_w2 += d_pre^T . [m_h | m_x], d_w1^T += x^T . d_m_x,
  // d_u1^T += h_prev^T . d_m_h -- the last in EVERY wave, so that one read of accU1 (trailing_done) proves that all of
  // them have retired.  Every wave issues the d_w1 product (its operands stay zero outside waves 2 and 3): no branch
  // inside the block these MFMAs share with the element-wise work.
  auto trailing_products_a = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) accUW[a][c] = mfma6_k16(Adp[a], Bm[c], accUW[a][c]);
  };
  auto trailing_products_b = [&]() __attribute__((always_inline)) {
    accW1 = mfma6_k16(Ax, Bdmx, accW1);
#pragma unroll
    for (int a = 0; a < NT; ++a) accU1[a] = mfma6_k16(Ahp[a], Bdm, accU1[a]);
  };
  constexpr int N_TRAILING_A = NT * 2 * 6, N_TRAILING_B = (1 + NT) * 6;
  auto step = [&](int t, EwOps& e) __attribute__((always_inline)) {
    const int cur = t & 1;
    // ---- the PREVIOUS step's factor-gradient products are issued one by one BETWEEN the instructions of EW(t): two waves
    //      of a SIMD in the same phase do not overlap each other's matrix and vector work (a wave waits at the issue of
    //      an MFMA while the pipe is busy), the instruction stream of ONE wave does, for half of each MFMA's cycles
    // ---- EW(t): .cu:107-117 ------------------------------------------------------------------
    f32x4 dpv[NT];
    float sz8 = 0.f, sn8 = 0.f;
    auto ew_tile = [&](int mt) __attribute__((always_inline)) {
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
    };
    trailing_products_a(); trailing_products_b();
    ew_tile(0); ew_tile(1);
    kahan_add(pz, pz_c, sz8); kahan_add(pn, pn_c, sn8);
    // (the element-wise results are pinned HERE, in front of the scheduling directives: arithmetic is not ordered
    // against them otherwise and would be emitted behind the block's closing barrier)
    asm volatile("" : "+v"(dpv[0]), "+v"(dpv[1]), "+v"(dh[0]), "+v"(dh[1]), "+v"(sbz[0]), "+v"(sbz[1]), "+v"(sbh[0]),
                 "+v"(sbh[1]), "+v"(pz), "+v"(pn), "+v"(pz_c), "+v"(pn_c));
#pragma unroll
    for (int k = 0; k < N_TRAILING_A + N_TRAILING_B; ++k) {   // one MFMA, three vector instructions (the transcendentals go where they fit), ...
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    trailing_done();                                 // ... all retired: their operand registers are free from here
    SPLIT_STAMP(0)
    // ---- planes of d_pre_t (B operand of the rank-space partial as they are; transposed, the A operand of the
    //      d_u2|d_w2 products) and of h_{t-1} through the private image; one value each of m_t and x_t into the shared ones
    // (EW(t) has consumed them; the previous step's MFMAs have retired.  No branch around any of the refills: step 0
    // re-requests its own operands -- second rule of DESIGN.md 4.0, these requests follow LDS writes)
    const int tn = t > 0 ? t - 1 : 0;
    load_ew_ga(tn, e, 0); load_ew_ga(tn, e, 1);
    __builtin_amdgcn_sched_barrier(0);
    const Frag3 dfr = split3(dpv[0], dpv[1]);
    put_img(dfr);
    tr_half(0, Adp[0]); tr_half(1, Adp[1]);
    put_img(split3(e.h[0], e.h[1]));                 // (stays in the image until the products behind the barrier want it)
    __builtin_amdgcn_sched_barrier(0);
    if (BF) lds_writes_landed();                     // (bf16 sequences: the fp32 h0 of step 0 is a branch of its own)
    load_ew_h(tn, e);
    __builtin_amdgcn_sched_barrier(0);
    {
      unsigned char* const so = my_s + cur * (6 * IMG_PLANE);
      unsigned short s0, s1, s2;
      split_one(e.m1, s0, s1, s2);
      *reinterpret_cast<unsigned short*>(so) = s0;
      *reinterpret_cast<unsigned short*>(so + IMG_PLANE) = s1;
      *reinterpret_cast<unsigned short*>(so + 2 * IMG_PLANE) = s2;
      split_one(e.x1, s0, s1, s2);
      *reinterpret_cast<unsigned short*>(so + 3 * IMG_PLANE) = s0;
      *reinterpret_cast<unsigned short*>(so + 4 * IMG_PLANE) = s1;
      *reinterpret_cast<unsigned short*>(so + 5 * IMG_PLANE) = s2;
    }
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STAMP(1)
    load_ew_mx(tn, e);
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STAMP(2)
    // ---- rank-space partial over own units: B operand = this lane's fragment of d_pre ----
    // big and small terms in accumulators of their own, as in the dense scans (mfma6_hl: inside one MFMA the
    // addends are chopped at the largest one, a one-signed loss that showed in d_zeta / d_nu at B = 4096)
    f32x4 mh = f32x4{0.f, 0.f, 0.f, 0.f}, mx = mh, mhl = mh, mxl = mh;
    mfma6_hl(UW2Tf[0], dfr, mh, mhl);
    mfma6_hl(UW2Tf[1], dfr, mx, mxl);
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<f32x4*>(&mp[cur][wv][i][4 * g]) = mh + mhl;
    *reinterpret_cast<f32x4*>(&mp[cur][wv][i][16 + 4 * g]) = mx + mxl;
    SPLIT_STAMP(3)
    lds_barrier();
    SPLIT_STAMP(4)
    // ---- behind the barrier: operands of everything that follows, then [d_m_h | d_m_x]_t of utterance i (rank rows
    //      4g..4g+3 of either half) as the sum of the eight partials, in wave order
    const unsigned trs_cur = trs_off + cur * (6 * IMG_PLANE);
    tr_half_at(trs_cur, Bm[0]); tr_half_at(trs_cur + 32, Bm[1]);
    if (xwave) tr_half_at(trs_cur + 3 * IMG_PLANE + (wv & 1) * 32, Ax);
    f32x4 dmh = f32x4{0.f, 0.f, 0.f, 0.f}, dmx = dmh;
#pragma unroll
    for (int w2i = 0; w2i < NW; ++w2i) dmh += *reinterpret_cast<const f32x4*>(&mp[cur][w2i][i][4 * g]);
    if (wv < 4) {                                    // (wave-uniform) d_x in waves 0, 1; d_w1 in waves 2, 3
#pragma unroll
      for (int w2i = 0; w2i < NW; ++w2i) dmx += *reinterpret_cast<const f32x4*>(&mp[cur][w2i][i][16 + 4 * g]);
    }
    Half3 mBh, mBx;
    {
      uint2 q0, q1, q2;
      split_quad(dmh, q0, q1, q2);
      mBh.p[0] = __builtin_bit_cast(s16x4, q0); mBh.p[1] = __builtin_bit_cast(s16x4, q1); mBh.p[2] = __builtin_bit_cast(s16x4, q2);
      split_quad(dmx, q0, q1, q2);
      mBx.p[0] = __builtin_bit_cast(s16x4, q0); mBx.p[1] = __builtin_bit_cast(s16x4, q1); mBx.p[2] = __builtin_bit_cast(s16x4, q2);
    }
    // ---- d_old_h for the own units (C-in = z*g) and d_x ---------------------------------------------
    {
      __builtin_amdgcn_sched_barrier(0);
      f32x4 dlo[NT];
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) { dlo[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; mfma6_hl_k16(U1Tf[mt], mBh, dh[mt], dlo[mt]); }
      if (wv < 2) {                                  // (wave-uniform) feature tile wv
        f32x4 dxv = f32x4{0.f, 0.f, 0.f, 0.f}, dxl = dxv;
        mfma6_hl_k16(W1Tf, mBx, dxv, dxl);
        dxv += dxl;
        {                                            // lanes beyond a ragged batch: the sink rows (dx_base)
          char* o = dx_base + (size_t)t * dx_step;
          if (BF) st4_bf16(o, dxv); else *reinterpret_cast<f32x4*>(o) = dxv;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) dh[mt] += dlo[mt];
      // (this sum read the youngest of the products above: their operand registers are free for the reads below)
      asm volatile("" : "+v"(dh[0]), "+v"(dh[1]));
      __builtin_amdgcn_sched_barrier(0);
    }
    SPLIT_STAMP(5)
    // ---- operands of the step's factor-gradient products (issued inside the next step's element-wise work):
    // h_{t-1} comes back out of the private image, transposed; then the planes of [d_m_h | d_m_x]_t go through it:
    // the B operand of d_u1^T (columns 0..15) and, in waves 2 and 3, of d_w1^T (16..31)
    tr_half(0, Ahp[0]); tr_half(1, Ahp[1]);          // (in front of the image's next use: LDS runs a wave's instructions in order)
    img_fence();
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<s16x4*>(my_img + pl * IMG_PLANE + i * 64 + g * 8) = mBh.p[pl];
    if (xwave) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<s16x4*>(my_img + pl * IMG_PLANE + i * 64 + 32 + g * 8) = mBx.p[pl];
    }
    img_fence();
    tr_half(0, Bdm);
    if (xwave) tr_half(1, Bdmx);
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STAMP(6)
  };

  EwOps ea;                                          // ONE operand set, refilled right behind its use (see step)
  __syncthreads();                                   // sbias
  load_ew(Tn - 1, ea);
  for (int t = Tn - 1; t >= 0; --t) step(t, ea);
  trailing_products_a(); trailing_products_b();      // step 0's
  __builtin_amdgcn_sched_barrier(0);
#ifdef FASTGRNN_DIAG_STAMPS
  if (blockIdx.x == 7 && (tid & 63) == 0) { for (int k2 = 0; k2 < 8; ++k2) g_sdiag[wv][k2] = dsum[k2]; }
#endif
  trailing_done();                                   // the last step's MFMAs have retired before anything below reuses their operand registers
  // ---- flush ---------------------------------------------------------------------------------
  // (lane geometry recomputed from the lane id: nothing of it has to stay in a register across the scan)
  const unsigned lf = fresh_lane();
  const int fi = (int)(lf & 15u), fg = (int)(lf >> 4);
  const int fb = blockIdx.x * 16 + fi, fn0 = wv * UPW + fg * UPL;
  if (!RAGGED || fb < B) {
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) st4(d_h0 + (size_t)fb * H + fn0 + 4 * mt, dh[mt]);
  }
  float* const ps = part + (size_t)blockIdx.x * SLAB_LR;
  // D row 4g + r of tile a = own unit 8g + 4a + r (see the parked U1^T fragments); column i = rank index j
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wv * UPW + 16 * a + 4 * fg + r;
      ps[SL_UW2 + n * 32 + fi] = accUW[a][0][r];
      ps[SL_UW2 + n * 32 + 16 + fi] = accUW[a][1][r];
      ps[SL_U1T + n * 16 + fi] = accU1[a][r];
    }
  if (xwave) {
#pragma unroll
    for (int r = 0; r < 4; ++r) ps[SL_W1T + ((wv & 1) * 16 + 4 * fg + r) * 16 + fi] = accW1[r];
  }
#pragma unroll
  for (int mt = 0; mt < NT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = sbz[mt][r], c = sbh[mt][r];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); c += __shfl_xor(c, m); }
      if (fi == 0) {
        ps[SL_BZ + fn0 + 4 * mt + r] = a;
        ps[SL_BZ + H + fn0 + 4 * mt + r] = c;
      }
    }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { pz += __shfl_xor(pz, m); pn += __shfl_xor(pn, m); }
  if ((lf & 63u) == 0) { red[wv] = pz; red[NW + wv] = pn; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int w2i = 0; w2i < NW; ++w2i) { a += red[w2i]; c += red[NW + w2i]; }
    ps[SL_ZN] = a; ps[SL_ZN + 1] = c;
  }
}

// All gradients of the low-rank backward that are sums over workgroups: fixed-order sum of the slabs, written in the
// operator's shapes (d_u2 [H,ru], d_w2 [H,rw], d_u1 [ru,H], d_w1 [rw,F]; rank padding dropped), raw-zeta / raw-nu
// chain rule applied (.cu:116-117,544-545).
__global__ __launch_bounds__(1024) void reduce_lowrank_slabs(int nwg, const float* __restrict__ part,
                                                             const float* __restrict__ zeta, const float* __restrict__ nu,
                                                             int ru, int rw, float* __restrict__ d_u1,
                                                             float* __restrict__ d_u2, float* __restrict__ d_w1,
                                                             float* __restrict__ d_w2, float* __restrict__ d_bz,
                                                             float* __restrict__ d_bh, float* __restrict__ d_zeta,
                                                             float* __restrict__ d_nu) {
  __shared__ float sm[16][64];
  const int o = threadIdx.x & 63, pid = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;               // 0 .. SL_ZN + 1
  float a = 0.f;
  if (idx < SL_ZN + 2) {
    for (int wg0 = pid; wg0 < nwg; wg0 += 64) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int wg = wg0 + 16 * j; v[j] = wg < nwg ? part[(size_t)wg * SLAB_LR + idx] : 0.f; }
      a += (v[0] + v[1]) + (v[2] + v[3]);
    }
  }
  sm[pid][o] = a;
  __syncthreads();
  if (pid == 0 && idx < SL_ZN + 2) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += sm[j][o];
    if (idx < SL_U1T) {
      const int n = idx >> 5, j = idx & 31;
      if (j < 16) { if (j < ru) d_u2[n * ru + j] = t; } else if (j - 16 < rw) d_w2[n * rw + (j - 16)] = t;
    } else if (idx < SL_W1T) {
      const int k = idx - SL_U1T, n = k >> 4, j = k & 15;
      if (j < ru) d_u1[j * 256 + n] = t;
    } else if (idx < SL_BZ) {
      const int k = idx - SL_W1T, f = k >> 4, j = k & 15;
      if (j < rw) d_w1[j * 32 + f] = t;
    } else if (idx < SL_BZ + 256) d_bz[idx - SL_BZ] = t;
    else if (idx < SL_ZN) d_bh[idx - SL_BZ - 256] = t;
    else if (idx == SL_ZN) { const float sz = 1.0f / (1.0f + expf(-zeta[0])); d_zeta[0] = t * sz * (1.0f - sz); }   // .cu:116,544
    else { const float sn = 1.0f / (1.0f + expf(-nu[0])); d_nu[0] = t * sn * (1.0f - sn); }                         // .cu:117,545
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

struct LowrankBwdWs { size_t part, sink, xt, dxt, total; };
LowrankBwdWs lowrank_bwd_layout(const fastgrnn_desc& d) {
  const size_t TB = (size_t)d.T * d.B, nwg = (d.B + 15) / 16;
  LowrankBwdWs L; size_t o = 0;
  L.part = o; o += align256(nwg * SLAB_LR * 4);      // one slab of partial sums per workgroup
  L.sink = o; o += align256(16 * 32 * 4);            // d_x rows of the lanes beyond a ragged batch
  L.xt = L.dxt = o;
  if (d.flags & FASTGRNN_FLAG_X_BFT) {               // time-major copies of x and d_x
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
  float* part = (float*)(base + L.part);
  float* sink = (float*)(base + L.sink);
  const int nwg = (d.B + 15) / 16;
  const bool bf = d.dtype == FASTGRNN_BF16_IO, bft = (d.flags & FASTGRNN_FLAG_X_BFT) != 0;
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
                       (d.flags & FASTGRNN_FLAG_GRAD_LAST) ? 1 : 0, (const float*)ghs, (const float*)xs, (const float*)hs,
                       (const float*)pre_s, (const float*)m_s, (const float*)h0, (const float*)p.w1, (const float*)p.w2,
                       (const float*)p.u1, (const float*)p.u2, (const float*)p.bias_gate, (const float*)p.bias_update,
                       (const float*)p.zeta, (const float*)p.nu, (float*)dxs, (float*)g.d_h0, sink, part);
  };
  // 8 waves (two per SIMD) for full and ragged batches alike (lanes beyond a ragged batch only get a zero gradient,
  // which needs no extra registers; the first ragged variant masked five values per element and spilled)
  if (bf) { if (d.B % 16) go(bwd_scan_lowrank_split<GATE, true, true>); else go(bwd_scan_lowrank_split<GATE, false, true>); }
  else    { if (d.B % 16) go(bwd_scan_lowrank_split<GATE, true, false>); else go(bwd_scan_lowrank_split<GATE, false, false>); }
  if (bft) {
    if (bf) hipLaunchKernelGGL((bft_transpose<unsigned short, false>), dim3(d.B), dim3(256), 0, s, d.B, d.T,
                               (const unsigned short*)(base + L.dxt), (unsigned short*)g.d_x);
    else hipLaunchKernelGGL((bft_transpose<float, false>), dim3(d.B), dim3(256), 0, s, d.B, d.T,
                            (const float*)(base + L.dxt), (float*)g.d_x);
  }
  // every parameter gradient is a sum over the workgroups' slabs (.cu:544-555, factorised)
  hipLaunchKernelGGL(reduce_lowrank_slabs, dim3((SL_ZN + 2 + 63) / 64), dim3(1024), 0, s, nwg, part, (const float*)p.zeta,
                     (const float*)p.nu, d.u_rank, d.w_rank, (float*)g.d_u1, (float*)g.d_u2, (float*)g.d_w1,
                     (float*)g.d_w2, (float*)g.d_bias_gate, (float*)g.d_bias_update, (float*)g.d_zeta, (float*)g.d_nu);
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
  const bool preact = (d.flags & FASTGRNN_FLAG_SAVE_PREACT) != 0;
  const int aux = zs == nullptr ? 0 : (preact ? 2 : 1);
  auto pick = [&](auto bf_tag) __attribute__((always_inline)) {
    constexpr bool BFv = decltype(bf_tag)::value;
    if (d.flags & FASTGRNN_FLAG_HS_LAST) {           // inference: aux == 0 (lowrank_forward)
      if (ragged) go(fwd_scan_lowrank_split<GATE, 0, true, BFv, true>); else go(fwd_scan_lowrank_split<GATE, 0, false, BFv, true>);
    } else if (aux == 2) {
      if (ragged) go(fwd_scan_lowrank_split<GATE, 2, true, BFv>); else go(fwd_scan_lowrank_split<GATE, 2, false, BFv>);
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
  if ((double)d.T * d.B * 256 * 4.0 >= 4294967296.0) return false;
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
  if (preact && (!cs || !zs) && !(d.flags & FASTGRNN_FLAG_HS_LAST)) return FASTGRNN_ERR_NULL_POINTER;   // pre-activation and rank-space vector
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
  if (!cs || !zs) return FASTGRNN_ERR_NULL_POINTER;  // the rank-space vector and the pre-activation saved by the forward
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
