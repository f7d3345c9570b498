// MFMA-tiled fp32 FastGRNN scan for gfx950 (dense W/U, tanh update, sigmoid/relu/tanh gate).
//
// Geometry.  One workgroup = 4 waves = one tile of 16 utterances for ALL T frames.  The
// recurrent product is evaluated transposed, D[n][b] = sum_k U[n][k] h[b][k], with
// v_mfma_f32_16x16x4_f32 (exact fp32 fma chain): the 16 utterances sit on the MFMA's N
// (lane&15) axis, hidden units on M.  Wave w owns the H/4 hidden units
//     n_own(g,mt,r) = w*H/4 + g*4*MT + mt*4 + r        (g = lane>>4, MT = H/64 tiles, r = reg)
// so a lane's 4*MT results are CONTIGUOUS in memory (float4 stores of hs/z/c; four
// lane groups cover one 128-B line per utterance).  The A operands (slices of U, W, or
// of their transposes in the backward) are loaded ONCE into VGPRs and stay there for
// the whole scan: U is never re-read from LDS or HBM.  Per step only the 16 x H state
// tile crosses waves, through an 8 KB LDS image [k/4][b] of float4 that is written with
// ds_write_b128 and read back, conflict-free, as the next step's B operand.
//
// The K index of every product is permuted (k = 16q + 4g + r for MFMA step 4q+r) so that
// the B operand of a lane is 4 contiguous floats per q; A is loaded with the same
// permutation, which leaves the sum unchanged.
//
// Backward additionally keeps the per-workgroup dW/dU partial sums in accumulator VGPRs
// across all T steps (d_pre^T . [x | h_prev] with K = the 16 utterances) and flushes them
// once; a tiny second kernel reduces the per-workgroup slabs deterministically.
//
// Reference semantics: forward .cu:42-60 + .cu:367-413; backward .cu:91-119 + .cu:473-545.
#include "common.h"

namespace fastgrnn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float LOG2E = 1.4426950408889634f;

// v_exp_f32 / v_rcp_f32 are 1-ulp; absolute error of these forms is < 3e-7 (the parity
// bar is 1e-5).  Saturation is exact: exp2(+inf) -> rcp(inf) = 0.
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

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int H, int F, int GATE>
__global__ __launch_bounds__(256) void fwd_scan_mfma(
    int Tn, int B, const float* __restrict__ x, const float* __restrict__ h0,
    const float* __restrict__ w, const float* __restrict__ u,
    const float* __restrict__ bz, const float* __restrict__ bh,
    const float* __restrict__ zeta, const float* __restrict__ nu,
    float* __restrict__ hs, float* __restrict__ zs, float* __restrict__ cs) {
  constexpr int HS = H / 4;        // hidden units per wave
  constexpr int MT = HS / 16;      // 16-row MFMA tiles per wave
  constexpr int KH = H / 4;        // MFMA steps over the hidden dim
  constexpr int KX = F / 4;        // MFMA steps over the feature dim
  constexpr int NQ = H / 16;
  static_assert(H % 64 == 0 && F % 16 == 0, "tile shape");
  __shared__ f32x4 hl[2][(H / 4) * 16];   // state tile, [k/4][b] float4, double-buffered

  const int tid = threadIdx.x;
  const int wv = tid >> 6, l = tid & 63, i = l & 15, g = l >> 4;
  const int b = blockIdx.x * 16 + i;
  const bool valid = b < B;
  const int bc = valid ? b : B - 1;                 // clamped row for loads
  const int n0 = wv * HS + g * (4 * MT);            // first of this lane's 4*MT hidden units

  // ---- resident A operands -----------------------------------------------------------------
  float Uf[MT][KH], Wf[MT][KX];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int nA = wv * HS + (i >> 2) * (4 * MT) + mt * 4 + (i & 3);   // A row i of tile mt
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4 v = ld4(u + (size_t)nA * H + 16 * q + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) Uf[mt][4 * q + r] = v[r];
    }
#pragma unroll
    for (int kq = 0; kq < KX / 4; ++kq) {
      f32x4 v = ld4(w + (size_t)nA * F + g * KX + 4 * kq);
#pragma unroll
      for (int r = 0; r < 4; ++r) Wf[mt][4 * kq + r] = v[r];
    }
  }
  f32x4 bzv[MT], bhv[MT], hown[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    bzv[mt] = ld4(bz + n0 + 4 * mt);
    bhv[mt] = ld4(bh + n0 + 4 * mt);
    hown[mt] = ld4(h0 + (size_t)bc * H + n0 + 4 * mt);
    hl[0][(wv * (HS / 4) + g * MT + mt) * 16 + i] = hown[mt];
  }
  const float sz = fsigmoid(zeta[0]), sn = fsigmoid(nu[0]);

  // x fragment of frame 0, and W.x_0
  float xB[KX];
  {
    const float* xp = x + (size_t)bc * F + g * KX;
#pragma unroll
    for (int kq = 0; kq < KX / 4; ++kq) {
      f32x4 v = ld4(xp + 4 * kq);
#pragma unroll
      for (int r = 0; r < 4; ++r) xB[4 * kq + r] = v[r];
    }
  }
  f32x4 accx[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) accx[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < KX; ++kk)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) accx[mt] = mfma4(Wf[mt][kk], xB[kk], accx[mt]);
  // features of frame 1, carried across the loop back-edge: frame t+2 is requested at the
  // top of step t and first read at the bottom of step t+1, one whole step later
  f32x4 xv[KX / 4];
  {
    const float* xp = x + ((size_t)(Tn > 1 ? 1 : 0) * B + bc) * F + g * KX;
#pragma unroll
    for (int kq = 0; kq < KX / 4; ++kq) xv[kq] = ld4(xp + 4 * kq);
  }
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < Tn; ++t) {
    const int tn = (t + 2 < Tn) ? t + 2 : Tn - 1;      // clamped: the tail loads are unused
    f32x4 xn[KX / 4];
    {
      const float* xp = x + ((size_t)tn * B + bc) * F + g * KX;
#pragma unroll
      for (int kq = 0; kq < KX / 4; ++kq) xn[kq] = ld4(xp + 4 * kq);
    }
    __builtin_amdgcn_sched_barrier(0);
    // B operand: the whole state tile h_{t-1}
    float hB[KH];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4 v = hl[cur][(4 * q + g) * 16 + i];
#pragma unroll
      for (int r = 0; r < 4; ++r) hB[4 * q + r] = v[r];
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = accx[mt];
#pragma unroll
    for (int kk = 0; kk < KH; ++kk)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma4(Uf[mt][kk], hB[kk], acc[mt]);   // .cu:368
    // W.x_{t+1}: independent of h_t, issued behind the recurrent chain so that the
    // epilogue's VALU work below overlaps it
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) accx[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KX; ++kk)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) accx[mt] = mfma4(Wf[mt][kk], xv[kk >> 2][kk & 3], accx[mt]);
    // epilogue (.cu:55-58)
    f32x4 zv[MT], cv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = acc[mt][r];
        const float z = gate_act<GATE>(pre + bzv[mt][r]);
        const float c = ftanh(pre + bhv[mt][r]);
        hown[mt][r] = (sz * (1.0f - z) + sn) * c + hown[mt][r] * z;
        zv[mt][r] = z; cv[mt][r] = c;
      }
      hl[cur ^ 1][(wv * (HS / 4) + g * MT + mt) * 16 + i] = hown[mt];
    }
    if (valid) {
      const size_t o = ((size_t)t * B + b) * H + n0;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) st4(hs + o + 4 * mt, hown[mt]);
      if (zs) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { st4(zs + o + 4 * mt, zv[mt]); st4(cs + o + 4 * mt, cv[mt]); }
      }
    }
#pragma unroll
    for (int kq = 0; kq < KX / 4; ++kq) xv[kq] = xn[kq];
    __syncthreads();
    cur ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
// floats per workgroup slab of backward partial sums, padded to 64
__host__ __device__ constexpr int slab_stride(int H, int F) { return (H * H + H * F + 2 * H + 2 + 63) & ~63; }

template <int H, int F>
struct BwdLds {
  static constexpr int HS = H / 4;
  f32x4 P[2][(H / 4) * 16];          // d_pre tile, [n/4][b] float4   (B operand of d_h, d_x)
  float Tt[2][4][HS][20];            // d_pre^T per wave, [n_local][b] (A operand of dW, dU)
  float Hp[2][16][H + 4];            // h_prev tile, [b][k]            (B operand of dU)
  f32x4 DX[2][3][64];                // d_x partial sums from the waves that split K
  float red[8];
};

template <int H, int F, int GATE>
__global__ __launch_bounds__(256) void bwd_scan_mfma(
    int Tn, int B, const float* __restrict__ ghs, const float* __restrict__ x,
    const float* __restrict__ hs, const float* __restrict__ zs, const float* __restrict__ cs,
    const float* __restrict__ h0, const float* __restrict__ w, const float* __restrict__ u,
    const float* __restrict__ zeta, const float* __restrict__ nu,
    float* __restrict__ d_x, float* __restrict__ d_h0,
    float* __restrict__ part)      // [nwg][slab_stride(H,F)]: dU | dW | d_bz | d_bh | (d_zeta, d_nu) sums
{
  constexpr int HS = H / 4, MT = HS / 16, KH = H / 4, NQ = H / 16;
  constexpr int NCT = H / 16;      // column tiles of dU
  constexpr int NFT = F / 16;      // feature tiles (d_x rows, dW columns)
  constexpr int NSPLIT = 4 / NFT;  // waves that split the K (=n) range of one d_x tile
  constexpr int KD = KH / NSPLIT;  // MFMA steps of d_x per wave
  static_assert(NFT == 1 || NFT == 2 || NFT == 4, "F must be 16, 32 or 64");
  __shared__ BwdLds<H, F> S;

  const int tid = threadIdx.x;
  const int wv = tid >> 6, l = tid & 63, i = l & 15, g = l >> 4;
  const int b = blockIdx.x * 16 + i;
  const bool valid = b < B;
  const int bc = valid ? b : B - 1;
  const int n0 = wv * HS + g * (4 * MT);
  const int ft = wv % NFT, nh = wv / NFT;           // d_x role of this wave
  const float sz = fsigmoid(zeta[0]), sn = fsigmoid(nu[0]);

  // ---- resident A operands -----------------------------------------------------------------
  // d_h[k][b] = sum_n U[n][k] d_pre[b][n]:  A row i of tile mt is k = wv*HS + (i>>2)*4MT + mt*4 + (i&3)
  float UTf[MT][KH];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int kA = wv * HS + (i >> 2) * (4 * MT) + mt * 4 + (i & 3);
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
      const int n = 16 * (kk >> 2) + 4 * g + (kk & 3);
      UTf[mt][kk] = u[(size_t)n * H + kA];
    }
  }
  // d_x[f][b] = sum_n W[n][f] d_pre[b][n]:  tile ft, K range [nh*KD, nh*KD+KD)
  float WTf[KD];
#pragma unroll
  for (int kk = 0; kk < KD; ++kk) {
    const int k2 = nh * KD + kk;
    const int n = 16 * (k2 >> 2) + 4 * g + (k2 & 3);
    WTf[kk] = w[(size_t)n * F + ft * 16 + i];
  }

  f32x4 accU[MT][NCT], accW[MT][NFT];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) accU[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NFT; ++c) accW[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 sbz[MT], sbh[MT], dh[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    sbz[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; sbh[mt] = sbz[mt]; dh[mt] = sbz[mt];
  }
  float pz = 0.f, pn = 0.f;
  f32x4 dx_keep = f32x4{0.f, 0.f, 0.f, 0.f};

  // Operands of step t are requested during step t+1 (loop-carried registers), so a whole
  // step of MFMA work covers the HBM latency.
  f32x4 gv[MT], zv[MT], cv[MT], hp[MT];
  float xT[NFT][4];
  auto load_step = [&](int t, f32x4 (&gq)[MT], f32x4 (&zq)[MT], f32x4 (&cq)[MT], f32x4 (&hq)[MT],
                       float (&xq)[NFT][4]) {
    const size_t o = ((size_t)t * B + bc) * H + n0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      gq[mt] = ld4(ghs + o + 4 * mt);
      zq[mt] = ld4(zs + o + 4 * mt);
      cq[mt] = ld4(cs + o + 4 * mt);
      hq[mt] = (t == 0) ? ld4(h0 + (size_t)bc * H + n0 + 4 * mt) : ld4(hs + o - (size_t)B * H + 4 * mt);  // .cu:478-481
    }
    // x_t^T fragments for dW: B[k = utterance 4g+kk][j = feature f2*16 + i]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int bb = blockIdx.x * 16 + 4 * g + kk;
      const bool ok = bb < B;
      const float* xp = x + ((size_t)t * B + (ok ? bb : B - 1)) * F + i;
#pragma unroll
      for (int f2 = 0; f2 < NFT; ++f2) { float v = xp[16 * f2]; xq[f2][kk] = ok ? v : 0.0f; }
    }
  };
  load_step(Tn - 1, gv, zv, cv, hp, xT);

  for (int t = Tn - 1; t >= 0; --t) {
    const int buf = t & 1;
    f32x4 gn[MT], zn[MT], cn[MT], hn[MT];
    float xn[NFT][4];
    load_step(t > 0 ? t - 1 : 0, gn, zn, cn, hn, xn);
    __builtin_amdgcn_sched_barrier(0);
    // ---- elementwise (.cu:107-117) ---------------------------------------------------------
    f32x4 dp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gg = gv[mt][r] + dh[mt][r];                                  // .cu:474
        const float z = zv[mt][r], c = cv[mt][r];
        float dcp = (sz * (1.0f - z) + sn) * (1.0f - c * c) * gg;                // .cu:109
        float dzp = (hp[mt][r] - sz * c) * gate_dact<GATE>(z) * gg;              // .cu:110
        float zg = z * gg;                                                        // .cu:108
        float tz = (1.0f - z) * c * gg, tn = c * gg;                              // .cu:114-115
        if (!valid) { dcp = 0.f; dzp = 0.f; zg = 0.f; tz = 0.f; tn = 0.f; }
        sbz[mt][r] += dzp; sbh[mt][r] += dcp; pz += tz; pn += tn;
        dp[mt][r] = dzp + dcp;                                                    // .cu:113
        dh[mt][r] = zg;
        if (!valid) hp[mt][r] = 0.f;
      }
      S.P[buf][(wv * (HS / 4) + g * MT + mt) * 16 + i] = dp[mt];
      *reinterpret_cast<f32x4*>(&S.Hp[buf][i][n0 + 4 * mt]) = hp[mt];
#pragma unroll
      for (int r = 0; r < 4; ++r) S.Tt[buf][wv][g * (4 * MT) + mt * 4 + r][i] = dp[mt][r];
    }
    __syncthreads();
    // ---- finish d_x of the PREVIOUS step (its partial sums were published before this barrier)
    if (NSPLIT > 1 && nh == 0 && t + 1 < Tn) {
      f32x4 s = dx_keep;
#pragma unroll
      for (int k = 1; k < NSPLIT; ++k) s += S.DX[buf ^ 1][(k - 1) * NFT + ft][l];
      if (valid) st4(d_x + ((size_t)(t + 1) * B + b) * F + ft * 16 + 4 * g, s);
    }
    // ---- operands from LDS ------------------------------------------------------------------
    float dpB[KH];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4 v = S.P[buf][(4 * q + g) * 16 + i];
#pragma unroll
      for (int r = 0; r < 4; ++r) dpB[4 * q + r] = v[r];
    }
    f32x4 dpT[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) dpT[a] = *reinterpret_cast<const f32x4*>(&S.Tt[buf][wv][a * 16 + i][4 * g]);
    // ---- d_h chain (.cu:537): C-in = z*g --------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < KH; ++kk)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) dh[mt] = mfma4(UTf[mt][kk], dpB[kk], dh[mt]);
    // ---- d_x partial (.cu:538) -------------------------------------------------------------
    f32x4 accx = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KD; ++kk) accx = mfma4(WTf[kk], dpB[nh * KD + kk], accx);
    // ---- dW (.cu:539), dU (.cu:540): K = the 16 utterances ---------------------------------
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int f2 = 0; f2 < NFT; ++f2) accW[a][f2] = mfma4(dpT[a][kk], xT[f2][kk], accW[a][f2]);
    }
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float hv = S.Hp[buf][4 * g + kk][c * 16 + i];
#pragma unroll
        for (int a = 0; a < MT; ++a) accU[a][c] = mfma4(dpT[a][kk], hv, accU[a][c]);
      }
    }
    // publish / keep the d_x partial of this step
    if (NSPLIT > 1) {
      if (nh == 0) dx_keep = accx;
      else S.DX[buf][(nh - 1) * NFT + ft][l] = accx;
    } else if (valid) {
      st4(d_x + ((size_t)t * B + b) * F + ft * 16 + 4 * g, accx);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { gv[mt] = gn[mt]; zv[mt] = zn[mt]; cv[mt] = cn[mt]; hp[mt] = hn[mt]; }
#pragma unroll
    for (int f2 = 0; f2 < NFT; ++f2)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) xT[f2][kk] = xn[f2][kk];
  }
  __syncthreads();
  if (NSPLIT > 1 && nh == 0) {       // d_x of t = 0
    f32x4 s = dx_keep;
#pragma unroll
    for (int k = 1; k < NSPLIT; ++k) s += S.DX[0][(k - 1) * NFT + ft][l];
    if (valid) st4(d_x + (size_t)b * F + ft * 16 + 4 * g, s);
  }
  // ---- flush ---------------------------------------------------------------------------------
  if (valid) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) st4(d_h0 + (size_t)b * H + n0 + 4 * mt, dh[mt]);
  }
  // dU / dW slabs: D row 4g+r of tile a is n = wv*HS + a*16 + 4g + r; column = c*16 + i
  {
    float* pu = part + (size_t)blockIdx.x * slab_stride(H, F);
    float* pw = pu + H * H;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = wv * HS + a * 16 + 4 * g + r;
#pragma unroll
        for (int c = 0; c < NCT; ++c) pu[(size_t)n * H + c * 16 + i] = accU[a][c][r];
#pragma unroll
        for (int f2 = 0; f2 < NFT; ++f2) pw[(size_t)n * F + f2 * 16 + i] = accW[a][f2][r];
      }
  }
  // bias partials: sum over the 16 utterance lanes of each group
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = sbz[mt][r], c = sbh[mt][r];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); c += __shfl_xor(c, m); }
      if (i == 0) {
        float* pb = part + (size_t)blockIdx.x * slab_stride(H, F) + H * H + H * F;
        pb[n0 + 4 * mt + r] = a;
        pb[H + n0 + 4 * mt + r] = c;
      }
    }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { pz += __shfl_xor(pz, m); pn += __shfl_xor(pn, m); }
  if (l == 0) { S.red[wv] = pz; S.red[4 + wv] = pn; }
  __syncthreads();
  if (tid == 0) {
    float* pzn = part + (size_t)blockIdx.x * slab_stride(H, F) + H * H + H * F + 2 * H;
    pzn[0] = S.red[0] + S.red[1] + S.red[2] + S.red[3];
    pzn[1] = S.red[4] + S.red[5] + S.red[6] + S.red[7];
  }
}

// Deterministic reduction of the per-workgroup slabs in ONE launch: out[i] = sum_wg part[wg][i]
// for the whole gradient vector dU | dW | d_bz | d_bh | d_zeta | d_nu (.cu:542-545).  A block
// owns 64 consecutive outputs; its 1024 threads split the workgroups 16 ways (all loads of a
// thread in flight at once), then 64 threads add the 16 partials in a fixed order.
__global__ __launch_bounds__(1024) void reduce_slabs(int nwg, int H, int F, int stride,
                                                     const float* __restrict__ part,
                                                     const float* __restrict__ zeta, const float* __restrict__ nu,
                                                     float* __restrict__ d_u, float* __restrict__ d_w,
                                                     float* __restrict__ d_bz, float* __restrict__ d_bh,
                                                     float* __restrict__ d_zeta, float* __restrict__ d_nu) {
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
        v[j] = wg < nwg ? part[(size_t)wg * stride + idx] : 0.f;
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
    else if (idx < oBz) d_w[idx - oW] = t;
    else if (idx < oBh) d_bz[idx - oBz] = t;
    else if (idx < oZ) d_bh[idx - oBh] = t;
    else if (idx == oZ) { const float sz = 1.0f / (1.0f + expf(-zeta[0])); d_zeta[0] = t * sz * (1.0f - sz); }   // .cu:116,544
    else { const float sn = 1.0f / (1.0f + expf(-nu[0])); d_nu[0] = t * sn * (1.0f - sn); }                      // .cu:117,545
  }
}

size_t bwd_ws_bytes(const fastgrnn_desc& d) {
  size_t nwg = (d.B + 15) / 16;
  return align256(nwg * (size_t)slab_stride(d.H, d.F) * 4);
}

template <int H, int F>
int launch_fwd(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs, void* zs,
               void* cs, hipStream_t s) {
  dim3 grid((d.B + 15) / 16), block(256);
  auto args = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, block, 0, s, d.T, d.B, (const float*)x, (const float*)h0, (const float*)p.w,
                       (const float*)p.u, (const float*)p.bias_gate, (const float*)p.bias_update,
                       (const float*)p.zeta, (const float*)p.nu, (float*)hs, (float*)zs, (float*)cs);
  };
  switch (d.gate_nl) {
    case FASTGRNN_NL_SIGMOID: args(fwd_scan_mfma<H, F, FASTGRNN_NL_SIGMOID>); break;
    case FASTGRNN_NL_RELU: args(fwd_scan_mfma<H, F, FASTGRNN_NL_RELU>); break;
    default: args(fwd_scan_mfma<H, F, FASTGRNN_NL_TANH>); break;
  }
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

template <int H, int F>
int launch_bwd(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x, const void* hs,
               const void* zs, const void* cs, const void* h0, const fastgrnn_grads& g, void* ws, hipStream_t s) {
  float* part = reinterpret_cast<float*>(ws);
  const int nwg = (d.B + 15) / 16;
  dim3 grid(nwg), block(256);
  auto args = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, block, 0, s, d.T, d.B, (const float*)ghs, (const float*)x, (const float*)hs,
                       (const float*)zs, (const float*)cs, (const float*)h0, (const float*)p.w, (const float*)p.u,
                       (const float*)p.zeta, (const float*)p.nu, (float*)g.d_x, (float*)g.d_h0, part);
  };
  switch (d.gate_nl) {
    case FASTGRNN_NL_SIGMOID: args(bwd_scan_mfma<H, F, FASTGRNN_NL_SIGMOID>); break;
    case FASTGRNN_NL_RELU: args(bwd_scan_mfma<H, F, FASTGRNN_NL_RELU>); break;
    default: args(bwd_scan_mfma<H, F, FASTGRNN_NL_TANH>); break;
  }
  const int ntot = H * H + H * F + 2 * H + 2;
  hipLaunchKernelGGL(reduce_slabs, dim3((ntot + 63) / 64), dim3(1024), 0, s, nwg, H, F, slab_stride(H, F), part,
                     (const float*)p.zeta, (const float*)p.nu, (float*)g.d_u, (float*)g.d_w,
                     (float*)g.d_bias_gate, (float*)g.d_bias_update, (float*)g.d_zeta, (float*)g.d_nu);
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

bool shape_ok(int H, int F) { return (H == 128 && F == 32) || (H == 64 && F == 32) || (H == 128 && F == 64); }

}  // namespace

bool mfma_supported(const fastgrnn_desc& d, int /*direction*/) {
  return d.dtype == FASTGRNN_F32 && d.w_rank == 0 && d.u_rank == 0 && d.update_nl == FASTGRNN_NL_TANH &&
         d.gate_nl >= FASTGRNN_NL_SIGMOID && d.gate_nl <= FASTGRNN_NL_TANH && shape_ok(d.H, d.F);
}
size_t mfma_forward_ws(const fastgrnn_desc&) { return 0; }
size_t mfma_backward_ws(const fastgrnn_desc& d) { return bwd_ws_bytes(d); }

int mfma_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs,
                 void* zs, void* cs, void*, hipStream_t s) {
  if ((zs == nullptr) != (cs == nullptr)) return FASTGRNN_ERR_NULL_POINTER;
  if (d.H == 128 && d.F == 32) return launch_fwd<128, 32>(d, p, x, h0, hs, zs, cs, s);
  if (d.H == 64 && d.F == 32) return launch_fwd<64, 32>(d, p, x, h0, hs, zs, cs, s);
  if (d.H == 128 && d.F == 64) return launch_fwd<128, 64>(d, p, x, h0, hs, zs, cs, s);
  return FASTGRNN_ERR_UNSUPPORTED;
}

int mfma_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x, const void* hs,
                  const void* zs, const void* cs, const void* h0, const fastgrnn_grads& g, void* ws,
                  hipStream_t s) {
  if (d.H == 128 && d.F == 32) return launch_bwd<128, 32>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s);
  if (d.H == 64 && d.F == 32) return launch_bwd<64, 32>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s);
  if (d.H == 128 && d.F == 64) return launch_bwd<128, 64>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s);
  return FASTGRNN_ERR_UNSUPPORTED;
}

}  // namespace fastgrnn
