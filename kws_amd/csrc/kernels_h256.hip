// Split-precision FastGRNN scans for the dense H = 256, F = 32 layer (gfx950): the first layer of the reference's
// default stack (trainingConfig.py:12-15: 32 -> 256 -> 128; model.py:196-203).
//
// Same arithmetic, geometry and rules as the H = 128 scans of kernels_split.hip -- fp32 operands as exact bf16 /
// fp16 planes on v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulation, one workgroup of 8 waves = 16 utterances for all
// T, the operand rule of DESIGN.md 4.0 -- but U is 4x larger: 256 KB as fp32, more than the register file of a CU
// can hold as three bf16 planes next to anything else.  What is resident where:
//   forward, bounded state (gate with z in [0,1] and a bounded h0, checked on the device):  U as TWO fp16 planes in
//       registers (128 per lane), the state product costs 3 MFMAs per K-step;
//   forward otherwise:  bf16 planes 0 and 1 of U in registers, plane 2 in LDS in fragment order (128 KB: each wave
//       re-reads only its own fragments, 16 ds_read_b128 per step, so the LDS acts as a second register file); six
//       MFMA terms per K-step as everywhere else;
//   backward:  U^T as two fp16 planes, hi in registers (64 per lane), lo in LDS; d_pre as two fp16 planes scaled by
//       an exact power of two per utterance and step -- see bwd_scan_h256.
// Wave w owns hidden units 32w..32w+31 (two 16-row tiles, eight K-steps).  W (32 wide) stays fused in the forward as
// in the H = 128 kernel.  The backward keeps the recurrence only and writes d_pre[T,B,H]: dU, dW and d_x are batched
// GEMMs afterwards (kernels_gemm.hip; .cu:538-540 does them per step) -- neither U^T's planes nor the 256 KB of dU
// accumulators would fit on chip beside each other.
// Reference semantics: forward .cu:42-60 + .cu:367-413; backward .cu:91-119 + .cu:473-545.
#include "split_common.h"

namespace fastgrnn {
namespace {

constexpr int H2 = 256, F2 = 32, KS2 = 8;
constexpr int ROWH2 = 528;                      // bytes per utterance row of a 256-wide 16-bit plane (512 + 16)
constexpr int ROWX2 = 80;                       // ... of a 32-wide plane (64 + 16)
constexpr int PLH2 = 16 * ROWH2;                // 8448: one state plane
constexpr int PLX2 = 16 * ROWX2;                // 1280
constexpr int U2L = 8 * 2 * KS2 * 64 * 16;      // 131072: plane 2 of every wave's weight fragments
constexpr int SLAB2 = 576;                      // floats per workgroup: d_bz[256] | d_bh[256] | (zeta, nu) sums, padded

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// AUX: 0 = hs only; 1 = also z_s, h_prime_s (the reference operator's outputs); 2 = also the pre-activation
// W.x + U.h into zs (FASTGRNN_FLAG_SAVE_PREACT); 3 = hs is [B,H] and receives h_T alone (FASTGRNN_FLAG_HS_LAST).
// MODE 0: bf16 path, unconditionally.  MODE 1: fp16 path for the workgroups whose rows of h0 allow it; the others
// set their word of `flags` and leave at once.  MODE 2: bf16 path for exactly those workgroups (launched right behind
// MODE 1 on the same stream; a workgroup whose flag is clear leaves at once).  Two kernels instead of one with both
// paths inside: together they needed a dozen registers more than a wave has, i.e. spill reloads inside the loop.
// PREIN: the layer's input is wider than 32 (F = 64: the reference's default feature type, mfcc + delta,
// trainingConfig.py:36, mfccProcessor.py:27-28; F = 128) -- W no longer fits beside U, and its product with the frames
// has no recurrence in it: `x` is then P[T*B, 256] = X . W^T, written by the batched frame GEMM (kernels_gemm.hip) in
// front of this launch, and the scan reads P(t) where it otherwise runs its W.x MFMAs (as the H = 128 wide layers do).
// BF: FASTGRNN_BF16_IO -- x and hs are bf16 in HBM (P, the saved pre-activation, h0 and the state itself stay fp32).
template <int GATE, int AUX, bool RAGGED, int MODE, bool PREIN = false, bool BF = false>
__global__ __launch_bounds__(512) void fwd_scan_h256(
    int Tn, int B, unsigned hsT, unsigned hsB, unsigned xsT, unsigned xsB,
    const float* __restrict__ x, const float* __restrict__ h0,
    const float* __restrict__ w, const float* __restrict__ u,
    const float* __restrict__ bz, const float* __restrict__ bh,
    const float* __restrict__ zeta, const float* __restrict__ nu,
    float* __restrict__ hs, float* __restrict__ zs, float* __restrict__ cs, unsigned* __restrict__ flags) {
  // hsT / hsB: element strides of one step / one utterance in hs, zs, cs (and in P under PREIN, whose rows follow the
  // frames' order); xsT / xsB: the same for x.  Time-major: (B*256, 256) and (B*32, 32); FASTGRNN_FLAG_BATCH_MAJOR:
  // (256, T*256) and (32, T*32) -- rnn.py:812-813 transposes instead.
  constexpr bool F16H = MODE == 1;
  if (MODE == 2 && flags[blockIdx.x] == 0u) return;  // (whole workgroup; no barrier has been passed)
  // One byte array carved per path (static LDS is the maximum over both):
  //   fp16 path:  state planes [2 buffers][2 planes] | feature planes [2][3] | W planes in fragment order (48 KB: with
  //               128 registers of U planes per lane the 24 of W's are re-read each step instead of kept)
  //   bf16 path:  U plane 2 (fragment order) | state planes [1][3] | feature planes [1][3]   (single-buffered: two
  //               barriers per step; 160 256 bytes, the CU has 163 840)
  __shared__ __attribute__((aligned(16))) unsigned char smem[U2L + 3 * PLH2 + 3 * PLX2];
  __shared__ __attribute__((aligned(16))) float sbias[2][H2];
  __shared__ float hmax_s[8];

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, i = l & 15, g = l >> 4;
  const int b = blockIdx.x * 16 + i;
  const bool valid = !RAGGED || b < B;
  const int bc = valid ? b : B - 1;
  const int n0 = wv * 32 + g * 4;                  // this lane's units: n0 + 16 mt + r, mt = 0, 1
  constexpr bool hs_last = AUX == 3;
  const int xu = tid >> 5, xf = tid & 31;          // the feature value this lane converts each step
  const int xb = blockIdx.x * 16 + xu;
  const int xbc = (!RAGGED || xb < B) ? xb : B - 1;

  if (tid < H2) { sbias[0][tid] = bz[tid]; sbias[1][tid] = bh[tid]; }
  f32x4 hown[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) hown[mt] = ld4(h0 + (size_t)bc * H2 + n0 + 16 * mt);
  const float sz = fsigmoid(zeta[0]), sn = fsigmoid(nu[0]);

  // The fp16 path needs |h| inside fp16's range for the whole scan: see fwd_scan_split_w8 (kernels_split.hip) for the
  // bound |h_t| <= max(|h0|, 1) + t under gates with z in [0,1]; the workgroup checks its own rows of h0.
  bool use_h16 = false;
  if constexpr (F16H) {
    float hm = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) hm = fmaxf(hm, fabsf(hown[mt][r]));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) if (!(hown[mt][r] == hown[mt][r])) hm = 3.0e38f;   // NaN: no range assumption
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) hm = fmaxf(hm, __shfl_xor(hm, m));
    if (l == 0) hmax_s[wv] = hm;
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) hm = fmaxf(hm, hmax_s[k2]);
    use_h16 = __builtin_amdgcn_readfirstlane((int)(hm + (float)Tn + 2.0f < 3.0e4f)) != 0;
    if (tid == 0) flags[blockIdx.x] = use_h16 ? 0u : 1u;
    if (!use_h16) return;                            // the bf16 launch behind this one takes the workgroup
  }

  auto scan = [&](auto h16_tag) __attribute__((always_inline)) {
  constexpr bool H16 = decltype(h16_tag)::value;
  constexpr int NPL = H16 ? 2 : 3, NBUF = H16 ? 2 : 1;
  unsigned char* const hpl = smem + (H16 ? 0 : U2L);                   // [NBUF][NPL][PLH2]
  unsigned char* const xpl = hpl + NBUF * NPL * PLH2;                  // [NBUF][3][PLX2]
  u32x4* const u2l = reinterpret_cast<u32x4*>(smem);                   // bf16 path: [(wv*2 + mt)*8 + s][lane]
  u32x4* const wfl = reinterpret_cast<u32x4*>(xpl + NBUF * 3 * PLX2);  // fp16 path: [(wv*2 + mt)*3 + plane][lane]

  // ---- resident A operands ----------------------------------------------------------------------
  Frag2h Uh[2][KS2];               // fp16 path
  u32x4 U0[2][KS2], U1[2][KS2];    // bf16 path: planes 0 and 1 (plane 2 in LDS)
  Frag3 Wf[2];
  float u_unscale = 1.0f;
  {
    float umax = 0.f;
    if (H16) {                     // one exact power of two for the wave's rows: largest element into [2^12, 2^13)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
          const float* up = u + (size_t)(wv * 32 + 16 * mt + i) * H2 + 32 * s + 8 * g;
          const f32x4 a = ld4(up), c = ld4(up + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) umax = fmaxf(umax, fmaxf(fabsf(a[j]), fabsf(c[j])));
        }
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) umax = fmaxf(umax, __shfl_xor(umax, m));
    }
    int e = 0;
    if (umax > 0.f && umax < 3.0e38f) (void)frexpf(umax, &e);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const float u_scale = ldexpf(1.0f, 13 - e);
    if (H16) u_unscale = ldexpf(1.0f, e - 13);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int nA = wv * 32 + 16 * mt + i;          // A row i of tile mt, K in natural unit order
#pragma unroll
      for (int s = 0; s < KS2; ++s) {
        const float* up = u + (size_t)nA * H2 + 32 * s + 8 * g;
        const f32x4 a = ld4(up), c = ld4(up + 4);
        if (H16) {
          Uh[mt][s] = split2h8(a * u_scale, c * u_scale);
        } else {
          const Frag3 f = split3(a, c);
          U0[mt][s] = f.p[0]; U1[mt][s] = f.p[1];
          u2l[((wv * 2 + mt) * KS2 + s) * 64 + l] = f.p[2];
        }
      }
      if (!PREIN) {
        const float* wp = w + (size_t)nA * F2 + 8 * g;
        Wf[mt] = split3(ld4(wp), ld4(wp + 4));
        if (H16) {
#pragma unroll
          for (int p = 0; p < 3; ++p) wfl[((wv * 2 + mt) * 3 + p) * 64 + l] = Wf[mt].p[p];
        }
      }
    }
  }

  auto publish_h = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const unsigned off = (unsigned)(buf * NPL * PLH2 + i * ROWH2 + (n0 + 16 * mt) * 2);
      if (H16) {
        uint2 hi, lo;
        split2h(hown[mt][0], hown[mt][1], hi.x, lo.x); split2h(hown[mt][2], hown[mt][3], hi.y, lo.y);
        *reinterpret_cast<uint2*>(hpl + off) = hi;
        *reinterpret_cast<uint2*>(hpl + PLH2 + off) = lo;
      } else {
        uint2 q0, q1, q2;
        split_quad(hown[mt], q0, q1, q2);
        *reinterpret_cast<uint2*>(hpl + off) = q0;
        *reinterpret_cast<uint2*>(hpl + PLH2 + off) = q1;
        *reinterpret_cast<uint2*>(hpl + 2 * PLH2 + off) = q2;
      }
    }
  };
  auto publish_x = [&](int buf, unsigned bits) __attribute__((always_inline)) {   // bits: as load_x returned them
    const float v = BF ? bitsf(bits << 16) : bitsf(bits);
    unsigned short s0, s1, s2;
    split_one(v, s0, s1, s2);
    const unsigned off = (unsigned)(buf * 3 * PLX2 + xu * ROWX2 + xf * 2);
    *reinterpret_cast<unsigned short*>(xpl + off) = s0;
    *reinterpret_cast<unsigned short*>(xpl + PLX2 + off) = s1;
    *reinterpret_cast<unsigned short*>(xpl + 2 * PLX2 + off) = s2;
  };
  // this lane's value of frame t, as raw bits (a bf16 value is widened when it is published, not here: the conversion
  // would wait for the load at once)
  const char* xlane = reinterpret_cast<const char*>(x) + ((size_t)xbc * xsB + xf) * (BF ? 2 : 4);
  auto load_x = [&](int t) __attribute__((always_inline)) -> unsigned {
    if (BF) return (unsigned)*reinterpret_cast<const unsigned short*>(xlane + (size_t)t * xsT * 2);
    return *reinterpret_cast<const unsigned*>(xlane + (size_t)t * xsT * 4);
  };
  // PREIN: this lane's eight values of P(t) (rows of lanes beyond a ragged batch: the last utterance's)
  const float* plane = x + (size_t)bc * hsB + n0;
  auto load_p = [&](int t, f32x4 (&q)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) q[mt] = ld4(plane + (size_t)t * hsT + 16 * mt);
  };
  const unsigned lane_hs = (unsigned)b * hsB + n0, lane_bh = (unsigned)b * H2 + n0;   // in a sequence / in a [B,H] tensor
  auto store_step = [&](int t, const f32x4* aux) __attribute__((always_inline)) {   // hown holds h_t
    if (hs_last && t != Tn - 1) return;               // (wave-uniform) the classifier reads h_T only: model.py:227
    if (valid) {
      const size_t eo = hs_last ? (size_t)lane_bh : (size_t)t * hsT + lane_hs;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (BF) st4_bf16(reinterpret_cast<unsigned short*>(hs) + eo + 16 * mt, hown[mt]);
        else st4(hs + eo + 16 * mt, hown[mt]);
      }
      if (AUX == 2) {
        float* po = zs + (size_t)t * hsT + lane_hs;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) st4(po + 16 * mt, aux[mt]);
      }
    }
  };

  __syncthreads();                                   // sbias staged; bf16 path: every wave's plane-2 fragments written
  publish_h(0);
  unsigned xnext = 0u;
  f32x4 pnext[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (PREIN) {
    load_p(0, pnext);
  } else {
    publish_x(0, load_x(0));
    xnext = load_x(Tn > 1 ? 1 : 0);                  // frame t+1, published during step t
  }
  __syncthreads();

  f32x4 aux_prev[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  for (int t = 0; t < Tn; ++t) {
    const int cur = H16 ? (t & 1) : 0, nxt = H16 ? (cur ^ 1) : 0;
    const unsigned xpub = xnext;
    const f32x4 pcur[2] = {pnext[0], pnext[1]};
    if (PREIN) load_p(t + 1 < Tn ? t + 1 : Tn - 1, pnext);
    else xnext = load_x(t + 2 < Tn ? t + 2 : Tn - 1);
    const unsigned char* hp = hpl + cur * NPL * PLH2;
    const unsigned char* xp = xpl + cur * 3 * PLX2;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a[2] = {z4, z4}, alo[2] = {z4, z4};        // W.x (+ U.h on the bf16 path): big / small terms
    f32x4 ah[2] = {z4, z4}, ahl[2] = {z4, z4};       // fp16 path: U.h scaled by 2^k
    constexpr int KB = H16 ? 4 : 2;                  // K-steps per fragment batch
    // every fragment read of a batch is ISSUED, each into registers of its own, before the first MFMA that reads
    // one; a batch's registers are reloaded only after its MFMAs have retired (operand rule)
    if constexpr (PREIN) {
      if (t > 0) store_step(t - 1, aux_prev);
      a[0] = pcur[0]; a[1] = pcur[1];                // W.x_t, from the frame GEMM (.cu:356 for all steps at once)
    } else {
      // ---- batch W: the frame product W.x_t (.cu:356) ---------------------------------------------------------
      Frag3 xB, Wl[2];
#pragma unroll
      for (int p = 0; p < 3; ++p) xB.p[p] = *reinterpret_cast<const u32x4*>(xp + p * PLX2 + i * ROWX2 + 16 * g);
      if (H16) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int p = 0; p < 3; ++p) Wl[mt].p[p] = wfl[((wv * 2 + mt) * 3 + p) * 64 + l];
      }
      if (t > 0) store_step(t - 1, aux_prev);        // h_{t-1} (+ its pre-activation): issued during the LDS round trip
      if (H16) publish_x(nxt, xpub);                 // (double-buffered: the next frame's planes go out here)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const Frag3& Wm = H16 ? Wl[mt] : Wf[mt];
        if (BF) {                                    // a bf16 frame is its own first plane: three of the six terms, same bits
          alo[mt] = mfma_bf16(Wm.p[2], xB.p[0], alo[mt]);
          alo[mt] = mfma_bf16(Wm.p[1], xB.p[0], alo[mt]);
          a[mt] = mfma_bf16(Wm.p[0], xB.p[0], a[mt]);
        } else {
          mfma6_hl(Wm, xB, a[mt], alo[mt]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      float touch = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) touch += a[mt][0] + alo[mt][0];
      completion_read(touch);
      if (H16) {                                     // fp16 path: W.x is complete; one sum instead of two accumulators
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt] += alo[mt];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k0 = 0; k0 < KS2; k0 += KB) {
      Frag2h hH[KB];
      Frag3 hB[KB];
      u32x4 U2[2][KB];
#pragma unroll
      for (int s = 0; s < KB; ++s) {
        const unsigned o = (unsigned)(i * ROWH2 + 64 * (k0 + s) + 16 * g);
        if (H16) {
          hH[s].hi = *reinterpret_cast<const u32x4*>(hp + o);
          hH[s].lo = *reinterpret_cast<const u32x4*>(hp + PLH2 + o);
        } else {
#pragma unroll
          for (int p = 0; p < 3; ++p) hB[s].p[p] = *reinterpret_cast<const u32x4*>(hp + p * PLH2 + o);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) U2[mt][s] = u2l[((wv * 2 + mt) * KS2 + k0 + s) * 64 + l];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < KB; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (H16) {
            mfma3h_hl(Uh[mt][k0 + s], hH[s], ah[mt], ahl[mt]);                          // .cu:368, scaled by 2^k
          } else {                                                                       // .cu:368, six terms
            alo[mt] = mfma_bf16(U2[mt][s], hB[s].p[0], alo[mt]);
            alo[mt] = mfma_bf16(U1[mt][k0 + s], hB[s].p[1], alo[mt]);
            alo[mt] = mfma_bf16(U0[mt][k0 + s], hB[s].p[2], alo[mt]);
            alo[mt] = mfma_bf16(U1[mt][k0 + s], hB[s].p[0], alo[mt]);
            alo[mt] = mfma_bf16(U0[mt][k0 + s], hB[s].p[1], alo[mt]);
            a[mt] = mfma_bf16(U0[mt][k0 + s], hB[s].p[0], a[mt]);
          }
        }
      __builtin_amdgcn_sched_barrier(0);             // (the scheduler otherwise sinks MFMAs below the read)
      float touch = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) touch += H16 ? (ah[mt][0] + ahl[mt][0]) : (a[mt][0] + alo[mt][0]);
      completion_read(touch);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: .cu:55-58 -----------------------------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const f32x4 pre = H16 ? a[mt] + (ah[mt] + ahl[mt]) * u_unscale : a[mt] + alo[mt];
      const f32x4 bzq = *reinterpret_cast<const f32x4*>(&sbias[0][n0 + 16 * mt]);
      const f32x4 bhq = *reinterpret_cast<const f32x4*>(&sbias[1][n0 + 16 * mt]);
      f32x4 zq, cq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = gate_act<GATE>(pre[r] + bzq[r]);
        const float c = ftanh(pre[r] + bhq[r]);
        hown[mt][r] = (sz * (1.0f - z) + sn) * c + hown[mt][r] * z;
        zq[r] = z; cq[r] = c;
      }
      if (AUX == 1 && valid) {                       // reference operator outputs: stored at once
        const size_t o = (size_t)t * hsT + lane_hs + 16 * mt;
        st4(zs + o, zq); st4(cs + o, cq);
      }
      aux_prev[mt] = pre;
    }
    if (!H16) {                                      // single-buffered planes: everyone has read h_{t-1}, x_t by now
      lds_barrier();
      if (!PREIN) publish_x(0, xpub);
    }
    publish_h(nxt);
    lds_barrier();
  }
  store_step(Tn - 1, aux_prev);
  {                                                  // the weight fragments stay allocated through the last step
    float probe = hown[0][0];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (!H16 && !PREIN) keep_alive(probe, Wf[mt]);
#pragma unroll
      for (int s = 0; s < KS2; ++s) {
        if (H16) asm volatile("" : "+v"(probe) : "v"(Uh[mt][s].hi), "v"(Uh[mt][s].lo));
        else asm volatile("" : "+v"(probe) : "v"(U0[mt][s]), "v"(U1[mt][s]));
      }
    }
  }
  };
  if constexpr (F16H) scan(std::true_type{}); else scan(std::false_type{});
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
// Reverse scan, recurrence only: per step t EW of the lane's 8 elements (z, c recomputed from the saved
// pre-activation under PREACT, else read) -> d_pre_t to the workspace -> chain  d_h = z*g + U^T d_pre  for the
// wave's own 32 units over all eight K-steps.
//
// Where U^T lives decides everything here.  Three bf16 planes are 384 KB; two of them in registers (128 per lane at
// two waves per SIMD) plus one in LDS was built first and could not be compiled without spilling ~70 registers of
// weight fragments inside the loop (the compiler wants ~160 registers for the rest of the kernel; one wave per SIMD
// with 512 registers did no better: the MFMA operands have to pass through the 256 architectural VGPRs).  So the
// chain runs on fp16 TWO-plane operands, hi = fp16(v), lo = fp16(v - hi), three MFMAs per K-step (lo.hi, hi.lo,
// hi.hi) like the forward's state product: U^T's hi plane in registers (64 per lane), its lo plane in LDS in
// fragment order (128 KB, each wave re-reading only its own fragments).  U^T is bounded and pre-scaled by one exact
// power of two per wave.  d_pre is NOT bounded -- the reason kernels_split.hip keeps bf16 planes for gradients: a
// small-magnitude gradient falls into fp16's subnormal range -- so every utterance's row of d_pre_t is scaled by its
// own exact power of two that puts the row's largest magnitude in [2^11, 2^12) before the split into THREE fp16
// planes (exact for everything within 2^-26 of that largest element).  The maximum over the row's 256 units crosses
// the waves through LDS on the barrier the step has anyway; with ONE scale per utterance the chain's eight K-steps
// accumulate in the matrix pipe and are un-scaled once per result (the first form scaled per 32-unit producer slice
// to avoid the cross-wave maximum, and paid an add and an fma per result and K-step for it).
// d_bz, d_bh, d_zeta, d_nu partial sums per workgroup.  mode bit 1 (FASTGRNN_FLAG_GRAD_LAST): ghs is [B,H], the
// gradient of the last state alone.
// BF: FASTGRNN_BF16_IO -- grad_hs and hs are bf16 in HBM (the saved pre-activation, h0, d_pre and d_h0 stay fp32).
template <int GATE, bool PREACT, bool RAGGED, bool BF = false>
__global__ __launch_bounds__(512) void bwd_scan_h256(
    int Tn, int B, int mode, unsigned hsT, unsigned hsB, const float* __restrict__ ghs, const float* __restrict__ hs,
    const float* __restrict__ aux0, const float* __restrict__ aux1, const float* __restrict__ h0,
    const float* __restrict__ u, const float* __restrict__ bz, const float* __restrict__ bh,
    const float* __restrict__ zeta, const float* __restrict__ nu,
    float* __restrict__ d_h0, float* __restrict__ dpre_ws, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[U2L + 3 * PLH2];
  __shared__ __attribute__((aligned(16))) float smax[16][8];        // largest |d_pre_t| of (utterance, producer wave)
  __shared__ __attribute__((aligned(16))) float sbias[2][H2];
  __shared__ float red[16];

  const bool g_last = (mode & 2) != 0;
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, i = l & 15, g = l >> 4;
  const int b = blockIdx.x * 16 + i;
  const bool valid = !RAGGED || b < B;
  const int bc = valid ? b : B - 1;
  const int n0 = wv * 32 + g * 4;                    // this lane's units: n0 + 16 mt + r
  const float sz = fsigmoid(zeta[0]), sn = fsigmoid(nu[0]);
  if (PREACT && tid < H2) { sbias[0][tid] = bz[tid]; sbias[1][tid] = bh[tid]; }
  u32x4* const ulo = reinterpret_cast<u32x4*>(smem); // lo plane of U^T: [(wv*2 + mt)*8 + s][lane]
  unsigned char* const dpl = smem + U2L;             // [3][PLH2]: the three fp16 planes of the scaled d_pre_t, [utterance][unit]

  // ---- resident A operands: d_h[k][b] = sum_n U[n][k] d_pre[b][n]; A row i of tile mt is k = 32w + 16mt + i ----
  u32x4 UTh[2][KS2];
  float u_unscale;
  {
    f32x4 lo[2][KS2], hi[2][KS2];
    float umax = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int k = wv * 32 + 16 * mt + i;
#pragma unroll
      for (int s = 0; s < KS2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          lo[mt][s][j] = u[(size_t)(32 * s + 8 * g + j) * H2 + k];
          hi[mt][s][j] = u[(size_t)(32 * s + 8 * g + 4 + j) * H2 + k];
          umax = fmaxf(umax, fmaxf(fabsf(lo[mt][s][j]), fabsf(hi[mt][s][j])));
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) umax = fmaxf(umax, __shfl_xor(umax, m));
    int e = 0;
    if (umax > 0.f && umax < 3.0e38f) (void)frexpf(umax, &e);      // umax = f * 2^e, f in [0.5, 1)
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const float u_scale = ldexpf(1.0f, 13 - e);                      // largest element into [2^12, 2^13)
    u_unscale = ldexpf(1.0f, e - 13);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int s = 0; s < KS2; ++s) {
        const Frag2h f = split2h8(lo[mt][s] * u_scale, hi[mt][s] * u_scale);
        UTh[mt][s] = f.hi;
        ulo[((wv * 2 + mt) * KS2 + s) * 64 + l] = f.lo;
      }
  }

  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 sbz[2] = {z4, z4}, sbh[2] = {z4, z4}, dh[2] = {z4, z4};
  float pz = 0.f, pn = 0.f, pz_c = 0.f, pn_c = 0.f;   // d_zeta / d_nu partial sums, compensated

  // Addresses: a wave-uniform step base (scalar registers) + a 32-bit lane offset (the host rejects B*H*4 >= 2^31)
  // hsT / hsB: element strides of one step / one utterance in grad_hs, hs, the saved tensors AND the d_pre workspace,
  // whose rows follow the sequences' order (time-major: B*256, 256; FASTGRNN_FLAG_BATCH_MAJOR: 256, T*256) so that the
  // GEMMs behind this scan pair its rows with the caller's rows of x and hs as they lie.
  const unsigned lane_c = ((unsigned)bc * hsB + n0) * 4u, lane_v = ((unsigned)b * hsB + n0) * 4u;   // BYTE offsets in a sequence
  const unsigned lane_c0 = ((unsigned)bc * H2 + n0) * 4u, lane_v0 = ((unsigned)b * H2 + n0) * 4u;   // ... in a [B,H] tensor
  auto ldg = [](const float* base, unsigned off) __attribute__((always_inline)) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
  };
  // four bf16 sequence elements as raw bits in .xy (fp32 byte offset halved); widened where they are consumed
  auto ldgb = [](const float* base, unsigned off) __attribute__((always_inline)) {
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(base) + (off >> 1));
    return f32x4{bitsf(r.x), bitsf(r.y), 0.f, 0.f};
  };
  auto widen = [](const f32x4 r) __attribute__((always_inline)) {
    const unsigned a = fbits(r[0]), c = fbits(r[1]);
    return f32x4{bitsf(a << 16), bitsf(a & 0xFFFF0000u), bitsf(c << 16), bitsf(c & 0xFFFF0000u)};
  };
  const unsigned dp_step = valid ? hsT * 4u : 0u;
  const unsigned dp_off = valid ? lane_v : (((unsigned)Tn * (unsigned)B + i) * H2 + n0) * 4u;
  auto stg = [](float* base, unsigned off, f32x4 v) __attribute__((always_inline)) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + off) = v;
  };
  struct EwOps { f32x4 g[2], a0[2], a1[2], h[2]; };
  auto load_ew = [&](int t, EwOps& e) __attribute__((always_inline)) {
    const size_t step = (size_t)t * hsT;                                     // uniform
    const size_t sstep = BF ? step / 2 : step;                               // (in floats) of a bf16 sequence
    const float* gt = g_last ? ghs : ghs + sstep;
    const float* p0 = aux0 + step;
    const float* p1 = PREACT ? aux0 : aux1 + step;
    const float* ht = (t == 0) ? h0 : hs + (sstep - (BF ? (size_t)hsT / 2 : (size_t)hsT));   // .cu:478-481
    // (bf16: step 0 reads the fp32 h0, the others a bf16 row of hs.  BOTH loads are issued every step -- the bf16 one
    // from row 0 of hs at step 0, the fp32 one from h0, an L2 hit -- and the value is picked with selects: no
    // wave-uniform branch around memory instructions inside the scan)
    const float* hb = (t == 0) ? hs : ht;
    const unsigned lane_g = g_last ? lane_c0 : lane_c, lane_h = (t == 0) ? lane_c0 : lane_c;
    const bool g_zero = (g_last && t != Tn - 1) || (RAGGED && !valid);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      // lanes beyond a ragged batch: the last utterance's rows with a ZERO gradient (dh starts at zero: gg, d_pre and
      // every sum they enter stay exactly zero for them)
      e.g[mt] = g_zero ? z4 : (BF ? ldgb(gt, lane_g + 64u * mt) : ldg(gt, lane_g + 64u * mt));
      e.a0[mt] = ldg(p0, lane_c + 64u * mt);
      if (!PREACT) e.a1[mt] = ldg(p1, lane_c + 64u * mt);
      if (BF) {
        const f32x4 h32 = ldg(h0, lane_c0 + 64u * mt), h16 = ldgb(hb, lane_c + 64u * mt);
        const bool first = t == 0;
        e.h[mt] = f32x4{first ? h32[0] : h16[0], first ? h32[1] : h16[1], first ? h32[2] : 0.f, first ? h32[3] : 0.f};
      } else {
        e.h[mt] = ldg(ht, lane_h + 64u * mt);
      }
    }
  };

  // ONE operand set: EW(t) consumes it at the top of the step and the requests for EW(t-1) refill it right behind
  // the step's LDS hand-off (two sets and a loop unrolled by two cost ~40 registers more and spilled)
#ifdef FASTGRNN_DIAG_STAMPS
  unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dlast)::"memory");
#endif
  auto step = [&](int t, EwOps& e) __attribute__((always_inline)) {
    SPLIT_STAMP(0)
    // ---- EW(t): .cu:107-117 ---------------------------------------------------------------------------------
    f32x4 dpv[2];
    float amax = 0.f, sn8 = 0.f, sz8 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 bzq = z4, bhq = z4;
      if (PREACT) {
        bzq = *reinterpret_cast<const f32x4*>(&sbias[0][n0 + 16 * mt]);
        bhq = *reinterpret_cast<const f32x4*>(&sbias[1][n0 + 16 * mt]);
      }
      if (BF) {
        e.g[mt] = widen(e.g[mt]);
        const f32x4 hw = widen(e.h[mt]);               // (selects, no branch: see load_ew)
        const bool first = t == 0;
        e.h[mt] = f32x4{first ? e.h[mt][0] : hw[0], first ? e.h[mt][1] : hw[1], first ? e.h[mt][2] : hw[2], first ? e.h[mt][3] : hw[3]};
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float z, c;
        if (PREACT) {
          z = gate_act<GATE>(e.a0[mt][r] + bzq[r]);
          c = ftanh(e.a0[mt][r] + bhq[r]);
        } else {
          z = e.a0[mt][r]; c = e.a1[mt][r];
        }
        const float gg = e.g[mt][r] + dh[mt][r];                                 // .cu:474
        const float dcp = (sz * (1.0f - z) + sn) * (1.0f - c * c) * gg;          // .cu:109
        const float dzp = (e.h[mt][r] - sz * c) * gate_dact<GATE>(z) * gg;       // .cu:110
        const float cg = c * gg;
        sbz[mt][r] += dzp; sbh[mt][r] += dcp;
        sn8 += cg; sz8 += cg - z * cg;                                           // .cu:114-115
        dpv[mt][r] = dzp + dcp;                                                  // .cu:113
        dh[mt][r] = z * gg;                                                      // .cu:108: C-in of the chain
        amax = fmaxf(amax, fabsf(dpv[mt][r]));
      }
    }
    kahan_add(pn, pn_c, sn8); kahan_add(pz, pz_c, sz8);
    {
      // d_pre_t for the weight-gradient / d_x GEMMs.  Every lane stores, without a branch: lanes beyond a ragged
      // batch write to 16 sink rows behind the T*B rows (step stride 0) -- no conditional code around memory
      // instructions in the steady state of a scan (the wide-layer backward's masked store block)
      char* o = reinterpret_cast<char*>(dpre_ws) + (size_t)t * dp_step + dp_off;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) *reinterpret_cast<f32x4*>(o + 64 * mt) = dpv[mt];
    }
    // ---- the utterance's power of two: largest |d_pre_t| over all 256 units into [2^11, 2^12).  Each wave leaves the
    //      maximum over ITS 32 units in LDS before the barrier the step needs anyway, and reads all eight behind it: one
    //      exact scale per utterance and step, so the chain's eight K-steps accumulate in the matrix pipe (C-in) and are
    //      un-scaled ONCE -- round 3's first form scaled per (utterance, 32-unit slice) and paid an add and an fma per
    //      result and K-step for it (128 fp32 vector instructions per wave and step: vector time adds to matrix time,
    //      DESIGN.md 4.0).  Values below 2^-26 of the utterance's largest lose low bits in the planes: an absolute error
    //      below 2^-36 of that largest element, against the 2^-24 an fp32 sum over them carries.
    amax = fmaxf(amax, __shfl_xor(amax, 16));
    amax = fmaxf(amax, __shfl_xor(amax, 32));
    if (g == 0) smax[i][wv] = amax;
    SPLIT_STAMP(1)
    lds_barrier();                                   // every wave has finished reading the planes of step t+1
    SPLIT_STAMP(2)
    {
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(&smax[i][0]), m1 = *reinterpret_cast<const f32x4*>(&smax[i][4]);
      amax = fmaxf(fmaxf(fmaxf(m0[0], m0[1]), fmaxf(m0[2], m0[3])), fmaxf(fmaxf(m1[0], m1[1]), fmaxf(m1[2], m1[3])));
    }
    int ex = 12;                                     // all-zero rows (or inf / NaN, which propagate anyway): scale 1
    if (amax > 0.f && amax < 3.0e38f) (void)frexpf(amax, &ex);       // amax = f * 2^ex, f in [0.5, 1)
    ex = ex < -112 ? -112 : ex;                      // (2^(12-ex) must stay a normal float)
    const float dscale = ldexpf(1.0f, 12 - ex);
    // (the factor of U^T belongs to the CONSUMER's rows -- this wave's own -- and is applied with the un-scaling below.
    // Round 2 shipped the producer's u_unscale folded into the slice's factor: right only while every wave's block of
    // U^T has its maximum in the same binade, which random 0.1-scale matrices happen to satisfy and a product U2.U1 does not)
    const float unscale = ldexpf(1.0f, ex - 12) * u_unscale;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      uint2 p0q, p1q, p2q;
      split3h(dpv[mt][0] * dscale, dpv[mt][1] * dscale, p0q.x, p1q.x, p2q.x);
      split3h(dpv[mt][2] * dscale, dpv[mt][3] * dscale, p0q.y, p1q.y, p2q.y);
      const unsigned off = (unsigned)(i * ROWH2 + (n0 + 16 * mt) * 2);
      *reinterpret_cast<uint2*>(dpl + off) = p0q;
      *reinterpret_cast<uint2*>(dpl + PLH2 + off) = p1q;
      *reinterpret_cast<uint2*>(dpl + 2 * PLH2 + off) = p2q;
    }
    SPLIT_STAMP(3)
    lds_barrier();
    SPLIT_STAMP(4)
    // requests for EW(t-1): behind the reads of dh above (they may land in registers the last chain's fragments used)
    if (t > 0) load_ew(t - 1, e);
    __builtin_amdgcn_sched_barrier(0);
    // ---- chain(t): d_h = z*g + U^T d_pre_t (.cu:537): per K-step five fp16 MFMAs per row tile, accumulated over the
    //      eight K-steps in the matrix pipe and un-scaled into dh once ------------------------------------------------
    // Software pipeline over the K-steps, two operand sets and two accumulator sets (even / odd K-steps): while the
    // MFMAs of K-step k execute, a read of K-step k-1's accumulators proves that one has retired and the fragments of
    // K-step k+1 are requested into the operand set it used.  (One accumulator set would make that read wait for
    // K-step k as well; taking one K-step at a time cost 350 cycles per K-step, tools/diag_h256.hip.)
    struct KOps { Frag2h dB; u32x4 d2, Ul[2]; };
    auto req = [&](int k, KOps& o) __attribute__((always_inline)) {
      const unsigned off = (unsigned)(i * ROWH2 + 64 * k + 16 * g);
      o.dB.hi = *reinterpret_cast<const u32x4*>(dpl + off);
      o.dB.lo = *reinterpret_cast<const u32x4*>(dpl + PLH2 + off);
      o.d2 = *reinterpret_cast<const u32x4*>(dpl + 2 * PLH2 + off);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) o.Ul[mt] = ulo[((wv * 2 + mt) * KS2 + k) * 64 + l];
    };
    // Five terms per K-step and row tile (round 3): U^T as two fp16 planes (22 bits: a fixed relative perturbation of
    // the weights of 2^-23, the size of their own fp32 rounding), d_pre as THREE (exact), the one dropped term
    // (U^T's low plane against d_pre's third) below 2^-33.  The four small terms go into an accumulator of their own
    // (inside an MFMA the addends are chopped at the largest one: mfma6_hl).  With three terms -- two
    // planes of d_pre, lo.lo dropped -- d_zeta / d_nu sat at 4-5.5e-5 of the result at B = 4096 and 4.5e-4 on 8x weights.
    // pr[mt]: big terms, pr[2 + mt]: small terms of one accumulator set
    auto issue = [&](int k, const KOps& o, f32x4* pr) __attribute__((always_inline)) {
      const bool first = k < 2;                        // (compile-time: the set's first K-step starts from the constant 0)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 a = mfma_f16(o.Ul[mt], o.dB.lo, first ? z4 : pr[2 + mt]);   // smallest first
        a = mfma_f16(UTh[mt][k], o.d2, a);
        a = mfma_f16(o.Ul[mt], o.dB.hi, a);
        pr[2 + mt] = mfma_f16(UTh[mt][k], o.dB.lo, a);
        pr[mt] = mfma_f16(UTh[mt][k], o.dB.hi, first ? z4 : pr[mt]);
      }
    };
    auto retire = [&](const KOps& o, const f32x4* pr) __attribute__((always_inline)) {
      // one element of every accumulator the K-step wrote (the scheduler orders the two tiles' chains freely), combined
      // with integer instructions (they ride between MFMAs for free, fp32 additions do not: DESIGN.md 4.0); the
      // operand set stays allocated up to here -- the compiler considers it dead once its MFMAs have ISSUED
      const unsigned bits = (fbits(pr[0][0]) | fbits(pr[1][0])) | (fbits(pr[2][0]) | fbits(pr[3][0]));
      completion_read(bitsf(bits));
      asm volatile("" :: "v"(o.dB.hi), "v"(o.dB.lo), "v"(o.d2), "v"(o.Ul[0]), "v"(o.Ul[1]));
    };
    KOps o0, o1;
    f32x4 p0[4], p1[4];
    req(0, o0);
    req(1, o1);
    __builtin_amdgcn_sched_barrier(0);
    static_for<KS2>([&](auto k_tag) __attribute__((always_inline)) {
      constexpr int k = decltype(k_tag)::value;
      if constexpr ((k & 1) == 0) issue(k, o0, p0); else issue(k, o1, p1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (k >= 1) {
        // K-step k-1 has retired: its operand set is free for K-step k+1
        if constexpr ((k & 1) == 0) { retire(o1, p1); __builtin_amdgcn_sched_barrier(0); if constexpr (k + 1 < KS2) req(k + 1, o1); }
        else                        { retire(o0, p0); __builtin_amdgcn_sched_barrier(0); if constexpr (k + 1 < KS2) req(k + 1, o0); }
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    retire(o1, p1);                                  // K-step 7
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dh[mt][r] = fmaf((p0[mt][r] + p1[mt][r]) + (p0[2 + mt][r] + p1[2 + mt][r]), unscale, dh[mt][r]);
  };

  EwOps ea;
  __syncthreads();                                   // sbias, lo-plane fragments
  load_ew(Tn - 1, ea);
  for (int t = Tn - 1; t >= 0; --t) step(t, ea);
#ifdef FASTGRNN_DIAG_STAMPS
  { SPLIT_STAMP(5) }
  if (blockIdx.x == 7 && l == 0) { for (int k2 = 0; k2 < 8; ++k2) g_sdiag[wv][k2] = dsum[k2]; }
#endif
  // ---- flush ---------------------------------------------------------------------------------
  if (valid) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) stg(d_h0, lane_v0 + 64u * mt, dh[mt]);
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = sbz[mt][r], c = sbh[mt][r];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); c += __shfl_xor(c, m); }
      if (i == 0) {
        float* pb = part + (size_t)blockIdx.x * SLAB2;
        pb[n0 + 16 * mt + r] = a;
        pb[H2 + n0 + 16 * mt + r] = c;
      }
    }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { pz += __shfl_xor(pz, m); pn += __shfl_xor(pn, m); }
  if (l == 0) { red[wv] = pz; red[8 + wv] = pn; }
  __syncthreads();
  if (tid == 0) {
    float* pzn = part + (size_t)blockIdx.x * SLAB2 + 2 * H2;
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a += red[k]; c += red[8 + k]; }
    pzn[0] = a; pzn[1] = c;
  }
  {
    float probe = dh[0][0];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int s = 0; s < KS2; ++s) asm volatile("" : "+v"(probe) : "v"(UTh[mt][s]));
  }
}

// bias / zeta / nu gradients: fixed-order sum over workgroups
__global__ __launch_bounds__(1024) void reduce_h256_small(int nwg, const float* __restrict__ part,
                                                          const float* __restrict__ zeta, const float* __restrict__ nu,
                                                          float* __restrict__ d_bz, float* __restrict__ d_bh,
                                                          float* __restrict__ d_zeta, float* __restrict__ d_nu) {
  __shared__ float sm[16][64];
  const int o = threadIdx.x & 63, pid = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;               // 0 .. 2*256+1
  float a = 0.f;
  if (idx < 2 * H2 + 2) {
    for (int wg0 = pid; wg0 < nwg; wg0 += 64) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int wg = wg0 + 16 * j; v[j] = wg < nwg ? part[(size_t)wg * SLAB2 + idx] : 0.f; }
      a += (v[0] + v[1]) + (v[2] + v[3]);
    }
  }
  sm[pid][o] = a;
  __syncthreads();
  if (pid == 0 && idx < 2 * H2 + 2) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += sm[j][o];
    if (idx < H2) d_bz[idx] = t;
    else if (idx < 2 * H2) d_bh[idx - H2] = t;
    else if (idx == 2 * H2) { const float sz = 1.0f / (1.0f + expf(-zeta[0])); d_zeta[0] = t * sz * (1.0f - sz); }   // .cu:116,544
    else { const float sn = 1.0f / (1.0f + expf(-nu[0])); d_nu[0] = t * sn * (1.0f - sn); }                          // .cu:117,545
  }
}

static inline size_t h256_flag_bytes(const fastgrnn_desc& d) { return align256((size_t)((d.B + 15) / 16) * sizeof(unsigned)); }

struct H256BwdWs { size_t part, dpre, tn, xtm, total; };
H256BwdWs h256_bwd_layout(const fastgrnn_desc& d) {
  const size_t TB = (size_t)d.T * d.B, nwg = (d.B + 15) / 16;
  H256BwdWs L; size_t o = 0;
  L.part = o; o += align256(nwg * SLAB2 * 4);
  L.dpre = o; o += align256((TB + 16) * H2 * 4);     // + 16 sink rows for the lanes beyond a ragged batch
  const size_t tn_u = tn_gemm_big_ws(TB, H2, H2), tn_w = tn_gemm_big_ws(TB, H2, d.F);
  L.tn = o; o += tn_u > tn_w ? tn_u : tn_w;
  // FASTGRNN_FLAG_X_BFT: the time-major copy of x for the dW GEMM; the d_x GEMM then writes over it and the result is
  // transposed into the caller's [B,F,T] tensor
  L.xtm = o; if (d.flags & FASTGRNN_FLAG_X_BFT) o += align256(TB * F2 * 4);
  L.total = o;
  return L;
}

template <int GATE>
void launch_fwd(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs, void* zs,
                void* cs, void* ws, hipStream_t s) {
  dim3 grid((d.B + 15) / 16), block(512);
  const bool ragged = (d.B % 16) != 0;
  const int aux = (d.flags & FASTGRNN_FLAG_HS_LAST) ? 3 : (zs == nullptr ? 0 : ((d.flags & FASTGRNN_FLAG_SAVE_PREACT) ? 2 : 1));
  unsigned* flags = reinterpret_cast<unsigned*>(ws);
  // FASTGRNN_FLAG_X_BFT: the loader's [B,F,T] frames are transposed into the workspace first (25 us at B = 4096; a
  // lane's frame-by-frame read of [B,F,T] in place touches 64 cache lines per wave load and cost the scan 120 us)
  if (d.flags & FASTGRNN_FLAG_X_BFT) {
    float* xtm = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + h256_flag_bytes(d));
    bft_transpose_f32(d.B, d.T, (const float*)x, xtm, true, s);
    x = xtm;
  }
  // a wider input (F = 64 / 128): the batched frame GEMM  P = X . W^T  into the workspace, then the PREIN scan on P
  const bool prein = d.F != F2, bf = d.dtype == FASTGRNN_BF16_IO;
  if (prein) {
    float* P = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + h256_flag_bytes(d));
    rows_gemm((size_t)d.T * d.B, H2, d.F, false, x, (const float*)p.w, P, bf, false, s);   // (bf16 frames, fp32 P)
    x = P;
  }
  // strides of the sequences (elements): FASTGRNN_FLAG_BATCH_MAJOR lays hs / zs / cs and x out as [B,T,*]; the
  // workspace copy of [B,F,T] frames is time-major whatever the flag says; P's rows follow the order of x's
  const bool bm = (d.flags & FASTGRNN_FLAG_BATCH_MAJOR) != 0, xbm = bm && !(d.flags & FASTGRNN_FLAG_X_BFT);
  const unsigned hsT = bm ? H2 : (unsigned)d.B * H2, hsB = bm ? (unsigned)d.T * H2 : H2;
  const unsigned xsT = xbm ? F2 : (unsigned)d.B * F2, xsB = xbm ? (unsigned)d.T * F2 : F2;
  auto go = [&](auto kern) __attribute__((always_inline)) {
    hipLaunchKernelGGL(kern, grid, block, 0, s, d.T, d.B, hsT, hsB, xsT, xsB, (const float*)x, (const float*)h0, (const float*)p.w,
                       (const float*)p.u, (const float*)p.bias_gate, (const float*)p.bias_update, (const float*)p.zeta,
                       (const float*)p.nu, (float*)hs, (float*)zs, (float*)cs, flags);
  };
  // fp16 two-plane state product only for gates that keep z in [0,1] (see fwd_scan_split_w8); FWD_BF16X3: A/B
  constexpr bool BOUNDED = GATE == FASTGRNN_NL_SIGMOID || GATE == FASTGRNN_NL_QUANT_SIGM || GATE == FASTGRNN_NL_QUANT_SIGM4;
  const bool h16 = BOUNDED && !(d.flags & FASTGRNN_FLAG_FWD_BF16X3);
  auto pick = [&](auto aux_tag) __attribute__((always_inline)) {
    constexpr int A = decltype(aux_tag)::value;
    auto with = [&](auto prein_tag, auto bf_tag) __attribute__((always_inline)) {
      constexpr bool PI = decltype(prein_tag)::value, BFV = decltype(bf_tag)::value;
      if constexpr (BOUNDED) {
        if (h16) {                                   // fp16 launch, then the bf16 one for workgroups it turned down
          if (ragged) { go(fwd_scan_h256<GATE, A, true, 1, PI, BFV>); go(fwd_scan_h256<GATE, A, true, 2, PI, BFV>); }
          else        { go(fwd_scan_h256<GATE, A, false, 1, PI, BFV>); go(fwd_scan_h256<GATE, A, false, 2, PI, BFV>); }
          return;
        }
      }
      if (ragged) go(fwd_scan_h256<GATE, A, true, 0, PI, BFV>); else go(fwd_scan_h256<GATE, A, false, 0, PI, BFV>);
    };
    // bf16 sequences (h256_supported: gates sigmoid / relu / tanh, hs alone or the one-saved-tensor contract)
    if constexpr (GATE <= FASTGRNN_NL_TANH && (A == 0 || A == 2)) {
      if (bf) {
        if (prein) with(std::true_type{}, std::true_type{}); else with(std::false_type{}, std::true_type{});
        return;
      }
    }
    if (prein) with(std::true_type{}, std::false_type{}); else with(std::false_type{}, std::false_type{});
  };
  if (aux == 3) pick(std::integral_constant<int, 3>{});
  else if (aux == 2) pick(std::integral_constant<int, 2>{});
  else if (aux == 1) pick(std::integral_constant<int, 1>{});
  else pick(std::integral_constant<int, 0>{});
}

template <int GATE>
void launch_bwd(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x, const void* hs,
                const void* a0, const void* a1, const void* h0, const fastgrnn_grads& g, void* ws, hipStream_t s) {
  const H256BwdWs L = h256_bwd_layout(d);
  char* base = reinterpret_cast<char*>(ws);
  float* part = (float*)(base + L.part); float* dpre = (float*)(base + L.dpre); float* tn = (float*)(base + L.tn);
  const int nwg = (d.B + 15) / 16;
  const bool ragged = (d.B % 16) != 0, preact = (d.flags & FASTGRNN_FLAG_SAVE_PREACT) != 0;
  const bool bm = (d.flags & FASTGRNN_FLAG_BATCH_MAJOR) != 0;
  const unsigned hsT = bm ? H2 : (unsigned)d.B * H2, hsB = bm ? (unsigned)d.T * H2 : H2;
  auto go = [&](auto kern) __attribute__((always_inline)) {
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 0, s, d.T, d.B, (d.flags & FASTGRNN_FLAG_GRAD_LAST) ? 2 : 0, hsT, hsB,
                       (const float*)ghs, (const float*)hs, (const float*)a0, (const float*)a1, (const float*)h0,
                       (const float*)p.u, (const float*)p.bias_gate, (const float*)p.bias_update, (const float*)p.zeta,
                       (const float*)p.nu, (float*)g.d_h0, dpre, part);
  };
  const bool bf = d.dtype == FASTGRNN_BF16_IO;       // (h256_supported: under SAVE_PREACT, gates sigmoid / relu / tanh)
  bool launched = false;
  if constexpr (GATE <= FASTGRNN_NL_TANH) {
    if (bf) {
      if (ragged) go(bwd_scan_h256<GATE, true, true, true>); else go(bwd_scan_h256<GATE, true, false, true>);
      launched = true;
    }
  }
  if (!launched) {
    if (preact) { if (ragged) go(bwd_scan_h256<GATE, true, true>); else go(bwd_scan_h256<GATE, true, false>); }
    else        { if (ragged) go(bwd_scan_h256<GATE, false, true>); else go(bwd_scan_h256<GATE, false, false>); }
  }
  hipLaunchKernelGGL(reduce_h256_small, dim3((2 * H2 + 2 + 63) / 64), dim3(1024), 0, s, nwg, part, (const float*)p.zeta,
                     (const float*)p.nu, (float*)g.d_bias_gate, (float*)g.d_bias_update, (float*)g.d_zeta, (float*)g.d_nu);
  const size_t TB = (size_t)d.T * d.B;
  // dU = d_pre^T . H_prev (rows of t = 0 are h0, the rest hs[t-1]);  dW = d_pre^T . X   (.cu:539-540 over all steps)
  // (batch-major: row b*T + t of d_pre pairs with hs row b*T + t - 1, every T-th row with h0[b])
  if (bm) tn_gemm_big_run_periodic(TB, H2, H2, dpre, H2, (const float*)h0, hs, (size_t)d.T, H2, tn, (float*)g.d_u, H2, s, bf);
  else tn_gemm_big_run(TB, H2, H2, dpre, H2, (const float*)h0, hs, (size_t)d.B, H2, tn, (float*)g.d_u, H2, s, bf);
  const bool bft = (d.flags & FASTGRNN_FLAG_X_BFT) != 0;
  float* xtm = (float*)(base + L.xtm);
  if (bft) bft_transpose_f32(d.B, d.T, (const float*)x, xtm, true, s);
  const float* xr = bft ? xtm : (const float*)x;
  tn_gemm_big_run(TB, H2, d.F, dpre, H2, xr, xr, (size_t)0, d.F, tn, (float*)g.d_w, d.F, s, bf);
  // d_x = d_pre . W   (.cu:538; W is [H,F] = [K,N]); skipped when the caller does not want the input's gradient
  // (g.d_x == NULL: the first layer of a model, whose input is data)
  if (g.d_x) {
    rows_gemm(TB, d.F, H2, true, dpre, (const float*)p.w, bft ? (void*)xtm : g.d_x, false, bf, s);   // (bf16 d_x)
    if (bft) bft_transpose_f32(d.B, d.T, xtm, (float*)g.d_x, false, s);
  }
}

}  // namespace

bool h256_shape(const fastgrnn_desc& d) {
  // B < 2^21: a step's rows are addressed with 32-bit lane offsets (B*H*4 bytes < 2^31)
  return d.w_rank == 0 && d.u_rank == 0 && d.H == H2 && (d.F == F2 || d.F == 64 || d.F == 128) && d.B < (1 << 21);
}

// fp32 sequences, time- or batch-major, every gate, both saved-tensor contracts, full or last-state outputs / gradients;
// bf16 sequences with the limits below
bool h256_supported(const fastgrnn_desc& d, int direction) {
  if (!h256_shape(d) || (d.dtype != FASTGRNN_F32 && d.dtype != FASTGRNN_BF16_IO)) return false;
  if (d.dtype == FASTGRNN_BF16_IO) {
    // bf16 sequences (x, hs, grad_hs, d_x): gates sigmoid / relu / tanh; the forward with hs alone or under
    // FASTGRNN_FLAG_SAVE_PREACT, the backward under it; no [B,F,T] frames, no last-state forward
    if (d.gate_nl > FASTGRNN_NL_TANH || (d.flags & (FASTGRNN_FLAG_X_BFT | FASTGRNN_FLAG_HS_LAST))) return false;
    if (direction == 1 && !(d.flags & FASTGRNN_FLAG_SAVE_PREACT)) return false;
    if (direction == 1 && (d.flags & FASTGRNN_FLAG_BATCH_MAJOR)) return false;   // (no bf16 variant of the periodic dU GEMM)
  }
  if (d.F != F2 && (d.flags & FASTGRNN_FLAG_X_BFT)) return false;      // (the loader's [B,F,T] batches: 32 features)
  // batch-major sequences: two-stride rows in both scans; the backward's d_pre rows then follow [B,T], which a
  // time-major workspace copy of [B,F,T] frames would not match
  if (direction == 1 && (d.flags & FASTGRNN_FLAG_BATCH_MAJOR) && (d.flags & FASTGRNN_FLAG_X_BFT)) return false;
  // 32-bit byte offsets inside the d_pre workspace, its 16 sink rows included
  if (((double)d.T * d.B + 16.0) * H2 * 4.0 >= 4294967296.0) return false;
  if (direction == 0 && (d.flags & FASTGRNN_FLAG_SAVE_PREACT) && (d.flags & FASTGRNN_FLAG_HS_LAST)) return false;
  const bool preact = (d.flags & FASTGRNN_FLAG_SAVE_PREACT) != 0;
  return d.gate_nl <= FASTGRNN_NL_TANH || direction == 0 || preact;
}

size_t h256_backward_ws(const fastgrnn_desc& d) { return h256_bwd_layout(d).total; }
// one word per workgroup: "my rows of h0 are outside the fp16 path's range" (fwd_scan_h256 MODE 1 -> MODE 2)
// + under FASTGRNN_FLAG_X_BFT the time-major copy of x
// + for F = 64 / 128 the frame product P[T*B, 256]
size_t h256_forward_ws(const fastgrnn_desc& d) {
  return h256_flag_bytes(d) + ((d.flags & FASTGRNN_FLAG_X_BFT) ? align256((size_t)d.T * d.B * F2 * 4) : 0) +
         (d.F != F2 ? align256((size_t)d.T * d.B * H2 * 4) : 0);
}

int h256_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs, void* zs,
                 void* cs, void* ws, hipStream_t s) {
  if (!ws) return FASTGRNN_ERR_WORKSPACE;
  switch (d.gate_nl) {
    case FASTGRNN_NL_SIGMOID: launch_fwd<FASTGRNN_NL_SIGMOID>(d, p, x, h0, hs, zs, cs, ws, s); break;
    case FASTGRNN_NL_RELU: launch_fwd<FASTGRNN_NL_RELU>(d, p, x, h0, hs, zs, cs, ws, s); break;
    case FASTGRNN_NL_TANH: launch_fwd<FASTGRNN_NL_TANH>(d, p, x, h0, hs, zs, cs, ws, s); break;
    case FASTGRNN_NL_QUANT_TANH: launch_fwd<FASTGRNN_NL_QUANT_TANH>(d, p, x, h0, hs, zs, cs, ws, s); break;
    case FASTGRNN_NL_QUANT_SIGM: launch_fwd<FASTGRNN_NL_QUANT_SIGM>(d, p, x, h0, hs, zs, cs, ws, s); break;
    default: launch_fwd<FASTGRNN_NL_QUANT_SIGM4>(d, p, x, h0, hs, zs, cs, ws, s); break;
  }
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

int h256_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x, const void* hs,
                  const void* zs, const void* cs, const void* h0, const fastgrnn_grads& g, void* ws, hipStream_t s) {
  switch (d.gate_nl) {
    case FASTGRNN_NL_SIGMOID: launch_bwd<FASTGRNN_NL_SIGMOID>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    case FASTGRNN_NL_RELU: launch_bwd<FASTGRNN_NL_RELU>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    case FASTGRNN_NL_TANH: launch_bwd<FASTGRNN_NL_TANH>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    case FASTGRNN_NL_QUANT_TANH: launch_bwd<FASTGRNN_NL_QUANT_TANH>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    case FASTGRNN_NL_QUANT_SIGM: launch_bwd<FASTGRNN_NL_QUANT_SIGM>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
    default: launch_bwd<FASTGRNN_NL_QUANT_SIGM4>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s); break;
  }
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

}  // namespace fastgrnn
