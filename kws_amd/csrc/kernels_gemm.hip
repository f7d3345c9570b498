// Batched split-precision GEMMs around the scans (gfx950): the genuinely dense, non-recurrent contractions of a
// FastGRNN layer whose input or hidden width is too large for the register-resident scans of kernels_split.hip
// (the reference's default stack 32 -> 256 -> 128, trainingConfig.py:12-15; model.py:196-203).
//
//   rows_gemm_split   C[R,N] = A[R,K] . Wt[N,K]^T     R = T*B rows, (N,K) small: the frame GEMM  X . W^T  in front of
//                     the recurrence (.cu:356,368 evaluates it per step) and  d_x = d_pre . W  behind it (.cu:538)
//   tn_gemm_big       C[M,N] = A[R,M]^T . B[R,N]       the weight gradients  dW = d_pre^T X,  dU = d_pre^T H_prev
//                     (.cu:539-540 accumulates them per step) for M, N up to 256
//   tn_gemm_w4        the same product for N = 256 as one wave per SIMD with the operand split between its own MFMAs
//
// Same arithmetic as the scans: every fp32 operand is split exactly into three bf16 planes, six MFMA terms per
// product on v_mfma_f32_16x16x32_bf16, fp32 accumulation, big and small terms in separate accumulators; results
// are fp32 tensors with fp32-level accuracy.  An operand that ARRIVES as bf16 (FASTGRNN_BF16_IO sequences) is its own
// first plane, the other two are zero: it is published from its raw bytes and multiplied with three terms
// (rows_gemm_split<.., BF_IN>, tn_gemm_big<.., BF_B, PURE>).  All kernels are written to the operand rule of
// DESIGN.md 4.0 (fragment reads before the first MFMA of a batch, completion reads before registers are reused) and
// are checked by tools/war_scan.py and tools/lds_branch_vmem_scan.py.
#include "split_common.h"

namespace fastgrnn {
namespace {

// ------------------------------------------------------------------------------------------
// C[R,N] = A[R,K] . Wt[N,K]^T
// ------------------------------------------------------------------------------------------
// Workgroup = 8 waves, persistent over a contiguous range of 32-row stages.  The weights are the MFMA A operand
// (rows = output columns n), resident in registers as planes for the whole launch: wave (wn, wr) owns NPW
// n-tiles and every WR-th 16-row tile of a stage.  A stage of A is loaded with coalesced 16-byte reads, split
// ONCE per workgroup, and published as natural [row][k] plane images (double-buffered); the waves read their B
// fragments (columns = rows of A) from there with conflict-free ds_read_b128.
//   TRANS_W: Wt[n][k] = W[k * N + n] (the caller's matrix is [K,N]: d_x = d_pre . W with W:[H,F]).
//   BF_IN / BF_OUT: A / C are bf16 (FASTGRNN_BF16_IO sequences); arithmetic and W stay fp32.
constexpr int RG_ROWS = 32;                       // rows per stage (two MFMA column tiles)

template <int NT, int KS, bool TRANS_W, bool BF_IN = false, bool BF_OUT = false>
__global__ __launch_bounds__(512) void rows_gemm_split(size_t R, int stages_per_wg, const void* __restrict__ Av,
                                                       const float* __restrict__ W, void* __restrict__ Cv) {
  constexpr int K = 32 * KS, N = 16 * NT;
  constexpr int WN = NT >= 8 ? 8 : NT, WR = 8 / WN, NPW = NT / WN, RT = RG_ROWS / 16;
  constexpr int ROWB = K * 2 + 16;                // bytes per row of a plane image (+16: conflict-free b128 reads)
  constexpr int VPT = RG_ROWS * K / 4 / 512;      // float4 per thread per stage
  constexpr int KB = KS < 4 ? KS : 4;             // K-steps per fragment batch
  static_assert(NT % WN == 0 && VPT >= 1 && KS % KB == 0, "shape");
  constexpr int PA = BF_IN ? 1 : 3;               // planes of A in LDS: a bf16 value is its own first plane, the others zero
  __shared__ __attribute__((aligned(16))) unsigned char pl[2][PA][RG_ROWS * ROWB];

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, i = l & 15, g = l >> 4;
  const int wn = wv % WN, wr = wv / WN;

  // ---- resident weights: planes of Wt rows n = 16 * (wn * NPW + a) + i, K in natural order ------------------
  Frag3 Wf[NPW][KS];
#pragma unroll
  for (int a = 0; a < NPW; ++a) {
    const int n = 16 * (wn * NPW + a) + i;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f32x4 lo, hi;
      if (TRANS_W) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          lo[j] = W[(size_t)(32 * s + 8 * g + j) * N + n];
          hi[j] = W[(size_t)(32 * s + 8 * g + 4 + j) * N + n];
        }
      } else {
        const float* wp = W + (size_t)n * K + 32 * s + 8 * g;
        lo = ld4(wp); hi = ld4(wp + 4);
      }
      Wf[a][s] = split3(lo, hi);
    }
  }

  const size_t nstages = (R + RG_ROWS - 1) / RG_ROWS;
  const size_t s_begin = (size_t)blockIdx.x * stages_per_wg;
  const size_t s_end = (s_begin + stages_per_wg < nstages) ? s_begin + stages_per_wg : nstages;
  if (s_begin >= s_end) return;                   // (whole workgroup: no barrier has been passed)

  // two register sets: a stage's rows are requested two stages ahead (see tn_gemm_big)
  struct Stage {
    f32x4 va[VPT];
    uint2 vraw[VPT];                              // bf16 input: four values per 8 bytes, unpacked at publish time
  };
  auto load_stage = [&](size_t st, Stage& S) __attribute__((always_inline)) {
    f32x4 (&va)[VPT] = S.va; uint2 (&vraw)[VPT] = S.vraw;
    const size_t r0 = st * RG_ROWS;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int idx = tid + 512 * j, row = idx / (K / 4);
      const size_t e = r0 * K + (size_t)idx * 4;  // rows are contiguous: element offset of this thread's four values
      const bool ok = r0 + row < R;
      if (BF_IN) vraw[j] = ok ? *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(Av) + e) : uint2{0u, 0u};
      else va[j] = ok ? ld4(reinterpret_cast<const float*>(Av) + e) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto publish = [&](int buf, const Stage& S) __attribute__((always_inline)) {
    const f32x4 (&va)[VPT] = S.va; const uint2 (&vraw)[VPT] = S.vraw;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int idx = tid + 512 * j, row = idx / (K / 4), c4 = idx % (K / 4);
      const unsigned off = (unsigned)(row * ROWB + c4 * 8);
      if (BF_IN) {                                  // the raw 8 bytes ARE plane 0 (no vector arithmetic at all)
        *reinterpret_cast<uint2*>(&pl[buf][0][off]) = vraw[j];
      } else {
        uint2 q0, q1, q2;
        split_quad(va[j], q0, q1, q2);
        *reinterpret_cast<uint2*>(&pl[buf][0][off]) = q0;
        *reinterpret_cast<uint2*>(&pl[buf][PA > 1 ? 1 : 0][off]) = q1;
        *reinterpret_cast<uint2*>(&pl[buf][PA > 2 ? 2 : 0][off]) = q2;
      }
    }
  };

  Stage S0, S1;
  load_stage(s_begin, S0);
  publish(0, S0);
  __syncthreads();
  if (s_begin + 1 < s_end) load_stage(s_begin + 1, S0);
  if (s_begin + 2 < s_end) load_stage(s_begin + 2, S1);

  auto stage = [&](size_t st, int buf, Stage& Snext) __attribute__((always_inline)) {
    const size_t r0 = st * RG_ROWS;
    // ---- this wave's tiles of the stage ----------------------------------------------------------------
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (WR > 1 && (rt % WR) != wr) continue;     // (wave-uniform) with fewer than eight n-tiles the r-tiles are dealt out
      const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 hi[NPW], lo[NPW];
#pragma unroll
      for (int a = 0; a < NPW; ++a) { hi[a] = z4; lo[a] = z4; }
#pragma unroll
      for (int k0 = 0; k0 < KS; k0 += KB) {
        // all fragment reads of the batch are issued, each into registers of its own, before its first MFMA
        Frag3 Bf[KB];
#pragma unroll
        for (int s = 0; s < KB; ++s)
#pragma unroll
          for (int p = 0; p < PA; ++p)
            Bf[s].p[p] = *reinterpret_cast<const u32x4*>(&pl[buf][p][(rt * 16 + i) * ROWB + (32 * (k0 + s) + 8 * g) * 2]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KB; ++s)
#pragma unroll
          for (int a = 0; a < NPW; ++a) {
            if (BF_IN) {                            // three of mfma6_hl's six terms, in its order (the others multiply zeros)
              lo[a] = mfma_bf16(Wf[a][k0 + s].p[2], Bf[s].p[0], lo[a]);
              lo[a] = mfma_bf16(Wf[a][k0 + s].p[1], Bf[s].p[0], lo[a]);
              hi[a] = mfma_bf16(Wf[a][k0 + s].p[0], Bf[s].p[0], hi[a]);
            } else {
              mfma6_hl(Wf[a][k0 + s], Bf[s], hi[a], lo[a]);
            }
          }
        __builtin_amdgcn_sched_barrier(0);        // (the scheduler otherwise sinks MFMAs below the read)
        float touch = 0.f;                         // every accumulator of the batch has retired before Bf is reloaded
#pragma unroll
        for (int a = 0; a < NPW; ++a) touch += hi[a][0] + lo[a][0];
        completion_read(touch);
        __builtin_amdgcn_sched_barrier(0);
      }
      const size_t r = r0 + rt * 16 + i;
      if (r < R) {
#pragma unroll
        for (int a = 0; a < NPW; ++a) {
          const f32x4 o = hi[a] + lo[a];
          const size_t e = r * N + 16 * (wn * NPW + a) + 4 * g;
          if (BF_OUT) st4_bf16(reinterpret_cast<unsigned short*>(Cv) + e, o);
          else st4(reinterpret_cast<float*>(Cv) + e, o);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- next stage: planes into the other buffer, then the request for the stage after it -----------------
    if (st + 1 < s_end) publish(buf ^ 1, Snext);
    lds_barrier();
    if (st + 3 < s_end) load_stage(st + 3, Snext);
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    size_t st = s_begin;
    for (; st + 1 < s_end; st += 2) {
      stage(st, 0, S0);
      stage(st + 1, 1, S1);
    }
    if (st < s_end) stage(st, 0, S0);
  }
  {                                               // the weight fragments stay allocated through the last stage
    float probe = 0.f;
#pragma unroll
    for (int a = 0; a < NPW; ++a)
#pragma unroll
      for (int s = 0; s < KS; ++s) keep_alive(probe, Wf[a][s]);
  }
}

template <int NT, int KS, bool TRANS_W>
void launch_rows_gemm(size_t R, const void* A, const float* W, void* C, bool bf_in, bool bf_out, hipStream_t s) {
  const size_t nstages = (R + RG_ROWS - 1) / RG_ROWS;
  const int nwg = (int)(nstages < 256 ? nstages : 256);                  // one workgroup per CU (LDS: 50-100 KB)
  const int spw = (int)((nstages + nwg - 1) / nwg);
  if (bf_in && bf_out) hipLaunchKernelGGL((rows_gemm_split<NT, KS, TRANS_W, true, true>), dim3(nwg), dim3(512), 0, s, R, spw, A, W, C);
  else if (bf_in) hipLaunchKernelGGL((rows_gemm_split<NT, KS, TRANS_W, true, false>), dim3(nwg), dim3(512), 0, s, R, spw, A, W, C);
  else if (bf_out) hipLaunchKernelGGL((rows_gemm_split<NT, KS, TRANS_W, false, true>), dim3(nwg), dim3(512), 0, s, R, spw, A, W, C);
  else hipLaunchKernelGGL((rows_gemm_split<NT, KS, TRANS_W, false, false>), dim3(nwg), dim3(512), 0, s, R, spw, A, W, C);
}

// ------------------------------------------------------------------------------------------
// C[M,N] = A[R,M]^T . B[R,N]   (M = 128 per workgroup row block, N = 16 * NT <= 256)
// ------------------------------------------------------------------------------------------
// Workgroup = 8 waves = one chunk of rows x one 128-column block of A.  Stages of 32 rows (one MFMA
// K-step): global fp32 -> three exact bf16 planes in LDS in natural [row][column] order (double-buffered, the
// next stage's loads in flight under the MFMAs) -> hardware-transposed fragment reads (K = rows) -> 6-term MFMAs
// into register accumulators.  A wave owns MA m-tiles x NT/MA n-tiles (4 x NT/4 for NT >= 8, else 2 x NT/2): its A
// fragments are read once per stage, its B fragments in batches.  Each workgroup leaves its partial C in the
// workspace; tn_big_reduce sums them in a fixed order (deterministic, no atomics).
// Rows of B below shiftB come from B0 (h0: H_prev of step 0), the rest from B1 shifted down by shiftB rows.
constexpr int TNB_STAGE = 32;
// Workgroup -> (row chunk, 128-column block of A).  The column blocks of ONE row chunk read the same rows of B: they
// are given to workgroups 8 apart, which the dispatcher places on the same XCD (workgroups go round-robin over the 8
// XCDs) in the same round, so the second read of those rows is a hit in that XCD's L2 and never reaches the fabric.
// (Round 2 had them on neighbouring workgroups = neighbouring XCDs: the second read came from the Infinity Cache,
// and the kernel sat at the ~5 TB/s the fabric delivers -- 1.25 GB per launch for 0.83 GB of operands.)
__device__ __forceinline__ int tnb_chunk(int bid, int nblk) { return ((bid >> 3) / nblk) * 8 + (bid & 7); }
__device__ __forceinline__ int tnb_mblk(int bid, int nblk) { return (bid >> 3) % nblk; }
#ifndef TNW4_PRE
#define TNW4_PRE 0                                // tn_gemm_w4: value pairs split before a stage's first MFMA (A/B knob)
#endif

// PERIODIC: row r of B is B0[r / shiftB] where r is a multiple of shiftB, else B1[r - 1] (H_prev of batch-major
// sequences: shiftB = T; the first step of every utterance reads h0, the others the row before their own).
// BF_B: B1 holds bf16 (FASTGRNN_BF16_IO sequences: x, hs), B0 (h0) stays fp32.  A bf16 value IS its own first plane
// (the other two are zero): its rows are published from the raw 8 bytes with no vector arithmetic.
// PURE (with BF_B): EVERY row of B is bf16 (no h0 rows: shiftB = 0) -- planes 1 and 2 of B are zero, are neither
// published nor read, and a product is three of mfma6's six terms, in its order (same bits as the six).
template <int NT, bool PERIODIC = false, bool BF_B = false, bool PURE = false>
__global__ __launch_bounds__(512) void tn_gemm_big(size_t R, int stages_per_wg, int nblk, int nchunk, const float* __restrict__ A, int lda,
                                                   const float* __restrict__ B0, const float* __restrict__ B1,
                                                   size_t shiftB, int ldb, float* __restrict__ part) {
  // wave tiling of the 8 x NT output tiles: MA m-tiles x NH n-tiles per wave.  4 x NT/4 where NT allows it (per stage
  // 4 A + NT/4 B fragment sets per wave instead of 2 + NT/2: a fifth fewer LDS reads at NT = 16, the kernel's
  // co-bottleneck), else 2 x NT/2
  constexpr int MB = 128, N = 16 * NT, MA = NT >= 8 ? 4 : 2, NWM = 8 / MA, NH = NT / MA;
  constexpr int NBATCH = MA == 4 ? (NH < 2 ? NH : 2) : (NH < 4 ? NH : 4);
  constexpr int ROWA = MB * 2 + 32, ROWB = N * 2 + 32;
  constexpr int VA = TNB_STAGE * MB / 4 / 512, VB = (TNB_STAGE * N / 4 + 511) / 512;   // float4 per thread per stage
  static_assert(NT % MA == 0 && NH % NBATCH == 0 && VA >= 1 && (!PURE || (BF_B && !PERIODIC)), "shape");
  constexpr int PB = PURE ? 1 : 3;                   // planes of B in LDS
  __shared__ __attribute__((aligned(16))) unsigned char la[2][3][TNB_STAGE * ROWA];
  __shared__ __attribute__((aligned(16))) unsigned char lb[2][PB][TNB_STAGE * ROWB];

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, g = l >> 4, q = (l & 15) >> 2, pp = l & 3;
  const int mq = wv % NWM, nh = wv / NWM;
#ifdef FASTGRNN_DIAG_STAMPS
  unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dlast)::"memory");
#endif
  const int mblk = tnb_mblk(blockIdx.x, nblk), chunk = tnb_chunk(blockIdx.x, nblk);
  if (chunk >= nchunk) return;                     // (grid padded to whole rounds of the XCDs; no barrier passed yet)
  const size_t nstages = (R + TNB_STAGE - 1) / TNB_STAGE;
  const size_t s_begin = (size_t)chunk * stages_per_wg;
  const size_t s_end = (s_begin + stages_per_wg < nstages) ? s_begin + stages_per_wg : nstages;

  // Two register sets: the rows of stage st + 2 are requested while stage st + 1 still waits in the other set, so a
  // request is in flight for two iterations (one was not enough to keep HBM busy: the kernel ran at 2.9-3.7 TB/s
  // with one workgroup's single stage in flight per CU).
  struct Stage { f32x4 va[VA], vb[VB]; };
  auto load_stage = [&](size_t st, Stage& S) __attribute__((always_inline)) {
    f32x4 (&va)[VA] = S.va; f32x4 (&vb)[VB] = S.vb;
    const size_t r0 = st * TNB_STAGE;
#pragma unroll
    for (int j = 0; j < VA; ++j) {
      const int idx = tid + 512 * j, row = idx / (MB / 4), c4 = idx % (MB / 4);
      const size_t r = r0 + row;
      va[j] = r < R ? ld4(A + r * lda + mblk * MB + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < VB; ++j) {
      const int idx = tid + 512 * j, row = idx / (N / 4), c4 = idx % (N / 4);
      const size_t r = r0 + row;
      bool from0;                                    // the row comes from B0 (h0)
      size_t e1;                                     // ... else element offset of the row in B1
      if (PERIODIC) {                                // (R < 2^32: the workspace that holds A is below 4 GB)
        const unsigned ru = (unsigned)r, qd = ru / (unsigned)shiftB;
        from0 = qd * (unsigned)shiftB == ru;
        e1 = from0 ? (size_t)qd * ldb : (size_t)(ru - 1u) * ldb;
      } else {
        from0 = r < shiftB;                          // H_prev: rows of t = 0 are h0
        e1 = from0 ? r * (size_t)ldb : (r - shiftB) * (size_t)ldb;
      }
      const bool ok = idx < TNB_STAGE * N / 4 && r < R;
      if (BF_B) {
        // (raw bits; unpacked when the stage is published -- converting here would wait for the load at once)
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!PURE && ok && from0) v = ld4(B0 + e1 + 4 * c4);
        if (ok && !from0) {
          const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(B1) + e1 + 4 * c4);
          v = f32x4{bitsf(raw.x), bitsf(raw.y), 0.f, 0.f};
        }
        vb[j] = v;
      } else {
        vb[j] = ok ? ld4((from0 ? B0 : B1) + e1 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto row_is_bf16 = [&](size_t r) __attribute__((always_inline)) {   // (BF_B) row r of B was loaded as raw bf16
    if (PERIODIC) { const unsigned ru = (unsigned)r; return (ru / (unsigned)shiftB) * (unsigned)shiftB != ru; }
    return r >= shiftB;
  };
  auto put = [&](unsigned char* p0, int plane_bytes, unsigned off, const f32x4 v) __attribute__((always_inline)) {
    uint2 q0, q1, q2;
    split_quad(v, q0, q1, q2);
    *reinterpret_cast<uint2*>(p0 + off) = q0;
    *reinterpret_cast<uint2*>(p0 + plane_bytes + off) = q1;
    *reinterpret_cast<uint2*>(p0 + 2 * plane_bytes + off) = q2;
  };
  auto publish = [&](int buf, const Stage& S, size_t st) __attribute__((always_inline)) {   // st: the stage S holds
    const f32x4 (&va)[VA] = S.va; const f32x4 (&vb)[VB] = S.vb;
#pragma unroll
    for (int j = 0; j < VA; ++j) {
      const int idx = tid + 512 * j, row = idx / (MB / 4), c4 = idx % (MB / 4);
      put(&la[buf][0][0], TNB_STAGE * ROWA, (unsigned)(row * ROWA + c4 * 8), va[j]);
    }
#pragma unroll
    for (int j = 0; j < VB; ++j) {
      const int idx = tid + 512 * j, row = idx / (N / 4), c4 = idx % (N / 4);
      if (idx < TNB_STAGE * N / 4) {
        const unsigned off = (unsigned)(row * ROWB + c4 * 8);
        if (PURE) {
          *reinterpret_cast<uint2*>(&lb[buf][0][0] + off) = uint2{fbits(vb[j][0]), fbits(vb[j][1])};
        } else if (BF_B && row_is_bf16(st * TNB_STAGE + row)) {
          unsigned char* p0 = &lb[buf][0][0] + off;
          *reinterpret_cast<uint2*>(p0) = uint2{fbits(vb[j][0]), fbits(vb[j][1])};
          *reinterpret_cast<uint2*>(p0 + TNB_STAGE * ROWB) = uint2{0u, 0u};
          *reinterpret_cast<uint2*>(p0 + 2 * TNB_STAGE * ROWB) = uint2{0u, 0u};
        } else {
          put(&lb[buf][0][0], TNB_STAGE * ROWB, off, vb[j]);
        }
      }
    }
  };

  f32x4 acc[MA][NH];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NH; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (s_begin < s_end) {
    const unsigned la0 = (unsigned)(size_t)&la[0][0][0], lb0 = (unsigned)(size_t)&lb[0][0][0];
    // transposed fragment of this lane: rows 8g + q (+4) of the stage, 4 columns at 4*pp of a 16-column tile
    const unsigned trA = la0 + (8 * g + q) * ROWA + (mq * MA * 16 + 4 * pp) * 2;
    const unsigned trB = lb0 + (8 * g + q) * ROWB + (nh * NH * 16 + 4 * pp) * 2;
    Stage S0, S1;
    load_stage(s_begin, S0);
    publish(0, S0, s_begin);
    __syncthreads();
    if (s_begin + 1 < s_end) load_stage(s_begin + 1, S0);
    if (s_begin + 2 < s_end) load_stage(s_begin + 2, S1);
    // one stage: products from buffer buf, then the planes of stage st + 1 (waiting in Snext) into buffer buf ^ 1,
    // then -- behind the barrier, i.e. behind every MFMA of the stage -- the request for stage st + 3 into Snext
    auto stage = [&](size_t st, int buf, Stage& Snext) __attribute__((always_inline)) {
      const unsigned oa = (unsigned)buf * 3 * TNB_STAGE * ROWA, ob = (unsigned)buf * PB * TNB_STAGE * ROWB;
      SPLIT_STAMP(0)
      Frag3 Af[MA];
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int p = 0; p < 3; ++p) Af[a].p[p] = tr_frag(trA + oa + p * (TNB_STAGE * ROWA) + a * 32, ROWA);
#pragma unroll
      for (int c0 = 0; c0 < NH; c0 += NBATCH) {
        Frag3 Bf[NBATCH];
#pragma unroll
        for (int c = 0; c < NBATCH; ++c)
#pragma unroll
          for (int p = 0; p < PB; ++p) Bf[c].p[p] = tr_frag(trB + ob + p * (TNB_STAGE * ROWB) + (c0 + c) * 32, ROWB);
        __builtin_amdgcn_sched_barrier(0);        // every fragment read of the batch is issued before its first MFMA
        SPLIT_STAMP(1)                            // (diagnostic build: + the wait for the fragments)
#pragma unroll
        for (int c = 0; c < NBATCH; ++c)
#pragma unroll
          for (int a = 0; a < MA; ++a) {
            if (PURE) {
              f32x4 v = mfma_bf16(Af[a].p[2], Bf[c].p[0], acc[a][c0 + c]);
              v = mfma_bf16(Af[a].p[1], Bf[c].p[0], v);
              acc[a][c0 + c] = mfma_bf16(Af[a].p[0], Bf[c].p[0], v);
            } else {
              acc[a][c0 + c] = mfma6(Af[a], Bf[c], acc[a][c0 + c]);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
        float touch = 0.f;                         // all of the batch's MFMAs have retired before Bf / Af are reloaded
#pragma unroll
        for (int c = 0; c < NBATCH; ++c)
#pragma unroll
          for (int a = 0; a < MA; ++a) touch += acc[a][c0 + c][0];
        completion_read(touch);
        __builtin_amdgcn_sched_barrier(0);
        SPLIT_STAMP(2)
      }
      // (tried, same-run A/B: half of the waves publishing BEFORE their products, so that one wave's VALU split runs
      // beside its SIMD partner's MFMAs -- for either pairing of the waves 9-16 % slower: 281 -> 309 us at N = 256)
      // (tried: the planes first and the requests in portions behind each batch's MFMAs, because cycle stamps --
      // tools/diag_gemm.hip -- put 1 300-1 800 of a stage's 6 800 stamped cycles into issuing the six loads right
      // behind the barrier: 276 -> 291 us in the same run)
      if (st + 1 < s_end) publish(buf ^ 1, Snext, st + 1);
      SPLIT_STAMP(3)
      lds_barrier();
      SPLIT_STAMP(4)
      if (st + 3 < s_end) load_stage(st + 3, Snext);
      __builtin_amdgcn_sched_barrier(0);
    };
    size_t st = s_begin;
    for (; st + 1 < s_end; st += 2) {
      stage(st, 0, S0);
      stage(st + 1, 1, S1);
    }
    if (st < s_end) stage(st, 0, S0);
  }
#ifdef FASTGRNN_DIAG_STAMPS
  if (blockIdx.x == 7 && l == 0) { for (int k = 0; k < 8; ++k) g_sdiag[wv][k] = dsum[k]; }
#endif
  // D row 4g + r of tile (mt, nt) is m = 16 mt + 4g + r, column n = 16 nt + (l & 15)
  float* pc = part + ((size_t)chunk * nblk + mblk) * MB * N;
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int c = 0; c < NH; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pc[(size_t)((mq * MA + a) * 16 + 4 * g + r) * N + (nh * NH + c) * 16 + (l & 15)] = acc[a][c][r];
}

// ------------------------------------------------------------------------------------------
// tn_gemm_w4 (N = 256): ONE wave per SIMD (workgroup = 4 waves), each 64 x 128 of the block (128 accumulator
// registers in the AGPR half of the 512 a lone wave has), 192 MFMAs per stage with the split of its quarter of the NEXT
// stage between them.  Why this shape (same-run A/B and cycle stamps, tools/gemm_ab_bench.hip):
//   * tn_gemm_big's stage is 6 200 ticks for 3 072 MFMA cycles per SIMD: its two waves per SIMD do requests, fragment
//     reads, products, split and publication one after the other and in lockstep.
//   * Splitting the waves by ROLE (one MFMA wave + one load/split wave per SIMD; built, bit-identical, removed) is no
//     faster: beside a wave that always has an MFMA ready the partner's vector instructions are issued at ~1 per 20
//     ticks -- 5 100 ticks for the 252 of a stage's split, 67 of them waiting for data.  The MFMA wave alone runs a
//     stage in 4 100 ticks (3 400 for the MFMAs = 17.7 each, 490 for the first fragment reads, 210 at the barrier).
//   * Vector work between the MFMAs of the SAME wave is not free either (tools/mfma_shadow_probe.hip: +2.9 ticks per
//     fp32 add / mul / fma behind a bf16 MFMA, ~0 for ONE integer or convert instruction, +3-4 for the second), but it
//     costs about half of what it costs in a partner wave: 4 550 ticks for 192 MFMAs + 298 vector instructions.
//   * v_dot2c_f32_bf16 for the residuals (7 instead of 11 instructions per value pair) is slower still (5 800): it
//     competes with the MFMAs for the matrix pipe.  Splitting part of the next stage while the stage's first fragment
//     reads are under way (PRE > 0) moves time from one segment to the other, no more.
// Each 16-byte piece is re-requested for the stage after next (buffer load: descriptor in scalar registers, ONE
// loop-invariant lane offset per operand, rows beyond R cut off by the descriptor's size -- no vector instruction) as
// soon as it has been split: one stage of rows (48 KB per CU) is in flight at all times, every request has a whole
// stage period to land, and the staging registers are 48.  Same sums in the same order as tn_gemm_big: bit-identical.
// Requires shiftB to be 0, >= R or a multiple of the stage (no stage takes rows from both B0 and B1): the launcher
// gives other shapes to tn_gemm_big.
template <int NT, int PRE>
__global__ __launch_bounds__(256) ONE_WAVE_PER_SIMD void tn_gemm_w4(size_t R, int stages_per_wg, int nblk, int nchunk, const float* __restrict__ A, int lda,
                                                  const float* __restrict__ B0, const float* __restrict__ B1,
                                                  size_t shiftB, int ldb, float* __restrict__ part) {
  constexpr int MB = 128, N = 16 * NT, NH = NT / 2, NP = 256;
  constexpr int ROWA = MB * 2 + 32, ROWB = N * 2 + 32;
  constexpr int VA = TNB_STAGE * MB / 4 / NP, VB = TNB_STAGE * N / 4 / NP, VQ = VA + VB;   // 16-byte pieces per lane and stage
  constexpr int PAIRS = 2 * VQ, PPT = (PAIRS - PRE) / NH;          // value pairs per lane and stage, per B tile
  static_assert(NT % 2 == 0 && NH >= 3 && (PAIRS - PRE) % NH == 0 && PRE % 2 == 0 && (TNB_STAGE * N / 4) % NP == 0 && (TNB_STAGE * MB / 4) % NP == 0, "shape");
  __shared__ __attribute__((aligned(16))) unsigned char la[2][3][TNB_STAGE * ROWA];
  __shared__ __attribute__((aligned(16))) unsigned char lb[2][3][TNB_STAGE * ROWB];

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, g = l >> 4, q = (l & 15) >> 2, pp = l & 3;
  const int mq = wv & 1, nh = wv >> 1;
  const int mblk = tnb_mblk(blockIdx.x, nblk), chunk = tnb_chunk(blockIdx.x, nblk);
  if (chunk >= nchunk) return;
  const size_t nstages = (R + TNB_STAGE - 1) / TNB_STAGE;
  const size_t s_begin = (size_t)chunk * stages_per_wg;
  const size_t s_end = (s_begin + stages_per_wg < nstages) ? s_begin + stages_per_wg : nstages;
  const int n = s_begin < s_end ? (int)(s_end - s_begin) : 0;
#ifdef FASTGRNN_DIAG_STAMPS
  unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dlast)::"memory");
#endif

  // ---- requests: buffer loads.  A stage's rows of one operand are one resource descriptor -- base = the stage's first
  // row, size = its valid rows, so rows beyond R read as zero without a comparison -- built in scalar registers; a lane
  // keeps ONE loop-invariant byte offset per operand and piece j adds a scalar offset.
  f32x4 sv[VQ];                                      // sv[0..VA) pieces of A, sv[VA..VQ) pieces of B
  const unsigned va_off = (unsigned)(tid / (MB / 4)) * (unsigned)lda * 4u + (unsigned)(tid % (MB / 4)) * 16u;
  const unsigned vb_off = (unsigned)(tid / (N / 4)) * (unsigned)ldb * 4u + (unsigned)(tid % (N / 4)) * 16u;
  struct Desc { __amdgpu_buffer_rsrc_t a, b; };
  auto desc_of = [&](int i) __attribute__((always_inline)) {   // stage s_begin + min(i, n - 1)
    const size_t r0 = (s_begin + (size_t)(i < n ? i : n - 1)) * TNB_STAGE;
    const unsigned rows = R - r0 < (size_t)TNB_STAGE ? (unsigned)(R - r0) : (unsigned)TNB_STAGE;   // >= 1
    const float* bb = r0 < shiftB ? B0 + r0 * (size_t)ldb : B1 + (r0 - shiftB) * (size_t)ldb;
    Desc d;
    d.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + r0 * (size_t)lda + mblk * MB), 0,
                                            (int)((rows - 1) * (unsigned)lda * 4u + MB * 4u), 0x00020000);
    d.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bb), 0, (int)((rows - 1) * (unsigned)ldb * 4u + N * 4u), 0x00020000);
    return d;
  };
  auto request = [&](const Desc& d, int j) __attribute__((always_inline)) {
    if (j < VA)
      sv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d.a, (int)va_off, (int)((unsigned)(j * (NP / (MB / 4))) * (unsigned)lda * 4u), 0));
    else
      sv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d.b, (int)vb_off, (int)((unsigned)((j - VA) * (NP / (N / 4))) * (unsigned)ldb * 4u), 0));
  };
  // LDS address of piece j in plane 0 of buffer 0
  const unsigned la0 = (unsigned)(size_t)&la[0][0][0], lb0 = (unsigned)(size_t)&lb[0][0][0];
  const unsigned pa_off = (unsigned)(tid / (MB / 4)) * ROWA + (unsigned)(tid % (MB / 4)) * 8u;
  const unsigned pb_off = (unsigned)(tid / (N / 4)) * ROWB + (unsigned)(tid % (N / 4)) * 8u;
  // (buf: the lane's plane-0 addresses of pieces 0 in the buffer written to)
  auto put_piece = [&](unsigned char* pa, unsigned char* pb, int j, const uint2 q0, const uint2 q1, const uint2 q2) __attribute__((always_inline)) {
    unsigned char* p0;
    int plane;
    if (j < VA) { p0 = pa + j * (NP / (MB / 4)) * ROWA; plane = TNB_STAGE * ROWA; }
    else { p0 = pb + (j - VA) * (NP / (N / 4)) * ROWB; plane = TNB_STAGE * ROWB; }
    *reinterpret_cast<uint2*>(p0) = q0;
    *reinterpret_cast<uint2*>(p0 + plane) = q1;
    *reinterpret_cast<uint2*>(p0 + 2 * plane) = q2;
  };
  constexpr unsigned BUFA = 3 * TNB_STAGE * ROWA, BUFB = 3 * TNB_STAGE * ROWB;   // bytes per buffer

  f32x4 acc[4][NH];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < NH; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned trA = la0 + (8 * g + q) * ROWA + (mq * 64 + 4 * pp) * 2;
  const unsigned trB = lb0 + (8 * g + q) * ROWB + (nh * NH * 16 + 4 * pp) * 2;

  if (n > 0) {
    const Desc d0 = desc_of(0), d1 = desc_of(1);
#pragma unroll
    for (int j = 0; j < VQ; ++j) request(d0, j);
#pragma unroll
    for (int j = 0; j < VQ; ++j) {
      uint2 q0, q1, q2;
      split_quad(sv[j], q0, q1, q2);
      put_piece(&la[0][0][0] + pa_off, &lb[0][0][0] + pb_off, j, q0, q1, q2);
    }
#pragma unroll
    for (int j = 0; j < VQ; ++j) request(d1, j);
    lds_barrier();
  }
  // stage i: products from buffer i & 1; between them the planes of stage i + 1 (waiting in sv) into the other buffer,
  // each piece re-requested for stage i + 2 as soon as it is split.  Requests and plane writes are unconditional (the
  // last stages re-request / re-publish the last stage): no branch between LDS writes and loads (second rule of
  // DESIGN.md 4.0).
  for (int i = 0; i < n; ++i) {
    const unsigned par = (unsigned)(i & 1);
    const unsigned oa = par * BUFA, ob = par * BUFB;
    unsigned char* const wpa = &la[0][0][0] + pa_off + (par ^ 1u) * BUFA;
    unsigned char* const wpb = &lb[0][0][0] + pb_off + (par ^ 1u) * BUFB;
    SPLIT_STAMP(0)
    const Desc dn = desc_of(i + 2);
    Frag3 Af[4], Bf[NH];
    auto read_b = [&](int c, Frag3& F) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 3; ++p) F.p[p] = tr_frag(trB + ob + p * (TNB_STAGE * ROWB) + c * 32, ROWB);
    };
    read_b(0, Bf[0]);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int p = 0; p < 3; ++p) Af[a].p[p] = tr_frag(trA + oa + p * (TNB_STAGE * ROWA) + a * 32, ROWA);
    __builtin_amdgcn_sched_barrier(0);
    uint2 q0, q1, q2;
    auto split_one_pair = [&](int pr) __attribute__((always_inline)) {   // pair pr = values 2 * (pr & 1) .. + 1 of piece pr / 2
      const int j = pr >> 1;
      if (pr & 1) {
        split_pair(sv[j][2], sv[j][3], q0.y, q1.y, q2.y);
        asm volatile("" : "+v"(q0.y), "+v"(q1.y), "+v"(q2.y));
        put_piece(wpa, wpb, j, q0, q1, q2);
        request(dn, j);
      } else {
        split_pair(sv[j][0], sv[j][1], q0.x, q1.x, q2.x);
        asm volatile("" : "+v"(q0.x), "+v"(q1.x), "+v"(q2.x));
      }
    };
    // the first PRE pairs while the fragment reads above are under way (the matrix pipe has nothing to do until they land)
#pragma unroll
    for (int pr = 0; pr < PRE; ++pr) split_one_pair(pr);
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STAMP(1)
    constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};   // mfma6's terms in its order
#pragma unroll
    for (int c = 0; c < NH; ++c) {
      if (c + 1 < NH) {                              // tile c + 1's B fragments, one tile ahead.  Every tile of the stage
        read_b(c + 1, Bf[c + 1]);                    // has registers of its own (a lone wave has them): no fragment
        __builtin_amdgcn_sched_barrier(0);           // register is written while an MFMA of the stage may still read it
      }
      // 24 MFMAs (four m-tiles interleaved: consecutive ones are independent) and PPT value pairs of the next stage
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][c] = mfma_bf16(Af[a].p[TA[t]], Bf[c].p[TB[t]], acc[a][c]);
#pragma unroll
      for (int k = 0; k < PPT; ++k) split_one_pair(PRE + c * PPT + k);
#pragma unroll
      for (int k = 0; k < 24; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (PPT >= 3 || k % 4 != 3) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        if (k % 8 == 7) {
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    {                                                // every MFMA of the stage has retired before its fragments are replaced
      float touch = 0.f;                             // (the last tile's four chains: the scheduler orders them freely; a
#pragma unroll                                       // scheduling barrier separates the tiles)
      for (int a = 0; a < 4; ++a) touch += acc[a][NH - 1][0];
      completion_read(touch);
#pragma unroll
      for (int c = 0; c < NH; ++c) asm volatile("" :: "v"(Bf[c].p[0]), "v"(Bf[c].p[1]), "v"(Bf[c].p[2]));
#pragma unroll
      for (int a = 0; a < 4; ++a) asm volatile("" :: "v"(Af[a].p[0]), "v"(Af[a].p[1]), "v"(Af[a].p[2]));
    }
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_STAMP(2)
    lds_barrier();                                   // stage i consumed, stage i + 1 published
  }
#ifdef FASTGRNN_DIAG_STAMPS
  if (blockIdx.x == 7 && l == 0) { for (int k = 0; k < 8; ++k) g_sdiag[wv][k] = dsum[k]; }
#endif
  float* pc = part + ((size_t)chunk * nblk + mblk) * MB * N;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < NH; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pc[(size_t)((mq * 4 + a) * 16 + 4 * g + r) * N + (nh * NH + c) * 16 + (l & 15)] = acc[a][c][r];
}

// C[(mblk*128 + m) * ldc + n] = sum over row chunks of part[chunk][mblk][m][n], fixed order
__global__ __launch_bounds__(1024) void tn_big_reduce(int nchunk, int nblk, int N, const float* __restrict__ part,
                                                      float* __restrict__ C, int ldc) {
  __shared__ float sm[16][64];
  const int o = threadIdx.x & 63, pid = threadIdx.x >> 6;
  const int per_blk = 128 * N, total = nblk * per_blk;
  const int idx = blockIdx.x * 64 + o;
  float a = 0.f;
  if (idx < total) {
    const int blk = idx / per_blk, e = idx - blk * per_blk;
    for (int c0 = pid; c0 < nchunk; c0 += 64) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + 16 * j;
        v[j] = c < nchunk ? part[((size_t)c * nblk + blk) * per_blk + e] : 0.f;
      }
      a += (v[0] + v[1]) + (v[2] + v[3]);
    }
  }
  sm[pid][o] = a;
  __syncthreads();
  if (pid == 0 && idx < total) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += sm[j][o];
    const int blk = idx / per_blk, e = idx - blk * per_blk, m = e / N, n = e - m * N;
    C[(size_t)(blk * 128 + m) * ldc + n] = t;
  }
}

}  // namespace

// ---- launchers (declared in common.h) -------------------------------------------------------------------------
bool rows_gemm_supported(int N, int K, bool trans_w) {
  // N * K <= 128 * 256: the weight planes of a wave's tiles (N * K * 12 / 512 registers per lane) fit beside the
  // fragments; N = 256 with K = 256 would spill (and spill reloads are loads: operand rule)
  const bool k_ok = K == 64 || K == 128 || K == 256;
  if (!k_ok || N * K > 128 * 256) return false;
  if (!trans_w) return N == 128 || N == 256;
  return N == 32 || N == 64 || N == 128 || N == 256;
}

int rows_gemm(size_t R, int N, int K, bool trans_w, const void* A, const float* W, void* C, bool bf_in, bool bf_out,
              hipStream_t s) {
#define RG_CASE(n, k)                                                                     \
  if (N == n && K == k) {                                                                 \
    if (trans_w) launch_rows_gemm<n / 16, k / 32, true>(R, A, W, C, bf_in, bf_out, s);    \
    else launch_rows_gemm<n / 16, k / 32, false>(R, A, W, C, bf_in, bf_out, s);           \
    return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;           \
  }
  if (!rows_gemm_supported(N, K, trans_w)) return FASTGRNN_ERR_UNSUPPORTED;
  RG_CASE(128, 64) RG_CASE(128, 128) RG_CASE(128, 256)
  RG_CASE(256, 64) RG_CASE(256, 128)
  RG_CASE(32, 64) RG_CASE(32, 128) RG_CASE(32, 256) RG_CASE(64, 64) RG_CASE(64, 128) RG_CASE(64, 256)
#undef RG_CASE
  return FASTGRNN_ERR_UNSUPPORTED;
}

static inline int tnb_chunks(size_t R, int nblk, int* spw) {
  const size_t nstages = (R + TNB_STAGE - 1) / TNB_STAGE;
  size_t want = 256 / (size_t)nblk;                // one workgroup per CU over the whole grid
  if (want < 1) want = 1;
  if (want > nstages) want = nstages;
  *spw = (int)((nstages + want - 1) / want);
  return (int)((nstages + *spw - 1) / *spw);
}

int tn_gemm_big_run_periodic(size_t R, int M, int N, const float* A, int lda, const float* B0, const void* B1, size_t period,
                             int ldb, float* part, float* C, int ldc, hipStream_t s, bool bf_b) {
  // (bf16 rows with the periodic mapping: the instantiation spills inside its loop -- spill reloads are loads, operand
  // rule -- and is not built; h256_supported keeps batch-major bf16 backwards off this path)
  if (N != 256 || (M != 128 && M != 256) || period == 0 || R >= ((size_t)1 << 32) || bf_b) return FASTGRNN_ERR_UNSUPPORTED;
  int spw;
  const int nblk = M / 128, nch = tnb_chunks(R, nblk, &spw);
  dim3 grid(((nch + 7) / 8) * 8 * nblk);
  hipLaunchKernelGGL((tn_gemm_big<16, true>), grid, dim3(512), 0, s, R, spw, nblk, nch, A, lda, B0, (const float*)B1, period, ldb, part);
  hipLaunchKernelGGL(tn_big_reduce, dim3((M * N + 63) / 64), dim3(1024), 0, s, nch, nblk, N, (const float*)part, C, ldc);
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

bool tn_gemm_big_supported(int M, int N) { return (M == 128 || M == 256) && (N == 32 || N == 64 || N == 128 || N == 256); }

size_t tn_gemm_big_ws(size_t R, int M, int N) {
  int spw;
  const int nblk = M / 128, nch = tnb_chunks(R, nblk, &spw);
  // (twice the chunks of the product: with bf16 rows of B the rows that pair with fp32 h0 are a product of their own,
  // its partials behind the others')
  return align256((size_t)2 * nch * nblk * 128 * N * sizeof(float));
}

int tn_gemm_big_run(size_t R, int M, int N, const float* A, int lda, const float* B0, const void* B1v, size_t shiftB,
                    int ldb, float* part, float* C, int ldc, hipStream_t s, bool bf_b) {
  if (!tn_gemm_big_supported(M, N)) return FASTGRNN_ERR_UNSUPPORTED;
  int spw;
  const int nblk = M / 128;
  int nch = tnb_chunks(R, nblk, &spw);
  dim3 grid(((nch + 7) / 8) * 8 * nblk);
  const float* B1 = (const float*)B1v;
  // bf16 rows of B: with shiftB = 0 every row is bf16 (the PURE variant: three MFMA terms); otherwise the first
  // shiftB rows pair with fp32 rows of B0 (h0) -- a product of its own over those rows, its partials behind the
  // others' in the workspace, one reduction over both
  int nch0 = 0;                                      // chunks of the fp32 head product
  if (bf_b && shiftB > 0 && shiftB < R) {
    const size_t R0 = shiftB, Rb = R - shiftB;
    // the bf16 body: rows shiftB .. R-1 of A against rows 0 .. of B1
    nch = tnb_chunks(Rb, nblk, &spw);
    grid = dim3(((nch + 7) / 8) * 8 * nblk);
    int spw0;
    nch0 = tnb_chunks(R0, nblk, &spw0);
    float* part0 = part + (size_t)nch * nblk * 128 * N;      // (behind the body's partials: one reduction over both)
    dim3 grid0(((nch0 + 7) / 8) * 8 * nblk);
#define TNB_HEAD(n) \
    if (N == n) hipLaunchKernelGGL((tn_gemm_big<n / 16>), grid0, dim3(512), 0, s, R0, spw0, nblk, nch0, A, lda, B0, B0, R0, ldb, part0);
    TNB_HEAD(32) TNB_HEAD(64) TNB_HEAD(128) TNB_HEAD(256)
#undef TNB_HEAD
    A += R0 * (size_t)lda;
    R = Rb;
    shiftB = 0;
  }
  const bool pure = bf_b && shiftB == 0;
#define TNB_CASE(n)                                                                                                               \
  if (N == n) {                                                                                                                   \
    if (pure) hipLaunchKernelGGL((tn_gemm_big<n / 16, false, true, true>), grid, dim3(512), 0, s, R, spw, nblk, nch, A, lda, B0, B1, shiftB, ldb, part); \
    else if (bf_b) hipLaunchKernelGGL((tn_gemm_big<n / 16, false, true>), grid, dim3(512), 0, s, R, spw, nblk, nch, A, lda, B0, B1, shiftB, ldb, part); \
    else hipLaunchKernelGGL((tn_gemm_big<n / 16>), grid, dim3(512), 0, s, R, spw, nblk, nch, A, lda, B0, B1, shiftB, ldb, part);   \
  }
  TNB_CASE(32) TNB_CASE(64) TNB_CASE(128)
#undef TNB_CASE
  if (N == 256) {
    // (a stage of tn_gemm_w4 takes its rows of B from ONE source, fp32)
    if (pure)
      hipLaunchKernelGGL((tn_gemm_big<16, false, true, true>), grid, dim3(512), 0, s, R, spw, nblk, nch, A, lda, B0, B1, shiftB, ldb, part);
    else if (bf_b)
      hipLaunchKernelGGL((tn_gemm_big<16, false, true>), grid, dim3(512), 0, s, R, spw, nblk, nch, A, lda, B0, B1, shiftB, ldb, part);
    else if (shiftB == 0 || shiftB >= R || shiftB % TNB_STAGE == 0)
      hipLaunchKernelGGL((tn_gemm_w4<16, TNW4_PRE>), grid, dim3(256), 0, s, R, spw, nblk, nch, A, lda, B0, B1, shiftB, ldb, part);
    else
      hipLaunchKernelGGL((tn_gemm_big<16>), grid, dim3(512), 0, s, R, spw, nblk, nch, A, lda, B0, B1, shiftB, ldb, part);
  }
  const int total = M * N;
  hipLaunchKernelGGL(tn_big_reduce, dim3((total + 63) / 64), dim3(1024), 0, s, nch + nch0, nblk, N, (const float*)part, C, ldc);
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

}  // namespace fastgrnn
