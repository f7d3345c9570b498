// Generic FastGRNN scan kernels: any (T,B,F,H), dense or factorised W/U, every
// nonlinearity of the table, fp32 and fp64.  LDS-staged hidden state, VALU FMAs,
// weights streamed from L2.  This is the correctness floor under the MFMA-tiled
// fast path (kernels_mfma.hip) and the only path for fp64 / odd shapes.
//
// Reference semantics: forward = rnn.py:273-297 per step, BaseRNN loop rnn.py:657-660,
// operator outputs .cu:414; backward = .cu:91-119 (per element) + .cu:473-555 (loop,
// reductions, low-rank chain) evaluated factorised.
#include "common.h"

namespace fastgrnn {
namespace {

constexpr int BT = 8;          // utterances per workgroup
constexpr int NTHREADS = 256;

template <typename T>
__global__ void transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int R, int C) {
  // dst[c][r] = src[r][c]
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)R * C) {
    int r = (int)(i / C), c = (int)(i % C);
    dst[(size_t)c * R + r] = src[i];
  }
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void fwd_scan_generic(
    int Tn, int B, int F, int H, int rw, int ru, int gate, int upd,
    const T* __restrict__ x, const T* __restrict__ h0,
    const T* __restrict__ wA,   // dense: wT[F][H]; low-rank: w2T[rw][H]
    const T* __restrict__ w1,   // [rw][F] or null
    const T* __restrict__ uA,   // dense: uT[H][H]; low-rank: u2T[ru][H]
    const T* __restrict__ u1,   // [ru][H] or null
    const T* __restrict__ bz, const T* __restrict__ bh,
    const T* __restrict__ zeta, const T* __restrict__ nu,
    T* __restrict__ hs, T* __restrict__ zs, T* __restrict__ cs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* hbuf = reinterpret_cast<T*>(smem_raw);      // [2][BT][H]
  T* xs = hbuf + 2 * BT * H;                     // [BT][F]
  T* mx = xs + BT * F;                           // [BT][rw]
  T* mh = mx + BT * rw;                          // [BT][ru]
  const int tid = threadIdx.x, nth = blockDim.x;
  const int b0 = blockIdx.x * BT;
  const T sz = sigmoid_acc<T>(zeta[0]), sn = sigmoid_acc<T>(nu[0]);

  for (int i = tid; i < BT * H; i += nth) {
    int b = i / H;
    hbuf[i] = (b0 + b < B) ? h0[(size_t)(b0 + b) * H + (i % H)] : T(0);
    lds_writes_landed();
  }
  int cur = 0;
  for (int t = 0; t < Tn; ++t) {
    const T* hc = hbuf + cur * BT * H;
    T* hn = hbuf + (cur ^ 1) * BT * H;
    for (int i = tid; i < BT * F; i += nth) {
      int b = i / F;
      xs[i] = (b0 + b < B) ? x[((size_t)t * B + b0 + b) * F + (i % F)] : T(0);
      lds_writes_landed();                         // (second rule of DESIGN.md 4.0: see fwd below)
    }
    __syncthreads();
    if (rw) {
      for (int i = tid; i < BT * rw; i += nth) {
        int b = i / rw, j = i % rw;
        T a = 0;
        for (int f = 0; f < F; ++f) a += xs[b * F + f] * w1[(size_t)j * F + f];
        mx[i] = a;
        lds_writes_landed();
      }
    }
    if (ru) {
      for (int i = tid; i < BT * ru; i += nth) {
        int b = i / ru, j = i % ru;
        T a = 0;
        for (int k = 0; k < H; ++k) a += hc[b * H + k] * u1[(size_t)j * H + k];
        mh[i] = a;
        lds_writes_landed();
      }
    }
    if (rw || ru) __syncthreads();
    for (int n = tid; n < H; n += nth) {
      T accw[BT], accu[BT];
#pragma unroll
      for (int b = 0; b < BT; ++b) { accw[b] = 0; accu[b] = 0; }
      if (rw) {
        for (int j = 0; j < rw; ++j) {
          T wv = wA[(size_t)j * H + n];
#pragma unroll
          for (int b = 0; b < BT; ++b) accw[b] += mx[b * rw + j] * wv;
        }
      } else {
        for (int f = 0; f < F; ++f) {
          T wv = wA[(size_t)f * H + n];
#pragma unroll
          for (int b = 0; b < BT; ++b) accw[b] += xs[b * F + f] * wv;
        }
      }
      if (ru) {
        for (int j = 0; j < ru; ++j) {
          T uv = uA[(size_t)j * H + n];
#pragma unroll
          for (int b = 0; b < BT; ++b) accu[b] += mh[b * ru + j] * uv;
        }
      } else {
        for (int k = 0; k < H; ++k) {
          T uv = uA[(size_t)k * H + n];
#pragma unroll
          for (int b = 0; b < BT; ++b) accu[b] += hc[b * H + k] * uv;
        }
      }
      const T bzn = bz[n], bhn = bh[n];
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        T pre = accw[b] + accu[b];                       // rnn.py:289
        T z = act<T>(pre + bzn, gate);                   // rnn.py:290
        T c = act<T>(pre + bhn, upd);                    // rnn.py:292
        T hv = z * hc[b * H + n] + (sz * (T(1) - z) + sn) * c;   // rnn.py:294-295
        if (b0 + b < B) {
          size_t o = ((size_t)t * B + b0 + b) * H + n;
          hs[o] = hv;
          if (zs) zs[o] = z;
          if (cs) cs[o] = c;
        }
        hn[b * H + n] = hv;      // (behind the conditional stores: no LDS write in front of a branch with memory
                                 //  instructions behind it -- DESIGN.md 4.0, tools/lds_branch_vmem_scan.py)
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    cur ^= 1;
  }
}

// Reverse scan.  Emits d_pre[T,B,H] to the workspace (weight gradients and d_x are
// contracted afterwards by the GEMM kernels below), carries d_h in LDS, accumulates the
// bias / zeta / nu partial sums per workgroup.
template <typename T>
__global__ __launch_bounds__(NTHREADS) void bwd_scan_generic(
    int Tn, int B, int H, int ru, int gate, int upd,
    const T* __restrict__ ghs, const T* __restrict__ hs, const T* __restrict__ zs,
    const T* __restrict__ cs, const T* __restrict__ h0,
    const T* __restrict__ u,    // dense [H][H] or null
    const T* __restrict__ u1,   // [ru][H]
    const T* __restrict__ u2,   // [H][ru]
    const T* __restrict__ zeta, const T* __restrict__ nu,
    T* __restrict__ dpre_out,   // [T][B][H]
    T* __restrict__ d_h0,       // [B][H]
    T* __restrict__ part_bz,    // [nWG][H]
    T* __restrict__ part_bh,    // [nWG][H]
    T* __restrict__ part_zn)    // [nWG][2]
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* dh = reinterpret_cast<T*>(smem_raw);   // [BT][H]
  T* dp = dh + BT * H;                      // [BT][H]
  T* sbz = dp + BT * H;                     // [H]
  T* sbh = sbz + H;                         // [H]
  T* dmh = sbh + H;                         // [BT][ru]
  T* red = dmh + BT * ru;                   // [2*NTHREADS]
  const int tid = threadIdx.x, nth = blockDim.x;
  const int b0 = blockIdx.x * BT;
  const T sz = sigmoid_acc<T>(zeta[0]), sn = sigmoid_acc<T>(nu[0]);
  for (int i = tid; i < BT * H; i += nth) dh[i] = 0;
  for (int i = tid; i < H; i += nth) { sbz[i] = 0; sbh[i] = 0; }
  T pz = 0, pn = 0;
  __syncthreads();
  for (int t = Tn - 1; t >= 0; --t) {
    for (int n = tid; n < H; n += nth) {
      T az = 0, ah = 0;
      T dpa[BT], zga[BT];
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        T dpv = 0, zg = 0;
        if (b0 + b < B) {
          size_t o = ((size_t)t * B + b0 + b) * H + n;
          T g = ghs[o] + dh[b * H + n];                                    // .cu:474
          T z = zs[o], c = cs[o];
          T hp = (t == 0) ? h0[(size_t)(b0 + b) * H + n] : hs[o - (size_t)B * H];  // .cu:478-481
          T dcp = (sz * (T(1) - z) + sn) * dact<T>(c, upd) * g;            // .cu:109
          T dzp = (hp - sz * c) * dact<T>(z, gate) * g;                    // .cu:110
          dpv = dzp + dcp;                                                 // .cu:113
          az += dzp; ah += dcp;
          pz += (T(1) - z) * c * g;                                        // .cu:114
          pn += c * g;                                                     // .cu:115
          zg = z * g;                                                      // .cu:108
          dpre_out[o] = dpv;
        }
        dpa[b] = dpv; zga[b] = zg;
      }
      // the LDS writes behind all conditional loads / stores of this unit, and waited for before the loop's branch:
      // no LDS write in front of a branch with memory instructions behind it (lds_branch_vmem_scan.py)
#pragma unroll
      for (int b = 0; b < BT; ++b) { dp[b * H + n] = dpa[b]; dh[b * H + n] = zga[b]; }
      sbz[n] += az; sbh[n] += ah;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (ru) {
      for (int i = tid; i < BT * ru; i += nth) {
        int b = i / ru, j = i % ru;
        T a = 0;
        for (int n = 0; n < H; ++n) a += dp[b * H + n] * u2[(size_t)n * ru + j];
        dmh[i] = a;
        lds_writes_landed();
      }
      __syncthreads();
      for (int k = tid; k < H; k += nth) {
        T acc[BT];
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[b] = 0;
        for (int j = 0; j < ru; ++j) {
          T uv = u1[(size_t)j * H + k];
#pragma unroll
          for (int b = 0; b < BT; ++b) acc[b] += dmh[b * ru + j] * uv;
        }
#pragma unroll
        for (int b = 0; b < BT; ++b) dh[b * H + k] += acc[b];
        lds_writes_landed();
      }
    } else {
      for (int k = tid; k < H; k += nth) {
        T acc[BT];
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[b] = 0;
        for (int n = 0; n < H; ++n) {
          T uv = u[(size_t)n * H + k];
#pragma unroll
          for (int b = 0; b < BT; ++b) acc[b] += dp[b * H + n] * uv;      // .cu:537
        }
#pragma unroll
        for (int b = 0; b < BT; ++b) dh[b * H + k] += acc[b];
        lds_writes_landed();
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < BT * H; i += nth) {
    int b = i / H;
    if (b0 + b < B) d_h0[(size_t)(b0 + b) * H + (i % H)] = dh[i];
  }
  for (int i = tid; i < H; i += nth) {
    part_bz[(size_t)blockIdx.x * H + i] = sbz[i];
    part_bh[(size_t)blockIdx.x * H + i] = sbh[i];
  }
  red[tid] = pz; red[NTHREADS + tid] = pn;
  __syncthreads();
  for (int s = NTHREADS / 2; s > 0; s >>= 1) {
    if (tid < s) { red[tid] += red[tid + s]; red[NTHREADS + tid] += red[NTHREADS + tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { part_zn[2 * blockIdx.x] = red[0]; part_zn[2 * blockIdx.x + 1] = red[NTHREADS]; }
}

// d_bias_* = sum over workgroups; d_zeta/d_nu = sum * sigma'(raw)   (.cu:542-545,116-117)
template <typename T>
__global__ void finalize_small_grads(int nwg, int H, const T* __restrict__ part_bz,
                                     const T* __restrict__ part_bh, const T* __restrict__ part_zn,
                                     const T* __restrict__ zeta, const T* __restrict__ nu,
                                     T* __restrict__ d_bz, T* __restrict__ d_bh,
                                     T* __restrict__ d_zeta, T* __restrict__ d_nu) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < H) {
    T a = 0, b = 0;
    for (int w = 0; w < nwg; ++w) { a += part_bz[(size_t)w * H + n]; b += part_bh[(size_t)w * H + n]; }
    d_bz[n] = a; d_bh[n] = b;
  }
  if (n == 0) {
    T a = 0, b = 0;
    for (int w = 0; w < nwg; ++w) { a += part_zn[2 * w]; b += part_zn[2 * w + 1]; }
    T sz = sigmoid_acc<T>(zeta[0]), sn = sigmoid_acc<T>(nu[0]);
    d_zeta[0] = a * sz * (T(1) - sz);
    d_nu[0] = b * sn * (T(1) - sn);
  }
}

// Row r of a [R,ld] matrix whose first `shift` rows live in `first` and the rest in `rest`
// (used for H_prev: rows of t=0 are h0, rows of t>=1 are hs[t-1]).
template <typename T>
__device__ __forceinline__ const T* row_ptr(const T* first, const T* rest, size_t shift, size_t r, int ld) {
  return r < shift ? first + r * ld : rest + (r - shift) * ld;
}

// C[M,N] = A[M,K] . op(B);  M huge, K,N small.  transB: B is [N,K] (C = A.B^T) else [K,N].
template <typename T>
__global__ void gemm_rows(size_t M, int N, int K, const T* __restrict__ A0, const T* __restrict__ A1,
                          size_t shiftA, const T* __restrict__ Bm, int transB, T* __restrict__ C) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * (size_t)N) return;
  size_t m = i / N; int n = (int)(i % N);
  const T* a = row_ptr<T>(A0, A1, shiftA, m, K);
  T acc = 0;
  if (transB) { for (int k = 0; k < K; ++k) acc += a[k] * Bm[(size_t)n * K + k]; }
  else        { for (int k = 0; k < K; ++k) acc += a[k] * Bm[(size_t)k * N + n]; }
  C[i] = acc;
}

// part[s][M,N] = sum_{r in chunk s} A[r][m] * B[r][n];  A:[R, lda] (first M columns used),
// B:[R, ldb] (first N columns used; B with row shift).
template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_splitk(size_t R, int M, int N, size_t chunk,
                                                      const T* __restrict__ A, int lda,
                                                      const T* __restrict__ B0, const T* __restrict__ B1, size_t shiftB,
                                                      int ldb, T* __restrict__ part) {
  __shared__ T As[16][17];
  __shared__ T Bs[16][17];
  const int tn = (N + 15) / 16;
  const int tile = blockIdx.x, s = blockIdx.y;
  const int m0 = (tile / tn) * 16, n0 = (tile % tn) * 16;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  size_t r0 = (size_t)s * chunk, r1 = r0 + chunk; if (r1 > R) r1 = R;
  T acc = 0;
  for (size_t r = r0; r < r1; r += 16) {
    size_t rr = r + ty;
    // (both predicated loads first, then the LDS writes: second rule of DESIGN.md 4.0)
    const T av = (rr < r1 && m0 + tx < M) ? A[rr * lda + m0 + tx] : T(0);
    const T bv = (rr < r1 && n0 + tx < N) ? row_ptr<T>(B0, B1, shiftB, rr, ldb)[n0 + tx] : T(0);
    As[ty][tx] = av;
    Bs[ty][tx] = bv;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += As[k][ty] * Bs[k][tx];
    __syncthreads();
  }
  if (m0 + ty < M && n0 + tx < N) part[((size_t)s * M + m0 + ty) * N + n0 + tx] = acc;
}

template <typename T>
__global__ void reduce_splitk(int nsplit, size_t MN, const T* __restrict__ part, T* __restrict__ C) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= MN) return;
  T a = 0;
  for (int s = 0; s < nsplit; ++s) a += part[(size_t)s * MN + i];
  C[i] = a;
}

constexpr size_t SPLITK_CHUNK = 2048;

struct BwdWs {
  size_t dpre, part_bz, part_bh, part_zn, mx, dmx, mh, dmh, splitk, total;
};

BwdWs bwd_layout(const fastgrnn_desc& d) {
  size_t es = d.dtype == FASTGRNN_F64 ? 8 : 4;
  size_t TB = (size_t)d.T * d.B;
  size_t nwg = (d.B + BT - 1) / BT;
  size_t nsplit = (TB + SPLITK_CHUNK - 1) / SPLITK_CHUNK;
  size_t maxMN = (size_t)d.H * (d.H > d.F ? d.H : d.F);
  BwdWs w; size_t o = 0;
  w.dpre = o; o += align256(TB * d.H * es);
  w.part_bz = o; o += align256(nwg * d.H * es);
  w.part_bh = o; o += align256(nwg * d.H * es);
  w.part_zn = o; o += align256(nwg * 2 * es);
  w.mx = o; o += align256(TB * d.w_rank * es);
  w.dmx = o; o += align256(TB * d.w_rank * es);
  w.mh = o; o += align256(TB * d.u_rank * es);
  w.dmh = o; o += align256(TB * d.u_rank * es);
  w.splitk = o; o += align256(nsplit * maxMN * es);
  w.total = o;
  return w;
}

// Same product, LDS-tiled: a workgroup owns 64 rows of A and streams K in chunks of 64; thread
// (row r = tid&63, column group cg = tid>>6) keeps up to 16 outputs n = cg + 4c in registers.  A
// chunk reads are coalesced (consecutive threads = consecutive k of one row), the A element of a
// row is read once per k from LDS (stride-65 rows: conflict-free) and the B row is a wave-wide
// broadcast.  N <= 64.
template <typename T>
__global__ __launch_bounds__(256) void gemm_rows_tiled(size_t M, int N, int K, const T* __restrict__ A0,
                                                       const T* __restrict__ A1, size_t shiftA,
                                                       const T* __restrict__ Bm, int transB, T* __restrict__ C) {
  constexpr int TM = 64, KC = 64, NC = 16;
  __shared__ T As[TM][KC + 1];
  __shared__ T Bs[KC][64];
  const int tid = threadIdx.x, r = tid & 63, cg = tid >> 6;
  const size_t m0 = (size_t)blockIdx.x * TM;
  T acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0;
  for (int k0 = 0; k0 < K; k0 += KC) {
    for (int idx = tid; idx < TM * KC; idx += 256) {
      const int rr = idx / KC, kk = idx % KC;
      const size_t m = m0 + rr;
      As[rr][kk] = (m < M && k0 + kk < K) ? row_ptr<T>(A0, A1, shiftA, m, K)[k0 + kk] : T(0);
      lds_writes_landed();                       // (second rule of DESIGN.md 4.0: the next iteration's predicated load)
    }
    for (int idx = tid; idx < KC * 64; idx += 256) {
      const int kk = idx / 64, n = idx % 64;
      T v = 0;
      if (n < N && k0 + kk < K) v = transB ? Bm[(size_t)n * K + k0 + kk] : Bm[(size_t)(k0 + kk) * N + n];
      Bs[kk][n] = v;
      lds_writes_landed();
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < KC; ++kk) {
      const T a = As[r][kk];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] += a * Bs[kk][cg + 4 * c];
    }
    __syncthreads();
  }
  const size_t m = m0 + r;
  if (m < M) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int n = cg + 4 * c;
      if (n < N) C[m * N + n] = acc[c];
    }
  }
}

template <typename T>
void launch_rows(size_t M, int N, int K, const T* A0, const T* A1, size_t shiftA, const T* Bm, int transB,
                 T* C, hipStream_t s) {
  if (N <= 64) {
    hipLaunchKernelGGL(gemm_rows_tiled<T>, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, s, M, N, K, A0, A1, shiftA,
                       Bm, transB, C);
    return;
  }
  size_t tot = M * (size_t)N;
  hipLaunchKernelGGL(gemm_rows<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, M, N, K, A0, A1, shiftA, Bm,
                     transB, C);
}

template <typename T>
void launch_tn_ld(size_t R, int M, int N, const T* A, int lda, const T* B0, const T* B1, size_t shiftB, int ldb,
                  T* part, T* C, hipStream_t s) {
  int nsplit = (int)((R + SPLITK_CHUNK - 1) / SPLITK_CHUNK);
  int tiles = ((M + 15) / 16) * ((N + 15) / 16);
  hipLaunchKernelGGL(gemm_tn_splitk<T>, dim3(tiles, nsplit), dim3(256), 0, s, R, M, N, SPLITK_CHUNK, A, lda, B0, B1,
                     shiftB, ldb, part);
  size_t MN = (size_t)M * N;
  hipLaunchKernelGGL(reduce_splitk<T>, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, s, nsplit, MN, part, C);
}

template <typename T>
void launch_tn(size_t R, int M, int N, const T* A, const T* B0, const T* B1, size_t shiftB, T* part, T* C,
               hipStream_t s) {
  launch_tn_ld<T>(R, M, N, A, M, B0, B1, shiftB, N, part, C, s);
}

template <typename T>
int generic_forward_t(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs,
                      void* zs, void* cs, void* ws, hipStream_t s) {
  const int F = d.F, H = d.H, rw = d.w_rank, ru = d.u_rank;
  T* wA = reinterpret_cast<T*>(ws);
  size_t wA_n = (size_t)(rw ? rw : F) * H;
  T* uA = reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + align256(wA_n * sizeof(T)));
  size_t uA_n = (size_t)(ru ? ru : H) * H;
  // [out,in] -> [in,out] so that lanes (one per output unit) read contiguous weights
  const T* wsrc = reinterpret_cast<const T*>(rw ? p.w2 : p.w);
  const T* usrc = reinterpret_cast<const T*>(ru ? p.u2 : p.u);
  hipLaunchKernelGGL(transpose_kernel<T>, dim3((unsigned)((wA_n + 255) / 256)), dim3(256), 0, s, wsrc, wA, H,
                     rw ? rw : F);
  hipLaunchKernelGGL(transpose_kernel<T>, dim3((unsigned)((uA_n + 255) / 256)), dim3(256), 0, s, usrc, uA, H,
                     ru ? ru : H);
  size_t lds = ((size_t)2 * BT * H + (size_t)BT * F + (size_t)BT * rw + (size_t)BT * ru) * sizeof(T);
  if (lds > 160 * 1024) return FASTGRNN_ERR_UNSUPPORTED;
  int nth = H >= NTHREADS ? NTHREADS : ((H + 63) / 64) * 64;
  unsigned nwg = (unsigned)((d.B + BT - 1) / BT);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_scan_generic<T>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(fwd_scan_generic<T>, dim3(nwg), dim3(nth), lds, s, d.T, d.B, F, H, rw, ru, d.gate_nl,
                     d.update_nl, reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(h0), wA,
                     reinterpret_cast<const T*>(p.w1), uA, reinterpret_cast<const T*>(p.u1),
                     reinterpret_cast<const T*>(p.bias_gate), reinterpret_cast<const T*>(p.bias_update),
                     reinterpret_cast<const T*>(p.zeta), reinterpret_cast<const T*>(p.nu),
                     reinterpret_cast<T*>(hs), reinterpret_cast<T*>(zs), reinterpret_cast<T*>(cs));
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

template <typename T>
int generic_backward_t(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs_, const void* x_,
                       const void* hs_, const void* zs, const void* cs, const void* h0_, const fastgrnn_grads& g,
                       void* ws, hipStream_t s) {
  const int F = d.F, H = d.H, rw = d.w_rank, ru = d.u_rank;
  const size_t TB = (size_t)d.T * d.B;
  BwdWs L = bwd_layout(d);
  char* base = reinterpret_cast<char*>(ws);
  T* dpre = reinterpret_cast<T*>(base + L.dpre);
  T* part_bz = reinterpret_cast<T*>(base + L.part_bz);
  T* part_bh = reinterpret_cast<T*>(base + L.part_bh);
  T* part_zn = reinterpret_cast<T*>(base + L.part_zn);
  T* mx = reinterpret_cast<T*>(base + L.mx);
  T* dmx = reinterpret_cast<T*>(base + L.dmx);
  T* mh = reinterpret_cast<T*>(base + L.mh);
  T* dmh = reinterpret_cast<T*>(base + L.dmh);
  T* splitk = reinterpret_cast<T*>(base + L.splitk);
  const T* x = reinterpret_cast<const T*>(x_);
  const T* hs = reinterpret_cast<const T*>(hs_);
  const T* h0 = reinterpret_cast<const T*>(h0_);
  unsigned nwg = (unsigned)((d.B + BT - 1) / BT);
  size_t lds = ((size_t)2 * BT * H + 2 * (size_t)H + (size_t)BT * ru + 2 * NTHREADS) * sizeof(T);
  if (lds > 160 * 1024) return FASTGRNN_ERR_UNSUPPORTED;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_scan_generic<T>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(bwd_scan_generic<T>, dim3(nwg), dim3(NTHREADS), lds, s, d.T, d.B, H, ru, d.gate_nl, d.update_nl,
                     reinterpret_cast<const T*>(ghs_), hs, reinterpret_cast<const T*>(zs),
                     reinterpret_cast<const T*>(cs), h0, reinterpret_cast<const T*>(p.u),
                     reinterpret_cast<const T*>(p.u1), reinterpret_cast<const T*>(p.u2),
                     reinterpret_cast<const T*>(p.zeta), reinterpret_cast<const T*>(p.nu), dpre,
                     reinterpret_cast<T*>(g.d_h0), part_bz, part_bh, part_zn);
  hipLaunchKernelGGL(finalize_small_grads<T>, dim3((H + 255) / 256), dim3(256), 0, s, (int)nwg, H, part_bz, part_bh,
                     part_zn, reinterpret_cast<const T*>(p.zeta), reinterpret_cast<const T*>(p.nu),
                     reinterpret_cast<T*>(g.d_bias_gate), reinterpret_cast<T*>(g.d_bias_update),
                     reinterpret_cast<T*>(g.d_zeta), reinterpret_cast<T*>(g.d_nu));
  // H_prev rows: first B rows are h0, the rest are hs[0..T-2]
  const size_t shiftH = (size_t)d.B;
  if (rw) {
    const T* w1 = reinterpret_cast<const T*>(p.w1);
    const T* w2 = reinterpret_cast<const T*>(p.w2);
    launch_rows<T>(TB, rw, F, x, x, 0, w1, 1, mx, s);                 // mx  = X . w1^T
    launch_rows<T>(TB, rw, H, dpre, dpre, 0, w2, 0, dmx, s);          // dmx = dpre . w2
    launch_tn<T>(TB, H, rw, dpre, mx, mx, 0, splitk, reinterpret_cast<T*>(g.d_w2), s);   // d_w2 = dpre^T . mx
    launch_tn<T>(TB, rw, F, dmx, x, x, 0, splitk, reinterpret_cast<T*>(g.d_w1), s);      // d_w1 = dmx^T . X
    launch_rows<T>(TB, F, rw, dmx, dmx, 0, w1, 0, reinterpret_cast<T*>(g.d_x), s);       // d_x  = dmx . w1
  } else {
    const T* w = reinterpret_cast<const T*>(p.w);
    launch_rows<T>(TB, F, H, dpre, dpre, 0, w, 0, reinterpret_cast<T*>(g.d_x), s);       // .cu:538
    launch_tn<T>(TB, H, F, dpre, x, x, 0, splitk, reinterpret_cast<T*>(g.d_w), s);       // .cu:539
  }
  if (ru) {
    const T* u1 = reinterpret_cast<const T*>(p.u1);
    const T* u2 = reinterpret_cast<const T*>(p.u2);
    launch_rows<T>(TB, ru, H, h0, hs, shiftH, u1, 1, mh, s);          // mh  = Hprev . u1^T
    launch_rows<T>(TB, ru, H, dpre, dpre, 0, u2, 0, dmh, s);          // dmh = dpre . u2
    launch_tn<T>(TB, H, ru, dpre, mh, mh, 0, splitk, reinterpret_cast<T*>(g.d_u2), s);
    launch_tn<T>(TB, ru, H, dmh, h0, hs, shiftH, splitk, reinterpret_cast<T*>(g.d_u1), s);
  } else {
    launch_tn<T>(TB, H, H, dpre, h0, hs, shiftH, splitk, reinterpret_cast<T*>(g.d_u), s);  // .cu:540
  }
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

}  // namespace

size_t tn_gemm_f32_ws(size_t R, int M, int N) {
  return align256(((R + SPLITK_CHUNK - 1) / SPLITK_CHUNK) * (size_t)M * N * 4);
}
void tn_gemm_f32(size_t R, int M, int N, const float* A, int lda, const float* B0, const float* B1, size_t shiftB,
                 int ldb, float* part, float* C, hipStream_t s) {
  launch_tn_ld<float>(R, M, N, A, lda, B0, B1, shiftB, ldb, part, C, s);
}

size_t generic_forward_ws(const fastgrnn_desc& d) {
  size_t es = d.dtype == FASTGRNN_F64 ? 8 : 4;
  return align256((size_t)(d.w_rank ? d.w_rank : d.F) * d.H * es) +
         align256((size_t)(d.u_rank ? d.u_rank : d.H) * d.H * es);
}
size_t generic_backward_ws(const fastgrnn_desc& d) { return bwd_layout(d).total; }

int generic_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs,
                    void* zs, void* cs, void* ws, hipStream_t s) {
  return d.dtype == FASTGRNN_F64 ? generic_forward_t<double>(d, p, x, h0, hs, zs, cs, ws, s)
                                 : generic_forward_t<float>(d, p, x, h0, hs, zs, cs, ws, s);
}
int generic_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x,
                     const void* hs, const void* zs, const void* cs, const void* h0, const fastgrnn_grads& g,
                     void* ws, hipStream_t s) {
  return d.dtype == FASTGRNN_F64 ? generic_backward_t<double>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s)
                                 : generic_backward_t<float>(d, p, ghs, x, hs, zs, cs, h0, g, ws, s);
}

}  // namespace fastgrnn
