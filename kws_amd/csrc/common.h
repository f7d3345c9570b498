// Shared device helpers and internal launcher declarations for libfastgrnn_hip.so.
// gfx950 only; see include/fastgrnn_hip.h for the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/fastgrnn_hip.h"

namespace fastgrnn {

// ---- nonlinearities (reference table: rnn.py:40-67; device forms .cu:17-40) -------------
template <typename T> __device__ __forceinline__ T sigmoid_acc(T a);
template <> __device__ __forceinline__ float sigmoid_acc<float>(float a) { return 1.0f / (1.0f + expf(-a)); }
template <> __device__ __forceinline__ double sigmoid_acc<double>(double a) { return 1.0 / (1.0 + exp(-a)); }
template <typename T> __device__ __forceinline__ T tanh_acc(T a);
template <> __device__ __forceinline__ float tanh_acc<float>(float a) { return tanhf(a); }
template <> __device__ __forceinline__ double tanh_acc<double>(double a) { return tanh(a); }

template <typename T> __device__ __forceinline__ T act(T a, int nl) {
  switch (nl) {
    case FASTGRNN_NL_SIGMOID: return sigmoid_acc<T>(a);
    case FASTGRNN_NL_RELU: return a > T(0) ? a : T(0);
    case FASTGRNN_NL_TANH: return tanh_acc<T>(a);
    case FASTGRNN_NL_QUANT_TANH: return a > T(1) ? T(1) : (a < T(-1) ? T(-1) : a);
    case FASTGRNN_NL_QUANT_SIGM: { T v = (a + T(1)) / T(2); return v > T(1) ? T(1) : (v < T(0) ? T(0) : v); }
    default: { T v = (a + T(2)) / T(4); return v > T(1) ? T(1) : (v < T(0) ? T(0) : v); }
  }
}

// derivative expressed through the OUTPUT y, as the reference kernels do (.cu:27-40).
// tanh uses 1-y^2 (the reference's unrolled tanh-gate backward wrongly uses d_sigmoid,
// .cu:519-521; not reproduced).  relu tests y > 0 (== .cu:33-35 where differentiable).
template <typename T> __device__ __forceinline__ T dact(T y, int nl) {
  switch (nl) {
    case FASTGRNN_NL_SIGMOID: return (T(1) - y) * y;
    case FASTGRNN_NL_RELU: return y > T(0) ? T(1) : T(0);
    case FASTGRNN_NL_TANH: return T(1) - y * y;
    case FASTGRNN_NL_QUANT_TANH: return (y < T(1) && y > T(-1)) ? T(1) : T(0);
    case FASTGRNN_NL_QUANT_SIGM: return (y < T(1) && y > T(0)) ? T(0.5) : T(0);
    default: return (y < T(1) && y > T(0)) ? T(0.25) : T(0);
  }
}

static inline size_t align256(size_t n) { return (n + 255) & ~size_t(255); }

// ---- internal launchers (implemented in kernels_generic.hip / kernels_mfma.hip) ---------
size_t generic_forward_ws(const fastgrnn_desc& d);
size_t generic_backward_ws(const fastgrnn_desc& d);
int generic_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0,
                    void* hs, void* zs, void* cs, void* ws, hipStream_t s);
int generic_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x,
                     const void* hs, const void* zs, const void* cs, const void* h0,
                     const fastgrnn_grads& g, void* ws, hipStream_t s);

// C[M,N] = A[:, :M]^T . B[:, :N] over R rows (deterministic split-K); rows of B below shiftB come from
// B0, the rest from B1.  part: tn_gemm_f32_ws(R, M, N) bytes of workspace.
size_t tn_gemm_f32_ws(size_t R, int M, int N);
void tn_gemm_f32(size_t R, int M, int N, const float* A, int lda, const float* B0, const float* B1, size_t shiftB,
                 int ldb, float* part, float* C, hipStream_t s);

bool mfma_supported(const fastgrnn_desc& d, int direction);
size_t mfma_forward_ws(const fastgrnn_desc& d);
size_t mfma_backward_ws(const fastgrnn_desc& d);
int mfma_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0,
                 void* hs, void* zs, void* cs, void* ws, hipStream_t s);
int mfma_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x,
                  const void* hs, const void* zs, const void* cs, const void* h0,
                  const fastgrnn_grads& g, void* ws, hipStream_t s);

// split-precision (3 x bf16 planes, 6 MFMA terms) scan on the bf16 matrix pipe (kernels_split.hip)
bool split_supported(const fastgrnn_desc& d, int direction);
size_t split_forward_ws(const fastgrnn_desc& d);
size_t split_backward_ws(const fastgrnn_desc& d);
bool split_forward_ws_optional(const fastgrnn_desc& d);   // the forward workspace is only used when z_s is NULL
bool split_dx_optional(const fastgrnn_desc& d);           // the backward accepts d_x == NULL (no input gradient)
int split_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0,
                  void* hs, void* zs, void* cs, void* ws, hipStream_t s);
int split_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x,
                   const void* hs, const void* zs, const void* cs, const void* h0,
                   const fastgrnn_grads& g, void* ws, hipStream_t s);

// batched split-precision GEMMs around the scans (kernels_gemm.hip)
//   rows_gemm:   C[R,N] = A[R,K] . Wt^T, Wt[n][k] = trans_w ? W[k*N + n] : W[n*K + k]; A / C may be bf16 sequences
//   tn_gemm_big: C[M,N] (row stride ldc) = A[R,M]^T . B[R,N]; rows of B below shiftB come from B0, the rest from B1
//                shifted down by shiftB rows; part: tn_gemm_big_ws(R, M, N) bytes
bool rows_gemm_supported(int N, int K, bool trans_w);
int rows_gemm(size_t R, int N, int K, bool trans_w, const void* A, const float* W, void* C, bool bf_in, bool bf_out,
              hipStream_t s);
bool tn_gemm_big_supported(int M, int N);
size_t tn_gemm_big_ws(size_t R, int M, int N);
int tn_gemm_big_run(size_t R, int M, int N, const float* A, int lda, const float* B0, const void* B1, size_t shiftB,
                    int ldb, float* part, float* C, int ldc, hipStream_t s, bool bf_b = false);   // bf_b: B1 holds bf16
// ... with row r of B = B0[r / period] where r is a multiple of period, else B1[r - 1] (N = 256 only)
int tn_gemm_big_run_periodic(size_t R, int M, int N, const float* A, int lda, const float* B0, const void* B1, size_t period,
                             int ldb, float* part, float* C, int ldc, hipStream_t s, bool bf_b = false);

// dense H = 256 / F = 32 scans (kernels_h256.hip), dispatched through split_supported / split_forward / split_backward
bool h256_shape(const fastgrnn_desc& d);
bool h256_supported(const fastgrnn_desc& d, int direction);
size_t h256_forward_ws(const fastgrnn_desc& d);
size_t h256_backward_ws(const fastgrnn_desc& d);
int h256_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs, void* zs,
                 void* cs, void* ws, hipStream_t s);
int h256_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x, const void* hs,
                  const void* zs, const void* cs, const void* h0, const fastgrnn_grads& g, void* ws, hipStream_t s);

// low-rank H = 256 / F = 32 scans, ranks <= 16 (kernels_lowrank.hip), dispatched like the H = 256 dense ones
void bft_transpose_f32(int B, int T, const float* src, float* dst, bool to_time_major, hipStream_t s);
bool lowrank_shape(const fastgrnn_desc& d);
bool lowrank_supported(const fastgrnn_desc& d, int direction);
size_t lowrank_forward_ws(const fastgrnn_desc& d);
size_t lowrank_backward_ws(const fastgrnn_desc& d);
int lowrank_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs, void* zs,
                    void* cs, void* ws, hipStream_t s);
int lowrank_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x, const void* hs,
                     const void* zs, const void* cs, const void* h0, const fastgrnn_grads& g, void* ws, hipStream_t s);

// other factorised H = 256 / F = 32 cells (rank above 16, or only one of W, U factorised) on the dense H = 256 scans:
// the factors are multiplied out per call and the dense gradients projected back (kernels_densify.hip)
bool densified_shape(const fastgrnn_desc& d);
bool densified_supported(const fastgrnn_desc& d, int direction);
size_t densified_forward_ws(const fastgrnn_desc& d);
size_t densified_backward_ws(const fastgrnn_desc& d);
int densified_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs, void* zs,
                      void* cs, void* ws, hipStream_t s);
int densified_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x, const void* hs,
                       const void* zs, const void* cs, const void* h0, const fastgrnn_grads& g, void* ws, hipStream_t s);

// classifier head on the last state: Linear + log_softmax + NLL, forward and backward (kernels_head.hip)
bool head_supported(int B, int H, int C);
size_t head_ws_bytes(int B, int H, int C);
int head_xent(int B, int H, int C, const void* h_last, const void* fc_w, const void* fc_b, const void* labels,
              void* loss, void* logp, void* d_h, void* d_w, void* d_b, void* ws, hipStream_t s);

// test hook (kernels_debug.hip): fill every CU's LDS and vector registers with a bit pattern
int debug_poison(unsigned pattern, hipStream_t s);

namespace {
// Completion read (operand rule): ONE vector instruction the compiler cannot drop reads `v` -- an
// element of the youngest accumulator, or a sum of several: MFMAs retire in order, so once it has executed every MFMA
// issued before it has read its operands.  v_readfirstlane into a scalar register that an empty asm consumes: no
// branch, no memory instruction (the first form, `if (v == <never>) <store>`, put a conditional store into scan loops:
// second rule of 4.0).
__device__ __forceinline__ void completion_read(float v) {
  const int s = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v));
  asm volatile("" :: "s"(s));
}

// Second rule of DESIGN.md 4.0, by construction: no LDS write is pending when a conditional branch -- exec-masked or
// wave-uniform -- with a vector-memory instruction behind it is taken.  Placed in front of such regions where LDS
// writes precede them in the same loop iteration without a barrier in between.
__device__ __forceinline__ void lds_writes_landed() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

}  // namespace

}  // namespace fastgrnn
