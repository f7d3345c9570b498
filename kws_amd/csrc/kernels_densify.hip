// Factorised cells that the register-resident low-rank scans do not cover (any rank on the dense shapes H = 128 with
// F = 32 / 64 / 128 / 256; H = 256 / F = 32 with a rank above 16 or with only one of W, U factorised: rnn.py:783-798)
// on the DENSE split-precision scans: W = W2.W1 and U = U2.U1 are multiplied out once per call -- what the reference's
// CUDA operator does for every low-rank cell (.cu:353-362) -- the dense kernels run on them, and the backward projects
// the dense gradients onto the factors (the chain rule the reference applies at .cu:546-555):
// dW1 = W2^T dW,  dW2 = dW W1^T,  dU1 = U2^T dU,  dU2 = dU U1^T.  Tiny GEMMs (<= 256 x 256 x 256, fp64 accumulation, one
// thread per output, fixed order) against scans of hundreds of microseconds.  The dense shape's own limits apply
// (layouts, dtypes, gates, flags).
#include "common.h"

namespace fastgrnn {
namespace {

// C[m][n] = sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]
__global__ __launch_bounds__(256) void small_gemm(int M, int N, int K, const float* __restrict__ A, int sam, int sak,
                                                  const float* __restrict__ B, int sbk, int sbn, float* __restrict__ C) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * N) return;
  const int m = idx / N, n = idx - m * N;
  double acc = 0.0;
  for (int k = 0; k < K; ++k) acc += (double)A[(size_t)m * sam + (size_t)k * sak] * (double)B[(size_t)k * sbk + (size_t)n * sbn];
  C[idx] = (float)acc;
}

void gemm(int M, int N, int K, const float* A, int sam, int sak, const float* B, int sbk, int sbn, float* C, hipStream_t s) {
  hipLaunchKernelGGL(small_gemm, dim3((M * N + 255) / 256), dim3(256), 0, s, M, N, K, A, sam, sak, B, sbk, sbn, C);
}

fastgrnn_desc dense_desc(const fastgrnn_desc& d) {
  fastgrnn_desc e = d;
  e.w_rank = 0; e.u_rank = 0;
  return e;
}

struct DenseWs { size_t wd, ud, dwd, dud, inner, total; };
DenseWs layout(const fastgrnn_desc& d, bool backward) {
  const size_t H2 = d.H, F2 = d.F;
  DenseWs L; size_t o = 0;
  L.wd = o; o += align256(H2 * F2 * 4);
  L.ud = o; o += align256(H2 * H2 * 4);
  L.dwd = o; if (backward) o += align256(H2 * F2 * 4);
  L.dud = o; if (backward) o += align256(H2 * H2 * 4);
  L.inner = o;
  const fastgrnn_desc e = dense_desc(d);
  o += backward ? split_backward_ws(e) : split_forward_ws(e);
  L.total = o;
  return L;
}

// the dense matrices of the cell: multiplied out into the workspace, or the caller's own where it is dense already
void densify(const fastgrnn_desc& d, const fastgrnn_params& p, char* base, const DenseWs& L, fastgrnn_params& q,
             hipStream_t s) {
  const int H2 = d.H, F2 = d.F;
  q = p;
  if (d.w_rank) {                                    // W[H,F] = W2[H,r] . W1[r,F]
    gemm(H2, F2, d.w_rank, (const float*)p.w2, d.w_rank, 1, (const float*)p.w1, F2, 1, (float*)(base + L.wd), s);
    q.w = base + L.wd;
  }
  if (d.u_rank) {                                    // U[H,H] = U2[H,r] . U1[r,H]
    gemm(H2, H2, d.u_rank, (const float*)p.u2, d.u_rank, 1, (const float*)p.u1, H2, 1, (float*)(base + L.ud), s);
    q.u = base + L.ud;
  }
  q.w1 = q.w2 = q.u1 = q.u2 = nullptr;
}

}  // namespace

bool densified_shape(const fastgrnn_desc& d) {
  return (d.w_rank > 0 || d.u_rank > 0) && d.w_rank <= d.H && d.u_rank <= d.H && !lowrank_shape(d) &&
         ((d.H == 256 && (d.F == 32 || d.F == 64 || d.F == 128)) ||
          (d.H == 128 && (d.F == 32 || d.F == 64 || d.F == 128 || d.F == 256)));
}

bool densified_supported(const fastgrnn_desc& d, int direction) {
  if (d.flags & FASTGRNN_FLAG_FWD_4WAVE) return false;
  if (d.H == 256 && d.dtype != FASTGRNN_F32) return false;   // (bf16 sequences: the dense H = 256 scans take them, this path is untested with them)
  return split_supported(dense_desc(d), direction);
}

size_t densified_forward_ws(const fastgrnn_desc& d) { return layout(d, false).total; }
size_t densified_backward_ws(const fastgrnn_desc& d) { return layout(d, true).total; }

int densified_forward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* x, const void* h0, void* hs, void* zs,
                      void* cs, void* ws, hipStream_t s) {
  if (!ws) return FASTGRNN_ERR_WORKSPACE;
  const DenseWs L = layout(d, false);
  char* base = reinterpret_cast<char*>(ws);
  fastgrnn_params q;
  densify(d, p, base, L, q, s);
  // (under FASTGRNN_FLAG_SAVE_PREACT the dense contract saves the pre-activation alone: c_s is not used)
  return split_forward(dense_desc(d), q, x, h0, hs, zs, (d.flags & FASTGRNN_FLAG_SAVE_PREACT) ? nullptr : cs, base + L.inner, s);
}

int densified_backward(const fastgrnn_desc& d, const fastgrnn_params& p, const void* ghs, const void* x, const void* hs,
                       const void* zs, const void* cs, const void* h0, const fastgrnn_grads& g, void* ws, hipStream_t s) {
  if (!ws) return FASTGRNN_ERR_WORKSPACE;
  const DenseWs L = layout(d, true);
  char* base = reinterpret_cast<char*>(ws);
  fastgrnn_params q;
  densify(d, p, base, L, q, s);
  fastgrnn_grads gd = g;
  gd.d_w = d.w_rank ? (void*)(base + L.dwd) : g.d_w;
  gd.d_u = d.u_rank ? (void*)(base + L.dud) : g.d_u;
  gd.d_w1 = gd.d_w2 = gd.d_u1 = gd.d_u2 = nullptr;
  const int st = split_backward(dense_desc(d), q, ghs, x, hs, zs, cs, h0, gd, base + L.inner, s);
  if (st != FASTGRNN_OK) return st;
  const int H2 = d.H, F2 = d.F;
  if (d.w_rank) {
    const float* dW = (const float*)(base + L.dwd);
    // dW1[r,F] = W2^T . dW :  A[m=i][k=h] = W2[h*r + i],  B[k=h][n=f] = dW[h*F + f]
    gemm(d.w_rank, F2, H2, (const float*)p.w2, 1, d.w_rank, dW, F2, 1, (float*)g.d_w1, s);
    // dW2[H,r] = dW . W1^T :  A[m=h][k=f] = dW[h*F + f],  B[k=f][n=i] = W1[i*F + f]
    gemm(H2, d.w_rank, F2, dW, F2, 1, (const float*)p.w1, 1, F2, (float*)g.d_w2, s);
  }
  if (d.u_rank) {
    const float* dU = (const float*)(base + L.dud);
    gemm(d.u_rank, H2, H2, (const float*)p.u2, 1, d.u_rank, dU, H2, 1, (float*)g.d_u1, s);     // dU1[r,H] = U2^T . dU
    gemm(H2, d.u_rank, H2, dU, H2, 1, (const float*)p.u1, 1, H2, (float*)g.d_u2, s);           // dU2[H,r] = dU . U1^T
  }
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

}  // namespace fastgrnn
