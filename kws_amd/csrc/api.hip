// C ABI of libfastgrnn_hip.so (include/fastgrnn_hip.h): argument validation and
// dispatch between the MFMA-tiled fp32 scan and the generic scan.  Pure launches:
// no allocation, no synchronisation, no global state.
#include "common.h"

using namespace fastgrnn;

namespace {

bool nl_ok(int nl) { return nl >= FASTGRNN_NL_SIGMOID && nl <= FASTGRNN_NL_QUANT_SIGM4; }

int check_desc(const fastgrnn_desc* d) {
  if (!d) return FASTGRNN_ERR_NULL_POINTER;
  if (d->T < 1 || d->B < 1 || d->F < 1 || d->H < 1) return FASTGRNN_ERR_BAD_SHAPE;
  if (d->w_rank < 0 || d->u_rank < 0) return FASTGRNN_ERR_BAD_SHAPE;
  // every tensor is indexed with size_t inside the kernels; keep T*B*max(H,F) below 2^40
  if ((double)d->T * d->B * (d->H > d->F ? d->H : d->F) > 1099511627776.0) return FASTGRNN_ERR_BAD_SHAPE;
  if (!nl_ok(d->gate_nl) || !nl_ok(d->update_nl)) return FASTGRNN_ERR_BAD_NONLINEARITY;
  if (d->dtype != FASTGRNN_F32 && d->dtype != FASTGRNN_F64 && d->dtype != FASTGRNN_BF16_IO) return FASTGRNN_ERR_BAD_DTYPE;
  return FASTGRNN_OK;
}

int check_params(const fastgrnn_desc* d, const fastgrnn_params* p) {
  if (!p) return FASTGRNN_ERR_NULL_POINTER;
  if (d->w_rank ? (!p->w1 || !p->w2) : !p->w) return FASTGRNN_ERR_NULL_POINTER;
  if (d->u_rank ? (!p->u1 || !p->u2) : !p->u) return FASTGRNN_ERR_NULL_POINTER;
  if (!p->bias_gate || !p->bias_update || !p->zeta || !p->nu) return FASTGRNN_ERR_NULL_POINTER;
  return FASTGRNN_OK;
}

int check_ws(void* ws, size_t have, size_t need) {
  if (need == 0) return FASTGRNN_OK;
  if (!ws || have < need || (reinterpret_cast<uintptr_t>(ws) & 255u)) return FASTGRNN_ERR_WORKSPACE;
  return FASTGRNN_OK;
}

// 0 = generic scan, 1 = fp32-MFMA scan, 2 = split-precision scan on the bf16 matrix pipe
int pick_path(const fastgrnn_desc* d, int direction) {
  if (d->flags & FASTGRNN_FLAG_FORCE_GENERIC) return 0;
  if (!(d->flags & FASTGRNN_FLAG_FORCE_F32_MFMA) && split_supported(*d, direction)) return 2;
  return mfma_supported(*d, direction) ? 1 : 0;
}

}  // namespace

extern "C" {

int fastgrnn_hip_abi_version(void) { return FASTGRNN_HIP_ABI_VERSION; }

const char* fastgrnn_hip_status_string(int status) {
  switch (status) {
    case FASTGRNN_OK: return "ok";
    case FASTGRNN_ERR_NULL_POINTER: return "a required pointer is NULL";
    case FASTGRNN_ERR_BAD_SHAPE: return "bad shape (T,B,F,H must be >= 1, ranks >= 0)";
    case FASTGRNN_ERR_BAD_NONLINEARITY: return "unknown nonlinearity code";
    case FASTGRNN_ERR_BAD_DTYPE: return "unsupported dtype";
    case FASTGRNN_ERR_WORKSPACE: return "workspace missing, too small or not 256-byte aligned";
    case FASTGRNN_ERR_LAUNCH: return "kernel launch failed";
    case FASTGRNN_ERR_UNSUPPORTED: return "configuration not supported";
    default: return "unknown status";
  }
}

int fastgrnn_hip_kernel_path(const fastgrnn_desc* d, int direction) {
  if (check_desc(d) != FASTGRNN_OK) return -1;
  return pick_path(d, direction);
}

size_t fastgrnn_hip_forward_workspace_bytes(const fastgrnn_desc* d) {
  if (check_desc(d) != FASTGRNN_OK) return 0;
  switch (pick_path(d, 0)) {
    case 2: return split_forward_ws(*d);
    case 1: return mfma_forward_ws(*d);
    default: return generic_forward_ws(*d);
  }
}

size_t fastgrnn_hip_backward_workspace_bytes(const fastgrnn_desc* d) {
  if (check_desc(d) != FASTGRNN_OK) return 0;
  switch (pick_path(d, 1)) {
    case 2: return split_backward_ws(*d);
    case 1: return mfma_backward_ws(*d);
    default: return generic_backward_ws(*d);
  }
}

int fastgrnn_hip_forward_unroll(const fastgrnn_desc* d, const fastgrnn_params* p, const void* x, const void* h0,
                                void* hs, void* z_s, void* c_s, void* workspace, size_t workspace_bytes,
                                void* stream) {
  int st = check_desc(d);
  if (st) return st;
  if ((st = check_params(d, p))) return st;
  if (!x || !h0 || !hs) return FASTGRNN_ERR_NULL_POINTER;
  if (((d->flags & (FASTGRNN_FLAG_SAVE_PREACT | FASTGRNN_FLAG_BATCH_MAJOR | FASTGRNN_FLAG_X_BFT | FASTGRNN_FLAG_HS_LAST)) ||
       d->dtype == FASTGRNN_BF16_IO) && pick_path(d, 0) != 2)
    return FASTGRNN_ERR_UNSUPPORTED;
  if ((d->flags & FASTGRNN_FLAG_HS_LAST) && z_s) return FASTGRNN_ERR_UNSUPPORTED;   // nothing is saved for a backward
  // (path 2 only needs its workspace when no auxiliary output is requested: see split_forward_ws)
  const size_t need = (pick_path(d, 0) == 2 && z_s && split_forward_ws_optional(*d)) ? 0 : fastgrnn_hip_forward_workspace_bytes(d);
  if ((st = check_ws(workspace, workspace_bytes, need))) return st;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (pick_path(d, 0)) {
    case 2: return split_forward(*d, *p, x, h0, hs, z_s, c_s, workspace, s);
    case 1: return mfma_forward(*d, *p, x, h0, hs, z_s, c_s, workspace, s);
    default: return generic_forward(*d, *p, x, h0, hs, z_s, c_s, workspace, s);
  }
}

int fastgrnn_hip_backward_unroll(const fastgrnn_desc* d, const fastgrnn_params* p, const void* grad_hs,
                                 const void* x, const void* hs, const void* z_s, const void* c_s, const void* h0,
                                 const fastgrnn_grads* g, void* workspace, size_t workspace_bytes, void* stream) {
  int st = check_desc(d);
  if (st) return st;
  if ((st = check_params(d, p))) return st;
  const bool preact = (d->flags & FASTGRNN_FLAG_SAVE_PREACT) != 0;
  if ((preact || (d->flags & (FASTGRNN_FLAG_BATCH_MAJOR | FASTGRNN_FLAG_X_BFT | FASTGRNN_FLAG_GRAD_LAST)) || d->dtype == FASTGRNN_BF16_IO) && pick_path(d, 1) != 2)
    return FASTGRNN_ERR_UNSUPPORTED;
  if (!grad_hs || !x || !hs || !z_s || (!c_s && !preact) || !h0 || !g) return FASTGRNN_ERR_NULL_POINTER;
  // d_x may be NULL (the input's gradient is not wanted) where it is a GEMM of its own behind the scan
  const bool dx_optional = pick_path(d, 1) == 2 && split_dx_optional(*d);
  if ((!g->d_x && !dx_optional) || !g->d_bias_gate || !g->d_bias_update || !g->d_zeta || !g->d_nu || !g->d_h0)
    return FASTGRNN_ERR_NULL_POINTER;
  if (d->w_rank ? (!g->d_w1 || !g->d_w2) : !g->d_w) return FASTGRNN_ERR_NULL_POINTER;
  if (d->u_rank ? (!g->d_u1 || !g->d_u2) : !g->d_u) return FASTGRNN_ERR_NULL_POINTER;
  if ((st = check_ws(workspace, workspace_bytes, fastgrnn_hip_backward_workspace_bytes(d)))) return st;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (pick_path(d, 1)) {
    case 2: return split_backward(*d, *p, grad_hs, x, hs, z_s, c_s, h0, *g, workspace, s);
    case 1: return mfma_backward(*d, *p, grad_hs, x, hs, z_s, c_s, h0, *g, workspace, s);
    default: return generic_backward(*d, *p, grad_hs, x, hs, z_s, c_s, h0, *g, workspace, s);
  }
}

// Single-step operators are the T = 1 case of the unrolled ones: hs[0] = new_h, and the
// backward's H_prev is old_h for t = 0 (hs itself is never read when T == 1).
int fastgrnn_hip_forward(const fastgrnn_desc* d, const fastgrnn_params* p, const void* x, const void* old_h,
                         void* new_h, void* z, void* c, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d) return FASTGRNN_ERR_NULL_POINTER;
  if (d->T != 1) return FASTGRNN_ERR_BAD_SHAPE;
  return fastgrnn_hip_forward_unroll(d, p, x, old_h, new_h, z, c, workspace, workspace_bytes, stream);
}

int fastgrnn_hip_backward(const fastgrnn_desc* d, const fastgrnn_params* p, const void* grad_h, const void* x,
                          const void* old_h, const void* z, const void* c, const fastgrnn_grads* g,
                          void* workspace, size_t workspace_bytes, void* stream) {
  if (!d) return FASTGRNN_ERR_NULL_POINTER;
  if (d->T != 1) return FASTGRNN_ERR_BAD_SHAPE;
  // hs is required non-NULL by the unrolled entry but never dereferenced at T == 1
  return fastgrnn_hip_backward_unroll(d, p, grad_h, x, /*hs=*/old_h, z, c, old_h, g, workspace, workspace_bytes,
                                      stream);
}

size_t fastgrnn_hip_head_workspace_bytes(int32_t B, int32_t H, int32_t C) {
  return head_supported(B, H, C) ? head_ws_bytes(B, H, C) : 0;
}

int fastgrnn_hip_head_xent(int32_t B, int32_t H, int32_t C, const void* h_last, const void* fc_w, const void* fc_b,
                           const int64_t* labels, void* loss, void* log_probs, void* d_h_last, void* d_fc_w,
                           void* d_fc_b, void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 1 || H < 1 || C < 1) return FASTGRNN_ERR_BAD_SHAPE;
  if (!head_supported(B, H, C)) return FASTGRNN_ERR_UNSUPPORTED;
  if (!h_last || !fc_w || !fc_b || !labels || !loss || !d_h_last || !d_fc_w || !d_fc_b) return FASTGRNN_ERR_NULL_POINTER;
  int st = check_ws(workspace, workspace_bytes, head_ws_bytes(B, H, C));
  if (st) return st;
  return head_xent(B, H, C, h_last, fc_w, fc_b, labels, loss, log_probs, d_h_last, d_fc_w, d_fc_b, workspace,
                   reinterpret_cast<hipStream_t>(stream));
}

int fastgrnn_hip_frame_gemm(size_t rows, int32_t H, int32_t F, const void* x, const void* w, void* p, int32_t dtype,
                            void* stream) {
  if (!x || !w || !p) return FASTGRNN_ERR_NULL_POINTER;
  if (rows < 1 || H < 1 || F < 1) return FASTGRNN_ERR_BAD_SHAPE;
  if (dtype != FASTGRNN_F32 && dtype != FASTGRNN_BF16_IO) return FASTGRNN_ERR_BAD_DTYPE;
  if (!rows_gemm_supported(H, F, false)) return FASTGRNN_ERR_UNSUPPORTED;
  return rows_gemm(rows, H, F, false, x, reinterpret_cast<const float*>(w), p, dtype == FASTGRNN_BF16_IO, false,
                   reinterpret_cast<hipStream_t>(stream));
}

int fastgrnn_hip_debug_poison_cu_state(uint32_t pattern, void* stream) {
  return debug_poison(pattern, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
