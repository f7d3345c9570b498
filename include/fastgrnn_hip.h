/*
 * fastgrnn_hip.h -- C ABI of libfastgrnn_hip.so: the MI355X (gfx950) FastGRNN
 * recurrent cell, forward and backward, single-step and unrolled over T frames.
 *
 * This is the drop-in boundary for the reference's native operator module
 * `fastgrnn_cuda` (/root/reference cuda/fastgrnn_cuda.cpp:235-240), whose four
 * entry points -- forward, backward, forward_unroll, backward_unroll -- are the
 * only native calls on the hot path (called from rnn.py:894,903,910,955).
 *
 * Conventions (all taken from the reference boundary, file:line cited per item):
 *   - plain device pointers + sizes; no torch types; the library allocates
 *     NOTHING and never synchronises: the caller passes outputs and a workspace
 *     (size from the *_workspace_bytes queries) and a hipStream_t.  Kernels are
 *     launched on that stream; the stream's device must be current.
 *   - every tensor is dense row-major ("contiguous", fastgrnn_cuda.cpp:69-71).
 *   - weight layout is the CUDA classes' [out,in] (rnn.py:783-798):
 *       w:[H,F]  u:[H,H]  w1:[w_rank,F]  w2:[H,w_rank]  u1:[u_rank,H]  u2:[H,u_rank]
 *       pre = x . w^T + h . u^T            (.cu:356,361,368)
 *     low-rank is evaluated FACTORISED, (x.w1^T).w2^T + (h.u1^T).u2^T, the CPU
 *     cell's association order (rnn.py:280-287); w_rank/u_rank == 0 means dense
 *     and the unused pointers may be NULL (reference: torch.empty(0),
 *     rnn.py:783-798, .cu:138-139).
 *   - zeta and nu are RAW device scalars; sigmoid is applied inside
 *     (.cu:148-149,364-365) and d_zeta/d_nu are w.r.t. the raw values
 *     (.cu:116-117).
 *   - gate code table {sigmoid:0, relu:1, tanh:2} is rnn.py:478,751; codes 3..5
 *     add the CPU cell's quantTanh/quantSigm/quantSigm4 (rnn.py:53-60; kernel path 2 under
 *     FASTGRNN_FLAG_SAVE_PREACT for the dense H=128/F=32 shape, the generic scan otherwise).  The
 *     reference's CUDA path fixes the update nonlinearity to tanh (.cu:57);
 *     update_nl is exposed because the CPU cell allows it (rnn.py:292-293).
 *   - every function returns a fastgrnn_status; nonzero means nothing useful
 *     was launched (argument errors) or the launch itself failed.
 *   - re-entrant, no global mutable state; safe from the autograd worker thread.
 */
#ifndef FASTGRNN_HIP_H
#define FASTGRNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FASTGRNN_HIP_ABI_VERSION 1

typedef enum fastgrnn_status {
  FASTGRNN_OK = 0,
  FASTGRNN_ERR_NULL_POINTER = 1,   /* a required pointer is NULL */
  FASTGRNN_ERR_BAD_SHAPE = 2,      /* T,B,F,H < 1, rank < 0, rank > dims, or size overflow */
  FASTGRNN_ERR_BAD_NONLINEARITY = 3,
  FASTGRNN_ERR_BAD_DTYPE = 4,
  FASTGRNN_ERR_WORKSPACE = 5,      /* workspace NULL/too small/misaligned (needs 256 B) */
  FASTGRNN_ERR_LAUNCH = 6,         /* hipGetLastError() != hipSuccess after a launch */
  FASTGRNN_ERR_UNSUPPORTED = 7
} fastgrnn_status;

typedef enum fastgrnn_dtype {
  FASTGRNN_F32 = 0,                /* mandatory type of the reference (AT_DISPATCH_FLOATING_TYPES, .cu:158) */
  FASTGRNN_F64 = 1,                /* gradcheck type of the reference dispatch */
  /* BASELINE config "bf16 with fp32 master grads" (new; the reference has no such type): the SEQUENCES
   * x, hs, grad_hs and d_x are bf16 (2 bytes per element); parameters, h0 / d_h0, the saved
   * pre-activation and every parameter gradient stay fp32, and so does all arithmetic (the state is
   * carried in fp32 and only its stored copy is rounded, to nearest even).  Kernel path 2, dense
   * H=128/F=32; forward without gates or with FASTGRNN_FLAG_SAVE_PREACT, backward with
   * FASTGRNN_FLAG_SAVE_PREACT; anything else answers FASTGRNN_ERR_UNSUPPORTED. */
  FASTGRNN_BF16_IO = 2
} fastgrnn_dtype;

typedef enum fastgrnn_nonlinearity {
  FASTGRNN_NL_SIGMOID = 0, FASTGRNN_NL_RELU = 1, FASTGRNN_NL_TANH = 2,       /* rnn.py:478 */
  FASTGRNN_NL_QUANT_TANH = 3, FASTGRNN_NL_QUANT_SIGM = 4, FASTGRNN_NL_QUANT_SIGM4 = 5 /* rnn.py:53-60 */
} fastgrnn_nonlinearity;

/* flags */
#define FASTGRNN_FLAG_FORCE_GENERIC 1u   /* bypass the MFMA-tiled kernels (testing / A-B) */
#define FASTGRNN_FLAG_FORCE_F32_MFMA 2u  /* use the fp32-MFMA scan instead of the split-precision one */
/* Training-only contract between forward_unroll and backward_unroll (kernel path 2 only): the
 * forward writes ONE auxiliary tensor, the pre-activation W.x_t + U.h_{t-1} (no bias), into z_s and
 * ignores c_s; the backward reads it from z_s (c_s ignored, may be NULL), recomputes z_t and
 * h_prime_t from it and therefore needs params->bias_gate / bias_update.  Saves one [T,B,H] write
 * and one read per step against the reference operator's (z_s, h_prime_s) pair. */
#define FASTGRNN_FLAG_SAVE_PREACT 4u
/* A/B only: run the dense H=128/F=32 split-precision forward in its older 4-wave shape (one wave per SIMD, two
 * row tiles per wave) instead of the default 8-wave one.  Same results to fp32 rounding.  Ignored by the other
 * shapes of kernel path 2. */
#define FASTGRNN_FLAG_FWD_4WAVE 8u
/* Batch-major sequences (the trainer's batch_first layout, rnn.py:812-813,823-825): x, hs, z_s, c_s,
 * grad_hs and d_x are [B,T,.] instead of [T,B,.]; h0/d_h0 stay [B,H].  Kernel path 2 (dense H=128 and the
 * low-rank H=256 scans; not the dense H=256 ones) -- anything else answers FASTGRNN_ERR_UNSUPPORTED and the
 * caller transposes as the reference does.  Removes the transpose(0,1).contiguous() copies around the operator. */
#define FASTGRNN_FLAG_BATCH_MAJOR 16u
/* x and d_x are [B,F,T]: what the trainer's data loader delivers and permute(2,0,1)s into a [T,B,F] VIEW
 * (trainClassifier.py:204,299) that the reference then copies with .contiguous() (rnn.py:910).  Independent
 * of FASTGRNN_FLAG_BATCH_MAJOR (which then only governs hs, the saved tensor and grad_hs).  Kernel path 2:
 * dense H=128/F=32 (read and written in place; backward under FASTGRNN_FLAG_SAVE_PREACT); dense H=256/F=32 and the
 * low-rank H=256/F=32 scans through a time-major copy in the workspace (what the reference's .contiguous() makes,
 * without the tensor it keeps alive for the backward; low-rank backward under FASTGRNN_FLAG_SAVE_PREACT). */
#define FASTGRNN_FLAG_X_BFT 128u
/* A/B only: keep the forward's state product U.h on three bf16 planes (6 MFMAs per K-step) instead of the
 * default fp16 two-plane operands with a per-wave power-of-two scale of U (3 MFMAs per K-step). */
#define FASTGRNN_FLAG_FWD_BF16X3 64u
/* SURVEY 8(f) N2 -- the classifier's view of the LAST layer (model.py:227 reads hs[T-1] alone):
 *   GRAD_LAST  backward_unroll: grad_hs is [B,H], the gradient of the last state; every other step's is zero and
 *              is neither materialised nor read (a dense zero [T,B,H] is what autograd would otherwise write and the
 *              kernel read: 2 x 208 MB at B=4096).  hs, the saved tensors and every output keep their shapes.
 *   HS_LAST    forward_unroll (inference: z_s must be NULL): hs is [B,H] and receives h_T only (a separately
 *              compiled kernel variant: equal to the last row of the full forward to fp32 rounding).
 * Both: dense H=128 (F = 32 and the wide-input layers) and low-rank H=256/F=32, any sequence layout;
 * FASTGRNN_ERR_UNSUPPORTED otherwise. */
#define FASTGRNN_FLAG_GRAD_LAST 256u
#define FASTGRNN_FLAG_HS_LAST 512u

/* Problem descriptor.  T = 1 for the single-step operators. */
typedef struct fastgrnn_desc {
  int32_t T, B, F, H;       /* frames, utterances, features per frame, hidden units */
  int32_t w_rank, u_rank;   /* 0 = dense */
  int32_t gate_nl;          /* fastgrnn_nonlinearity */
  int32_t update_nl;        /* fastgrnn_nonlinearity; the reference CUDA boundary means TANH */
  int32_t dtype;            /* fastgrnn_dtype */
  uint32_t flags;
} fastgrnn_desc;

/* Parameters (device pointers, boundary layout above). */
typedef struct fastgrnn_params {
  const void *w, *u;                 /* dense matrices or NULL */
  const void *w1, *w2, *u1, *u2;     /* factors or NULL */
  const void *bias_gate;             /* [1,H]  (bias_z) */
  const void *bias_update;           /* [1,H]  (bias_h_prime) */
  const void *zeta, *nu;             /* [1,1] raw */
} fastgrnn_params;

/* Gradient outputs, the 12-tuple of .cu:556 in that order.  d_w/d_u are written
 * for dense operands, d_w1,d_w2 / d_u1,d_u2 for factorised ones; the others are
 * ignored and may be NULL (reference returns torch::empty(0), .cu:221-224). */
typedef struct fastgrnn_grads {
  void *d_x;            /* [T,B,F]; may be NULL on kernel path 2 for dense H=256 and dense H=128 with F > 32 (the
                           input's gradient is then not computed: one GEMM less -- a model's first layer) */
  void *d_bias_gate;    /* [1,H] */
  void *d_bias_update;  /* [1,H] */
  void *d_zeta;         /* [1,1] */
  void *d_nu;           /* [1,1] */
  void *d_h0;           /* [B,H]  (d_old_h) */
  void *d_w, *d_u;      /* [H,F], [H,H] */
  void *d_w1, *d_w2;    /* [w_rank,F], [H,w_rank] */
  void *d_u1, *d_u2;    /* [u_rank,H], [H,u_rank] */
} fastgrnn_grads;

int fastgrnn_hip_abi_version(void);
const char *fastgrnn_hip_status_string(int status);

/* Which kernel family a descriptor dispatches to: 0 = generic LDS/VALU scan,
 * 1 = fp32-MFMA scan (v_mfma_f32_16x16x4_f32; 16 utterances per workgroup, U in registers),
 * 2 = split-precision scan: every fp32 operand as three exact bf16 planes, six
 *     v_mfma_f32_16x16x32_bf16 terms per product, fp32 accumulation (error O(2^-24)).
 * direction: 0 forward, 1 backward.  Pure function of the descriptor.
 *
 * Shapes on path 2 (fp32 or bf16 sequences unless noted; everything else runs on paths 1 / 0, 20-30x slower at
 * B = 4096 -- ask this function before assuming):
 *   dense  H=128, F=32            every gate; update tanh or quantTanh (quantTanh: fp32, SAVE_PREACT backward);
 *                                 all layout flags.  Backward with the reference's (z_s, h_prime_s) tensors for the
 *                                 sigmoid / relu / tanh gates, otherwise under FASTGRNN_FLAG_SAVE_PREACT.
 *   dense  H=128, F=64/128/256    (the reference's second layer) time- or batch-major; last-state flags (fp32).
 *                                 bf16 sequences: gates sigmoid / relu / tanh, no last-state flags, backward under
 *                                 FASTGRNN_FLAG_SAVE_PREACT.
 *   dense  H=256, F=32            (the reference's first layer) time- or batch-major (two-stride rows in
 *                                 both scans, the dU GEMM pairs row b*T+t with hs row b*T+t-1 and every T-th row
 *                                 with h0); x in the sequences' layout or the
 *                                 loader's [B,F,T] (FASTGRNN_FLAG_X_BFT: transposed into the workspace by both
 *                                 calls, 25 us at B = 4096; d_x comes back as [B,F,T]; the backward with BOTH
 *                                 FASTGRNN_FLAG_X_BFT and FASTGRNN_FLAG_BATCH_MAJOR is not on path 2); last-state
 *                                 flags; gates sigmoid / relu / tanh with the reference's (z_s, h_prime_s)
 *                                 tensors, every gate under FASTGRNN_FLAG_SAVE_PREACT.  bf16 sequences: gates
 *                                 sigmoid / relu / tanh; forward with hs alone or under FASTGRNN_FLAG_SAVE_PREACT
 *                                 (time- or batch-major), backward under it (time-major); no [B,F,T] frames, no
 *                                 FASTGRNN_FLAG_HS_LAST.
 *   dense  H=256, F=64/128        (F = 64: the reference's DEFAULT first layer, feature_type='delta' = 32 MFCCs + 32
 *                                 deltas, trainingConfig.py:36, mfccProcessor.py:27-28) time- or batch-major; as F=32
 *                                 but without FASTGRNN_FLAG_X_BFT: the frame product X.W^T is one batched GEMM into
 *                                 the workspace (T*B*256*4 bytes more of it) in front of the scan.
 *   low-rank H=256, F=32, both W and U factorised with 1 <= rank <= 16 (the two ranks may differ; ranks are
 *                                 zero-extended to 16 inside the kernels): gates sigmoid / relu / tanh; all layout
 *                                 flags; backward under FASTGRNN_FLAG_SAVE_PREACT only.
 *   every other factorised cell on one of the dense shapes above (H=128 with any ranks; H=256: F=32 with a rank of
 *                                 17..256 or with only one of W, U factorised, F=64/128 with any ranks:
 *                                 rnn.py:783-798): the factors are
 *                                 multiplied out per call, the dense kernels run, the dense gradients are
 *                                 projected onto the factors (as the reference's CUDA operator does for every
 *                                 low-rank cell, .cu:353-362,546-555); the dense shape's limits and flags apply.
 *                                 No rank-space vector is saved (c_s is ignored under FASTGRNN_FLAG_SAVE_PREACT).
 * Under FASTGRNN_FLAG_SAVE_PREACT a factorised forward with both ranks in 1..16 also writes, through c_s, the rank-space vector
 * [U1.h_{t-1} | W1.x_t] as a time-major fp32 [T*B, 32] tensor (each half zero-extended to 16 columns) that the
 * backward takes back through c_s (with z_s, the pre-activation): its factor gradients are contracted inside the
 * scan, d_u2|d_w2 against exactly this vector.  (Round 2's option of passing z_s = NULL and having the backward
 * recompute the pre-activation from c_s is gone: the scan that also contracts the factor gradients has neither the
 * registers nor the LDS for a second copy of [U2|W2]; z_s = NULL is FASTGRNN_ERR_NULL_POINTER again.) */
int fastgrnn_hip_kernel_path(const fastgrnn_desc *d, int direction);

/* Workspace sizes in bytes (0 is a valid answer).  Workspace must be 256-B aligned.  The forward answer covers a
 * call without auxiliary outputs; dense H=128 layers with F > 32 park the frame product X.W^T in z_s / c_s when
 * the caller passes them and then accept workspace == NULL. */
size_t fastgrnn_hip_forward_workspace_bytes(const fastgrnn_desc *d);
size_t fastgrnn_hip_backward_workspace_bytes(const fastgrnn_desc *d);

/* forward_unroll -- replaces fastgrnn_unroll_forward (fastgrnn_cuda.cpp:147-180 ->
 * .cu:320-415).  x:[T,B,F], h0:[B,H] -> hs:[T,B,H]; z_s,c_s:[T,B,H] are the
 * reference's z_s / h_prime_s outputs and may be NULL when the caller does not
 * need them (forward-only use). */
int fastgrnn_hip_forward_unroll(const fastgrnn_desc *d, const fastgrnn_params *p,
                                const void *x, const void *h0,
                                void *hs, void *z_s, void *c_s,
                                void *workspace, size_t workspace_bytes, void *stream);

/* backward_unroll -- replaces fastgrnn_unroll_backward (fastgrnn_cuda.cpp:182-232 ->
 * .cu:417-557).  grad_hs:[T,B,H] is dL/d(hs[t]) for every t; z_s,c_s are the
 * forward's outputs.  Writes every applicable member of *g (overwrites, does not
 * accumulate). */
int fastgrnn_hip_backward_unroll(const fastgrnn_desc *d, const fastgrnn_params *p,
                                 const void *grad_hs, const void *x, const void *hs,
                                 const void *z_s, const void *c_s, const void *h0,
                                 const fastgrnn_grads *g,
                                 void *workspace, size_t workspace_bytes, void *stream);

/* forward -- replaces fastgrnn_forward (fastgrnn_cuda.cpp:73-107 -> .cu:123-198).
 * x:[B,F], old_h:[B,H] -> new_h, z, c each [B,H].  d->T must be 1. */
int fastgrnn_hip_forward(const fastgrnn_desc *d, const fastgrnn_params *p,
                         const void *x, const void *old_h,
                         void *new_h, void *z, void *c,
                         void *workspace, size_t workspace_bytes, void *stream);

/* backward -- replaces fastgrnn_backward (fastgrnn_cuda.cpp:109-145 -> .cu:200-318).
 * grad_h:[B,H]; g->d_x is [B,F], g->d_h0 is d_old_h.  d->T must be 1. */
int fastgrnn_hip_backward(const fastgrnn_desc *d, const fastgrnn_params *p,
                          const void *grad_h, const void *x, const void *old_h,
                          const void *z, const void *c,
                          const fastgrnn_grads *g,
                          void *workspace, size_t workspace_bytes, void *stream);

/* ---- classifier head on the last state ------------------------------------------------------
 * head_xent -- replaces, for training, the three torch modules the reference chains after the last layer:
 *   keyword_scores = F.log_softmax(self.hidden2keyword(hs[T-1]), dim=1)     (model.py:86-88, 226-230)
 *   loss = nn.NLLLoss()(keyword_scores, labels)                              (trainClassifier.py:154,236; mean)
 * and their backward.  fp32.  h_last:[B,H] (e.g. the FASTGRNN_FLAG_HS_LAST / last_state output), fc_w:[C,H],
 * fc_b:[C] (nn.Linear layout), labels:[B] int64 in [0,C) or -100 (nn.NLLLoss's default ignore_index: such rows add
 * neither loss nor gradient and the mean runs over the others; any other out-of-range label makes the loss NaN,
 * where torch raises a device-side assert).  Writes loss[1], log_probs:[B,C] (may be NULL),
 * d_h_last:[B,H] = dLoss/dh_last, d_fc_w:[C,H], d_fc_b:[C] (overwritten).  Deterministic (fixed-order reduction).
 * H <= 256, C <= 64, else FASTGRNN_ERR_UNSUPPORTED.  workspace: fastgrnn_hip_head_workspace_bytes(B,H,C). */
size_t fastgrnn_hip_head_workspace_bytes(int32_t B, int32_t H, int32_t C);
int fastgrnn_hip_head_xent(int32_t B, int32_t H, int32_t C, const void *h_last, const void *fc_w,
                           const void *fc_b, const int64_t *labels, void *loss, void *log_probs,
                           void *d_h_last, void *d_fc_w, void *d_fc_b,
                           void *workspace, size_t workspace_bytes, void *stream);

/* frame_gemm -- the one genuinely dense, non-recurrent contraction of a layer, as a call of its own:
 *   P[rows, H] = X[rows, F] . W^T,   W:[H,F]  (rows = T*B; the reference computes it per step, `mm` at .cu:356).
 * forward_unroll runs exactly this launch in front of the scan for layers whose input is wider than 32 (the scan of a
 * 32-feature layer has the product fused); it is exported so that its MFMA utilisation can be measured and reported by
 * itself (bench.py: `wx_gemm`).  fp32 (dtype FASTGRNN_F32) or bf16 x with fp32 p (FASTGRNN_BF16_IO); (H, F) one of
 * the wide-layer shapes of the table above, else FASTGRNN_ERR_UNSUPPORTED.  No workspace. */
int fastgrnn_hip_frame_gemm(size_t rows, int32_t H, int32_t F, const void *x, const void *w, void *p, int32_t dtype,
                            void *stream);

/* Test hook, not part of the reference boundary: one launch that leaves `pattern` in every CU's LDS and vector
 * registers (on-chip state is not cleared between kernels).  tests/test_hip_state_independence.py runs it before the
 * operators with a NaN pattern: results must not change, i.e. no kernel reads LDS or registers it did not write. */
int fastgrnn_hip_debug_poison_cu_state(uint32_t pattern, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FASTGRNN_HIP_H */
