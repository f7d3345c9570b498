// Classifier head on the last hidden state: keyword scores = log_softmax(Linear(h_T)) and their
// NLL loss, with every gradient, in one pass.  gfx950 only.
//   reference: model.py:86-88 (hidden2keyword = nn.Linear(H, C)), model.py:226-230 (Linear on hs[T-1], then
//   F.log_softmax(dim=1)), trainClassifier.py:154,236 (nn.NLLLoss(), mean over the batch).
// The loss is always differentiated in training, so the backward is computed beside the forward (the classic
// fused softmax-cross-entropy): one launch produces log-probabilities, the loss and d_h / d_W / d_b per
// workgroup, a second one reduces the per-workgroup partials in a fixed order (deterministic, no atomics).
// HBM-bound integer/float byte work: B*H*4 bytes in, B*H*4 out; nothing here belongs on the matrix pipe
// (2*B*H*C flops, C ~ 12).
#include "common.h"

namespace fastgrnn {
namespace {

constexpr int HEAD_UTT = 16;        // utterances per workgroup (B = 4096 -> 256 workgroups, one per CU)
constexpr int HEAD_THREADS = 256;
constexpr int HEAD_MAX_C = 64, HEAD_MAX_H = 256;
constexpr long long HEAD_IGNORE_INDEX = -100;   // torch.nn.NLLLoss default

// dynamic LDS: hT[HEAD_UTT][H+1] | W[C][H+1] | dl[HEAD_UTT][C+1]
__global__ __launch_bounds__(HEAD_THREADS) void head_xent_fwd_bwd(
    int B, int H, int C, const float* __restrict__ h_last, const float* __restrict__ fc_w,
    const float* __restrict__ fc_b, const long long* __restrict__ labels,
    float* __restrict__ logp, float* __restrict__ d_h, float* __restrict__ part) {
  extern __shared__ float lds[];
  const int HP = H + 1, CP = C + 1;
  float* sh = lds;                               // [HEAD_UTT][HP]
  float* sw = sh + HEAD_UTT * HP;                // [C][HP]
  float* sdl = sw + C * HP;                      // [HEAD_UTT][CP]: logits, then d_logits
  float* sloss = sdl + HEAD_UTT * CP;            // [HEAD_UTT]
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * HEAD_UTT;
  const int nu = min(HEAD_UTT, B - b0);

  // nn.NLLLoss() semantics (trainClassifier.py:154: default reduction 'mean', ignore_index = -100): rows whose
  // label is -100 contribute neither loss nor gradient and the mean runs over the other rows.  Every workgroup
  // counts them itself (B labels from L2, fixed order: deterministic).  Any other label outside [0,C) is an
  // error in torch (device-side assert); here it makes the loss NaN instead of silently dropping the row.
  __shared__ int s_cnt[HEAD_THREADS / 64];
  {
    int cnt = 0;
    for (int e = tid; e < B; e += HEAD_THREADS) cnt += (labels[e] != HEAD_IGNORE_INDEX) ? 1 : 0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) cnt += __shfl_xor(cnt, m);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = cnt;
  }
  __syncthreads();
  int n_rows = 0;
#pragma unroll
  for (int k = 0; k < HEAD_THREADS / 64; ++k) n_rows += s_cnt[k];
  const float inv_B = n_rows > 0 ? 1.0f / (float)n_rows : __builtin_nanf("");   // torch: mean over zero rows is NaN

  // ---- stage the tile of h_T (coalesced rows) and the weights
  for (int e = tid; e < HEAD_UTT * H; e += HEAD_THREADS) {
    const int u = e / H, n = e - u * H;
    sh[u * HP + n] = (u < nu) ? h_last[(size_t)(b0 + u) * H + n] : 0.f;
    lds_writes_landed();                           // (second rule of DESIGN.md 4.0: the next iteration's load)
  }
  for (int e = tid; e < C * H; e += HEAD_THREADS) {
    const int c = e / H, n = e - c * H;
    sw[c * HP + n] = fc_w[e];
    lds_writes_landed();                           // (second rule of DESIGN.md 4.0: the next iteration's load)
  }
  __syncthreads();
  // ---- logits[u][c] = b[c] + W[c,:] . h[u,:]            (model.py:227)
  for (int e = tid; e < HEAD_UTT * C; e += HEAD_THREADS) {
    const int u = e / C, c = e - u * C;
    float acc = fc_b[c];
    const float* hp = sh + u * HP;
    const float* wp = sw + c * HP;
    for (int n = 0; n < H; ++n) acc = fmaf(wp[n], hp[n], acc);
    sdl[u * CP + c] = acc;
    lds_writes_landed();                           // (second rule of DESIGN.md 4.0: the next iteration's load)
  }
  __syncthreads();
  // ---- per utterance: log_softmax (model.py:229), NLL term and d_logits = (softmax - onehot) / B
  if (tid < HEAD_UTT) {
    const int u = tid;
    float* row = sdl + u * CP;
    float m = row[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(row[c] - m);
    const float lse = m + logf(s);
    float l = 0.f;
    if (u < nu) {
      const long long y = labels[b0 + u];
      const bool ignored = y == HEAD_IGNORE_INDEX;
      if (!ignored && (y < 0 || y >= C)) l = __builtin_nanf("");      // invalid label: loud, not dropped
      for (int c = 0; c < C; ++c) {
        const float lp = row[c] - lse;
        if (logp) logp[(size_t)(b0 + u) * C + c] = lp;
        if (c == (int)y) l = -lp;                                       // trainClassifier.py:236
        row[c] = ignored ? 0.f : (expf(lp) - (c == (int)y ? 1.f : 0.f)) * inv_B;
        lds_writes_landed();                       // (the next class's conditional logp store)
      }
    } else {
      for (int c = 0; c < C; ++c) { row[c] = 0.f; lds_writes_landed(); }
    }
    sloss[u] = l;
  }
  __syncthreads();
  // ---- d_h[u][n] = sum_c d_logits[u][c] W[c][n]
  for (int e = tid; e < HEAD_UTT * H; e += HEAD_THREADS) {
    const int u = e / H, n = e - u * H;
    if (u < nu) {
      float acc = 0.f;
      for (int c = 0; c < C; ++c) acc = fmaf(sdl[u * CP + c], sw[c * HP + n], acc);
      d_h[(size_t)(b0 + u) * H + n] = acc;
    }
  }
  // ---- partials of this workgroup: d_W[c][n] | d_b[c] | loss
  float* pw = part + (size_t)blockIdx.x * (C * H + C + 1);
  for (int e = tid; e < C * H; e += HEAD_THREADS) {
    const int c = e / H, n = e - c * H;
    float acc = 0.f;
    for (int u = 0; u < HEAD_UTT; ++u) acc = fmaf(sdl[u * CP + c], sh[u * HP + n], acc);
    pw[e] = acc;
  }
  if (tid < C) {
    float acc = 0.f;
    for (int u = 0; u < HEAD_UTT; ++u) acc += sdl[u * CP + tid];
    pw[C * H + tid] = acc;
  }
  if (tid == 0) {
    float acc = 0.f;
    for (int u = 0; u < HEAD_UTT; ++u) acc += sloss[u];
    pw[C * H + C] = acc * inv_B;
  }
}

// out[e] = sum over workgroups of part[wg][e], e < C*H + C + 1 (d_W | d_b | loss), in a fixed order: a workgroup
// owns 32 outputs; its 8 thread rows each sum every 8th workgroup's partial (32 independent loads in flight per
// thread instead of one dependent chain of nwg), then the 8 row sums are added in row order.
__global__ __launch_bounds__(256) void head_reduce(int nwg, int n, int CH, int C, const float* __restrict__ part,
                                                   float* __restrict__ d_w, float* __restrict__ d_b,
                                                   float* __restrict__ loss) {
  __shared__ float rows[8][32];
  const int col = threadIdx.x & 31, row = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + col;
  float acc = 0.f;
  if (e < n) {
#pragma unroll 8
    for (int w = row; w < nwg; w += 8) acc += part[(size_t)w * n + e];
  }
  rows[row][col] = acc;
  __syncthreads();
  if (row == 0 && e < n) {
    float t = rows[0][col];
#pragma unroll
    for (int r = 1; r < 8; ++r) t += rows[r][col];
    if (e < CH) d_w[e] = t;
    else if (e < CH + C) d_b[e - CH] = t;
    else loss[0] = t;
  }
}

size_t head_lds_bytes(int H, int C) {
  return (size_t)(HEAD_UTT * (H + 1) + C * (H + 1) + HEAD_UTT * (C + 1) + HEAD_UTT) * sizeof(float);
}

}  // namespace

bool head_supported(int B, int H, int C) { return B >= 1 && H >= 1 && H <= HEAD_MAX_H && C >= 1 && C <= HEAD_MAX_C; }

size_t head_ws_bytes(int B, int H, int C) {
  return align256((size_t)((B + HEAD_UTT - 1) / HEAD_UTT) * (C * H + C + 1) * sizeof(float));
}

int head_xent(int B, int H, int C, const void* h_last, const void* fc_w, const void* fc_b, const void* labels,
              void* loss, void* logp, void* d_h, void* d_w, void* d_b, void* ws, hipStream_t s) {
  const int nwg = (B + HEAD_UTT - 1) / HEAD_UTT;
  const size_t lds = head_lds_bytes(H, C);
  // > 64 KB of dynamic LDS needs the opt-in; the attribute is per device, so it is set before every launch
  // that needs it (no process-wide state: the library is re-entrant and device-agnostic)
  if (lds > 64 * 1024)                              // (what this launch needs, not the CU's 160 KB: the kernel also has static LDS)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(head_xent_fwd_bwd),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float* part = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(head_xent_fwd_bwd, dim3(nwg), dim3(HEAD_THREADS), lds, s, B, H, C, (const float*)h_last,
                     (const float*)fc_w, (const float*)fc_b, (const long long*)labels, (float*)logp,
                     (float*)d_h, part);
  const int n = C * H + C + 1;
  hipLaunchKernelGGL(head_reduce, dim3((n + 31) / 32), dim3(256), 0, s, nwg, n, C * H, C, (const float*)part,
                     (float*)d_w, (float*)d_b, (float*)loss);
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

}  // namespace fastgrnn
