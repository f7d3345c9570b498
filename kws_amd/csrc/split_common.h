// Split-precision building blocks shared by kernels_split.hip (the scans) and kernels_gemm.hip (the batched
// row / TN GEMMs around them): exact three-bf16-plane and two-fp16-plane splits, the six- / three-term MFMA
// products, the hardware-transposed LDS fragment read.  gfx950 only.  See kernels_split.hip's header for why
// fp32 operands go to the bf16 matrix pipe this way, and DESIGN.md 4.0 for the operand rule the helpers'
// users are written to.
#pragma once
#include "common.h"
#include <type_traits>
#include <utility>

namespace fastgrnn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ float fsigmoid(float a) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * a));
}
__device__ __forceinline__ float ftanh(float a) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((2.0f * LOG2E) * a));
}
// gate nonlinearities: the reference's GPU table {sigmoid, relu, tanh} (rnn.py:478) and, on the 8-wave dense
// kernels, the CPU cell's quantised family (rnn.py:53-60); derivatives through the output as in
// common.h / .cu:27-40
template <int GATE> __device__ __forceinline__ float gate_act(float a) {
  if (GATE == FASTGRNN_NL_SIGMOID) return fsigmoid(a);
  if (GATE == FASTGRNN_NL_RELU) return a > 0.0f ? a : 0.0f;
  if (GATE == FASTGRNN_NL_QUANT_TANH) return fminf(fmaxf(a, -1.0f), 1.0f);
  if (GATE == FASTGRNN_NL_QUANT_SIGM) return fminf(fmaxf((a + 1.0f) * 0.5f, 0.0f), 1.0f);
  if (GATE == FASTGRNN_NL_QUANT_SIGM4) return fminf(fmaxf((a + 2.0f) * 0.25f, 0.0f), 1.0f);
  return ftanh(a);
}
template <int GATE> __device__ __forceinline__ float gate_dact(float y) {
  if (GATE == FASTGRNN_NL_SIGMOID) return (1.0f - y) * y;
  if (GATE == FASTGRNN_NL_RELU) return y > 0.0f ? 1.0f : 0.0f;
  if (GATE == FASTGRNN_NL_QUANT_TANH) return (y < 1.0f && y > -1.0f) ? 1.0f : 0.0f;
  if (GATE == FASTGRNN_NL_QUANT_SIGM) return (y < 1.0f && y > 0.0f) ? 0.5f : 0.0f;
  if (GATE == FASTGRNN_NL_QUANT_SIGM4) return (y < 1.0f && y > 0.0f) ? 0.25f : 0.0f;
  return 1.0f - y * y;
}

// compensated (Kahan) running sum: s += y with the rounding error of every addition carried in c.  Used for the two
// scalar gradients d_zeta / d_nu, whose per-lane sums run over T * (elements per lane) terms of either sign: a plain
// fp32 running sum is good to ~sqrt(n) * 6e-8 of the partial sum, which showed as 2.5e-5 of the result at B = 64.
__device__ __forceinline__ void kahan_add(float& s, float& c, float y) {
  const float yy = y - c;
  const float t = s + yy;
  c = (t - s) - yy;
  s = t;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// bf16 I/O variant (FASTGRNN_BF16_IO): sequences x / hs / grad_hs / d_x are bf16 in HBM, everything else fp32
__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
// round to nearest even on the hardware converter (v_cvt_pk_bf16_f32): a NaN stays a NaN (the integer form
// (u + 0x7FFF + lsb) >> 16 turns some NaNs into 0 / inf -- MI355X_MICROARCH.md, correctness boundaries)
__device__ __forceinline__ unsigned f32_to_bf16_rne(float f) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ f32x4 ld4_bf16(const void* p) {               // 4 consecutive bf16 -> 4 floats
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xFFFF0000u),
               __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xFFFF0000u)};
}
__device__ __forceinline__ void st4_bf16(void* p, const f32x4 v) {
  typedef __bf16 bf16x2_cv __attribute__((ext_vector_type(2)));
  typedef float f32x2_cv __attribute__((ext_vector_type(2)));
  const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_cv{v[0], v[1]}, bf16x2_cv));
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_cv{v[2], v[3]}, bf16x2_cv));
  *reinterpret_cast<uint2*>(p) = uint2{lo, hi};
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#ifdef FASTGRNN_DIAG_STAMPS
// Diagnostic build only (tools/diag_split.hip): per-segment cycle sums of each wave of block 7.
__device__ unsigned long long g_sdiag[8][8];
#define SPLIT_STAMP(idx)                                                                  \
  {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    unsigned long long now_;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    dsum[idx] += now_ - dlast; dlast = now_;                                              \
  }
#else
#define SPLIT_STAMP(idx)
#endif

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                 0, 0);
}

// The 4-wave scans are written for ONE wave per SIMD: a second workgroup on the same CU (possible for the
// leaner instantiations once B > 4096) would put another wave's MFMAs between an MFMA's issue and its
// operand fetch, and the compiler reloads fragment registers right behind the MFMAs that read them
// (see bwd_scan_split_w8::weight_grads).  Pin them to one wave per SIMD.
#define ONE_WAVE_PER_SIMD __attribute__((amdgpu_waves_per_eu(1, 1)))

// Three bf16 planes of 8 fp32 values (one MFMA fragment each).  Exact: p0+p1+p2 == v.
struct Frag3 { u32x4 p[3]; };

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned x) { return __builtin_bit_cast(float, x); }
// {hi16(b), hi16(a)} -> one dword of two bf16 (element order a, b)
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// ---- the exact three-plane split -----------------------------------------------------------------------
// v = p0 + p1 + p2 with p0 = bf16(v), p1 = bf16(v - p0), p2 = v - p0 - p1 (<= 8 significant bits left: exact),
// every conversion ROUND-TO-NEAREST-EVEN on the hardware converter (v_cvt_pk_bf16_f32, two values per
// instruction).  Rounding, not truncating, matters: truncated planes all carry the sign of v, so the three
// dropped cross terms (p1.q2 + p2.q1 + p2.q2) pushed every product toward zero by ~4e-8 relative -- harmless
// per product, but one-signed, and it showed in the two scalar gradients that sum a million terms (d_zeta,
// d_nu: 3e-4 relative against 1e-5 for the fp32 paths).  With rounded planes the
// dropped terms are zero-mean and below 2^-26.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {            // {bf16(b), bf16(a)}: element order a, b
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
// two values -> one dword (two bf16) per plane
__device__ __forceinline__ void split_pair(float a, float b, unsigned& q0, unsigned& q1, unsigned& q2) {
  q0 = pk_bf16(a, b);
  const float ra = a - bitsf(q0 << 16), rb = b - bitsf(q0 & 0xFFFF0000u);
  q1 = pk_bf16(ra, rb);
  q2 = pk_bf16(ra - bitsf(q1 << 16), rb - bitsf(q1 & 0xFFFF0000u));
}
// four values -> 8 bytes (four bf16) per plane
__device__ __forceinline__ void split_quad(const f32x4 v, uint2& p0, uint2& p1, uint2& p2) {
  split_pair(v[0], v[1], p0.x, p1.x, p2.x);
  split_pair(v[2], v[3], p0.y, p1.y, p2.y);
}
// one value -> one bf16 per plane
__device__ __forceinline__ void split_one(float v, unsigned short& s0, unsigned short& s1, unsigned short& s2) {
  unsigned q0, q1, q2;
  split_pair(v, 0.0f, q0, q1, q2);
  s0 = (unsigned short)q0; s1 = (unsigned short)q1; s2 = (unsigned short)q2;
}
__device__ __forceinline__ Frag3 split3(const f32x4 lo, const f32x4 hi) {
  uint2 a0, a1, a2, b0, b1, b2;
  split_quad(lo, a0, a1, a2);
  split_quad(hi, b0, b1, b2);
  Frag3 f;
  f.p[0] = u32x4{a0.x, a0.y, b0.x, b0.y};
  f.p[1] = u32x4{a1.x, a1.y, b1.x, b1.y};
  f.p[2] = u32x4{a2.x, a2.y, b2.x, b2.y};
  return f;
}

// fp16 two-plane operands (forward state product only, see fwd_scan_split_w8): v = hi + lo with hi = fp16(v),
// lo = fp16(v - hi): 22-23 significant bits while v is in fp16's normal range
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
struct Frag2h { u32x4 hi, lo; };
__device__ __forceinline__ void split2h(float a, float b, unsigned& hi, unsigned& lo) {   // two values -> one dword per plane
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;                                       // round to nearest even
  const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
  hi = __builtin_bit_cast(unsigned, f16x2{ha, hb});
  lo = __builtin_bit_cast(unsigned, f16x2{la, lb});
}
// three fp16 planes: exact for a value scaled into fp16's normal range (11 + 11 + 11 bits cover fp32's 24)
__device__ __forceinline__ void split3h(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;
  const float ra = a - (float)ha, rb = b - (float)hb;
  const _Float16 la = (_Float16)ra, lb = (_Float16)rb;
  const _Float16 ma = (_Float16)(ra - (float)la), mb = (_Float16)(rb - (float)lb);
  p0 = __builtin_bit_cast(unsigned, f16x2{ha, hb});
  p1 = __builtin_bit_cast(unsigned, f16x2{la, lb});
  p2 = __builtin_bit_cast(unsigned, f16x2{ma, mb});
}
__device__ __forceinline__ Frag2h split2h8(const f32x4 lo4, const f32x4 hi4) {
  unsigned h[4], l[4];
  split2h(lo4[0], lo4[1], h[0], l[0]); split2h(lo4[2], lo4[3], h[1], l[1]);
  split2h(hi4[0], hi4[1], h[2], l[2]); split2h(hi4[2], hi4[3], h[3], l[3]);
  Frag2h f;
  f.hi = u32x4{h[0], h[1], h[2], h[3]}; f.lo = u32x4{l[0], l[1], l[2], l[3]};
  return f;
}
__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// acc += A.B with the three retained plane pairs (lo.lo, 2^-22 relative, dropped; small terms first)
__device__ __forceinline__ f32x4 mfma3h(const Frag2h& a, const Frag2h& b, f32x4 acc) {
  acc = mfma_f16(a.lo, b.hi, acc);
  acc = mfma_f16(a.hi, b.lo, acc);
  acc = mfma_f16(a.hi, b.hi, acc);
  return acc;
}

__device__ __forceinline__ void mfma3h_hl(const Frag2h& a, const Frag2h& b, f32x4& hi, f32x4& lo) {   // see mfma6_hl
  lo = mfma_f16(a.lo, b.hi, lo);
  lo = mfma_f16(a.hi, b.lo, lo);
  hi = mfma_f16(a.hi, b.hi, hi);
}

// acc += sum over the six retained plane pairs of A[pa] . B[pb]   (small terms first)
__device__ __forceinline__ f32x4 mfma6(const Frag3& a, const Frag3& b, f32x4 acc) {
  acc = mfma_bf16(a.p[2], b.p[0], acc);
  acc = mfma_bf16(a.p[1], b.p[1], acc);
  acc = mfma_bf16(a.p[0], b.p[2], acc);
  acc = mfma_bf16(a.p[1], b.p[0], acc);
  acc = mfma_bf16(a.p[0], b.p[1], acc);
  acc = mfma_bf16(a.p[0], b.p[0], acc);
  return acc;
}

// The same six terms with the five small ones in an accumulator of their own (lo, started at zero by the caller
// and added to hi at the end): inside one MFMA every product is aligned to the largest addend -- including C --
// and chopped there, so small-term products added straight into a large running sum lose their low bits
// (tools/mfma_round_probe.hip).
__device__ __forceinline__ void mfma6_hl(const Frag3& a, const Frag3& b, f32x4& hi, f32x4& lo) {
  lo = mfma_bf16(a.p[2], b.p[0], lo);
  lo = mfma_bf16(a.p[1], b.p[1], lo);
  lo = mfma_bf16(a.p[0], b.p[2], lo);
  lo = mfma_bf16(a.p[1], b.p[0], lo);
  lo = mfma_bf16(a.p[0], b.p[1], lo);
  hi = mfma_bf16(a.p[0], b.p[0], hi);
}

// K = 16 form (v_mfma_f32_16x16x16_bf16: lane group g holds k = 4g..4g+3, two registers per operand) of the six-term
// product, for contractions whose K is the 16 utterances of a workgroup: one hardware-transposed LDS read
// (ds_read_b64_tr_b16) returns exactly one plane of such an operand.  Same issue rate as the K = 32 form.
struct Half3 { s16x4 p[3]; };
__device__ __forceinline__ f32x4 mfma_bf16_k16(s16x4 a, s16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma6_k16(const Half3& a, const Half3& b, f32x4 acc) {
  acc = mfma_bf16_k16(a.p[2], b.p[0], acc);
  acc = mfma_bf16_k16(a.p[1], b.p[1], acc);
  acc = mfma_bf16_k16(a.p[0], b.p[2], acc);
  acc = mfma_bf16_k16(a.p[1], b.p[0], acc);
  acc = mfma_bf16_k16(a.p[0], b.p[1], acc);
  acc = mfma_bf16_k16(a.p[0], b.p[0], acc);
  return acc;
}

__device__ __forceinline__ void mfma6_hl_k16(const Half3& a, const Half3& b, f32x4& hi, f32x4& lo) {   // see mfma6_hl
  lo = mfma_bf16_k16(a.p[2], b.p[0], lo);
  lo = mfma_bf16_k16(a.p[1], b.p[1], lo);
  lo = mfma_bf16_k16(a.p[0], b.p[2], lo);
  lo = mfma_bf16_k16(a.p[1], b.p[0], lo);
  lo = mfma_bf16_k16(a.p[0], b.p[1], lo);
  hi = mfma_bf16_k16(a.p[0], b.p[0], hi);
}

// Keeps a fragment's registers allocated up to this point, ordered after whatever produced `tie` (pass a value
// read from the youngest accumulator: once that has been read the matrix pipe has drained).  No instruction.
__device__ __forceinline__ void keep_alive(float& tie, const Frag3& f) {
  asm volatile("" : "+v"(tie) : "v"(f.p[0]), "v"(f.p[1]), "v"(f.p[2]));
}

template <typename Fn, int... Is>
__device__ __forceinline__ void static_for_impl(Fn&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
// compile-time loop: the index is a constant expression inside the body ("i" asm operands need that)
template <int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }


// One transposed fragment (8 bf16: utterances +0..7 of this lane's 8-row block, one unit column) =
// two ds_read_b64_tr_b16 (rows +0..3, +4..7), through the compiler builtin so that hipcc tracks
// their lgkmcnt and registers itself.  EXEC is all ones wherever this is used (the transpose
// gathers across the 16 lanes of a group).
__device__ __forceinline__ u32x4 tr_frag(unsigned lds_byte_addr, int rowb) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>((size_t)lds_byte_addr));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>((size_t)(lds_byte_addr + 4 * rowb)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(u32x4, v);
}

}  // namespace
}  // namespace fastgrnn
