// Low-precision-MFMA building blocks shared by the split-precision kernels (slu_gemm_bf16.hip, slu_gru_bf16.hip,
// slu_wconv_bf16.hip, slu_pool.hip).  A "split scheme" NS writes every fp32 operand as NS 16-bit terms
// ("planes") and forms the contraction from a few 16-bit MFMA products with fp32 accumulation:
//
// NS = 3, "bf16x3".  x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): three 8-bit
//   significands cover fp32's 24 bits, i.e. the triple is (barring underflow) EXACT.  A product a*b is then the sum of
//   nine bf16 x bf16 products; the three with combined order >= 5 — (2,3), (3,2), (3,3) — are below 2^-24 |a b| and are
//   dropped, the other six are formed by v_mfma_f32_16x16x32_bf16: 6/16 of the fp32-MFMA cycles.  bf16 has fp32's
//   exponent: no range restriction.
// NS = 2, "f16x2" (default of the frozen stages).  x = hi + 2^-11 lo with hi = fp16(x) and lo = fp16(2^11 (x - hi)):
//   two 11-bit significands = 22 bits, |x - hi - 2^-11 lo| <= 2^-22 |x|.  Of the four products hi_a hi_b goes to one
//   accumulator, hi_a lo_b + lo_a hi_b (both carry the factor 2^11) to a second one, lo_a lo_b (<= 2^-22 |a b|) is
//   dropped; the result is acc0 + 2^-11 acc1.  THREE v_mfma_f32_16x16x32_f16 — 3/16 of the fp32-MFMA cycles, half of
//   bf16x3 — for an error of <= 3 * 2^-22 per product in the worst case; measured (tests/test_hip_bf16.py): 1e-7 of
//   sum |a b| against float64 for the GEMMs — torch's fp32 GEMM: 2e-7 — and the fp32 kernels' own round-off over a
//   300-step recurrence.  Scaling lo by 2^11 keeps it in fp16's NORMAL range
//   whenever hi is; values below fp16's smallest normal (6.1e-5) get hi = 0 and are carried by lo alone (11 bits,
//   absolute error <= 1.5e-8) so that nothing depends on how the MFMA treats fp16 denormals.  Range: |x| < 65504
//   (fp16's largest finite value; beyond it hi is infinite and the result NaN — SLU_FROZEN_MATH=bf16x3 has no limit).
// NS = 1 is plain bf16 (BASELINE configs[4]: bf16 weights / activations, fp32 accumulation).
//
// Activations travel between the frozen stages as NS planes of 16-bit terms, plane p = element-wise term p,
// rows padded with zeros to a multiple of 32 columns (one MFMA k-chunk).
#pragma once
#include "slu_common.h"

namespace slu {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// The products kept by a split scheme: q-th product = (plane PA(q) of A) x (plane PB(q) of B), added to accumulator
// ACC(q).  Small terms first: the accumulation order adds the corrections before the dominant product.
template <int NS> struct Split;
template <> struct Split<1> {
  static constexpr int NPAIR = 1, NACC = 1;
  static __device__ __host__ constexpr int PA(int) { return 0; }
  static __device__ __host__ constexpr int PB(int) { return 0; }
  static __device__ __host__ constexpr int ACC(int) { return 0; }
};
template <> struct Split<3> {
  static constexpr int NPAIR = 6, NACC = 1;      // (1,1) (2,0) (0,2) (1,0) (0,1) (0,0)
  static __device__ __host__ constexpr int PA(int q) { return (0x001021 >> (4 * q)) & 15; }
  static __device__ __host__ constexpr int PB(int q) { return (0x010201 >> (4 * q)) & 15; }
  static __device__ __host__ constexpr int ACC(int) { return 0; }
};
template <> struct Split<2> {
  static constexpr int NPAIR = 3, NACC = 2;      // (1,0) (0,1) -> accumulator 1 (scaled 2^11); (0,0) -> accumulator 0
  static __device__ __host__ constexpr int PA(int q) { return q == 0 ? 1 : 0; }
  static __device__ __host__ constexpr int PB(int q) { return q == 1 ? 1 : 0; }
  static __device__ __host__ constexpr int ACC(int q) { return q == 2 ? 0 : 1; }
};
constexpr float F16X2_LO_SCALE = 2048.0f, F16X2_LO_INV = 1.0f / 2048.0f;
constexpr float F16_MIN_NORMAL = 6.103515625e-05f;

// round to nearest even on the gfx950 conversion unit (v_cvt_pk_bf16_f32: one instruction instead of the
// five-instruction integer sequence; the splits are VALU-bound in the recurrence and the staging loops)
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float x) {
  return __builtin_bit_cast(unsigned short, (__bf16)x);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// x -> up to three bf16 terms (h[0] + h[1] + h[2] == x exactly unless a term underflows)
template <int NS>
__device__ __forceinline__ void split_bf16(float x, unsigned short (&h)[NS]) {
  h[0] = f32_to_bf16_rne(x);
  if (NS > 1) {
    const float r1 = x - bf16_to_f32(h[0]);    // exact
    h[1] = f32_to_bf16_rne(r1);
    if (NS > 2) {
      const float r2 = r1 - bf16_to_f32(h[1]); // exact
      h[2] = f32_to_bf16_rne(r2);
    }
  }
}

// x -> (hi, lo) fp16 terms of the f16x2 scheme: x ~= hi + 2^-11 lo (see the header comment)
__device__ __forceinline__ void split_f16x2(float x, unsigned short (&h)[2]) {
  const float hi = __builtin_fabsf(x) >= F16_MIN_NORMAL ? (float)(_Float16)x : 0.0f;
  h[0] = __builtin_bit_cast(unsigned short, (_Float16)hi);                              // exact
  h[1] = __builtin_bit_cast(unsigned short, (_Float16)((x - hi) * F16X2_LO_SCALE));     // x - hi is exact
}

// Two values -> the three bf16 planes' words, each word = term of a | term of b << 16 (what the plane layouts store for two
// consecutive elements): the same roundings and exact residuals as split_bf16<3> on a and b separately, on the PACKED
// conversion (one v_cvt_pk_bf16_f32 per level) — 11 VALU instructions per pair instead of ~20 with per-element conversions
// and 16-bit packing.
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2s{a, b}, bf16x2v));
}
__device__ __forceinline__ void split_bf16x3_pair(float a, float b, unsigned (&w)[3]) {
  w[0] = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(w[0] << 16), rb = b - __uint_as_float(w[0] & 0xffff0000u);          // exact
  w[1] = cvt_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(w[1] << 16), sb = rb - __uint_as_float(w[1] & 0xffff0000u);        // exact
  w[2] = cvt_pk_bf16(sa, sb);
}

// The same pair under the rule v_cvt_pk_f16_f32 follows when MODE.FP_DENORM[7:6] = 0 (fp16 denormal results flushed).
// Measured exhaustively over all 2^32 inputs (tools/probes/f16_flush_probe.cpp): tininess is detected AFTER rounding to 11
// bits with an unbounded exponent — |x| >= 2^-14 - 2^-26 (bit pattern 0x387FF000) rounds up to the smallest normal and is
// kept, anything smaller becomes a signed zero (flushing the denormal of the default-mode conversion would keep 4096 more
// patterns per sign, flushing every |x| < 2^-14 4096 fewer).  Written with a compare + select for kernels that run in the
// default (denormal-keeping) mode.  hi differs from split_f16x2's only on those 4096 patterns and in the sign of a flushed
// zero; lo loses the values below 2^-25 (absolute error <= 3e-8).  Both are valid f16x2 pairs for every consumer; the rule
// exists so that the recurrence's dropout + pool epilogue (slu_gru_bf16.hip, which runs in flush mode and splits with two
// packed conversions) and dropout_pool_fwd4_kernel write IDENTICAL planes.
__device__ __forceinline__ unsigned short f16_cvt_flush(float x) {
  const unsigned short h = __builtin_bit_cast(unsigned short, (_Float16)x);
  return (__float_as_uint(x) & 0x7fffffffu) >= 0x387FF000u ? h : (unsigned short)(h & 0x8000u);
}
__device__ __forceinline__ void split_f16x2_flush(float x, unsigned short (&h)[2]) {
  h[0] = f16_cvt_flush(x);
  const float hf = (float)__builtin_bit_cast(_Float16, h[0]);
  h[1] = f16_cvt_flush((x - hf) * F16X2_LO_SCALE);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

// MODE.FP_DENORM[7:6] (f16 / f64 denormals) = 0: v_cvt_pk_f16_f32 then returns ZERO for every result below fp16's smallest
// normal — the "hi = 0, lo carries the value" rule of the f16x2 split (slu_bf16.h) without a compare + select per element.
// f32 denormal handling (bits [5:4]) is untouched; the kernel has no other f16 / f64 arithmetic.
__device__ __forceinline__ void f16_denorm_flush() { __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0); }
__device__ __forceinline__ void f16_denorm_keep() { __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 3); }   // the default
// f16x2 terms of two values under f16_denorm_flush(): (hi0 | hi1 << 16), (lo0 | lo1 << 16).  hi = one packed conversion;
// lo = fp16(2048 (x - hi)) with 2048 (x - hi) = fma(hi, -2048, 2048 x) exact (both products are exact, the difference of x and
// its 11-bit rounding is representable): the compiler folds the widening of hi into v_fma_mix_f32.
__device__ __forceinline__ void split_f16x2_pair_flush(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f16x2v h = __builtin_convertvector(f32x2{x0, x1}, f16x2v);
  const float l0 = __builtin_fmaf((float)h[0], -F16X2_LO_SCALE, x0 * F16X2_LO_SCALE);
  const float l1 = __builtin_fmaf((float)h[1], -F16X2_LO_SCALE, x1 * F16X2_LO_SCALE);
  const f16x2v l = __builtin_convertvector(f32x2{l0, l1}, f16x2v);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

// the NS terms of scheme NS
template <int NS>
__device__ __forceinline__ void split_terms(float x, unsigned short (&h)[NS]) {
  if constexpr (NS == 2) split_f16x2(x, h);
  else split_bf16<NS>(x, h);
}

__device__ __forceinline__ f32x4 mfma_bf16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_f16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// one 16 x 16 x 32 product of scheme NS's element type
template <int NS>
__device__ __forceinline__ f32x4 mfma_split(const uint4& a, const uint4& b, f32x4 c) {
  if constexpr (NS == 2) return mfma_f16(a, b, c);
  else return mfma_bf16(a, b, c);
}
// the result of scheme NS from its accumulators (a1 is ignored when the scheme has one)
template <int NS>
__device__ __forceinline__ f32x4 split_result(const f32x4& a0, const f32x4& a1) {
  if constexpr (NS == 2) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(a1[e], F16X2_LO_INV, a0[e]);
    return r;
  } else {
    return a0;
  }
}

// LDS image of a (rows x 32) bf16 operand tile: 64 bytes (four 16-byte slots) per row, dense; slot kg of
// row r is stored at slot kg ^ SWZ[(r >> 2) & 3].  With the ds_read_b128 lane groups of gfx950
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) every group of 16 fragment reads (lane -> row base + lane%16,
// slot lane/16) then touches 16 distinct bank quads: conflict-free without padding.
__device__ __forceinline__ int swz_slot(int row, int kg) {
  const int f = (0x1230 >> (((row >> 2) & 3) * 4)) & 3;    // {0, 3, 2, 1}[(row >> 2) & 3]
  return kg ^ f;
}

}  // namespace slu
