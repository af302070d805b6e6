// bf16 MFMA building blocks shared by the split-precision kernels (slu_gemm_bf16.hip, slu_gru_bf16.hip,
// slu_wconv_bf16.hip).
//
// Split precision.  An fp32 value x is written as x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2): three 8-bit significands cover fp32's 24 bits, i.e. the triple is (barring
// underflow) EXACT.  A product a*b is then the sum of nine bf16 x bf16 products; the three with combined
// order >= 5 — (2,3), (3,2), (3,3) — are below 2^-24 |a b| and are dropped, the other six are formed by
// v_mfma_f32_16x16x32_bf16 (exact products, fp32 accumulation).  Six bf16 MFMAs cost 6/16 of one fp32
// MFMA of the same shape, so an "fp32-class" contraction runs at up to 2.67x the fp32 MFMA rate with an
// error of a few 2^-24 per product (same class as the fp32 fmaf chain; tests hold it to the same 1e-4).
// NS = 1 is plain bf16 (BASELINE configs[4]: bf16 weights / activations, fp32 accumulation).
//
// Activations travel between the frozen stages as NS planes of bf16, plane p = element-wise x_{p+1},
// rows padded with zeros to a multiple of 32 columns (one MFMA k-chunk).
#pragma once
#include "slu_common.h"

namespace slu {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// the product pairs (plane of A, plane of B) kept for a given split count
template <int NS> struct SplitPairs;
template <> struct SplitPairs<1> { static constexpr int N = 1; static constexpr int A[1] = {0}; static constexpr int B[1] = {0}; };
template <> struct SplitPairs<3> {
  static constexpr int N = 6;
  // small terms first: the accumulation order adds the corrections before the dominant product
  static constexpr int A[6] = {1, 2, 0, 1, 0, 0};
  static constexpr int B[6] = {1, 0, 2, 0, 1, 0};
};

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

__device__ __forceinline__ f32x4 mfma_bf16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
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
