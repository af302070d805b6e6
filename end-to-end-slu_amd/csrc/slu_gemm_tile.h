// Operand-tile helpers of the exact-fp32 MFMA GEMM (slu_gemm.hip), shared with the fused projection + recurrence launch
// (slu_gru_proj.hip): global -> register -> LDS staging of (32*WT x 32) operand tiles and the MFMA fragment reads.
#pragma once
#include "slu_common.h"

namespace slu {

constexpr int GM_BK = 32, GM_THREADS = 256;
// WT = 16x16 MFMA tiles per wave per dimension: workgroup tile (32*WT)^2, WT float4 loads per thread
// per operand k-tile.  WT = 2 (64 x 64) is the default, WT = 4 (128 x 128) for very large problems.
constexpr int GM_KPV = GM_BK + 4;           // LDS row stride of a k-contiguous operand: float4 stores/reads
constexpr int GM_KPS = GM_BK + 1;           // LDS row stride of a row-contiguous operand: scalar, conflict-free

struct GemmParams {
  const float* A; long long a_rs, a_cs;
  const float* B; long long b_rs, b_cs;
  float* C; long long c_rs, c_cs;
  const float* bias;
  float* ws;            // split-K partials [KS][M][N] or null
  int M, N, K;
  int k_per_split;      // multiple of GM_BK
  int accumulate;
};

// Loads the 4*WT elements thread `tid` owns of a (32*WT x GM_BK) operand tile into r[].
//   X(row, k) = X[row*rs + k*cs], rows [row0, row0+128) limited by nrows, k in [k0, k0+BK) limited by kend.
//   KFAST: k is the contiguous dimension -> thread owns k-quad tid % (BK/4) of rows tid/(BK/4) + 1024/BK*h.
//   else : row is contiguous          -> thread owns row-quad tid % (8*WT) of k = tid/(8*WT) + 32/WT*h.
template <bool KFAST, int WT>
__device__ __forceinline__ void load_tile(const float* __restrict__ X, long long rs, long long cs,
                                          int row0, int nrows, int k0, int kend, int tid, float (&r)[4 * WT]) {
  constexpr int TPR = GM_BK / 4;              // threads per row (k-fast)
  constexpr int RQ = 8 * WT;                  // row quads per tile (row-fast)
#pragma unroll
  for (int h = 0; h < WT; ++h) {
    if (KFAST) {
      const int row = row0 + tid / TPR + (GM_THREADS / TPR) * h;
      const int k = k0 + 4 * (tid % TPR);
      const float* p = X + (long long)row * rs + (long long)k * cs;
      if (row < nrows && k + 3 < kend && cs == 1 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        r[4 * h + 0] = v.x; r[4 * h + 1] = v.y; r[4 * h + 2] = v.z; r[4 * h + 3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[4 * h + j] = (row < nrows && k + j < kend) ? p[(long long)j * cs] : 0.0f;
      }
    } else {
      const int k = k0 + tid / RQ + (GM_THREADS / RQ) * h;
      const int row = row0 + 4 * (tid % RQ);
      const float* p = X + (long long)row * rs + (long long)k * cs;
      if (k < kend && row + 3 < nrows && rs == 1 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        r[4 * h + 0] = v.x; r[4 * h + 1] = v.y; r[4 * h + 2] = v.z; r[4 * h + 3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[4 * h + j] = (k < kend && row + j < nrows) ? p[(long long)j * rs] : 0.0f;
      }
    }
  }
}

// LDS tiles are ROW-major with k contiguous: s[row][k].  A k-contiguous operand keeps its float4s
// (row stride 36 floats = 144 B: 16-byte aligned, and the 16 rows of a fragment read land on 16
// distinct 16-byte slots); a row-contiguous operand is transposed on the way in with scalar stores
// (row stride 33: conflict-free for both the stores and the scalar fragment reads).
template <bool KFAST, int WT>
__device__ __forceinline__ void store_tile(float* __restrict__ s, int tid, const float (&r)[4 * WT]) {
  constexpr int TPR = GM_BK / 4;
  constexpr int RQ = 8 * WT;
#pragma unroll
  for (int h = 0; h < WT; ++h) {
    if (KFAST) {
      const int row = tid / TPR + (GM_THREADS / TPR) * h;
      const int k = 4 * (tid % TPR);
      *reinterpret_cast<float4*>(&s[row * GM_KPV + k]) = make_float4(r[4 * h], r[4 * h + 1], r[4 * h + 2], r[4 * h + 3]);
    } else {
      const int k = tid / RQ + (GM_THREADS / RQ) * h;
      const int row = 4 * (tid % RQ);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[(row + j) * GM_KPS + k] = r[4 * h + j];
    }
  }
}

// Fragment of one 16-row tile: lane (i = lane & 15, kg = lane >> 4) gets row (base + i), k = kg*8 .. kg*8+7.
// MFMA step kk of the k-tile consumes element kk from every lane group, i.e. k = kg*8 + kk for A and B alike.
template <bool KFAST>
__device__ __forceinline__ void load_frag(const float* __restrict__ s, int row, int kg, float (&f)[8]) {
  if (KFAST) {
    const float4 lo = *reinterpret_cast<const float4*>(&s[row * GM_KPV + kg * 8]);
    const float4 hi = *reinterpret_cast<const float4*>(&s[row * GM_KPV + kg * 8 + 4]);
    f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = s[row * GM_KPS + kg * 8 + k];
  }
}

}  // namespace slu
