// Exact-fp32 MFMA GEMM with generic element strides (GRU input projections x @ W_ih^T + b_ih of
// torch.nn.GRU — models.py:232,262,686 — and the three gradient GEMMs of their backward pass).
//
//   C(m,n) = [C(m,n)] + sum_k A(m,k) B(k,n) + [bias_n(n)]
//
// 128 x 128 x 32 workgroup tile, 256 threads = 2 x 2 waves of 64 x 64 (4 x 4 tiles of
// v_mfma_f32_16x16x4_f32, 128 MFMAs per wave per k-tile = 4096 cycles, enough to cover the global
// load latency of the next tile even with one workgroup per CU).  Operands are staged k-major in
// LDS ([k][m], row stride 144 floats, so the four k-groups of a wave hit disjoint banks) through
// registers with a one-tile prefetch (next tile's global loads are in flight during the MFMAs).  The global->register mapping follows whichever
// operand dimension is contiguous, so A may be row- or column-major (likewise B) without a
// transposed copy.  Small-output / long-K problems (weight gradients) are split along K into a
// workspace and reduced in a fixed order (deterministic).
#include "slu_common.h"

namespace slu {

constexpr int GM_BM = 128, GM_BN = 128, GM_BK = 32, GM_LD = 144, GM_THREADS = 256;
constexpr int GM_PASSES = GM_BK / 8;        // float4 loads per thread per operand tile
constexpr int GM_REGS = 4 * GM_PASSES;

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

// Loads the GM_REGS elements thread `tid` owns of a (128 x GM_BK) operand tile into r[].
//   X(row, k) = X[row*rs + k*cs], rows [row0, row0+128) limited by nrows, k in [k0, k0+BK) limited by kend.
//   KFAST: k is the contiguous dimension -> thread owns k-quad tid % (BK/4) of rows tid/(BK/4) + 1024/BK*h.
//   else : row is contiguous          -> thread owns row-quad tid % 32 of k = tid/32 + 8*h.
template <bool KFAST>
__device__ __forceinline__ void load_tile(const float* __restrict__ X, long long rs, long long cs,
                                          int row0, int nrows, int k0, int kend, int tid, float (&r)[GM_REGS]) {
  constexpr int TPR = GM_BK / 4;              // threads per row (k-fast)
#pragma unroll
  for (int h = 0; h < GM_PASSES; ++h) {
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
      const int k = k0 + (tid >> 5) + 8 * h;
      const int row = row0 + 4 * (tid & 31);
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

// Stores r[] into the k-major LDS tile s[k][row] (row stride GM_LD).
template <bool KFAST>
__device__ __forceinline__ void store_tile(float* __restrict__ s, int tid, const float (&r)[GM_REGS]) {
  constexpr int TPR = GM_BK / 4;
#pragma unroll
  for (int h = 0; h < GM_PASSES; ++h) {
    if (KFAST) {
      const int row = tid / TPR + (GM_THREADS / TPR) * h;
      const int k = 4 * (tid % TPR);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[(k + j) * GM_LD + row] = r[4 * h + j];
    } else {
      const int k = (tid >> 5) + 8 * h;
      const int row = 4 * (tid & 31);
      *reinterpret_cast<float4*>(&s[k * GM_LD + row]) = make_float4(r[4 * h], r[4 * h + 1], r[4 * h + 2], r[4 * h + 3]);
    }
  }
}

template <bool A_KFAST, bool B_KFAST>
__global__ void __launch_bounds__(GM_THREADS)
gemm_f32_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) float sA[GM_BK * GM_LD];
  __shared__ __attribute__((aligned(16))) float sB[GM_BK * GM_LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * GM_BM, n0 = blockIdx.x * GM_BN;
  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int ntiles = (kend - kbeg + GM_BK - 1) / GM_BK;
  const int i = lane & 15, kg = lane >> 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  float ra[GM_REGS], rb[GM_REGS];
  // B(k,n) = B[k*b_rs + n*b_cs]: as an (n-rows x k) operand its row stride is b_cs, k stride b_rs.
  if (ntiles > 0) {
    load_tile<A_KFAST>(p.A, p.a_rs, p.a_cs, m0, p.M, kbeg, kend, tid, ra);
    load_tile<B_KFAST>(p.B, p.b_cs, p.b_rs, n0, p.N, kbeg, kend, tid, rb);
  }
  const float* __restrict__ a_s = sA + wm * 64 + i;
  const float* __restrict__ b_s = sB + wn * 64 + i;
  for (int t = 0; t < ntiles; ++t) {
    store_tile<A_KFAST>(sA, tid, ra);
    store_tile<B_KFAST>(sB, tid, rb);
    __syncthreads();
    if (t + 1 < ntiles) {                       // next tile's loads fly during this tile's MFMAs
      const int k0 = kbeg + (t + 1) * GM_BK;
      load_tile<A_KFAST>(p.A, p.a_rs, p.a_cs, m0, p.M, k0, kend, tid, ra);
      load_tile<B_KFAST>(p.B, p.b_cs, p.b_rs, n0, p.N, k0, kend, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < GM_BK / 4; ++kk) {
      const int krow = (kk * 4 + kg) * GM_LD;
      float af[4], bf[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) af[a] = a_s[krow + a * 16];
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[b] = b_s[krow + b * 16];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(af[a], bf[b], acc[a][b]);
    }
    __syncthreads();
  }

  // epilogue: lane holds D[row = 4*kg + r][col = i] of each 16x16 tile
  if (p.ws) {
    float* __restrict__ w = p.ws + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int n = n0 + wn * 64 + b * 16 + i;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 64 + a * 16 + 4 * kg + r;
          if (m < p.M) w[(size_t)m * p.N + n] = acc[a][b][r];
        }
      }
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int n = n0 + wn * 64 + b * 16 + i;
        if (n >= p.N) continue;
        const float bias = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 64 + a * 16 + 4 * kg + r;
          if (m >= p.M) continue;
          float* c = p.C + (long long)m * p.c_rs + (long long)n * p.c_cs;
          float v = acc[a][b][r] + bias;
          if (p.accumulate) v += *c;
          *c = v;
        }
      }
  }
}

__global__ void __launch_bounds__(256)
gemm_splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, long long c_rs,
                          long long c_cs, const float* __restrict__ bias, int M, int N, int KS,
                          int accumulate) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx - (long long)m * N);
  float s = 0.0f;
  for (int k = 0; k < KS; ++k) s += ws[(size_t)k * M * N + idx];
  if (bias) s += bias[n];
  float* c = C + (long long)m * c_rs + (long long)n * c_cs;
  if (accumulate) s += *c;
  *c = s;
}

// out[n] = sum_m X[m*rs + n]; one workgroup per 64 columns, rows strided over 4 waves.
__global__ void __launch_bounds__(256)
colsum_kernel(const float* __restrict__ X, long long rs, float* __restrict__ out, int M, int N,
              int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  float s = 0.0f;
  if (n < N)
    for (int m = w; m < M; m += 4) s += X[(long long)m * rs + n];
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && n < N) {
    const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    out[n] = accumulate ? out[n] + t : t;
  }
}

// Two-stage, atomic-free column sum for tall matrices: partial[rsplit][n] then a fixed-order reduce.
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ X, long long rs, float* __restrict__ part, int M,
                      int N, int rows_per_split) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int m0 = blockIdx.y * rows_per_split;
  const int m1 = min(M, m0 + rows_per_split);
  float s = 0.0f;
  if (n < N)
    for (int m = m0 + w; m < m1; m += 4) s += X[(long long)m * rs + n];
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && n < N)
    part[(size_t)blockIdx.y * N + n] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

__global__ void __launch_bounds__(256)
colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int RS) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.0f;
  for (int r = 0; r < RS; ++r) s += part[(size_t)r * N + n];
  out[n] = s;
}

int colsum_splits(int64_t M) {
  int64_t rs = cdiv(M, 64);
  return (int)(rs > 256 ? 256 : (rs < 1 ? 1 : rs));
}

// out[n] = sum_m X[m*rs + n] using `ws` (colsum_splits(M) * N floats).
int colsum_two_stage(const float* X, int64_t rs, float* out, int64_t M, int64_t N, float* ws,
                     hipStream_t st) {
  const int RS = colsum_splits(M);
  const int rows = (int)cdiv(M, RS);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)cdiv(N, 64), (unsigned)RS), dim3(256), 0, st,
                     X, (long long)rs, ws, (int)M, (int)N, rows);
  SLU_CHECK_LAUNCH("colsum_partial_kernel");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, st,
                     (const float*)ws, out, (int)N, RS);
  SLU_CHECK_LAUNCH("colsum_final_kernel");
  return SLU_OK;
}

static void split_plan(int64_t M, int64_t N, int64_t K, int* KS, int* kper) {
  const int64_t tiles = cdiv(M, GM_BM) * cdiv(N, GM_BN);
  int64_t ks = 1;
  if (tiles < 128 && K >= 512) {
    ks = cdiv(256, tiles);
    const int64_t max_ks = K / 128;            // keep >= 128 k per split
    if (ks > max_ks) ks = max_ks;
    if (ks < 1) ks = 1;
  }
  int64_t per = cdiv(cdiv(K, ks), GM_BK) * GM_BK;
  *kper = (int)per;
  *KS = (int)cdiv(K, per);
}

}  // namespace slu

using namespace slu;

extern "C" size_t slu_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  int KS, kper;
  split_plan(M, N, K, &KS, &kper);
  return KS > 1 ? (size_t)KS * M * N * sizeof(float) : 0;
}

extern "C" int slu_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B,
                            int64_t b_rs, int64_t b_cs, float* C, int64_t c_rs, int64_t c_cs,
                            const float* bias_n, int64_t M, int64_t N, int64_t K, int accumulate,
                            void* workspace, size_t workspace_bytes, void* stream) {
  SLU_REQUIRE(A && B && C, "slu_gemm_f32: null pointer");
  SLU_REQUIRE(M > 0 && N > 0 && K > 0, "slu_gemm_f32: non-positive size");
  SLU_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "slu_gemm_f32: size overflow");
  hipStream_t st = (hipStream_t)stream;
  int KS, kper;
  split_plan(M, N, K, &KS, &kper);
  GemmParams p;
  p.A = A; p.a_rs = a_rs; p.a_cs = a_cs;
  p.B = B; p.b_rs = b_rs; p.b_cs = b_cs;
  p.C = C; p.c_rs = c_rs; p.c_cs = c_cs;
  p.bias = bias_n; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.k_per_split = kper; p.accumulate = accumulate;
  p.ws = nullptr;
  if (KS > 1) {
    const size_t need = (size_t)KS * M * N * sizeof(float);
    if (!workspace || workspace_bytes < need)
      SLU_FAIL(SLU_ERR_WORKSPACE, "slu_gemm_f32: workspace too small (%zu < %zu)", workspace_bytes, need);
    p.ws = reinterpret_cast<float*>(workspace);
  }
  dim3 grid((unsigned)cdiv(N, GM_BN), (unsigned)cdiv(M, GM_BM), (unsigned)KS);
  SLU_REQUIRE(grid.y <= 65535, "slu_gemm_f32: M too large for one launch");
  const bool akf = (a_cs == 1) || (a_rs != 1);     // k-fast mapping unless M is the contiguous dim
  const bool bkf = (b_rs == 1) || (b_cs != 1);     // B(k,n): k contiguous when b_rs == 1
  if (akf && bkf) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(GM_THREADS), 0, st, p);
  else if (akf && !bkf) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(GM_THREADS), 0, st, p);
  else if (!akf && bkf) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(GM_THREADS), 0, st, p);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(GM_THREADS), 0, st, p);
  SLU_CHECK_LAUNCH("gemm_f32_kernel");
  if (KS > 1) {
    const long long total = (long long)M * N;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st,
                       (const float*)p.ws, C, (long long)c_rs, (long long)c_cs, bias_n, (int)M, (int)N,
                       KS, accumulate);
    SLU_CHECK_LAUNCH("gemm_splitk_reduce_kernel");
  }
  return SLU_OK;
}

extern "C" int slu_colsum_f32(const float* X, int64_t x_rs, float* out, int64_t M, int64_t N,
                              int accumulate, void* stream) {
  SLU_REQUIRE(X && out && M > 0 && N > 0, "slu_colsum_f32: bad argument");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream,
                     X, (long long)x_rs, out, (int)M, (int)N, accumulate);
  SLU_CHECK_LAUNCH("colsum_kernel");
  return SLU_OK;
}
