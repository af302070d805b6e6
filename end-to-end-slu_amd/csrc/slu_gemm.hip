// Exact-fp32 MFMA GEMM with generic element strides (GRU input projections x @ W_ih^T + b_ih of
// torch.nn.GRU — models.py:232,262,686 — and the three gradient GEMMs of their backward pass).
//
//   C(m,n) = [C(m,n)] + sum_k A(m,k) B(k,n) + [bias_n(n)]
//
// 128 x 128 x 32 workgroup tile, 256 threads = 2 x 2 waves of 64 x 64 (4 x 4 tiles of
// v_mfma_f32_16x16x4_f32, 128 MFMAs per wave per k-tile = 4096 cycles, enough to cover the global
// load latency of the next tile even with one workgroup per CU).  Operands are staged through
// registers (one-tile prefetch: the next tile's global loads are in flight during the MFMAs) into
// row-major LDS tiles with k contiguous; the MFMA k-index is remapped so that each lane group owns 8
// CONSECUTIVE k of the tile: a fragment is two ds_read_b128 instead of eight ds_read_b32, and all
// 16 fragment reads of a k-tile are issued before its 128 back-to-back MFMAs.  The global->register mapping follows whichever
// operand dimension is contiguous, so A may be row- or column-major (likewise B) without a
// transposed copy.  Small-output / long-K problems (weight gradients) are split along K into a
// workspace and reduced in a fixed order (deterministic).
#include "slu_common.h"
#include "slu_gemm_tile.h"
#include <cstdlib>

namespace slu {

template <bool A_KFAST, bool B_KFAST, int WT>
__global__ void __launch_bounds__(GM_THREADS)
gemm_f32_kernel(const GemmParams p) {
  constexpr int GM_BM = 32 * WT, GM_BN = 32 * WT, WR = 16 * WT;
  __shared__ __attribute__((aligned(16))) float smem[(GM_BM + GM_BN) * GM_KPV];
  float* const sA = smem;
  float* const sB = smem + GM_BM * GM_KPV;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order.  Workgroup L = (y * gridDim.x + x) is dispatched to XCD L % 8 (MI355X_MICROARCH.md),
  // each XCD has its own L2: the gridDim.x column tiles that re-read one A row-tile must run on ONE XCD, or
  // that tile is fetched through eight L2s (measured with rocprofv3 FETCH_SIZE on the input projections:
  // 7.8x the algorithmic A bytes).  Renumber so that each XCD walks a contiguous range of (row-tile, column-
  // tile) pairs: V = (L % 8) * ceil(total / 8) + L / 8.  Tiles past the end (total not a multiple of 8) are
  // taken by the workgroups whose V falls outside: they fetch the leftover ids instead.
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int total = gridDim.x * gridDim.y;
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    if ((total & 7) == 0) {
      const int V = (L & 7) * (total >> 3) + (L >> 3);
      by = V / gridDim.x;
      bx = V - by * gridDim.x;
    }
  }
  const int m0 = by * GM_BM, n0 = bx * GM_BN;
  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int ntiles = (kend - kbeg + GM_BK - 1) / GM_BK;
  const int i = lane & 15, kg = lane >> 4;

  f32x4 acc[WT][WT];
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < WT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  float ra[4 * WT], rb[4 * WT];
  // B(k,n) = B[k*b_rs + n*b_cs]: as an (n-rows x k) operand its row stride is b_cs, k stride b_rs.
  if (ntiles > 0) {
    load_tile<A_KFAST, WT>(p.A, p.a_rs, p.a_cs, m0, p.M, kbeg, kend, tid, ra);
    load_tile<B_KFAST, WT>(p.B, p.b_cs, p.b_rs, n0, p.N, kbeg, kend, tid, rb);
  }
  for (int t = 0; t < ntiles; ++t) {
    store_tile<A_KFAST, WT>(sA, tid, ra);
    store_tile<B_KFAST, WT>(sB, tid, rb);
    __syncthreads();
    if (t + 1 < ntiles) {                       // next tile's loads fly during this tile's MFMAs
      const int k0 = kbeg + (t + 1) * GM_BK;
      load_tile<A_KFAST, WT>(p.A, p.a_rs, p.a_cs, m0, p.M, k0, kend, tid, ra);
      load_tile<B_KFAST, WT>(p.B, p.b_cs, p.b_rs, n0, p.N, k0, kend, tid, rb);
    }
    float af[WT][8], bf[WT][8];
#pragma unroll
    for (int a = 0; a < WT; ++a) load_frag<A_KFAST>(sA, wm * WR + a * 16 + i, kg, af[a]);
#pragma unroll
    for (int b = 0; b < WT; ++b) load_frag<B_KFAST>(sB, wn * WR + b * 16 + i, kg, bf[b]);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b) acc[a][b] = mfma16(af[a][kk], bf[b][kk], acc[a][b]);
    __syncthreads();
  }

  // epilogue: lane holds D[row = 4*kg + r][col = i] of each 16x16 tile
  float* __restrict__ outp = p.ws ? p.ws + (size_t)blockIdx.z * p.M * p.N : p.C;
  const long long o_rs = p.ws ? p.N : p.c_rs, o_cs = p.ws ? 1 : p.c_cs;
  const bool with_bias = !p.ws && p.bias, with_acc = !p.ws && p.accumulate;
  if (WT == 2 && o_cs == 1 && (o_rs & 3) == 0 && ((reinterpret_cast<uintptr_t>(outp) & 15) == 0)) {
    // Row-major output: stage the 64 x 64 tile in LDS (the operand tiles are dead) and write whole
    // 256-byte rows with float4 stores; a direct store from the MFMA layout is 64-byte pieces.
    constexpr int LDC = GM_BN + 4;     // 4 k-groups x 4 rows land 16 banks apart: conflict-free
    static_assert(GM_BM * LDC <= 2 * GM_BM * GM_KPV || WT != 2, "C tile must fit in the operand tiles");
    float* __restrict__ sC = sA;       // sA and sB are adjacent: 2 * 64 * 36 floats >= 64 * 68
    __syncthreads();
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
      for (int b = 0; b < WT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sC[(wm * WR + a * 16 + 4 * kg + r) * LDC + wn * WR + b * 16 + i] = acc[a][b][r];
    __syncthreads();
    const int col = 4 * (tid & 15);
    const int n = n0 + col;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (with_bias && n + 3 < p.N) {
      bv.x = p.bias[n]; bv.y = p.bias[n + 1]; bv.z = p.bias[n + 2]; bv.w = p.bias[n + 3];
    }
#pragma unroll
    for (int h = 0; h < GM_BM / 16; ++h) {
      const int row = (tid >> 4) + 16 * h;
      const int m = m0 + row;
      if (m >= p.M) continue;
      const float4 v = *reinterpret_cast<const float4*>(&sC[row * LDC + col]);
      float* c = outp + (long long)m * o_rs + n;
      if (n + 3 < p.N) {
        float4 o = make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w);
        if (with_acc) {
          const float4 old = *reinterpret_cast<const float4*>(c);
          o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *reinterpret_cast<float4*>(c) = o;
      } else {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (n + j >= p.N) continue;
          float o = vv[j] + (with_bias ? p.bias[n + j] : 0.0f);
          if (with_acc) o += c[j];
          c[j] = o;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < WT; ++b) {
      const int n = n0 + wn * WR + b * 16 + i;
      if (n >= p.N) continue;
      const float bias = with_bias ? p.bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * WR + a * 16 + 4 * kg + r;
        if (m >= p.M) continue;
        float* c = outp + (long long)m * o_rs + (long long)n * o_cs;
        float v = acc[a][b][r] + bias;
        if (with_acc) v += *c;
        *c = v;
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
  if (n < N) {
    // eight independent partial sums per thread: eight loads in flight instead of one dependent load per iteration
    // (a 1536-row sum took 70 us as a 384-step chain of L2 round trips); fixed order, deterministic
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int m = w;
    for (; m + 28 < M; m += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += X[(long long)(m + 4 * u) * rs + n];
    }
    for (; m < M; m += 4) a[0] += X[(long long)m * rs + n];
    s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
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

// 64 columns x 4 row groups per workgroup: the (up to 256) partial rows are summed by four waves in
// parallel (fixed order: deterministic), not by one serial loop per column.
__global__ void __launch_bounds__(256)
colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int RS) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  float s0 = 0.0f, s1 = 0.0f;
  if (n < N) {
    int r = w;
#pragma unroll 4
    for (; r + 4 < RS; r += 8) { s0 += part[(size_t)r * N + n]; s1 += part[(size_t)(r + 4) * N + n]; }
    if (r < RS) s0 += part[(size_t)r * N + n];
  }
  red[w][lane] = s0 + s1;
  __syncthreads();
  if (w == 0 && n < N) out[n] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// ---- batched small "A^T B" GEMMs: the weight gradients of ONE small GRU layer in one launch -------------------
//   C_q (M_q x N_q) = A_q^T B_q,  A_q (K_q x M_q, row stride lda), B_q (K_q x N_q, row stride ldb), q < count <= 4
// (dW_ih = d_gx^T x and dW_hh = d_gh^T h_prev of each direction).  With K = T*B of a few thousand rows (the intent
// layer of the look-ahead pipeline: 1216) the generic kernel needs split-K plus a reduce launch per matrix to fill
// the device: three GEMMs + three reduces, 66-75 us on the training stream's 64 CUs for 0.7 GFLOP.  Here a workgroup
// owns a 64 x 32 output tile of one of the matrices, its four waves split the k range, operands go straight from
// global memory into the exact-fp32 MFMA (both are k-slow: for a fixed k sixteen lanes read 256 / 128 contiguous bytes)
// and the four partial tiles are summed through LDS in a fixed order (deterministic, no workspace, no second launch).
struct TnProblem {
  const float* A; const float* B; float* C;
  long long lda, ldb, ldc;
  int M, N, K;
  int tile_end;       // exclusive prefix of tile counts
  int tiles_n;
};
struct TnArgs {
  TnProblem p[4];
  int count;
  // optional extra job of the same launch: rs_dst[c] = sum_r rs_src[r * rs_cols + c] (rows added in order), done by
  // the workgroups past the last tile — the layer's bias gradients (per-tile partial sums of the BPTT kernel)
  const float* rs_src; float* rs_dst; int rs_rows, rs_cols, tiles;
  int rs_blocks;      // the row-sum job takes the FIRST rs_blocks workgroups (it runs beside the tiles, not after them)
};

// the row-sum job of a batched launch: 16 rows in flight at a time (clamped row index instead of a branch around the
// loads: one memory round trip per 16 rows, not one per row), added in row order
__device__ __forceinline__ void tn_rowsum(const TnArgs& a, int c) {
  if (c >= a.rs_cols) return;
  float s = 0.0f;
  for (int r0 = 0; r0 < a.rs_rows; r0 += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = a.rs_src[(long long)min(r0 + u, a.rs_rows - 1) * a.rs_cols + c];
#pragma unroll
    for (int u = 0; u < 16; ++u) s = (r0 + u < a.rs_rows) ? (r0 + u == 0 ? v[u] : s + v[u]) : s;
  }
  a.rs_dst[c] = s;
}

__global__ void __launch_bounds__(256)
gemm_tn_small_kernel(const TnArgs a) {
  __shared__ float red[4][8][256];                     // [wave][tile][lane*4 + r]
  if ((int)blockIdx.x < a.rs_blocks) {                 // the row-sum job
    tn_rowsum(a, (int)blockIdx.x * 256 + threadIdx.x);
    return;
  }
  const int bid = (int)blockIdx.x - a.rs_blocks;
  int q = 0;
  while (q + 1 < a.count && bid >= a.p[q].tile_end) ++q;
  const TnProblem& P = a.p[q];
  const int tile = bid - (q ? a.p[q - 1].tile_end : 0);
  const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
  const int m0 = tm * 64, n0 = tn * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kg = lane >> 4;
  // this wave's k range: quarters of the FULL 4-row MFMA steps; a partial last step (K % 4 != 0) goes to wave 3.
  // Rows / columns of the tile past M / N read column 0 instead: the accumulator rows / columns they feed are never
  // stored, so no select sits between a load and its MFMA (a select there makes the compiler wait for each load
  // where it is issued: measured 80 us instead of 50 for the intent layer's three matrices).
  const int steps = P.K >> 2;
  const int per = ((((steps + 3) >> 2) + 7) >> 3) << 3;   // a multiple of the batch size U = 8: no per-step tail loop
  const int s0 = min(steps, w * per), s1 = min(steps, s0 + per);
  // ONE 16-byte (A) and ONE 8-byte (B) load per k row feed all MFMA tiles of a 64 x 32 output tile: lane i holds
  // columns 4i..4i+3 of A and 2i, 2i+1 of B, i.e. MFMA row-tile x covers the rows m0 + 4*(0..15) + x and column-tile
  // y the columns n0 + 2*(0..15) + y (a permutation, undone at the store).  A workgroup moves (64 + 32) K floats
  // for 64 x 32 x K MACs: the kernel is bound by what one CU can pull through its L1 (~60 GB/s), 32 x 32 tiles
  // moved 1.5x as much per MAC.  (M % 4 == 0, N % 2 == 0: checked by the launcher.)
  const float* __restrict__ pa = P.A + (m0 + 4 * i + 3 < P.M ? m0 + 4 * i : 0);
  const float* __restrict__ pb = P.B + (n0 + 2 * i + 1 < P.N ? n0 + 2 * i : 0);
  f32x4 acc[4][2];
#pragma unroll
  for (int x = 0; x < 4; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // batches of U = 8 MFMA steps (16 loads per lane), software-pipelined over two register sets: the next batch's
  // loads are issued before the current batch's MFMAs (the batch index is clamped instead of branching around the
  // loads, which would send the register arrays through scratch memory)
  constexpr int U = 8;
  const int nb = (s1 > s0) ? (s1 - s0) / U : 0;
  float4 avA[U], avB[U];
  float2 bvA[U], bvB[U];
#define TN_LOAD(av, bv, batch)                                                        \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                      \
    const long long k = 4 * (s0 + (batch) * U + u) + kg;                               \
    av[u] = *reinterpret_cast<const float4*>(pa + k * P.lda);                          \
    bv[u] = *reinterpret_cast<const float2*>(pb + k * P.ldb);                          \
  }                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#define TN_STEP(a_, b_)                                                                \
  acc[0][0] = mfma16(a_.x, b_.x, acc[0][0]); acc[0][1] = mfma16(a_.x, b_.y, acc[0][1]); \
  acc[1][0] = mfma16(a_.y, b_.x, acc[1][0]); acc[1][1] = mfma16(a_.y, b_.y, acc[1][1]); \
  acc[2][0] = mfma16(a_.z, b_.x, acc[2][0]); acc[2][1] = mfma16(a_.z, b_.y, acc[2][1]); \
  acc[3][0] = mfma16(a_.w, b_.x, acc[3][0]); acc[3][1] = mfma16(a_.w, b_.y, acc[3][1]);
#define TN_MFMA(av, bv)                                                                \
  _Pragma("unroll") for (int u = 0; u < U; ++u) { TN_STEP(av[u], bv[u]) }              \
  __builtin_amdgcn_sched_barrier(0);
  if (nb > 0) {
    TN_LOAD(avA, bvA, 0)
    for (int bt = 0; bt < nb; bt += 2) {
      TN_LOAD(avB, bvB, min(bt + 1, nb - 1))
      TN_MFMA(avA, bvA)
      TN_LOAD(avA, bvA, min(bt + 2, nb - 1))
      if (bt + 1 < nb) { TN_MFMA(avB, bvB) }
    }
  }
  int sb = s0 + nb * U;
  for (; sb < s1; ++sb) {                              // fewer than U full steps left
    const long long k = 4 * sb + kg;
    const float4 av = *reinterpret_cast<const float4*>(pa + k * P.lda);
    const float2 bv = *reinterpret_cast<const float2*>(pb + k * P.ldb);
    TN_STEP(av, bv)
  }
  if (w == 3 && (P.K & 3)) {                           // partial last step: zero the rows past K
    const long long k = 4 * steps + kg;
    const bool kok = k < P.K;
    const long long kc = kok ? k : 0;
    float4 av = *reinterpret_cast<const float4*>(pa + kc * P.lda);
    const float2 bv = *reinterpret_cast<const float2*>(pb + kc * P.ldb);
    av.x = kok ? av.x : 0.0f; av.y = kok ? av.y : 0.0f; av.z = kok ? av.z : 0.0f; av.w = kok ? av.w : 0.0f;
    TN_STEP(av, bv)
  }
#undef TN_MFMA
#undef TN_STEP
#undef TN_LOAD
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][2 * x + y][lane * 4 + r] = acc[x][y][r];
  __syncthreads();
  // wave w finishes the two tiles of row-tile x = w: element (lane, r) is row 4*kg + r, column i of a 16 x 16 tile
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = lane * 4 + r, t = 2 * w + y;
      const float v = ((red[0][t][e] + red[1][t][e]) + red[2][t][e]) + red[3][t][e];
      const int m = m0 + 4 * (4 * kg + r) + w, n = n0 + 2 * i + y;
      if (m < P.M && n < P.N) P.C[(long long)m * P.ldc + n] = v;
    }
}

// 64 x 64 output tiles (four columns of B per lane: one 16-byte load): a third fewer operand bytes per MAC than the
// 64 x 32 kernel, twice the MFMAs per wave.  N % 4 == 0, ldb % 4 == 0, B 16-byte aligned.
__global__ void __launch_bounds__(256)
gemm_tn_small_wide_kernel(const TnArgs a) {
  __shared__ float red[4][16][256];                    // [wave][tile][lane*4 + r]
  if ((int)blockIdx.x < a.rs_blocks) {                 // the row-sum job
    tn_rowsum(a, (int)blockIdx.x * 256 + threadIdx.x);
    return;
  }
  const int bid = (int)blockIdx.x - a.rs_blocks;
  int q = 0;
  while (q + 1 < a.count && bid >= a.p[q].tile_end) ++q;
  const TnProblem& P = a.p[q];
  const int tile = bid - (q ? a.p[q - 1].tile_end : 0);
  const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
  const int m0 = tm * 64, n0 = tn * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kg = lane >> 4;
  // this wave's k range: quarters of the FULL 4-row MFMA steps; a partial last step (K % 4 != 0) goes to wave 3.
  // Rows / columns of the tile past M / N read column 0 instead: the accumulator rows / columns they feed are never
  // stored, so no select sits between a load and its MFMA (a select there makes the compiler wait for each load
  // where it is issued: measured 80 us instead of 50 for the intent layer's three matrices).
  const int steps = P.K >> 2;
  const int per = ((((steps + 3) >> 2) + 7) >> 3) << 3;   // a multiple of the batch size U = 8: no per-step tail loop
  const int s0 = min(steps, w * per), s1 = min(steps, s0 + per);
  // ONE 16-byte (A) and ONE 8-byte (B) load per k row feed all MFMA tiles of a 64 x 32 output tile: lane i holds
  // columns 4i..4i+3 of A and 2i, 2i+1 of B, i.e. MFMA row-tile x covers the rows m0 + 4*(0..15) + x and column-tile
  // y the columns n0 + 2*(0..15) + y (a permutation, undone at the store).  A workgroup moves (64 + 32) K floats
  // for 64 x 32 x K MACs: the kernel is bound by what one CU can pull through its L1 (~60 GB/s), 32 x 32 tiles
  // moved 1.5x as much per MAC.  (M % 4 == 0, N % 2 == 0: checked by the launcher.)
  const float* __restrict__ pa = P.A + (m0 + 4 * i + 3 < P.M ? m0 + 4 * i : 0);
  const float* __restrict__ pb = P.B + (n0 + 4 * i + 3 < P.N ? n0 + 4 * i : 0);
  f32x4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  // batches of U = 8 MFMA steps (16 loads per lane), software-pipelined over two register sets: the next batch's
  // loads are issued before the current batch's MFMAs (the batch index is clamped instead of branching around the
  // loads, which would send the register arrays through scratch memory)
  constexpr int U = 8;
  const int nb = (s1 > s0) ? (s1 - s0) / U : 0;
  float4 avA[U], avB[U];
  float4 bvA[U], bvB[U];
#define TN_LOAD(av, bv, batch)                                                        \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                      \
    const long long k = 4 * (s0 + (batch) * U + u) + kg;                               \
    av[u] = *reinterpret_cast<const float4*>(pa + k * P.lda);                          \
    bv[u] = *reinterpret_cast<const float4*>(pb + k * P.ldb);                          \
  }                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#define TN_ROW(x_, av_, b_)                                                            \
  acc[x_][0] = mfma16(av_, b_.x, acc[x_][0]); acc[x_][1] = mfma16(av_, b_.y, acc[x_][1]); \
  acc[x_][2] = mfma16(av_, b_.z, acc[x_][2]); acc[x_][3] = mfma16(av_, b_.w, acc[x_][3]);
#define TN_STEP(a_, b_) TN_ROW(0, a_.x, b_) TN_ROW(1, a_.y, b_) TN_ROW(2, a_.z, b_) TN_ROW(3, a_.w, b_)
#define TN_MFMA(av, bv)                                                                \
  _Pragma("unroll") for (int u = 0; u < U; ++u) { TN_STEP(av[u], bv[u]) }              \
  __builtin_amdgcn_sched_barrier(0);
  if (nb > 0) {
    TN_LOAD(avA, bvA, 0)
    for (int bt = 0; bt < nb; bt += 2) {
      TN_LOAD(avB, bvB, min(bt + 1, nb - 1))
      TN_MFMA(avA, bvA)
      TN_LOAD(avA, bvA, min(bt + 2, nb - 1))
      if (bt + 1 < nb) { TN_MFMA(avB, bvB) }
    }
  }
  int sb = s0 + nb * U;
  for (; sb < s1; ++sb) {                              // fewer than U full steps left
    const long long k = 4 * sb + kg;
    const float4 av = *reinterpret_cast<const float4*>(pa + k * P.lda);
    const float4 bv = *reinterpret_cast<const float4*>(pb + k * P.ldb);
    TN_STEP(av, bv)
  }
  if (w == 3 && (P.K & 3)) {                           // partial last step: zero the rows past K
    const long long k = 4 * steps + kg;
    const bool kok = k < P.K;
    const long long kc = kok ? k : 0;
    float4 av = *reinterpret_cast<const float4*>(pa + kc * P.lda);
    const float4 bv = *reinterpret_cast<const float4*>(pb + kc * P.ldb);
    av.x = kok ? av.x : 0.0f; av.y = kok ? av.y : 0.0f; av.z = kok ? av.z : 0.0f; av.w = kok ? av.w : 0.0f;
    TN_STEP(av, bv)
  }
#undef TN_MFMA
#undef TN_STEP
#undef TN_ROW
#undef TN_LOAD
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][4 * x + y][lane * 4 + r] = acc[x][y][r];
  __syncthreads();
  // wave w finishes the four tiles of row-tile x = w: element (lane, r) is row 4*kg + r, column i of a 16 x 16 tile
#pragma unroll
  for (int y = 0; y < 4; ++y)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = lane * 4 + r, t = 4 * w + y;
      const float v = ((red[0][t][e] + red[1][t][e]) + red[2][t][e]) + red[3][t][e];
      const int m = m0 + 4 * (4 * kg + r) + w, n = n0 + 4 * i + y;
      if (m < P.M && n < P.N) P.C[(long long)m * P.ldc + n] = v;
    }
}


// The 64 x 64 kernel with the k range ALSO split over workgroups, for the weight gradients of the long layers (K = T * B
// = 2 400 ... 19 200 rows: a layer's 72 tiles alone leave most of the chip idle, and the generic k-slow GEMM + reduce
// launches it replaced ran at 14 % of the fp32 MFMA peak).  Workgroup (tile, ks) multiplies k steps [ks, ks + 1) * sps of
// its tile exactly as above (four waves, fixed-order LDS sum) and writes the 64 x 64 partial to the workspace; the LAST
// workgroup of a tile to arrive (ticket) adds the ksplit partials in the order ks = 0, 1, ... — the same sum whichever
// workgroup it is — and stores C.  One launch per layer, deterministic, no reduce launch.
// workspace: ksplit * tiles partial tiles of 4096 floats; tickets: one zero word per tile, left zero.
__global__ void __launch_bounds__(256)
gemm_tn_wide_splitk_kernel(const TnArgs a, const int ksplit, float* __restrict__ ws, unsigned* __restrict__ tickets) {
  __shared__ float red[4][16][256];                    // [wave][tile][lane*4 + r]
  __shared__ int s_last;
  if ((int)blockIdx.x < a.rs_blocks) {                 // the row-sum job
    tn_rowsum(a, (int)blockIdx.x * 256 + threadIdx.x);
    return;
  }
  const int bid = (int)blockIdx.x - a.rs_blocks;
  const int gtile = bid / ksplit, ks = bid - gtile * ksplit;
  int q = 0;
  while (q + 1 < a.count && gtile >= a.p[q].tile_end) ++q;
  const TnProblem& P = a.p[q];
  const int tile = gtile - (q ? a.p[q - 1].tile_end : 0);
  const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
  const int m0 = tm * 64, n0 = tn * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kg = lane >> 4;
  // this workgroup's k steps (4 rows each): a multiple of 32 per split, so that every wave runs whole batches of U = 8
  const int steps = P.K >> 2;
  const int sps = ((((steps + ksplit - 1) / ksplit) + 31) >> 5) << 5;
  const int g0 = min(steps, ks * sps), g1 = min(steps, g0 + sps);
  const int per = sps >> 2;                            // a multiple of 8
  const int s0 = min(g1, g0 + w * per), s1 = min(g1, s0 + per);
  const float* __restrict__ pa = P.A + (m0 + 4 * i + 3 < P.M ? m0 + 4 * i : 0);
  const float* __restrict__ pb = P.B + (n0 + 4 * i + 3 < P.N ? n0 + 4 * i : 0);
  f32x4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;
  const int nb = (s1 > s0) ? (s1 - s0) / U : 0;
  float4 avA[U], avB[U];
  float4 bvA[U], bvB[U];
#define TN_LOAD(av, bv, batch)                                                        \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                      \
    const long long k = 4 * (s0 + (batch) * U + u) + kg;                               \
    av[u] = *reinterpret_cast<const float4*>(pa + k * P.lda);                          \
    bv[u] = *reinterpret_cast<const float4*>(pb + k * P.ldb);                          \
  }                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#define TN_ROW(x_, av_, b_)                                                            \
  acc[x_][0] = mfma16(av_, b_.x, acc[x_][0]); acc[x_][1] = mfma16(av_, b_.y, acc[x_][1]); \
  acc[x_][2] = mfma16(av_, b_.z, acc[x_][2]); acc[x_][3] = mfma16(av_, b_.w, acc[x_][3]);
#define TN_STEP(a_, b_) TN_ROW(0, a_.x, b_) TN_ROW(1, a_.y, b_) TN_ROW(2, a_.z, b_) TN_ROW(3, a_.w, b_)
#define TN_MFMA(av, bv)                                                                \
  _Pragma("unroll") for (int u = 0; u < U; ++u) { TN_STEP(av[u], bv[u]) }              \
  __builtin_amdgcn_sched_barrier(0);
  if (nb > 0) {
    TN_LOAD(avA, bvA, 0)
    for (int bt = 0; bt < nb; bt += 2) {
      TN_LOAD(avB, bvB, min(bt + 1, nb - 1))
      TN_MFMA(avA, bvA)
      TN_LOAD(avA, bvA, min(bt + 2, nb - 1))
      if (bt + 1 < nb) { TN_MFMA(avB, bvB) }
    }
  }
  int sb = s0 + nb * U;
  for (; sb < s1; ++sb) {                              // fewer than U full steps left
    const long long k = 4 * sb + kg;
    const float4 av = *reinterpret_cast<const float4*>(pa + k * P.lda);
    const float4 bv = *reinterpret_cast<const float4*>(pb + k * P.ldb);
    TN_STEP(av, bv)
  }
  if (ks == ksplit - 1 && w == 3 && (P.K & 3)) {       // partial last step of the whole k range: zero the rows past K
    const long long k = 4 * steps + kg;
    const bool kok = k < P.K;
    const long long kc = kok ? k : 0;
    float4 av = *reinterpret_cast<const float4*>(pa + kc * P.lda);
    const float4 bv = *reinterpret_cast<const float4*>(pb + kc * P.ldb);
    av.x = kok ? av.x : 0.0f; av.y = kok ? av.y : 0.0f; av.z = kok ? av.z : 0.0f; av.w = kok ? av.w : 0.0f;
    TN_STEP(av, bv)
  }
#undef TN_MFMA
#undef TN_STEP
#undef TN_ROW
#undef TN_LOAD
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][4 * x + y][lane * 4 + r] = acc[x][y][r];
  __syncthreads();
  // wave w finishes the four 16 x 16 tiles of row-tile x = w: element (lane, r) is row 4*kg + r, column i
  float4 part[4];
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    const int t = 4 * w + y;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = lane * 4 + r;
      v[r] = ((red[0][t][e] + red[1][t][e]) + red[2][t][e]) + red[3][t][e];
    }
    part[y] = make_float4(v[0], v[1], v[2], v[3]);
  }
  if (ksplit > 1) {
    // partial tile -> workspace, [16 sub-tiles][256] in the layout of `red`: 16 bytes per lane, 1 KB per wave and store
    float* mine = ws + ((size_t)gtile * ksplit + ks) * 4096;
#pragma unroll
    for (int y = 0; y < 4; ++y) *reinterpret_cast<float4*>(mine + (4 * w + y) * 256 + lane * 4) = part[y];
    __threadfence();                                   // this workgroup's partial visible device-wide before its ticket
    __syncthreads();
    if (tid == 0) s_last = (atomicAdd(tickets + gtile, 1u) == (unsigned)(ksplit - 1)) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                   // acquire: the other workgroups' partials
    const float* all = ws + (size_t)gtile * ksplit * 4096;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      float4 s4 = *reinterpret_cast<const float4*>(all + (4 * w + y) * 256 + lane * 4);
      for (int k2 = 1; k2 < ksplit; ++k2) {
        const float4 o = *reinterpret_cast<const float4*>(all + (size_t)k2 * 4096 + (4 * w + y) * 256 + lane * 4);
        s4.x += o.x; s4.y += o.y; s4.z += o.z; s4.w += o.w;
      }
      part[y] = s4;
    }
    if (tid == 0) tickets[gtile] = 0u;                 // ready for the next launch (stream order)
  }
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    const float v[4] = {part[y].x, part[y].y, part[y].z, part[y].w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * (4 * kg + r) + w, n = n0 + 4 * i + y;
      if (m < P.M && n < P.N) P.C[(long long)m * P.ldc + n] = v[r];
    }
  }
}


// The same kernel with MT floats of A per lane and k row: 3 -> 48 x 32 output tiles, 12-byte loads, for matrices whose
// M is not a multiple of 4 (or whose A is not 16-byte aligned).  A tile's k range is summed exactly as above.  Kept
// apart from the 64-row kernel: writing that one as the MT = 4 instance of this template cost it 14 us (53 -> 67 us
// for the intent layer; same loads and MFMAs, a different schedule).
template <int MT> struct TnVec;
template <> struct TnVec<4> { typedef float4 type; };
template <> struct TnVec<3> { typedef float3 type; };
template <int MT> __device__ __forceinline__ float tn_elem(const typename TnVec<MT>::type& v, int x);
template <> __device__ __forceinline__ float tn_elem<4>(const float4& v, int x) { return x == 0 ? v.x : x == 1 ? v.y : x == 2 ? v.z : v.w; }
template <> __device__ __forceinline__ float tn_elem<3>(const float3& v, int x) { return x == 0 ? v.x : x == 1 ? v.y : v.z; }

template <int MT>
__global__ void __launch_bounds__(256)
gemm_tn_small_mt_kernel(const TnArgs a) {
  typedef typename TnVec<MT>::type avec;
  __shared__ float red[4][2 * MT][256];                // [wave][tile][lane*4 + r]
  if ((int)blockIdx.x < a.rs_blocks) {                 // the row-sum job
    tn_rowsum(a, (int)blockIdx.x * 256 + threadIdx.x);
    return;
  }
  const int bid = (int)blockIdx.x - a.rs_blocks;
  int q = 0;
  while (q + 1 < a.count && bid >= a.p[q].tile_end) ++q;
  const TnProblem& P = a.p[q];
  const int tile = bid - (q ? a.p[q - 1].tile_end : 0);
  const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
  const int m0 = tm * (16 * MT), n0 = tn * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kg = lane >> 4;
  // this wave's k range: quarters of the FULL 4-row MFMA steps; a partial last step (K % 4 != 0) goes to wave 3.
  // Rows / columns of the tile past M / N read column 0 instead: the accumulator rows / columns they feed are never
  // stored, so no select sits between a load and its MFMA (a select there makes the compiler wait for each load
  // where it is issued: measured 80 us instead of 50 for the intent layer's three matrices).
  const int steps = P.K >> 2;
  const int per = ((((steps + 3) >> 2) + 7) >> 3) << 3;   // a multiple of the batch size U = 8: no per-step tail loop
  const int s0 = min(steps, w * per), s1 = min(steps, s0 + per);
  // ONE 4*MT-byte (A) and ONE 8-byte (B) load per k row feed all MFMA tiles of the (16 MT) x 32 output tile: lane i
  // holds columns MT*i .. MT*i + MT-1 of A and 2i, 2i+1 of B, i.e. MFMA row-tile x covers the rows m0 + MT*(0..15) + x
  // and column-tile y the columns n0 + 2*(0..15) + y (a permutation, undone at the store).  (M % MT == 0, N % 2 == 0:
  // checked by the launcher.)
  const float* __restrict__ pa = P.A + (m0 + MT * i + MT - 1 < P.M ? m0 + MT * i : 0);
  const float* __restrict__ pb = P.B + (n0 + 2 * i + 1 < P.N ? n0 + 2 * i : 0);
  f32x4 acc[MT][2];
#pragma unroll
  for (int x = 0; x < MT; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // batches of U = 8 MFMA steps (16 loads per lane), software-pipelined over two register sets: the next batch's
  // loads are issued before the current batch's MFMAs (the batch index is clamped instead of branching around the
  // loads, which would send the register arrays through scratch memory)
  constexpr int U = 8;
  const int nb = (s1 > s0) ? (s1 - s0) / U : 0;
  avec avA[U], avB[U];
  float2 bvA[U], bvB[U];
#define TN_LOAD(av, bv, batch)                                                        \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                      \
    const long long k = 4 * (s0 + (batch) * U + u) + kg;                               \
    av[u] = *reinterpret_cast<const avec*>(pa + k * P.lda);                            \
    bv[u] = *reinterpret_cast<const float2*>(pb + k * P.ldb);                          \
  }                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#define TN_STEP(a_, b_)                                                                \
  _Pragma("unroll") for (int x = 0; x < MT; ++x) {                                     \
    acc[x][0] = mfma16(tn_elem<MT>(a_, x), b_.x, acc[x][0]);                           \
    acc[x][1] = mfma16(tn_elem<MT>(a_, x), b_.y, acc[x][1]);                           \
  }
#define TN_MFMA(av, bv)                                                                \
  _Pragma("unroll") for (int u = 0; u < U; ++u) { TN_STEP(av[u], bv[u]) }              \
  __builtin_amdgcn_sched_barrier(0);
  if (nb > 0) {
    TN_LOAD(avA, bvA, 0)
    for (int bt = 0; bt < nb; bt += 2) {
      TN_LOAD(avB, bvB, min(bt + 1, nb - 1))
      TN_MFMA(avA, bvA)
      TN_LOAD(avA, bvA, min(bt + 2, nb - 1))
      if (bt + 1 < nb) { TN_MFMA(avB, bvB) }
    }
  }
  int sb = s0 + nb * U;
  for (; sb < s1; ++sb) {                              // fewer than U full steps left
    const long long k = 4 * sb + kg;
    const avec av = *reinterpret_cast<const avec*>(pa + k * P.lda);
    const float2 bv = *reinterpret_cast<const float2*>(pb + k * P.ldb);
    TN_STEP(av, bv)
  }
  if (w == 3 && (P.K & 3)) {                           // partial last step: zero the rows past K
    const long long k = 4 * steps + kg;
    const bool kok = k < P.K;
    const long long kc = kok ? k : 0;
    avec av = *reinterpret_cast<const avec*>(pa + kc * P.lda);
    const float2 bv = *reinterpret_cast<const float2*>(pb + kc * P.ldb);
    av.x = kok ? av.x : 0.0f; av.y = kok ? av.y : 0.0f; av.z = kok ? av.z : 0.0f;
    if constexpr (MT == 4) av.w = kok ? av.w : 0.0f;
    TN_STEP(av, bv)
  }
#undef TN_MFMA
#undef TN_STEP
#undef TN_LOAD
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][2 * x + y][lane * 4 + r] = acc[x][y][r];
  __syncthreads();
  // wave w finishes the two tiles of row-tile x = w: element (lane, r) is row 4*kg + r, column i of a 16 x 16 tile
  if (w >= MT) return;
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = lane * 4 + r, t = 2 * w + y;
      const float v = ((red[0][t][e] + red[1][t][e]) + red[2][t][e]) + red[3][t][e];
      const int m = m0 + MT * (4 * kg + r) + w, n = n0 + 2 * i + y;
      if (m < P.M && n < P.N) P.C[(long long)m * P.ldc + n] = v;
    }
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
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(N, 64)), dim3(256), 0, st,
                     (const float*)ws, out, (int)N, RS);
  SLU_CHECK_LAUNCH("colsum_final_kernel");
  return SLU_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped small-M GEMMs in ONE launch (the seq2seq decoder's per-step products: 64 utterances x a few hundred
// columns x K <= 768, reference models.py:427-485): up to four independent problems
//     C_q (M x N_q) = [C_q +] A_q (M x K_q, k fast) * op(B_q) + bias_q,
//     op(B) = B^T for mode 0 (B is (N x K), k fast: a Linear / GRUCell weight as stored), B for mode 1 (B is (K x N),
//     n fast: the same weights in the data-gradient products).
// These launches are latency-bound, so the kernel is built around ONE memory round trip: a 64 x 16 output tile per
// workgroup, EIGHT waves that split K in 16-wide chunks (wave w takes chunks w, w + 8, ...), every load of a pass of
// eight chunks per wave issued before the first MFMA (operands straight from global memory / L2: a 16-byte A load per
// lane and m-tile feeds four k steps — the MFMA's k index is remapped so that lane group kg owns k = 4 kg .. 4 kg + 3
// of a chunk), exact fp32 MFMA (v_mfma_f32_16x16x4_f32), cross-wave reduction in LDS in fixed wave order (deterministic),
// bias / accumulate in the epilogue.  K % 4 == 0, lda % 4 == 0, 16-byte aligned A (and B, ldb % 4 == 0, for mode 0).
// ---------------------------------------------------------------------------------------------------------------------
struct SmallProblem {
  const float* A; const float* B; float* C; const float* bias;
  long long lda, ldb, ldc;
  int M, N, K, mode, accumulate;
  int tiles_n, tile_end;      // column tiles; exclusive prefix of (m tiles x n tiles)
};
struct SmallArgs { SmallProblem p[4]; int count; };

constexpr int SM_WAVES = 8;
constexpr int SM_CHUNKS = 4;    // chunks of 16 k per wave and pass (8 waves x 4 x 16 = 512 k per pass)

__global__ void __launch_bounds__(SM_WAVES * 64, 4)       // two workgroups per CU (<= 128 VGPRs)
gemm_small_batched_kernel(const SmallArgs a) {
  __shared__ float red[SM_WAVES][64 * 16];
  int q = 0;
  while (q + 1 < a.count && (int)blockIdx.x >= a.p[q].tile_end) ++q;
  const SmallProblem& P = a.p[q];
  const int t = blockIdx.x - (q ? a.p[q - 1].tile_end : 0);
  const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
  const int m0 = tm * 64, n0 = tn * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kg = lane >> 4;
  const int nchunks = (P.K + 15) >> 4;

  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // row pointers (clamped: rows / columns outside the problem read a valid address and are masked to zero)
  const float* arow[4];
  float amask[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + mt * 16 + i;
    amask[mt] = m < P.M ? 1.0f : 0.0f;
    arow[mt] = P.A + (long long)min(m, P.M - 1) * P.lda;
  }
  const int n = n0 + i;
  const float bmask = n < P.N ? 1.0f : 0.0f;
  const int nc = min(n, P.N - 1);

  for (int c0 = w; c0 < nchunks; c0 += SM_WAVES * SM_CHUNKS) {
    // chunks this wave really has in this pass (wave-uniform): the others cost neither loads nor MFMAs (a first
    // version issued all of them masked: 10 us per workgroup whatever K)
    const int nv = min(SM_CHUNKS, (nchunks - c0 + SM_WAVES - 1) / SM_WAVES);
    float4 fa[SM_CHUNKS][4];
    float4 fb[SM_CHUNKS];
#pragma unroll
    for (int u = 0; u < SM_CHUNKS; ++u) {
      if (u >= nv) break;
      const int c = c0 + u * SM_WAVES;
      const int k = c * 16 + 4 * kg;                       // this lane's four k of the chunk
      const bool ok = k < P.K;                             // K % 4 == 0: a lane's four k are all in or all out
      const int kc = ok ? k : 0;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float4 v = *reinterpret_cast<const float4*>(arow[mt] + kc);
        const float f = ok ? amask[mt] : 0.0f;
        fa[u][mt] = make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
      }
      float4 b;
      if (P.mode == 0) {
        b = *reinterpret_cast<const float4*>(P.B + (long long)nc * P.ldb + kc);
      } else {
        const float* bp = P.B + (long long)kc * P.ldb + nc;
        b = make_float4(bp[0], bp[P.ldb], bp[2 * P.ldb], bp[3 * P.ldb]);
      }
      const float f = ok ? bmask : 0.0f;
      fb[u] = make_float4(b.x * f, b.y * f, b.z * f, b.w * f);
    }
#pragma unroll
    for (int u = 0; u < SM_CHUNKS; ++u) {
      if (u >= nv) break;
      // k sub-step outermost: the four m-tiles are four independent accumulator chains
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma16(fa[u][mt].x, fb[u].x, acc[mt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma16(fa[u][mt].y, fb[u].y, acc[mt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma16(fa[u][mt].z, fb[u].z, acc[mt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma16(fa[u][mt].w, fb[u].w, acc[mt]);
    }
  }
  // D[row = 4 kg + r][col = i] of m-tile mt -> red[w][(mt * 16 + 4 kg + r) * 16 + i]
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][(mt * 16 + 4 * kg + r) * 16 + i] = acc[mt][r];
  __syncthreads();
  for (int e = tid; e < 64 * 16; e += SM_WAVES * 64) {
    float s = red[0][e];
#pragma unroll
    for (int v = 1; v < SM_WAVES; ++v) s += red[v][e];
    const int m = m0 + (e >> 4), nn = n0 + (e & 15);
    if (m < P.M && nn < P.N) {
      if (P.bias) s += P.bias[nn];
      float* c = P.C + (long long)m * P.ldc + nn;
      *c = P.accumulate ? *c + s : s;
    }
  }
}

static void split_plan(int64_t M, int64_t N, int64_t K, int* KS, int* kper, int* wt) {
  // measured on MI355X: the 64-tile wins or ties up to 8192^2 x 1024 (more waves per SIMD hide the
  // per-k-tile latency chain; L2 absorbs the extra operand re-reads); 128-tiles only pay off when
  // both the grid and K are large.
  *wt = (cdiv(M, 128) * cdiv(N, 128) >= 1024 && K >= 2048) ? 4 : 2;
  const int64_t bm = 32 * *wt;
  const int64_t tiles = cdiv(M, bm) * cdiv(N, bm);
  int64_t ks = 1;
  if (tiles < 128 && K >= 512) {
    // two co-resident workgroups per CU (each is four waves with a barrier per k-tile: the second one fills
    // the first one's staging phases) and never a partial second round: at most 512 workgroups.  The plan is a
    // function of the shape only: the summation order (hence every bit of the result) must not depend on which
    // stream / CU partition the GEMM is launched on (pipelined == sequential training, bit for bit).
    ks = 512 / tiles;
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
  int KS, kper, wt;
  split_plan(M, N, K, &KS, &kper, &wt);
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
  int KS, kper, wt;
  split_plan(M, N, K, &KS, &kper, &wt);
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
  const int bm = 32 * wt;
  dim3 grid((unsigned)cdiv(N, bm), (unsigned)cdiv(M, bm), (unsigned)KS);
  SLU_REQUIRE(grid.y <= 65535, "slu_gemm_f32: M too large for one launch");
  const bool akf = (a_cs == 1) || (a_rs != 1);     // k-fast mapping unless M is the contiguous dim
  const bool bkf = (b_rs == 1) || (b_cs != 1);     // B(k,n): k contiguous when b_rs == 1
#define SLU_GEMM_LAUNCH(AK, BK_, W) \
  hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, W>), grid, dim3(GM_THREADS), 0, st, p)
  if (wt == 4) {
    if (akf && bkf) SLU_GEMM_LAUNCH(true, true, 4);
    else if (akf && !bkf) SLU_GEMM_LAUNCH(true, false, 4);
    else if (!akf && bkf) SLU_GEMM_LAUNCH(false, true, 4);
    else SLU_GEMM_LAUNCH(false, false, 4);
  } else {
    if (akf && bkf) SLU_GEMM_LAUNCH(true, true, 2);
    else if (akf && !bkf) SLU_GEMM_LAUNCH(true, false, 2);
    else if (!akf && bkf) SLU_GEMM_LAUNCH(false, true, 2);
    else SLU_GEMM_LAUNCH(false, false, 2);
  }
#undef SLU_GEMM_LAUNCH
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

extern "C" int slu_gemm_tn_batched(const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
                                   float* const* C, const int64_t* ldc, const int64_t* M, const int64_t* N,
                                   const int64_t* K, int64_t count, const float* rowsum_src, int64_t rowsum_rows,
                                   int64_t rowsum_cols, float* rowsum_dst, void* stream) {
  SLU_REQUIRE(A && B && C && lda && ldb && ldc && M && N && K, "slu_gemm_tn_batched: null pointer");
  SLU_REQUIRE((rowsum_src == nullptr) == (rowsum_dst == nullptr), "slu_gemm_tn_batched: rowsum_src and rowsum_dst go together");
  SLU_REQUIRE(!rowsum_src || (rowsum_rows >= 1 && rowsum_cols >= 1 && rowsum_rows < (1LL << 30) && rowsum_cols < (1LL << 30)),
              "slu_gemm_tn_batched: bad row-sum size");
  SLU_REQUIRE(count >= 1 && count <= 4, "slu_gemm_tn_batched: 1..4 problems per call");
  // tile height: 64 rows (16-byte A loads) where the shapes allow it, else 48 rows (12-byte loads).  Measured for the
  // intent layer (144 tiles of 64 rows vs 192 of 48 on 64 CUs): 52 vs 57 us — the even spread of the 48-row tiles does
  // not pay for their extra B traffic per MAC.
  int mt = 4;
  {
    bool ok4 = true, ok3 = true;
    for (int q = 0; q < (int)count; ++q) {
      ok4 = ok4 && ((M[q] | lda[q]) & 3) == 0 && ((uintptr_t)A[q] & 15) == 0;
      ok3 = ok3 && (M[q] % 3) == 0;
    }
    if (!ok4 && !ok3)
      SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gemm_tn_batched: M must be a multiple of 3, or M and lda multiples of 4 with A 16-byte aligned");
    if (!ok4) mt = 3;
    // 64 x 64 tiles (mt = 8: gemm_tn_small_wide_kernel) where B allows 16-byte loads and the tiles still make a few
    // dozen workgroups: the kernel is bound by the operand bytes a CU pulls through its L1 per MAC — measured for the
    // intent layer's three matrices on 128 CUs: 64 x 64 tiles 26.6 us, 64 x 32 tiles 34.6 us, 32 x 32 tiles 45.1 us
    // (SLU_TN_WIDE=0: 64 x 32 tiles always)
    static const int wide_env = [] { const char* e = getenv("SLU_TN_WIDE"); return e ? atoi(e) : 1; }();
    if (ok4 && wide_env) {
      bool okw = true;
      int64_t wt = 0;
      for (int q = 0; q < (int)count; ++q) {
        okw = okw && ((N[q] | ldb[q]) & 3) == 0 && ((uintptr_t)B[q] & 15) == 0;
        wt += cdiv(M[q], 64) * cdiv(N[q], 64);
      }
      if (okw && wt >= 32) mt = 8;
    }
  }
  TnArgs a;
  int tiles = 0;
  for (int q = 0; q < (int)count; ++q) {
    SLU_REQUIRE(A[q] && B[q] && C[q] && M[q] > 0 && N[q] > 0 && K[q] > 0, "slu_gemm_tn_batched: bad problem %d", q);
    SLU_REQUIRE(M[q] < (1LL << 30) && N[q] < (1LL << 30) && K[q] < (1LL << 30), "slu_gemm_tn_batched: size overflow");
    if ((N[q] | ldb[q]) & 1 || ((uintptr_t)B[q] & 7))
      SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gemm_tn_batched: N and ldb must be even (B 8-byte aligned)");
    a.p[q].A = A[q]; a.p[q].B = B[q]; a.p[q].C = C[q];
    a.p[q].lda = lda[q]; a.p[q].ldb = ldb[q]; a.p[q].ldc = ldc[q];
    a.p[q].M = (int)M[q]; a.p[q].N = (int)N[q]; a.p[q].K = (int)K[q];
    const int tile_n = mt == 8 ? 64 : 32, tile_m = mt == 8 ? 64 : 16 * mt;
    a.p[q].tiles_n = (int)cdiv(N[q], tile_n);
    tiles += (int)(cdiv(M[q], tile_m) * cdiv(N[q], tile_n));
    a.p[q].tile_end = tiles;
  }
  a.count = (int)count;
  a.tiles = tiles;
  a.rs_src = rowsum_src; a.rs_dst = rowsum_dst; a.rs_rows = (int)rowsum_rows; a.rs_cols = (int)rowsum_cols;
  const int extra = rowsum_src ? (int)cdiv(rowsum_cols, 256) : 0;
  a.rs_blocks = extra;
  if (mt == 8)
    hipLaunchKernelGGL(gemm_tn_small_wide_kernel, dim3((unsigned)(tiles + extra)), dim3(256), 0, (hipStream_t)stream, a);
  else if (mt == 4)
    hipLaunchKernelGGL(gemm_tn_small_kernel, dim3((unsigned)(tiles + extra)), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(gemm_tn_small_mt_kernel<3>, dim3((unsigned)(tiles + extra)), dim3(256), 0, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("gemm_tn_small_kernel");
  return SLU_OK;
}

// Split-K form of slu_gemm_tn_batched for long k ranges (include/slu_hip.h).
static int tn_splitk_factor(int64_t tiles, int64_t kmin, int64_t max_wg = 0) {
  // ONE round of workgroups: two fit a CU (216 VGPRs), 512 on the chip — 654 workgroups (nine splits of a layer's 72 tiles)
  // ran as two rounds, the second a quarter full: 168 us per launch against 141-150 with seven; at least 256 k rows per
  // split, at most 16.  (Eight splits put split ks on XCD ks — linear id tile * 8 + ks — so that a k row crosses HBM -> L2
  // once: FETCH_SIZE 266 -> 91 MB per launch, L2 hits 39 -> 75 %, and the launch got SLOWER, 174 us: nine eighths of a round,
  // and the bound is not HBM but the L1's outstanding requests — k-slow 16-byte loads, 8 MAC per operand byte, ~16 B/clk per
  // CU needed at the MFMA peak; profiles/r04_am_pmc_tn_splitk.txt.)
  // max_wg > 0 (slu_gemm_tn_batched_splitk_wg): a smaller budget — a launch that runs BESIDE a latency-bound recurrence on a
  // branch of its own must leave whole CUs empty for it (192-216 workgroups spread one per CU)
  int64_t ks = (max_wg > 0 ? max_wg : 512) / tiles;
  ks = ks < 1 ? 1 : ks;
  if (ks > 16) ks = 16;
  const int64_t by_k = kmin / 256 < 1 ? 1 : kmin / 256;
  return (int)(ks < by_k ? ks : by_k);
}

static bool tn_splitk_shapes_ok(const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
                                const int64_t* M, const int64_t* N, int64_t count) {
  for (int q = 0; q < (int)count; ++q) {
    if (((M[q] | lda[q]) & 3) || ((uintptr_t)A[q] & 15)) return false;
    if (((N[q] | ldb[q]) & 3) || ((uintptr_t)B[q] & 15)) return false;
  }
  return true;
}

extern "C" size_t slu_gemm_tn_splitk_workspace_bytes_wg(const int64_t* M, const int64_t* N, const int64_t* K, int64_t count,
                                                        int64_t max_workgroups) {
  if (!M || !N || !K || count < 1 || count > 4) return 0;
  int64_t tiles = 0, kmin = K[0];
  for (int q = 0; q < (int)count; ++q) {
    tiles += cdiv(M[q], 64) * cdiv(N[q], 64);
    kmin = K[q] < kmin ? K[q] : kmin;
  }
  return (size_t)tiles * tn_splitk_factor(tiles, kmin, max_workgroups) * 4096 * sizeof(float);
}

extern "C" size_t slu_gemm_tn_splitk_workspace_bytes(const int64_t* M, const int64_t* N, const int64_t* K, int64_t count) {
  return slu_gemm_tn_splitk_workspace_bytes_wg(M, N, K, count, 0);
}

extern "C" int slu_gemm_tn_batched_splitk_wg(const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
                                             float* const* C, const int64_t* ldc, const int64_t* M, const int64_t* N,
                                             const int64_t* K, int64_t count, const float* rowsum_src, int64_t rowsum_rows,
                                             int64_t rowsum_cols, float* rowsum_dst, void* workspace, size_t workspace_bytes,
                                             uint32_t* tickets, int64_t n_tickets, int64_t max_workgroups, void* stream) {
  SLU_REQUIRE(A && B && C && lda && ldb && ldc && M && N && K, "slu_gemm_tn_batched_splitk: null pointer");
  SLU_REQUIRE((rowsum_src == nullptr) == (rowsum_dst == nullptr), "slu_gemm_tn_batched_splitk: rowsum_src and rowsum_dst go together");
  SLU_REQUIRE(!rowsum_src || (rowsum_rows >= 1 && rowsum_cols >= 1 && rowsum_rows < (1LL << 30) && rowsum_cols < (1LL << 30)),
              "slu_gemm_tn_batched_splitk: bad row-sum size");
  SLU_REQUIRE(count >= 1 && count <= 4, "slu_gemm_tn_batched_splitk: 1..4 problems per call");
  if (!tn_splitk_shapes_ok(A, lda, B, ldb, M, N, count))
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gemm_tn_batched_splitk: M, lda, N, ldb must be multiples of 4 and A, B 16-byte aligned");
  TnArgs a;
  int tiles = 0;
  int64_t kmin = K[0];
  for (int q = 0; q < (int)count; ++q) {
    SLU_REQUIRE(A[q] && B[q] && C[q] && M[q] > 0 && N[q] > 0 && K[q] > 0, "slu_gemm_tn_batched_splitk: bad problem %d", q);
    SLU_REQUIRE(M[q] < (1LL << 30) && N[q] < (1LL << 30) && K[q] < (1LL << 30), "slu_gemm_tn_batched_splitk: size overflow");
    a.p[q].A = A[q]; a.p[q].B = B[q]; a.p[q].C = C[q];
    a.p[q].lda = lda[q]; a.p[q].ldb = ldb[q]; a.p[q].ldc = ldc[q];
    a.p[q].M = (int)M[q]; a.p[q].N = (int)N[q]; a.p[q].K = (int)K[q];
    a.p[q].tiles_n = (int)cdiv(N[q], 64);
    tiles += (int)(cdiv(M[q], 64) * cdiv(N[q], 64));
    a.p[q].tile_end = tiles;
    kmin = K[q] < kmin ? K[q] : kmin;
  }
  SLU_REQUIRE(max_workgroups >= 0, "slu_gemm_tn_batched_splitk: negative workgroup budget");
  const int ksplit = tn_splitk_factor(tiles, kmin, max_workgroups);
  if (ksplit > 1) {
    const size_t need = (size_t)tiles * ksplit * 4096 * sizeof(float);
    if (!workspace || workspace_bytes < need)
      SLU_FAIL(SLU_ERR_WORKSPACE, "slu_gemm_tn_batched_splitk: workspace too small (%zu < %zu)", workspace_bytes, need);
    SLU_REQUIRE(tickets && n_tickets >= tiles, "slu_gemm_tn_batched_splitk: needs %d zeroed ticket words", tiles);
  }
  a.count = (int)count;
  a.tiles = tiles;
  a.rs_src = rowsum_src; a.rs_dst = rowsum_dst; a.rs_rows = (int)rowsum_rows; a.rs_cols = (int)rowsum_cols;
  const int extra = rowsum_src ? (int)cdiv(rowsum_cols, 256) : 0;
  a.rs_blocks = extra;
  hipLaunchKernelGGL(gemm_tn_wide_splitk_kernel, dim3((unsigned)(tiles * ksplit + extra)), dim3(256), 0, (hipStream_t)stream,
                     a, ksplit, reinterpret_cast<float*>(workspace), (unsigned*)tickets);
  SLU_CHECK_LAUNCH("gemm_tn_wide_splitk_kernel");
  return SLU_OK;
}

extern "C" int slu_gemm_tn_batched_splitk(const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
                                          float* const* C, const int64_t* ldc, const int64_t* M, const int64_t* N,
                                          const int64_t* K, int64_t count, const float* rowsum_src, int64_t rowsum_rows,
                                          int64_t rowsum_cols, float* rowsum_dst, void* workspace, size_t workspace_bytes,
                                          uint32_t* tickets, int64_t n_tickets, void* stream) {
  return slu_gemm_tn_batched_splitk_wg(A, lda, B, ldb, C, ldc, M, N, K, count, rowsum_src, rowsum_rows, rowsum_cols, rowsum_dst,
                                       workspace, workspace_bytes, tickets, n_tickets, 0, stream);
}

extern "C" int slu_colsum_f32(const float* X, int64_t x_rs, float* out, int64_t M, int64_t N,
                              int accumulate, void* stream) {
  SLU_REQUIRE(X && out && M > 0 && N > 0, "slu_colsum_f32: bad argument");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream,
                     X, (long long)x_rs, out, (int)M, (int)N, accumulate);
  SLU_CHECK_LAUNCH("colsum_kernel");
  return SLU_OK;
}

extern "C" int slu_gemm_small_batched(const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
                                      const int* mode, float* const* C, const int64_t* ldc, const float* const* bias,
                                      const int* accumulate, const int64_t* M, const int64_t* N, const int64_t* K,
                                      int64_t count, void* stream) {
  SLU_REQUIRE(A && lda && B && ldb && mode && C && ldc && bias && accumulate && M && N && K, "slu_gemm_small_batched: null pointer");
  SLU_REQUIRE(count >= 1 && count <= 4, "slu_gemm_small_batched: 1..4 problems per call");
  SmallArgs a;
  int tiles = 0;
  for (int q = 0; q < (int)count; ++q) {
    SLU_REQUIRE(A[q] && B[q] && C[q] && M[q] > 0 && N[q] > 0 && K[q] > 0, "slu_gemm_small_batched: bad problem %d", q);
    SLU_REQUIRE(M[q] < (1LL << 24) && N[q] < (1LL << 24) && K[q] < (1LL << 24), "slu_gemm_small_batched: size overflow");
    SLU_REQUIRE(mode[q] == 0 || mode[q] == 1, "slu_gemm_small_batched: mode must be 0 (B is N x K) or 1 (B is K x N)");
    if ((K[q] & 3) || (lda[q] & 3) || ((uintptr_t)A[q] & 15) || (mode[q] == 0 && ((ldb[q] & 3) || ((uintptr_t)B[q] & 15))))
      SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gemm_small_batched: problem %d needs K %% 4 == 0, lda (ldb for mode 0) %% 4 == 0 and "
               "16-byte aligned operands", q);
    SmallProblem& P = a.p[q];
    P.A = A[q]; P.B = B[q]; P.C = C[q]; P.bias = bias[q];
    P.lda = lda[q]; P.ldb = ldb[q]; P.ldc = ldc[q];
    P.M = (int)M[q]; P.N = (int)N[q]; P.K = (int)K[q]; P.mode = mode[q]; P.accumulate = accumulate[q];
    P.tiles_n = (int)cdiv(N[q], 16);
    tiles += (int)(cdiv(M[q], 64) * P.tiles_n);
    P.tile_end = tiles;
  }
  a.count = (int)count;
  hipLaunchKernelGGL(gemm_small_batched_kernel, dim3((unsigned)tiles), dim3(SM_WAVES * 64), 0, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("gemm_small_batched_kernel");
  return SLU_OK;
}
