// Split-precision (bf16 MFMA) GEMM for the input projections of the FROZEN GRU layers:
//   C(m, n) = sum_k A(m, k) W(n, k) + bias(n)          (x W_ih^T + b_ih of torch.nn.GRU, models.py:232/:262)
// A: NS bf16 planes (M x Kp, Kp = K rounded up to 32, zero padded) written by the producing stage;
// W: (N x K) fp32 weights, split and re-laid in MFMA B-fragment order once per launch by gemm_bf_pack;
// C: fp32 (the recurrence's gx).  NS = 3: fp32-class result from six bf16 products (slu_bf16.h), NS = 1: bf16.
//
// Workgroup tile 128 x 64, 4 waves as 2 x 2 (64 x 32 each = 4 x 2 tiles of v_mfma_f32_16x16x32_bf16), one
// 32-wide k-chunk per stage, three workgroups per CU:
//   * A goes through LDS (double buffered, swizzled 64-byte rows -> conflict-free ds_read_b128 fragments),
//     filled by LDS-DMA (global_load_lds_dwordx4, swizzle on the source address) during the previous chunk's MFMAs;
//   * W fragments are read straight from L2 in fragment order (1 KiB coalesced per wave-load, one chunk ahead):
//     the packed weights are <= 1.2 MB and shared by every workgroup;
//   * the workgroup -> tile map is XCD-aware (the N/128 column tiles of one row tile run on one XCD, whose L2
//     then serves five of the six reads of that A tile);
//   * the epilogue adds the bias and writes full 256-byte row segments (per-wave LDS transpose).
// fp32 output makes the K = 60 projection HBM-write bound (708 MB per 768-sequence super-batch) and the
// K = 256 ones MFMA bound at 6/16 of the fp32-MFMA cycle count.
#include "slu_bf16.h"

#include <cstdlib>

namespace slu {

constexpr int GB_BM = 128, GB_BN = 64, GB_THREADS = 256;     // wave tile 64 x 32: 152 live VGPRs -> 3 waves / SIMD

struct GemmBfParams {
  const unsigned short* A;    // planes: A + p * a_plane, rows of lda bf16
  long long a_plane, lda;
  const uint4* wp;            // packed W: [plane][kc][nt][lane] uint4 (8 bf16)
  const float* bias;          // (N) or null
  float* C; long long ldc;
  int M, N, KC;               // KC = number of 32-wide k-chunks
};

// W (N x K, row stride ldw) fp32 -> packed bf16 planes in B-fragment order, zero beyond N / K
template <int NS>
__global__ void __launch_bounds__(256)
gemm_bf_pack_kernel(const float* __restrict__ W, long long ldw, long long w_cs, uint4* __restrict__ wp, int N, int K, int KC, int NT) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (kc, nt, lane)
  if (idx >= KC * NT * 64) return;
  const int lane = idx & 63, nt = (idx >> 6) % NT, kc = (idx >> 6) / NT;
  const int n = nt * 16 + (lane & 15), k0 = kc * 32 + (lane >> 4) * 8;
  unsigned short h[NS][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = (n < N && k0 + e < K) ? W[(long long)n * ldw + (long long)(k0 + e) * w_cs] : 0.0f;
    unsigned short s[NS];
    split_terms<NS>(v, s);
#pragma unroll
    for (int p = 0; p < NS; ++p) h[p][e] = s[p];
  }
#pragma unroll
  for (int p = 0; p < NS; ++p) {
    uint4 o;
    o.x = h[p][0] | ((unsigned)h[p][1] << 16); o.y = h[p][2] | ((unsigned)h[p][3] << 16);
    o.z = h[p][4] | ((unsigned)h[p][5] << 16); o.w = h[p][6] | ((unsigned)h[p][7] << 16);
    wp[(size_t)p * KC * NT * 64 + idx] = o;
  }
}

// fp32 (rows x K, row stride ldx) -> NS bf16 planes (rows x Kp), zero padded: the entry into the split format
// for tensors that were not produced by a split-writing epilogue
template <int NS>
__global__ void __launch_bounds__(256)
split_planes_kernel(const float* __restrict__ x, long long ldx, unsigned short* __restrict__ out, long long plane,
                    long long rows, int K, int Kp) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (row, 8-column group)
  const int groups = Kp / 8;
  if (idx >= rows * groups) return;
  const long long r = idx / groups;
  const int c0 = (int)(idx - r * groups) * 8;
  unsigned short h[NS][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = (c0 + e < K) ? x[r * ldx + c0 + e] : 0.0f;
    unsigned short s[NS];
    split_terms<NS>(v, s);
#pragma unroll
    for (int p = 0; p < NS; ++p) h[p][e] = s[p];
  }
#pragma unroll
  for (int p = 0; p < NS; ++p) {
    uint4 o;
    o.x = h[p][0] | ((unsigned)h[p][1] << 16); o.y = h[p][2] | ((unsigned)h[p][3] << 16);
    o.z = h[p][4] | ((unsigned)h[p][5] << 16); o.w = h[p][6] | ((unsigned)h[p][7] << 16);
    *reinterpret_cast<uint4*>(out + (size_t)p * plane + (size_t)r * Kp + c0) = o;
  }
}

template <int NS>
__global__ void __launch_bounds__(GB_THREADS, 3)
gemm_bf_kernel(const GemmBfParams p) {
  // A stage: NS planes x 128 rows x 64 B; two stages.  The epilogue reuses the memory (4 waves x 32 x 68 floats).
  constexpr int STAGE_U4 = NS * GB_BM * 4;
  constexpr int EPI_FLOATS = 4 * 32 * 36;
  constexpr int SMEM_BYTES = (2 * STAGE_U4 * 16 > EPI_FLOATS * 4) ? 2 * STAGE_U4 * 16 : EPI_FLOATS * 4;
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  uint4* const sA = reinterpret_cast<uint4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, kg = lane >> 4;
  const int NT = p.N / 16;

  // XCD-aware tile order (see gemm_f32_kernel): workgroup L runs on XCD L % 8
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
  const int m0 = by * GB_BM, n0 = bx * GB_BN;

  typedef Split<NS> SP;
  f32x4 accs[SP::NACC][4][2];
#pragma unroll
  for (int c = 0; c < SP::NACC; ++c)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) accs[c][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A staging by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write): a wave-load fills
  // 1 KiB of LDS linearly (lane l -> bytes [16 l, 16 l + 16) = row l / 4, slot l % 4 of a 16-row group), so the
  // swizzle is applied to the SOURCE address: the lane fetches the k-slot that belongs in its LDS slot.
  // Rows past M read row 0: rows of C are independent and those are never stored.
  const int srow = tid >> 2, sslot = tid & 3;
  const unsigned short* a_src[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = srow + 64 * h, m = m0 + r;
    a_src[h] = p.A + (size_t)(m < p.M ? m : 0) * p.lda + swz_slot(r, sslot) * 8;
  }
  const uint4* b_src = p.wp + (size_t)(n0 / 16 + wn * 2) * 64 + lane;
  const size_t b_plane = (size_t)p.KC * NT * 64, b_chunk = (size_t)NT * 64;
  int a_frag[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = wm * 64 + a * 16 + i;
    a_frag[a] = r * 4 + swz_slot(r, kg);
  }
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  uint4 rb0[NS][2], rb1[NS][2];
#define GB_DMA_A(kc_, buf_)                                                                          \
  _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) _Pragma("unroll") for (int h = 0; h < 2; ++h)    \
    __builtin_amdgcn_global_load_lds((gptr_t)(a_src[h] + (size_t)pl * p.a_plane + (kc_) * 32),       \
                                     (lptr_t)(sA + (buf_) * STAGE_U4 + (pl * GB_BM + 64 * h + 16 * wave) * 4), 16, 0, 0);
#define GB_FETCH_B(kc_, dst)                                                                         \
  _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) _Pragma("unroll") for (int b = 0; b < 2; ++b)    \
    dst[pl][b] = b_src[(size_t)pl * b_plane + (size_t)(kc_) * b_chunk + b * 64];
  // one k-chunk: W fragments `cur` were fetched a chunk ago; the next chunk's A tile (DMA into the other LDS
  // buffer, last read before the previous barrier) and W fragments (`nxt`) are in flight during this chunk's
  // MFMAs.  Two alternating fragment sets: no register copies.  Unconditional (the last chunk re-fetches
  // itself into the idle buffer): no control flow around the register arrays.
#define GB_STEP(kc_, buf, cur, nxt)   /* buf: literal 0 / 1 = (kc_) & 1 */                           \
  {                                                                                                  \
    const int kn = min((kc_) + 1, p.KC - 1);                                                         \
    /* fragment reads first: the compiler orders every LDS read behind a pending LDS-DMA (vmcnt(0)), \
       so the DMA of the next tile is issued after them and then flies during the MFMAs */           \
    uint4 fa[NS][4];                                                                                 \
    _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) _Pragma("unroll") for (int a = 0; a < 4; ++a)  \
      fa[pl][a] = sA[buf * STAGE_U4 + pl * GB_BM * 4 + a_frag[a]];                                   \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    GB_DMA_A(kn, 1 - buf) GB_FETCH_B(kn, nxt)                                                        \
    __builtin_amdgcn_sched_barrier(0);   /* the loads are issued HERE, ahead of the MFMAs */          \
    _Pragma("unroll") for (int q = 0; q < SP::NPAIR; ++q) {                                          \
      _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)    \
        accs[SP::ACC(q)][a][b] = mfma_split<NS>(fa[SP::PA(q)][a], cur[SP::PB(q)][b], accs[SP::ACC(q)][a][b]); \
    }                                                                                                \
    __builtin_amdgcn_sched_barrier(0);   /* ... and the MFMAs stay on this side of the barrier */     \
    __syncthreads();   /* (drains the DMA and the W loads: both are needed right after) */           \
  }
  GB_DMA_A(0, 0)
  GB_FETCH_B(0, rb0)
  __syncthreads();
  for (int kc = 0; kc < p.KC; kc += 2) {
    GB_STEP(kc, 0, rb0, rb1)
    if (kc + 1 < p.KC) GB_STEP(kc + 1, 1, rb1, rb0)
  }
#undef GB_STEP
#undef GB_FETCH_B
#undef GB_DMA_A
  f32x4 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = split_result<NS>(accs[0][a][b], accs[SP::NACC - 1][a][b]);

  // epilogue: two passes of 32 rows per wave through LDS, float4 row-contiguous stores (+ bias)
  float* const sC = reinterpret_cast<float*>(smem) + wave * (32 * 36);
  const int col4 = (lane & 7) * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  const int ncol = n0 + wn * 32 + col4;
  if (p.bias) { bv.x = p.bias[ncol]; bv.y = p.bias[ncol + 1]; bv.z = p.bias[ncol + 2]; bv.w = p.bias[ncol + 3]; }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(a2 * 16 + 4 * kg + r) * 36 + b * 16 + i] = acc[2 * h + a2][b][r];
    // a wave only reads what it wrote itself: no workgroup barrier needed, LDS ops of a wave are ordered
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = (lane >> 3) + 8 * it;
      const int m = m0 + wm * 64 + h * 32 + rl;
      const float4 v = *reinterpret_cast<const float4*>(&sC[rl * 36 + col4]);
      if (m < p.M)
        *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + ncol) = make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w);
    }
  }
}


// ---- K <= 64 (one or two k-chunks: the first frozen GRU layer's projection, K = 60 -> 768 outputs) ---------------------
// That shape is HBM-WRITE bound (a 128-row panel reads 48 KB of A planes and writes 384 KB of fp32), and in the tiled
// kernel above each 128 x 64 tile is a latency chain of its own (A DMA -> two chunks -> stores: 2.3 TB/s).  Here a
// workgroup owns a 128-row PANEL: its A planes (all of K) are fetched once by LDS-DMA and stay in LDS, then it walks
// the N/64 column tiles: W fragments and the bias of tile j+1 are in flight (registers, two alternating sets) during
// the MFMAs and the stores of tile j, so the only thing a panel ever waits for after its first tile is the store
// queue.  No LDS-DMA inside the loop, hence no conservative vmcnt(0) before LDS reads; no workgroup barrier either
// (the epilogue buffer is per wave).  Two workgroups per CU (66 KB of LDS each).
template <int NS, int KC>
__global__ void __launch_bounds__(GB_THREADS, 2)
gemm_bf_panel_kernel(const GemmBfParams p) {
  constexpr int STAGE_U4 = NS * GB_BM * 4;
  constexpr int EPI_FLOATS = 4 * 32 * 36;
  __shared__ __attribute__((aligned(16))) char smem[KC * STAGE_U4 * 16 + EPI_FLOATS * 4];
  uint4* const sA = reinterpret_cast<uint4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, kg = lane >> 4;
  const int NT = p.N / 16, NJ = p.N / GB_BN;
  const int m0 = blockIdx.x * GB_BM;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  {
    const int srow = tid >> 2, sslot = tid & 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = srow + 64 * h, m = m0 + r;
      const unsigned short* src = p.A + (size_t)(m < p.M ? m : 0) * p.lda + swz_slot(r, sslot) * 8;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
          __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)pl * p.a_plane + kc * 32),
                                           (lptr_t)(sA + kc * STAGE_U4 + (pl * GB_BM + 64 * h + 16 * wave) * 4), 16, 0, 0);
    }
  }
  const uint4* b_src = p.wp + (size_t)(wn * 2) * 64 + lane;
  const size_t b_plane = (size_t)p.KC * NT * 64, b_chunk = (size_t)NT * 64;
  int a_frag[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = wm * 64 + a * 16 + i;
    a_frag[a] = r * 4 + swz_slot(r, kg);
  }
  const int col4 = (lane & 7) * 4;
  float* const sC = reinterpret_cast<float*>(smem + KC * STAGE_U4 * 16) + wave * (32 * 36);
  typedef Split<NS> SP;
  uint4 rb0[KC][NS][2], rb1[KC][NS][2];
  float4 bias0 = make_float4(0.f, 0.f, 0.f, 0.f), bias1 = bias0;
  // W fragments + bias of column tile j_ (clamped: the last tile prefetches itself; no branch around the arrays)
#define GP_FETCH(j_, dst, bdst)                                                                         \
  {                                                                                                     \
    const int jj = min((j_), NJ - 1);                                                                   \
    _Pragma("unroll") for (int kc = 0; kc < KC; ++kc) _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                     \
        dst[kc][pl][b] = b_src[(size_t)pl * b_plane + (size_t)kc * b_chunk + (size_t)(jj * 4 + b) * 64]; \
    if (p.bias) bdst = *reinterpret_cast<const float4*>(p.bias + jj * GB_BN + wn * 32 + col4);          \
  }
#define GP_TILE(j_, cur, bcur, nxt, bnxt)                                                               \
  {                                                                                                     \
    GP_FETCH((j_) + 1, nxt, bnxt)                                                                       \
    __builtin_amdgcn_sched_barrier(0);   /* next tile's loads are issued HERE, ahead of this tile's MFMAs */ \
    f32x4 accs[SP::NACC][4][2];                                                                         \
    _Pragma("unroll") for (int c = 0; c < SP::NACC; ++c)                                                \
      _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)       \
        accs[c][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};                                                      \
    _Pragma("unroll") for (int kc = 0; kc < KC; ++kc) {                                                 \
      uint4 fa[NS][4];                                                                                  \
      _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) _Pragma("unroll") for (int a = 0; a < 4; ++a)   \
        fa[pl][a] = sA[kc * STAGE_U4 + pl * GB_BM * 4 + a_frag[a]];                                     \
      _Pragma("unroll") for (int q = 0; q < SP::NPAIR; ++q) {                                           \
        _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)     \
          accs[SP::ACC(q)][a][b] = mfma_split<NS>(fa[SP::PA(q)][a], cur[kc][SP::PB(q)][b], accs[SP::ACC(q)][a][b]); \
      }                                                                                                 \
    }                                                                                                   \
    f32x4 acc[4][2];                                                                                    \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)         \
      acc[a][b] = split_result<NS>(accs[0][a][b], accs[SP::NACC - 1][a][b]);                            \
    const int ncol = (j_) * GB_BN + wn * 32 + col4;                                                     \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                     \
      _Pragma("unroll") for (int a2 = 0; a2 < 2; ++a2) _Pragma("unroll") for (int b = 0; b < 2; ++b)    \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
          sC[(a2 * 16 + 4 * kg + r) * 36 + b * 16 + i] = acc[2 * h + a2][b][r];                         \
      _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                                \
        const int rl = (lane >> 3) + 8 * it;                                                            \
        const int m = m0 + wm * 64 + h * 32 + rl;                                                       \
        const float4 v = *reinterpret_cast<const float4*>(&sC[rl * 36 + col4]);                         \
        if (m < p.M)                                                                                    \
          *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + ncol) =                                  \
              make_float4(v.x + bcur.x, v.y + bcur.y, v.z + bcur.z, v.w + bcur.w);                      \
      }                                                                                                 \
    }                                                                                                   \
  }
  GP_FETCH(0, rb0, bias0)
  __syncthreads();                       // the A panel (all waves' DMA) and the first W fragments have landed
  for (int j = 0; j < NJ; j += 2) {
    GP_TILE(j, rb0, bias0, rb1, bias1)
    if (j + 1 < NJ) GP_TILE(j + 1, rb1, bias1, rb0, bias0)
  }
#undef GP_TILE
#undef GP_FETCH
}


// ---- 64 < K <= 256: row panels of 96 rows, A resident for the whole panel -----------------------------------------------
// The tiled kernel moves 48 KB through the CU's L2 port per 768 MFMA cycles and exposes one L2/HBM round trip per k-chunk
// (its LDS reads wait for ALL pending LDS-DMA).  Here a workgroup fetches its 96-row panel of A (KC x NS planes, 147 KB
// for K = 256) ONCE by LDS-DMA, waits once, and then walks the N/64 column tiles with nothing but LDS fragment reads,
// MFMAs and W fragments from L2 (24 KB per 576 MFMA cycles; three chunks ahead in a ring of four register sets — one
// workgroup per CU, one wave per SIMD, i.e. the whole 512-register file per wave).  2 x 2 waves: 48 rows x 32 columns
// each.  The panel load is exposed (~20 % of the panel's MFMA time); everything after it is not.
constexpr int GP_ROWS = 96;
// Below: the tiled kernel.  Round 6, with the loads interleaved between the MFMAs (G9_STEP), 160 CUs, bf16x3, panel | tiled:
// M = 192 000: 558 | -, 115 200: 359 | 412, 96 000: 311 | 351, 57 600: 190 | 199, 48 640: 172 | 174, 29 184: 98 | 112 us
// (profiles/r06_b_gemm_panel96.txt; cutting short launches in two along N — twice the workgroups, a shorter tail round, the
// A panel fetched twice — was slower everywhere: the panel load is the exposed part)
constexpr int64_t GP_PANEL_MIN_M = 16 * 1024;

template <int NS, int KC>
__global__ void __launch_bounds__(GB_THREADS, 1)
gemm_bf_panel96_kernel(const GemmBfParams p) {
  constexpr int CH_U4 = NS * GP_ROWS * 4;                  // uint4 per k-chunk of the panel
  typedef Split<NS> SP;
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  uint4* const sA = reinterpret_cast<uint4*>(dsmem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, kg = lane >> 4;
  const int NT = p.N / 16, NJ = p.N / GB_BN;
  const int m0 = blockIdx.x * GP_ROWS;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  {
    // (k-chunk, plane, 16-row group) units of 1 KiB, dealt round-robin to the four waves
    const int rrow = lane >> 2, rslot = lane & 3;
    for (int u = wave; u < KC * NS * (GP_ROWS / 16); u += 4) {
      const int rg = u % (GP_ROWS / 16), pl = (u / (GP_ROWS / 16)) % NS, kc = u / ((GP_ROWS / 16) * NS);
      const int r = rg * 16 + rrow, m = m0 + r;
      const unsigned short* src = p.A + (size_t)pl * p.a_plane + (size_t)(m < p.M ? m : 0) * p.lda + kc * 32 + swz_slot(r, rslot) * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + kc * CH_U4 + (pl * GP_ROWS + rg * 16) * 4), 16, 0, 0);
    }
  }
  const uint4* b_src = p.wp + (size_t)(wn * 2) * 64 + lane;
  const size_t b_plane = (size_t)p.KC * NT * 64, b_chunk = (size_t)NT * 64;
  int a_frag[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int r = wm * 48 + a * 16 + i;
    a_frag[a] = r * 4 + swz_slot(r, kg);
  }
  const int col4 = (lane & 7) * 4;
  float* const sC = reinterpret_cast<float*>(dsmem + (size_t)KC * CH_U4 * 16) + wave * (16 * 36);

  // (Round 6, measured and not kept: the W fragments through a buffer descriptor — base + constant VGPR offset + uniform SGPR
  // offset, no per-lane 64-bit address per load: VALU instructions per 288 MFMAs 106 -> 34, and the launches got 1.5 % SLOWER,
  // 521 / 281 / 154 -> 529 / 288 / 156 us: with one wave per SIMD those additions sat in the MFMAs' shadow already.)
  uint4 wr[4][NS][2];                  // ring of W fragment sets: chunk c of a tile lives in set c % 4 (KC % 4 == 0 or
                                       // the ring position is carried: see SETOF)
  // flattened (tile, chunk) sequence: step s = j * KC + kc; its set is s % 4
#define G9_FETCH(s_, set_)                                                                              \
  {                                                                                                     \
    const int ss = min((s_), NJ * KC - 1);                                                              \
    const int jj = ss / KC, kk = ss - jj * KC;                                                          \
    _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) _Pragma("unroll") for (int b = 0; b < 2; ++b)     \
      wr[set_][pl][b] = b_src[(size_t)pl * b_plane + (size_t)kk * b_chunk + (size_t)(jj * 4 + b) * 64]; \
  }
  G9_FETCH(0, 0) G9_FETCH(1, 1) G9_FETCH(2, 2)
  __syncthreads();                     // the panel has landed (every wave's DMA)

  f32x4 accs[SP::NACC][3][2];
  float4 bias_v = make_float4(0.f, 0.f, 0.f, 0.f);
  // one (tile, chunk) step with ring set SET (literal): W fragments of step s + 3 into set (SET + 3) % 4 and the A
  // fragments of step s + 1 (LDS) are requested first, then the MFMAs of step s run on what was requested earlier —
  // with one wave per SIMD nothing else would hide the LDS latency
#define G9_READ_A(s_, dst)                                                                              \
  {                                                                                                     \
    const int kc_ = (s_) % KC;                                                                          \
    _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) _Pragma("unroll") for (int a = 0; a < 3; ++a)     \
      dst[pl][a] = sA[kc_ * CH_U4 + pl * GP_ROWS * 4 + a_frag[a]];                                      \
  }
#ifndef SLU_G9_SCHED
#define SLU_G9_SCHED 1
#endif
#define G9_STEP(s_, SET, fcur, fnxt)                                                                    \
  {                                                                                                     \
    G9_FETCH((s_) + 3, (SET + 3) % 4)                                                                   \
    G9_READ_A((s_) + 1, fnxt)                                                                           \
    if constexpr (SLU_G9_SCHED == 0) __builtin_amdgcn_sched_barrier(0);                                 \
    _Pragma("unroll") for (int q = 0; q < SP::NPAIR; ++q) {                                             \
      _Pragma("unroll") for (int a = 0; a < 3; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)       \
        accs[SP::ACC(q)][a][b] = mfma_split<NS>(fcur[SP::PA(q)][a], wr[SET][SP::PB(q)][b], accs[SP::ACC(q)][a][b]); \
    }                                                                                                   \
    if constexpr (SLU_G9_SCHED == 0) {                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                \
    } else {                                                                                            \
      /* one wave per SIMD issues in order: a load placed BETWEEN two MFMAs costs nothing (the wave would wait for the    \
         matrix pipe anyway), a block of loads in front of 36 back-to-back MFMAs costs its whole issue time (round 6:   \
         MFMA-busy 51 % of the wave cycles with the block form).  Pattern: MFMA + W load (x 2 NS), MFMA + A fragment     \
         read (x 3 NS), the remaining MFMAs */                                                                          \
      _Pragma("unroll") for (int g_ = 0; g_ < 2 * NS; ++g_) {                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                              \
      }                                                                                                 \
      _Pragma("unroll") for (int g_ = 0; g_ < 3 * NS; ++g_) {                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                              \
      }                                                                                                 \
      __builtin_amdgcn_sched_group_barrier(0x008, SP::NPAIR * 6 - 5 * NS, 0);                           \
    }                                                                                                   \
  }
  uint4 fa0[NS][3], fa1[NS][3];
  G9_READ_A(0, fa0)
  static_assert(KC % 4 == 0, "the register ring assumes a multiple of four k-chunks per tile");
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int c = 0; c < SP::NACC; ++c)
#pragma unroll
      for (int a = 0; a < 3; ++a) { accs[c][a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accs[c][a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (p.bias) bias_v = *reinterpret_cast<const float4*>(p.bias + j * GB_BN + wn * 32 + col4);
#pragma unroll
    for (int k4 = 0; k4 < KC; k4 += 4) {
      G9_STEP(j * KC + k4 + 0, 0, fa0, fa1)
      G9_STEP(j * KC + k4 + 1, 1, fa1, fa0)
      G9_STEP(j * KC + k4 + 2, 2, fa0, fa1)
      G9_STEP(j * KC + k4 + 3, 3, fa1, fa0)
    }
    // epilogue of the tile: three passes of 16 rows per wave through its private LDS strip, row-contiguous stores
    f32x4 acc[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = split_result<NS>(accs[0][a][b], accs[SP::NACC - 1][a][b]);
    const int ncol = j * GB_BN + wn * 32 + col4;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(4 * kg + r) * 36 + b * 16 + i] = acc[a][b][r];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int rl = (lane >> 3) + 8 * it;
        const int m = m0 + wm * 48 + a * 16 + rl;
        const float4 v = *reinterpret_cast<const float4*>(&sC[rl * 36 + col4]);
        if (m < p.M)
          *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + ncol) =
              make_float4(v.x + bias_v.x, v.y + bias_v.y, v.z + bias_v.z, v.w + bias_v.w);
      }
    }
  }
#undef G9_STEP
#undef G9_READ_A
#undef G9_FETCH
}


// ---- fp32 A, split on the fly: the GEMMs of TRAINABLE layers (forward projection x W_ih^T, data gradient d_gx W_ih) --------
// Same tile and MFMA schedule as gemm_bf_kernel, but A arrives as plain fp32 (M x K, k fast, row stride lda) — an
// activation or a gradient produced a moment ago by an exact-fp32 kernel — and is split into the scheme's planes in
// registers on its way into LDS (one split per element and column tile; a separate split pass would write and re-read
// 2 NS bytes per element through HBM).  The weights come packed (gemm_bf_pack_kernel reads them with any strides, i.e.
// W or W^T in place).  Chunk kc+2's fp32 rows are in flight (16 registers per thread) during chunk kc's MFMAs and are
// split into the idle LDS buffer after them; N % 4 == 0 (stores are guarded), K % 4 == 0.
template <int NS>
__global__ void __launch_bounds__(GB_THREADS, 2)
gemm_bf_a32_kernel(const float* __restrict__ A32, long long lda32, int K, const GemmBfParams p) {
  constexpr int STAGE_U4 = NS * GB_BM * 4;
  constexpr int EPI_FLOATS = 4 * 32 * 36;
  constexpr int SMEM_BYTES = (2 * STAGE_U4 * 16 > EPI_FLOATS * 4) ? 2 * STAGE_U4 * 16 : EPI_FLOATS * 4;
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  uint4* const sA = reinterpret_cast<uint4*>(smem);
  typedef Split<NS> SP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, kg = lane >> 4;
  const int NT = (p.N + 15) / 16;
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
  const int m0 = by * GB_BM, n0 = bx * GB_BN;

  f32x4 accs[SP::NACC][4][2];
#pragma unroll
  for (int c = 0; c < SP::NACC; ++c)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) accs[c][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: thread -> row tid / 2, k half tid % 2 of the 128 x 32 chunk: four float4 loads (64 contiguous bytes),
  // two 16-byte slots per plane.  Rows past M read row 0 (never stored); k past K reads k = 0 and is zeroed.
  const int srow = tid >> 1, shalf = tid & 1;
  const float* __restrict__ arow = A32 + (size_t)(m0 + srow < p.M ? m0 + srow : 0) * lda32;
  const int s_slot0 = srow * 4 + swz_slot(srow, 2 * shalf), s_slot1 = srow * 4 + swz_slot(srow, 2 * shalf + 1);
  float4 pre[4];
#define GA_LOAD(kc_)                                                                         \
  _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                            \
    const int k = (kc_) * 32 + shalf * 16 + v * 4;                                           \
    pre[v] = *reinterpret_cast<const float4*>(arow + (k < K ? k : 0));   /* zeroed at the split */ \
  }
#define GA_STORE(buf_, kc_)                                                                  \
  {                                                                                          \
    unsigned short h[16][NS];                                                                \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                          \
      if ((kc_) * 32 + shalf * 16 + v * 4 >= K) pre[v] = make_float4(0.f, 0.f, 0.f, 0.f);    \
      split_terms<NS>(pre[v].x, h[4 * v]); split_terms<NS>(pre[v].y, h[4 * v + 1]);          \
      split_terms<NS>(pre[v].z, h[4 * v + 2]); split_terms<NS>(pre[v].w, h[4 * v + 3]);      \
    }                                                                                        \
    _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) {                                      \
      uint4 o0, o1;                                                                          \
      o0.x = h[0][pl] | ((unsigned)h[1][pl] << 16);   o0.y = h[2][pl] | ((unsigned)h[3][pl] << 16);   \
      o0.z = h[4][pl] | ((unsigned)h[5][pl] << 16);   o0.w = h[6][pl] | ((unsigned)h[7][pl] << 16);   \
      o1.x = h[8][pl] | ((unsigned)h[9][pl] << 16);   o1.y = h[10][pl] | ((unsigned)h[11][pl] << 16); \
      o1.z = h[12][pl] | ((unsigned)h[13][pl] << 16); o1.w = h[14][pl] | ((unsigned)h[15][pl] << 16); \
      sA[(buf_) * STAGE_U4 + pl * GB_BM * 4 + s_slot0] = o0;                                 \
      sA[(buf_) * STAGE_U4 + pl * GB_BM * 4 + s_slot1] = o1;                                 \
    }                                                                                        \
  }
  // the last column tile of the grid may reach past the NT packed 16-column tiles (N % 64 != 0): those fragment
  // reads are clamped to the last tile (their columns are never stored)
  const uint4* b_src = p.wp + lane;
  const int b_tile[2] = {min(n0 / 16 + wn * 2, NT - 1) * 64, min(n0 / 16 + wn * 2 + 1, NT - 1) * 64};
  const size_t b_plane = (size_t)p.KC * NT * 64, b_chunk = (size_t)NT * 64;
  int a_frag[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = wm * 64 + a * 16 + i;
    a_frag[a] = r * 4 + swz_slot(r, kg);
  }
  uint4 rb0[NS][2], rb1[NS][2];
#define GA_FETCH_B(kc_, dst)                                                                         \
  _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) _Pragma("unroll") for (int b = 0; b < 2; ++b)    \
    dst[pl][b] = b_src[(size_t)pl * b_plane + (size_t)(kc_) * b_chunk + b_tile[b]];
#define GA_STEP(kc_, buf, cur, nxt)   /* buf: literal 0 / 1 = (kc_) & 1 */                           \
  {                                                                                                  \
    const int kn = min((kc_) + 1, p.KC - 1);                                                         \
    uint4 fa[NS][4];                                                                                 \
    _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) _Pragma("unroll") for (int a = 0; a < 4; ++a)  \
      fa[pl][a] = sA[buf * STAGE_U4 + pl * GB_BM * 4 + a_frag[a]];                                   \
    GA_FETCH_B(kn, nxt)                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    _Pragma("unroll") for (int q = 0; q < SP::NPAIR; ++q) {                                          \
      _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)    \
        accs[SP::ACC(q)][a][b] = mfma_split<NS>(fa[SP::PA(q)][a], cur[SP::PB(q)][b], accs[SP::ACC(q)][a][b]); \
    }                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    if ((kc_) + 1 < p.KC) {          /* `pre` holds chunk kc + 1: into the idle buffer, then fetch kc + 2 */ \
      GA_STORE(1 - buf, (kc_) + 1)                                                                   \
      if ((kc_) + 2 < p.KC) { GA_LOAD((kc_) + 2) }                                                   \
    }                                                                                                \
    __syncthreads();                                                                                 \
  }
  GA_LOAD(0)
  GA_FETCH_B(0, rb0)
  GA_STORE(0, 0)
  if (p.KC > 1) { GA_LOAD(1) }
  __syncthreads();
  for (int kc = 0; kc < p.KC; kc += 2) {
    GA_STEP(kc, 0, rb0, rb1)
    if (kc + 1 < p.KC) GA_STEP(kc + 1, 1, rb1, rb0)
  }
#undef GA_STEP
#undef GA_FETCH_B
#undef GA_STORE
#undef GA_LOAD
  f32x4 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = split_result<NS>(accs[0][a][b], accs[SP::NACC - 1][a][b]);

  float* const sC = reinterpret_cast<float*>(smem) + wave * (32 * 36);
  const int col4 = (lane & 7) * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  const int ncol = n0 + wn * 32 + col4;
  const bool col_ok = ncol < p.N;                      // N % 4 == 0: a float4 is inside or outside as a whole
  if (p.bias && col_ok) { bv.x = p.bias[ncol]; bv.y = p.bias[ncol + 1]; bv.z = p.bias[ncol + 2]; bv.w = p.bias[ncol + 3]; }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(a2 * 16 + 4 * kg + r) * 36 + b * 16 + i] = acc[2 * h + a2][b][r];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = (lane >> 3) + 8 * it;
      const int m = m0 + wm * 64 + h * 32 + rl;
      const float4 v = *reinterpret_cast<const float4*>(&sC[rl * 36 + col4]);
      if (m < p.M && col_ok)
        *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + ncol) = make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w);
    }
  }
}

}  // namespace slu

using namespace slu;

extern "C" size_t slu_gemm_bf16_pack_bytes(int64_t N, int64_t K, int nsplit) {
  return (size_t)nsplit * cdiv(K, 32) * cdiv(N, 16) * 64 * sizeof(uint4);
}

extern "C" int slu_gemm_bf16_pack(const float* W, int64_t ldw, int64_t w_cs, void* packed, int64_t N, int64_t K, int nsplit,
                                  void* stream) {
  SLU_REQUIRE(W && packed && N > 0 && K > 0, "slu_gemm_bf16_pack: bad argument");
  SLU_REQUIRE(nsplit >= 1 && nsplit <= 3, "slu_gemm_bf16_pack: nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  const int KC = (int)cdiv(K, 32), NT = (int)cdiv(N, 16);
  const int total = KC * NT * 64;
#define SLU_PACK(NS_) hipLaunchKernelGGL(gemm_bf_pack_kernel<NS_>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, \
                                         (long long)ldw, (long long)w_cs, (uint4*)packed, (int)N, (int)K, KC, NT)
  if (nsplit == 3) SLU_PACK(3); else if (nsplit == 2) SLU_PACK(2); else SLU_PACK(1);
#undef SLU_PACK
  SLU_CHECK_LAUNCH("gemm_bf_pack_kernel");
  return SLU_OK;
}

extern "C" int slu_split_bf16(const float* x, int64_t ldx, void* planes, int64_t plane_stride, int64_t rows,
                              int64_t K, int nsplit, void* stream) {
  SLU_REQUIRE(x && planes && rows > 0 && K > 0, "slu_split_bf16: bad argument");
  SLU_REQUIRE(nsplit >= 1 && nsplit <= 3, "slu_split_bf16: nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  const int Kp = (int)(cdiv(K, 32) * 32);
  SLU_REQUIRE(plane_stride >= rows * Kp, "slu_split_bf16: plane stride smaller than rows * round_up(K, 32)");
  const long long total = rows * (Kp / 8);
#define SLU_SPLIT(NS_) hipLaunchKernelGGL(split_planes_kernel<NS_>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, \
                                          (long long)ldx, (unsigned short*)planes, (long long)plane_stride, (long long)rows, (int)K, Kp)
  if (nsplit == 3) SLU_SPLIT(3); else if (nsplit == 2) SLU_SPLIT(2); else SLU_SPLIT(1);
#undef SLU_SPLIT
  SLU_CHECK_LAUNCH("split_planes_kernel");
  return SLU_OK;
}

extern "C" int slu_gemm_bf16(const void* A_planes, int64_t a_plane_stride, int64_t lda, const void* w_packed,
                             const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                             int nsplit, void* stream) {
  SLU_REQUIRE(A_planes && w_packed && C, "slu_gemm_bf16: null pointer");
  SLU_REQUIRE(M > 0 && N > 0 && K > 0, "slu_gemm_bf16: non-positive size");
  SLU_REQUIRE(nsplit >= 1 && nsplit <= 3, "slu_gemm_bf16: nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  if (N % GB_BN != 0) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gemm_bf16: N = %lld is not a multiple of %d", (long long)N, GB_BN);
  const int64_t Kp = cdiv(K, 32) * 32;
  SLU_REQUIRE(lda >= Kp && (lda % 8) == 0, "slu_gemm_bf16: lda must be >= round_up(K, 32) and a multiple of 8");
  SLU_REQUIRE((ldc % 4) == 0, "slu_gemm_bf16: ldc must be a multiple of 4");
  GemmBfParams p;
  p.A = (const unsigned short*)A_planes; p.a_plane = a_plane_stride; p.lda = lda;
  p.wp = (const uint4*)w_packed; p.bias = bias; p.C = C; p.ldc = ldc;
  p.M = (int)M; p.N = (int)N; p.KC = (int)(Kp / 32);
  if (p.KC <= 2 && N >= 2 * GB_BN && (bias == nullptr || ((uintptr_t)bias & 15) == 0)) {
    // row-panel kernel: A resident in LDS, column tiles walked by the workgroup (HBM-write-bound shapes)
    const dim3 pg((unsigned)cdiv(M, GB_BM));
    hipStream_t st = (hipStream_t)stream;
#define SLU_PANEL(NS_)                                                                                    \
    { if (p.KC == 2) hipLaunchKernelGGL((gemm_bf_panel_kernel<NS_, 2>), pg, dim3(GB_THREADS), 0, st, p); \
      else hipLaunchKernelGGL((gemm_bf_panel_kernel<NS_, 1>), pg, dim3(GB_THREADS), 0, st, p); }
    if (nsplit == 3) SLU_PANEL(3) else if (nsplit == 2) SLU_PANEL(2) else SLU_PANEL(1)
#undef SLU_PANEL
    SLU_CHECK_LAUNCH("gemm_bf_panel_kernel");
    return SLU_OK;
  }
  // panels pay their A load once per panel and leave a tail of idle CUs in the last round (GP_PANEL_MIN_M has the figures)
  if ((p.KC == 4 || p.KC == 8) && N >= 2 * GB_BN && M >= GP_PANEL_MIN_M && (bias == nullptr || ((uintptr_t)bias & 15) == 0)
      && !(getenv("SLU_GEMM_PANEL96") && atoi(getenv("SLU_GEMM_PANEL96")) == 0)) {     // SLU_GEMM_PANEL96=0: tiled kernel
    const size_t lds = (size_t)p.KC * nsplit * GP_ROWS * 64 + 4 * 16 * 36 * sizeof(float);
    const dim3 pg((unsigned)cdiv(M, GP_ROWS));
    hipStream_t st = (hipStream_t)stream;
#define SLU_P96(NS_, KC_)                                                                                            \
    {                                                                                                                \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_bf_panel96_kernel<NS_, KC_>,                              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
      if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "slu_gemm_bf16: cannot raise the dynamic LDS cap to %zu: %s", lds, hipGetErrorString(e)); \
      hipLaunchKernelGGL((gemm_bf_panel96_kernel<NS_, KC_>), pg, dim3(GB_THREADS), lds, st, p);                      \
    }
    if (nsplit == 3 && p.KC == 8) SLU_P96(3, 8)
    else if (nsplit == 3) SLU_P96(3, 4)
    else if (nsplit == 2 && p.KC == 8) SLU_P96(2, 8)
    else if (nsplit == 2) SLU_P96(2, 4)
    else if (p.KC == 8) SLU_P96(1, 8)
    else SLU_P96(1, 4)
#undef SLU_P96
    SLU_CHECK_LAUNCH("gemm_bf_panel96_kernel");
    return SLU_OK;
  }
  dim3 grid((unsigned)(N / GB_BN), (unsigned)cdiv(M, GB_BM));
  SLU_REQUIRE(grid.y <= 65535, "slu_gemm_bf16: M too large for one launch");
  if (nsplit == 3) hipLaunchKernelGGL(gemm_bf_kernel<3>, grid, dim3(GB_THREADS), 0, (hipStream_t)stream, p);
  else if (nsplit == 2) hipLaunchKernelGGL(gemm_bf_kernel<2>, grid, dim3(GB_THREADS), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gemm_bf_kernel<1>, grid, dim3(GB_THREADS), 0, (hipStream_t)stream, p);
  SLU_CHECK_LAUNCH("gemm_bf_kernel");
  return SLU_OK;
}

extern "C" int slu_gemm_bf16_a32(const float* A, int64_t lda, const void* w_packed, const float* bias, float* C, int64_t ldc,
                                 int64_t M, int64_t N, int64_t K, int nsplit, void* stream) {
  SLU_REQUIRE(A && w_packed && C, "slu_gemm_bf16_a32: null pointer");
  SLU_REQUIRE(M > 0 && N > 0 && K > 0, "slu_gemm_bf16_a32: non-positive size");
  SLU_REQUIRE(nsplit >= 1 && nsplit <= 3, "slu_gemm_bf16_a32: nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  if ((N & 3) || (K & 3) || (lda & 3) || (ldc & 3) || ((uintptr_t)A & 15) || ((uintptr_t)C & 15) || lda < K)
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gemm_bf16_a32: needs N, K, lda, ldc multiples of 4 and 16-byte aligned A, C "
             "(N %lld, K %lld, lda %lld, ldc %lld)", (long long)N, (long long)K, (long long)lda, (long long)ldc);
  GemmBfParams p;
  p.A = nullptr; p.a_plane = 0; p.lda = 0;
  p.wp = (const uint4*)w_packed; p.bias = bias; p.C = C; p.ldc = ldc;
  p.M = (int)M; p.N = (int)N; p.KC = (int)cdiv(K, 32);
  dim3 grid((unsigned)cdiv(N, GB_BN), (unsigned)cdiv(M, GB_BM));
  SLU_REQUIRE(grid.y <= 65535, "slu_gemm_bf16_a32: M too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  if (nsplit == 3) hipLaunchKernelGGL(gemm_bf_a32_kernel<3>, grid, dim3(GB_THREADS), 0, st, A, (long long)lda, (int)K, p);
  else if (nsplit == 2) hipLaunchKernelGGL(gemm_bf_a32_kernel<2>, grid, dim3(GB_THREADS), 0, st, A, (long long)lda, (int)K, p);
  else hipLaunchKernelGGL(gemm_bf_a32_kernel<1>, grid, dim3(GB_THREADS), 0, st, A, (long long)lda, (int)K, p);
  SLU_CHECK_LAUNCH("gemm_bf_a32_kernel");
  return SLU_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// TN GEMM on split-precision operands for the weight gradients (NS = 1: bf16, BASELINE configs[4] / SLU_DTYPE=bf16;
// NS = 2: f16x2, fp32-class, the default of the trainable layers' weight gradients; NS = 3: bf16x3):
//     C (M x N) = A^T B,   A (K x M), B (K x N) fp32 row-major (k is the SLOW index of both: d_gx / d_gh and x / h_prev
//     as the BPTT and the forward pass leave them), K = T * B rows up to tens of thousands.
// The bf16 MFMA wants 8 consecutive k per lane, so a chunk of 32 k-rows is read with coalesced float4 loads (a row of 64
// m / n values is 256 contiguous bytes), rounded to bf16 in registers and written to LDS TRANSPOSED ([m][k], 80-byte rows):
// a fragment is then one ds_read_b128.  64 x 64 output tile per workgroup, 2 x 2 waves of 32 x 32, fp32 accumulation;
// the next chunk's global loads are in flight during the MFMAs (register prefetch, LDS double buffer, one barrier per
// chunk).  Split-K over gridDim.z into a workspace [KS][M][N], summed in fixed order by gemm_tn_bf16_reduce_kernel
// (deterministic).  Operand traffic comes from L2 / MALL (each operand is re-read by the other dimension's tiles).
// ---------------------------------------------------------------------------------------------------------------------
namespace slu {

constexpr int TNB_T = 64;          // tile edge (M and N)
constexpr int TNB_K = 32;          // k rows per chunk
constexpr int TNB_LD = 40;         // bf16 elements per LDS row: 32 + 8 padding (80 bytes, 16-byte aligned fragments)

struct TnBfParams {
  const float* A; long long lda;
  const float* B; long long ldb;
  float* out; long long ldc;       // C (KS == 1) or the workspace slab base (row stride N)
  long long slab;                  // elements between split slabs (M * N), 0 when KS == 1
  int M, N, K, k_per_split;
};

// rows c4 .. c4 + 3 of the transposed tile, k columns kp and kp + 1 (one 4-byte store per row and plane)
template <int NS>
__device__ __forceinline__ void tnb_stage(unsigned short* __restrict__ s, const float4& v0, const float4& v1, int c4, int kp) {
  const float e0[4] = {v0.x, v0.y, v0.z, v0.w}, e1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned short h0[NS], h1[NS];
    split_terms<NS>(e0[j], h0);
    split_terms<NS>(e1[j], h1);
#pragma unroll
    for (int pl = 0; pl < NS; ++pl)
      *reinterpret_cast<unsigned*>(s + pl * (TNB_T * TNB_LD) + (c4 + j) * TNB_LD + kp) = h0[pl] | ((unsigned)h1[pl] << 16);
  }
}

template <int NS>
__global__ void __launch_bounds__(256)
gemm_tn_bf16_kernel(const TnBfParams p) {
  typedef Split<NS> SP;
  __shared__ __attribute__((aligned(16))) unsigned short sA[2][NS * TNB_T * TNB_LD];
  __shared__ __attribute__((aligned(16))) unsigned short sB[2][NS * TNB_T * TNB_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m0 = blockIdx.y * TNB_T, n0 = blockIdx.x * TNB_T;
  const int k_begin = blockIdx.z * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int kp = (tid >> 4) * 2, c4 = (tid & 15) * 4;            // loader: k rows kp, kp + 1; columns c4 .. c4 + 3
  // M, N multiples of 4: a float4 of columns is inside or outside the matrix as a whole (outside: zeros, never stored)
  const bool a_in = m0 + c4 < p.M, b_in = n0 + c4 < p.N;
  const float* __restrict__ Ap = p.A + (a_in ? m0 + c4 : 0);
  const float* __restrict__ Bp = p.B + (b_in ? n0 + c4 : 0);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);

  auto load = [&](int k0, float4& a0, float4& a1, float4& b0, float4& b1) {
    const int ka = k0 + kp, kb = k0 + kp + 1;
    a0 = (a_in && ka < k_end) ? *reinterpret_cast<const float4*>(Ap + (long long)ka * p.lda) : zero;
    a1 = (a_in && kb < k_end) ? *reinterpret_cast<const float4*>(Ap + (long long)kb * p.lda) : zero;
    b0 = (b_in && ka < k_end) ? *reinterpret_cast<const float4*>(Bp + (long long)ka * p.ldb) : zero;
    b1 = (b_in && kb < k_end) ? *reinterpret_cast<const float4*>(Bp + (long long)kb * p.ldb) : zero;
  };

  float4 a0, a1, b0, b1;
  load(k_begin, a0, a1, b0, b1);
  tnb_stage<NS>(sA[0], a0, a1, c4, kp);
  tnb_stage<NS>(sB[0], b0, b1, c4, kp);
  __syncthreads();

  const int wm = w >> 1, wn = w & 1;
  const int i = lane & 15, kg = lane >> 4;
  f32x4 accs[SP::NACC][2][2];
#pragma unroll
  for (int c = 0; c < SP::NACC; ++c)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) accs[c][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int buf = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += TNB_K) {
    const bool more = k0 + TNB_K < k_end;
    if (more) load(k0 + TNB_K, a0, a1, b0, b1);                 // in flight during this chunk's MFMAs
    uint4 fa[NS][2], fb[NS][2];
#pragma unroll
    for (int pl = 0; pl < NS; ++pl) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        fa[pl][mt] = *reinterpret_cast<const uint4*>(&sA[buf][pl * (TNB_T * TNB_LD) + (wm * 32 + mt * 16 + i) * TNB_LD + kg * 8]);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        fb[pl][nt] = *reinterpret_cast<const uint4*>(&sB[buf][pl * (TNB_T * TNB_LD) + (wn * 32 + nt * 16 + i) * TNB_LD + kg * 8]);
    }
#pragma unroll
    for (int q = 0; q < SP::NPAIR; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          accs[SP::ACC(q)][mt][nt] = mfma_split<NS>(fa[SP::PA(q)][mt], fb[SP::PB(q)][nt], accs[SP::ACC(q)][mt][nt]);
    if (more) {
      tnb_stage<NS>(sA[buf ^ 1], a0, a1, c4, kp);
      tnb_stage<NS>(sB[buf ^ 1], b0, b1, c4, kp);
    }
    __syncthreads();
    buf ^= 1;
  }

  // D[row = 4 * kg + r][col = i] of each 16 x 16 tile: rows index m (the A operand), columns n
  float* __restrict__ out = p.out + (size_t)blockIdx.z * p.slab;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 acc = split_result<NS>(accs[0][mt][nt], accs[SP::NACC - 1][mt][nt]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + mt * 16 + 4 * kg + r, n = n0 + wn * 32 + nt * 16 + i;
        if (m < p.M && n < p.N) out[(size_t)m * p.ldc + n] = acc[r];
      }
    }
}

__global__ void __launch_bounds__(256)
gemm_tn_bf16_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, long long ldc, int M, int N, int KS) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx - (long long)m * N);
  float s = 0.0f;
  for (int k = 0; k < KS; ++k) s += ws[(size_t)k * M * N + idx];
  C[(long long)m * ldc + n] = s;
}

static void tnb_plan(int64_t M, int64_t N, int64_t K, int* KS, int* kper) {
  const int64_t tiles = cdiv(M, TNB_T) * cdiv(N, TNB_T);
  int64_t ks = 512 / tiles;                       // a function of the shape only (bit-reproducible on any stream)
  const int64_t max_ks = cdiv(K, 256);            // >= 256 k rows per split
  if (ks > max_ks) ks = max_ks;
  if (ks < 1) ks = 1;
  const int64_t per = cdiv(cdiv(K, ks), TNB_K) * TNB_K;
  *kper = (int)per;
  *KS = (int)cdiv(K, per);
}

}  // namespace slu

extern "C" size_t slu_gemm_tn_bf16_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0 || (M & 3) || (N & 3)) return 0;
  int KS, kper;
  slu::tnb_plan(M, N, K, &KS, &kper);
  return KS > 1 ? (size_t)KS * M * N * sizeof(float) : 0;
}

extern "C" int slu_gemm_tn_bf16(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                                int64_t M, int64_t N, int64_t K, int nsplit, void* workspace, size_t workspace_bytes,
                                void* stream) {
  SLU_REQUIRE(A && B && C, "slu_gemm_tn_bf16: null pointer");
  SLU_REQUIRE(nsplit >= 1 && nsplit <= 3, "slu_gemm_tn_bf16: nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  SLU_REQUIRE(M > 0 && N > 0 && K > 0 && K < (1LL << 31), "slu_gemm_tn_bf16: bad size");
  if ((M & 3) || (N & 3))
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gemm_tn_bf16: M = %lld and N = %lld must be multiples of 4", (long long)M, (long long)N);
  if ((lda | ldb) & 3 || ((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gemm_tn_bf16: lda / ldb must be multiples of 4 and A / B 16-byte aligned");
  int KS, kper;
  tnb_plan(M, N, K, &KS, &kper);
  TnBfParams p;
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.k_per_split = kper;
  if (KS > 1) {
    const size_t need = (size_t)KS * M * N * sizeof(float);
    if (!workspace || workspace_bytes < need)
      SLU_FAIL(SLU_ERR_WORKSPACE, "slu_gemm_tn_bf16: workspace too small (%zu < %zu)", workspace_bytes, need);
    p.out = reinterpret_cast<float*>(workspace); p.ldc = N; p.slab = M * N;
  } else {
    p.out = C; p.ldc = ldc; p.slab = 0;
  }
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(N, TNB_T), (unsigned)cdiv(M, TNB_T), (unsigned)KS);
  if (nsplit == 3) hipLaunchKernelGGL(gemm_tn_bf16_kernel<3>, grid, dim3(256), 0, st, p);
  else if (nsplit == 2) hipLaunchKernelGGL(gemm_tn_bf16_kernel<2>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(gemm_tn_bf16_kernel<1>, grid, dim3(256), 0, st, p);
  SLU_CHECK_LAUNCH("gemm_tn_bf16_kernel");
  if (KS > 1) {
    hipLaunchKernelGGL(gemm_tn_bf16_reduce_kernel, dim3((unsigned)cdiv(M * N, 256)), dim3(256), 0, st,
                       (const float*)workspace, C, (long long)ldc, (int)M, (int)N, KS);
    SLU_CHECK_LAUNCH("gemm_tn_bf16_reduce_kernel");
  }
  return SLU_OK;
}
