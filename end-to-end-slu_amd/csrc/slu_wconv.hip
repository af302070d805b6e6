// Windowed-GEMM convolution block for gfx950 (reference: models.py:108 SincLayer conv, :200 Conv1d,
// :163-168 Abs, :205 MaxPool1d(ceil_mode), :211 LeakyReLU).
//
// With channels-last activations a 1-D convolution over (B, L, Cin) is a correlation of the FLAT
// row in[b][0 .. L*Cin) with a window of Kw = Kt*Cin floats that advances S = stride*Cin floats per
// output frame — exactly the shape of the SincNet layer (Cin = 1, Kw = 401, S = 80).  One kernel
// serves all three CNN layers:
//     out[b][l][c] = sum_q  in_flat[b][l*S + q - pad] * W[c][q]
// as an fp32 MFMA GEMM (v_mfma_f32_16x16x4_f32, exact fp32): M = frames, N = channels, K = q.
//   * the input window of a workgroup's frames is staged ONCE into LDS (coalesced global reads,
//     zero fill outside the row) in rows of S floats with an odd row stride, so that the A-operand
//     read "frame i, tap q" -> lds[(i + q/S)*Sp + q%S] is bank-conflict free across the 16 frames;
//   * the filter matrix is pre-packed in B-fragment order (wconv_pack_kernel) so that every
//     B-operand load is one fully coalesced 256-byte read shared by all waves through L1/L2;
//   * bias, abs, max-pool (pairs of frames live in adjacent accumulator registers of one lane),
//     LeakyReLU and the layout change (channels-last or time-major) are fused into the epilogue.
#include "slu_common.h"

namespace slu {

constexpr int WC_THREADS = 256;   // 4 waves

struct WconvParams {
  const float* in;      // (B, in_row) flat rows
  const float* wp;      // packed filters [KK][NT][64]
  const float* bias;    // (c_out) or null
  float* out;
  uint8_t* route;       // (B, l_out, c_out) or null
  long long in_row;     // floats per batch row (l_in * c_in)
  long long out_sb, out_sl;
  int S, Sp, Kw, KK, pad;
  int l_conv, l_out, c_out;
  int do_abs, pool;
  float slope;
  int nrows;            // LDS rows staged per workgroup
};

// mode 0: forward filters   W(c, q)  = w[c][q % c_in][q / c_in]                (c < c_out)
// mode 1: data gradient     W'(c, q) = w[q % c_out][c][k_t - 1 - q / c_out]    (c < c_in)
__global__ void wconv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int c_out,
                                  int c_in, int k_t, int NT, int KK, int mode) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= KK * NT * 64) return;
  const int lane = idx & 63;
  const int nt = (idx >> 6) % NT;
  const int kk = (idx >> 6) / NT;
  const int c = nt * 16 + (lane & 15);
  const int q = 4 * kk + (lane >> 4);
  float v = 0.0f;
  if (mode == 0) {
    if (c < c_out && q < k_t * c_in) {
      const int k = q / c_in, ci = q - k * c_in;
      v = w[((size_t)c * c_in + ci) * k_t + k];
    }
  } else {
    if (c < c_in && q < k_t * c_out) {
      const int k = q / c_out, co = q - k * c_out;
      v = w[((size_t)co * c_in + c) * k_t + (k_t - 1 - k)];
    }
  }
  wp[idx] = v;
}

template <int MT, int NT>
__global__ void __launch_bounds__(WC_THREADS)
wconv_fwd_kernel(const WconvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = reinterpret_cast<float*>(smem);
  constexpr int F = 64 * MT;                      // frames per workgroup
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * F;
  const float* __restrict__ inb = p.in + (size_t)b * p.in_row;

  // ---- stage the window [l0*S - pad, ...) into LDS rows of S floats (row stride Sp, odd) ----
  {
    const long long u0 = (long long)l0 * p.S - p.pad;
    const int total = p.nrows * p.S;
    const float invS = 1.0f / (float)p.S;
    // eight independent global loads in flight per thread, then their LDS stores (a load -> store chain per
    // element would expose the full memory latency once per element)
    constexpr int U = 8;
    for (int base = 0; base < total; base += WC_THREADS * U) {
      float v[U];
      int off[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int idx = base + j * WC_THREADS + tid;
        int row = (int)((float)idx * invS);
        int col = idx - row * p.S;
        if (col < 0) { col += p.S; --row; }
        if (col >= p.S) { col -= p.S; ++row; }
        const long long u = u0 + idx;
        off[j] = idx < total ? row * p.Sp + col : -1;
        v[j] = (idx < total && u >= 0 && u < p.in_row) ? inb[u] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < U; ++j)
        if (off[j] >= 0) lds[off[j]] = v[j];
    }
  }
  __syncthreads();

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int i = lane & 15, kg = lane >> 4;
  int qd = kg / p.S, qm = kg - qd * p.S;
  int abase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) abase[m] = (wave * 16 * MT + m * 16 + i) * p.Sp;
  const float* __restrict__ wp = p.wp + lane;

  // operands of k-step kk+1 are fetched (LDS window, L1/L2-resident packed filters) before the MFMAs
  // of k-step kk are issued: the load latency hides under MT*NT*32 MFMA cycles instead of stalling
  float a_n[MT], b_n[NT];
  {
    const int off = qd * p.Sp + qm;
#pragma unroll
    for (int m = 0; m < MT; ++m) a_n[m] = lds[abase[m] + off];
#pragma unroll
    for (int n = 0; n < NT; ++n) b_n[n] = wp[(size_t)n * 64];
  }
  for (int kk = 0; kk < p.KK; ++kk) {
    float a[MT], bv[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = a_n[m];
#pragma unroll
    for (int n = 0; n < NT; ++n) bv[n] = b_n[n];
    if (kk + 1 < p.KK) {
      qm += 4;
      while (qm >= p.S) { qm -= p.S; ++qd; }
      const int off = qd * p.Sp + qm;
#pragma unroll
      for (int m = 0; m < MT; ++m) a_n[m] = lds[abase[m] + off];
#pragma unroll
      for (int n = 0; n < NT; ++n) b_n[n] = wp[((size_t)(kk + 1) * NT + n) * 64];
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = mfma16(a[m], bv[n], acc[m][n]);
  }

  // ---- epilogue: bias, abs, max-pool over frame pairs, LeakyReLU, strided store ----
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int fbase = l0 + wave * 16 * MT + m * 16 + 4 * kg;   // multiple of 4
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int c = n * 16 + i;
      if (c >= p.c_out) continue;
      const float bias = p.bias ? p.bias[c] : 0.0f;
      float v[4];
      bool neg[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = acc[m][n][r] + bias;
        neg[r] = t < 0.0f;
        v[r] = p.do_abs ? fabsf(t) : t;
      }
      if (p.pool == 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int f0 = fbase + 2 * h;
          if (f0 >= p.l_conv) continue;
          const bool has1 = (f0 + 1) < p.l_conv;                // ceil_mode: last window may be partial
          const bool pick1 = has1 && (v[2 * h + 1] > v[2 * h]);
          const float pooled = pick1 ? v[2 * h + 1] : v[2 * h];
          const bool sgn = pick1 ? neg[2 * h + 1] : neg[2 * h];
          const float y = pooled > 0.0f ? pooled : pooled * p.slope;
          const long long lo = f0 >> 1;
          p.out[(size_t)b * p.out_sb + lo * p.out_sl + c] = y;
          if (p.route) p.route[((size_t)b * p.l_out + lo) * p.c_out + c] = (uint8_t)((pick1 ? 1 : 0) | (sgn ? 2 : 0));
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = fbase + r;
          if (f >= p.l_conv) continue;
          const float y = v[r] > 0.0f ? v[r] : v[r] * p.slope;
          p.out[(size_t)b * p.out_sb + (long long)f * p.out_sl + c] = y;
          if (p.route) p.route[((size_t)b * p.l_out + f) * p.c_out + c] = (uint8_t)(neg[r] ? 2 : 0);
        }
      }
    }
  }
}

// d_conv (B, l_conv, c_out) contiguous from dy / y (strided, pooled resolution).
__global__ void __launch_bounds__(256)
wconv_bwd_act_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                     const uint8_t* __restrict__ route, float* __restrict__ d_conv, long long total,
                     int l_conv, int l_out, int c_out, int do_abs, int pool, float slope,
                     long long sb, long long sl) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % c_out);
  const long long bl = idx / c_out;
  const int lo = (int)(bl % l_out);
  const long long b = bl / l_out;
  const size_t src = (size_t)b * sb + (size_t)lo * sl + c;
  const float yv = y[src];
  float g = dy[src] * (yv > 0.0f ? 1.0f : slope);
  const uint8_t rt = route ? route[idx] : 0;
  if (do_abs && (rt & 2)) g = -g;
  if (pool == 2) {
    const int f0 = 2 * lo;
    const size_t d0 = ((size_t)b * l_conv + f0) * c_out + c;
    const bool pick1 = rt & 1;
    d_conv[d0] = pick1 ? 0.0f : g;
    if (f0 + 1 < l_conv) d_conv[d0 + c_out] = pick1 ? g : 0.0f;
  } else {
    d_conv[((size_t)b * l_conv + lo) * c_out + c] = g;
  }
}

// ---- weight gradient:  dW(c, q) = sum_{b,l} d_conv[b][l][c] * in_flat[b][l*S + q - pad] ----
// MFMA with M = channels, N = taps q, K = frames.  Workgroup (qg, ks): taps [qg*128, qg*128+128),
// K-split ks = a contiguous range of 64-frame chunks.  Both operands of a chunk are CONTIGUOUS pieces
// of global memory — the d_conv slab (64 frames x c_out) and the flat input window
// [l0*S + q0 - pad, + 64*S + 128) — so they are copied to LDS as they lie (float4 where aligned, zero
// outside the row), and tap q of frame f is simply window[f*S + q].  The next chunk's pieces are
// fetched into registers while the current chunk's MFMAs run (one-chunk prefetch: without it every
// chunk paid a full global-load round trip between two barriers, 0.04-0.13 of the MFMA peak).
// The bias gradient rides along as the virtual tap q = Kw whose input is the constant 1.
// Partial sums go to ws[ks][c_out][Kw1]; wconv_dw_reduce sums the splits in fixed order
// (deterministic) and scatters to the torch (c_out, c_in, k_t) layout (+ d_bias).
constexpr int DW_FC = 64;     // frames per chunk
constexpr int DW_QG = 128;    // taps per workgroup (4 waves x 2 N-tiles x 16)
constexpr int DW_NIN = 9;     // float4 of the input window per thread: 64*S + 128 <= 9216 floats (S <= 142)

struct WconvDwParams {
  const float* d_conv;  // (B, l_conv, c_out)
  const float* in;      // (B, in_row)
  float* ws;            // [KS][c_out][Kw1]
  long long in_row;
  int S, Kw, Kw1, pad, l_conv, c_out;
  int chunks_per_row, chunks_total, chunks_per_split;
  int win;              // floats of the input window per chunk (multiple of 4)
  int bias_tap;         // Kw when the bias gradient is wanted, else -1
};

// 4 consecutive floats base[idx .. idx+3] with zero outside [0, n)
__device__ __forceinline__ float4 load4_bounded(const float* __restrict__ base, long long idx, long long n) {
  const float* p = base + idx;
  if (idx >= 0 && idx + 3 < n && ((reinterpret_cast<uintptr_t>(p) & 15) == 0))
    return *reinterpret_cast<const float4*>(p);
  float4 v;
  v.x = (idx >= 0 && idx < n) ? p[0] : 0.0f;
  v.y = (idx + 1 >= 0 && idx + 1 < n) ? p[1] : 0.0f;
  v.z = (idx + 2 >= 0 && idx + 2 < n) ? p[2] : 0.0f;
  v.w = (idx + 3 >= 0 && idx + 3 < n) ? p[3] : 0.0f;
  return v;
}

template <int MTC>
__global__ void __launch_bounds__(WC_THREADS)
wconv_dw_kernel(const WconvDwParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_in = reinterpret_cast<float*>(smem);            // [win]
  float* lds_d = lds_in + p.win;                             // [DW_FC * c_out] (+ MTC*16 slack), as in memory
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q0 = blockIdx.x * DW_QG;
  const int ks = blockIdx.y;
  const int i = lane & 15, kg = lane >> 4;
  const int c_out = p.c_out;
  const int dsz = DW_FC * c_out;                             // floats of a d_conv slab

  f32x4 acc[MTC][2];
#pragma unroll
  for (int m = 0; m < MTC; ++m) { acc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // per-lane tap offsets inside the window for its two N-tiles; the bias tap reads the constant 1
  int boff[2];
  bool is_bias[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    boff[n] = wave * 32 + n * 16 + i;
    is_bias[n] = (q0 + boff[n]) == p.bias_tap;
  }

  const int c_first = ks * p.chunks_per_split;
  const int c_last = min(c_first + p.chunks_per_split, p.chunks_total);

  float4 rin[DW_NIN], rd[MTC];
  auto fetch = [&](int ch) {
    const int b = ch / p.chunks_per_row;
    const int l0 = (ch - b * p.chunks_per_row) * DW_FC;
    const float* __restrict__ inb = p.in + (size_t)b * p.in_row;
    const long long u0 = (long long)l0 * p.S + q0 - p.pad;
#pragma unroll
    for (int v = 0; v < DW_NIN; ++v) {
      const int idx = 4 * (tid + v * WC_THREADS);
      if (idx < p.win) rin[v] = load4_bounded(inb, u0 + idx, p.in_row);
    }
    // slab: frames l0 .. l0+63 of row b are contiguous; frames beyond the row end contribute zero
    const float* __restrict__ db = p.d_conv + ((size_t)b * p.l_conv + l0) * c_out;
    const long long nvalid = (long long)min(DW_FC, p.l_conv - l0) * c_out;
#pragma unroll
    for (int v = 0; v < MTC; ++v) {
      const int idx = 4 * (tid + v * WC_THREADS);
      if (idx < dsz) rd[v] = load4_bounded(db, idx, nvalid);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int v = 0; v < DW_NIN; ++v) {
      const int idx = 4 * (tid + v * WC_THREADS);
      if (idx < p.win) *reinterpret_cast<float4*>(lds_in + idx) = rin[v];
    }
#pragma unroll
    for (int v = 0; v < MTC; ++v) {
      const int idx = 4 * (tid + v * WC_THREADS);
      if (idx < dsz) *reinterpret_cast<float4*>(lds_d + idx) = rd[v];
    }
  };

  // slack behind the slab (rows c >= c_out of the last frame read it; those accumulator rows are dropped)
  for (int x = tid; x < MTC * 16; x += WC_THREADS) lds_d[dsz + x] = 0.0f;
  if (c_first < c_last) fetch(c_first);
  for (int ch = c_first; ch < c_last; ++ch) {
    __syncthreads();                     // the previous chunk's MFMAs are done with the LDS tiles
    stash();
    __syncthreads();
    if (ch + 1 < c_last) fetch(ch + 1);  // in flight during this chunk's MFMAs
#pragma unroll 4
    for (int kk = 0; kk < DW_FC / 4; ++kk) {
      const int fr = 4 * kk + kg;
      float a[MTC];
#pragma unroll
      for (int m = 0; m < MTC; ++m) a[m] = lds_d[fr * c_out + m * 16 + i];
      float b0 = lds_in[fr * p.S + boff[0]];
      float b1 = lds_in[fr * p.S + boff[1]];
      b0 = is_bias[0] ? 1.0f : b0;
      b1 = is_bias[1] ? 1.0f : b1;
#pragma unroll
      for (int m = 0; m < MTC; ++m) {
        acc[m][0] = mfma16(a[m], b0, acc[m][0]);
        acc[m][1] = mfma16(a[m], b1, acc[m][1]);
      }
    }
  }
  float* __restrict__ ws = p.ws + (size_t)ks * c_out * p.Kw1;
#pragma unroll
  for (int m = 0; m < MTC; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int q = q0 + wave * 32 + n * 16 + i;
      if (q >= p.Kw1) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = m * 16 + 4 * kg + r;
        if (c < c_out) ws[(size_t)c * p.Kw1 + q] = acc[m][n][r];
      }
    }
}

__global__ void __launch_bounds__(256)
wconv_dw_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ dbias,
                       int KS, int c_out, int c_in, int k_t, int Kw1) {
  const int Kw = c_in * k_t;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= c_out * Kw1) return;
  // KS is 128-170 for the model's layers: eight independent chains keep eight loads in flight per
  // thread (two chains left the kernel waiting on one L2 round trip per pair, 19-26 us a launch);
  // the order is fixed (chain j takes the splits k = j mod 8, then a fixed tree), so the result is
  // deterministic.
  const size_t plane = (size_t)c_out * Kw1;
  float part[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) part[j] = 0.0f;
  int k = 0;
  for (; k + 8 <= KS; k += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) part[j] += ws[(size_t)(k + j) * plane + idx];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (k + j < KS) part[j] += ws[(size_t)(k + j) * plane + idx];
  const float s = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
  const int c = idx / Kw1, q = idx - c * Kw1;
  if (q == Kw) {
    if (dbias) dbias[c] = s;
    return;
  }
  const int kt = q / c_in, ci = q - kt * c_in;
  dw[((size_t)c * c_in + ci) * k_t + kt] = s;
}

// ---------------------------------------------------------------------------------------------
static inline int nt_for(int64_t c) {
  const int need = (int)cdiv(c, 16);
  if (need <= 1) return 1;
  if (need <= 2) return 2;
  if (need <= 4) return 4;
  if (need <= 5) return 5;
  if (need <= 8) return 8;
  return -1;
}

template <int MT, int NT>
static int raise_lds_cap(size_t lds) {
  if (lds <= 64 * 1024) return SLU_OK;
  hipError_t e = hipFuncSetAttribute((const void*)wconv_fwd_kernel<MT, NT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "wconv: cannot raise the dynamic LDS cap to %zu: %s", lds, hipGetErrorString(e));
  return SLU_OK;
}

template <int MT>
static int launch_fwd_nt(int NT, dim3 grid, size_t lds, hipStream_t st, const WconvParams& p) {
  int rc = SLU_OK;
  switch (NT) {
    case 1: rc = raise_lds_cap<MT, 1>(lds); break;
    case 2: rc = raise_lds_cap<MT, 2>(lds); break;
    case 4: rc = raise_lds_cap<MT, 4>(lds); break;
    case 5: rc = raise_lds_cap<MT, 5>(lds); break;
    case 8: rc = raise_lds_cap<MT, 8>(lds); break;
    default: break;
  }
  if (rc) return rc;
  switch (NT) {
    case 1: hipLaunchKernelGGL((wconv_fwd_kernel<MT, 1>), grid, dim3(WC_THREADS), lds, st, p); break;
    case 2: hipLaunchKernelGGL((wconv_fwd_kernel<MT, 2>), grid, dim3(WC_THREADS), lds, st, p); break;
    case 4: hipLaunchKernelGGL((wconv_fwd_kernel<MT, 4>), grid, dim3(WC_THREADS), lds, st, p); break;
    case 5: hipLaunchKernelGGL((wconv_fwd_kernel<MT, 5>), grid, dim3(WC_THREADS), lds, st, p); break;
    case 8: hipLaunchKernelGGL((wconv_fwd_kernel<MT, 8>), grid, dim3(WC_THREADS), lds, st, p); break;
    default: SLU_FAIL(SLU_ERR_UNSUPPORTED, "wconv: unsupported channel-tile count %d", NT);
  }
  SLU_CHECK_LAUNCH("wconv_fwd_kernel");
  return SLU_OK;
}

// Generic launcher shared by forward and data-gradient.
//   n_out channels produced, n_in channels consumed, pad_t / stride_t in frames, mode = pack mode.
static int wconv_launch(const float* in, const float* weight, const float* bias, float* out,
                        uint8_t* route, int64_t B, int64_t l_in, int64_t c_in, int64_t c_out,
                        int64_t k_t, int64_t stride_t, int64_t pad_t, int mode, int do_abs, int pool,
                        float slope, int64_t out_sb, int64_t out_sl, void* workspace,
                        size_t workspace_bytes, hipStream_t st) {
  // mode 1 swaps the roles of c_in / c_out (weight stays in the forward (c_out, c_in, k_t) layout)
  const int64_t n_in = (mode == 0) ? c_in : c_out;
  const int64_t n_out = (mode == 0) ? c_out : c_in;
  const int NT = nt_for(n_out);
  if (NT < 0) SLU_FAIL(SLU_ERR_UNSUPPORTED, "wconv: at most 128 output channels are supported (got %lld)", (long long)n_out);
  const int64_t Kw = k_t * n_in;
  const int64_t KK = cdiv(Kw, 4);
  const size_t need = (size_t)KK * NT * 64 * sizeof(float);
  if (!workspace || workspace_bytes < need)
    SLU_FAIL(SLU_ERR_WORKSPACE, "wconv: workspace too small (%zu < %zu)", workspace_bytes, need);
  const int64_t l_conv = (l_in + 2 * pad_t - k_t) / stride_t + 1;
  SLU_REQUIRE(l_conv > 0, "wconv: input shorter than the filter");
  SLU_REQUIRE(B <= 65535, "wconv: B must be <= 65535");
  float* wp = reinterpret_cast<float*>(workspace);
  {
    const int total = (int)(KK * NT * 64);
    hipLaunchKernelGGL(wconv_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, weight, wp,
                       (int)c_out, (int)c_in, (int)k_t, NT, (int)KK, mode);
    SLU_CHECK_LAUNCH("wconv_pack_kernel");
  }
  WconvParams p;
  p.in = in; p.wp = wp; p.bias = bias; p.out = out; p.route = route;
  p.in_row = l_in * n_in;
  p.out_sb = out_sb; p.out_sl = out_sl;
  p.S = (int)(stride_t * n_in);
  p.Sp = p.S | 1;
  p.Kw = (int)Kw; p.KK = (int)KK; p.pad = (int)(pad_t * n_in);
  p.l_conv = (int)l_conv;
  p.l_out = (int)cdiv(l_conv, pool);
  p.c_out = (int)n_out;
  p.do_abs = do_abs; p.pool = pool; p.slope = slope;
  // Choose frames per workgroup: 128 when that still yields >= 256 workgroups, else 64.
  const int MT = (B * cdiv(l_conv, 128) >= 256) ? 2 : 1;
  const int F = 64 * MT;
  p.nrows = F + (int)cdiv(4 * KK + 3, p.S) + 1;
  const size_t lds = (size_t)p.nrows * p.Sp * sizeof(float);
  if (lds > 160 * 1024) SLU_FAIL(SLU_ERR_UNSUPPORTED, "wconv: window of %zu bytes exceeds the 160 KiB LDS", lds);
  dim3 grid((unsigned)cdiv(l_conv, F), (unsigned)B);
  if (MT == 2) return launch_fwd_nt<2>(NT, grid, lds, st, p);
  return launch_fwd_nt<1>(NT, grid, lds, st, p);
}

}  // namespace slu

using namespace slu;

extern "C" size_t slu_wconv_workspace_bytes(int64_t c_out, int64_t c_in, int64_t k_t) {
  // large enough for both the forward pack (N = c_out, Kw = k_t*c_in) and the data-gradient pack
  const int64_t cmax = c_out > c_in ? c_out : c_in;
  const int64_t nt = cdiv(cmax, 16) <= 8 ? 8 : cdiv(cmax, 16);
  const int64_t kw = k_t * cmax;
  return (size_t)(cdiv(kw, 4) * nt * 64) * sizeof(float);
}

extern "C" int slu_wconv_fwd(const float* in, const float* weight, const float* bias, float* out,
                             uint8_t* route, int64_t B, int64_t l_in, int64_t c_in, int64_t c_out,
                             int64_t k_t, int64_t stride_t, int do_abs, int pool, float slope,
                             int64_t out_sb, int64_t out_sl, void* workspace,
                             size_t workspace_bytes, void* stream) {
  SLU_REQUIRE(in && weight && out, "slu_wconv_fwd: null pointer");
  SLU_REQUIRE(B > 0 && l_in > 0 && c_in > 0 && c_out > 0 && k_t > 0 && stride_t > 0, "slu_wconv_fwd: non-positive size");
  SLU_REQUIRE(pool == 1 || pool == 2, "slu_wconv_fwd: pool must be 1 or 2 (got %d)", pool);
  return wconv_launch(in, weight, bias, out, route, B, l_in, c_in, c_out, k_t, stride_t, k_t / 2, 0,
                      do_abs, pool, slope, out_sb, out_sl, workspace, workspace_bytes,
                      (hipStream_t)stream);
}

extern "C" int slu_wconv_bwd_act(const float* dy, const float* y, const uint8_t* route,
                                 float* d_conv, int64_t B, int64_t l_conv, int64_t c_out,
                                 int do_abs, int pool, float slope, int64_t sb, int64_t sl,
                                 void* stream) {
  SLU_REQUIRE(dy && y && d_conv, "slu_wconv_bwd_act: null pointer");
  SLU_REQUIRE(pool == 1 || pool == 2, "slu_wconv_bwd_act: pool must be 1 or 2");
  SLU_REQUIRE(route || (pool == 1 && !do_abs), "slu_wconv_bwd_act: route is required when pool == 2 or do_abs");
  const int64_t l_out = cdiv(l_conv, pool);
  const long long total = (long long)B * l_out * c_out;
  hipLaunchKernelGGL(wconv_bwd_act_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, y, route, d_conv, total, (int)l_conv, (int)l_out,
                     (int)c_out, do_abs, pool, slope, (long long)sb, (long long)sl);
  SLU_CHECK_LAUNCH("wconv_bwd_act_kernel");
  return SLU_OK;
}

extern "C" int slu_wconv_bwd_data(const float* d_conv, const float* weight, float* d_in, int64_t B,
                                  int64_t l_in, int64_t c_in, int64_t c_out, int64_t k_t,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  SLU_REQUIRE(d_conv && weight && d_in, "slu_wconv_bwd_data: null pointer");
  const int64_t pad_t = k_t / 2;
  const int64_t l_conv = l_in + 2 * pad_t - k_t + 1;
  // transposed correlation: "input" = d_conv (l_conv frames of c_out), output = l_in frames of c_in
  return wconv_launch(d_conv, weight, nullptr, d_in, nullptr, B, l_conv, c_in, c_out, k_t, 1,
                      k_t - 1 - pad_t, 1, 0, 1, 1.0f, l_in * c_in, c_in, workspace, workspace_bytes,
                      (hipStream_t)stream);
}

static void dw_geometry(int64_t B, int64_t l_in, int64_t c_in, int64_t c_out, int64_t k_t,
                        int64_t stride_t, int64_t* l_conv, int* QGn, int* KS, int* cps, int* cpr) {
  *l_conv = (l_in + 2 * (k_t / 2) - k_t) / stride_t + 1;
  const int64_t Kw1 = k_t * c_in + 1;                 // + the bias tap
  *QGn = (int)cdiv(Kw1, DW_QG);
  *cpr = (int)cdiv(*l_conv, DW_FC);
  const int64_t chunks = B * *cpr;
  int64_t ks = 512 / *QGn;
  if (ks < 1) ks = 1;
  if (ks > chunks) ks = chunks;
  *cps = (int)cdiv(chunks, ks);
  *KS = (int)cdiv(chunks, *cps);
  (void)c_out;
}

extern "C" size_t slu_wconv_bwd_weight_workspace_bytes(int64_t B, int64_t l_in, int64_t c_in,
                                                       int64_t c_out, int64_t k_t, int64_t stride_t) {
  int64_t l_conv; int QGn, KS, cps, cpr;
  dw_geometry(B, l_in, c_in, c_out, k_t, stride_t, &l_conv, &QGn, &KS, &cps, &cpr);
  return ((size_t)KS * c_out * (k_t * c_in + 1)) * sizeof(float);
}

extern "C" int slu_wconv_bwd_weight(const float* d_conv, const float* in, float* d_weight,
                                    float* d_bias, int64_t B, int64_t l_in, int64_t c_in,
                                    int64_t c_out, int64_t k_t, int64_t stride_t, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  SLU_REQUIRE(d_conv && in && d_weight, "slu_wconv_bwd_weight: null pointer");
  hipStream_t st = (hipStream_t)stream;
  int64_t l_conv; int QGn, KS, cps, cpr;
  dw_geometry(B, l_in, c_in, c_out, k_t, stride_t, &l_conv, &QGn, &KS, &cps, &cpr);
  const int64_t Kw = k_t * c_in, Kw1 = Kw + 1;
  const size_t need = (size_t)KS * c_out * Kw1 * sizeof(float);
  if (!workspace || workspace_bytes < need)
    SLU_FAIL(SLU_ERR_WORKSPACE, "slu_wconv_bwd_weight: workspace too small (%zu < %zu)", workspace_bytes, need);
  const int MTC = (int)cdiv(c_out, 16);
  if (MTC > 8) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_wconv_bwd_weight: at most 128 output channels");
  WconvDwParams p;
  p.d_conv = d_conv; p.in = in; p.ws = reinterpret_cast<float*>(workspace);
  p.in_row = l_in * c_in;
  p.S = (int)(stride_t * c_in);
  p.Kw = (int)Kw; p.Kw1 = (int)Kw1; p.pad = (int)((k_t / 2) * c_in);
  p.l_conv = (int)l_conv; p.c_out = (int)c_out;
  p.chunks_per_row = cpr; p.chunks_total = (int)(B * cpr); p.chunks_per_split = cps;
  p.win = (int)(cdiv((int64_t)(DW_FC - 1) * p.S + DW_QG, 4) * 4);
  p.bias_tap = d_bias ? (int)Kw : -1;
  if (p.win > DW_NIN * WC_THREADS * 4)
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_wconv_bwd_weight: stride*c_in = %d too large (window of %d floats)", p.S, p.win);
  dim3 grid((unsigned)QGn, (unsigned)KS);
  const size_t lds = ((size_t)p.win + (size_t)DW_FC * c_out + (size_t)MTC * 16) * sizeof(float);
  if (lds > 160 * 1024) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_wconv_bwd_weight: LDS need %zu exceeds 160 KiB", lds);
  if (lds > 64 * 1024) {
    const void* fn = nullptr;
    switch (MTC) {
      case 1: fn = (const void*)wconv_dw_kernel<1>; break;
      case 2: fn = (const void*)wconv_dw_kernel<2>; break;
      case 3: fn = (const void*)wconv_dw_kernel<3>; break;
      case 4: fn = (const void*)wconv_dw_kernel<4>; break;
      case 5: fn = (const void*)wconv_dw_kernel<5>; break;
      case 6: fn = (const void*)wconv_dw_kernel<6>; break;
      case 7: fn = (const void*)wconv_dw_kernel<7>; break;
      default: fn = (const void*)wconv_dw_kernel<8>; break;
    }
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "slu_wconv_bwd_weight: cannot raise the dynamic LDS cap to %zu: %s", lds, hipGetErrorString(e));
  }
  switch (MTC) {
    case 1: hipLaunchKernelGGL(wconv_dw_kernel<1>, grid, dim3(WC_THREADS), lds, st, p); break;
    case 2: hipLaunchKernelGGL(wconv_dw_kernel<2>, grid, dim3(WC_THREADS), lds, st, p); break;
    case 3: hipLaunchKernelGGL(wconv_dw_kernel<3>, grid, dim3(WC_THREADS), lds, st, p); break;
    case 4: hipLaunchKernelGGL(wconv_dw_kernel<4>, grid, dim3(WC_THREADS), lds, st, p); break;
    case 5: hipLaunchKernelGGL(wconv_dw_kernel<5>, grid, dim3(WC_THREADS), lds, st, p); break;
    case 6: hipLaunchKernelGGL(wconv_dw_kernel<6>, grid, dim3(WC_THREADS), lds, st, p); break;
    case 7: hipLaunchKernelGGL(wconv_dw_kernel<7>, grid, dim3(WC_THREADS), lds, st, p); break;
    default: hipLaunchKernelGGL(wconv_dw_kernel<8>, grid, dim3(WC_THREADS), lds, st, p); break;
  }
  SLU_CHECK_LAUNCH("wconv_dw_kernel");
  const int total = (int)(c_out * Kw1);
  hipLaunchKernelGGL(wconv_dw_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st,
                     (const float*)p.ws, d_weight, d_bias, KS, (int)c_out, (int)c_in, (int)k_t, (int)Kw1);
  SLU_CHECK_LAUNCH("wconv_dw_reduce_kernel");
  return SLU_OK;
}
