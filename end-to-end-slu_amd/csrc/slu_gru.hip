// Persistent GRU recurrence for gfx950 (reference: torch.nn.GRU at models.py:232, :262, :686;
// h0 = 0, gate order [r; z; n], one layer, optional reverse direction; RNNSelect models.py:138-149).
//
// The recurrence is independent per sequence, so the grid is (16-sequence tile) x (direction): one
// workgroup of H/16 waves runs the whole time loop with no inter-workgroup communication.
//   * Wave w owns hidden units [16w, 16w+16).  Its slice of W_hh (3 gates x 16 units x H) stays
//     RESIDENT IN VGPRs for all T steps as the MFMA B operand (3H/4 = 96 registers for H = 128).
//   * Per step: G = h_{t-1} (16 x H) * W_hh^T via v_mfma_f32_16x16x4_f32 (exact fp32), three
//     independent accumulator chains (r, z, n) per wave so the 40-cycle MFMA dependency latency is
//     covered; h_{t-1} is read from LDS (row stride H+2 floats: conflict-free ds_read_b64).
//   * The MFMA C layout gives each lane the r, z, n pre-activations of the SAME (sequence, unit)
//     for 4 sequences, so sigmoid/tanh/hadamard/blend are fused in registers and h_{t-1} for the
//     blend never leaves the lane.  h_t goes to LDS (next step's A operand, double buffered ->
//     one barrier per step) and to HBM.
//   * x-side pre-activations gx = x W_ih^T + b_ih come from slu_gemm_f32 and are prefetched one
//     step ahead.
// The backward kernel mirrors this with W_hh consumed transposed (dh_{t-1} += dG W_hh).
#include "slu_common.h"
#include <stdlib.h>
#include <cstdlib>

namespace slu {

// Gate non-linearities use v_exp_f32 / v_rcp_f32 (1 ulp each): sigmoid(x) = rcp(1 + exp2(-x log2 e)),
// tanh(x) = 1 - 2 rcp(1 + exp2(2 x log2 e)); absolute error of the gate values < 3e-7.

__device__ __forceinline__ float act_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float act_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

struct GruFwdParams {
  const float* gx;        // (T, B, D*3H)
  const float* w_hh[2];   // (3H, H) per direction
  const float* b_hh[2];   // (3H)
  float* out;             // (T, B, D*H)
  float* reserve;         // [D][T][NBT][NW][5][64][4] or null
  int T, B, D;
};

template <int H>
__global__ void __launch_bounds__(H * 4)
gru_seq_fwd_kernel(const GruFwdParams p) {
  constexpr int NW = H / 16;      // waves
  constexpr int KQ = H / 4;       // k range per lane group
  constexpr int LD = H + 2;       // LDS row stride (floats)
  __shared__ __attribute__((aligned(16))) float hbuf[2][16 * LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kg = lane >> 4;
  const int btile = blockIdx.x, dir = blockIdx.y;
  const int NBT = gridDim.x;
  const int b0 = btile * 16;
  const int j = w * 16 + i;       // hidden unit of this lane's outputs
  const int T = p.T, B = p.B, D = p.D;

  // resident W_hh slice: B[k = kg*KQ + kk][j] = W_hh[gate*H + j][kg*KQ + kk]
  float wr[KQ], wz[KQ], wn[KQ];
  {
    const float* __restrict__ W = p.w_hh[dir];
    const float* pr = W + ((size_t)(0 * H + j)) * H + kg * KQ;
    const float* pz = W + ((size_t)(1 * H + j)) * H + kg * KQ;
    const float* pn = W + ((size_t)(2 * H + j)) * H + kg * KQ;
#pragma unroll
    for (int k = 0; k < KQ; ++k) { wr[k] = pr[k]; wz[k] = pz[k]; wn[k] = pn[k]; }
  }
  const float bhr = p.b_hh[dir][j], bhz = p.b_hh[dir][H + j], bhn = p.b_hh[dir][2 * H + j];

  for (int x = tid; x < 2 * 16 * LD; x += H * 4) (&hbuf[0][0])[x] = 0.0f;   // h0 = 0
  float hprev[4] = {0.f, 0.f, 0.f, 0.f};
  bool rowok[4];
  size_t grow[4];   // gx / out row offsets exclude t
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * kg + r;
    rowok[r] = b < B;
    grow[r] = (size_t)(rowok[r] ? b : 0);
  }
  const size_t gx_ts = (size_t)B * D * 3 * H;      // stride of t in gx
  const size_t out_ts = (size_t)B * D * H;
  const float* __restrict__ gxd = p.gx + (size_t)dir * 3 * H + j;
  float* __restrict__ outd = p.out + (size_t)dir * H + j;

  float gr[4], gz[4], gn[4];
  {
    const int t0 = dir ? T - 1 : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* g = gxd + (size_t)t0 * gx_ts + grow[r] * D * 3 * H;
      gr[r] = g[0]; gz[r] = g[H]; gn[r] = g[2 * H];   // padded rows read row 0 (never stored)
    }
  }
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const int cur = s & 1;
    // prefetch next step's x-side pre-activations
    float ngr[4], ngz[4], ngn[4];
    if (s + 1 < T) {
      const int tn = dir ? t - 1 : t + 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* g = gxd + (size_t)tn * gx_ts + grow[r] * D * 3 * H;
        ngr[r] = g[0]; ngz[r] = g[H]; ngn[r] = g[2 * H];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) { ngr[r] = 0.f; ngz[r] = 0.f; ngn[r] = 0.f; }
    }

    f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = {0.f, 0.f, 0.f, 0.f}, an = {0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ hrow = &hbuf[cur][i * LD + kg * KQ];
    float2 af[KQ / 2];
#pragma unroll
    for (int v = 0; v < KQ / 2; ++v) af[v] = *reinterpret_cast<const float2*>(hrow + 2 * v);
    __builtin_amdgcn_sched_barrier(0);     // keep all 16 LDS reads in flight ahead of the MFMA chain
#pragma unroll
    for (int v = 0; v < KQ / 2; ++v) {
      const float2 a = af[v];
      ar = mfma16(a.x, wr[2 * v], ar);
      az = mfma16(a.x, wz[2 * v], az);
      an = mfma16(a.x, wn[2 * v], an);
      ar = mfma16(a.y, wr[2 * v + 1], ar);
      az = mfma16(a.y, wz[2 * v + 1], az);
      an = mfma16(a.y, wn[2 * v + 1], an);
    }

    float rr[4], zz[4], nn[4], qq[4], hn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      rr[r] = act_sigmoid(gr[r] + (ar[r] + bhr));
      zz[r] = act_sigmoid(gz[r] + (az[r] + bhz));
      qq[r] = an[r] + bhn;
      nn[r] = act_tanh(gn[r] + rr[r] * qq[r]);
      hn[r] = (1.0f - zz[r]) * nn[r] + zz[r] * hprev[r];
    }
    float* __restrict__ hnext = &hbuf[cur ^ 1][0];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hnext[(4 * kg + r) * LD + j] = hn[r];
      if (rowok[r]) outd[(size_t)t * out_ts + grow[r] * D * H] = hn[r];
    }
    if (p.reserve) {
      float4* __restrict__ rs = reinterpret_cast<float4*>(
          p.reserve + ((((size_t)dir * T + t) * NBT + btile) * NW + w) * (5 * 256)) + lane;
      rs[0 * 64] = make_float4(rr[0], rr[1], rr[2], rr[3]);
      rs[1 * 64] = make_float4(zz[0], zz[1], zz[2], zz[3]);
      rs[2 * 64] = make_float4(nn[0], nn[1], nn[2], nn[3]);
      rs[3 * 64] = make_float4(qq[0], qq[1], qq[2], qq[3]);
      rs[4 * 64] = make_float4(hprev[0], hprev[1], hprev[2], hprev[3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { hprev[r] = hn[r]; gr[r] = ngr[r]; gz[r] = ngz[r]; gn[r] = ngn[r]; }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// 4-sequence variant for small batches.  With 16 sequences per workgroup a B = 64 layer is 8
// workgroups on a 256-CU chip, each paying the full 16-row MFMA chain per step.  Here a workgroup of
// H/32 waves owns only FOUR sequences and uses v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 blocks
// per instruction, 8 cycles): lane l = (half = l/32, unit u = l%32) owns hidden unit 32w+u and the
// k-range [half*H/2, (half+1)*H/2) of the reduction, so each of the three gate tiles is
// [32 units, k low | the same 32 units, k high] and no MFMA lane is wasted: H/2 * 3 instructions of
// 8 cycles per wave per step (1536 cycles for H = 128, a quarter of the 16-sequence chain), with the
// same 3H/4 resident W_hh registers per lane.  The A operand (h_{t-1}, identical for every block of
// a half) uses the cbsz/abid block broadcast: register q of the four lanes of block a holds
// h[seq][k(q, a)], so a lane reads H/16 floats of h per step (two ds_read_b128) instead of H/2.
// v_permlane32_swap folds the two k-halves; the lower half-wave then finishes sequences 0, 1 and the
// upper one sequences 2, 3 (gates, blend, stores).  Reserve layout and results interoperate with the
// 16-sequence kernels (summation order differs: not bit-identical between the variants).
template <int H>
__global__ void __launch_bounds__(H * 2)
gru_seq_fwd4_kernel(const GruFwdParams p, const int NBT16) {
  constexpr int NW16 = H / 16;   // waves of the 16-sequence layout (reserve indexing)
  constexpr int KS = H / 2;      // k range of a half-wave
  constexpr int NQ = KS / 8;     // A registers per lane (8 blocks x NQ = KS)
  constexpr int LD = H + 16;     // LDS row stride: rows 16 banks apart -> conflict-free b128 reads
  static_assert(NQ % 4 == 0, "gru_seq_fwd4_kernel needs H >= 64");
  __shared__ __attribute__((aligned(16))) float hbuf[2][4 * LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u = lane & 31, half = lane >> 5, blk = (lane >> 2) & 7, si = lane & 3;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 4;
  const int j = w * 32 + u;
  const int T = p.T, B = p.B, D = p.D;

  // k index (within the half) consumed by the MFMA that uses A register q with abid = a
  //   k(q, a) = (q / 4) * 32 + 4 * a + (q % 4):  register group q/4 is one ds_read_b128 at offset 4*blk.
  float wr[KS], wz[KS], wn[KS];
  {
    const float* __restrict__ W = p.w_hh[dir];
    const float* pr = W + ((size_t)(0 * H + j)) * H + half * KS;
    const float* pz = W + ((size_t)(1 * H + j)) * H + half * KS;
    const float* pn = W + ((size_t)(2 * H + j)) * H + half * KS;
#pragma unroll
    for (int k = 0; k < KS; ++k) { wr[k] = pr[k]; wz[k] = pz[k]; wn[k] = pn[k]; }
  }
  const float bhr = p.b_hh[dir][j], bhz = p.b_hh[dir][H + j], bhn = p.b_hh[dir][2 * H + j];

  for (int x = tid; x < 2 * 4 * LD; x += H * 2) (&hbuf[0][0])[x] = 0.0f;   // h0 = 0
  // this lane finishes sequences b0 + 2*half + {0, 1}
  float hprev[2] = {0.f, 0.f};
  bool rowok[2];
  size_t grow[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int b = b0 + 2 * half + e;
    rowok[e] = b < B;
    grow[e] = (size_t)(rowok[e] ? b : 0);
  }
  const size_t gx_ts = (size_t)B * D * 3 * H;
  const size_t out_ts = (size_t)B * D * H;
  const float* __restrict__ gxd = p.gx + (size_t)dir * 3 * H + j;
  float* __restrict__ outd = p.out + (size_t)dir * H + j;

  float gr[2], gz[2], gn[2];
  {
    const int t0 = dir ? T - 1 : 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float* g = gxd + (size_t)t0 * gx_ts + grow[e] * D * 3 * H;
      gr[e] = g[0]; gz[e] = g[H]; gn[e] = g[2 * H];
    }
  }
  // reserve element (sequence b, unit j, component c) lives where the 16-sequence kernel puts it
  const size_t rsv_lane = ((size_t)((j & 15) + 16 * ((b0 & 15) >> 2))) * 4 + 2 * half;
  const size_t rsv_wave = (size_t)(b0 >> 4) * NW16 + (j >> 4);
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const int cur = s & 1;
    float ngr[2], ngz[2], ngn[2];
    if (s + 1 < T) {
      const int tn = dir ? t - 1 : t + 1;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float* g = gxd + (size_t)tn * gx_ts + grow[e] * D * 3 * H;
        ngr[e] = g[0]; ngz[e] = g[H]; ngn[e] = g[2 * H];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e) { ngr[e] = 0.f; ngz[e] = 0.f; ngn[e] = 0.f; }
    }

    float af[NQ];
    {
      const float* __restrict__ hrow = &hbuf[cur][si * LD + half * KS + 4 * blk];
#pragma unroll
      for (int v = 0; v < NQ / 4; ++v) {
        const float4 x = *reinterpret_cast<const float4*>(hrow + 32 * v);
        af[4 * v + 0] = x.x; af[4 * v + 1] = x.y; af[4 * v + 2] = x.z; af[4 * v + 3] = x.w;
      }
    }
    f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = {0.f, 0.f, 0.f, 0.f}, an = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int kb = (q / 4) * 32 + (q % 4);
#define SLU_GRU4_STEP(a)                                                        \
      ar = __builtin_amdgcn_mfma_f32_4x4x1f32(af[q], wr[kb + 4 * a], ar, 3, a, 0); \
      az = __builtin_amdgcn_mfma_f32_4x4x1f32(af[q], wz[kb + 4 * a], az, 3, a, 0); \
      an = __builtin_amdgcn_mfma_f32_4x4x1f32(af[q], wn[kb + 4 * a], an, 3, a, 0);
      SLU_GRU4_STEP(0) SLU_GRU4_STEP(1) SLU_GRU4_STEP(2) SLU_GRU4_STEP(3)
      SLU_GRU4_STEP(4) SLU_GRU4_STEP(5) SLU_GRU4_STEP(6) SLU_GRU4_STEP(7)
#undef SLU_GRU4_STEP
    }
    // fold the k-halves: register pair (seq e, seq e+2) -> lower half-wave gets seq e, upper seq e+2
    float hr[2], hz[2], hq[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      auto sr = __builtin_amdgcn_permlane32_swap(__float_as_uint(ar[e]), __float_as_uint(ar[e + 2]), false, false);
      auto sz = __builtin_amdgcn_permlane32_swap(__float_as_uint(az[e]), __float_as_uint(az[e + 2]), false, false);
      auto sn = __builtin_amdgcn_permlane32_swap(__float_as_uint(an[e]), __float_as_uint(an[e + 2]), false, false);
      hr[e] = __uint_as_float(sr[0]) + __uint_as_float(sr[1]);
      hz[e] = __uint_as_float(sz[0]) + __uint_as_float(sz[1]);
      hq[e] = __uint_as_float(sn[0]) + __uint_as_float(sn[1]);
    }

    float rr[2], zz[2], nn[2], qq[2], hn[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      rr[e] = act_sigmoid(gr[e] + (hr[e] + bhr));
      zz[e] = act_sigmoid(gz[e] + (hz[e] + bhz));
      qq[e] = hq[e] + bhn;
      nn[e] = act_tanh(gn[e] + rr[e] * qq[e]);
      hn[e] = (1.0f - zz[e]) * nn[e] + zz[e] * hprev[e];
    }
    float* __restrict__ hnext = &hbuf[cur ^ 1][0];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      hnext[(2 * half + e) * LD + j] = hn[e];
      if (rowok[e]) outd[(size_t)t * out_ts + grow[e] * D * H] = hn[e];
    }
    if (p.reserve) {
      float* __restrict__ rs = p.reserve + ((((size_t)dir * T + t) * NBT16) * NW16 + rsv_wave) * (5 * 256) + rsv_lane;
      *reinterpret_cast<float2*>(rs + 0 * 256) = make_float2(rr[0], rr[1]);
      *reinterpret_cast<float2*>(rs + 1 * 256) = make_float2(zz[0], zz[1]);
      *reinterpret_cast<float2*>(rs + 2 * 256) = make_float2(nn[0], nn[1]);
      *reinterpret_cast<float2*>(rs + 3 * 256) = make_float2(qq[0], qq[1]);
      *reinterpret_cast<float2*>(rs + 4 * 256) = make_float2(hprev[0], hprev[1]);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) { hprev[e] = hn[e]; gr[e] = ngr[e]; gz[e] = ngz[e]; gn[e] = ngn[e]; }
    __syncthreads();
  }
}

struct GruBwdParams {
  const float* d_out;     // (T, B, D*H)
  const float* reserve;
  const float* w_hh[2];
  float* d_gx;            // (T, B, D*3H)  gradient w.r.t. x W_ih^T + b_ih         = [dr_pre, dz_pre, dn_pre]
  float* d_gh;            // (T, B, D*3H)  gradient w.r.t. h_{t-1} W_hh^T + b_hh   = [dr_pre, dz_pre, dq]
  float* d_bias_part;     // [NBT][D][6H] or null: sums over t and the tile's sequences of d_gx | d_gh
  int T, B, D;
};

template <int H>
__global__ void __launch_bounds__(H * 4)
gru_seq_bwd_kernel(const GruBwdParams p) {
  constexpr int NW = H / 16;
  constexpr int KQ = H / 4;
  constexpr int LDB = 3 * H + 2;
  __shared__ __attribute__((aligned(16))) float gbuf[2][16 * LDB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kg = lane >> 4;
  const int btile = blockIdx.x, dir = blockIdx.y;
  const int NBT = gridDim.x;
  const int b0 = btile * 16;
  const int j = w * 16 + i;
  const int T = p.T, B = p.B, D = p.D;

  // resident transposed slice: B[k = g][n = j] = W_hh[gate*H + kg*KQ + kk][j]
  float wr[KQ], wz[KQ], wn[KQ];
  {
    const float* __restrict__ W = p.w_hh[dir];
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
      wr[k] = W[((size_t)(0 * H + kg * KQ + k)) * H + j];
      wz[k] = W[((size_t)(1 * H + kg * KQ + k)) * H + j];
      wn[k] = W[((size_t)(2 * H + kg * KQ + k)) * H + j];
    }
  }
  bool rowok[4];
  size_t grow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * kg + r;
    rowok[r] = b < B;
    grow[r] = (size_t)(rowok[r] ? b : 0);
  }
  const size_t out_ts = (size_t)B * D * H;
  const size_t gx_ts = (size_t)B * D * 3 * H;
  const float* __restrict__ dod = p.d_out + (size_t)dir * H + j;
  float* __restrict__ dgxd = p.d_gx + (size_t)dir * 3 * H + j;
  float* __restrict__ dghd = p.d_gh + (size_t)dir * 3 * H + j;

  float dcarry[4] = {0.f, 0.f, 0.f, 0.f};
  float sbr = 0.f, sbz = 0.f, sbn = 0.f, sbq = 0.f;

  // step s of the backward pass visits the time index the forward pass visited LAST first
  auto tindex = [&](int s) { return dir ? s : T - 1 - s; };
  auto rsv = [&](int t) {
    return reinterpret_cast<const float4*>(
               p.reserve + ((((size_t)dir * T + t) * NBT + btile) * NW + w) * (5 * 256)) + lane;
  };

  // Per step, everything that does not depend on dh_t is folded into three coefficients per element
  //   cN = (1-z)(1-n^2)      dn_pre = dh * cN
  //   cZ = (h_prev-n) z(1-z)  dz_pre = dh * cZ
  //   cR = q r(1-r)           dr_pre = dn_pre * cR,   dq = dn_pre * r,   dh_direct = dh * z
  // computed for step s+1 in the shadow of step s's MFMAs, so that only 5 multiplies per element sit
  // between the arrival of dh (previous MFMA chain) and the LDS store that feeds the next one.
  float cN[4], cZ[4], cR[4], cr[4], cz[4], c_do[4];
  {
    const int t = tindex(0);
    const float4* rs = rsv(t);
    const float4 v_r = rs[0], v_z = rs[64], v_n = rs[128], v_q = rs[192], v_h = rs[256];
    const float rr[4] = {v_r.x, v_r.y, v_r.z, v_r.w}, zz[4] = {v_z.x, v_z.y, v_z.z, v_z.w};
    const float nn[4] = {v_n.x, v_n.y, v_n.z, v_n.w}, qq[4] = {v_q.x, v_q.y, v_q.z, v_q.w};
    const float hp[4] = {v_h.x, v_h.y, v_h.z, v_h.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float omz = 1.0f - zz[r];
      cN[r] = omz * (1.0f - nn[r] * nn[r]);
      cZ[r] = (hp[r] - nn[r]) * (zz[r] * omz);
      cR[r] = qq[r] * (rr[r] * (1.0f - rr[r]));
      cr[r] = rr[r]; cz[r] = zz[r];
      const float v = dod[(size_t)t * out_ts + grow[r] * D * H];
      c_do[r] = rowok[r] ? v : 0.f;
    }
  }

  for (int s = 0; s < T; ++s) {
    const int t = tindex(s);
    const int cur = s & 1;
    // next step's saved gates / upstream gradient (the last step re-reads its own: unused)
    const int tn = tindex(s + 1 < T ? s + 1 : s);
    const float4* rsn = rsv(tn);
    const float4 n_r = rsn[0], n_z = rsn[64], n_n = rsn[128], n_q = rsn[192], n_h = rsn[256];
    float n_do[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = dod[(size_t)tn * out_ts + grow[r] * D * H];
      n_do[r] = rowok[r] ? v : 0.f;
    }

    float ddirect[4];
    float* __restrict__ gcur = &gbuf[cur][0];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dh = dcarry[r] + c_do[r];
      const float dn_pre = dh * cN[r];
      const float dz_pre = dh * cZ[r];
      const float dq = dn_pre * cr[r];
      const float dr_pre = dn_pre * cR[r];
      ddirect[r] = dh * cz[r];
      const int brow = 4 * kg + r;
      gcur[brow * LDB + 0 * H + j] = dr_pre;
      gcur[brow * LDB + 1 * H + j] = dz_pre;
      gcur[brow * LDB + 2 * H + j] = dq;
      if (rowok[r]) {
        float* g = dgxd + (size_t)t * gx_ts + grow[r] * D * 3 * H;
        g[0] = dr_pre; g[H] = dz_pre; g[2 * H] = dn_pre;
        float* gh = dghd + (size_t)t * gx_ts + grow[r] * D * 3 * H;
        gh[0] = dr_pre; gh[H] = dz_pre; gh[2 * H] = dq;
        sbr += dr_pre; sbz += dz_pre; sbn += dn_pre; sbq += dq;
      }
    }
    __syncthreads();

    f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = {0.f, 0.f, 0.f, 0.f}, an = {0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ grow_l = &gbuf[cur][i * LDB + kg * KQ];
    const float nr[4] = {n_r.x, n_r.y, n_r.z, n_r.w}, nz[4] = {n_z.x, n_z.y, n_z.z, n_z.w};
    const float nnv[4] = {n_n.x, n_n.y, n_n.z, n_n.w}, nq[4] = {n_q.x, n_q.y, n_q.z, n_q.w};
    const float nh[4] = {n_h.x, n_h.y, n_h.z, n_h.w};
    float t0[4], t1[4], xN[4], xZ[4], xR[4];
    constexpr int NV = KQ / 2;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float2 a0 = *reinterpret_cast<const float2*>(grow_l + 0 * H + 2 * v);
      const float2 a1 = *reinterpret_cast<const float2*>(grow_l + 1 * H + 2 * v);
      const float2 a2 = *reinterpret_cast<const float2*>(grow_l + 2 * H + 2 * v);
      ar = mfma16(a0.x, wr[2 * v], ar);
      az = mfma16(a1.x, wz[2 * v], az);
      an = mfma16(a2.x, wn[2 * v], an);
      ar = mfma16(a0.y, wr[2 * v + 1], ar);
      az = mfma16(a1.y, wz[2 * v + 1], az);
      an = mfma16(a2.y, wn[2 * v + 1], an);
      // coefficient stages for step s+1, one per MFMA group from the middle of the chain on
      // (by then the prefetched gates have landed); all at the end for small H
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool here = (NV >= 8) ? (k == v - NV / 2) : (v == NV - 1);
        if (!here) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (k == 0) { t0[r] = 1.0f - nz[r]; t1[r] = 1.0f - nnv[r] * nnv[r]; }
          if (k == 1) { xN[r] = t0[r] * t1[r]; t0[r] = nz[r] * t0[r]; }
          if (k == 2) { xZ[r] = (nh[r] - nnv[r]) * t0[r]; t1[r] = nr[r] * (1.0f - nr[r]); }
          if (k == 3) { xR[r] = nq[r] * t1[r]; }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dcarry[r] = ddirect[r] + ((ar[r] + az[r]) + an[r]);
      cN[r] = xN[r]; cZ[r] = xZ[r]; cR[r] = xR[r]; cr[r] = nr[r]; cz[r] = nz[r];
      c_do[r] = n_do[r];
    }
    // gbuf[cur] is rewritten two steps from now; the barrier of the next step orders that.
  }

  if (p.d_bias_part) {
    // sum the 4 lane groups (same unit j, different sequences)
    sbr += __shfl_xor(sbr, 16); sbr += __shfl_xor(sbr, 32);
    sbz += __shfl_xor(sbz, 16); sbz += __shfl_xor(sbz, 32);
    sbn += __shfl_xor(sbn, 16); sbn += __shfl_xor(sbn, 32);
    sbq += __shfl_xor(sbq, 16); sbq += __shfl_xor(sbq, 32);
    if (kg == 0) {
      float* o = p.d_bias_part + ((size_t)btile * D + dir) * 6 * H;
      o[j] = sbr; o[H + j] = sbz; o[2 * H + j] = sbn;                 // d(b_ih)
      o[3 * H + j] = sbr; o[4 * H + j] = sbz; o[5 * H + j] = sbq;     // d(b_hh)
    }
  }
}

// 4-sequence BPTT (see gru_seq_fwd4_kernel): dh_{t-1} += [dr_pre, dz_pre, dq] (4 x 3H) * W_hh (3H x H).
// Lane (half, u) of wave w owns output unit 32w+u and, for each of the three gate row-blocks of W_hh,
// the k-range [half*H/2, (half+1)*H/2): 3H/4 resident registers, H/2 * 3 MFMAs per step, three
// accumulator chains (one per gate block) summed at the end.  Element ownership after the fold is the
// forward kernel's: lane (half, u) handles sequences 2*half + {0, 1} of unit 32w+u.
// d_bias_part rows are per 4-sequence tile here: [cdiv(B,4)][D][6H] (slu_gru_bias_tiles).
template <int H>
__global__ void __launch_bounds__(H * 2)
gru_seq_bwd4_kernel(const GruBwdParams p, const int NBT16) {
  constexpr int NW16 = H / 16;
  constexpr int KS = H / 2;
  constexpr int NQ = KS / 8;
  constexpr int LDB = 3 * H + 16;
  static_assert(NQ % 4 == 0, "gru_seq_bwd4_kernel needs H >= 64");
  __shared__ __attribute__((aligned(16))) float gbuf[2][4 * LDB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u = lane & 31, half = lane >> 5, blk = (lane >> 2) & 7, si = lane & 3;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 4;
  const int j = w * 32 + u;
  const int T = p.T, B = p.B, D = p.D;

  float wr[KS], wz[KS], wn[KS];   // W_hh[gate*H + half*KS + k][j]
  {
    const float* __restrict__ W = p.w_hh[dir] + j;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      wr[k] = W[((size_t)(0 * H + half * KS + k)) * H];
      wz[k] = W[((size_t)(1 * H + half * KS + k)) * H];
      wn[k] = W[((size_t)(2 * H + half * KS + k)) * H];
    }
  }
  bool rowok[2];
  size_t grow[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int b = b0 + 2 * half + e;
    rowok[e] = b < B;
    grow[e] = (size_t)(rowok[e] ? b : 0);
  }
  const size_t out_ts = (size_t)B * D * H;
  const size_t gx_ts = (size_t)B * D * 3 * H;
  const float* __restrict__ dod = p.d_out + (size_t)dir * H + j;
  float* __restrict__ dgxd = p.d_gx + (size_t)dir * 3 * H + j;
  float* __restrict__ dghd = p.d_gh + (size_t)dir * 3 * H + j;
  const size_t rsv_lane = ((size_t)((j & 15) + 16 * ((b0 & 15) >> 2))) * 4 + 2 * half;
  const size_t rsv_wave = (size_t)(b0 >> 4) * NW16 + (j >> 4);

  float dcarry[2] = {0.f, 0.f};
  float sbr = 0.f, sbz = 0.f, sbn = 0.f, sbq = 0.f;
  auto tindex = [&](int s) { return dir ? s : T - 1 - s; };
  auto rsv = [&](int t) {
    return p.reserve + ((((size_t)dir * T + t) * NBT16) * NW16 + rsv_wave) * (5 * 256) + rsv_lane;
  };

  // saved gates / upstream gradient of the current step (prefetched one step ahead)
  float2 c_r, c_z, c_n, c_q, c_h;
  float c_do[2];
  {
    const int t = tindex(0);
    const float* rs = rsv(t);
    c_r = *reinterpret_cast<const float2*>(rs);
    c_z = *reinterpret_cast<const float2*>(rs + 256);
    c_n = *reinterpret_cast<const float2*>(rs + 512);
    c_q = *reinterpret_cast<const float2*>(rs + 768);
    c_h = *reinterpret_cast<const float2*>(rs + 1024);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float v = dod[(size_t)t * out_ts + grow[e] * D * H];
      c_do[e] = rowok[e] ? v : 0.f;
    }
  }

  for (int s = 0; s < T; ++s) {
    const int t = tindex(s);
    const int cur = s & 1;
    const int tn = tindex(s + 1 < T ? s + 1 : s);
    const float* rsn = rsv(tn);
    const float2 n_r = *reinterpret_cast<const float2*>(rsn);
    const float2 n_z = *reinterpret_cast<const float2*>(rsn + 256);
    const float2 n_n = *reinterpret_cast<const float2*>(rsn + 512);
    const float2 n_q = *reinterpret_cast<const float2*>(rsn + 768);
    const float2 n_h = *reinterpret_cast<const float2*>(rsn + 1024);
    float n_do[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float v = dod[(size_t)tn * out_ts + grow[e] * D * H];
      n_do[e] = rowok[e] ? v : 0.f;
    }

    const float rr[2] = {c_r.x, c_r.y}, zz[2] = {c_z.x, c_z.y}, nn[2] = {c_n.x, c_n.y};
    const float qq[2] = {c_q.x, c_q.y}, hp[2] = {c_h.x, c_h.y};
    float ddirect[2];
    float* __restrict__ gcur = &gbuf[cur][0];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float dh = dcarry[e] + c_do[e];
      const float omz = 1.0f - zz[e];
      const float dn_pre = dh * (omz * (1.0f - nn[e] * nn[e]));
      const float dz_pre = dh * ((hp[e] - nn[e]) * (zz[e] * omz));
      const float dq = dn_pre * rr[e];
      const float dr_pre = dn_pre * (qq[e] * (rr[e] * (1.0f - rr[e])));
      ddirect[e] = dh * zz[e];
      const int row = 2 * half + e;
      gcur[row * LDB + 0 * H + j] = dr_pre;
      gcur[row * LDB + 1 * H + j] = dz_pre;
      gcur[row * LDB + 2 * H + j] = dq;
      if (rowok[e]) {
        float* g = dgxd + (size_t)t * gx_ts + grow[e] * D * 3 * H;
        g[0] = dr_pre; g[H] = dz_pre; g[2 * H] = dn_pre;
        float* gh = dghd + (size_t)t * gx_ts + grow[e] * D * 3 * H;
        gh[0] = dr_pre; gh[H] = dz_pre; gh[2 * H] = dq;
        sbr += dr_pre; sbz += dz_pre; sbn += dn_pre; sbq += dq;
      }
    }
    __syncthreads();

    float a0[NQ], a1[NQ], a2[NQ];
    {
      const float* __restrict__ grow_l = &gbuf[cur][si * LDB + half * KS + 4 * blk];
#pragma unroll
      for (int v = 0; v < NQ / 4; ++v) {
        const float4 x0 = *reinterpret_cast<const float4*>(grow_l + 0 * H + 32 * v);
        const float4 x1 = *reinterpret_cast<const float4*>(grow_l + 1 * H + 32 * v);
        const float4 x2 = *reinterpret_cast<const float4*>(grow_l + 2 * H + 32 * v);
        a0[4 * v] = x0.x; a0[4 * v + 1] = x0.y; a0[4 * v + 2] = x0.z; a0[4 * v + 3] = x0.w;
        a1[4 * v] = x1.x; a1[4 * v + 1] = x1.y; a1[4 * v + 2] = x1.z; a1[4 * v + 3] = x1.w;
        a2[4 * v] = x2.x; a2[4 * v + 1] = x2.y; a2[4 * v + 2] = x2.z; a2[4 * v + 3] = x2.w;
      }
    }
    f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = {0.f, 0.f, 0.f, 0.f}, an = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int kb = (q / 4) * 32 + (q % 4);
#define SLU_GRU4_STEP(a)                                                        \
      ar = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[q], wr[kb + 4 * a], ar, 3, a, 0); \
      az = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[q], wz[kb + 4 * a], az, 3, a, 0); \
      an = __builtin_amdgcn_mfma_f32_4x4x1f32(a2[q], wn[kb + 4 * a], an, 3, a, 0);
      SLU_GRU4_STEP(0) SLU_GRU4_STEP(1) SLU_GRU4_STEP(2) SLU_GRU4_STEP(3)
      SLU_GRU4_STEP(4) SLU_GRU4_STEP(5) SLU_GRU4_STEP(6) SLU_GRU4_STEP(7)
#undef SLU_GRU4_STEP
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float lo = (ar[e] + az[e]) + an[e];
      const float hi = (ar[e + 2] + az[e + 2]) + an[e + 2];
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
      dcarry[e] = ddirect[e] + (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
      c_do[e] = n_do[e];
    }
    c_r = n_r; c_z = n_z; c_n = n_n; c_q = n_q; c_h = n_h;
    // gbuf[cur] is rewritten two steps from now; the barrier of the next step orders that.
  }

  if (p.d_bias_part) {
    sbr += __shfl_xor(sbr, 32); sbz += __shfl_xor(sbz, 32);
    sbn += __shfl_xor(sbn, 32); sbq += __shfl_xor(sbq, 32);
    if (half == 0) {
      float* o = p.d_bias_part + ((size_t)blockIdx.x * D + dir) * 6 * H;
      o[j] = sbr; o[H + j] = sbz; o[2 * H + j] = sbn;
      o[3 * H + j] = sbr; o[4 * H + j] = sbz; o[5 * H + j] = sbq;
    }
  }
}

}  // namespace slu

// H-generic path (slu_gru_step.hip): one launch per time step
namespace slu {
int gru_step_bias_splits(int64_t T, int64_t B);
size_t gru_step_reserve_floats(int64_t T, int64_t B, int64_t H, int64_t D);
int gru_step_fwd(const float* gx, const float* const w_hh[2], const float* const b_hh[2], float* out,
                 float* reserve, int64_t T, int64_t B, int64_t H, int64_t D, hipStream_t st);
int gru_step_bwd(const float* d_out, const float* reserve, const float* const w_hh[2], float* d_gx,
                 float* d_gh, float* d_bias_part, int64_t T, int64_t B, int64_t H, int64_t D, hipStream_t st);
}  // namespace slu

using namespace slu;

// hidden sizes with a persistent (W_hh resident in VGPRs) instantiation; every other H runs step by step
static bool gru_persistent(int64_t H) { return H == 16 || H == 32 || H == 64 || H == 128; }

extern "C" size_t slu_gru_reserve_bytes(int64_t T, int64_t B, int64_t H, int64_t D) {
  if (!gru_persistent(H)) return gru_step_reserve_floats(T, B, H, D) * sizeof(float);
  const int64_t nbt = cdiv(B, 16);
  return (size_t)(D * T * nbt * (H / 16) * 5 * 256) * sizeof(float);
}

// 4-sequence workgroups while the 16-sequence grid would leave CUs idle (SLU_GRU_TILE=4|16 forces one)
static bool gru_use_seq4(int64_t B, int64_t H, int64_t D) {
  const char* env = getenv("SLU_GRU_TILE");          // read per call: tests switch it at run time
  const int forced = env ? atoi(env) : 0;
  if (H != 64 && H != 128) return false;
  if (forced == 4) return true;
  if (forced == 16) return false;
  return cdiv(B, 16) * D < 256;
}

extern "C" int64_t slu_gru_bias_tiles(int64_t T, int64_t B, int64_t H, int64_t D) {
  if (!gru_persistent(H)) return gru_step_bias_splits(T, B);
  return gru_use_seq4(B, H, D) ? cdiv(B, 4) : cdiv(B, 16);
}

static int gru_check(const char* who, int64_t T, int64_t B, int64_t H, int64_t D) {
  SLU_REQUIRE(T > 0 && B > 0, "%s: non-positive T or B", who);
  SLU_REQUIRE(D == 1 || D == 2, "%s: D must be 1 or 2", who);
  SLU_REQUIRE(H > 0 && H <= 8192, "%s: hidden size %lld outside [1, 8192]", who, (long long)H);
  SLU_REQUIRE(cdiv(B, 16) <= 65535, "%s: B too large", who);
  return SLU_OK;
}

extern "C" int slu_gru_seq_fwd(const float* gx, const float* w_hh_fwd, const float* w_hh_rev,
                               const float* b_hh_fwd, const float* b_hh_rev, float* out,
                               float* reserve, int64_t T, int64_t B, int64_t H, int64_t D,
                               void* stream) {
  SLU_REQUIRE(gx && w_hh_fwd && b_hh_fwd && out, "slu_gru_seq_fwd: null pointer");
  SLU_REQUIRE(D == 1 || (w_hh_rev && b_hh_rev), "slu_gru_seq_fwd: reverse weights missing");
  int rc = gru_check("slu_gru_seq_fwd", T, B, H, D);
  if (rc) return rc;
  GruFwdParams p;
  p.gx = gx; p.w_hh[0] = w_hh_fwd; p.w_hh[1] = w_hh_rev; p.b_hh[0] = b_hh_fwd; p.b_hh[1] = b_hh_rev;
  p.out = out; p.reserve = reserve; p.T = (int)T; p.B = (int)B; p.D = (int)D;
  hipStream_t st = (hipStream_t)stream;
  if (!gru_persistent(H)) return gru_step_fwd(gx, p.w_hh, p.b_hh, out, reserve, T, B, H, D, st);
  if (gru_use_seq4(B, H, D)) {
    dim3 grid4((unsigned)cdiv(B, 4), (unsigned)D);
    const int nbt16 = (int)cdiv(B, 16);
    if (H == 64) hipLaunchKernelGGL(gru_seq_fwd4_kernel<64>, grid4, dim3(128), 0, st, p, nbt16);
    else hipLaunchKernelGGL(gru_seq_fwd4_kernel<128>, grid4, dim3(256), 0, st, p, nbt16);
    SLU_CHECK_LAUNCH("gru_seq_fwd4_kernel");
    return SLU_OK;
  }
  dim3 grid((unsigned)cdiv(B, 16), (unsigned)D);
  switch (H) {
    case 16: hipLaunchKernelGGL(gru_seq_fwd_kernel<16>, grid, dim3(64), 0, st, p); break;
    case 32: hipLaunchKernelGGL(gru_seq_fwd_kernel<32>, grid, dim3(128), 0, st, p); break;
    case 64: hipLaunchKernelGGL(gru_seq_fwd_kernel<64>, grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL(gru_seq_fwd_kernel<128>, grid, dim3(512), 0, st, p); break;
  }
  SLU_CHECK_LAUNCH("gru_seq_fwd_kernel");
  return SLU_OK;
}

extern "C" int slu_gru_seq_bwd(const float* d_out, const float* reserve, const float* w_hh_fwd,
                               const float* w_hh_rev, float* d_gx, float* d_gh, float* d_bias_part,
                               int64_t T, int64_t B, int64_t H, int64_t D, void* stream) {
  SLU_REQUIRE(d_out && reserve && w_hh_fwd && d_gx && d_gh, "slu_gru_seq_bwd: null pointer");
  SLU_REQUIRE(D == 1 || w_hh_rev, "slu_gru_seq_bwd: reverse weights missing");
  int rc = gru_check("slu_gru_seq_bwd", T, B, H, D);
  if (rc) return rc;
  GruBwdParams p;
  p.d_out = d_out; p.reserve = reserve; p.w_hh[0] = w_hh_fwd; p.w_hh[1] = w_hh_rev;
  p.d_gx = d_gx; p.d_gh = d_gh; p.d_bias_part = d_bias_part; p.T = (int)T; p.B = (int)B; p.D = (int)D;
  hipStream_t st = (hipStream_t)stream;
  if (!gru_persistent(H)) return gru_step_bwd(d_out, reserve, p.w_hh, d_gx, d_gh, d_bias_part, T, B, H, D, st);
  if (gru_use_seq4(B, H, D)) {
    dim3 grid4((unsigned)cdiv(B, 4), (unsigned)D);
    const int nbt16 = (int)cdiv(B, 16);
    if (H == 64) hipLaunchKernelGGL(gru_seq_bwd4_kernel<64>, grid4, dim3(128), 0, st, p, nbt16);
    else hipLaunchKernelGGL(gru_seq_bwd4_kernel<128>, grid4, dim3(256), 0, st, p, nbt16);
    SLU_CHECK_LAUNCH("gru_seq_bwd4_kernel");
    return SLU_OK;
  }
  dim3 grid((unsigned)cdiv(B, 16), (unsigned)D);
  switch (H) {
    case 16: hipLaunchKernelGGL(gru_seq_bwd_kernel<16>, grid, dim3(64), 0, st, p); break;
    case 32: hipLaunchKernelGGL(gru_seq_bwd_kernel<32>, grid, dim3(128), 0, st, p); break;
    case 64: hipLaunchKernelGGL(gru_seq_bwd_kernel<64>, grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL(gru_seq_bwd_kernel<128>, grid, dim3(512), 0, st, p); break;
  }
  SLU_CHECK_LAUNCH("gru_seq_bwd_kernel");
  return SLU_OK;
}
