// H-generic GRU recurrence for gfx950: hidden sizes the persistent kernels of slu_gru.hip are not
// instantiated for (reference: torch.nn.GRU accepts any hidden_size, models.py:232/:262/:686; SURVEY
// §8.0-A asks for an H = 512 point).  For H > 128 the W_hh slice of a sequence tile no longer fits one
// CU's registers (H = 512: 3 MB), so the time loop moves to the host side of the C ABI: ONE launch per
// time step, both directions, the whole chip working on that step:
//   grid = (unit tiles of 16) x (sequence tiles of 16) x D, four waves per workgroup splitting the
//   reduction (k) range, v_mfma_f32_16x16x4_f32 (exact fp32) with the k index remapped so that a lane
//   group owns four consecutive k (one 16-byte load per operand row and 16 k), cross-wave reduction in
//   LDS, gates fused.  h_{t-1} is read back from the output buffer (written by the previous launch),
//   W_hh comes from L2 (3 MB at H = 512).  B = 64, H = 512: 4 x 32 x 2 = 256 workgroups, one per CU.
// The launches of consecutive steps are ordered by the stream; under hipGraph capture they become T
// kernel nodes.  Saved gates use a plain [D][T][B][5][H] layout private to this path.
#include "slu_common.h"

namespace slu {

__device__ __forceinline__ float step_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float step_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// 4 consecutive floats p[0..3] of a row of `n` valid elements starting at element index `idx`
__device__ __forceinline__ float4 ld4(const float* __restrict__ row, int idx, int n) {
  const float* p = row + idx;
  if (idx + 3 < n && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) return *reinterpret_cast<const float4*>(p);
  float4 v;
  v.x = idx < n ? p[0] : 0.0f;
  v.y = idx + 1 < n ? p[1] : 0.0f;
  v.z = idx + 2 < n ? p[2] : 0.0f;
  v.w = idx + 3 < n ? p[3] : 0.0f;
  return v;
}

struct GruStepFwd {
  const float* gx;        // (T, B, D*3H)
  const float* w_hh[2];   // (3H, H)
  const float* b_hh[2];   // (3H)
  float* out;             // (T, B, D*H)
  float* reserve;         // [D][T][B][5][H] or null
  int T, B, D, H;
  int s;                  // step index: direction 0 visits t = s, direction 1 t = T-1-s
};

__global__ void __launch_bounds__(256)
gru_step_fwd_kernel(const GruStepFwd p) {
  __shared__ float red[3][4][256];                  // [gate][wave][lane*4 + r]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kg = lane >> 4;
  const int H = p.H, B = p.B, D = p.D, T = p.T;
  const int dir = blockIdx.z;
  const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int t = dir ? T - 1 - p.s : p.s;
  const int tprev = dir ? t + 1 : t - 1;
  const size_t out_ts = (size_t)B * D * H;

  f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = {0.f, 0.f, 0.f, 0.f}, an = {0.f, 0.f, 0.f, 0.f};
  if (p.s > 0) {                                    // h0 = 0: the first step has no recurrent term
    // this wave's share of the reduction range, in chunks of 16 k
    const int chunks = (H + 15) / 16;
    const int per = (chunks + 3) / 4;
    const int c0 = w * per, c1 = min(chunks, c0 + per);
    const int brow = b0 + i, jrow = j0 + i;
    const bool bok = brow < B, jok = jrow < H;
    const float* __restrict__ hrow = p.out + (size_t)tprev * out_ts + (size_t)(bok ? brow : 0) * D * H + (size_t)dir * H;
    const float* __restrict__ W = p.w_hh[dir];
    const float* __restrict__ wr = W + (size_t)(0 * H + (jok ? jrow : 0)) * H;
    const float* __restrict__ wz = W + (size_t)(1 * H + (jok ? jrow : 0)) * H;
    const float* __restrict__ wn = W + (size_t)(2 * H + (jok ? jrow : 0)) * H;
    for (int c = c0; c < c1; ++c) {
      const int k = c * 16 + kg * 4;
      float4 a = ld4(hrow, k, bok ? H : 0);
      float4 vr = ld4(wr, k, jok ? H : 0);
      float4 vz = ld4(wz, k, jok ? H : 0);
      float4 vn = ld4(wn, k, jok ? H : 0);
      ar = mfma16(a.x, vr.x, ar); az = mfma16(a.x, vz.x, az); an = mfma16(a.x, vn.x, an);
      ar = mfma16(a.y, vr.y, ar); az = mfma16(a.y, vz.y, az); an = mfma16(a.y, vn.y, an);
      ar = mfma16(a.z, vr.z, ar); az = mfma16(a.z, vz.z, az); an = mfma16(a.z, vn.z, an);
      ar = mfma16(a.w, vr.w, ar); az = mfma16(a.w, vz.w, az); an = mfma16(a.w, vn.w, an);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[0][w][lane * 4 + r] = ar[r];
    red[1][w][lane * 4 + r] = az[r];
    red[2][w][lane * 4 + r] = an[r];
  }
  __syncthreads();
  // thread tid finishes element e = tid of the 16 x 16 tile: MFMA layout element (lane l, reg r) is
  // row 4*(l >> 4) + r, column l & 15  ->  e = l*4 + r
  {
    const int l = tid >> 2, r = tid & 3;
    const int row = 4 * (l >> 4) + r, col = l & 15;
    const int b = b0 + row, j = j0 + col;
    if (b < B && j < H) {
      const float hr_ = ((red[0][0][tid] + red[0][1][tid]) + red[0][2][tid]) + red[0][3][tid];
      const float hz_ = ((red[1][0][tid] + red[1][1][tid]) + red[1][2][tid]) + red[1][3][tid];
      const float hn_ = ((red[2][0][tid] + red[2][1][tid]) + red[2][2][tid]) + red[2][3][tid];
      const float* __restrict__ g = p.gx + ((size_t)t * B + b) * D * 3 * H + (size_t)dir * 3 * H + j;
      const float* __restrict__ bh = p.b_hh[dir];
      const float hprev = p.s > 0 ? p.out[(size_t)tprev * out_ts + (size_t)b * D * H + (size_t)dir * H + j] : 0.0f;
      const float rr = step_sigmoid(g[0] + (hr_ + bh[j]));
      const float zz = step_sigmoid(g[H] + (hz_ + bh[H + j]));
      const float qq = hn_ + bh[2 * H + j];
      const float nn = step_tanh(g[2 * H] + rr * qq);
      const float hn = (1.0f - zz) * nn + zz * hprev;
      p.out[(size_t)t * out_ts + (size_t)b * D * H + (size_t)dir * H + j] = hn;
      if (p.reserve) {
        float* __restrict__ rs = p.reserve + ((((size_t)dir * T + t) * B + b) * 5) * H + j;
        rs[0] = rr; rs[(size_t)H] = zz; rs[2 * (size_t)H] = nn; rs[3 * (size_t)H] = qq; rs[4 * (size_t)H] = hprev;
      }
    }
  }
}

struct GruStepBwd {
  const float* d_out;     // (T, B, D*H)
  const float* reserve;   // [D][T][B][5][H]
  const float* w_hh[2];
  float* d_gx;            // (T, B, D*3H)
  float* d_gh;            // (T, B, D*3H)
  float* ddirect;         // [2][D][B][H] ping-pong: dh_t * z_t of the step processed last
  int T, B, D, H;
  int s;                  // backward step index: visits the time index the forward pass visited LAST first
};

// dh_t = d_out[t] + dh_{t+}*z_{t+} + d_gh[t+] (B x 3H) * W_hh (3H x H)   (t+ = the step processed before),
// then the gate gradients of step t (same formulas as gru_seq_bwd_kernel).
__global__ void __launch_bounds__(256)
gru_step_bwd_kernel(const GruStepBwd p) {
  __shared__ float red[4][256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kg = lane >> 4;
  const int H = p.H, B = p.B, D = p.D, T = p.T;
  const int dir = blockIdx.z;
  const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int t = dir ? p.s : T - 1 - p.s;
  const int tnext = dir ? t - 1 : t + 1;            // the step processed just before this one
  const size_t gx_ts = (size_t)B * D * 3 * H;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (p.s > 0) {
    const int K = 3 * H;
    const int chunks = (K + 15) / 16;
    const int per = (chunks + 3) / 4;
    const int c0 = w * per, c1 = min(chunks, c0 + per);
    const int brow = b0 + i, jcol = j0 + i;
    const bool bok = brow < B, jok = jcol < H;
    const float* __restrict__ grow = p.d_gh + (size_t)tnext * gx_ts + (size_t)(bok ? brow : 0) * D * 3 * H + (size_t)dir * 3 * H;
    const float* __restrict__ W = p.w_hh[dir] + (jok ? jcol : 0);
    for (int c = c0; c < c1; ++c) {
      const int k = c * 16 + kg * 4;
      const float4 a = ld4(grow, k, bok ? K : 0);
      const float b0v = (jok && k < K) ? W[(size_t)k * H] : 0.0f;
      const float b1v = (jok && k + 1 < K) ? W[(size_t)(k + 1) * H] : 0.0f;
      const float b2v = (jok && k + 2 < K) ? W[(size_t)(k + 2) * H] : 0.0f;
      const float b3v = (jok && k + 3 < K) ? W[(size_t)(k + 3) * H] : 0.0f;
      acc = mfma16(a.x, b0v, acc);
      acc = mfma16(a.y, b1v, acc);
      acc = mfma16(a.z, b2v, acc);
      acc = mfma16(a.w, b3v, acc);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][lane * 4 + r] = acc[r];
  __syncthreads();
  {
    const int l = tid >> 2, r = tid & 3;
    const int row = 4 * (l >> 4) + r, col = l & 15;
    const int b = b0 + row, j = j0 + col;
    if (b < B && j < H) {
      const size_t dd = ((size_t)dir * B + b) * H + j;
      const size_t plane = (size_t)D * B * H;
      float dh = p.d_out[((size_t)t * B + b) * D * H + (size_t)dir * H + j];
      if (p.s > 0) dh += (((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid]) + p.ddirect[((p.s - 1) & 1) * plane + dd];
      const float* __restrict__ rs = p.reserve + ((((size_t)dir * T + t) * B + b) * 5) * H + j;
      const float rr = rs[0], zz = rs[(size_t)H], nn = rs[2 * (size_t)H], qq = rs[3 * (size_t)H], hp = rs[4 * (size_t)H];
      const float omz = 1.0f - zz;
      const float dn_pre = dh * (omz * (1.0f - nn * nn));
      const float dz_pre = dh * ((hp - nn) * (zz * omz));
      const float dq = dn_pre * rr;
      const float dr_pre = dn_pre * (qq * (rr * (1.0f - rr)));
      p.ddirect[(p.s & 1) * plane + dd] = dh * zz;
      float* __restrict__ g = p.d_gx + (size_t)t * gx_ts + (size_t)b * D * 3 * H + (size_t)dir * 3 * H + j;
      g[0] = dr_pre; g[H] = dz_pre; g[2 * H] = dn_pre;
      float* __restrict__ gh = p.d_gh + (size_t)t * gx_ts + (size_t)b * D * 3 * H + (size_t)dir * 3 * H + j;
      gh[0] = dr_pre; gh[H] = dz_pre; gh[2 * H] = dq;
    }
  }
}

// d_bias_part[split][d][0:3H] = sum over the split's (t, b) rows of d_gx[., d, :]; [3H:6H] likewise of d_gh.
__global__ void __launch_bounds__(256)
gru_step_bias_kernel(const float* __restrict__ d_gx, const float* __restrict__ d_gh, float* __restrict__ part,
                     int rows, int D, int H, int rows_per_split) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;            // in [0, D*3H)
  const int split = blockIdx.y;
  const int r0 = split * rows_per_split, r1 = min(rows, r0 + rows_per_split);
  const int W3 = D * 3 * H;
  float sx = 0.f, sh = 0.f;
  if (col < W3)
    for (int r = r0 + w; r < r1; r += 4) { sx += d_gx[(size_t)r * W3 + col]; sh += d_gh[(size_t)r * W3 + col]; }
  red[0][w][lane] = sx; red[1][w][lane] = sh;
  __syncthreads();
  if (w == 0 && col < W3) {
    const int d = col / (3 * H), c = col - d * 3 * H;
    float* o = part + ((size_t)split * D + d) * 6 * H;
    o[c] = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
    o[3 * H + c] = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
  }
}

int gru_step_bias_splits(int64_t T, int64_t B) {
  const int64_t rows = T * B;
  int64_t s = cdiv(rows, 64);
  return (int)(s > 64 ? 64 : (s < 1 ? 1 : s));
}

size_t gru_step_reserve_floats(int64_t T, int64_t B, int64_t H, int64_t D) {
  // saved gates [D][T][B][5][H] + the BPTT's ping-pong dh*z buffer [2][D][B][H]
  return (size_t)(D * T * B * 5 * H) + (size_t)(2 * D * B * H);
}

int gru_step_fwd(const float* gx, const float* const w_hh[2], const float* const b_hh[2], float* out,
                 float* reserve, int64_t T, int64_t B, int64_t H, int64_t D, hipStream_t st) {
  GruStepFwd p;
  p.gx = gx; p.w_hh[0] = w_hh[0]; p.w_hh[1] = w_hh[1]; p.b_hh[0] = b_hh[0]; p.b_hh[1] = b_hh[1];
  p.out = out; p.reserve = reserve; p.T = (int)T; p.B = (int)B; p.D = (int)D; p.H = (int)H;
  dim3 grid((unsigned)cdiv(H, 16), (unsigned)cdiv(B, 16), (unsigned)D);
  for (int s = 0; s < (int)T; ++s) {
    p.s = s;
    hipLaunchKernelGGL(gru_step_fwd_kernel, grid, dim3(256), 0, st, p);
  }
  SLU_CHECK_LAUNCH("gru_step_fwd_kernel");
  return SLU_OK;
}

int gru_step_bwd(const float* d_out, const float* reserve, const float* const w_hh[2], float* d_gx,
                 float* d_gh, float* d_bias_part, int64_t T, int64_t B, int64_t H, int64_t D, hipStream_t st) {
  GruStepBwd p;
  p.d_out = d_out; p.reserve = reserve; p.w_hh[0] = w_hh[0]; p.w_hh[1] = w_hh[1];
  p.d_gx = d_gx; p.d_gh = d_gh; p.T = (int)T; p.B = (int)B; p.D = (int)D; p.H = (int)H;
  // the ping-pong buffer lives behind the saved gates (the reserve is caller-owned scratch in backward)
  p.ddirect = const_cast<float*>(reserve) + (size_t)(D * T * B * 5 * H);
  dim3 grid((unsigned)cdiv(H, 16), (unsigned)cdiv(B, 16), (unsigned)D);
  for (int s = 0; s < (int)T; ++s) {
    p.s = s;
    hipLaunchKernelGGL(gru_step_bwd_kernel, grid, dim3(256), 0, st, p);
  }
  SLU_CHECK_LAUNCH("gru_step_bwd_kernel");
  if (d_bias_part) {
    const int splits = gru_step_bias_splits(T, B);
    const int rows = (int)(T * B);
    const int rps = (int)cdiv(rows, splits);
    hipLaunchKernelGGL(gru_step_bias_kernel, dim3((unsigned)cdiv(D * 3 * H, 64), (unsigned)splits), dim3(256), 0, st,
                       (const float*)d_gx, (const float*)d_gh, d_bias_part, rows, (int)D, (int)H, rps);
    SLU_CHECK_LAUNCH("gru_step_bias_kernel");
  }
  return SLU_OK;
}

}  // namespace slu
