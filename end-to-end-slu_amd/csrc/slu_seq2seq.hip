// Kernels of the seq2seq intent decoder (reference models.py:418-438 Attention, :440-485 DecoderRNN, :504-557
// Seq2SeqDecoder.forward) that are not plain GEMMs: the GRUCell gate math (+ the nn.Dropout that follows each cell),
// dot-product attention over the encoder states, and the log-softmax / label pick of one decoding step — forward and
// backward, so that the teacher-forced decoder trains on the HIP kernels (the Linear layers and the cells'
// projections go through slu_gemm_f32, their weight gradients through one GEMM per weight over the whole
// (steps x batch) history).
//
// Sizes are tiny (batch x 256 hidden, 19-63 encoder frames, ~100 labels) and every step depends on the previous
// one: these kernels are latency-bound by construction; they are written to be exact (expf / tanhf, fixed
// reduction trees — deterministic) rather than clever.  One workgroup per utterance where a reduction is needed.
#include "slu_common.h"
#include "slu_philox.h"

namespace slu {

// ------------------------------------------------------------------------------------------------------------
// GRUCell: r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) n + z h
// (torch.nn.GRUCell, gate order [r; z; n]; gi / gh include their biases)
// ------------------------------------------------------------------------------------------------------------
struct CellArgs {
  const float* gi; const float* gh;          // (B, 3H) contiguous
  const float* h_prev; long long ld_prev;    // (B, H), row stride
  float* h_out; long long ld_out;            // (B, H), row stride
  float* save;                               // null or (4, B, H): r, z, n, gh_n
  float* drop_out;                           // null or (B, H) contiguous: h' * keep / (1 - p)
  const float* mask;                         // null (Philox) or (B, H) contiguous {0, 1}
  float p, scale;
  unsigned long long seed, offset; const unsigned long long* offset_dev; unsigned long long idx_base;
  int B, H;
};

__device__ __forceinline__ float cell_keep(const float* mask, float p, float scale, unsigned long long seed,
                                           unsigned long long offset, const unsigned long long* offset_dev,
                                           unsigned long long idx_base, int e) {
  if (p <= 0.0f) return 1.0f;
  if (mask) return mask[e] * scale;
  const unsigned long long off = offset + (offset_dev ? *offset_dev : 0ull);
  return philox_uniform(seed, off, idx_base + (unsigned long long)e) < (1.0f - p) ? scale : 0.0f;
}

__global__ void __launch_bounds__(256)
gru_cell_fwd_kernel(const CellArgs a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= a.B * a.H) return;
  const int b = e / a.H, j = e - b * a.H;
  const float* gi = a.gi + (size_t)b * 3 * a.H;
  const float* gh = a.gh + (size_t)b * 3 * a.H;
  const float r = sigmoidf_acc(gi[j] + gh[j]);
  const float z = sigmoidf_acc(gi[a.H + j] + gh[a.H + j]);
  const float ghn = gh[2 * a.H + j];
  const float n = tanhf(gi[2 * a.H + j] + r * ghn);
  const float hp = a.h_prev[(size_t)b * a.ld_prev + j];
  const float h = (1.0f - z) * n + z * hp;
  a.h_out[(size_t)b * a.ld_out + j] = h;
  if (a.save) {
    const size_t BH = (size_t)a.B * a.H;
    a.save[e] = r; a.save[BH + e] = z; a.save[2 * BH + e] = n; a.save[3 * BH + e] = ghn;
  }
  if (a.drop_out) a.drop_out[e] = h * cell_keep(a.mask, a.p, a.scale, a.seed, a.offset, a.offset_dev, a.idx_base, e);
}

struct CellBwdArgs {
  const float* d_h; long long ld_dh;         // (B, H): gradient w.r.t. h' (the state slice)
  const float* d_drop;                       // null or (B, H) contiguous: gradient w.r.t. the dropped output
  const float* save;                         // (4, B, H)
  const float* h_prev; long long ld_prev;
  float* d_gi; float* d_gh;                  // (B, 3H) contiguous
  float* d_h_prev; long long ld_dprev;       // (B, H): = dh * z (may alias d_h)
  const float* mask; float p, scale;
  unsigned long long seed, offset; const unsigned long long* offset_dev; unsigned long long idx_base;
  int B, H;
};

__global__ void __launch_bounds__(256)
gru_cell_bwd_kernel(const CellBwdArgs a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= a.B * a.H) return;
  const int b = e / a.H, j = e - b * a.H;
  const size_t BH = (size_t)a.B * a.H;
  float dh = a.d_h ? a.d_h[(size_t)b * a.ld_dh + j] : 0.0f;
  if (a.d_drop) dh += a.d_drop[e] * cell_keep(a.mask, a.p, a.scale, a.seed, a.offset, a.offset_dev, a.idx_base, e);
  const float r = a.save[e], z = a.save[BH + e], n = a.save[2 * BH + e], ghn = a.save[3 * BH + e];
  const float hp = a.h_prev[(size_t)b * a.ld_prev + j];
  const float dn = dh * (1.0f - z);
  const float dz = dh * (hp - n);
  const float dpn = dn * (1.0f - n * n);
  const float dpr = dpn * ghn * r * (1.0f - r);
  const float dpz = dz * z * (1.0f - z);
  float* gi = a.d_gi + (size_t)b * 3 * a.H;
  float* gh = a.d_gh + (size_t)b * 3 * a.H;
  gi[j] = dpr; gi[a.H + j] = dpz; gi[2 * a.H + j] = dpn;
  gh[j] = dpr; gh[a.H + j] = dpz; gh[2 * a.H + j] = dpn * r;
  a.d_h_prev[(size_t)b * a.ld_dprev + j] = dh * z;
}

// ------------------------------------------------------------------------------------------------------------
// Attention (models.py:427-438): scores_t = <keys[b,t], query[b]> / sqrt(key_dim); a = softmax_t(scores);
// context[b] = sum_t a_t values[b,t].  keys / values are addressed as ptr + t * s_t + b * s_b (+ k): time-major or
// batch-major alike.  One workgroup of 256 threads per utterance.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum / max of one value per thread (256 threads), result broadcast; `red` = 4 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

struct AttArgs {
  const float* keys; const float* values; long long s_t, s_b, v_t, v_b;
  const float* query; long long ld_q;         // (B, Kd)
  float* ctx; long long ld_ctx;               // (B, Vd)
  float* weights;                             // (B, T) contiguous (saved for backward)
  float inv_scale;
  int B, T, Kd, Vd;
};

__global__ void __launch_bounds__(256)
attention_fwd_kernel(const AttArgs a) {
  extern __shared__ float lds[];              // q[Kd] | s[T] | red[4]
  float* q = lds; float* s = lds + a.Kd; float* red = s + a.T;
  const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  for (int k = tid; k < a.Kd; k += 256) q[k] = a.query[(size_t)b * a.ld_q + k];
  __syncthreads();
  for (int t = w; t < a.T; t += 4) {          // one wave per frame: dot product over the key dimension
    const float* kp = a.keys + (size_t)t * a.s_t + (size_t)b * a.s_b;
    float acc = 0.0f;
    for (int k = lane; k < a.Kd; k += 64) acc = fmaf(kp[k], q[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) s[t] = acc * a.inv_scale;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int t = tid; t < a.T; t += 256) m = fmaxf(m, s[t]);
  m = block_max(m, red);
  float z = 0.0f;
  for (int t = tid; t < a.T; t += 256) { const float ex = expf(s[t] - m); s[t] = ex; z += ex; }
  z = block_sum(z, red);
  const float inv = 1.0f / z;
  __syncthreads();
  for (int t = tid; t < a.T; t += 256) { const float p = s[t] * inv; s[t] = p; a.weights[(size_t)b * a.T + t] = p; }
  __syncthreads();
  for (int v = tid; v < a.Vd; v += 256) {
    float acc = 0.0f;
    for (int t = 0; t < a.T; ++t) acc = fmaf(s[t], a.values[(size_t)t * a.v_t + (size_t)b * a.v_b + v], acc);
    a.ctx[(size_t)b * a.ld_ctx + v] = acc;
  }
}

struct AttBwdArgs {
  const float* keys; const float* values; long long s_t, s_b, v_t, v_b;
  const float* query; long long ld_q;
  const float* d_ctx; long long ld_dctx;      // (B, Vd)
  const float* weights;                       // (B, T)
  float* d_keys; float* d_values;             // same addressing as keys / values; ACCUMULATED into (+=)
  float* d_query; long long ld_dq;            // (B, Kd), overwritten
  float inv_scale;
  int B, T, Kd, Vd;
};

__global__ void __launch_bounds__(256)
attention_bwd_kernel(const AttBwdArgs a) {
  extern __shared__ float lds[];              // q[Kd] | dc[Vd] | w[T] | ds[T] | red[4]
  float* q = lds; float* dc = q + a.Kd; float* wt = dc + a.Vd; float* ds = wt + a.T; float* red = ds + a.T;
  const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  for (int k = tid; k < a.Kd; k += 256) q[k] = a.query[(size_t)b * a.ld_q + k];
  for (int v = tid; v < a.Vd; v += 256) dc[v] = a.d_ctx[(size_t)b * a.ld_dctx + v];
  for (int t = tid; t < a.T; t += 256) wt[t] = a.weights[(size_t)b * a.T + t];
  __syncthreads();
  for (int t = w; t < a.T; t += 4) {          // d a_t = <values[b,t], d_ctx>; d_values[b,t] += a_t d_ctx
    const size_t o = (size_t)t * a.v_t + (size_t)b * a.v_b;
    const float at = wt[t];
    float acc = 0.0f;
    for (int v = lane; v < a.Vd; v += 64) {
      acc = fmaf(a.values[o + v], dc[v], acc);
      a.d_values[o + v] += at * dc[v];
    }
    acc = wave_sum(acc);
    if (lane == 0) ds[t] = acc;
  }
  __syncthreads();
  float dot = 0.0f;
  for (int t = tid; t < a.T; t += 256) dot += wt[t] * ds[t];
  dot = block_sum(dot, red);
  __syncthreads();
  for (int t = tid; t < a.T; t += 256) ds[t] = wt[t] * (ds[t] - dot) * a.inv_scale;     // d score_t
  __syncthreads();
  for (int k = tid; k < a.Kd; k += 256) {
    float acc = 0.0f;
    const float qk = q[k];
    for (int t = 0; t < a.T; ++t) {
      const size_t o = (size_t)t * a.s_t + (size_t)b * a.s_b + k;
      acc = fmaf(ds[t], a.keys[o], acc);
      a.d_keys[o] += ds[t] * qk;
    }
    a.d_query[(size_t)b * a.ld_dq + k] = acc;
  }
}

// ------------------------------------------------------------------------------------------------------------
// One decoding step's score (models.py:541-545): out = log_softmax(logits); log p(y_u) = sum_v out_v * y_v (y one-hot
// in the reference, any float row here); logp_acc[b] += that (the reference's running sum, same order over steps).
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
logsoftmax_dot_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ y, long long ld_y,
                          float* __restrict__ logp_acc, float* __restrict__ lse_out, int V) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* lg = logits + (size_t)b * V;
  const float* yy = y + (size_t)b * ld_y;
  float m = -INFINITY;
  for (int v = tid; v < V; v += 256) m = fmaxf(m, lg[v]);
  m = block_max(m, red);
  float z = 0.0f;
  for (int v = tid; v < V; v += 256) z += expf(lg[v] - m);
  z = block_sum(z, red);
  const float lse = m + logf(z);
  float acc = 0.0f;
  for (int v = tid; v < V; v += 256) acc = fmaf(lg[v] - lse, yy[v], acc);
  acc = block_sum(acc, red);
  if (tid == 0) {
    logp_acc[b] += acc;
    if (lse_out) lse_out[b] = lse;
  }
}

// d_logits[b, v] = g[b] * (y_v - softmax_v * sum_v' y_v'),  g[b] = d loss / d log p(y_b) (same for every step)
__global__ void __launch_bounds__(256)
logsoftmax_dot_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ y, long long ld_y,
                          const float* __restrict__ lse, const float* __restrict__ g, long long g_stride,
                          float* __restrict__ d_logits, int V) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* lg = logits + (size_t)b * V;
  const float* yy = y + (size_t)b * ld_y;
  float sy = 0.0f;
  for (int v = tid; v < V; v += 256) sy += yy[v];
  sy = block_sum(sy, red);
  const float gb = g[(size_t)b * g_stride], l = lse[b];
  for (int v = tid; v < V; v += 256) d_logits[(size_t)b * V + v] = gb * (yy[v] - expf(lg[v] - l) * sy);
}

// loss = -mean_b logp[b] (models.py:825); d loss / d logp[b] = -g / B is produced by the same kernel when d_out given
__global__ void __launch_bounds__(256)
neg_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
  __shared__ float red[4];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) acc += x[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = -acc / (float)n;
}

__global__ void __launch_bounds__(256)
fill_scaled_kernel(float* __restrict__ dst, int n, const float* __restrict__ g, float scale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = g[0] * scale;
}

// dst[b, :] = src[:] for every row b (the decoder's initial state, models.py:521)
__global__ void __launch_bounds__(256)
broadcast_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long long ld_dst, int rows, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * n) return;
  const int b = i / n, j = i - b * n;
  dst[(size_t)b * ld_dst + j] = src[j];
}

}  // namespace slu

using namespace slu;

extern "C" int slu_gru_cell_fwd(const float* gi, const float* gh, const float* h_prev, int64_t ld_prev, float* h_out,
                                int64_t ld_out, float* save, float* drop_out, const float* mask, float p, uint64_t seed,
                                uint64_t offset, const uint64_t* offset_dev, uint64_t idx_base, int64_t B, int64_t H,
                                void* stream) {
  SLU_REQUIRE(gi && gh && h_prev && h_out, "slu_gru_cell_fwd: null pointer");
  SLU_REQUIRE(B > 0 && H > 0 && B * H < (1LL << 31), "slu_gru_cell_fwd: bad size");
  SLU_REQUIRE(p >= 0.0f && p < 1.0f, "slu_gru_cell_fwd: dropout p must be in [0,1)");
  CellArgs a;
  a.gi = gi; a.gh = gh; a.h_prev = h_prev; a.ld_prev = ld_prev; a.h_out = h_out; a.ld_out = ld_out; a.save = save;
  a.drop_out = drop_out; a.mask = mask; a.p = p; a.scale = 1.0f / (1.0f - p); a.seed = seed; a.offset = offset;
  a.offset_dev = (const unsigned long long*)offset_dev; a.idx_base = idx_base; a.B = (int)B; a.H = (int)H;
  hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3((unsigned)cdiv(B * H, 256)), dim3(256), 0, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("gru_cell_fwd_kernel");
  return SLU_OK;
}

extern "C" int slu_gru_cell_bwd(const float* d_h, int64_t ld_dh, const float* d_drop, const float* save,
                                const float* h_prev, int64_t ld_prev, float* d_gi, float* d_gh, float* d_h_prev,
                                int64_t ld_dprev, const float* mask, float p, uint64_t seed, uint64_t offset,
                                const uint64_t* offset_dev, uint64_t idx_base, int64_t B, int64_t H, void* stream) {
  SLU_REQUIRE(save && h_prev && d_gi && d_gh && d_h_prev, "slu_gru_cell_bwd: null pointer");
  SLU_REQUIRE(B > 0 && H > 0 && B * H < (1LL << 31), "slu_gru_cell_bwd: bad size");
  SLU_REQUIRE(p >= 0.0f && p < 1.0f, "slu_gru_cell_bwd: dropout p must be in [0,1)");
  CellBwdArgs a;
  a.d_h = d_h; a.ld_dh = ld_dh; a.d_drop = d_drop; a.save = save; a.h_prev = h_prev; a.ld_prev = ld_prev;
  a.d_gi = d_gi; a.d_gh = d_gh; a.d_h_prev = d_h_prev; a.ld_dprev = ld_dprev; a.mask = mask; a.p = p;
  a.scale = 1.0f / (1.0f - p); a.seed = seed; a.offset = offset; a.offset_dev = (const unsigned long long*)offset_dev;
  a.idx_base = idx_base; a.B = (int)B; a.H = (int)H;
  hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3((unsigned)cdiv(B * H, 256)), dim3(256), 0, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("gru_cell_bwd_kernel");
  return SLU_OK;
}

extern "C" int slu_attention_fwd(const float* keys, int64_t k_st, int64_t k_sb, const float* values, int64_t v_st,
                                 int64_t v_sb, const float* query, int64_t ld_q, float* ctx, int64_t ld_ctx,
                                 float* weights, float inv_scale, int64_t B, int64_t T, int64_t Kd, int64_t Vd,
                                 void* stream) {
  SLU_REQUIRE(keys && values && query && ctx && weights, "slu_attention_fwd: null pointer");
  SLU_REQUIRE(B > 0 && T > 0 && Kd > 0 && Vd > 0, "slu_attention_fwd: non-positive size");
  const size_t lds = (size_t)(Kd + T + 4) * sizeof(float);
  if (lds > 64 * 1024) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_attention_fwd: key_dim + T too large (%zu bytes of LDS)", lds);
  AttArgs a;
  a.keys = keys; a.values = values; a.s_t = k_st; a.s_b = k_sb; a.v_t = v_st; a.v_b = v_sb; a.query = query; a.ld_q = ld_q;
  a.ctx = ctx; a.ld_ctx = ld_ctx; a.weights = weights; a.inv_scale = inv_scale;
  a.B = (int)B; a.T = (int)T; a.Kd = (int)Kd; a.Vd = (int)Vd;
  hipLaunchKernelGGL(attention_fwd_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("attention_fwd_kernel");
  return SLU_OK;
}

extern "C" int slu_attention_bwd(const float* keys, int64_t k_st, int64_t k_sb, const float* values, int64_t v_st,
                                 int64_t v_sb, const float* query, int64_t ld_q, const float* d_ctx, int64_t ld_dctx,
                                 const float* weights, float* d_keys, float* d_values, float* d_query, int64_t ld_dq,
                                 float inv_scale, int64_t B, int64_t T, int64_t Kd, int64_t Vd, void* stream) {
  SLU_REQUIRE(keys && values && query && d_ctx && weights && d_keys && d_values && d_query, "slu_attention_bwd: null pointer");
  SLU_REQUIRE(B > 0 && T > 0 && Kd > 0 && Vd > 0, "slu_attention_bwd: non-positive size");
  const size_t lds = (size_t)(Kd + Vd + 2 * T + 4) * sizeof(float);
  if (lds > 64 * 1024) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_attention_bwd: key_dim + value_dim + 2 T too large (%zu bytes of LDS)", lds);
  AttBwdArgs a;
  a.keys = keys; a.values = values; a.s_t = k_st; a.s_b = k_sb; a.v_t = v_st; a.v_b = v_sb; a.query = query; a.ld_q = ld_q;
  a.d_ctx = d_ctx; a.ld_dctx = ld_dctx; a.weights = weights; a.d_keys = d_keys; a.d_values = d_values;
  a.d_query = d_query; a.ld_dq = ld_dq; a.inv_scale = inv_scale;
  a.B = (int)B; a.T = (int)T; a.Kd = (int)Kd; a.Vd = (int)Vd;
  hipLaunchKernelGGL(attention_bwd_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("attention_bwd_kernel");
  return SLU_OK;
}

extern "C" int slu_logsoftmax_dot_fwd(const float* logits, const float* y, int64_t ld_y, float* logp_acc, float* lse,
                                      int64_t B, int64_t V, void* stream) {
  SLU_REQUIRE(logits && y && logp_acc, "slu_logsoftmax_dot_fwd: null pointer");
  SLU_REQUIRE(B > 0 && V > 0, "slu_logsoftmax_dot_fwd: non-positive size");
  hipLaunchKernelGGL(logsoftmax_dot_fwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logits, y,
                     (long long)ld_y, logp_acc, lse, (int)V);
  SLU_CHECK_LAUNCH("logsoftmax_dot_fwd_kernel");
  return SLU_OK;
}

extern "C" int slu_logsoftmax_dot_bwd(const float* logits, const float* y, int64_t ld_y, const float* lse, const float* g,
                                      int64_t g_stride, float* d_logits, int64_t B, int64_t V, void* stream) {
  SLU_REQUIRE(logits && y && lse && g && d_logits, "slu_logsoftmax_dot_bwd: null pointer");
  SLU_REQUIRE(B > 0 && V > 0, "slu_logsoftmax_dot_bwd: non-positive size");
  hipLaunchKernelGGL(logsoftmax_dot_bwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logits, y,
                     (long long)ld_y, lse, g, (long long)g_stride, d_logits, (int)V);
  SLU_CHECK_LAUNCH("logsoftmax_dot_bwd_kernel");
  return SLU_OK;
}

extern "C" int slu_neg_mean_f32(const float* x, float* out, int64_t n, void* stream) {
  SLU_REQUIRE(x && out && n > 0 && n < (1LL << 31), "slu_neg_mean_f32: bad arguments");
  hipLaunchKernelGGL(neg_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, out, (int)n);
  SLU_CHECK_LAUNCH("neg_mean_kernel");
  return SLU_OK;
}

extern "C" int slu_fill_scaled_f32(float* dst, int64_t n, const float* g, float scale, void* stream) {
  SLU_REQUIRE(dst && g && n > 0 && n < (1LL << 31), "slu_fill_scaled_f32: bad arguments");
  hipLaunchKernelGGL(fill_scaled_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, (int)n, g, scale);
  SLU_CHECK_LAUNCH("fill_scaled_kernel");
  return SLU_OK;
}

extern "C" int slu_broadcast_rows_f32(const float* src, float* dst, int64_t ld_dst, int64_t rows, int64_t n, void* stream) {
  SLU_REQUIRE(src && dst && rows > 0 && n > 0 && rows * n < (1LL << 31), "slu_broadcast_rows_f32: bad arguments");
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3((unsigned)cdiv(rows * n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     (long long)ld_dst, (int)rows, (int)n);
  SLU_CHECK_LAUNCH("broadcast_rows_kernel");
  return SLU_OK;
}
