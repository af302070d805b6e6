// Frame-level softmax cross-entropy with an ignore index, for the two ASR pre-training heads of
// PretrainedModel.forward (reference models.py:291-331): logits (N = B*T' rows, V classes; the
// phoneme_linear / word_linear output, V = 42 / 10 000) against integer targets where -1 marks frames
// without a label (F.cross_entropy(..., ignore_index=-1), mean over the labelled frames) and the
// accuracy over the labelled frames (logits.max(1)[1] == y).
//
// Three launches, no host synchronisation:
//   frame_ce_count   : n_kept = #{y != ignore}                       (1 workgroup)
//   frame_ce_rows    : one workgroup per row: max / first arg-max, log-sum-exp, row loss, row hit;
//                      optionally overwrites the row IN PLACE with d loss / d logits
//                      = (softmax - onehot) / n_kept for labelled rows, 0 otherwise
//   frame_ce_reduce  : deterministic sums -> out[0] = loss, out[1] = accuracy, out[2] = n_kept
// HBM-bound: logits are read twice (second pass from L2 for V <= 10 000: 40 KB per row) and written
// once.  The linear layers themselves are slu_gemm_f32 calls.
#include "slu_common.h"

namespace slu {

constexpr int FCE_THREADS = 256;

__global__ void __launch_bounds__(FCE_THREADS)
frame_ce_count_kernel(const long long* __restrict__ y, long long n, long long ignore, float* __restrict__ out) {
  __shared__ int part[FCE_THREADS];
  int c = 0;
  for (long long i = threadIdx.x; i < n; i += FCE_THREADS) c += (y[i] != ignore) ? 1 : 0;
  part[threadIdx.x] = c;
  __syncthreads();
  for (int s = FCE_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[2] = (float)part[0];
}

__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__global__ void __launch_bounds__(FCE_THREADS)
frame_ce_rows_kernel(float* __restrict__ logits, const long long* __restrict__ y, int V, long long ignore,
                     const float* __restrict__ out, float* __restrict__ row_stats, int write_grad) {
  __shared__ float s_val[FCE_THREADS];
  __shared__ int s_idx[FCE_THREADS];
  const long long row = blockIdx.x;
  float* __restrict__ lr = logits + row * (long long)V;
  const long long target = y[row];
  const int tid = threadIdx.x;
  // pass 1: max and first arg-max
  float mv = -INFINITY;
  int mi = 0x7fffffff;
  for (int v = tid; v < V; v += FCE_THREADS) {
    const float x = lr[v];
    if (x > mv) { mv = x; mi = v; }
  }
  s_val[tid] = mv; s_idx[tid] = mi;
  __syncthreads();
  for (int s = FCE_THREADS / 2; s > 0; s >>= 1) {
    if (tid < s) {
      float a = s_val[tid]; int ai = s_idx[tid];
      argmax_combine(a, ai, s_val[tid + s], s_idx[tid + s]);
      s_val[tid] = a; s_idx[tid] = ai;
    }
    __syncthreads();
  }
  const float row_max = s_val[0];
  const int row_arg = s_idx[0];
  __syncthreads();
  // pass 2: sum of exp
  float se = 0.0f;
  for (int v = tid; v < V; v += FCE_THREADS) se += __expf(lr[v] - row_max);
  s_val[tid] = se;
  __syncthreads();
  for (int s = FCE_THREADS / 2; s > 0; s >>= 1) {
    if (tid < s) s_val[tid] += s_val[tid + s];
    __syncthreads();
  }
  const float sum_exp = s_val[0];
  const bool kept = target != ignore;
  if (tid == 0) {
    const float lse = row_max + __logf(sum_exp);
    row_stats[2 * row + 0] = kept ? (lse - lr[target]) : 0.0f;
    row_stats[2 * row + 1] = (kept && row_arg == (int)target) ? 1.0f : 0.0f;
  }
  if (write_grad) {
    __syncthreads();                     // lr[target] has been read
    const float scale = kept ? 1.0f / out[2] : 0.0f;
    const float inv = 1.0f / sum_exp;
    for (int v = tid; v < V; v += FCE_THREADS) {
      const float p = __expf(lr[v] - row_max) * inv;
      lr[v] = kept ? (p - ((long long)v == target ? 1.0f : 0.0f)) * scale : 0.0f;
    }
  }
}

__global__ void __launch_bounds__(FCE_THREADS)
frame_ce_reduce_kernel(const float* __restrict__ row_stats, long long n, float* __restrict__ out) {
  __shared__ float sl[FCE_THREADS], sa[FCE_THREADS];
  float l = 0.0f, a = 0.0f;
  for (long long i = threadIdx.x; i < n; i += FCE_THREADS) { l += row_stats[2 * i]; a += row_stats[2 * i + 1]; }
  sl[threadIdx.x] = l; sa[threadIdx.x] = a;
  __syncthreads();
  for (int s = FCE_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sl[threadIdx.x] += sl[threadIdx.x + s]; sa[threadIdx.x] += sa[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float nk = out[2];
    out[0] = sl[0] / nk;                  // 0/0 = NaN when no frame is labelled, like F.cross_entropy
    out[1] = sa[0] / nk;
  }
}

}  // namespace slu

using namespace slu;

extern "C" int slu_frame_ce_fwd(float* logits, const int64_t* y, int64_t N, int64_t V, int64_t ignore_index,
                                int write_grad, float* row_stats, float* out3, void* stream) {
  SLU_REQUIRE(logits && y && row_stats && out3, "slu_frame_ce_fwd: null pointer");
  SLU_REQUIRE(N > 0 && V > 0 && V < (1LL << 31) && N < (1LL << 31), "slu_frame_ce_fwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(frame_ce_count_kernel, dim3(1), dim3(FCE_THREADS), 0, st, (const long long*)y,
                     (long long)N, (long long)ignore_index, out3);
  SLU_CHECK_LAUNCH("frame_ce_count_kernel");
  hipLaunchKernelGGL(frame_ce_rows_kernel, dim3((unsigned)N), dim3(FCE_THREADS), 0, st, logits,
                     (const long long*)y, (int)V, (long long)ignore_index, (const float*)out3, row_stats,
                     write_grad);
  SLU_CHECK_LAUNCH("frame_ce_rows_kernel");
  hipLaunchKernelGGL(frame_ce_reduce_kernel, dim3(1), dim3(FCE_THREADS), 0, st, (const float*)row_stats,
                     (long long)N, out3);
  SLU_CHECK_LAUNCH("frame_ce_reduce_kernel");
  return SLU_OK;
}
