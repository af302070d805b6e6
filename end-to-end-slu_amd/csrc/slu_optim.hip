// Adam update for up to SLU_ADAM_MAX_TENSORS parameter tensors in ONE launch (reference:
// torch.optim.Adam(model.parameters(), lr) at training.py:19, defaults betas (0.9, 0.999), eps 1e-8,
// no weight decay / amsgrad).  The training step of the frozen-encoder configuration updates ten small
// tensors (1.2 MB): a launch per tensor, or torch's capturable multi-tensor kernel (45 us of per-element
// double-precision pow), costs more than the whole intent-GRU recurrence.  Here the tensor list
// travels by value in the kernel arguments (hipGraph-safe), the step count lives in device memory so a
// captured graph can be replayed, and the bias corrections are computed once per workgroup.
//   m <- m + (g - m)(1 - b1);  v <- b2 v + (1 - b2) g^2
//   p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)        (torch's formulation)
#include "slu_common.h"

#include <cmath>

namespace slu {

constexpr int ADAM_MAX_TENSORS = 32;
constexpr int ADAM_CHUNK = 1024;       // elements per workgroup (256 threads x 4)

struct AdamList {
  void* p[ADAM_MAX_TENSORS];
  const void* g[ADAM_MAX_TENSORS];
  void* m[ADAM_MAX_TENSORS];
  void* v[ADAM_MAX_TENSORS];
  long long n[ADAM_MAX_TENSORS];
  int chunk_end[ADAM_MAX_TENSORS];     // exclusive prefix of chunk counts
  int count;
};

template <typename T>
__global__ void __launch_bounds__(256)
adam_multi_kernel(const AdamList L, long long* step_dev, const double lr,
                  const double b1, const double b2, const double eps, const double grad_div,
                  unsigned int* ticket) {
  __shared__ float s_coef[2];
  const long long step_now = *step_dev;                    // read by every thread before the block takes its ticket
  if (threadIdx.x == 0) {
    const double t = (double)(step_now + 1);
    s_coef[0] = (float)(lr / (1.0 - pow(b1, t)));          // step size
    s_coef[1] = (float)sqrt(1.0 - pow(b2, t));             // sqrt of bias correction 2
  }
  int k = 0;
  while (k + 1 < L.count && (int)blockIdx.x >= L.chunk_end[k]) ++k;
  const int chunk = blockIdx.x - (k ? L.chunk_end[k - 1] : 0);
  const float fb1 = (float)b1, fb2 = (float)b2, feps = (float)eps, fdiv = (float)grad_div;
  T* __restrict__ p = reinterpret_cast<T*>(L.p[k]);
  const T* __restrict__ g = reinterpret_cast<const T*>(L.g[k]);
  T* __restrict__ m = reinterpret_cast<T*>(L.m[k]);
  T* __restrict__ v = reinterpret_cast<T*>(L.v[k]);
  const long long n = L.n[k];
  const long long base = (long long)chunk * ADAM_CHUNK;
  // the operands are requested BEFORE the barrier: their round trip passes while thread 0 evaluates the two double-precision
  // pow() of the bias corrections (round 6: the launch is a chain of latencies — counter, pow, operands, ticket — on the
  // latency-bound training stream; this takes one of them out)
  constexpr int U = ADAM_CHUNK / 256;
  T pg[U], pm[U], pv[U], pp[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long e = base + u * 256 + threadIdx.x;
    const long long ec = e < n ? e : n - 1;          // clamped: loaded unconditionally, used only when e < n
    pg[u] = g[ec]; pm[u] = m[ec]; pv[u] = v[ec]; pp[u] = p[ec];
  }
  __syncthreads();
  const float step_size = s_coef[0], bc2_sqrt = s_coef[1];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long e = base + u * 256 + threadIdx.x;
    if (e >= n) continue;
    if (sizeof(T) == 4) {
      float gg = (float)pg[u];
      if (fdiv != 1.0f) gg = gg / fdiv;            // data-parallel mean of the all-reduced sum
      float mm = (float)pm[u], vv = (float)pv[u];
      mm = mm + (gg - mm) * (1.0f - fb1);
      vv = fb2 * vv + (1.0f - fb2) * gg * gg;
      const float denom = sqrtf(vv) / bc2_sqrt + feps;
      p[e] = (T)((float)pp[u] - step_size * (mm / denom));
      m[e] = (T)mm; v[e] = (T)vv;
    } else {                       // float64 parameters (the Sinc band edges): double arithmetic
      double gg = (double)pg[u];
      if (grad_div != 1.0) gg = gg / grad_div;
      double mm = (double)pm[u], vv = (double)pv[u];
      mm = mm + (gg - mm) * (1.0 - b1);
      vv = b2 * vv + (1.0 - b2) * gg * gg;
      const double t = (double)(step_now + 1);
      const double denom = sqrt(vv) / sqrt(1.0 - pow(b2, t)) + eps;
      p[e] = (T)((double)pp[u] - (lr / (1.0 - pow(b1, t))) * (mm / denom));
      m[e] = (T)mm; v[e] = (T)vv;
    }
  }
  // Advance the step counter from inside the launch (saves the 1-thread increment kernel of every step): every
  // workgroup has read the counter before it takes a ticket, so the LAST one to finish may write it.
  if (ticket) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int done = atomicAdd(ticket, 1u);
      if (done == gridDim.x - 1) {
        *step_dev = step_now + 1;
        *ticket = 0u;                                      // ready for the next launch (stream order)
      }
    }
  }
}

__global__ void adam_step_inc_kernel(long long* step_dev, unsigned long long mask) {
  if ((mask >> threadIdx.x) & 1ull) step_dev[threadIdx.x] += 1;
}

}  // namespace slu

using namespace slu;

extern "C" int slu_adam_max_tensors(void) { return ADAM_MAX_TENSORS; }

// One Adam update of `count` tensors of one dtype (elem_bytes 4 = float32, 8 = float64).  The step
// counter (*step_dev, int64, number of updates done so far) is advanced by this launch when `ticket`
// (a zero-initialised device uint32 owned by the caller, left at zero again) is given — pass it with the
// LAST tensor list of an optimisation step that uses this counter — and left alone when ticket is null:
// then slu_adam_advance_step (adds 1 to the counters selected by a bit mask) does it.  Gradients are divided by grad_div first (the world size under data
// parallelism: the all-reduce delivers the sum), 1.0 = as they are.
extern "C" int slu_adam_multi(void* const* params, const void* const* grads, void* const* exp_avg,
                              void* const* exp_avg_sq, const int64_t* numel, int64_t count,
                              int elem_bytes, int64_t* step_dev, double lr, double beta1,
                              double beta2, double eps, double grad_div, uint32_t* ticket, void* stream) {
  SLU_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && step_dev, "slu_adam_multi: null pointer");
  SLU_REQUIRE(count > 0 && count <= ADAM_MAX_TENSORS, "slu_adam_multi: 1..%d tensors per call", ADAM_MAX_TENSORS);
  SLU_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "slu_adam_multi: elem_bytes must be 4 or 8");
  SLU_REQUIRE(grad_div > 0.0, "slu_adam_multi: grad_div must be positive");
  AdamList L;
  int chunks = 0;
  for (int k = 0; k < (int)count; ++k) {
    SLU_REQUIRE(params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k] && numel[k] > 0, "slu_adam_multi: bad tensor %d", k);
    L.p[k] = params[k]; L.g[k] = grads[k]; L.m[k] = exp_avg[k]; L.v[k] = exp_avg_sq[k]; L.n[k] = numel[k];
    chunks += (int)cdiv(numel[k], ADAM_CHUNK);
    L.chunk_end[k] = chunks;
  }
  L.count = (int)count;
  hipStream_t st = (hipStream_t)stream;
  if (elem_bytes == 4)
    hipLaunchKernelGGL(adam_multi_kernel<float>, dim3((unsigned)chunks), dim3(256), 0, st, L,
                       (long long*)step_dev, lr, beta1, beta2, eps, grad_div, (unsigned int*)ticket);
  else
    hipLaunchKernelGGL(adam_multi_kernel<double>, dim3((unsigned)chunks), dim3(256), 0, st, L,
                       (long long*)step_dev, lr, beta1, beta2, eps, grad_div, (unsigned int*)ticket);
  SLU_CHECK_LAUNCH("adam_multi_kernel");
  return SLU_OK;
}

// step_dev[i] += 1 for every bit i set in cohort_mask (i < 64): only the cohorts that were updated in
// this optimisation step advance, like torch.optim.Adam's per-parameter step counts.
extern "C" int slu_adam_advance_step(int64_t* step_dev, uint64_t cohort_mask, void* stream) {
  SLU_REQUIRE(step_dev, "slu_adam_advance_step: null pointer");
  if (cohort_mask == 0) return SLU_OK;
  hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)step_dev,
                     (unsigned long long)cohort_mask);
  SLU_CHECK_LAUNCH("adam_step_inc_kernel");
  return SLU_OK;
}
