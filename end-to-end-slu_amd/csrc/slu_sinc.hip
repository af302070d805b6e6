// SincNet filterbank construction and its gradient (reference: models.py:79-106, sinc() :17-24).
//
// One workgroup per filter.  The bank is tiny (80 x 401), so the kernels favour fidelity to the
// reference's float32 operation order over speed: every product the reference forms in float32 is
// formed in float32 here in the same order; sin/cos are evaluated in double and rounded once
// (within 1 ulp of the CPU's vectorised sinf/cosf).
#include "slu_common.h"
#include <math.h>

namespace slu {

constexpr int SINC_THREADS = 256;

__device__ __forceinline__ float block_max(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float m = red[0];
  for (int i = 1; i < SINC_THREADS / 64; ++i) m = fmaxf(m, red[i]);
  __syncthreads();
  return m;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < SINC_THREADS / 64; ++i) s += red[i];
  __syncthreads();
  return s;
}

// linspace(0, N, N)[k] as torch computes it in float32 (symmetric two-sided formula).
__device__ __forceinline__ float linspace_0_N(int k, int N) {
  const float step = (float)N / (float)(N - 1);
  return (k < N / 2) ? (float)k * step : (float)N - (float)(N - 1 - k) * step;
}

__device__ __forceinline__ float hamming(int k, int N) {
  const float TWO_PI = 6.283185307179586f;
  const float a = TWO_PI * linspace_0_N(k, N) / (float)N;   // models.py:94
  return 0.54f - 0.46f * (float)cos((double)a);
}

// low_pass = 2 f * sinc(f * fs, t_right)  (models.py:99-100), value at tap k of N.
__device__ __forceinline__ float low_pass_tap(float f, float fs, int k, int half) {
  if (k == half) return 2.0f * f;                           // sinc centre = 1 (models.py:22)
  const int j = (k > half) ? (k - half) : (half - k);       // flip(): symmetric (models.py:19)
  const float TWO_PI = 6.283185307179586f;
  const float t = (float)j / fs;                            // models.py:82
  const float band = f * fs;
  const float arg = (TWO_PI * band) * t;                    // models.py:18
  const float y = (float)sin((double)arg) / arg;
  return (2.0f * f) * y;
}

__global__ void __launch_bounds__(SINC_THREADS)
sinc_filters_fwd_kernel(const double* __restrict__ b1, const double* __restrict__ band,
                        float* __restrict__ filters, int N, float fs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bp = reinterpret_cast<float*>(smem);
  __shared__ float red[SINC_THREADS / 64];
  const int i = blockIdx.x;
  const int half = (N - 1) / 2;
  const double c = 50.0 / (double)fs;
  const double beg64 = fabs(b1[i]) + c;                     // models.py:88
  const double end64 = beg64 + (fabs(band[i]) + c);         // models.py:89
  const float begf = (float)beg64, endf = (float)end64;
  float lmax = -INFINITY;
  for (int k = threadIdx.x; k < N; k += SINC_THREADS) {
    const float v = low_pass_tap(endf, fs, k, half) - low_pass_tap(begf, fs, k, half);
    bp[k] = v;
    lmax = fmaxf(lmax, v);
  }
  const float m = block_max(lmax, red);
  for (int k = threadIdx.x; k < N; k += SINC_THREADS)
    filters[(size_t)i * N + k] = (bp[k] / m) * hamming(k, N);   // models.py:103,106
}

__global__ void __launch_bounds__(SINC_THREADS)
sinc_filters_bwd_kernel(const double* __restrict__ b1, const double* __restrict__ band,
                        const float* __restrict__ dF, double* __restrict__ db1,
                        double* __restrict__ dband, int N, float fs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bp = reinterpret_cast<float*>(smem);
  __shared__ float red[SINC_THREADS / 64];
  __shared__ double dred[SINC_THREADS / 64];
  __shared__ int s_arg;
  const int i = blockIdx.x;
  const int half = (N - 1) / 2;
  const double c = 50.0 / (double)fs;
  const double beg64 = fabs(b1[i]) + c;
  const double end64 = beg64 + (fabs(band[i]) + c);
  const float begf = (float)beg64, endf = (float)end64;
  float lmax = -INFINITY;
  for (int k = threadIdx.x; k < N; k += SINC_THREADS) {
    const float v = low_pass_tap(endf, fs, k, half) - low_pass_tap(begf, fs, k, half);
    bp[k] = v;
    lmax = fmaxf(lmax, v);
  }
  if (threadIdx.x == 0) s_arg = N;
  const float m = block_max(lmax, red);
  // first index attaining the max (torch.max backward routes the gradient to one element)
  for (int k = threadIdx.x; k < N; k += SINC_THREADS)
    if (bp[k] == m) atomicMin(&s_arg, k);
  __syncthreads();
  const int amax = s_arg;
  // g[k] = dF[k] * w[k] is the gradient w.r.t. bp/m
  double dm_part = 0.0;
  for (int k = threadIdx.x; k < N; k += SINC_THREADS) {
    const double g = (double)dF[(size_t)i * N + k] * (double)hamming(k, N);
    dm_part += g * (double)bp[k];
  }
  const double md = (double)m;
  const double dm = -block_sum(dm_part, dred) / (md * md);
  // d(low_pass(f))[k] / df = 2 cos(2 pi f j), j = |k - half|  (closed form of the reference graph)
  const double TWO_PI = 6.283185307179586;
  double de = 0.0, dbg = 0.0;
  for (int k = threadIdx.x; k < N; k += SINC_THREADS) {
    double dbp = (double)dF[(size_t)i * N + k] * (double)hamming(k, N) / md;
    if (k == amax) dbp += dm;
    const int j = (k > half) ? (k - half) : (half - k);
    de += dbp * 2.0 * cos(TWO_PI * (double)endf * (double)j);
    dbg -= dbp * 2.0 * cos(TWO_PI * (double)begf * (double)j);
  }
  const double d_end = block_sum(de, dred);
  const double d_beg = block_sum(dbg, dred);
  if (threadIdx.x == 0) {
    const double v1 = b1[i], v2 = band[i];
    const double s1 = (v1 > 0.0) - (v1 < 0.0), s2 = (v2 > 0.0) - (v2 < 0.0);
    db1[i] = s1 * (d_beg + d_end);      // end = beg + |band| + c, beg = |b1| + c
    dband[i] = s2 * d_end;
  }
}

}  // namespace slu

extern "C" int slu_sinc_filters_fwd(const double* filt_b1, const double* filt_band, float* filters,
                                    int64_t n_filt, int64_t filt_dim, double fs, void* stream) {
  SLU_REQUIRE(filt_b1 && filt_band && filters, "slu_sinc_filters_fwd: null pointer");
  SLU_REQUIRE(n_filt > 0 && filt_dim >= 3 && (filt_dim & 1) && filt_dim <= 16384,
              "slu_sinc_filters_fwd: filt_dim must be odd, 3..16383 (got %lld)", (long long)filt_dim);
  hipLaunchKernelGGL(slu::sinc_filters_fwd_kernel, dim3((unsigned)n_filt), dim3(slu::SINC_THREADS),
                     (size_t)filt_dim * sizeof(float), (hipStream_t)stream, filt_b1, filt_band,
                     filters, (int)filt_dim, (float)fs);
  SLU_CHECK_LAUNCH("sinc_filters_fwd_kernel");
  return SLU_OK;
}

extern "C" int slu_sinc_filters_bwd(const double* filt_b1, const double* filt_band,
                                    const float* d_filters, double* d_filt_b1, double* d_filt_band,
                                    int64_t n_filt, int64_t filt_dim, double fs, void* stream) {
  SLU_REQUIRE(filt_b1 && filt_band && d_filters && d_filt_b1 && d_filt_band,
              "slu_sinc_filters_bwd: null pointer");
  SLU_REQUIRE(n_filt > 0 && filt_dim >= 3 && (filt_dim & 1) && filt_dim <= 16384,
              "slu_sinc_filters_bwd: filt_dim must be odd, 3..16383 (got %lld)", (long long)filt_dim);
  hipLaunchKernelGGL(slu::sinc_filters_bwd_kernel, dim3((unsigned)n_filt), dim3(slu::SINC_THREADS),
                     (size_t)filt_dim * sizeof(float), (hipStream_t)stream, filt_b1, filt_band,
                     d_filters, d_filt_b1, d_filt_band, (int)filt_dim, (float)fs);
  SLU_CHECK_LAUNCH("sinc_filters_bwd_kernel");
  return SLU_OK;
}
