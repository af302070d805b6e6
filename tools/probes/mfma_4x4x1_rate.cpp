// Probe: issue rate of v_mfma_f32_4x4x1_16b_f32 with the cbsz/abid broadcast, as used by
// gru_seq_fwd4_kernel: 192 MFMAs per "step" in NCH independent accumulator chains, one wave per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_4x4x1_rate.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCH, int CBSZ>
__global__ void __launch_bounds__(256) probe(float* out, long long* cyc, int iters) {
  float w[64], a[8];
  for (int k = 0; k < 64; ++k) w[k] = 0.001f * (threadIdx.x + k);
  for (int k = 0; k < 8; ++k) a[k] = 0.01f * (threadIdx.x - k);
  f32x4 acc[NCH];
  for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 3; ++g)            // three gate tiles x 64 k
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#define STEP(ab) acc[(g * 64 + q * 8 + ab) % NCH] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[q], w[q * 8 + ab], acc[(g * 64 + q * 8 + ab) % NCH], CBSZ, (CBSZ ? ab : 0), 0);
        STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
#undef STEP
      }
    asm volatile("" ::: "memory");
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NCH, int CBSZ>
void run(const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  probe<NCH, CBSZ><<<256, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); probe<NCH, CBSZ><<<256, 256>>>(out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-28s: %7.1f ns per 192-MFMA step  (%.2f ns per MFMA; s_memtime ticks/step %.1f)\n", name,
         ms * 1e6 / iters, ms * 1e6 / iters / 192, (double)h[0] / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<3, 3>("3 chains, cbsz=3");
  run<6, 3>("6 chains, cbsz=3");
  run<12, 3>("12 chains, cbsz=3");
  run<3, 0>("3 chains, no broadcast");
  run<6, 0>("6 chains, no broadcast");
  return 0;
}
