// Probe: sustained issue interval of v_mfma_f32_16x16x4_f32 on gfx950 as a function of the number of
// independent accumulator chains and of the number of waves per SIMD (tuning aid, not product code).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_rate.cpp -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void probe(float* out, unsigned long long* cyc, int iters, float a0, float b0) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CH>
void run(int threads, int blocks) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks * (threads / 64));
  const int iters = 200;
  probe<CH><<<blocks, threads>>>(out, cyc, iters, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<CH><<<blocks, threads>>>(out, cyc, iters, 1.0f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[64];
  hipMemcpy(h, cyc, sizeof(unsigned long long) * (threads / 64), hipMemcpyDeviceToHost);
  const double n = (double)iters * 8 * CH;            // MFMAs per wave
  const int waves_per_simd = threads / 256;
  printf("chains=%d threads=%4d blocks=%3d: wave0 %.1f ticks/MFMA(per wave) -> %.1f ticks per MFMA per SIMD; kernel %.1f us\n",
         CH, threads, blocks, h[0] / n, h[0] / n / (waves_per_simd > 0 ? waves_per_simd : 1), ms * 1e3);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run<1>(threads, 1); run<2>(threads, 1); run<3>(threads, 1); run<4>(threads, 1); run<6>(threads, 1);
  }
  run<3>(512, 8); run<3>(512, 256); run<4>(256, 256);
  return 0;
}
