// Does a SIMD of gfx950 overlap one wave's MFMA stream with ANOTHER wave's VALU stream?  (round 6: the premise of the two-tile
// recurrence.)  One workgroup of 8 waves per CU; per-wave role: 0 = MFMA only, 1 = VALU only, 2 = both, fine-grained
// interleaved in program order, 3 = transcendental only.  Roles are assigned by the wave's RANK ON ITS SIMD (HW_ID.SIMD_ID + LDS counter).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap_probe.cpp -o /tmp/overlap_probe && /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV>
__device__ __forceinline__ void body(f32x4 (&acc)[4], float (&v)[8], const bf16x8& a, const bf16x8& b, int iters) {
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (NM > 0) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[(k * NM + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[(k * NM + m) & 3], 0, 0, 0);
      }
      if constexpr (NV > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[(k * NV + j) & 7] = __builtin_fmaf(v[(k * NV + j) & 7], 1.0001f, 0.5f);
      }
    }
  }
}

__global__ void __launch_bounds__(512) probe(int mode, int iters, float* out, unsigned* ids, long long* cycles) {
  __shared__ unsigned cnt[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < 4) cnt[tid] = 0;
  __syncthreads();
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  const unsigned simd = (hw >> 4) & 3u;
  unsigned tk = 0;
  if (lane == 0) tk = atomicAdd(&cnt[simd], 1u);
  tk = __builtin_amdgcn_readfirstlane(tk);
  if (blockIdx.x == 0 && lane == 0) ids[w] = hw | (tk << 28);
  // role by mode: 0 all MFMA; 1 all VALU; 2 rank 0 MFMA / rank 1 VALU (per SIMD); 3 all interleaved (1 MFMA : 3 VALU);
  // 4 waves 0-3 MFMA, 4-7 VALU (by wave index); 5 rank 0 MFMA only (rank 1 idle); 6 rank 0 VALU only; 7 all trans; 8 rank0 MFMA / rank1 trans
  f32x4 acc[4];
  float v[8];
  for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < 8; ++k) v[k] = lane * 0.001f + k;
  bf16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.01f * (lane + k)); b[k] = (__bf16)(0.02f * (lane - k)); }
  int role = -1;
  if (mode == 0) role = 0;
  else if (mode == 1) role = 1;
  else if (mode == 2) role = (tk & 1) ? 1 : 0;
  else if (mode == 3) role = 2;
  else if (mode == 4) role = w < 4 ? 0 : 1;
  else if (mode == 5) role = (tk & 1) ? -1 : 0;
  else if (mode == 6) role = (tk & 1) ? -1 : 1;
  else if (mode == 7) role = 3;
  else if (mode == 8) role = (tk & 1) ? 3 : 0;
  __syncthreads();
  const long long c0 = __builtin_readcyclecounter();
  if (role == 0) body<3, 0>(acc, v, a, b, iters);            // 24 MFMA per iteration
  else if (role == 1) body<0, 9>(acc, v, a, b, iters);       // 72 VALU per iteration
  else if (role == 2) body<3, 9>(acc, v, a, b, iters);       // both, interleaved 3 : 9 per inner step
  else if (role == 3) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int k = 0; k < 24; ++k) v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7]);
  }
  const long long c1 = __builtin_readcyclecounter();
  __syncthreads();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1];
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 512 + tid] = s;
  if (blockIdx.x == 0 && lane == 0) cycles[w] = c1 - c0;
}

int main() {
  float* out; unsigned* ids; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ids, 64); hipMalloc(&cyc, 64);
  const int iters = 2000;
  const char* names[] = {"all 8 waves MFMA (24/iter)", "all 8 waves VALU fma (72/iter)", "per SIMD: rank 0 MFMA, rank 1 VALU", "all waves both, interleaved 3 MFMA : 9 VALU",
                         "waves 0-3 MFMA, 4-7 VALU", "rank 0 MFMA, rank 1 idle", "rank 0 VALU, rank 1 idle", "all 8 waves v_exp_f32 (24/iter)", "rank 0 MFMA, rank 1 v_exp"};
  for (int mode = 0; mode < 9; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, mode, 10, out, ids, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, mode, iters, out, ids, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned h[8]; long long c[8];
    hipMemcpy(h, ids, 32, hipMemcpyDeviceToHost); hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("mode %d (%s): %.1f us; per iteration %.1f ns; wave cycles(counter ticks)/iter:", mode, names[mode], ms * 1e3, ms * 1e6 / iters);
    for (int w = 0; w < 8; ++w) printf(" %.0f", (double)c[w] / iters);
    printf("\n   waves (simd, rank):");
    for (int w = 0; w < 8; ++w) printf(" w%d=(%u,%u)", w, (h[w] >> 4) & 3, h[w] >> 28);
    printf("\n");
  }
  return 0;
}
