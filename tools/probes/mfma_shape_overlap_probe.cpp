// Does the MFMA SHAPE change how much of another wave's VALU stream a gfx950 SIMD hides?  Same set-up as
// mfma_valu_overlap_probe2.cpp (256 workgroups x 8 waves; waves w and w + 4 share a SIMD; rank 0 = MFMA stream, rank 1 = 72
// VALU instructions of one kind per iteration), with the MFMA stream issued as
//   shape 0: 24 x v_mfma_f32_16x16x32_bf16 (4 passes each)     shape 1: 12 x v_mfma_f32_32x32x16_bf16 (8 passes each)
//   shape 2: 24 x v_mfma_f32_16x16x16_bf16? (not on gfx950 - omitted)   shape 3: 6 x v_mfma_f32_32x32x16_bf16 + idle (half rate)
// i.e. the same flops per iteration for shapes 0 and 1.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_shape_overlap_probe.cpp -o tools/probes/bin/shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define V9(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
static const char* KNAME[] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "v_and_b32"};

template <int KIND>
__device__ __forceinline__ void valu_iter(float (&v)[8]) {
  if constexpr (KIND == 0) { V9(FMA) }
  else if constexpr (KIND == 1) { V9(EXP) }
  else if constexpr (KIND == 2) { V9(CVT) }
  else { V9(AND) }
}

// mode 0: rank 0 MFMA alone; 1: rank 1 VALU alone; 2: rank 0 MFMA beside rank 1 VALU; 3: both ranks MFMA
template <int KIND, int SHAPE>
__global__ void __launch_bounds__(512) probe(int mode, int iters, float* out) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int rank = w >> 2;
  f32x4 acc[4];
  f32x16 big[2];
  float v[8];
  for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < 2; ++k) for (int j = 0; j < 16; ++j) big[k][j] = 0.f;
  for (int k = 0; k < 8; ++k) v[k] = 1.0f + lane * 1e-6f + k * 1e-7f;
  bf16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.01f * (lane + k)); b[k] = (__bf16)(0.02f * (lane - k)); }
  __syncthreads();
  const bool do_m = (rank == 0 && (mode == 0 || mode == 2)) || mode == 3;
  const bool do_v = rank == 1 && (mode == 1 || mode == 2);
  if (do_m) {
    for (int it = 0; it < iters; ++it) {
      if constexpr (SHAPE == 0) {
#pragma unroll
        for (int k = 0; k < 24; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 3], 0, 0, 0);
      } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) big[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[k & 1], 0, 0, 0);
      }
    }
  } else if (do_v) {
    for (int it = 0; it < iters; ++it) valu_iter<KIND>(v);
  }
  __syncthreads();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1];
  for (int k = 0; k < 2; ++k) s += big[k][0] + big[k][5];
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 512 + tid] = s;
}

template <int KIND, int SHAPE>
void run(float* out) {
  const int iters = 2000;
  double t[4];
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND, SHAPE>), dim3(256), dim3(512), 0, 0, mode, 10, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, SHAPE>), dim3(256), dim3(512), 0, 0, mode, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    t[mode] = ms * 1e6 / iters;
  }
  printf("%-18s %s: MFMA alone %6.1f ns | 72 VALU alone %6.1f ns | side by side %6.1f ns -> hidden %.2f of the shorter | MFMA on both waves %6.1f ns\n",
         KNAME[KIND], SHAPE == 0 ? "24 x 16x16x32" : "12 x 32x32x16", t[0], t[1], t[2], (t[0] + t[1] - t[2]) / (t[0] < t[1] ? t[0] : t[1]), t[3]);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  run<0, 0>(out); run<0, 1>(out);
  run<1, 0>(out); run<1, 1>(out);
  run<2, 0>(out); run<2, 1>(out);
  run<3, 0>(out); run<3, 1>(out);
  return 0;
}
