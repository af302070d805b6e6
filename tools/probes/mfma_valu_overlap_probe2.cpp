// Which VALU instruction classes of gfx950 co-execute with ANOTHER wave's MFMA stream on the same SIMD (and with the same
// wave's)?  One workgroup of 8 waves per CU; waves w and w + 4 share a SIMD.  Wave roles: M = 24 independent
// v_mfma_f32_16x16x32_bf16 per iteration; V<kind> = 72 instructions of one kind per iteration (8 independent registers).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap_probe2.cpp -o tools/probes/bin/overlap_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define V9(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

template <int KIND>
__device__ __forceinline__ void valu_iter(float (&v)[8], float2 (&pv)[8]) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define PKF(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[i]) : "v"(pv[(i + 1) & 7]));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define LSH(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[i]));
#define PRM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
  if constexpr (KIND == 0) { V9(FMA) }
  else if constexpr (KIND == 1) { V9(MUL) }
  else if constexpr (KIND == 2) { V9(PKF) }
  else if constexpr (KIND == 3) { V9(EXP) }
  else if constexpr (KIND == 4) { V9(RCP) }
  else if constexpr (KIND == 5) { V9(CVT) }
  else if constexpr (KIND == 6) { V9(AND) }
  else if constexpr (KIND == 7) { V9(LSH) }
  else if constexpr (KIND == 8) { V9(PRM) }
  else if constexpr (KIND == 9) { V9(CND) }
  else if constexpr (KIND == 10) { V9(MOV) }
  else if constexpr (KIND == 11) { V9(SUB) }
}
static const char* KNAME[] = {"v_fma_f32", "v_mul_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_cvt_pk_bf16_f32", "v_and_b32", "v_lshlrev_b32",
                              "v_perm_b32", "v_cndmask_b32", "v_mov_b32", "v_sub_f32"};
constexpr int NKIND = 12;

__device__ __forceinline__ void mfma_iter(f32x4 (&acc)[4], const bf16x8& a, const bf16x8& b) {
#pragma unroll
  for (int k = 0; k < 24; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 3], 0, 0, 0);
}

// mode 0: rank 0 runs the VALU kind alone (rank 1 idle); 1: rank 0 MFMA, rank 1 VALU kind; 2: both ranks VALU kind;
// 3: every wave 1 MFMA : 3 VALU in program order (24 MFMA + 72 VALU per iteration)
template <int KIND>
__global__ void __launch_bounds__(512) probe(int mode, int iters, float* out, long long* cycles) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int rank = w >> 2;
  f32x4 acc[4];
  float v[8];
  float2 pv[8];
  for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < 8; ++k) { v[k] = 1.0f + lane * 1e-6f + k * 1e-7f; pv[k] = make_float2(v[k], v[k]); }
  bf16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.01f * (lane + k)); b[k] = (__bf16)(0.02f * (lane - k)); }
  __syncthreads();
  const long long c0 = __builtin_readcyclecounter();
  if (mode == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 24; ++k) {
        acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KIND == 0) { FMA(0) FMA(1) FMA(2) }
        else if constexpr (KIND == 2) { PKF(0) PKF(1) PKF(2) }
        else if constexpr (KIND == 3) { EXP(0) EXP(1) EXP(2) }
        else if constexpr (KIND == 5) { CVT(0) CVT(1) CVT(2) }
        else if constexpr (KIND == 6) { AND(0) AND(1) AND(2) }
        else { MUL(0) MUL(1) MUL(2) }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (rank == 0 && mode == 1) {
    for (int it = 0; it < iters; ++it) mfma_iter(acc, a, b);
  } else if (rank == 0 || mode != 0) {
    for (int it = 0; it < iters; ++it) valu_iter<KIND>(v, pv);
  }
  const long long c1 = __builtin_readcyclecounter();
  __syncthreads();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1];
  for (int k = 0; k < 8; ++k) s += v[k] + pv[k].x + pv[k].y;
  out[blockIdx.x * 512 + tid] = s;
  if (blockIdx.x == 0 && lane == 0) cycles[w] = c1 - c0;
}

template <int KIND>
void run(float* out, long long* cyc, double mfma_ns) {
  const int iters = 2000;
  double t[4];
  long long cw[4][8];
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, mode, 10, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    t[mode] = ms * 1e6 / iters;
    (void)hipMemcpy(cw[mode], cyc, 64, hipMemcpyDeviceToHost);
  }
  printf("%-18s alone (1 wave/SIMD, 72 instr) %6.1f ns | 2 waves/SIMD %6.1f ns | beside MFMA wave (24 MFMA = %.1f ns alone) %6.1f ns -> overlap %.2f "
         "(1 = hidden, 0 = serial) | in-wave 1 MFMA : 3 VALU, 2 waves/SIMD %6.1f ns (MFMA alone x2 = %.1f)\n",
         KNAME[KIND], t[0], t[2], mfma_ns, t[1], (mfma_ns + t[0] - t[1]) / (t[0] < mfma_ns ? t[0] : mfma_ns), t[3], 2 * mfma_ns);
}

__global__ void __launch_bounds__(512) mfma_only(int iters, float* out) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  f32x4 acc[4];
  for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.01f * (lane + k)); b[k] = (__bf16)(0.02f * (lane - k)); }
  if (w < 4) for (int it = 0; it < iters; ++it) mfma_iter(acc, a, b);
  out[blockIdx.x * 512 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_only, dim3(256), dim3(512), 0, 0, 10, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_only, dim3(256), dim3(512), 0, 0, 2000, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma_ns = ms * 1e6 / 2000;
  printf("MFMA alone: 24 x v_mfma_f32_16x16x32_bf16 per iteration, one wave per SIMD: %.1f ns per iteration\n", mfma_ns);
  run<0>(out, cyc, mfma_ns); run<1>(out, cyc, mfma_ns); run<2>(out, cyc, mfma_ns); run<3>(out, cyc, mfma_ns); run<4>(out, cyc, mfma_ns);
  run<5>(out, cyc, mfma_ns); run<6>(out, cyc, mfma_ns); run<7>(out, cyc, mfma_ns); run<8>(out, cyc, mfma_ns); run<9>(out, cyc, mfma_ns);
  run<10>(out, cyc, mfma_ns); run<11>(out, cyc, mfma_ns);
  return 0;
}
