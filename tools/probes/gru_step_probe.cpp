// Probe: cycles per recurrence step of a GRU-like loop (LDS A-operand reads, 96 resident-B MFMAs per
// wave in 3 chains, short VALU tail, LDS write, barrier) for 8 waves on one CU, with features toggled.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gru_step_probe.cpp -o tools/probes/bin/gru_step_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

// MODE bit0: LDS reads each step, bit1: tail transcendental math, bit2: barrier, bit3: LDS write,
// bit4: 12 scalar gx loads + 4 scalar h stores per step (plain (T,B,C) layout),
// bit5: 3 dwordx4 gx loads + 1 dwordx4 h store per step (16-row tiled layout)
template <int MODE>
__global__ void __launch_bounds__(512) probe(float* out, unsigned long long* cyc, int steps, const float* gx, float* hout) {
  constexpr int H = 128, KQ = 32, LD = H + 2;
  __shared__ __attribute__((aligned(16))) float hbuf[2][16 * LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kg = lane >> 4;
  float wr[KQ], wz[KQ], wn[KQ];
  for (int k = 0; k < KQ; ++k) { wr[k] = 0.01f * (k + lane); wz[k] = 0.02f * (k - lane); wn[k] = 0.003f * (k * 3 + lane); }
  for (int x = tid; x < 2 * 16 * LD; x += 512) (&hbuf[0][0])[x] = 0.001f * x;
  float hprev[4] = {0.f, 0.f, 0.f, 0.f};
  float2 af[KQ / 2];
  for (int v = 0; v < KQ / 2; ++v) af[v] = make_float2(0.1f * v, 0.2f * v);
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
    const int cur = s & 1;
    float g12[12];
    if (MODE & 16) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < 3; ++g) g12[r * 3 + g] = gx[((size_t)(s + 1) * 64 + 4 * kg + r) * 768 + g * 128 + w * 16 + i];
    }
    if (MODE & 32) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float4 v4 = *reinterpret_cast<const float4*>(gx + (((size_t)(s + 1) * 768 + g * 128 + w * 16 + i) * 16 + 4 * kg));
        g12[g * 4] = v4.x; g12[g * 4 + 1] = v4.y; g12[g * 4 + 2] = v4.z; g12[g * 4 + 3] = v4.w;
      }
    }
    if (MODE & 1) {
      const float* hrow = &hbuf[cur][i * LD + kg * KQ];
#pragma unroll
      for (int v = 0; v < KQ / 2; ++v) af[v] = *reinterpret_cast<const float2*>(hrow + 2 * v);
    }
    f32x4 ar = {0, 0, 0, 0}, az = {0, 0, 0, 0}, an = {0, 0, 0, 0};
#pragma unroll
    for (int v = 0; v < KQ / 2; ++v) {
      ar = MF(af[v].x, wr[2 * v], ar); az = MF(af[v].x, wz[2 * v], az); an = MF(af[v].x, wn[2 * v], an);
      ar = MF(af[v].y, wr[2 * v + 1], ar); az = MF(af[v].y, wz[2 * v + 1], az); an = MF(af[v].y, wn[2 * v + 1], an);
    }
    float hn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE & 2) {
        const float rr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44f * ar[r]));
        const float zz = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44f * az[r]));
        const float nn = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88f * (an[r] * rr)));
        hn[r] = (1.0f - zz) * nn + zz * hprev[r];
      } else {
        hn[r] = (ar[r] + az[r] + an[r]) * 1e-3f;
      }
      if (MODE & 48) hn[r] += 1e-9f * (g12[r] + g12[4 + r] + g12[8 + r]);
      hprev[r] = hn[r];
    }
    if (MODE & 16) {
#pragma unroll
      for (int r = 0; r < 4; ++r) hout[((size_t)s * 64 + 4 * kg + r) * 256 + w * 16 + i] = hn[r];
    }
    if (MODE & 32) *reinterpret_cast<float4*>(hout + (((size_t)s * 256 + w * 16 + i) * 16 + 4 * kg)) = make_float4(hn[0], hn[1], hn[2], hn[3]);
    if (MODE & 8) {
#pragma unroll
      for (int r = 0; r < 4; ++r) hbuf[cur ^ 1][(4 * kg + r) * LD + w * 16 + i] = hn[r];
    }
    if (MODE & 4) __syncthreads();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[tid] = hprev[0] + hprev[1] + hprev[2] + hprev[3] + af[3].x;
  if (lane == 0) cyc[w] = t1 - t0;
}

template <int MODE>
void run() {
  float* out; unsigned long long* cyc; float* gx; float* hout;
  hipMalloc(&out, 4 * 512); hipMalloc(&cyc, 8 * 8);
  const int steps = 300;
  hipMalloc(&gx, sizeof(float) * (steps + 2) * 64 * 768); hipMemset(gx, 0, sizeof(float) * (steps + 2) * 64 * 768);
  hipMalloc(&hout, sizeof(float) * (steps + 2) * 64 * 256);
  probe<MODE><<<1, 512>>>(out, cyc, steps, gx, hout);
  hipDeviceSynchronize();
  probe<MODE><<<1, 512>>>(out, cyc, steps, gx, hout);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("mode %2d (lds-read %d, tail-math %d, barrier %d, lds-write %d, scalar-vmem %d, tiled-vmem %d): cycles/step per wave:", MODE, MODE & 1, (MODE >> 1) & 1,
         (MODE >> 2) & 1, (MODE >> 3) & 1, (MODE >> 4) & 1, (MODE >> 5) & 1);
  for (int w = 0; w < 8; ++w) printf(" %5.0f", (double)h[w] / steps);
  printf("\n");
  hipFree(out); hipFree(cyc); hipFree(gx); hipFree(hout);
}

int main() {
  run<15>(); run<15 + 16>(); run<15 + 32>();
  return 0;
}
