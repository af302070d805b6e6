// Probe: lane layout of v_mfma_f32_4x4x1_16b_f32 incl. the cbsz/abid A-broadcast, and
// v_permlane32_swap, on gfx950.  Build: hipcc --offload-arch=gfx950 -O2 mfma_4x4x1_layout.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out, unsigned* sw) {
  const int l = threadIdx.x;
  // A[l] = 1000*block + 10*(l%4)  ; B[l] = 1 + 0.001*l   -> D identifies who multiplied whom
  const float a = 100.0f * (l >> 2) + 10.0f * (l & 3) + 1.0f;
  const float b = 1.0f + 0.001f * l;
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 0, 0, 0);
  f32x4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 3, 5, 0);   // cbsz=3, abid=5
  for (int r = 0; r < 4; ++r) { out[(0 * 64 + l) * 4 + r] = d0[r]; out[(1 * 64 + l) * 4 + r] = d1[r]; }
  unsigned x = 1000 + l, y = 2000 + l;
  auto s = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  sw[l * 2] = s[0]; sw[l * 2 + 1] = s[1];
}

int main() {
  float* out; unsigned* sw;
  hipMalloc(&out, 2 * 64 * 4 * 4); hipMalloc(&sw, 64 * 2 * 4);
  probe<<<1, 64>>>(out, sw);
  float h[2 * 64 * 4]; unsigned hs[128];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hs, sw, sizeof(hs), hipMemcpyDeviceToHost);
  for (int v = 0; v < 2; ++v) {
    printf("variant %d (a = 100*blk + 10*(l%%4) + 1, b = 1 + .001*l)\n", v);
    for (int l : {0, 1, 2, 3, 4, 5, 21, 33, 63}) {
      printf(" lane %2d:", l);
      for (int r = 0; r < 4; ++r) printf(" %9.3f", h[(v * 64 + l) * 4 + r]);
      printf("\n");
    }
  }
  printf("permlane32_swap(x=1000+l, y=2000+l): lane0 -> (%u,%u) lane 5 -> (%u,%u) lane 32 -> (%u,%u) lane 40 -> (%u,%u)\n",
         hs[0], hs[1], hs[10], hs[11], hs[64], hs[65], hs[80], hs[81]);
  return 0;
}
