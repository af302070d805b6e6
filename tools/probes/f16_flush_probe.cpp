// What does v_cvt_pk_f16_f32 return with MODE.FP_DENORM[7:6] = 0 (fp16 denormals flushed)?  Exhaustive over all 2^32 fp32
// bit patterns, against two software rules applied to the default-mode conversion:
//   rule A (post-rounding): round to fp16, then replace a DENORMAL result by a signed zero;
//   rule B (pre-rounding):  |x| < 2^-14 -> signed zero, else round to fp16.
// hipcc --offload-arch=gfx950 -O2 tools/probes/f16_flush_probe.cpp -o tools/probes/bin/f16_flush_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__global__ void probe(unsigned long long* counts, unsigned* examples) {
  __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);        // flush fp16 / fp64 denormals
  const unsigned long long n = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long a = 0, b = 0;
  for (unsigned long long v = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; v < (1ull << 32); v += n) {
    const unsigned bits = (unsigned)v;
    const float x = __uint_as_float(bits);
    if (x != x) continue;                                          // NaN payloads: not compared
    const f16x2 h = __builtin_convertvector(f32x2{x, x}, f16x2);
    const unsigned short hw = __builtin_bit_cast(unsigned, h) & 0xffffu;
    // default-mode conversion by integer arithmetic is not available here (the mode is global to the wave), so the
    // reference conversion is computed in fp32: |x| >= 2^-14 (1 - 2^-12) covers every value that rounds to a NORMAL fp16
    unsigned short rn;                                             // round-to-nearest-even result with denormals kept
    {
      const unsigned s = (bits >> 16) & 0x8000u;
      const unsigned e = (bits >> 23) & 0xff;
      const unsigned m = bits & 0x7fffffu;
      if (e == 0xff) rn = (unsigned short)(s | 0x7c00u);
      else if (e > 142) rn = (unsigned short)(s | 0x7c00u);        // >= 2^16: inf
      else if (e >= 113) {                                          // normal fp16 range (before rounding)
        unsigned r = ((e - 112) << 10) | (m >> 13);
        const unsigned rem = m & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
        rn = (unsigned short)(s | r);                               // may carry into the exponent / to inf
      } else if (e >= 102) {                                        // fp16 denormal range
        const unsigned mm = m | 0x800000u;
        const unsigned sh = 126 - e;                                // 14 .. 24
        unsigned r = mm >> sh;
        const unsigned rem = mm & ((1u << sh) - 1), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (r & 1))) ++r;
        rn = (unsigned short)(s | r);
      } else rn = (unsigned short)s;
    }
    const unsigned short ruleA = (rn & 0x7c00u) ? rn : (unsigned short)(rn & 0x8000u);
    const bool big = ((bits & 0x7fffffffu) >= 0x38800000u);        // |x| >= 2^-14
    const unsigned short ruleB = big ? rn : (unsigned short)(rn & 0x8000u);
    if (hw != ruleA) { if (a < 1) { examples[0] = bits; examples[1] = hw; examples[2] = ruleA; } ++a; }
    if (hw != ruleB) { if (b < 1) { examples[3] = bits; examples[4] = hw; examples[5] = ruleB; } ++b; }
  }
  atomicAdd(&counts[0], a);
  atomicAdd(&counts[1], b);
}

int main() {
  unsigned long long* counts; unsigned* ex;
  hipMalloc(&counts, 16); hipMalloc(&ex, 32); hipMemset(counts, 0, 16); hipMemset(ex, 0, 32);
  hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, counts, ex);
  unsigned long long h[2]; unsigned e[8];
  hipMemcpy(h, counts, 16, hipMemcpyDeviceToHost); hipMemcpy(e, ex, 32, hipMemcpyDeviceToHost);
  printf("v_cvt_pk_f16_f32 under fp16-denormal flush vs rule A (flush after rounding): %llu mismatches (e.g. x bits %08x: hw %04x, rule %04x)\n", h[0], e[0], e[1], e[2]);
  printf("                                              vs rule B (flush before rounding): %llu mismatches (e.g. x bits %08x: hw %04x, rule %04x)\n", h[1], e[3], e[4], e[5]);
  return 0;
}
