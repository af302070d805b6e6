// Shared host/device helpers for libslu_hip.so (gfx950 only — no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/slu_hip.h"

namespace slu {

void set_error(const char* fmt, ...);

#define SLU_FAIL(code, ...)            \
  do {                                 \
    ::slu::set_error(__VA_ARGS__);     \
    return (code);                     \
  } while (0)

#define SLU_REQUIRE(cond, ...)                          \
  do {                                                  \
    if (!(cond)) SLU_FAIL(SLU_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

// Launch check that does not synchronise: hipGetLastError only reports launch-time failures.
#define SLU_CHECK_LAUNCH(name)                                                        \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// two-stage deterministic column sum (slu_gemm.hip); ws must hold colsum_splits(M) * N floats
int colsum_splits(int64_t M);
int colsum_two_stage(const float* X, int64_t rs, float* out, int64_t M, int64_t N, float* ws,
                     hipStream_t st);

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D(16x16) += A(16x4) * B(4x16), exact fp32 (an fmaf chain over k).
// Lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; it receives
// D[row = 4 * (l >> 4) + reg][col = l & 15] in reg = 0..3.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace slu
