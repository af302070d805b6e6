// Philox4x32-10 counter-based generator shared by every kernel that draws dropout masks (slu_pool.hip,
// slu_seq2seq.hip, the recurrence epilogue of slu_gru_bf16.hip): element `idx` of the stream (seed, offset) is word
// idx % 4 of the block with counter (idx / 4, offset) — a mask does not depend on which kernel or lane draws it.
#pragma once
#include "slu_common.h"

namespace slu {

// Philox4x32-10 (Salmon et al.), counter = (element index / 4, offset), key = seed.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  // one 32 x 32 -> 64 product each (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: the generator is multiply-bound
  const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}

__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint64_t idx) {
  uint32_t c[4] = {(uint32_t)(idx >> 2), (uint32_t)(idx >> 34), (uint32_t)offset, (uint32_t)(offset >> 32)};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(c, k);
  const uint32_t x = c[idx & 3];
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// the four words of the block holding elements 4 * blk .. 4 * blk + 3
__device__ __forceinline__ void philox_block(uint64_t seed, uint64_t offset, uint64_t blk, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(c, k);
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__device__ __forceinline__ float philox_to_uniform(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// dropout keep factors (scale or 0) of elements idx .. idx + 3 (idx % 4 == 0) of the stream (seed, offset): element e
// is kept iff its uniform draw is below thr = 1 - p.  The one formula every kernel that applies or re-derives a mask uses.
__device__ __forceinline__ float4 philox_keep4(uint64_t seed, uint64_t offset, uint64_t idx, float thr, float scale) {
  uint32_t w[4];
  philox_block(seed, offset, idx >> 2, w);
  float4 o;
  o.x = (philox_to_uniform(w[0]) < thr) ? scale : 0.0f;
  o.y = (philox_to_uniform(w[1]) < thr) ? scale : 0.0f;
  o.z = (philox_to_uniform(w[2]) < thr) ? scale : 0.0f;
  o.w = (philox_to_uniform(w[3]) < thr) ? scale : 0.0f;
  return o;
}

}  // namespace slu
