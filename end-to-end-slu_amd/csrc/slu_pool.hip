// Fused Dropout + Downsample for time-major GRU outputs (reference: nn.Dropout models.py:246,276,700
// followed by Downsample models.py:26-46: "none" = strided slice, "avg"/"max" = pool1d with
// ceil_mode=True, where a partial last window uses only the frames that exist).
//
// Pure HBM-bound elementwise work.  Main path (C % 4 == 0): one thread per four consecutive channels
// (float4 accesses, a wave covers a contiguous 1 KB row segment), 2-D grid (time on y) so that all
// index arithmetic is 32-bit, and ONE Philox4x32-10 evaluation per four elements (its four output
// words are exactly the four channels).  A scalar kernel covers other channel counts.
#include "slu_bf16.h"
#include "slu_philox.h"

namespace slu {

struct PoolParams {
  const float* mask; long long m_st, m_sb;
  const unsigned* keep_bits;              // optional 1-bit mask (slu_dropout_bits): word (t*B + b) * C/32 + c/32, bit c % 32
  float p, scale;
  unsigned long long seed, offset;
  const unsigned long long* offset_dev;   // optional run-time addend (hipGraph replays)
  unsigned long long sub_stride;          // offset increment between sub-batches
  int sub_batch;                          // 0: the whole batch is one Philox index space
  int method, factor;
  int T, B, C, T_out;
};

// keep-factor (0 or 1/(1-p)) of element (t,b,c)
__device__ __forceinline__ float keep_scale(const PoolParams& q, int t, int b, int c) {
  if (q.p <= 0.0f) return 1.0f;
  if (q.mask) return q.mask[(long long)t * q.m_st + (long long)b * q.m_sb + c] * q.scale;
  if (q.keep_bits) return ((q.keep_bits[((size_t)t * q.B + b) * (q.C >> 5) + (c >> 5)] >> (c & 31)) & 1u) ? q.scale : 0.0f;
  uint64_t off = q.offset + (q.offset_dev ? *q.offset_dev : 0ull);
  uint64_t idx;
  if (q.sub_batch > 0) {
    const int k = b / q.sub_batch, bl = b - k * q.sub_batch;
    idx = ((uint64_t)t * q.sub_batch + bl) * q.C + c;
    off += (uint64_t)k * q.sub_stride;
  } else {
    idx = ((uint64_t)t * q.B + b) * q.C + c;
  }
  return philox_uniform(q.seed, off, idx) < (1.0f - q.p) ? q.scale : 0.0f;
}

__global__ void __launch_bounds__(256)
dropout_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, const PoolParams q) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)q.T_out * q.B * q.C;
  if (idx >= total) return;
  const int c = (int)(idx % q.C);
  const long long tb = idx / q.C;
  const int b = (int)(tb % q.B);
  const int to = (int)(tb / q.B);
  const int t0 = to * q.factor;
  const long long row = (long long)q.B * q.C;
  if (q.method == 0) {
    y[idx] = x[(long long)t0 * row + (long long)b * q.C + c] * keep_scale(q, t0, b, c);
    return;
  }
  const int t1 = min(q.T, t0 + q.factor);
  float acc = (q.method == 1) ? 0.0f : -INFINITY;
  for (int t = t0; t < t1; ++t) {
    const float v = x[(long long)t * row + (long long)b * q.C + c] * keep_scale(q, t, b, c);
    acc = (q.method == 1) ? acc + v : fmaxf(acc, v);
  }
  y[idx] = (q.method == 1) ? acc / (float)(t1 - t0) : acc;
}

__global__ void __launch_bounds__(256)
dropout_pool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                        float* __restrict__ dx, const PoolParams q) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)q.T * q.B * q.C;
  if (idx >= total) return;
  const int c = (int)(idx % q.C);
  const long long tb = idx / q.C;
  const int b = (int)(tb % q.B);
  const int t = (int)(tb / q.B);
  const int to = t / q.factor;
  const int t0 = to * q.factor;
  const long long row = (long long)q.B * q.C;
  const float g = dy[(long long)to * row + (long long)b * q.C + c];
  const float ks = keep_scale(q, t, b, c);
  float out;
  if (q.method == 0) {
    out = (t == t0) ? g * ks : 0.0f;
  } else if (q.method == 1) {
    const int t1 = min(q.T, t0 + q.factor);
    out = g * ks / (float)(t1 - t0);
  } else {
    const int t1 = min(q.T, t0 + q.factor);
    int arg = t0;
    float best = -INFINITY;
    for (int tt = t0; tt < t1; ++tt) {
      const float v = x[(long long)tt * row + (long long)b * q.C + c] * keep_scale(q, tt, b, c);
      if (v > best) { best = v; arg = tt; }
    }
    out = (arg == t) ? g * ks : 0.0f;
  }
  dx[idx] = out;
}

// ---- vector path: four consecutive channels per thread -----------------------------------------
__device__ __forceinline__ float4 keep_scale4(const PoolParams& q, uint64_t off_base, int t, int b, int c) {
  if (q.p <= 0.0f) return make_float4(1.f, 1.f, 1.f, 1.f);
  if (q.mask) {
    const float* m = q.mask + (long long)t * q.m_st + (long long)b * q.m_sb + c;
    return make_float4(m[0] * q.scale, m[1] * q.scale, m[2] * q.scale, m[3] * q.scale);
  }
  if (q.keep_bits) {          // c % 4 == 0: the four bits sit in one word
    const unsigned m = q.keep_bits[((size_t)t * q.B + b) * (q.C >> 5) + (c >> 5)] >> (c & 31);
    return make_float4((m & 1u) ? q.scale : 0.0f, (m & 2u) ? q.scale : 0.0f, (m & 4u) ? q.scale : 0.0f, (m & 8u) ? q.scale : 0.0f);
  }
  uint64_t off = off_base;
  uint64_t idx;
  if (q.sub_batch > 0) {
    const int k = b / q.sub_batch, bl = b - k * q.sub_batch;
    idx = ((uint64_t)t * q.sub_batch + bl) * q.C + c;
    off += (uint64_t)k * q.sub_stride;
  } else {
    idx = ((uint64_t)t * q.B + b) * q.C + c;
  }
  // idx % 4 == 0 (C % 4 == 0, c % 4 == 0): the four words of one Philox block are elements idx .. idx+3
  return philox_keep4(q.seed, off, idx, 1.0f - q.p, q.scale);
}

// (explicitly rounded products and sums: no fma contraction, so that the recurrence epilogue of slu_gru_bf16.hip, which
// applies the same mask and window to a value it holds in registers, reproduces these results bit for bit for any p)
__device__ __forceinline__ float4 mul4(float4 a, float4 b) {
  return make_float4(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y), __fmul_rn(a.z, b.z), __fmul_rn(a.w, b.w));
}

// grid: x over B*C/4 quads, y over output frames.  NS = 0: fp32 output y; NS = 1 / 2 / 3: the output goes straight
// into the split-precision activation format (NS 16-bit planes of (T_out*B) x C, plane stride `plane` elements)
// that the next frozen layer's input-projection GEMM reads (slu_gemm_bf16): no fp32 round trip, no split pass.
template <int NS>
__global__ void __launch_bounds__(256)
dropout_pool_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned short* __restrict__ planes,
                         long long plane, const PoolParams q) {
  const int C4 = q.C >> 2;
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  if (e >= (unsigned)q.B * C4) return;
  const int b = e / C4, c = (e - b * C4) * 4;
  const int to = blockIdx.y;
  const int t0 = to * q.factor;
  const size_t row = (size_t)q.B * q.C;
  const size_t col = (size_t)b * q.C + c;
  const uint64_t off = q.offset + (q.offset_dev ? *q.offset_dev : 0ull);
  float4 acc;
  if (q.method == 0) {
    acc = mul4(*reinterpret_cast<const float4*>(x + t0 * row + col), keep_scale4(q, off, t0, b, c));
  } else {
    const int t1 = min(q.T, t0 + q.factor);
    acc = (q.method == 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int t = t0; t < t1; ++t) {
      const float4 v = mul4(*reinterpret_cast<const float4*>(x + t * row + col), keep_scale4(q, off, t, b, c));
      if (q.method == 1) { acc.x = __fadd_rn(acc.x, v.x); acc.y = __fadd_rn(acc.y, v.y); acc.z = __fadd_rn(acc.z, v.z); acc.w = __fadd_rn(acc.w, v.w); }
      else { acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y); acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w); }
    }
    if (q.method == 1) {
      const float n = (float)(t1 - t0);
      acc.x = acc.x / n; acc.y = acc.y / n; acc.z = acc.z / n; acc.w = acc.w / n;
    }
  }
  if (NS == 0) {
    *reinterpret_cast<float4*>(y + to * row + col) = acc;
  } else {
    constexpr int NP = NS == 0 ? 1 : NS;
    unsigned short h[4][NP];
    if constexpr (NS == 2) {      // the flush rule shared with the recurrence's fused epilogue (slu_bf16.h)
      split_f16x2_flush(acc.x, h[0]); split_f16x2_flush(acc.y, h[1]); split_f16x2_flush(acc.z, h[2]); split_f16x2_flush(acc.w, h[3]);
    } else {
      split_terms<NP>(acc.x, h[0]); split_terms<NP>(acc.y, h[1]); split_terms<NP>(acc.z, h[2]); split_terms<NP>(acc.w, h[3]);
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
      *reinterpret_cast<uint2*>(planes + (size_t)pl * plane + to * row + col) =
          make_uint2(h[0][pl] | ((unsigned)h[1][pl] << 16), h[2][pl] | ((unsigned)h[3][pl] << 16));
  }
}

// The dropout mask of a FROZEN layer as a bit stream (one bit per element: word (t, b, c / 32), bit c % 32), consumed by the
// recurrence's fused epilogue (slu_gru_bf16.hip) and, for pooling modes that epilogue does not cover, by the kernels above
// (keep_bits) — one definition of a frozen layer's mask whichever path applies it.  Random bits are used economically:
//   p == 0.5 and C % 128 == 0 (every layer of the reference cfgs): element (t, b, c) is kept iff bit c % 32 of word
//     (c / 32) % 4 of the Philox block with counter (row * C/128 + c / 128, offset) is set — 128 elements per block, where the
//     one-uniform-per-element rule of the trainable layers' kernels spends a block on four (the T = 300 mask of a
//     1024-sequence super-batch: 111 us -> a few us on the look-ahead partition);
//   otherwise: a 16-bit draw per element — element e = row * C + c is kept iff halfword e % 2 of word (e / 2) % 4 of block
//     e / 8 is below round((1 - p) 2^16).
// row = t * B + b, or t * sub_batch + (b % sub_batch) with the offset advanced by sub_stride per sub-batch (several steps'
// frozen stages in one launch, each with its own step's stream).  grid: x over the (b, group) pairs, y over frames.
__global__ void __launch_bounds__(256)
dropout_bits_kernel(unsigned* __restrict__ bits, const PoolParams q, const int half_mode, const unsigned thr16) {
  const int G = half_mode ? (q.C >> 7) : (q.C >> 5);          // groups per row: 128 channels (one block) / 32 channels (one word)
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  if (e >= (unsigned)q.B * G) return;
  const int b = e / G, gi = e - b * G;
  const int t = blockIdx.y;
  uint64_t off = q.offset + (q.offset_dev ? *q.offset_dev : 0ull);
  uint64_t row;
  if (q.sub_batch > 0) {
    const int k = b / q.sub_batch, bl = b - k * q.sub_batch;
    row = (uint64_t)t * q.sub_batch + bl;
    off += (uint64_t)k * q.sub_stride;
  } else {
    row = (uint64_t)t * q.B + b;
  }
  unsigned* dst = bits + ((size_t)t * q.B + b) * (q.C >> 5);
  if (half_mode) {
    uint32_t w4[4];
    philox_block(q.seed, off, row * G + gi, w4);
    *reinterpret_cast<uint4*>(dst + gi * 4) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
  } else {
    const uint64_t blk0 = (row * q.C + (uint64_t)gi * 32) >> 3;
    unsigned word = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t w4[4];
      philox_block(q.seed, off, blk0 + k, w4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        word |= ((w4[j] & 0xffffu) < thr16 ? 1u : 0u) << (8 * k + 2 * j);
        word |= ((w4[j] >> 16) < thr16 ? 1u : 0u) << (8 * k + 2 * j + 1);
      }
    }
    dst[gi] = word;
  }
}

// grid: x over B*C/4 quads, y over OUTPUT frames; a thread writes dx for every input frame of its window
__global__ void __launch_bounds__(256)
dropout_pool_bwd4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                         float* __restrict__ dx, const PoolParams q) {
  const int C4 = q.C >> 2;
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  if (e >= (unsigned)q.B * C4) return;
  const int b = e / C4, c = (e - b * C4) * 4;
  const int to = blockIdx.y;
  const int t0 = to * q.factor;
  const int t1 = min(q.T, t0 + q.factor);
  const size_t row = (size_t)q.B * q.C;
  const size_t col = (size_t)b * q.C + c;
  const uint64_t off = q.offset + (q.offset_dev ? *q.offset_dev : 0ull);
  const float4 g = *reinterpret_cast<const float4*>(dy + to * row + col);
  if (q.method == 0) {
    *reinterpret_cast<float4*>(dx + t0 * row + col) = mul4(g, keep_scale4(q, off, t0, b, c));
    for (int t = t0 + 1; t < t1; ++t) *reinterpret_cast<float4*>(dx + t * row + col) = make_float4(0.f, 0.f, 0.f, 0.f);
  } else if (q.method == 1) {
    const float n = (float)(t1 - t0);
    for (int t = t0; t < t1; ++t) {
      const float4 k = keep_scale4(q, off, t, b, c);
      *reinterpret_cast<float4*>(dx + t * row + col) = make_float4(g.x * k.x / n, g.y * k.y / n, g.z * k.z / n, g.w * k.w / n);
    }
  } else {
    // arg-max per channel (first maximum wins, as in the scalar kernel and ATen's max_pool1d)
    int ax = t0, ay = t0, az = t0, aw = t0;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int t = t0; t < t1; ++t) {
      const float4 v = mul4(*reinterpret_cast<const float4*>(x + t * row + col), keep_scale4(q, off, t, b, c));
      if (v.x > best.x) { best.x = v.x; ax = t; }
      if (v.y > best.y) { best.y = v.y; ay = t; }
      if (v.z > best.z) { best.z = v.z; az = t; }
      if (v.w > best.w) { best.w = v.w; aw = t; }
    }
    for (int t = t0; t < t1; ++t) {
      const float4 k = keep_scale4(q, off, t, b, c);
      *reinterpret_cast<float4*>(dx + t * row + col) =
          make_float4(ax == t ? g.x * k.x : 0.f, ay == t ? g.y * k.y : 0.f, az == t ? g.z * k.z : 0.f, aw == t ? g.w * k.w : 0.f);
    }
  }
}

// ---- MaxPool1d(k, ceil_mode) + [Abs before it] + LeakyReLU / ReLU after it, for pool widths the convolution's
// epilogue does not fuse (k > 2; reference models.py:163-168, :205, :211-213 allow any cnn_max_pool_len).  x channels-last
// (B, L, C); y[b, lo, c] at b * out_sb + lo * out_sl + c; route (B, L_out, C) = arg-max offset | sign bit (abs) << 7.
__global__ void __launch_bounds__(256)
pool_act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ route, int B, int L,
                    int C, int L_out, int pool, int do_abs, float slope, long long out_sb, long long out_sl) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)B * L_out * C) return;
  const int c = (int)(e % C);
  const long long bl = e / C;
  const int lo = (int)(bl % L_out), b = (int)(bl / L_out);
  const int l0 = lo * pool, l1 = min(L, l0 + pool);
  float best = -INFINITY;
  int arg = 0, neg = 0;
  for (int l = l0; l < l1; ++l) {
    const float v = x[((size_t)b * L + l) * C + c];
    const float u = do_abs ? fabsf(v) : v;
    if (u > best) { best = u; arg = l - l0; neg = (do_abs && v < 0.0f) ? 1 : 0; }
  }
  y[(size_t)b * out_sb + (size_t)lo * out_sl + c] = best > 0.0f ? best : best * slope;
  if (route) route[e] = (unsigned char)(arg | (neg << 7));
}

__global__ void __launch_bounds__(256)
pool_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const unsigned char* __restrict__ route,
                    float* __restrict__ dx, int B, int L, int C, int L_out, int pool, float slope, long long out_sb,
                    long long out_sl) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)B * L_out * C) return;
  const int c = (int)(e % C);
  const long long bl = e / C;
  const int lo = (int)(bl % L_out), b = (int)(bl / L_out);
  const int l0 = lo * pool, l1 = min(L, l0 + pool);
  const size_t o = (size_t)b * out_sb + (size_t)lo * out_sl + c;
  const unsigned char r = route[e];
  float g = dy[o] * (y[o] > 0.0f ? 1.0f : slope);
  if (r & 0x80) g = -g;
  const int arg = r & 0x7f;
  for (int l = l0; l < l1; ++l) dx[((size_t)b * L + l) * C + c] = (l - l0 == arg) ? g : 0.0f;
}

static int pool_fill(PoolParams& q, const char* who, const float* mask, int64_t m_st, int64_t m_sb, const uint32_t* keep_bits,
                     float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                     int64_t sub_batch, uint64_t sub_stride, int method, int64_t factor, int64_t T,
                     int64_t B, int64_t C) {
  SLU_REQUIRE(sub_batch >= 0 && (sub_batch == 0 || B % sub_batch == 0), "%s: B must be a multiple of sub_batch", who);
  q.sub_batch = (int)sub_batch; q.sub_stride = sub_stride;
  SLU_REQUIRE(T > 0 && B > 0 && C > 0 && factor > 0, "%s: non-positive size", who);
  SLU_REQUIRE(method >= 0 && method <= 2, "%s: downsampling method must be 0 (none), 1 (avg) or 2 (max)", who);
  SLU_REQUIRE(p >= 0.0f && p < 1.0f, "%s: dropout p must be in [0,1)", who);
  SLU_REQUIRE(!keep_bits || (!mask && C % 32 == 0), "%s: keep_bits excludes a float mask and needs C %% 32 == 0", who);
  q.mask = mask; q.m_st = m_st; q.m_sb = m_sb; q.keep_bits = keep_bits; q.p = p; q.scale = 1.0f / (1.0f - p);
  q.seed = seed; q.offset = offset; q.offset_dev = (const unsigned long long*)offset_dev; q.method = method; q.factor = (int)factor;
  q.T = (int)T; q.B = (int)B; q.C = (int)C; q.T_out = (int)cdiv(T, factor);
  return SLU_OK;
}

// float4 path: channel count and every row start 16-byte aligned, grid within limits
static bool pool_vec_ok(const PoolParams& q, const float* a, const float* b, const float* mask) {
  if (q.C % 4 != 0 || q.T_out > 65535 || (long long)q.B * q.C >= (1LL << 31)) return false;
  if ((reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(b) & 15)) return false;
  (void)mask;   // explicit masks are read with scalar loads (arbitrary strides)
  return true;
}

}  // namespace slu

using namespace slu;

extern "C" int slu_dropout_pool_fwd(const float* x, const float* mask, int64_t m_st, int64_t m_sb, const uint32_t* keep_bits,
                                    float p, uint64_t seed, uint64_t offset,
                                    const uint64_t* offset_dev, int64_t sub_batch,
                                    uint64_t sub_stride, int method, int64_t factor, float* y,
                                    int64_t T, int64_t B, int64_t C, void* stream) {
  SLU_REQUIRE(x && y, "slu_dropout_pool_fwd: null pointer");
  PoolParams q;
  int rc = pool_fill(q, "slu_dropout_pool_fwd", mask, m_st, m_sb, keep_bits, p, seed, offset, offset_dev, sub_batch, sub_stride, method, factor, T, B, C);
  if (rc) return rc;
  if (pool_vec_ok(q, x, y, mask)) {
    hipLaunchKernelGGL(dropout_pool_fwd4_kernel<0>, dim3((unsigned)cdiv(B * (C / 4), 256), (unsigned)q.T_out),
                       dim3(256), 0, (hipStream_t)stream, x, y, (unsigned short*)nullptr, 0LL, q);
    SLU_CHECK_LAUNCH("dropout_pool_fwd4_kernel");
    return SLU_OK;
  }
  const long long total = (long long)q.T_out * B * C;
  hipLaunchKernelGGL(dropout_pool_fwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, y, q);
  SLU_CHECK_LAUNCH("dropout_pool_fwd_kernel");
  return SLU_OK;
}

extern "C" int slu_dropout_pool_fwd_planes(const float* x, const float* mask, int64_t m_st, int64_t m_sb,
                                           const uint32_t* keep_bits, float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                           int64_t sub_batch, uint64_t sub_stride, int method, int64_t factor,
                                           void* planes, int64_t plane_stride, int nsplit, int64_t T, int64_t B,
                                           int64_t C, void* stream) {
  SLU_REQUIRE(x && planes, "slu_dropout_pool_fwd_planes: null pointer");
  SLU_REQUIRE(nsplit >= 1 && nsplit <= 3, "slu_dropout_pool_fwd_planes: nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  PoolParams q;
  int rc = pool_fill(q, "slu_dropout_pool_fwd_planes", mask, m_st, m_sb, keep_bits, p, seed, offset, offset_dev, sub_batch, sub_stride, method, factor, T, B, C);
  if (rc) return rc;
  if (C % 32 != 0 || !pool_vec_ok(q, x, x, mask) || (reinterpret_cast<uintptr_t>(planes) & 15) || (plane_stride & 7))
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_dropout_pool_fwd_planes: needs C %% 32 == 0 (got %lld), aligned buffers, T_out <= 65535", (long long)C);
  SLU_REQUIRE(plane_stride >= (int64_t)q.T_out * B * C, "slu_dropout_pool_fwd_planes: plane stride too small");
  dim3 grid((unsigned)cdiv(B * (C / 4), 256), (unsigned)q.T_out);
#define SLU_DPP(NS_) hipLaunchKernelGGL(dropout_pool_fwd4_kernel<NS_>, grid, dim3(256), 0, (hipStream_t)stream, x, (float*)nullptr, \
                                        (unsigned short*)planes, (long long)plane_stride, q)
  if (nsplit == 3) SLU_DPP(3); else if (nsplit == 2) SLU_DPP(2); else SLU_DPP(1);
#undef SLU_DPP
  SLU_CHECK_LAUNCH("dropout_pool_fwd4_kernel(planes)");
  return SLU_OK;
}

extern "C" int slu_dropout_bits(uint32_t* bits, float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                int64_t sub_batch, uint64_t sub_stride, int64_t T, int64_t B, int64_t C, void* stream) {
  SLU_REQUIRE(bits, "slu_dropout_bits: null pointer");
  PoolParams q;
  int rc = pool_fill(q, "slu_dropout_bits", nullptr, 0, 0, nullptr, p, seed, offset, offset_dev, sub_batch, sub_stride, 1, 1, T, B, C);
  if (rc) return rc;
  if (C % 32 != 0 || T > 65535 || B * (C / 32) >= (1LL << 31) || ((uintptr_t)bits & 15))
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_dropout_bits: needs C %% 32 == 0 (got %lld), T <= 65535 and a 16-byte aligned buffer", (long long)C);
  const int half_mode = (p == 0.5f && C % 128 == 0) ? 1 : 0;
  const double t16 = (1.0 - (double)p) * 65536.0 + 0.5;
  const unsigned thr16 = t16 >= 65536.0 ? 65536u : (unsigned)t16;
  hipLaunchKernelGGL(dropout_bits_kernel, dim3((unsigned)cdiv(B * (half_mode ? C / 128 : C / 32), 256), (unsigned)T), dim3(256), 0,
                     (hipStream_t)stream, bits, q, half_mode, thr16);
  SLU_CHECK_LAUNCH("dropout_bits_kernel");
  return SLU_OK;
}

extern "C" int slu_dropout_pool_bwd(const float* dy, const float* x, const float* y,
                                    const float* mask, int64_t m_st, int64_t m_sb, float p,
                                    uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                    int64_t sub_batch, uint64_t sub_stride, int method,
                                    int64_t factor, float* dx, int64_t T, int64_t B, int64_t C,
                                    void* stream) {
  SLU_REQUIRE(dy && dx, "slu_dropout_pool_bwd: null pointer");
  SLU_REQUIRE(method != 2 || x, "slu_dropout_pool_bwd: x is required for max pooling");
  (void)y;
  PoolParams q;
  int rc = pool_fill(q, "slu_dropout_pool_bwd", mask, m_st, m_sb, nullptr, p, seed, offset, offset_dev, sub_batch, sub_stride, method, factor, T, B, C);
  if (rc) return rc;
  if (pool_vec_ok(q, dy, dx, mask) && (method != 2 || (reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
    hipLaunchKernelGGL(dropout_pool_bwd4_kernel, dim3((unsigned)cdiv(B * (C / 4), 256), (unsigned)q.T_out),
                       dim3(256), 0, (hipStream_t)stream, dy, x, dx, q);
    SLU_CHECK_LAUNCH("dropout_pool_bwd4_kernel");
    return SLU_OK;
  }
  const long long total = (long long)T * B * C;
  hipLaunchKernelGGL(dropout_pool_bwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, x, dx, q);
  SLU_CHECK_LAUNCH("dropout_pool_bwd_kernel");
  return SLU_OK;
}

extern "C" int slu_pool_act_fwd(const float* x, float* y, uint8_t* route, int64_t B, int64_t L, int64_t C, int64_t pool,
                                int do_abs, float slope, int64_t out_sb, int64_t out_sl, void* stream) {
  SLU_REQUIRE(x && y, "slu_pool_act_fwd: null pointer");
  SLU_REQUIRE(B > 0 && L > 0 && C > 0 && pool >= 1 && pool <= 127, "slu_pool_act_fwd: bad size (pool width 1..127)");
  const int64_t L_out = cdiv(L, pool);
  hipLaunchKernelGGL(pool_act_fwd_kernel, dim3((unsigned)cdiv(B * L_out * C, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                     route, (int)B, (int)L, (int)C, (int)L_out, (int)pool, do_abs, slope, (long long)out_sb, (long long)out_sl);
  SLU_CHECK_LAUNCH("pool_act_fwd_kernel");
  return SLU_OK;
}

extern "C" int slu_pool_act_bwd(const float* dy, const float* y, const uint8_t* route, float* dx, int64_t B, int64_t L,
                                int64_t C, int64_t pool, float slope, int64_t out_sb, int64_t out_sl, void* stream) {
  SLU_REQUIRE(dy && y && route && dx, "slu_pool_act_bwd: null pointer");
  SLU_REQUIRE(B > 0 && L > 0 && C > 0 && pool >= 1 && pool <= 127, "slu_pool_act_bwd: bad size (pool width 1..127)");
  const int64_t L_out = cdiv(L, pool);
  hipLaunchKernelGGL(pool_act_bwd_kernel, dim3((unsigned)cdiv(B * L_out * C, 256)), dim3(256), 0, (hipStream_t)stream, dy, y,
                     route, dx, (int)B, (int)L, (int)C, (int)L_out, (int)pool, slope, (long long)out_sb, (long long)out_sl);
  SLU_CHECK_LAUNCH("pool_act_bwd_kernel");
  return SLU_OK;
}
