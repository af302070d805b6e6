// Split-precision windowed-GEMM convolution block (forward only) for FROZEN CNN layers — the SincNet filterbank
// convolution and the two dense Conv1d layers with their fused epilogue (reference: models.py:108 / :200 conv,
// :163-168 Abs, :205 MaxPool1d(ceil_mode), :211 LeakyReLU).  Same formulation as slu_wconv.hip:
//     out[b][l][c] = sum_q  in_flat[b][l*S + q - pad] * W[c][q]        (M = frames, N = channels, K = taps q)
// but the contraction runs on v_mfma_f32_16x16x32_bf16 with both operands split into NS bf16 terms
// (slu_bf16.h: NS = 3 -> six products, fp32-class; NS = 1 -> bf16):
//   * the fp32 input window of a workgroup's frames is read once from HBM (eight loads in flight per thread),
//     split in registers and staged as NS bf16 planes in LDS, rows of S elements with a stride of S + 8: an
//     A fragment (frame i, taps 8 kg .. 8 kg + 7 of a 32-tap chunk) is ONE ds_read_b128 — S % 8 == 0 keeps the eight
//     taps inside a row, the +8 shifts consecutive frames by one 16-byte slot: conflict-free;
//   * channel counts that are not a multiple of 8 are padded in the LDS image and in the packed filters
//     (conv2: 60 -> 64 input channels, zero weights), the 401 Sinc taps to 416;
//   * filters are split and packed once per launch in B-fragment order (bf_wconv_pack_r3_kernel) and read from L2 one
//     chunk ahead; bias, abs, max-pool, LeakyReLU and the output layout are the fp32 kernel's epilogue.
#include "slu_bf16.h"   // round-3 kernel kept as an A/B baseline (tools/build_alt.sh EXTRA_UNITS); not part of the product

namespace slu {

constexpr int WB_THREADS = 256;

struct WconvBfParamsR3 {
  const float* in;      // (B, in_row) flat fp32 rows
  const float* const* in_tab;   // null, or a device table of base pointers: row b = in_tab[b / tab_rows] + (b % tab_rows) * in_row
  int tab_rows;                 // (a look-ahead super-batch reads its batches where they lie: no concatenation copy)
  const uint4* wp;      // packed filters [plane][KC][NT][64]
  const float* bias;    // (c_out) or null
  float* out;
  unsigned char* route;     // null, or (B, l_out, c_out): pool pick | sign << 1, as wconv_fwd_kernel writes it (trainable
                            // blocks in bf16 mode: bf16 forward, exact fp32 backward through slu_wconv_bwd_*)
  unsigned short* planes;   // null, or NS bf16 planes of (l_out * Bn) x Kp_out (time-major rows f * Bn + b, zero padded
  long long plane;          // columns): the split-precision activation format the next frozen GRU layer's GEMM reads
  int Kp_out, Bn;
  long long in_row;     // floats per batch row (l_in * c_in)
  long long out_sb, out_sl;
  int S, S_real, Sp;    // LDS row length (bf16 elements), global elements per row, LDS row stride
  int KC, pad;          // 32-tap chunks; left padding in GLOBAL elements
  int l_conv, l_out, c_out;
  int do_abs, pool;
  float slope;
  int nrows;            // LDS rows staged per workgroup
};

// mode: filters W(c, q') for padded tap index q' = k * c_pad + ci (c_in > 1) or q' = tap (c_in == 1)
template <int NS>
__global__ void __launch_bounds__(256)
bf_wconv_pack_r3_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int c_out, int c_in, int c_pad, int k_t,
                     int NT, int KC) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (kc, nt, lane)
  if (idx >= KC * NT * 64) return;
  const int lane = idx & 63, nt = (idx >> 6) % NT, kc = (idx >> 6) / NT;
  const int c = nt * 16 + (lane & 15), q0 = kc * 32 + (lane >> 4) * 8;
  unsigned short h[NS][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int q = q0 + e;
    const int k = q / c_pad, ci = q - k * c_pad;
    const float v = (c < c_out && k < k_t && ci < c_in) ? w[((size_t)c * c_in + ci) * k_t + k] : 0.0f;
    unsigned short sp[NS];
    split_terms<NS>(v, sp);
#pragma unroll
    for (int p = 0; p < NS; ++p) h[p][e] = sp[p];
  }
#pragma unroll
  for (int p = 0; p < NS; ++p) {
    uint4 o;
    o.x = h[p][0] | ((unsigned)h[p][1] << 16); o.y = h[p][2] | ((unsigned)h[p][3] << 16);
    o.z = h[p][4] | ((unsigned)h[p][5] << 16); o.w = h[p][6] | ((unsigned)h[p][7] << 16);
    wp[(size_t)p * KC * NT * 64 + idx] = o;
  }
}

// SPLITN: the four waves form a 2 x 2 grid (frames x channels) instead of 4 x 1: a wave then needs only half of the
// filter fragments of a k-chunk.  With 4 x 1 every wave fetches ALL NT x NS fragments from L2 — 48 KB per workgroup
// and chunk for 768 MFMA cycles = the whole 64 B/clk L2 port of the CU; 2 x 2 halves that (the A fragments, read
// from LDS by two waves each, take the difference: 62 B/clk of the LDS' 128).  NT even only.
template <int MT, int NT, int NS, bool SPLITN>
__global__ void __launch_bounds__(WB_THREADS, 2)
wconv_bf_fwd_r3_kernel(const WconvBfParamsR3 p) {
  constexpr int RT = SPLITN ? 2 * MT : MT;        // row (frame) tiles per wave
  constexpr int CT = SPLITN ? NT / 2 : NT;        // column (channel) tiles per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* lds = reinterpret_cast<unsigned short*>(smem);     // [NS][nrows][Sp]
  constexpr int F = 64 * MT;                      // frames per workgroup
  typedef Split<NS> SP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * F;
  const float* __restrict__ inb;
  if (p.in_tab) {
    const int e = b / p.tab_rows;
    inb = p.in_tab[e] + (size_t)(b - e * p.tab_rows) * p.in_row;
  } else {
    inb = p.in + (size_t)b * p.in_row;
  }
  const int plane = p.nrows * p.Sp;               // bf16 elements per LDS plane
  const int row0 = SPLITN ? (wave >> 1) * 16 * RT : wave * 16 * MT;   // this wave's first frame in the tile
  const int nb = SPLITN ? (wave & 1) * CT : 0;                        // ... and its first channel tile

  // ---- stage the window: LDS (row, col) <- global element u0 + row * S_real + col (col < S_real), zero elsewhere;
  //      two adjacent columns per thread and step (one 4-byte LDS store per plane), eight steps' loads in flight ----
  {
    const long long u0 = (long long)l0 * p.S_real - p.pad;
    const int half = p.S >> 1;                     // column pairs per row
    const int total = p.nrows * half;
    const float inv = 1.0f / (float)half;
    constexpr int U = 8;
    for (int base = 0; base < total; base += WB_THREADS * U) {
      float v0[U], v1[U];
      int off[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int idx = base + j * WB_THREADS + tid;
        int row = (int)((float)idx * inv);
        int cp = idx - row * half;
        if (cp < 0) { cp += half; --row; }
        if (cp >= half) { cp -= half; ++row; }
        const int col = 2 * cp;
        const long long u = u0 + (long long)row * p.S_real + col;
        const bool ok = idx < total;
        off[j] = ok ? row * p.Sp + col : -1;
        v0[j] = (ok && col < p.S_real && u >= 0 && u < p.in_row) ? inb[u] : 0.0f;
        v1[j] = (ok && col + 1 < p.S_real && u + 1 >= 0 && u + 1 < p.in_row) ? inb[u + 1] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if (off[j] < 0) continue;
        unsigned short a[NS], c[NS];
        split_terms<NS>(v0[j], a);
        split_terms<NS>(v1[j], c);
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
          *reinterpret_cast<unsigned*>(lds + pl * plane + off[j]) = a[pl] | ((unsigned)c[pl] << 16);
      }
    }
  }
  __syncthreads();

  f32x4 accs[SP::NACC][RT][CT];
#pragma unroll
  for (int a = 0; a < SP::NACC; ++a)
#pragma unroll
    for (int m = 0; m < RT; ++m)
#pragma unroll
      for (int n = 0; n < CT; ++n) accs[a][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int i = lane & 15, kg = lane >> 4;
  // tap chunk (kc, kg) starts at padded tap q = kc*32 + kg*8: LDS row offset q / S, column q % S (multiple of 8)
  int qd = (kg * 8) / p.S, qm = kg * 8 - qd * p.S;
  int abase[RT];
#pragma unroll
  for (int m = 0; m < RT; ++m) abase[m] = (row0 + m * 16 + i) * p.Sp;
  const uint4* __restrict__ wp = p.wp + (size_t)nb * 64 + lane;
  const size_t w_plane = (size_t)p.KC * NT * 64;

  uint4 fb[NS][CT], fbn[NS][CT];
#pragma unroll
  for (int pl = 0; pl < NS; ++pl)
#pragma unroll
    for (int n = 0; n < CT; ++n) fb[pl][n] = wp[pl * w_plane + (size_t)n * 64];
  for (int kc = 0; kc < p.KC; ++kc) {
    const int kn = min(kc + 1, p.KC - 1);          // unconditional prefetch (the last chunk re-reads itself)
#pragma unroll
    for (int pl = 0; pl < NS; ++pl)
#pragma unroll
      for (int n = 0; n < CT; ++n) fbn[pl][n] = wp[pl * w_plane + ((size_t)kn * NT + n) * 64];
    uint4 fa[NS][RT];
    const int aoff = qd * p.Sp + qm;
#pragma unroll
    for (int pl = 0; pl < NS; ++pl)
#pragma unroll
      for (int m = 0; m < RT; ++m) fa[pl][m] = *reinterpret_cast<const uint4*>(lds + pl * plane + abase[m] + aoff);
    __builtin_amdgcn_sched_barrier(0);             // next chunk's filter loads and this chunk's fragments issued HERE
#pragma unroll
    for (int q = 0; q < SP::NPAIR; ++q) {
#pragma unroll
      for (int m = 0; m < RT; ++m)
#pragma unroll
        for (int n = 0; n < CT; ++n)
          accs[SP::ACC(q)][m][n] = mfma_split<NS>(fa[SP::PA(q)][m], fb[SP::PB(q)][n], accs[SP::ACC(q)][m][n]);
    }
    qm += 32;
    while (qm >= p.S) { qm -= p.S; ++qd; }
#pragma unroll
    for (int pl = 0; pl < NS; ++pl)
#pragma unroll
      for (int n = 0; n < CT; ++n) fb[pl][n] = fbn[pl][n];
  }

  f32x4 acc[RT][CT];
#pragma unroll
  for (int m = 0; m < RT; ++m)
#pragma unroll
    for (int n = 0; n < CT; ++n) acc[m][n] = split_result<NS>(accs[0][m][n], accs[SP::NACC - 1][m][n]);

  // ---- epilogue: bias, abs, max-pool over frame pairs, LeakyReLU, strided store (as wconv_fwd_kernel) ----
#pragma unroll
  for (int m = 0; m < RT; ++m) {
    const int fbase = l0 + row0 + m * 16 + 4 * kg;   // multiple of 4
#pragma unroll
    for (int n = 0; n < CT; ++n) {
      const int c = (nb + n) * 16 + i;
      if (p.planes) {
        // straight into the split format (pool == 1 only): columns [c_out, Kp_out) are the zero padding
        if (c >= p.Kp_out) continue;
        const bool real = c < p.c_out;
        const float bias = (real && p.bias) ? p.bias[c] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = fbase + r;
          if (f >= p.l_conv) continue;
          float t = acc[m][n][r] + bias;
          t = p.do_abs ? fabsf(t) : t;
          t = real ? (t > 0.0f ? t : t * p.slope) : 0.0f;
          unsigned short sp[NS];
          split_terms<NS>(t, sp);
#pragma unroll
          for (int pl = 0; pl < NS; ++pl)
            p.planes[(size_t)pl * p.plane + ((size_t)f * p.Bn + b) * p.Kp_out + c] = sp[pl];
        }
        continue;
      }
      if (c >= p.c_out) continue;
      const float bias = p.bias ? p.bias[c] : 0.0f;
      float v[4];
      bool neg[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = acc[m][n][r] + bias;
        neg[r] = t < 0.0f;
        v[r] = p.do_abs ? fabsf(t) : t;
      }
      if (p.pool == 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int f0 = fbase + 2 * h;
          if (f0 >= p.l_conv) continue;
          const bool has1 = (f0 + 1) < p.l_conv;                // ceil_mode: last window may be partial
          const bool pick1 = has1 && v[2 * h + 1] > v[2 * h];
          const float pooled = pick1 ? v[2 * h + 1] : v[2 * h];
          p.out[(size_t)b * p.out_sb + (long long)(f0 >> 1) * p.out_sl + c] = pooled > 0.0f ? pooled : pooled * p.slope;
          if (p.route) {
            const bool sgn = pick1 ? neg[2 * h + 1] : neg[2 * h];
            p.route[((size_t)b * p.l_out + (f0 >> 1)) * p.c_out + c] = (unsigned char)((pick1 ? 1 : 0) | (sgn ? 2 : 0));
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = fbase + r;
          if (f >= p.l_conv) continue;
          p.out[(size_t)b * p.out_sb + (long long)f * p.out_sl + c] = v[r] > 0.0f ? v[r] : v[r] * p.slope;
          if (p.route) p.route[((size_t)b * p.l_out + f) * p.c_out + c] = (unsigned char)(neg[r] ? 2 : 0);
        }
      }
    }
  }
}

static inline int bf_nt_for(int64_t c) {
  const int need = (int)cdiv(c, 16);
  if (need <= 1) return 1;
  if (need <= 2) return 2;
  if (need <= 4) return 4;
  if (need <= 5) return 5;
  if (need <= 8) return 8;
  return -1;
}

template <int MT, int NT, int NS>
static int bf_launch(dim3 grid, size_t lds, hipStream_t st, const WconvBfParamsR3& p) {
  constexpr bool SPLITN = (NT % 2 == 0) && MT == 2;      // 2 x 2 waves where the channel tiles divide
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)wconv_bf_fwd_r3_kernel<MT, NT, NS, SPLITN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "wconv_bf16: cannot raise the dynamic LDS cap to %zu: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL((wconv_bf_fwd_r3_kernel<MT, NT, NS, SPLITN>), grid, dim3(WB_THREADS), lds, st, p);
  SLU_CHECK_LAUNCH("wconv_bf_fwd_r3_kernel");
  return SLU_OK;
}

template <int NS>
static int bf_launch_nt(int MT, int NT, dim3 grid, size_t lds, hipStream_t st, const WconvBfParamsR3& p) {
#define SLU_BF_CASE(M_, N_) if (MT == M_ && NT == N_) return bf_launch<M_, N_, NS>(grid, lds, st, p);
  SLU_BF_CASE(1, 1) SLU_BF_CASE(1, 2) SLU_BF_CASE(1, 4) SLU_BF_CASE(1, 5) SLU_BF_CASE(1, 8)
  SLU_BF_CASE(2, 1) SLU_BF_CASE(2, 2) SLU_BF_CASE(2, 4) SLU_BF_CASE(2, 5) SLU_BF_CASE(2, 8)
#undef SLU_BF_CASE
  SLU_FAIL(SLU_ERR_UNSUPPORTED, "wconv_bf16: unsupported tile configuration %d x %d", MT, NT);
}

}  // namespace slu

using namespace slu;

static int64_t bf_c_pad(int64_t c_in) { return c_in == 1 ? 1 : cdiv(c_in, 8) * 8; }

extern "C" size_t slu_wconv_bf16_workspace_bytes_r3(int64_t c_out, int64_t c_in, int64_t k_t, int nsplit) {
  const int64_t Kw = k_t * bf_c_pad(c_in);
  const int64_t nt = cdiv(c_out, 16) <= 8 ? 8 : cdiv(c_out, 16);
  return (size_t)nsplit * cdiv(Kw, 32) * nt * 64 * sizeof(uint4);
}

extern "C" int slu_wconv_fwd_bf16_r3(const float* in, const float* const* in_table, int64_t table_rows,
                                  const float* weight, const float* bias, float* out, uint8_t* route, int64_t B,
                                  int64_t l_in, int64_t c_in, int64_t c_out, int64_t k_t, int64_t stride_t,
                                  int do_abs, int pool, float slope, int64_t out_sb, int64_t out_sl,
                                  void* out_planes, int64_t out_plane_stride,
                                  void* workspace, size_t workspace_bytes, int packed_valid, int nsplit, void* stream) {
  SLU_REQUIRE((in || in_table) && weight && (out || out_planes), "slu_wconv_fwd_bf16_r3: null pointer");
  SLU_REQUIRE(!in_table || (table_rows >= 1 && table_rows <= B), "slu_wconv_fwd_bf16_r3: bad table_rows");
  SLU_REQUIRE(B > 0 && l_in > 0 && c_in > 0 && c_out > 0 && k_t > 0 && stride_t > 0, "slu_wconv_fwd_bf16_r3: non-positive size");
  SLU_REQUIRE(pool == 1 || pool == 2, "slu_wconv_fwd_bf16_r3: pool must be 1 or 2 (got %d)", pool);
  SLU_REQUIRE(nsplit >= 1 && nsplit <= 3, "slu_wconv_fwd_bf16_r3: nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  SLU_REQUIRE(B <= 65535, "slu_wconv_fwd_bf16_r3: B must be <= 65535");
  const int64_t c_pad = bf_c_pad(c_in);
  const int64_t S = stride_t * c_pad, S_real = stride_t * c_in;
  if (S % 8 != 0 || (c_in > 1 && stride_t != 1))
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_wconv_fwd_bf16_r3: needs stride * channels %% 8 == 0 (stride 1 for multi-channel "
             "inputs); got stride %lld, c_in %lld", (long long)stride_t, (long long)c_in);
  const int NT = bf_nt_for(c_out);
  if (NT < 0) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_wconv_fwd_bf16_r3: at most 128 output channels (got %lld)", (long long)c_out);
  const int64_t Kw = k_t * c_pad, KC = cdiv(Kw, 32);
  const size_t need = (size_t)nsplit * KC * NT * 64 * sizeof(uint4);
  if (!workspace || workspace_bytes < need)
    SLU_FAIL(SLU_ERR_WORKSPACE, "slu_wconv_fwd_bf16_r3: workspace too small (%zu < %zu)", workspace_bytes, need);
  const int64_t pad_t = k_t / 2;
  const int64_t l_conv = (l_in + 2 * pad_t - k_t) / stride_t + 1;
  SLU_REQUIRE(l_conv > 0, "slu_wconv_fwd_bf16_r3: input shorter than the filter");
  hipStream_t st = (hipStream_t)stream;
  uint4* wp = reinterpret_cast<uint4*>(workspace);
  if (!packed_valid) {     // else: the workspace still holds the pack of these very filters (frozen block, caller's cache)
    const int total = (int)(KC * NT * 64);
#define SLU_WPACK(NS_) hipLaunchKernelGGL(bf_wconv_pack_r3_kernel<NS_>, dim3((total + 255) / 256), dim3(256), 0, st, weight, wp, \
                                          (int)c_out, (int)c_in, (int)c_pad, (int)k_t, NT, (int)KC)
    if (nsplit == 3) SLU_WPACK(3); else if (nsplit == 2) SLU_WPACK(2); else SLU_WPACK(1);
#undef SLU_WPACK
    SLU_CHECK_LAUNCH("bf_wconv_pack_r3_kernel");
  }
  WconvBfParamsR3 p;
  p.in = in; p.wp = wp; p.bias = bias; p.out = out; p.route = route;
  SLU_REQUIRE(!route || (out && !out_planes), "slu_wconv_fwd_bf16_r3: route goes with the fp32 output");
  p.in_tab = in_table; p.tab_rows = (int)(in_table ? table_rows : 1);
  p.planes = (unsigned short*)out_planes; p.plane = out_plane_stride; p.Kp_out = (int)(cdiv(c_out, 32) * 32); p.Bn = (int)B;
  if (out_planes) {
    if (pool != 1 || NT * 16 < p.Kp_out)
      SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_wconv_fwd_bf16_r3: out_planes needs pool == 1 and a channel tiling that covers "
               "round_up(c_out, 32) columns (c_out %lld)", (long long)c_out);
    SLU_REQUIRE(out_plane_stride >= (int64_t)((l_in + 2 * (k_t / 2) - k_t) / stride_t + 1) * B * p.Kp_out,
                "slu_wconv_fwd_bf16_r3: plane stride too small");
  }
  p.in_row = l_in * c_in; p.out_sb = out_sb; p.out_sl = out_sl;
  p.S = (int)S; p.S_real = (int)S_real; p.Sp = (int)S + 8;
  p.KC = (int)KC; p.pad = (int)(pad_t * c_in);
  p.l_conv = (int)l_conv; p.l_out = (int)cdiv(l_conv, pool); p.c_out = (int)c_out;
  p.do_abs = do_abs; p.pool = pool; p.slope = slope;
  int MT = (B * cdiv(l_conv, 128) >= 256) ? 2 : 1;
  int F = 64 * MT;
  p.nrows = F + (int)cdiv(KC * 32, S) + 1;
  size_t lds = (size_t)nsplit * p.nrows * p.Sp * sizeof(unsigned short);
  if (lds > 160 * 1024 && MT == 2) {           // long hops (stride ~200 and up at three planes): 64-frame tiles still fit
    MT = 1; F = 64;
    p.nrows = F + (int)cdiv(KC * 32, S) + 1;
    lds = (size_t)nsplit * p.nrows * p.Sp * sizeof(unsigned short);
  }
  if (lds > 160 * 1024) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_wconv_fwd_bf16_r3: window of %zu bytes exceeds the 160 KiB LDS", lds);
  dim3 grid((unsigned)cdiv(l_conv, F), (unsigned)B);
  return nsplit == 3 ? bf_launch_nt<3>(MT, NT, grid, lds, st, p)
       : nsplit == 2 ? bf_launch_nt<2>(MT, NT, grid, lds, st, p) : bf_launch_nt<1>(MT, NT, grid, lds, st, p);
}
