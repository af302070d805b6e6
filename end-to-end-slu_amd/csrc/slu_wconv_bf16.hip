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
//   * filters are split and packed once per launch in B-fragment order (bf_wconv_pack_kernel) and read from L2 one
//     chunk ahead; bias, abs, max-pool, LeakyReLU and the output layout are the fp32 kernel's epilogue.
#include "slu_bf16.h"
#include <type_traits>

namespace slu {

constexpr int WB_THREADS = 256;

struct WconvBfParams {
  const float* in;      // (B, in_row) flat fp32 rows
  const float* const* in_tab;   // null, or a device table of base pointers: row b = in_tab[b / tab_rows] + (b % tab_rows) * in_row
  int tab_rows;                 // (a look-ahead super-batch reads its batches where they lie: no concatenation copy)
  int pcm16;                    // != 0: `in` / the table's pointers address int16 samples; value = sample * in_scale (PCM16 wavs
  float in_scale;               // cross PCIe as int16, half the bytes, and become sample / 32768 here: exact in fp32)
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
  int vec4;             // != 0: the window is staged four elements at a time through a buffer descriptor of the sequence's
                        // row (S_real, pad and in_row multiples of 4): 16-byte (8-byte for PCM16) loads, and the convolution's
                        // zero padding — reads before the start / past the end of the row — is the descriptor's bounds check
  int stage_out;        // != 0: the epilogue goes through LDS — the workgroup's output tile (fp32 rows or plane rows) is
                        // assembled in the window's LDS and leaves in whole rows of 16-byte stores (launcher: alignment, size)
  unsigned* amax;       // null, or the f16x2 range word of this launch: atomicMax of the bit pattern of |v| over every value
                        // the launch splits (input window, plane output) — what the host's guard reads (slu_hip.h)
};

// Range guard of the f16x2 scheme: the largest |v| a launch splits, as its IEEE bit pattern (integer maximum: NaN and
// infinity compare above every finite value, so one word reports overflow and non-finite inputs alike).
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ void amax_publish(unsigned* word, unsigned mx) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, d, 64));
  // the word only grows: a (possibly stale) read that already covers this wave's maximum makes the atomic unnecessary —
  // after the first workgroups of a launch nearly every wave skips it
  if ((threadIdx.x & 63) == 0 && mx > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(word, mx);
}

// mode: filters W(c, q') for padded tap index q' = k * c_pad + ci (c_in > 1) or q' = tap (c_in == 1)
template <int NS>
__global__ void __launch_bounds__(256)
bf_wconv_pack_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int c_out, int c_in, int c_pad, int k_t,
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

// How the four waves share the (4 MT row tiles) x (NT column tiles) of a workgroup — MAP:
//   0  4 x 1: a wave owns MT row tiles and ALL column tiles: it fetches all NT x NS filter fragments of a k-chunk from
//      L2 — for the Sinc layer (NT = 5, f16x2) 40 KB per workgroup and chunk for 480 MFMA cycles = 85 B/clk against the
//      CU's 64 B/clk L2 port: the round-3 Sinc launch ran at 0.26 of its partition's MFMA peak, L2-port bound;
//   1  2 x 2 (NT even, MT = 2): frames x channels, a wave needs half of the filter fragments (the A fragments, read
//      from LDS by two waves each, take the difference);
//   2  column ownership (NT = 5, MT = 2; round 4): wave w owns column tile w over ALL eight row tiles, and the fifth
//      column tile is shared by rows — wave w takes its row tiles 2w, 2w + 1.  Same 30 MFMAs per wave and chunk, but
//      4 filter fragments per wave instead of 10 (16 KB per workgroup and chunk: 34 B/clk), paid with LDS reads
//      (16 A fragments per wave and chunk instead of 4: 136 B/clk of the LDS' 256).  A wave numbers its row tiles from
//      2w (local tile m' = global tile (m' + 2w) mod 8), so the two fragments the shared column needs are local 0 and 1.
template <int MT, int NT, int NS, int MAP>
__global__ void __launch_bounds__(WB_THREADS, 2)
wconv_bf_fwd_kernel(const WconvBfParams p) {
  constexpr bool SPLITN = MAP == 1;
  constexpr bool COLS = MAP == 2;
  static_assert(!COLS || (NT == 5 && MT == 2), "column ownership is written for NT = 5, MT = 2");
  constexpr int RT = COLS ? 8 : (SPLITN ? 2 * MT : MT);        // row (frame) tiles per wave
  constexpr int CT = COLS ? 1 : (SPLITN ? NT / 2 : NT);        // column (channel) tiles per wave (+ the shared one: COLS)
  constexpr int XT = COLS ? 2 : 0;                             // row tiles of the shared column tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* lds = reinterpret_cast<unsigned short*>(smem);     // [NS][nrows][Sp]
  constexpr int F = 64 * MT;                      // frames per workgroup
  typedef Split<NS> SP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * F;
  const float* __restrict__ inb;
  const short* __restrict__ inb16;
  {
    const size_t roff = (size_t)(p.in_tab ? b % p.tab_rows : b) * p.in_row;
    const float* base = p.in_tab ? p.in_tab[b / p.tab_rows] : p.in;
    inb = base + roff;
    inb16 = reinterpret_cast<const short*>(base) + roff;
  }
  const int plane = p.nrows * p.Sp;               // bf16 elements per LDS plane
  unsigned amx = 0;                               // f16x2 range guard (NS == 2 with p.amax only)
  const int row0 = COLS ? 0 : (SPLITN ? (wave >> 1) * 16 * RT : wave * 16 * MT);   // this wave's first frame in the tile
  const int nb = COLS ? wave : (SPLITN ? (wave & 1) * CT : 0);                     // ... and its first channel tile
  // global row tile of local row tile m (COLS: rotated by 2 * wave)
  auto rtile = [&](int m) { return COLS ? ((m + 2 * wave) & 7) : m; };

  // Filter fragments come from L2 (the packed bank is shared by every workgroup) through a ring of NB register buffers,
  // the tap loop unrolled NB times so that buffer indices are compile-time (no copies): chunk kc + NB - 1 is requested
  // while chunk kc is multiplied.  Round 3 kept ONE chunk in flight: its MFMAs (480 cycles per wave, two waves per
  // SIMD) are shorter than an L2 round trip, and the ISA showed `s_waitcnt vmcnt(0)` at the top of every chunk.
  // The first NB - 1 chunks are requested here, before the window is staged.
  const uint4* __restrict__ wp = p.wp + (size_t)nb * 64 + lane;
  const uint4* __restrict__ wpx = p.wp + (size_t)(NT - 1) * 64 + lane;      // COLS: the shared column tile
  const size_t w_plane = (size_t)p.KC * NT * 64;
  constexpr int FREGS = NS * (CT + (COLS ? 1 : 0)) * 4;          // VGPRs per buffer
  constexpr int NB = FREGS <= 24 ? 3 : 2;
  uint4 fb[NB][NS][CT], fx[NB][NS];
#define SLU_WB_FETCH(buf_, kc_)                                                                        \
  {                                                                                                    \
    const int kf_ = min((kc_), p.KC - 1);          /* unconditional (the tail re-reads the last chunk) */ \
    _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) {                                                \
      _Pragma("unroll") for (int n = 0; n < CT; ++n) fb[buf_][pl][n] = wp[pl * w_plane + ((size_t)kf_ * NT + n) * 64]; \
      if constexpr (COLS) fx[buf_][pl] = wpx[pl * w_plane + (size_t)kf_ * NT * 64];                    \
    }                                                                                                  \
  }
#pragma unroll
  for (int d = 0; d < NB - 1; ++d) SLU_WB_FETCH(d, d)

  // ---- stage the window: LDS (row, col) <- global element u0 + row * S_real + col (col < S_real), zero elsewhere; two
  //      adjacent columns per thread and step (one 4-byte LDS store per plane).  The round-3 loop spent ~80 VALU
  //      instructions per pair (index arithmetic through a float reciprocal, 64-bit addresses and bounds, select-based
  //      splits) and made the kernel VALU-bound: SQ counters of the Sinc launch showed 2 700 VALU instructions per wave beside
  //      390 MFMAs, VALU-active 12.7 k cycles against 6.2 k of MFMA.  Now: (row, pair) advance incrementally in 32-bit
  //      integers, one unsigned compare covers both bounds of an index, and the f16x2 split is the packed flush-mode form
  //      (two conversions + two fused multiply-adds per pair). ----
  if constexpr (NS == 2) f16_denorm_flush();
  {
    const int u0 = l0 * p.S_real - p.pad;            // (the launcher checks in_row < 2^31)
    const int half = p.S >> 1;                       // column pairs per row
    const int total = p.nrows * half;
    const unsigned in_row = (unsigned)p.in_row;
    const int drow = WB_THREADS / half, dcp = WB_THREADS - drow * half;
    // pairs per thread and batch: ALL of a 128-frame window (135 rows x 40 pairs / 256 threads = 21.1) in one round — 44 loads
    // in flight per thread before the first split (the accumulators are not live yet: the registers are free).  With 8 per
    // round a workgroup paid three dependent HBM round trips before its first MFMA (same box, Sinc launch of a 1024-sequence
    // super-batch on 128 CUs: DESIGN.md section 7)
    constexpr int U = 22;
    // PCM: the rows hold int16 samples (decided once, outside the loop: a branch inside it would split the batch of loads)
    auto stage = [&](auto PCM) {
      int row = tid / half, cp = tid - row * half;   // this thread's next pair; + WB_THREADS pairs per step
      for (int base = tid; base < total; base += WB_THREADS * U) {
        float v0[U], v1[U];
        int off[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const bool ok = base + j * WB_THREADS < total;
          const int col = 2 * cp;
          const int u = u0 + row * p.S_real + col;
          const bool ok0 = ok && col < p.S_real && (unsigned)u < in_row;
          const bool ok1 = ok && col + 1 < p.S_real && (unsigned)(u + 1) < in_row;
          off[j] = ok ? row * p.Sp + col : -1;
          if constexpr (decltype(PCM)::value) {
            v0[j] = ok0 ? (float)inb16[u] : 0.0f;
            v1[j] = ok1 ? (float)inb16[u + 1] : 0.0f;
          } else {
            v0[j] = ok0 ? inb[u] : 0.0f;
            v1[j] = ok1 ? inb[u + 1] : 0.0f;
          }
          cp += dcp; row += drow;
          if (cp >= half) { cp -= half; ++row; }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
          if (off[j] < 0) continue;
          float a0 = v0[j], a1 = v1[j];
          if constexpr (decltype(PCM)::value) { a0 *= p.in_scale; a1 *= p.in_scale; }
          unsigned short* dst = lds + off[j];
          if constexpr (NS == 2) {
            amx = max(amx, max(abs_bits(a0), abs_bits(a1)));
            unsigned hi, lo;
            split_f16x2_pair_flush(a0, a1, hi, lo);
            *reinterpret_cast<unsigned*>(dst) = hi;
            *reinterpret_cast<unsigned*>(dst + plane) = lo;
          } else if constexpr (NS == 3) {
            unsigned w[3];               // (round 6: the packed pair split — one v_cvt_pk_bf16_f32 per level)
            split_bf16x3_pair(a0, a1, w);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<unsigned*>(dst + pl * plane) = w[pl];
          } else {
            unsigned short a[NS], c[NS];
            split_terms<NS>(a0, a);
            split_terms<NS>(a1, c);
#pragma unroll
            for (int pl = 0; pl < NS; ++pl) *reinterpret_cast<unsigned*>(dst + pl * plane) = a[pl] | ((unsigned)c[pl] << 16);
          }
        }
      }
    };
    // Four elements per step (S_real, pad, in_row multiples of 4).  The pair loop above costs ~40 VALU instructions per
    // pair (two bounds per element, 64-bit addresses, conditional loads compiled to branches): SQ counters of the Sinc
    // launch showed 2 360 VALU instructions per wave beside 390 MFMAs, and on this chip the VALU and MFMA issue of the
    // two waves of a SIMD add up rather than overlap — the kernel was bound by VALU + MFMA issue.
    // Here the row is read through a buffer descriptor whose bounds check returns 0 outside [0, in_row): no bounds
    // arithmetic, one 16-byte (PCM16: 8-byte) load, two packed splits and one 8-byte LDS store per plane and quad.
    auto stage4 = [&](auto PCM) {
      constexpr bool pcm = decltype(PCM)::value;
      constexpr int ESZ = pcm ? 2 : 4;
      const unsigned long long gbase = pcm ? (unsigned long long)inb16 : (unsigned long long)inb;
      const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)gbase), bhi = __builtin_amdgcn_readfirstlane((unsigned)(gbase >> 32));
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(((unsigned long long)bhi << 32) | blo), 0, (int)(in_row * ESZ), 0x00020000);
      const int quads = p.S >> 2;
      const int totq = p.nrows * quads;
      const int drq = WB_THREADS / quads, dcq = WB_THREADS - drq * quads;
      int row = tid / quads, cq = tid - row * quads;
      constexpr int UQ = 11;                           // 135 rows x 20 quads / 256 threads = 10.5: one round for the Sinc window
      for (int base = tid; base < totq; base += WB_THREADS * UQ) {
        float v[UQ][4];
        int off[UQ];
#pragma unroll
        for (int j = 0; j < UQ; ++j) {
          const bool ok = base + j * WB_THREADS < totq;
          const int col = 4 * cq;
          const int u = u0 + row * p.S_real + col;
          // columns [S_real, S) are the channel padding, rows past the window nothing: an offset the descriptor rejects
          const int voff = (ok && col < p.S_real) ? u * ESZ : (int)0x80000000;
          off[j] = ok ? row * p.Sp + col : -1;
          if constexpr (pcm) {
            const auto w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 0);
            v[j][0] = (float)(short)(w[0] & 0xffffu); v[j][1] = (float)(short)(w[0] >> 16);
            v[j][2] = (float)(short)(w[1] & 0xffffu); v[j][3] = (float)(short)(w[1] >> 16);
          } else {
            const auto w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
            v[j][0] = __uint_as_float(w[0]); v[j][1] = __uint_as_float(w[1]);
            v[j][2] = __uint_as_float(w[2]); v[j][3] = __uint_as_float(w[3]);
          }
          cq += dcq; row += drq;
          if (cq >= quads) { cq -= quads; ++row; }
        }
#pragma unroll
        for (int j = 0; j < UQ; ++j) {
          if (off[j] < 0) continue;
          float a[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = pcm ? v[j][e] * p.in_scale : v[j][e];
          unsigned short* dst = lds + off[j];
          if constexpr (NS == 2) {
            amx = max(amx, max(max(abs_bits(a[0]), abs_bits(a[1])), max(abs_bits(a[2]), abs_bits(a[3]))));
            uint2 hi, lo;
            split_f16x2_pair_flush(a[0], a[1], hi.x, lo.x);
            split_f16x2_pair_flush(a[2], a[3], hi.y, lo.y);
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + plane) = lo;
          } else if constexpr (NS == 3) {
            unsigned w01[3], w23[3];
            split_bf16x3_pair(a[0], a[1], w01);
            split_bf16x3_pair(a[2], a[3], w23);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(dst + pl * plane) = make_uint2(w01[pl], w23[pl]);
          } else {
            unsigned short t[4][NS];
#pragma unroll
            for (int e = 0; e < 4; ++e) split_terms<NS>(a[e], t[e]);
#pragma unroll
            for (int pl = 0; pl < NS; ++pl)
              *reinterpret_cast<uint2*>(dst + pl * plane) = make_uint2(t[0][pl] | ((unsigned)t[1][pl] << 16), t[2][pl] | ((unsigned)t[3][pl] << 16));
          }
        }
      }
    };
    if (p.vec4) { if (p.pcm16) stage4(std::true_type{}); else stage4(std::false_type{}); }
    else if (p.pcm16) stage(std::true_type{}); else stage(std::false_type{});
  }
  if constexpr (NS == 2) f16_denorm_keep();          // the epilogue's plane output follows the default-mode rule (slu_bf16.h)
  __syncthreads();

  f32x4 accs[SP::NACC][RT][CT];
#pragma unroll
  for (int a = 0; a < SP::NACC; ++a)
#pragma unroll
    for (int m = 0; m < RT; ++m)
#pragma unroll
      for (int n = 0; n < CT; ++n) accs[a][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int XTA = XT > 0 ? XT : 1;
  f32x4 accx[SP::NACC][XTA];                      // COLS: the shared column tile (NT - 1), local row tiles 0 .. XT-1
#pragma unroll
  for (int a = 0; a < SP::NACC; ++a)
#pragma unroll
    for (int m = 0; m < XTA; ++m) accx[a][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int i = lane & 15, kg = lane >> 4;
  // tap chunk (kc, kg) starts at padded tap q = kc*32 + kg*8: LDS row offset q / S, column q % S (multiple of 8)
  int qd = (kg * 8) / p.S, qm = kg * 8 - qd * p.S;
  int abase[RT];
#pragma unroll
  for (int m = 0; m < RT; ++m) abase[m] = (row0 + rtile(m) * 16 + i) * p.Sp;
  // (the first NB - 1 chunks were requested before the window was staged; every load of the staging has been waited
  // for since — saying so keeps the compiler from merging "loads pending at loop entry" into a vmcnt(0) at the top of
  // every trip through the loop)
  __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
  for (int kc0 = 0; kc0 < p.KC; kc0 += NB) {
#pragma unroll
    for (int sb = 0; sb < NB; ++sb) {
      if (kc0 + sb >= p.KC) break;                 // uniform
      SLU_WB_FETCH((sb + NB - 1) % NB, kc0 + sb + NB - 1)
      uint4 fa[NS][RT];
      const int aoff = qd * p.Sp + qm;
#pragma unroll
      for (int pl = 0; pl < NS; ++pl)
#pragma unroll
        for (int m = 0; m < RT; ++m) fa[pl][m] = *reinterpret_cast<const uint4*>(lds + pl * plane + abase[m] + aoff);
      __builtin_amdgcn_sched_barrier(0);           // the filter request and this chunk's fragments are issued HERE
#pragma unroll
      for (int q = 0; q < SP::NPAIR; ++q) {
#pragma unroll
        for (int m = 0; m < RT; ++m)
#pragma unroll
          for (int n = 0; n < CT; ++n)
            accs[SP::ACC(q)][m][n] = mfma_split<NS>(fa[SP::PA(q)][m], fb[sb][SP::PB(q)][n], accs[SP::ACC(q)][m][n]);
        if constexpr (COLS) {
#pragma unroll
          for (int m = 0; m < XT; ++m)
            accx[SP::ACC(q)][m] = mfma_split<NS>(fa[SP::PA(q)][m], fx[sb][SP::PB(q)], accx[SP::ACC(q)][m]);
        }
      }
      qm += 32;
      while (qm >= p.S) { qm -= p.S; ++qd; }
    }
  }
#undef SLU_WB_FETCH

  // tiles of this wave: (local row tile m, column slot n < CT) and, COLS, (local row tile m < XT, the shared column)
  constexpr int NTILE = RT * CT + XT;
  f32x4 acc[NTILE];
#pragma unroll
  for (int m = 0; m < RT; ++m)
#pragma unroll
    for (int n = 0; n < CT; ++n) acc[m * CT + n] = split_result<NS>(accs[0][m][n], accs[SP::NACC - 1][m][n]);
#pragma unroll
  for (int m = 0; m < XT; ++m) acc[RT * CT + m] = split_result<NS>(accx[0][m], accx[SP::NACC - 1][m]);

  // ---- epilogue: bias, abs, max-pool over frame pairs, LeakyReLU, strided store (as wconv_fwd_kernel); row bases in 64
  //      bits once, 32-bit offsets inside (the launcher checks l_out * out_sl + c_out < 2^31) ----
  float* __restrict__ outb = p.out ? p.out + (size_t)b * p.out_sb : nullptr;
  unsigned char* __restrict__ routeb = p.route ? p.route + (size_t)b * p.l_out * p.c_out : nullptr;
  const int osl = (int)p.out_sl;
  if (p.stage_out) {
    // ---- LDS-staged epilogue.  Written straight from the accumulator layout, a store instruction covers 4 rows x 16
    //      channels: 64-byte (fp32) or 32-byte (plane) segments, 20-40 store instructions per lane — probe builds put the
    //      Sinc launch's stores at 50 of its 270 us (tools/build_wconv_probe.sh, profiles/r04_r_wconv_probe_bits.txt).  Here
    //      the tile is assembled in the window's LDS (every wave is done with it after the tap loop) and leaves as whole
    //      rows, 16 bytes per lane. ----
    __syncthreads();
    if (p.planes) {
      const int LDP = p.Kp_out + 8;                                      // plane row stride in LDS (elements; 16-byte rows)
#pragma unroll
      for (int tile = 0; tile < NTILE; ++tile) {
        const bool shared = tile >= RT * CT;
        const int m = shared ? tile - RT * CT : tile / CT, n = shared ? 0 : tile % CT;
        const int lr0 = row0 + rtile(m) * 16 + 4 * kg;                   // local frame of acc[tile][0]
        const int c = (shared ? NT - 1 : nb + n) * 16 + i;
        if (c >= p.Kp_out) continue;
        const bool real = c < p.c_out;
        const float bias = (real && p.bias) ? p.bias[c] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = acc[tile][r] + bias;
          t = p.do_abs ? fabsf(t) : t;
          t = real ? (t > 0.0f ? t : t * p.slope) : 0.0f;
          if (l0 + lr0 + r >= p.l_conv) t = 0.0f;                       // rows past the end: never stored, keep them out of the guard
          unsigned short sp[NS];
          if constexpr (NS == 2) amx = max(amx, abs_bits(t));
          split_terms<NS>(t, sp);
#pragma unroll
          for (int pl = 0; pl < NS; ++pl) lds[(pl * F + lr0 + r) * LDP + c] = sp[pl];
        }
      }
      __syncthreads();
      const int cpr = p.Kp_out >> 3;                                     // 16-byte chunks per row
      const int per_plane = F * cpr;
      for (int q = tid; q < NS * per_plane; q += WB_THREADS) {
        const int pl = q / per_plane, rem = q - pl * per_plane;
        const int lr = rem / cpr, ch = rem - lr * cpr;
        const int f = l0 + lr;
        if (f >= p.l_conv) continue;
        *reinterpret_cast<uint4*>(p.planes + (size_t)pl * p.plane + ((size_t)f * p.Bn + b) * p.Kp_out + 8 * ch) =
            *reinterpret_cast<const uint4*>(lds + (pl * F + lr) * LDP + 8 * ch);
      }
    } else {
      float* so = reinterpret_cast<float*>(smem);                        // [F / pool][c_out + 4]
      const int LDO = p.c_out + 4;
      const bool full = l0 + F <= p.l_conv;                              // uniform: no partial pooling window in this tile
      // |x| >= 0: LeakyReLU is the identity after Abs (0 * slope = 0, NaN stays NaN) — skipped, bit for bit the same
      const bool leaky = !p.do_abs;
      // this lane's element of a 16 x 16 tile: rows 4 kg + r, column i (pooled: rows 2 kg + h)
      const int lane_off = (p.pool == 2 ? 2 * kg : 4 * kg) * LDO + i;
#pragma unroll
      for (int tile = 0; tile < NTILE; ++tile) {
        const bool shared = tile >= RT * CT;
        const int m = shared ? tile - RT * CT : tile / CT, n = shared ? 0 : tile % CT;
        const int ut = row0 + rtile(m) * 16;                             // wave-uniform: first local frame of the tile
        const int ct = (shared ? NT - 1 : nb + n) * 16;                  // wave-uniform: first channel of the tile
        if (ct + i >= p.c_out) continue;
        const float bias = p.bias ? p.bias[ct + i] : 0.0f;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = acc[tile][r] + bias;
          v[r] = p.do_abs ? fabsf(t) : t;
        }
        if (p.pool == 2) {
          float* dst = so + (ut >> 1) * LDO + ct + lane_off;
          float o[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            // ceil_mode: the last window of a sequence may hold one frame (a tile past l_conv stores nothing: read-out skips it)
            const bool has1 = full || (l0 + ut + 4 * kg + 2 * h + 1) < p.l_conv;
            const float pooled = (has1 && v[2 * h + 1] > v[2 * h]) ? v[2 * h + 1] : v[2 * h];
            o[h] = leaky ? (pooled > 0.0f ? pooled : pooled * p.slope) : pooled;
          }
          dst[0] = o[0];
          dst[LDO] = o[1];
        } else {
          float* dst = so + ut * LDO + ct + lane_off;
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[r * LDO] = leaky ? (v[r] > 0.0f ? v[r] : v[r] * p.slope) : v[r];
        }
      }
      __syncthreads();
      const int cpr = p.c_out >> 2;                                      // float4 chunks per row
      const int rows = F / p.pool, o0 = l0 / p.pool;                     // l0 is a multiple of 64
      for (int q = tid; q < rows * cpr; q += WB_THREADS) {
        const int lr = q / cpr, ch = q - lr * cpr;
        if (o0 + lr >= p.l_out) continue;
        *reinterpret_cast<float4*>(outb + (size_t)(o0 + lr) * osl + 4 * ch) = *reinterpret_cast<const float4*>(so + lr * LDO + 4 * ch);
      }
    }
    if constexpr (NS == 2) { if (p.amax) amax_publish(p.amax, amx); }
    return;
  }
#pragma unroll
  for (int tile = 0; tile < NTILE; ++tile) {
    {
      const bool shared = tile >= RT * CT;                              // COLS: a tile of the shared column
      const int m = shared ? tile - RT * CT : tile / CT, n = shared ? 0 : tile % CT;
      const int fbase = l0 + row0 + rtile(m) * 16 + 4 * kg;             // multiple of 4
      const int c = (shared ? NT - 1 : nb + n) * 16 + i;
      if (p.planes) {
        // straight into the split format (pool == 1 only): columns [c_out, Kp_out) are the zero padding
        if (c >= p.Kp_out) continue;
        const bool real = c < p.c_out;
        const float bias = (real && p.bias) ? p.bias[c] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = fbase + r;
          if (f >= p.l_conv) continue;
          float t = acc[tile][r] + bias;
          t = p.do_abs ? fabsf(t) : t;
          t = real ? (t > 0.0f ? t : t * p.slope) : 0.0f;
          unsigned short sp[NS];
          if constexpr (NS == 2) amx = max(amx, abs_bits(t));
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
        const float t = acc[tile][r] + bias;
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
          outb[(f0 >> 1) * osl + c] = pooled > 0.0f ? pooled : pooled * p.slope;
          if (routeb) {
            const bool sgn = pick1 ? neg[2 * h + 1] : neg[2 * h];
            routeb[(f0 >> 1) * p.c_out + c] = (unsigned char)((pick1 ? 1 : 0) | (sgn ? 2 : 0));
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = fbase + r;
          if (f >= p.l_conv) continue;
          outb[f * osl + c] = v[r] > 0.0f ? v[r] : v[r] * p.slope;
          if (routeb) routeb[f * p.c_out + c] = (unsigned char)(neg[r] ? 2 : 0);
        }
      }
    }
  }
  if constexpr (NS == 2) { if (p.amax) amax_publish(p.amax, amx); }
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
static int bf_launch(dim3 grid, size_t lds, hipStream_t st, const WconvBfParams& p) {
  // 2 x 2 waves where the channel tiles divide; column ownership for the five-tile (Sinc, 80 filters) launch
  constexpr int SPLITN = ((NT % 2 == 0) && MT == 2) ? 1 : ((NT == 5 && MT == 2) ? 2 : 0);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)wconv_bf_fwd_kernel<MT, NT, NS, SPLITN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "wconv_bf16: cannot raise the dynamic LDS cap to %zu: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL((wconv_bf_fwd_kernel<MT, NT, NS, SPLITN>), grid, dim3(WB_THREADS), lds, st, p);
  SLU_CHECK_LAUNCH("wconv_bf_fwd_kernel");
  return SLU_OK;
}

template <int NS>
static int bf_launch_nt(int MT, int NT, dim3 grid, size_t lds, hipStream_t st, const WconvBfParams& p) {
#define SLU_BF_CASE(M_, N_) if (MT == M_ && NT == N_) return bf_launch<M_, N_, NS>(grid, lds, st, p);
  SLU_BF_CASE(1, 1) SLU_BF_CASE(1, 2) SLU_BF_CASE(1, 4) SLU_BF_CASE(1, 5) SLU_BF_CASE(1, 8)
  SLU_BF_CASE(2, 1) SLU_BF_CASE(2, 2) SLU_BF_CASE(2, 4) SLU_BF_CASE(2, 5) SLU_BF_CASE(2, 8)
#undef SLU_BF_CASE
  SLU_FAIL(SLU_ERR_UNSUPPORTED, "wconv_bf16: unsupported tile configuration %d x %d", MT, NT);
}

}  // namespace slu

using namespace slu;

static int64_t bf_c_pad(int64_t c_in) { return c_in == 1 ? 1 : cdiv(c_in, 8) * 8; }

extern "C" size_t slu_wconv_bf16_workspace_bytes(int64_t c_out, int64_t c_in, int64_t k_t, int nsplit) {
  const int64_t Kw = k_t * bf_c_pad(c_in);
  const int64_t nt = cdiv(c_out, 16) <= 8 ? 8 : cdiv(c_out, 16);
  return (size_t)nsplit * cdiv(Kw, 32) * nt * 64 * sizeof(uint4);
}

extern "C" int slu_wconv_fwd_bf16(const float* in, const float* const* in_table, int64_t table_rows,
                                  const float* weight, const float* bias, float* out, uint8_t* route, int64_t B,
                                  int64_t l_in, int64_t c_in, int64_t c_out, int64_t k_t, int64_t stride_t,
                                  int do_abs, int pool, float slope, int64_t out_sb, int64_t out_sl,
                                  void* out_planes, int64_t out_plane_stride,
                                  void* workspace, size_t workspace_bytes, int packed_valid, int nsplit,
                                  uint32_t* absmax_word, int in_pcm16, float in_scale, void* stream) {
  SLU_REQUIRE((in || in_table) && weight && (out || out_planes), "slu_wconv_fwd_bf16: null pointer");
  SLU_REQUIRE(!in_table || (table_rows >= 1 && table_rows <= B), "slu_wconv_fwd_bf16: bad table_rows");
  SLU_REQUIRE(B > 0 && l_in > 0 && c_in > 0 && c_out > 0 && k_t > 0 && stride_t > 0, "slu_wconv_fwd_bf16: non-positive size");
  SLU_REQUIRE(pool == 1 || pool == 2, "slu_wconv_fwd_bf16: pool must be 1 or 2 (got %d)", pool);
  SLU_REQUIRE(nsplit >= 1 && nsplit <= 3, "slu_wconv_fwd_bf16: nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  SLU_REQUIRE(B <= 65535, "slu_wconv_fwd_bf16: B must be <= 65535");
  SLU_REQUIRE(l_in * c_in < (1LL << 31) - 2, "slu_wconv_fwd_bf16: a row of l_in * c_in elements must fit 32-bit indexing");
  const int64_t c_pad = bf_c_pad(c_in);
  const int64_t S = stride_t * c_pad, S_real = stride_t * c_in;
  if (S % 8 != 0 || (c_in > 1 && stride_t != 1))
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_wconv_fwd_bf16: needs stride * channels %% 8 == 0 (stride 1 for multi-channel "
             "inputs); got stride %lld, c_in %lld", (long long)stride_t, (long long)c_in);
  const int NT = bf_nt_for(c_out);
  if (NT < 0) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_wconv_fwd_bf16: at most 128 output channels (got %lld)", (long long)c_out);
  const int64_t Kw = k_t * c_pad, KC = cdiv(Kw, 32);
  const size_t need = (size_t)nsplit * KC * NT * 64 * sizeof(uint4);
  if (!workspace || workspace_bytes < need)
    SLU_FAIL(SLU_ERR_WORKSPACE, "slu_wconv_fwd_bf16: workspace too small (%zu < %zu)", workspace_bytes, need);
  const int64_t pad_t = k_t / 2;
  const int64_t l_conv = (l_in + 2 * pad_t - k_t) / stride_t + 1;
  SLU_REQUIRE(l_conv > 0, "slu_wconv_fwd_bf16: input shorter than the filter");
  hipStream_t st = (hipStream_t)stream;
  uint4* wp = reinterpret_cast<uint4*>(workspace);
  if (!packed_valid) {     // else: the workspace still holds the pack of these very filters (frozen block, caller's cache)
    const int total = (int)(KC * NT * 64);
#define SLU_WPACK(NS_) hipLaunchKernelGGL(bf_wconv_pack_kernel<NS_>, dim3((total + 255) / 256), dim3(256), 0, st, weight, wp, \
                                          (int)c_out, (int)c_in, (int)c_pad, (int)k_t, NT, (int)KC)
    if (nsplit == 3) SLU_WPACK(3); else if (nsplit == 2) SLU_WPACK(2); else SLU_WPACK(1);
#undef SLU_WPACK
    SLU_CHECK_LAUNCH("bf_wconv_pack_kernel");
  }
  WconvBfParams p;
  p.in = in; p.wp = wp; p.bias = bias; p.out = out; p.route = route;
  SLU_REQUIRE(!route || (out && !out_planes), "slu_wconv_fwd_bf16: route goes with the fp32 output");
  p.in_tab = in_table; p.tab_rows = (int)(in_table ? table_rows : 1);
  SLU_REQUIRE(!in_pcm16 || (c_in == 1 && !route), "slu_wconv_fwd_bf16: PCM16 input is the waveform of a frozen first block (c_in == 1)");
  p.pcm16 = in_pcm16 ? 1 : 0; p.in_scale = in_pcm16 ? in_scale : 1.0f;
  p.planes = (unsigned short*)out_planes; p.plane = out_plane_stride; p.Kp_out = (int)(cdiv(c_out, 32) * 32); p.Bn = (int)B;
  if (out_planes) {
    if (pool != 1 || NT * 16 < p.Kp_out)
      SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_wconv_fwd_bf16: out_planes needs pool == 1 and a channel tiling that covers "
               "round_up(c_out, 32) columns (c_out %lld)", (long long)c_out);
    SLU_REQUIRE(out_plane_stride >= (int64_t)((l_in + 2 * (k_t / 2) - k_t) / stride_t + 1) * B * p.Kp_out,
                "slu_wconv_fwd_bf16: plane stride too small");
  }
  p.in_row = l_in * c_in; p.out_sb = out_sb; p.out_sl = out_sl;
  SLU_REQUIRE(out_planes || (cdiv(l_conv, pool) + 1) * out_sl + c_out < (1LL << 31), "slu_wconv_fwd_bf16: output row offsets must fit 32 bits");
  // LDS row stride: the smallest Sp >= S with Sp = 16 (mod 32) elements.  ds_read_b128 serves a wave in four groups of
  // sixteen lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) — and a fragment read has
  // lane (i, kg) at row i, 16-byte column kg: the sixteen lanes of a group hit sixteen distinct bank quads exactly when
  // the row stride is 8 (mod 16) dwords (tools/lds_stride_conflicts.py enumerates them).  Round 3's "S + 8" gave 44
  // (Sinc, conv1) and 36 (conv2) dwords: 6 and 14 of every 32 lanes collided — SQ_LDS_BANK_CONFLICT was 47 % of the
  // LDS-active cycles of the Sinc launch.
  p.S = (int)S; p.S_real = (int)S_real; p.Sp = (int)(S + ((16 - S % 32) + 32) % 32);
  p.KC = (int)KC; p.pad = (int)(pad_t * c_in);
  p.l_conv = (int)l_conv; p.l_out = (int)cdiv(l_conv, pool); p.c_out = (int)c_out;
  p.do_abs = do_abs; p.pool = pool; p.slope = slope;
  p.amax = nsplit == 2 ? absmax_word : nullptr;
  // plane offset (nrows * Sp elements) and row starts (Sp elements) keep the 8-byte LDS stores aligned: Sp % 16 == 0
  p.vec4 = (S_real % 4 == 0 && (pad_t * c_in) % 4 == 0 && (l_in * c_in) % 4 == 0 && l_in * c_in * 4 < (1LL << 31)
            && (in_table || (uintptr_t)in % (in_pcm16 ? 8 : 16) == 0)) ? 1 : 0;
  int MT = (B * cdiv(l_conv, 128) >= 256) ? 2 : 1;
  int F = 64 * MT;
  p.nrows = F + (int)cdiv(KC * 32, S) + 1;
  size_t lds = (size_t)nsplit * p.nrows * p.Sp * sizeof(unsigned short);
  if (lds > 160 * 1024 && MT == 2) {           // long hops (stride ~200 and up at three planes): 64-frame tiles still fit
    MT = 1; F = 64;
    p.nrows = F + (int)cdiv(KC * 32, S) + 1;
    lds = (size_t)nsplit * p.nrows * p.Sp * sizeof(unsigned short);
  }
  if (lds > 160 * 1024) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_wconv_fwd_bf16: window of %zu bytes exceeds the 160 KiB LDS", lds);
  // LDS-staged epilogue (whole-row 16-byte stores) where the output allows it: no route bytes, 16-byte aligned rows
  {
    size_t tile = 0;
    bool ok = false;
    if (out_planes) {
      ok = (out_plane_stride % 8 == 0) && ((uintptr_t)out_planes % 16 == 0);
      tile = (size_t)nsplit * F * (p.Kp_out + 8) * sizeof(unsigned short);
    } else {
      ok = !route && c_out % 4 == 0 && out_sl % 4 == 0 && out_sb % 4 == 0 && ((uintptr_t)out % 16 == 0);
      tile = (size_t)(F / pool) * (c_out + 4) * sizeof(float);
    }
    p.stage_out = (ok && tile <= 64 * 1024) ? 1 : 0;
    if (p.stage_out && tile > lds) lds = tile;
  }
  dim3 grid((unsigned)cdiv(l_conv, F), (unsigned)B);
  return nsplit == 3 ? bf_launch_nt<3>(MT, NT, grid, lds, st, p)
       : nsplit == 2 ? bf_launch_nt<2>(MT, NT, grid, lds, st, p) : bf_launch_nt<1>(MT, NT, grid, lds, st, p);
}
