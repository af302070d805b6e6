// Split-precision persistent GRU recurrence (forward) for FROZEN layers and for the bf16 forward of trainable ones:
// torch.nn.GRU semantics (models.py:232/:262: h0 = 0, gates [r; z; n], optional reverse direction), the
// hidden x hidden contraction on v_mfma_f32_16x16x32_{f16,bf16} with W_hh and h_{t-1} each split into NS 16-bit
// terms (slu_bf16.h): NS = 3 (bf16x3) keeps six products (fp32-class result at 6/16 of the fp32-MFMA cycles),
// NS = 2 (f16x2: two fp16 terms) three products (fp32-class at 3/16), NS = 1 is plain bf16 (BASELINE configs[4]).
//
// Geometry: grid (16-sequence tile) x (direction), H/16 waves, wave w owns hidden units [16w, 16w+16) of all three
// gates; its W_hh slice — 3 gates x H/32 k-chunks x NS planes of eight 16-bit terms per lane (96 VGPRs for H = 128 on
// f16x2) — is split once at kernel start and stays RESIDENT for all T steps; h_{t-1} lives in LDS as NS planes
// (double buffered, 16-byte slots XOR-swizzled by the row), in fp32 in the owning lane for the blend.
//
// gru_bf_fwd_kernel (round 4) computes the TRANSPOSED product per step — A = the wave's W_hh slice (rows = hidden
// units), B = h_{t-1}^T (columns = sequences) — so that the MFMA's C layout hands a lane FOUR CONSECUTIVE hidden units of
// ONE sequence (round 3: one unit of four sequences).  Same registers, same LDS reads, same products in the same order,
// but everything around the MFMAs shrinks: gx arrives as three 16-byte loads per lane and step
// (were twelve 4-byte loads), the output leaves as one 16-byte store (were four), the split h goes to LDS as one 8-byte
// store per plane (were four 2-byte stores), and the step's barrier waits for the LDS only (lds_barrier: the round-3
// __syncthreads also drained the global stores of the step).  A lane's four units are exactly one Philox block of the
// layer's dropout mask and the two frames of an average-pooling window arrive at the same lane on consecutive steps, so
// the Dropout + Downsample(avg, 2) that follow the layer (models.py:246-251, 26-46) are applied HERE (EPI 1 / 2): the lane
// keeps the masked h of a window's first frame and writes the pooled value — as the next frozen layer's 16-bit planes or
// as fp32 — when the second one arrives.  The fp32 (T, B, D*H) output, its re-read and the dropout_pool launch are gone.
// The keep bits come from slu_dropout_bits (one bit per element, drawn with the element -> counter map of
// dropout_pool_fwd4_kernel: the fused and the two-launch paths produce identical bits).
#include "slu_bf16.h"
#include <type_traits>

namespace slu {

__device__ __forceinline__ float bf_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float bf_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// Workgroup barrier that orders LDS traffic only: the step's global loads (next step's gx, prefetched) and stores (this
// step's output) stay in flight across it.  __syncthreads() = workgroup fence + s_barrier waits for vmcnt(0) as well,
// i.e. for the write acknowledgement of the stores issued a few instructions earlier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Separately ROUNDED product / sum: the operations carry no `contract` flag, so the backend cannot fuse them into an fma
// (HIP compiles with -ffp-contract=fast, and __fmul_rn / __fadd_rn are plain x * y / x + y of a header compiled under it:
// whether "h * scale" and the pooling window's addition fused depended on the surrounding code — round 6 saw the forward
// direction fuse them after an unrelated change to the mask arithmetic: 1-ulp differences from the two-launch path for
// every dropout scale that is not a power of two, caught by test_gru_dropout_pool_epilogue_equals_the_two_launch_path).
__device__ __forceinline__ float mul_rnd(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rnd(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}

struct GruBfParams {
  const float* gx;        // (T, B, D*3H); unused by the fused-input kernels
  // fused input projection (KI > 0: K <= 32 KI input channels): x as NS planes of (T*B) x (32 KI) 16-bit terms
  // (plane stride x_plane elements), W_ih packed by gemm_bf_pack_kernel for N = D*3H columns, b_ih (D*3H)
  const unsigned short* xp; long long x_plane;
  const uint4* wih; const float* b_ih;
  const float* w_hh[2];   // (3H, H) fp32
  const float* b_hh[2];   // (3H)
  float* out;             // EPI 0: (T, B, D*H); EPI 2: the pooled (ceil(T/2), B, D*H)
  float* reserve;         // gru_bf_fwd_rs_kernel only: the saved gates in gru_seq_bwd_kernel's layout [D][T][NBT][NW][5][64][4]
  unsigned short* planes; long long plane;   // EPI 1: NS planes of (ceil(T/2) * B) x (D*H) 16-bit terms
  const unsigned* keep;   // EPI > 0: keep bits (T, B, D*H/32), bit c % 32 of word c / 32 = element (t, b, c); null: no dropout
  float keep_scale;       // 1 / (1 - p)
  int T, B, D;
};

// ---------------------------------------------------------------------------------------------------------------------
// EPI: 0 = fp32 output of every step; 1 = Dropout + avg-pool(2, ceil) -> NS planes; 2 = the same -> fp32.
// KI > 0: the input projection x_t W_ih^T + b_ih is computed HERE instead of being read as gx — for layers whose input has
// at most 32 KI channels (the first GRU layer: K = 60) the wave's W_ih slice fits beside W_hh (3 gates x KI chunks x NS
// planes = 48 registers for KI = 2 on f16x2), the fragments of x_t are read straight from the previous stage's planes (one
// 16-byte load per chunk and plane, a step ahead), and the extra MFMAs do not depend on h_{t-1}.  Saves the projection
// GEMM and the fp32 gx round trip (T*B*D*3H*8 bytes) of that layer.
//
// Global memory goes through LDS (measured).  In the compute layout a lane owns 16 bytes of ONE
// sequence's row and its neighbours own other sequences (rows 1 - 3 KB apart): a global_load_dwordx4 / store in that
// layout is 64 separate 16-byte requests touching 16 lines four times each.  Timing probes on the first version of this
// kernel (1024 sequences, T = 300, 128 CUs): 1.41 us per step; 1.02 without the gx loads; 1.13 without the output store —
// the compute core was already below round 3's 1.29 us and the scattered accesses gave it all back.  So a wave moves
// WHOLE ROWS: rows 16/NW w .. of the tile, consecutive lanes = consecutive 16 bytes (a half-wave = one 512-byte output row,
// 1.5 waves = one 1536-byte gx row), and the transposition between the two layouts happens in padded, conflict-free LDS
// staging tiles: gx of step s+1 is requested at the top of step s, written to LDS at its end (a full step later: the wait
// is free) and read in the compute layout at the top of step s+1; the output of step s is written to LDS in the compute
// layout, read back row-wise after the barrier and stored at the top of step s+1.
//
// Order of a step:
//   A. gx (from LDS) + b_hh seed the MFMA chains (r, z) — the bias additions leave the dependent tail of the step;
//   B. next step's gx rows / x fragments / keep bits are requested; the previous step's output rows are stored;
//   C. h fragments from LDS, 36 (f16x2) MFMAs, gate math, split h to LDS, output to its staging tile;
//   D. the gx rows requested in B go to their staging tile; barrier (LDS only).
// Gate arithmetic (fp32-class, not the operation order of round 3): the blend is n + z (h - n), the f16x2 split of h uses
// split_f16x2_pair_flush.  (Measured and not kept, DESIGN.md section 7: gate-outer MFMA order with the r / z gate arithmetic
// interleaved between the remaining MFMAs; static wave priorities.)
template <int H, int NS, int KI, int EPI>
struct GruLds {
  static constexpr int ROWB = H * 2;                       // bytes per h row (one sequence, one plane)
  static constexpr int GXROW = 3 * H + 8;                  // floats per staged gx row (+ 32 B: conflict-free b128 reads)
  static constexpr int OROW = H + 4;                       // floats per staged fp32 output row
  static constexpr int PROWB = H * 2 + 16;                 // bytes per staged plane row
  static constexpr int HBUF = 2 * NS * 16 * ROWB;
  static constexpr int BIAS = KI > 0 ? 2 * 3 * H * 4 : 0;
  static constexpr int GXS = KI > 0 ? 0 : 2 * 16 * GXROW * 4;
  static constexpr int OST = EPI == 1 ? 2 * NS * 16 * PROWB : 2 * 16 * OROW * 4;
  static constexpr int BYTES = HBUF + BIAS + GXS + OST;
};

// REV (round 6): the direction is a template parameter of the body — blockIdx.y picks the instantiation —, so that the ~20
// selects per step on the (workgroup-uniform) direction in the Dropout + pooling epilogue and the time arithmetic are
// resolved at compile time: the loop is bound by instruction issue, every VALU slot counts (same arithmetic, same bits).
template <int H, int NS, int KI, int EPI, bool REV>
__device__ __forceinline__ void gru_bf_fwd_body(const GruBfParams& p) {
  constexpr int NW = H / 16;          // waves
  constexpr int KC = H / 32;          // 32-wide k-chunks
  constexpr int SLOTS = H / 8;        // 16-byte slots per h row
  constexpr bool BIAS_LDS = KI > 0;   // the fused kernel keeps its 24 bias values in LDS (register budget)
  typedef Split<NS> SP;
  typedef GruLds<H, NS, KI, EPI> LD;
  constexpr int ROWB = LD::ROWB, GXROW = LD::GXROW, OROW = LD::OROW, PROWB = LD::PROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const hbuf = smem;                                                   // [2][NS][16 * ROWB]
  float* const bias_s = reinterpret_cast<float*>(smem + LD::HBUF);                    // [2][3H] (KI > 0)
  float* const gxs = reinterpret_cast<float*>(smem + LD::HBUF + LD::BIAS);            // [2][16][GXROW] (KI == 0)
  unsigned char* const ost = smem + LD::HBUF + LD::BIAS + LD::GXS;                    // output staging, see LD::OST
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kg = lane >> 4;      // compute layout: i = sequence of the tile (MFMA column), kg = unit quad
  constexpr int dir = REV ? 1 : 0;
  const int b0 = blockIdx.x * 16;
  const int u0 = w * 16 + kg * 4;               // first of this lane's four hidden units
  const int T = p.T, B = p.B, D = p.D;
  const int seq = b0 + i;
  const int seqc = seq < B ? seq : B - 1;       // rows past B repeat the last sequence and are never stored

  if constexpr (NS == 2) f16_denorm_flush();

  // resident W_hh fragments (MFMA A operand, row = unit 16w + i): wb[g][c][pl] = 8 terms of
  // W_hh[g*H + 16w + i][c*32 + kg*8 .. +7], plane pl
  uint4 wb[3][KC][NS];
  {
    const float* __restrict__ W = p.w_hh[dir];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float* src = W + (size_t)(g * H + w * 16 + i) * H + c * 32 + kg * 8;
        const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        unsigned short s[8][NS];
#pragma unroll
        for (int e = 0; e < 8; ++e) split_terms<NS>(v[e], s[e]);
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) {
          uint4 o;
          o.x = s[0][pl] | ((unsigned)s[1][pl] << 16); o.y = s[2][pl] | ((unsigned)s[3][pl] << 16);
          o.z = s[4][pl] | ((unsigned)s[5][pl] << 16); o.w = s[6][pl] | ((unsigned)s[7][pl] << 16);
          wb[g][c][pl] = o;
        }
      }
  }
  // biases of this lane's four units: registers, or (fused kernel) LDS rows re-read every step
  float bh[3][4];
  if constexpr (BIAS_LDS) {
    for (int x = tid; x < 3 * H; x += H * 4) {
      bias_s[x] = p.b_hh[dir][x];
      bias_s[3 * H + x] = p.b_ih[(size_t)dir * 3 * H + x];
    }
  } else {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(p.b_hh[dir] + g * H + u0);
      bh[g][0] = v.x; bh[g][1] = v.y; bh[g][2] = v.z; bh[g][3] = v.w;
    }
  }
  // fused input projection: the wave's W_ih fragments (tile dir * 3H/16 + g * H/16 + w of the packed matrix)
  constexpr int KIA = KI > 0 ? KI : 1;
  uint4 wi[3][KIA][NS];
  if constexpr (KI > 0) {
    const int NTI = p.D * 3 * NW;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < KI; ++c)
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
          wi[g][c][pl] = p.wih[(size_t)pl * KI * NTI * 64 + ((size_t)c * NTI + dir * 3 * NW + g * NW + w) * 64 + lane];
  }
  // x_t W_ih^T + b_ih for this lane's (sequence, four units) from the fragments of x_t: the accumulation order
  // of gemm_bf_panel_kernel (k-chunks outside, products inside), i.e. the gx the GEMM would write
  auto xproj = [&](const uint4 (&xa)[KIA][NS], float (&o)[3][4]) {
    f32x4 ax[SP::NACC][3];
#pragma unroll
    for (int a = 0; a < SP::NACC; ++a)
#pragma unroll
      for (int g = 0; g < 3; ++g) ax[a][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KIA; ++c)
#pragma unroll
      for (int q = 0; q < SP::NPAIR; ++q)
#pragma unroll
        for (int g = 0; g < 3; ++g)
          ax[SP::ACC(q)][g] = mfma_split<NS>(wi[g][c][SP::PB(q)], xa[c][SP::PA(q)], ax[SP::ACC(q)][g]);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const f32x4 v = split_result<NS>(ax[0][g], ax[SP::NACC - 1][g]);
      const float4 bi = *reinterpret_cast<const float4*>(&bias_s[BIAS_LDS ? 3 * H + g * H + u0 : 0]);
      o[g][0] = v[0] + bi.x; o[g][1] = v[1] + bi.y; o[g][2] = v[2] + bi.z; o[g][3] = v[3] + bi.w;
    }
  };
  // fragment of x_t for this lane (MFMA B operand): column = sequence b0 + i (clamped), k slice kg of chunk c, plane pl
  auto xload = [&](int t_, uint4 (&xa)[KIA][NS]) {
    const unsigned short* base = p.xp + ((size_t)t_ * B + seqc) * (32 * KIA) + kg * 8;
#pragma unroll
    for (int c = 0; c < KIA; ++c)
#pragma unroll
      for (int pl = 0; pl < NS; ++pl)
        xa[c][pl] = *reinterpret_cast<const uint4*>(base + (size_t)pl * p.x_plane + c * 32);
  };

  // ---- row-wise (coalesced) side of the staging tiles: a wave moves rows RW w .. RW w + RW - 1 of the 16 ----
  constexpr int RW = 16 / NW;                   // rows (sequences) per wave
  // gx: 3H floats = 3H/4 lanes per row, RW rows = 192 lanes = three 16-byte loads per lane (any H)
  constexpr int GLR = 3 * H / 4;                // lanes per gx row
  // (plain scalars, not arrays handed to lambdas: those were demoted to scratch memory)
  auto gx_g = [&](int j) { const int f = j * 64 + lane; return min(b0 + RW * w + f / GLR, B - 1) * D * 3 * H + dir * 3 * H + (f % GLR) * 4; };
  auto gx_s = [&](int j) { const int f = j * 64 + lane; return (RW * w + f / GLR) * GXROW + (f % GLR) * 4; };
  const int gx_g0 = gx_g(0), gx_g1 = gx_g(1), gx_g2 = gx_g(2);   // global element offsets inside a time step
  const int gx_s0 = gx_s(0), gx_s1 = gx_s(1), gx_s2 = gx_s(2);   // float offsets inside a staging tile
  const size_t gx_ts = (size_t)B * D * 3 * H, out_ts = (size_t)B * D * H;
#define SLU_GX_REQUEST(t_, a, b, c)                                                         \
  do {                                                                                      \
    const float* gp__ = p.gx + (size_t)(t_) * gx_ts;                                        \
    a = *reinterpret_cast<const float4*>(gp__ + gx_g0);                                     \
    b = *reinterpret_cast<const float4*>(gp__ + gx_g1);                                     \
    c = *reinterpret_cast<const float4*>(gp__ + gx_g2);                                     \
  } while (0)
#define SLU_GX_STAGE(buf, a, b, c)                                                          \
  do {                                                                                      \
    float* gs__ = gxs + (buf) * (16 * GXROW);                                               \
    *reinterpret_cast<float4*>(gs__ + gx_s0) = a;                                           \
    *reinterpret_cast<float4*>(gs__ + gx_s1) = b;                                           \
    *reinterpret_cast<float4*>(gs__ + gx_s2) = c;                                           \
  } while (0)
  // output rows: fp32 (EPI 0 / 2): H/4 lanes per row, RW rows = 64 lanes = one 16-byte store per lane;
  // planes (EPI 1): H/8 lanes per row and plane, NS planes x 16 rows = NS * 2H lanes: wave-instructions c = w, w + NW, ...
  const int orow = RW * w + lane / (H / 4), oc16 = lane % (H / 4);
  const bool orow_ok = b0 + orow < B;
  const int o_goff = (b0 + orow) * D * H + dir * H + oc16 * 4;

  // compute-layout side
  int a_off[KC];                                // h fragment (B operand): row i, slot (c*4 + kg) ^ i
#pragma unroll
  for (int c = 0; c < KC; ++c) a_off[c] = i * ROWB + (((c * 4 + kg) ^ i) & (SLOTS - 1)) * 16;
  const int h_off = i * ROWB + ((((u0 >> 3) ^ i) & (SLOTS - 1)) * 16) + (u0 & 7) * 2;     // h store: 8 bytes at unit u0
  const int gs_off = i * GXROW + u0;            // gx seed read: + g * H
  const int os_off = i * OROW + u0;             // fp32 output write
  const int ps_off = i * PROWB + u0 * 2;        // plane output write (bytes)
  // dropout keep bits of (t, sequence, this lane's four channels)
  const int kw_off = seqc * ((D * H) >> 5) + ((dir * H + u0) >> 5);
  const int kw_sh = (dir * H + u0) & 31;
  const size_t kw_ts = (size_t)B * ((D * H) >> 5);
  const bool drop = EPI > 0 && p.keep != nullptr;

  for (int x = tid; x < 2 * NS * 16 * ROWB / 4; x += H * 4) reinterpret_cast<unsigned*>(hbuf)[x] = 0u;   // h0 = 0
  float hprev[4] = {0.f, 0.f, 0.f, 0.f};
  float gcur[3][4];
  unsigned kcur = 0;
  {
    const int t0 = dir ? T - 1 : 0;
    if constexpr (KI > 0) {
      __syncthreads();                 // bias_s
      uint4 xa0[KIA][NS];
      xload(t0, xa0);
      xproj(xa0, gcur);
    } else {
      float4 ga, gb, gc;
      SLU_GX_REQUEST(t0, ga, gb, gc);
      SLU_GX_STAGE(0, ga, gb, gc);
    }
    if (drop) kcur = p.keep[(size_t)t0 * kw_ts + kw_off];
  }
  float held[4] = {0.f, 0.f, 0.f, 0.f};
  int pend_t = -1;                     // frame (EPI 0) / pooled frame (EPI > 0) whose rows wait in ost[pend_buf], -1 = none
  int pend_buf = 0;
  // (Round 6, measured and not kept — profiles/r06_c_gru_deferred_epilogue.txt: running the Dropout + pooling + split epilogue
  // of step s - 1 INSIDE step s, branch-free, so that its ~60 VALU instructions could sit between the step's MFMAs.  The
  // scheduler places them in FRONT of the MFMAs whatever sched_group_barrier pattern asks for (their operands are ready, the
  // MFMAs wait for their LDS fragments), a non-emitting step then also pays the plane split, and the launch got 4 % SLOWER:
  // T = 300, 1280 sequences 636 -> 663 us.  Second attempt: the epilogue cut into 13 slices of <= 5 VALU instructions, each
  // FENCED (sched_barrier) behind a group of three MFMAs, h fragments of the next chunk fetched as soon as a plane's last
  // product has issued — the generated code does interleave then, bit-identical, and is STILL slower: 620 -> 644 us
  // (f16x2: 425 -> 457).  Instructions placed between MFMAs of one accumulator chain cost more than their issue slot on
  // gfx950 (MI355X_MICROARCH.md: "+43 cycles for the first extra state"): back-to-back MFMAs are the fast form of this loop.)
  // the previous step's output: staging tile -> global memory, whole rows
  auto flush = [&]() {
    if (pend_t < 0) return;
    if constexpr (EPI == 1) {
      constexpr int LPR = H / 8;                               // lanes per plane row
      constexpr int NCH = NS * 16 * LPR / 64;                  // wave-instructions for the whole tile
      for (int c = w; c < NCH; c += NW) {
        const int f = c * 64 + lane, pl = f / (16 * LPR), r = (f / LPR) % 16, c16 = f % LPR;
        if (b0 + r < B) {
          const uint4 v = *reinterpret_cast<const uint4*>(ost + ((pend_buf * NS + pl) * 16 + r) * PROWB + c16 * 16);
          *reinterpret_cast<uint4*>(p.planes + (size_t)pl * p.plane + (size_t)pend_t * out_ts + (b0 + r) * D * H + dir * H + c16 * 8) = v;
        }
      }
    } else {
      if (orow_ok) {
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ost) + (pend_buf * 16 + orow) * OROW + oc16 * 4);
        *reinterpret_cast<float4*>(p.out + (size_t)pend_t * out_ts + o_goff) = v;
      }
    }
  };
  __syncthreads();
  // Nothing may be pending when the loop is entered: the compiler sinks the W_hh split below the barrier above, and waits
  // it would otherwise place INSIDE the loop for the entry path would execute on every iteration.
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)

  // The loop is unrolled by two with the buffer index `cur` = s & 1 a compile-time constant of each half (round 6: staging
  // tile / h buffer addresses fold into the LDS instructions' immediate offsets, and in the forward direction the frame
  // parity of the pooling epilogue is static).
  auto step = [&](auto cur_c, const int s) {
    constexpr int cur = decltype(cur_c)::value;
    const int t = dir ? T - 1 - s : s;
    const int tn = (s + 1 < T) ? (dir ? t - 1 : t + 1) : t;      // last step: re-reads its own row (unused)

    // ---- A: seed the accumulator chains with gx + b_hh (r, z) and b_hh (n) ----
    if constexpr (BIAS_LDS) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(&bias_s[g * H + u0]);
        bh[g][0] = v.x; bh[g][1] = v.y; bh[g][2] = v.z; bh[g][3] = v.w;
      }
    }
    if constexpr (KI == 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(gxs + cur * (16 * GXROW) + gs_off + g * H);
        gcur[g][0] = v.x; gcur[g][1] = v.y; gcur[g][2] = v.z; gcur[g][3] = v.w;
      }
    }
    f32x4 accs[SP::NACC][3];
#pragma unroll
    for (int a = 0; a < SP::NACC; ++a)
#pragma unroll
      for (int g = 0; g < 3; ++g) accs[a][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      accs[0][0][r] = gcur[0][r] + bh[0][r];
      accs[0][1][r] = gcur[1][r] + bh[1][r];
      accs[0][2][r] = bh[2][r];
      gn[r] = gcur[2][r];
    }
    const unsigned kbits = kcur >> kw_sh;

    // ---- B: the next step's operands; the previous step's output ----
    float gnext[3][4];
    float4 gl0 = make_float4(0.f, 0.f, 0.f, 0.f), gl1 = gl0, gl2 = gl0;   // gx rows of the next step (row-wise layout)
    uint4 xan[KIA][NS];                                          // fused input: fragments of the NEXT step's x
    if constexpr (KI > 0) xload(tn, xan);                        // in flight during this step; multiplied at its end
    else SLU_GX_REQUEST(tn, gl0, gl1, gl2);
    unsigned knext = 0;
    if (drop) knext = p.keep[(size_t)tn * kw_ts + kw_off];
    flush();

    // ---- C: W_hh h_{t-1} (h plane = PA(q), W plane = PB(q)), gates, h_t ----
    const unsigned char* hc = hbuf + cur * (NS * 16 * ROWB);
    float hn[4];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      uint4 fa[NS];
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) fa[pl] = *reinterpret_cast<const uint4*>(hc + pl * (16 * ROWB) + a_off[c]);
#pragma unroll
      for (int q = 0; q < SP::NPAIR; ++q)
#pragma unroll
        for (int g = 0; g < 3; ++g)
          accs[SP::ACC(q)][g] = mfma_split<NS>(wb[g][c][SP::PB(q)], fa[SP::PA(q)], accs[SP::ACC(q)][g]);
    }
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = split_result<NS>(accs[0][g], accs[SP::NACC - 1][g]);

#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float rr = bf_sigmoid(acc[0][r]);
      const float zz = bf_sigmoid(acc[1][r]);
      const float nn = bf_tanh(gn[r] + rr * acc[2][r]);
      hn[r] = nn + zz * (hprev[r] - nn);
    }
    // h_t -> LDS as NS planes: four consecutive units = one 8-byte store per plane
    {
      unsigned char* __restrict__ hnext = hbuf + (cur ^ 1) * (NS * 16 * ROWB);
      if constexpr (NS == 2) {
        unsigned hi01, lo01, hi23, lo23;
        split_f16x2_pair_flush(hn[0], hn[1], hi01, lo01);
        split_f16x2_pair_flush(hn[2], hn[3], hi23, lo23);
        *reinterpret_cast<uint2*>(hnext + h_off) = make_uint2(hi01, hi23);
        *reinterpret_cast<uint2*>(hnext + 16 * ROWB + h_off) = make_uint2(lo01, lo23);
      } else if constexpr (NS == 3) {
        unsigned w01[3], w23[3];        // (round 6: the packed pair split, 11 VALU instructions per pair: 636 -> 620 us at T = 300)
        split_bf16x3_pair(hn[0], hn[1], w01);
        split_bf16x3_pair(hn[2], hn[3], w23);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          *reinterpret_cast<uint2*>(hnext + pl * (16 * ROWB) + h_off) = make_uint2(w01[pl], w23[pl]);
      } else {
        unsigned short sp[4][NS];
#pragma unroll
        for (int r = 0; r < 4; ++r) split_terms<NS>(hn[r], sp[r]);
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
          *reinterpret_cast<uint2*>(hnext + pl * (16 * ROWB) + h_off) =
              make_uint2(sp[0][pl] | ((unsigned)sp[1][pl] << 16), sp[2][pl] | ((unsigned)sp[3][pl] << 16));
      }
    }
    // this step's output -> its staging tile (compute layout); stored row-wise at the top of the next step
    if constexpr (EPI == 0) {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(ost) + cur * (16 * OROW) + os_off) = make_float4(hn[0], hn[1], hn[2], hn[3]);
      pend_t = t; pend_buf = cur;
    } else {
      // Dropout (keep bit ? h * scale : h * 0) and average pooling over frames (2 to, 2 to + 1), in the operation order of
      // dropout_pool_fwd4_kernel: acc = 0 + v(2 to); acc += v(2 to + 1); acc / n  (n = 1 for the partial last window)
      float m[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        m[r] = drop ? mul_rnd(hn[r], ((kbits >> r) & 1u) ? p.keep_scale : 0.0f) : hn[r];
      const bool even = dir ? (t & 1) == 0 : cur == 0;     // forward: t = s, its parity is the half's
      const bool single = even && t == T - 1;
      const bool emit = dir ? even : (!even || single);
      pend_t = -1;
      if (!emit) {
#pragma unroll
        for (int r = 0; r < 4; ++r) held[r] = dir ? m[r] : add_rnd(0.0f, m[r]);
      } else {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float first = dir ? add_rnd(0.0f, m[r]) : held[r];      // 0 + v(2 to)
          const float second = dir ? held[r] : m[r];                       // v(2 to + 1)
          v[r] = mul_rnd(add_rnd(first, second), 0.5f);
        }
        if (single) {                     // the partial last window (T odd): one step of the T — a branch, not four selects
          asm volatile("" ::: "memory");
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = add_rnd(0.0f, m[r]);
        }
        pend_t = t >> 1; pend_buf = cur;
        if constexpr (EPI == 2) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(ost) + cur * (16 * OROW) + os_off) = make_float4(v[0], v[1], v[2], v[3]);
        } else if constexpr (NS == 2) {
          // "round to fp16, flush denormals" (split_f16x2_flush of slu_bf16.h, the rule dropout_pool_fwd4_kernel<2> uses):
          // in this kernel's flush mode that is two packed conversions per pair
          unsigned hi01, lo01, hi23, lo23;
          split_f16x2_pair_flush(v[0], v[1], hi01, lo01);
          split_f16x2_pair_flush(v[2], v[3], hi23, lo23);
          *reinterpret_cast<uint2*>(ost + ((cur * NS + 0) * 16) * PROWB + ps_off) = make_uint2(hi01, hi23);
          *reinterpret_cast<uint2*>(ost + ((cur * NS + 1) * 16) * PROWB + ps_off) = make_uint2(lo01, lo23);
        } else if constexpr (NS == 3) {
          unsigned w01[3], w23[3];
          split_bf16x3_pair(v[0], v[1], w01);
          split_bf16x3_pair(v[2], v[3], w23);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<uint2*>(ost + ((cur * NS + pl) * 16) * PROWB + ps_off) = make_uint2(w01[pl], w23[pl]);
        } else {
          unsigned short sp[4][NS];
#pragma unroll
          for (int r = 0; r < 4; ++r) split_terms<NS>(v[r], sp[r]);
#pragma unroll
          for (int pl = 0; pl < NS; ++pl)
            *reinterpret_cast<uint2*>(ost + ((cur * NS + pl) * 16) * PROWB + ps_off) =
                make_uint2(sp[0][pl] | ((unsigned)sp[1][pl] << 16), sp[2][pl] | ((unsigned)sp[3][pl] << 16));
        }
      }
    }
    // the next step's x W_ih^T + b_ih: independent of h (round 3 measured this placement against two interleavings)
    if constexpr (KI > 0) {
      xproj(xan, gnext);
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) gcur[g][r] = gnext[g][r];
    } else {
      // ---- D: the gx rows requested in B (a step ago in wall time: the wait is free) -> the other staging tile ----
      SLU_GX_STAGE(cur ^ 1, gl0, gl1, gl2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) hprev[r] = hn[r];
    kcur = knext;
    lds_barrier();
  };
  {
    int s = 0;
    for (; s + 1 < T; s += 2) {
      step(std::integral_constant<int, 0>{}, s);
      step(std::integral_constant<int, 1>{}, s + 1);
    }
    if (s < T) step(std::integral_constant<int, 0>{}, s);
  }
  flush();
#undef SLU_GX_REQUEST
#undef SLU_GX_STAGE
}

template <int H, int NS, int KI, int EPI>
__global__ void __launch_bounds__(H * 4)
gru_bf_fwd_kernel(const GruBfParams p) {
  if (blockIdx.y) gru_bf_fwd_body<H, NS, KI, EPI, true>(p);
  else gru_bf_fwd_body<H, NS, KI, EPI, false>(p);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: TWO sequence tiles per workgroup, half a step apart (gru_bf2_fwd_kernel; opt-in: seq_tiles = 2).
//
// What bounds gru_bf_fwd_kernel is not a roofline but the issue of its own dependent chain (profiles/r05_z_pmc_gru_bf.txt):
// per wave and step 1152 cycles of MFMA issue (72 x 16) and ~1080 cycles of gate / split VALU work that need each other's
// results, and the SIMD's two waves sit in the same phase at the same time (one barrier per step): 2 x 1152 + 2 x 1080 of the
// 4750 cycles of a step.  Here a workgroup owns 32 sequences = two tiles A, B that share the waves' resident W_hh slices, and
// time is cut into PHASES of half a step:
//     phase 2s    :  M(A, s)  ||  G(B, s-1)        M(X, s) = the 72 MFMAs  W_hh h_{s-1}  of tile X (needs all of h_{s-1}^X)
//     phase 2s + 1:  M(B, s)  ||  G(A, s)          G(X, s) = gates, blend, split h_s -> LDS, Dropout + pooling, output tile
// with one LDS barrier per phase.  M and G of a phase belong to DIFFERENT tiles, i.e. they are independent: they form ONE
// scheduling region and the compiler interleaves the gate arithmetic of one tile with the MFMAs of the other.  Same products
// in the same order, same gate formulas, same epilogue as gru_bf_fwd_kernel: the results are bit-identical
// (tests/test_hip_bf16.py::test_gru_two_tiles_per_workgroup_equals_one_tile).
//
// MEASURED (profiles/r06_a_gru_two_tiles.txt, MI355X, T = 300, 2560 sequences on 160 CUs, bf16x3, pooled planes out): one tile
// per workgroup 4.10 us per pair of tile-steps, this kernel 3.78 (- 8 %); f16x2: 2.80 -> 3.05 (slower).  The premise "one wave
// on the matrix pipe while its SIMD neighbour does gate math doubles the throughput" does NOT hold on gfx950
// (tools/probes/mfma_valu_overlap_probe2.cpp, profiles/r06_a_mfma_valu_coissue.txt): beside a wave that issues
// v_mfma_f32_16x16x32_bf16 back to back, a second wave's v_fma / v_mul / v_sub stream hides 58 % of its time, v_and / v_mov /
// v_perm 74 %, v_cvt_pk_bf16_f32 66 %, v_exp / v_rcp 50 %, and the packed fp32 operations (v_pk_fma / v_pk_add / v_pk_mul_f32)
// NOTHING (0.07): an MFMA holds the SIMD's VALU issue port for about half of its 16 cycles.  The explicit form of the idea —
// the first wave of every SIMD (HW_ID.SIMD_ID + an LDS ticket) runs M then G, the second G then M — was built and is SLOWER
// than both waves in the same order (4.29 against 4.14 us; 3.94 against 3.53 with fp32 output): the wave that feeds the
// matrix pipe starves its neighbour's VALU stream and the phase ends when the slower one does.  Letting the compiler
// interleave inside one wave gives the 8 %.  Hence opt-in (SLU_GRU_TILES=2); the default stays one tile per workgroup.
//
// Memory: gx of chunk c (= tile c & 1, step c >> 1) is needed in phase c.  It is fetched by LDS-DMA (global_load_lds_dwordx4:
// no staging registers, no ds_write) into a ring of THREE chunk buffers, two phases ahead; the keep bits of the chunk ride
// along (one global_load_lds_dword).  The DMA is issued from inline assembly and awaited with an explicit
// s_waitcnt vmcnt(4) at the end of each phase (a wave issues exactly four per phase, AFTER the phase's stores; memory
// operations retire in order): the compiler would order EVERY later LDS read behind a pending LDS-DMA (vmcnt(0)), which
// would put the memory latency on the chain.  There is no other vector-memory LOAD in the loop, so the compiler inserts no
// waits of its own (a register spill would: the fp32 h_{s-1} and the pooling window's first frame therefore live in
// private LDS slots, not in registers).  The LDS image of a chunk is dense (16 rows x 3H floats) with the 16-byte pieces of
// row r XOR-swizzled by r — applied to the SOURCE address of the DMA — so that the compute layout's ds_read_b128
// (lane = (sequence, unit quad)) is conflict-free.  Loads of rows past B are clamped to sequence B - 1 (they recompute that
// row's values), stores of such rows are predicated off.
template <int H, int NS, int EPI>
struct Gru2Lds {
  static constexpr int ROWB = H * 2;                        // bytes per h row (one sequence, one plane)
  static constexpr int OROW = H + 4;                        // floats per staged fp32 output row
  static constexpr int PROWB = H * 2 + 16;                  // bytes per staged plane row
  static constexpr int HB = NS * 16 * ROWB;                 // h planes of ONE tile (single buffer: written in G, read in M)
  static constexpr int GXB = 16 * 3 * H * 4;                // one gx chunk
  static constexpr int KBB = 16 * (H / 32) * 4;             // keep words of one chunk (this direction's H / 32 words per sequence)
  static constexpr int OSB = EPI == 1 ? NS * 16 * PROWB : 16 * OROW * 4;   // output tile of ONE sequence tile
  static constexpr int HBUF = 0;
  static constexpr int GXR = HBUF + 2 * HB;
  static constexpr int KBR = GXR + 3 * GXB;
  static constexpr int BIAS = KBR + 3 * KBB;
  static constexpr int OST = BIAS + 3 * H * 4;
  static constexpr int HP = OST + 2 * OSB;                   // fp32 h_{s-1} of the lane's four units (private slots)
  static constexpr int HELD = HP + 2 * 16 * H * 4;          // EPI > 0: the masked h of a pooling window's first frame
  static constexpr int BYTES = HELD + (EPI > 0 ? 2 * 16 * H * 4 : 0);
};

// LDS-DMA from inline assembly (see above): 64 lanes x 16 (4) bytes from base + voff (bytes) to LDS bytes [m0, m0 + 1024 (256))
__device__ __forceinline__ void dma_b128(const void* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma_b32(const void* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}

template <int H, int NS, int EPI>
__global__ void __launch_bounds__(H * 4)
gru_bf2_fwd_kernel(const GruBfParams p) {
  static_assert(H == 128, "two-tile recurrence: H = 128 (8 waves = two per SIMD, one keep-word DMA per chunk)");
  constexpr int NW = H / 16, KC = H / 32, SLOTS = H / 8;
  typedef Split<NS> SP;
  typedef Gru2Lds<H, NS, EPI> LD;
  constexpr int ROWB = LD::ROWB, OROW = LD::OROW, PROWB = LD::PROWB;
  constexpr int PCS = 3 * H / 4;                // 16-byte pieces per gx row (96)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  float* const bias_s = reinterpret_cast<float*>(smem + LD::BIAS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kg = lane >> 4;      // compute layout: i = sequence of the tile (MFMA column), kg = unit quad
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 32;
  const int u0 = w * 16 + kg * 4;
  const int T = p.T, B = p.B, D = p.D;

  if constexpr (NS == 2) f16_denorm_flush();

  // resident W_hh fragments, as in gru_bf_fwd_kernel
  uint4 wb[3][KC][NS];
  {
    const float* __restrict__ W = p.w_hh[dir];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float* src = W + (size_t)(g * H + w * 16 + i) * H + c * 32 + kg * 8;
        const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        unsigned short s[8][NS];
#pragma unroll
        for (int e = 0; e < 8; ++e) split_terms<NS>(v[e], s[e]);
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) {
          uint4 o;
          o.x = s[0][pl] | ((unsigned)s[1][pl] << 16); o.y = s[2][pl] | ((unsigned)s[3][pl] << 16);
          o.z = s[4][pl] | ((unsigned)s[5][pl] << 16); o.w = s[6][pl] | ((unsigned)s[7][pl] << 16);
          wb[g][c][pl] = o;
        }
      }
  }
  for (int x = tid; x < 3 * H; x += H * 4) bias_s[x] = p.b_hh[dir][x];
  for (int x = tid; x < 2 * LD::HB / 4; x += H * 4) reinterpret_cast<unsigned*>(smem + LD::HBUF)[x] = 0u;   // h0 = 0
  __syncthreads();

  // ---- DMA side: this wave moves 1 KiB pieces 3w .. 3w + 2 of a chunk (gx) and the chunk's 256 bytes of keep words ----
  const bool drop = EPI > 0 && p.keep != nullptr;
  unsigned gv[2][3];                            // byte offsets inside a time step, per tile
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int P = (3 * w + j) * 64 + lane, row = P / PCS, q = P % PCS;
      gv[x][j] = (unsigned)((min(b0 + 16 * x + row, B - 1) * D * 3 * H + dir * 3 * H) * 4 + (q ^ (row & 15)) * 16);
    }
  unsigned kv[2];
  const int KW = (D * H) >> 5;                  // keep words per (t, sequence)
#pragma unroll
  for (int x = 0; x < 2; ++x)
    kv[x] = drop ? (unsigned)((min(b0 + 16 * x + (lane >> 2), B - 1) * KW + dir * (H >> 5) + (lane & 3)) * 4) : (unsigned)(lane * 4);
  const void* const kbase = drop ? (const void*)p.keep : (const void*)p.gx;      // no dropout: a harmless read (uniform counts)
  const size_t gx_ts = (size_t)B * D * 3 * H * 4, kw_ts = drop ? (size_t)B * KW * 4 : 0, out_ts = (size_t)B * D * H;
  const unsigned gx_m0 = lds0 + LD::GXR + 3 * w * 1024, kb_m0 = lds0 + LD::KBR;
  // chunk (tile x, time t) -> ring slot
  auto dma_chunk = [&](int x, int t_, int slot) {
    const char* gb = reinterpret_cast<const char*>(p.gx) + (size_t)t_ * gx_ts;
    const unsigned m = gx_m0 + slot * LD::GXB;
    const unsigned v0 = x ? gv[1][0] : gv[0][0], v1 = x ? gv[1][1] : gv[0][1], v2 = x ? gv[1][2] : gv[0][2];
    dma_b128(gb, v0, m);
    dma_b128(gb, v1, m + 1024);
    dma_b128(gb, v2, m + 2048);
    dma_b32(reinterpret_cast<const char*>(kbase) + (size_t)t_ * kw_ts, x ? kv[1] : kv[0], kb_m0 + slot * LD::KBB);
  };

  // ---- compute-layout side ----
  // h fragment (B operand) of chunk c: row i, slot (c*4 + kg) ^ i = ((kg ^ i) ^ 4c) — one register, c enters as an XOR of
  // byte-offset bits 6-7 (register budget: 144 resident W_hh registers + two tiles' accumulators)
  const int a_off0 = i * ROWB + ((kg ^ i) & (SLOTS - 1)) * 16;
  const int h_off = i * ROWB + ((((u0 >> 3) ^ i) & (SLOTS - 1)) * 16) + (u0 & 7) * 2;
  // gx seed read (bytes inside a chunk buffer): row i, piece (g*H/4 + u0/4) ^ i = ((u0/4) ^ i) + g*H/4 — gate g is an
  // immediate offset of g * 4H bytes
  const int gs_off0 = (i * PCS + ((u0 >> 2) ^ i)) * 16;
  const int hp_off = (i * H + u0) * 4;          // private fp32 slot of (sequence i, units u0 .. u0 + 3)
  const int kb_off = (i * (H >> 5) + (u0 >> 5)) * 4;
  const int kb_sh = u0 & 31;
  const int os_off = i * OROW + u0;             // fp32 output write (floats)
  const int ps_off = i * PROWB + u0 * 2;        // plane output write (bytes)
  // row-wise side of the output tiles (flush), rows clamped to B - 1
  const int orow = (16 / NW) * w + lane / (H / 4), oc16 = lane % (H / 4);

  f32x4 accs[2][SP::NACC][3];
  float gn[2][4];
  unsigned kb[2] = {0u, 0u};
  int pend[2] = {-1, -1};
#pragma unroll
  for (int x = 0; x < 2; ++x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) gn[x][r] = 0.f;
    *reinterpret_cast<float4*>(smem + LD::HP + x * (16 * H * 4) + hp_off) = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EPI > 0) *reinterpret_cast<float4*>(smem + LD::HELD + x * (16 * H * 4) + hp_off) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int a = 0; a < SP::NACC; ++a)
#pragma unroll
      for (int g = 0; g < 3; ++g) accs[x][a][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  // M(X): seed the chains of tile X with gx + b_hh from ring slot `slot`, then W_hh h_{s-1}
#define SLU_G2_M(X, slot)                                                                                              \
  do {                                                                                                                 \
    const unsigned char* gxc__ = smem + LD::GXR + (slot) * LD::GXB;                                                    \
    _Pragma("unroll") for (int g = 0; g < 3; ++g) {                                                                    \
      const float4 v = *reinterpret_cast<const float4*>(gxc__ + gs_off0 + g * (4 * H));                                \
      const float4 bb = *reinterpret_cast<const float4*>(&bias_s[g * H + u0]);                                         \
      _Pragma("unroll") for (int a = 0; a < SP::NACC; ++a) accs[X][a][g] = f32x4{0.f, 0.f, 0.f, 0.f};                  \
      if (g < 2) accs[X][0][g] = f32x4{v.x + bb.x, v.y + bb.y, v.z + bb.z, v.w + bb.w};                                \
      else { accs[X][0][g] = f32x4{bb.x, bb.y, bb.z, bb.w}; gn[X][0] = v.x; gn[X][1] = v.y; gn[X][2] = v.z; gn[X][3] = v.w; } \
    }                                                                                                                  \
    kb[X] = *reinterpret_cast<const unsigned*>(smem + LD::KBR + (slot) * LD::KBB + kb_off) >> kb_sh;                   \
    const unsigned char* hc__ = smem + LD::HBUF + (X) * LD::HB;                                                        \
    _Pragma("unroll") for (int c = 0; c < KC; ++c) {                                                                   \
      uint4 fa[NS];                                                                                                    \
      _Pragma("unroll") for (int pl = 0; pl < NS; ++pl) fa[pl] = *reinterpret_cast<const uint4*>(hc__ + pl * (16 * ROWB) + (a_off0 ^ (c * 64))); \
      _Pragma("unroll") for (int q = 0; q < SP::NPAIR; ++q)                                                            \
        _Pragma("unroll") for (int g = 0; g < 3; ++g)                                                                  \
          accs[X][SP::ACC(q)][g] = mfma_split<NS>(wb[g][c][SP::PB(q)], fa[SP::PA(q)], accs[X][SP::ACC(q)][g]);         \
    }                                                                                                                  \
  } while (0)

  // G(X) at time t_: gates, blend, split h -> LDS, Dropout + pooling, output tile (gru_bf_fwd_kernel's arithmetic)
#define SLU_G2_G(X, t_)                                                                                                \
  do {                                                                                                                 \
    f32x4 acc[3];                                                                                                      \
    _Pragma("unroll") for (int g = 0; g < 3; ++g) acc[g] = split_result<NS>(accs[X][0][g], accs[X][SP::NACC - 1][g]);  \
    float hn[4];                                                                                                       \
    float4* const hp__ = reinterpret_cast<float4*>(smem + LD::HP + (X) * (16 * H * 4) + hp_off);                       \
    const float4 hpv = *hp__;                                                                                          \
    const float hprev[4] = {hpv.x, hpv.y, hpv.z, hpv.w};                                                               \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                    \
      const float rr = bf_sigmoid(acc[0][r]);                                                                          \
      const float zz = bf_sigmoid(acc[1][r]);                                                                          \
      const float nn = bf_tanh(gn[X][r] + rr * acc[2][r]);                                                             \
      hn[r] = nn + zz * (hprev[r] - nn);                                                                               \
    }                                                                                                                  \
    *hp__ = make_float4(hn[0], hn[1], hn[2], hn[3]);                                                                   \
    {                                                                                                                  \
      unsigned char* __restrict__ hx = smem + LD::HBUF + (X) * LD::HB;                                                 \
      if constexpr (NS == 2) {                                                                                         \
        unsigned hi01, lo01, hi23, lo23;                                                                               \
        split_f16x2_pair_flush(hn[0], hn[1], hi01, lo01);                                                              \
        split_f16x2_pair_flush(hn[2], hn[3], hi23, lo23);                                                              \
        *reinterpret_cast<uint2*>(hx + h_off) = make_uint2(hi01, hi23);                                                \
        *reinterpret_cast<uint2*>(hx + 16 * ROWB + h_off) = make_uint2(lo01, lo23);                                    \
      } else if constexpr (NS == 3) {                                                                                  \
        unsigned w01[3], w23[3];                                                                                       \
        split_bf16x3_pair(hn[0], hn[1], w01);                                                                          \
        split_bf16x3_pair(hn[2], hn[3], w23);                                                                          \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                               \
          *reinterpret_cast<uint2*>(hx + pl * (16 * ROWB) + h_off) = make_uint2(w01[pl], w23[pl]);                     \
      } else {                                                                                                         \
        unsigned short sp[4][NS];                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) split_terms<NS>(hn[r], sp[r]);                                   \
        _Pragma("unroll") for (int pl = 0; pl < NS; ++pl)                                                              \
          *reinterpret_cast<uint2*>(hx + pl * (16 * ROWB) + h_off) =                                                   \
              make_uint2(sp[0][pl] | ((unsigned)sp[1][pl] << 16), sp[2][pl] | ((unsigned)sp[3][pl] << 16));            \
      }                                                                                                                \
    }                                                                                                                  \
    unsigned char* const ox__ = smem + LD::OST + (X) * LD::OSB;                                                        \
    if constexpr (EPI == 0) {                                                                                          \
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(ox__) + os_off) = make_float4(hn[0], hn[1], hn[2], hn[3]);   \
      pend[X] = (t_);                                                                                                  \
    } else {                                                                                                           \
      float m[4];                                                                                                      \
      _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                                    \
        m[r] = drop ? __fmul_rn(hn[r], ((kb[X] >> r) & 1u) ? p.keep_scale : 0.0f) : hn[r];                             \
      const bool even = ((t_) & 1) == 0;                                                                               \
      const bool single = even && (t_) == T - 1;                                                                       \
      const bool emit = dir ? even : (!even || single);                                                                \
      pend[X] = -1;                                                                                                    \
      float4* const hl__ = reinterpret_cast<float4*>(smem + LD::HELD + (X) * (16 * H * 4) + hp_off);                   \
      if (!emit) {                                                                                                     \
        float hd[4];                                                                                                   \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) hd[r] = dir ? m[r] : __fadd_rn(0.0f, m[r]);                      \
        *hl__ = make_float4(hd[0], hd[1], hd[2], hd[3]);                                                               \
      } else {                                                                                                         \
        const float4 hlv = *hl__;                                                                                      \
        const float held[4] = {hlv.x, hlv.y, hlv.z, hlv.w};                                                            \
        float v[4];                                                                                                    \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                \
          const float first = dir ? __fadd_rn(0.0f, m[r]) : held[r];                                                   \
          const float second = dir ? held[r] : m[r];                                                                   \
          v[r] = single ? __fadd_rn(0.0f, m[r]) : __fmul_rn(__fadd_rn(first, second), 0.5f);                           \
        }                                                                                                              \
        pend[X] = (t_) >> 1;                                                                                           \
        if constexpr (EPI == 2) {                                                                                      \
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(ox__) + os_off) = make_float4(v[0], v[1], v[2], v[3]);   \
        } else if constexpr (NS == 2) {                                                                                \
          unsigned hi01, lo01, hi23, lo23;                                                                             \
          split_f16x2_pair_flush(v[0], v[1], hi01, lo01);                                                              \
          split_f16x2_pair_flush(v[2], v[3], hi23, lo23);                                                              \
          *reinterpret_cast<uint2*>(ox__ + ps_off) = make_uint2(hi01, hi23);                                           \
          *reinterpret_cast<uint2*>(ox__ + 16 * PROWB + ps_off) = make_uint2(lo01, lo23);                              \
        } else if constexpr (NS == 3) {                                                                                \
          unsigned w01[3], w23[3];                                                                                     \
          split_bf16x3_pair(v[0], v[1], w01);                                                                          \
          split_bf16x3_pair(v[2], v[3], w23);                                                                          \
          _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                             \
            *reinterpret_cast<uint2*>(ox__ + (pl * 16) * PROWB + ps_off) = make_uint2(w01[pl], w23[pl]);               \
        } else {                                                                                                       \
          unsigned short sp[4][NS];                                                                                    \
          _Pragma("unroll") for (int r = 0; r < 4; ++r) split_terms<NS>(v[r], sp[r]);                                  \
          _Pragma("unroll") for (int pl = 0; pl < NS; ++pl)                                                            \
            *reinterpret_cast<uint2*>(ox__ + (pl * 16) * PROWB + ps_off) =                                             \
                make_uint2(sp[0][pl] | ((unsigned)sp[1][pl] << 16), sp[2][pl] | ((unsigned)sp[3][pl] << 16));          \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)

  // the output tile of tile X (assembled by G in the previous phase) -> global memory, whole rows.  EPI 1: NS planes x 16 rows
  // x 2H bytes = NS * 4 wave-instructions of 1 KiB: wave w < NS * 4 moves piece w (plane w / 4, rows 4 (w % 4) ..) and, for w < NS * 4 - 8,
  // piece w + 8.  Stores are issued BEFORE the phase's DMA, so their number does not enter the vmcnt(4) of the phase's end.
#define SLU_G2_FLUSH(X)                                                                                                \
  do {                                                                                                                 \
    if (pend[X] >= 0) {                                                                                                \
      const unsigned char* ox__ = smem + LD::OST + (X) * LD::OSB;                                                      \
      if constexpr (EPI == 1) {                                                                                        \
        const int r = (w & 3) * 4 + (lane >> 4), c16 = lane & 15;                                                      \
        unsigned short* const gp__ = p.planes + (size_t)pend[X] * out_ts + (size_t)((b0 + 16 * (X) + r) * D * H + dir * H + c16 * 8); \
        const bool ok__ = b0 + 16 * (X) + r < B;                                                                       \
        if (w < NS * 4) {                                                                                              \
          const int pl = w >> 2;                                                                                       \
          const uint4 v = *reinterpret_cast<const uint4*>(ox__ + (pl * 16 + r) * PROWB + c16 * 16);                    \
          if (ok__) *reinterpret_cast<uint4*>(gp__ + (size_t)pl * p.plane) = v;                                        \
        }                                                                                                              \
        if (w + 8 < NS * 4) {                                                                                          \
          const int pl = (w >> 2) + 2;                                                                                 \
          const uint4 v = *reinterpret_cast<const uint4*>(ox__ + (pl * 16 + r) * PROWB + c16 * 16);                    \
          if (ok__) *reinterpret_cast<uint4*>(gp__ + (size_t)pl * p.plane) = v;                                        \
        }                                                                                                              \
      } else {                                                                                                         \
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ox__) + orow * OROW + oc16 * 4); \
        if (b0 + 16 * (X) + orow < B)                                                                                  \
          *reinterpret_cast<float4*>(p.out + (size_t)pend[X] * out_ts + (b0 + 16 * (X) + orow) * D * H + dir * H + oc16 * 4) = v; \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)

  // one phase: X = the M tile (chunk in ring slot sl), 1 - X = the G tile (time tg); chunk (X, time tn) is requested into the
  // slot two ahead.  doM / doG are literals (the first and the last phases of a launch lack one of the two).
#define SLU_G2_PHASE(X, doM, doG, tg, tn)                                                                              \
  do {                                                                                                                 \
    SLU_G2_FLUSH(X);                                                                                                   \
    dma_chunk(X, tn, sl == 0 ? 2 : sl - 1);                                                                            \
    if constexpr (doM) SLU_G2_M(X, sl);          /* one scheduling region: the compiler interleaves M and G */         \
    if constexpr (doG) SLU_G2_G(1 - (X), tg);                                                                          \
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                                   \
    lds_barrier();                                                                                                     \
    sl = sl == 2 ? 0 : sl + 1;                                                                                         \
  } while (0)

  auto tt = [&](int s_) { const int c = s_ < T - 1 ? s_ : T - 1; return dir ? T - 1 - c : c; };   // time of step s_ (clamped)
  dma_chunk(0, tt(0), 0);
  dma_chunk(1, tt(0), 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int sl = 0;                                   // ring slot of the chunk the current phase reads: chunk c lives in slot c % 3
  SLU_G2_PHASE(0, true, false, 0, tt(1));                       // phase 0: M(A, 0); requests chunk 2 = (A, 1)
  for (int s = 0; s + 1 < T; ++s) {
    SLU_G2_PHASE(1, true, true, tt(s), tt(s + 1));              // phase 2s + 1: M(B, s) || G(A, s); requests (B, s + 1)
    SLU_G2_PHASE(0, true, true, tt(s), tt(s + 2));              // phase 2s + 2: M(A, s + 1) || G(B, s); requests (A, s + 2)
  }
  SLU_G2_PHASE(1, true, true, tt(T - 1), tt(T - 1));            // phase 2T - 1: M(B, T - 1) || G(A, T - 1)
  SLU_G2_PHASE(0, false, true, tt(T - 1), tt(T - 1));           // phase 2T: G(B, T - 1); the last output of tile A leaves
  SLU_G2_FLUSH(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may land after the workgroup has released its LDS
#undef SLU_G2_PHASE
#undef SLU_G2_FLUSH
#undef SLU_G2_G
#undef SLU_G2_M
}

// ---------------------------------------------------------------------------------------------------------------------
// Round-3 geometry (A = h: a lane owns one hidden unit of four sequences), kept for the one caller that needs the saved
// gates in the exact BPTT kernels' lane order: the bf16 forward of TRAINABLE layers (SLU_DTYPE=bf16; reserve != null).
template <int H, int NS>
__global__ void __launch_bounds__(H * 4)
gru_bf_fwd_rs_kernel(const GruBfParams p) {
  constexpr int NW = H / 16, KC = H / 32, ROWB = H * 2, SLOTS = H / 8;
  typedef Split<NS> SP;
  __shared__ __attribute__((aligned(16))) unsigned char hbuf[2][NS][16 * ROWB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kg = lane >> 4;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const int j = w * 16 + i;           // hidden unit of this lane's outputs
  const int T = p.T, B = p.B, D = p.D;

  uint4 wb[3][KC][NS];
  {
    const float* __restrict__ W = p.w_hh[dir];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float* src = W + (size_t)(g * H + j) * H + c * 32 + kg * 8;
        const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        unsigned short s[8][NS];
#pragma unroll
        for (int e = 0; e < 8; ++e) split_terms<NS>(v[e], s[e]);
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) {
          uint4 o;
          o.x = s[0][pl] | ((unsigned)s[1][pl] << 16); o.y = s[2][pl] | ((unsigned)s[3][pl] << 16);
          o.z = s[4][pl] | ((unsigned)s[5][pl] << 16); o.w = s[6][pl] | ((unsigned)s[7][pl] << 16);
          wb[g][c][pl] = o;
        }
      }
  }
  const float bhr = p.b_hh[dir][j], bhz = p.b_hh[dir][H + j], bhn = p.b_hh[dir][2 * H + j];
  for (int x = tid; x < 2 * NS * 16 * ROWB / 4; x += H * 4) reinterpret_cast<unsigned*>(&hbuf[0][0][0])[x] = 0u;   // h0 = 0
  float hprev[4] = {0.f, 0.f, 0.f, 0.f};
  int g_off[4], o_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * kg + r;
    g_off[r] = (b < B ? b : 0) * D * 3 * H;
    o_off[r] = b < B ? b * D * H : -1;
  }
  const size_t gx_ts = (size_t)B * D * 3 * H, out_ts = (size_t)B * D * H;
  const float* __restrict__ gxd = p.gx + (size_t)dir * 3 * H + j;
  float* __restrict__ outd = p.out + (size_t)dir * H + j;
  int a_off[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) a_off[c] = i * ROWB + (((c * 4 + kg) ^ i) & (SLOTS - 1)) * 16;
  int h_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * kg + r;
    h_off[r] = row * ROWB + ((((j >> 3) ^ row) & (SLOTS - 1)) * 16) + (j & 7) * 2;
  }
  float gr[4], gz[4], gn[4];
  {
    const int t0 = dir ? T - 1 : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* g = gxd + (size_t)t0 * gx_ts + g_off[r];
      gr[r] = g[0]; gz[r] = g[H]; gn[r] = g[2 * H];
    }
  }
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const int cur = s & 1;
    float ngr[4], ngz[4], ngn[4];
    const int tn = (s + 1 < T) ? (dir ? t - 1 : t + 1) : t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* g = gxd + (size_t)tn * gx_ts + g_off[r];
      ngr[r] = g[0]; ngz[r] = g[H]; ngn[r] = g[2 * H];
    }
    f32x4 accs[SP::NACC][3];
#pragma unroll
    for (int a = 0; a < SP::NACC; ++a)
#pragma unroll
      for (int g = 0; g < 3; ++g) accs[a][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      uint4 fa[NS];
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) fa[pl] = *reinterpret_cast<const uint4*>(&hbuf[cur][pl][a_off[c]]);
#pragma unroll
      for (int q = 0; q < SP::NPAIR; ++q)
#pragma unroll
        for (int g = 0; g < 3; ++g)
          accs[SP::ACC(q)][g] = mfma_split<NS>(fa[SP::PA(q)], wb[g][c][SP::PB(q)], accs[SP::ACC(q)][g]);
    }
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = split_result<NS>(accs[0][g], accs[SP::NACC - 1][g]);

    float hn[4], rr[4], zz[4], nn[4], qq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      rr[r] = bf_sigmoid(gr[r] + (acc[0][r] + bhr));
      zz[r] = bf_sigmoid(gz[r] + (acc[1][r] + bhz));
      qq[r] = acc[2][r] + bhn;
      nn[r] = bf_tanh(gn[r] + rr[r] * qq[r]);
      hn[r] = (1.0f - zz[r]) * nn[r] + zz[r] * hprev[r];
    }
    {     // the gates the exact BPTT kernels read
      float4* __restrict__ rs = reinterpret_cast<float4*>(
          p.reserve + ((((size_t)dir * T + t) * gridDim.x + blockIdx.x) * NW + w) * (5 * 256)) + lane;
      rs[0 * 64] = make_float4(rr[0], rr[1], rr[2], rr[3]);
      rs[1 * 64] = make_float4(zz[0], zz[1], zz[2], zz[3]);
      rs[2 * 64] = make_float4(nn[0], nn[1], nn[2], nn[3]);
      rs[3 * 64] = make_float4(qq[0], qq[1], qq[2], qq[3]);
      rs[4 * 64] = make_float4(hprev[0], hprev[1], hprev[2], hprev[3]);
    }
    unsigned char* __restrict__ hnext = &hbuf[cur ^ 1][0][0];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      unsigned short sp[NS];
      split_terms<NS>(hn[r], sp);
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) *reinterpret_cast<unsigned short*>(hnext + pl * (16 * ROWB) + h_off[r]) = sp[pl];
      if (o_off[r] >= 0) outd[(size_t)t * out_ts + o_off[r]] = hn[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { hprev[r] = hn[r]; gr[r] = ngr[r]; gz[r] = ngz[r]; gn[r] = ngn[r]; }
    __syncthreads();
  }
}

template <int H, int NS, int KI, int EPI>
static void gru_bf_launch(dim3 grid, hipStream_t st, const GruBfParams& p) {
  constexpr int lds = GruLds<H, NS, KI, EPI>::BYTES;
  static bool raised = false;      // > 64 KiB of dynamic LDS needs the function attribute (once per instantiation)
  if (lds > 64 * 1024 && !raised) {
    (void)hipFuncSetAttribute((const void*)gru_bf_fwd_kernel<H, NS, KI, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    raised = true;
  }
  hipLaunchKernelGGL((gru_bf_fwd_kernel<H, NS, KI, EPI>), grid, dim3(H * 4), lds, st, p);
}

template <int H, int NS, int EPI>
static void gru_bf2_launch(hipStream_t st, const GruBfParams& p) {
  constexpr int lds = Gru2Lds<H, NS, EPI>::BYTES;
  static bool raised = false;
  if (lds > 64 * 1024 && !raised) {
    (void)hipFuncSetAttribute((const void*)gru_bf2_fwd_kernel<H, NS, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    raised = true;
  }
  hipLaunchKernelGGL((gru_bf2_fwd_kernel<H, NS, EPI>), dim3((unsigned)cdiv(p.B, 32), (unsigned)p.D), dim3(H * 4), lds, st, p);
}

template <int H, int EPI>
static int gru_bf_dispatch(int nsplit, int ki, int seq_tiles, dim3 grid, hipStream_t st, const GruBfParams& p) {
  if (seq_tiles == 2) {        // two sequence tiles per workgroup (checked by the caller: H = 128, no fused input)
    if constexpr (H == 128) {
      if (nsplit == 3) gru_bf2_launch<128, 3, EPI>(st, p);
      else if (nsplit == 2) gru_bf2_launch<128, 2, EPI>(st, p);
      else gru_bf2_launch<128, 1, EPI>(st, p);
    }
    SLU_CHECK_LAUNCH("gru_bf2_fwd_kernel");
    return SLU_OK;
  }
  if (ki > 0) {        // fused input projection: f16x2 or bf16x3, H = 128 (checked by the caller)
    if constexpr (H == 128) {
      if (nsplit == 3) {
        if (ki == 1) gru_bf_launch<128, 3, 1, EPI>(grid, st, p); else gru_bf_launch<128, 3, 2, EPI>(grid, st, p);
      } else {
        if (ki == 1) gru_bf_launch<128, 2, 1, EPI>(grid, st, p); else gru_bf_launch<128, 2, 2, EPI>(grid, st, p);
      }
    }
  } else if (nsplit == 3) {
    gru_bf_launch<H, 3, 0, EPI>(grid, st, p);
  } else if (nsplit == 2) {
    gru_bf_launch<H, 2, 0, EPI>(grid, st, p);
  } else {
    gru_bf_launch<H, 1, 0, EPI>(grid, st, p);
  }
  SLU_CHECK_LAUNCH("gru_bf_fwd_kernel");
  return SLU_OK;
}

}  // namespace slu

using namespace slu;

static int gru_bf_common(const char* who, GruBfParams& p, const float* gx, const float* w_hh_fwd, const float* w_hh_rev,
                         const float* b_hh_fwd, const float* b_hh_rev, const void* x_planes, int64_t x_plane_stride,
                         int64_t K, const void* w_ih_packed, const float* b_ih, int64_t T, int64_t B, int64_t H, int64_t D,
                         int nsplit, bool has_reserve, int seq_tiles) {
  SLU_REQUIRE((gx || x_planes) && w_hh_fwd && b_hh_fwd, "%s: null pointer", who);
  SLU_REQUIRE(seq_tiles >= 0 && seq_tiles <= 2, "%s: seq_tiles must be 0 / 1 (one 16-sequence tile per workgroup) or 2", who);
  if (seq_tiles == 2 && !(H == 128 && gx && !x_planes && !has_reserve && B * D * 3 * H < (1LL << 30)))
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "%s: two sequence tiles per workgroup are instantiated for H = 128, gx input (no fused "
             "projection), no reserve, B * D * 3H < 2^30 (got H %lld, B %lld)", who, (long long)H, (long long)B);
  const bool fused = x_planes != nullptr;
  if (fused) {
    SLU_REQUIRE(!gx && w_ih_packed && b_ih && !has_reserve, "%s: the fused input projection takes x_planes, "
                "w_ih_packed and b_ih instead of gx, and no reserve (frozen layers only)", who);
    if (!((nsplit == 2 || nsplit == 3) && H == 128 && K >= 1 && K <= 64))
      SLU_FAIL(SLU_ERR_UNSUPPORTED, "%s: the fused input projection is instantiated for f16x2 / bf16x3 (nsplit 2 / 3), "
               "H = 128 and at most 64 input channels (got nsplit %d, H %lld, K %lld)", who, nsplit, (long long)H, (long long)K);
    SLU_REQUIRE(x_plane_stride >= T * B * (cdiv(K, 32) * 32) && ((uintptr_t)x_planes & 15) == 0 && (x_plane_stride & 7) == 0,
                "%s: x plane stride / alignment", who);
  }
  SLU_REQUIRE(D == 1 || (D == 2 && w_hh_rev && b_hh_rev), "%s: D must be 1 or 2 (with reverse weights)", who);
  SLU_REQUIRE(T > 0 && B > 0, "%s: non-positive T or B", who);
  SLU_REQUIRE(nsplit >= 1 && nsplit <= 3, "%s: nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)", who);
  if (H != 64 && H != 128)
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "%s: hidden size %lld not instantiated (64, 128)", who, (long long)H);
  SLU_REQUIRE(cdiv(B, 16) <= 65535 && B * D * 3 * H < (1LL << 31), "%s: B too large", who);
  SLU_REQUIRE(!gx || ((uintptr_t)gx & 15) == 0, "%s: gx must be 16-byte aligned", who);
  SLU_REQUIRE(((uintptr_t)b_hh_fwd & 15) == 0 && ((uintptr_t)b_hh_rev & 15) == 0, "%s: biases must be 16-byte aligned", who);
  p.xp = (const unsigned short*)x_planes; p.x_plane = x_plane_stride; p.wih = (const uint4*)w_ih_packed; p.b_ih = b_ih;
  p.gx = gx; p.w_hh[0] = w_hh_fwd; p.w_hh[1] = w_hh_rev; p.b_hh[0] = b_hh_fwd; p.b_hh[1] = b_hh_rev;
  p.out = nullptr; p.reserve = nullptr; p.planes = nullptr; p.plane = 0; p.keep = nullptr; p.keep_scale = 1.0f;
  p.T = (int)T; p.B = (int)B; p.D = (int)D;
  return SLU_OK;
}

extern "C" int slu_gru_seq_fwd_bf16(const float* gx, const float* w_hh_fwd, const float* w_hh_rev,
                                    const float* b_hh_fwd, const float* b_hh_rev, float* out, float* reserve,
                                    const void* x_planes, int64_t x_plane_stride, int64_t K, const void* w_ih_packed,
                                    const float* b_ih, int64_t T, int64_t B, int64_t H, int64_t D, int nsplit,
                                    int seq_tiles, void* stream) {
  SLU_REQUIRE(out, "slu_gru_seq_fwd_bf16: null pointer");
  GruBfParams p;
  int rc = gru_bf_common("slu_gru_seq_fwd_bf16", p, gx, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev, x_planes, x_plane_stride, K,
                         w_ih_packed, b_ih, T, B, H, D, nsplit, reserve != nullptr, seq_tiles);
  if (rc) return rc;
  p.out = out; p.reserve = reserve;
  dim3 grid((unsigned)cdiv(B, 16), (unsigned)D);
  hipStream_t st = (hipStream_t)stream;
  if (reserve) {       // bf16 forward of a trainable layer: the saved gates in the BPTT kernels' lane order
#define SLU_RS(H_, NS_) hipLaunchKernelGGL((gru_bf_fwd_rs_kernel<H_, NS_>), grid, dim3(H_ * 4), 0, st, p)
    if (H == 128) { if (nsplit == 3) SLU_RS(128, 3); else if (nsplit == 2) SLU_RS(128, 2); else SLU_RS(128, 1); }
    else { if (nsplit == 3) SLU_RS(64, 3); else if (nsplit == 2) SLU_RS(64, 2); else SLU_RS(64, 1); }
#undef SLU_RS
    SLU_CHECK_LAUNCH("gru_bf_fwd_rs_kernel");
    return SLU_OK;
  }
  SLU_REQUIRE(((uintptr_t)out & 15) == 0, "slu_gru_seq_fwd_bf16: out must be 16-byte aligned");
  const int ki = x_planes ? (K <= 32 ? 1 : 2) : 0;
  return H == 128 ? gru_bf_dispatch<128, 0>(nsplit, ki, seq_tiles, grid, st, p) : gru_bf_dispatch<64, 0>(nsplit, ki, seq_tiles, grid, st, p);
}

extern "C" int slu_gru_seq_fwd_pool_bf16(const float* gx, const float* w_hh_fwd, const float* w_hh_rev,
                                         const float* b_hh_fwd, const float* b_hh_rev, float* out_pooled,
                                         void* out_planes, int64_t out_plane_stride, const uint32_t* keep_bits, float p_drop,
                                         const void* x_planes, int64_t x_plane_stride, int64_t K, const void* w_ih_packed,
                                         const float* b_ih, int64_t T, int64_t B, int64_t H, int64_t D, int nsplit,
                                         int seq_tiles, void* stream) {
  SLU_REQUIRE((out_pooled != nullptr) != (out_planes != nullptr),
              "slu_gru_seq_fwd_pool_bf16: exactly one of out_pooled and out_planes");
  SLU_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f && (keep_bits || p_drop == 0.0f),
              "slu_gru_seq_fwd_pool_bf16: dropout p in [0, 1), with keep_bits when p > 0");
  GruBfParams p;
  int rc = gru_bf_common("slu_gru_seq_fwd_pool_bf16", p, gx, w_hh_fwd, w_hh_rev, b_hh_fwd, b_hh_rev, x_planes,
                         x_plane_stride, K, w_ih_packed, b_ih, T, B, H, D, nsplit, false, seq_tiles);
  if (rc) return rc;
  if ((D * H) % 32 != 0)
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gru_seq_fwd_pool_bf16: D * H must be a multiple of 32 (got %lld)", (long long)(D * H));
  const int64_t T_out = cdiv(T, 2);
  if (out_planes)
    SLU_REQUIRE(out_plane_stride >= T_out * B * D * H && ((uintptr_t)out_planes & 7) == 0 && (out_plane_stride & 3) == 0,
                "slu_gru_seq_fwd_pool_bf16: plane stride / alignment");
  else
    SLU_REQUIRE(((uintptr_t)out_pooled & 15) == 0, "slu_gru_seq_fwd_pool_bf16: out_pooled must be 16-byte aligned");
  p.out = out_pooled; p.planes = (unsigned short*)out_planes; p.plane = out_plane_stride;
  p.keep = p_drop > 0.0f ? keep_bits : nullptr; p.keep_scale = 1.0f / (1.0f - p_drop);
  dim3 grid((unsigned)cdiv(B, 16), (unsigned)D);
  hipStream_t st = (hipStream_t)stream;
  const int ki = x_planes ? (K <= 32 ? 1 : 2) : 0;
  if (out_planes)
    return H == 128 ? gru_bf_dispatch<128, 1>(nsplit, ki, seq_tiles, grid, st, p) : gru_bf_dispatch<64, 1>(nsplit, ki, seq_tiles, grid, st, p);
  return H == 128 ? gru_bf_dispatch<128, 2>(nsplit, ki, seq_tiles, grid, st, p) : gru_bf_dispatch<64, 2>(nsplit, ki, seq_tiles, grid, st, p);
}
