// Split-precision persistent GRU recurrence (forward only, no saved gates) for FROZEN layers:
// torch.nn.GRU semantics (models.py:232/:262: h0 = 0, gates [r; z; n], optional reverse direction), the
// hidden x hidden contraction on v_mfma_f32_16x16x32_bf16 with W_hh and h_{t-1} each split into NS bf16
// 16-bit terms (slu_bf16.h): NS = 3 (bf16x3) keeps six products (fp32-class result at 6/16 of the fp32-MFMA cycles),
// NS = 2 (f16x2: two fp16 terms, v_mfma_f32_16x16x32_f16) three products (fp32-class at 3/16), NS = 1 is plain bf16
// (BASELINE configs[4]).
//
// Geometry = gru_seq_fwd_kernel's: grid (16-sequence tile) x (direction), H/16 waves, wave w owns hidden
// units [16w, 16w+16) of all three gates, so each lane ends up with r, z, n of the same (sequence, unit) and
// the gate math is fused in registers.
//   * the wave's W_hh slice — 3 gates x H/32 k-chunks x NS planes of 8 bf16 per lane — is split once at
//     kernel start and stays RESIDENT in VGPRs (144 registers for H = 128, NS = 3) for all T steps;
//   * h_{t-1} lives in LDS as NS bf16 planes (double buffered, 16-byte slots XOR-swizzled by the row: the
//     ds_read_b128 fragments and the 2-byte stores are conflict-free), in fp32 in the owning lane for the blend;
//   * per step and wave 3 x H/32 x (6 | 1) MFMAs of 16 cycles on six accumulator chains:
//     72 x 16 = 1152 cycles per wave, 2304 per SIMD (two waves) for SIXTEEN sequences, against 1536 cycles for
//     FOUR sequences on the fp32 4x4x1 kernel: 2.7x the sequences per CU-cycle.
#include "slu_bf16.h"   // round-3 kernel kept as an A/B baseline (tools/build_alt.sh EXTRA_UNITS); not part of the product
#include <stdlib.h>

namespace slu {

__device__ __forceinline__ float bf_sigmoid_r3(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float bf_tanh_r3(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

#ifdef SLU_GRU_PROBE
#define SLU_BDBG(bit) (p.dbg & (bit))
#else
#define SLU_BDBG(bit) 0
#endif

struct GruBfParamsR3 {
#ifdef SLU_GRU_PROBE
  int dbg;                // ablation mask of the probe build (tools/gru_probe.py): never compiled into the product
#endif
  const float* gx;        // (T, B, D*3H); unused by the fused-input kernels
  // fused input projection (KI > 0: K <= 32 KI input channels): x as NS planes of (T*B) x (32 KI) 16-bit terms
  // (plane stride x_plane elements), W_ih packed by gemm_bf_pack_kernel for N = D*3H columns, b_ih (D*3H)
  const unsigned short* xp; long long x_plane;
  const uint4* wih; const float* b_ih;
  const float* w_hh[2];   // (3H, H) fp32
  const float* b_hh[2];   // (3H)
  float* out;             // (T, B, D*H)
  float* reserve;         // null, or the saved gates in gru_seq_bwd_kernel's layout [D][T][NBT][NW][5][64][4]
  int T, B, D;
};

// KI > 0: the input projection x_t W_ih^T + b_ih is computed HERE instead of being read as gx — for layers whose input has
// at most 32 KI channels (the first GRU layer: K = 60) the wave's W_ih slice fits beside W_hh (3 gates x KI chunks x NS
// planes = 48 registers for KI = 2 on f16x2; the registers that prefetched gx are free), the A fragments of x_t are read
// straight from the previous stage's planes (one 16-byte load per chunk and plane, a step ahead), and the extra MFMAs do
// not depend on h_{t-1}.  Saves the projection GEMM and the fp32 gx round trip (T*B*D*3H*8 bytes) of that layer.
template <int H, int NS, int KI>
__global__ void __launch_bounds__(H * 4)
gru_bf_fwd_r3_kernel(const GruBfParamsR3 p) {
  constexpr int NW = H / 16;          // waves
  constexpr int KC = H / 32;          // 32-wide k-chunks
  constexpr int ROWB = H * 2;         // bytes per LDS row (one sequence, one plane)
  constexpr int SLOTS = H / 8;        // 16-byte slots per row
  typedef Split<NS> SP;
  __shared__ __attribute__((aligned(16))) unsigned char hbuf[2][NS][16 * ROWB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kg = lane >> 4;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const int j = w * 16 + i;           // hidden unit of this lane's outputs
  const int T = p.T, B = p.B, D = p.D;

  // resident W_hh fragments: wb[g][c][pl] = 8 bf16 of W_hh[g*H + j][c*32 + kg*8 .. +7], plane pl
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
  // fused input projection: the wave's W_ih fragments (tile dir * 3H/16 + g * H/16 + w of the packed matrix) and biases
  constexpr int KIA = KI > 0 ? KI : 1;
  uint4 wi[3][KIA][NS];
  float bir = 0.f, biz = 0.f, bin = 0.f;
  if constexpr (KI > 0) {
    const int NTI = p.D * 3 * NW;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < KI; ++c)
#pragma unroll
        for (int pl = 0; pl < NS; ++pl)
          wi[g][c][pl] = p.wih[(size_t)pl * KI * NTI * 64 + ((size_t)c * NTI + dir * 3 * NW + g * NW + w) * 64 + lane];
    const float* bi = p.b_ih + (size_t)dir * 3 * H + j;
    bir = bi[0]; biz = bi[H]; bin = bi[2 * H];
  }
  // x_t W_ih^T + b_ih for this lane's four (sequence, unit) pairs from the A fragments of x_t: the accumulation order
  // of gemm_bf_panel_kernel (k-chunks outside, products inside), i.e. bit-identical to the gx the GEMM would write
  auto xproj = [&](const uint4 (&xa)[KIA][NS], float (&o_r)[4], float (&o_z)[4], float (&o_n)[4]) {
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
          ax[SP::ACC(q)][g] = mfma_split<NS>(xa[c][SP::PA(q)], wi[g][c][SP::PB(q)], ax[SP::ACC(q)][g]);
    const f32x4 vr = split_result<NS>(ax[0][0], ax[SP::NACC - 1][0]), vz = split_result<NS>(ax[0][1], ax[SP::NACC - 1][1]),
                vn = split_result<NS>(ax[0][2], ax[SP::NACC - 1][2]);
#pragma unroll
    for (int r = 0; r < 4; ++r) { o_r[r] = vr[r] + bir; o_z[r] = vz[r] + biz; o_n[r] = vn[r] + bin; }
  };
  // A fragment of x_t for this lane: row (sequence) b0 + i (clamped), k slice kg of chunk c, plane pl
  const int xrow = min(b0 + i, p.B - 1);
  auto xload = [&](int t_, uint4 (&xa)[KIA][NS]) {
    const unsigned short* base = p.xp + ((size_t)t_ * p.B + xrow) * (32 * KIA) + kg * 8;
#pragma unroll
    for (int c = 0; c < KIA; ++c)
#pragma unroll
      for (int pl = 0; pl < NS; ++pl)
        xa[c][pl] = *reinterpret_cast<const uint4*>(base + (size_t)pl * p.x_plane + c * 32);
  };

  for (int x = tid; x < 2 * NS * 16 * ROWB / 4; x += H * 4) reinterpret_cast<unsigned*>(&hbuf[0][0][0])[x] = 0u;   // h0 = 0
  float hprev[4] = {0.f, 0.f, 0.f, 0.f};
  // rows b0 + 4 kg + r of this lane: 32-bit offsets inside one time step (B * D * 3H < 2^31 is checked by the
  // launcher); rows past B read row 0 and are never stored (oob row offset -1)
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
  // A-fragment read: row i (sequence), slot (c*4 + kg) ^ i;  h store: row 4*kg + r, slot (j/8) ^ row, element j%8
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
    if constexpr (KI > 0) {
      uint4 xa0[KIA][NS];
      xload(t0, xa0);
      xproj(xa0, gr, gz, gn);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* g = gxd + (size_t)t0 * gx_ts + g_off[r];
        gr[r] = g[0]; gz[r] = g[H]; gn[r] = g[2 * H];
      }
    }
  }
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const int cur = s & 1;
    float ngr[4], ngz[4], ngn[4];
    uint4 xan[KIA][NS];                                          // fused input: A fragments of the NEXT step's x
    const int tn = (s + 1 < T) ? (dir ? t - 1 : t + 1) : t;      // last step: re-reads its own row (unused)
    if constexpr (KI > 0) {
      xload(tn, xan);                                            // in flight during this step; multiplied at its end
    } else if (SLU_BDBG(1)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { ngr[r] = gr[r]; ngz[r] = gz[r]; ngn[r] = gn[r]; }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* g = gxd + (size_t)tn * gx_ts + g_off[r];
        ngr[r] = g[0]; ngz[r] = g[H]; ngn[r] = g[2 * H];
      }
    }
    // one accumulator chain per gate (two for f16x2: the 2^11-scaled cross terms): three or six independent
    // chains per wave, two waves per SIMD
    f32x4 accs[SP::NACC][3];
#pragma unroll
    for (int a = 0; a < SP::NACC; ++a)
#pragma unroll
      for (int g = 0; g < 3; ++g) accs[a][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!SLU_BDBG(16))
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      uint4 fa[NS];                     // (W_hh takes 144 of the 256 registers: one fragment set at a time)
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) fa[pl] = *reinterpret_cast<const uint4*>(&hbuf[cur][pl][a_off[c]]);
#pragma unroll
      for (int q = 0; q < SP::NPAIR; ++q) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
          accs[SP::ACC(q)][g] = mfma_split<NS>(fa[SP::PA(q)], wb[g][c][SP::PB(q)], accs[SP::ACC(q)][g]);
      }
    }
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = split_result<NS>(accs[0][g], accs[SP::NACC - 1][g]);

    float hn[4], rr[4], zz[4], nn[4], qq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (SLU_BDBG(8)) {       // probe: gate math without the transcendentals
        rr[r] = 0.5f + 0.01f * (gr[r] + (acc[0][r] + bhr)); zz[r] = 0.5f + 0.01f * (gz[r] + (acc[1][r] + bhz));
        qq[r] = acc[2][r] + bhn; nn[r] = 0.01f * (gn[r] + rr[r] * qq[r]);
      } else {
      rr[r] = bf_sigmoid_r3(gr[r] + (acc[0][r] + bhr));
      zz[r] = bf_sigmoid_r3(gz[r] + (acc[1][r] + bhz));
      qq[r] = acc[2][r] + bhn;
      nn[r] = bf_tanh_r3(gn[r] + rr[r] * qq[r]);
      }
      hn[r] = (1.0f - zz[r]) * nn[r] + zz[r] * hprev[r];
    }
    if (p.reserve) {      // trainable layer (bf16 forward, fp32 BPTT): the gates the exact BPTT kernels read
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
      if (SLU_BDBG(32)) {      // probe: one rounding instead of the NS-way split
        sp[0] = f32_to_bf16_rne(hn[r]);
#pragma unroll
        for (int pl = 1; pl < NS; ++pl) sp[pl] = 0;
      } else {
        split_terms<NS>(hn[r], sp);
      }
#pragma unroll
      for (int pl = 0; pl < NS; ++pl)
        if (!SLU_BDBG(64)) *reinterpret_cast<unsigned short*>(hnext + pl * (16 * ROWB) + h_off[r]) = sp[pl];
      if (o_off[r] >= 0 && !SLU_BDBG(2)) outd[(size_t)t * out_ts + o_off[r]] = hn[r];
    }
    // the next step's x W_ih^T + b_ih: independent of h.  (Measured alternatives, T = 300 x 1024 sequences, this
    // placement 511 us: issued before the gate math and interleaved with it by sched_group_barrier hints, one MFMA per six
    // VALU instructions — 662 us; THIS step's projection at the top of the step, under the h fragments' LDS latency, with
    // the fragments carried across the barrier — 565 us, 256 VGPRs and 8 spilled.)
    if constexpr (KI > 0) xproj(xan, ngr, ngz, ngn);
#pragma unroll
    for (int r = 0; r < 4; ++r) { hprev[r] = hn[r]; gr[r] = ngr[r]; gz[r] = ngz[r]; gn[r] = ngn[r]; }
    __syncthreads();
  }
}

}  // namespace slu

using namespace slu;

extern "C" int slu_gru_seq_fwd_bf16_r3(const float* gx, const float* w_hh_fwd, const float* w_hh_rev,
                                    const float* b_hh_fwd, const float* b_hh_rev, float* out, float* reserve,
                                    const void* x_planes, int64_t x_plane_stride, int64_t K, const void* w_ih_packed,
                                    const float* b_ih, int64_t T, int64_t B, int64_t H, int64_t D, int nsplit,
                                    void* stream) {
  SLU_REQUIRE((gx || x_planes) && w_hh_fwd && b_hh_fwd && out, "slu_gru_seq_fwd_bf16_r3: null pointer");
  const bool fused = x_planes != nullptr;
  if (fused) {
    SLU_REQUIRE(!gx && w_ih_packed && b_ih && !reserve, "slu_gru_seq_fwd_bf16_r3: the fused input projection takes x_planes, "
                "w_ih_packed and b_ih instead of gx, and no reserve (frozen layers only)");
    if (!(nsplit == 2 && H == 128 && K >= 1 && K <= 64))
      SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gru_seq_fwd_bf16_r3: the fused input projection is instantiated for f16x2 (nsplit 2), "
               "H = 128 and at most 64 input channels (got nsplit %d, H %lld, K %lld)", nsplit, (long long)H, (long long)K);
    SLU_REQUIRE(x_plane_stride >= T * B * (cdiv(K, 32) * 32) && ((uintptr_t)x_planes & 15) == 0 && (x_plane_stride & 7) == 0,
                "slu_gru_seq_fwd_bf16_r3: x plane stride / alignment");
  }
  SLU_REQUIRE(D == 1 || (D == 2 && w_hh_rev && b_hh_rev), "slu_gru_seq_fwd_bf16_r3: D must be 1 or 2 (with reverse weights)");
  SLU_REQUIRE(T > 0 && B > 0, "slu_gru_seq_fwd_bf16_r3: non-positive T or B");
  SLU_REQUIRE(nsplit >= 1 && nsplit <= 3, "slu_gru_seq_fwd_bf16_r3: nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  if (H != 64 && H != 128)
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gru_seq_fwd_bf16_r3: hidden size %lld not instantiated (64, 128)", (long long)H);
  SLU_REQUIRE(cdiv(B, 16) <= 65535 && B * D * 3 * H < (1LL << 31), "slu_gru_seq_fwd_bf16_r3: B too large");
  GruBfParamsR3 p;
#ifdef SLU_GRU_PROBE
  { const char* e = getenv("SLU_GRU_DBG"); p.dbg = e ? atoi(e) : 0; }
#endif
  p.xp = (const unsigned short*)x_planes; p.x_plane = x_plane_stride; p.wih = (const uint4*)w_ih_packed; p.b_ih = b_ih;
  p.gx = gx; p.w_hh[0] = w_hh_fwd; p.w_hh[1] = w_hh_rev; p.b_hh[0] = b_hh_fwd; p.b_hh[1] = b_hh_rev;
  p.out = out; p.reserve = reserve; p.T = (int)T; p.B = (int)B; p.D = (int)D;
  dim3 grid((unsigned)cdiv(B, 16), (unsigned)D);
  hipStream_t st = (hipStream_t)stream;
  if (fused) {
    if (K <= 32) hipLaunchKernelGGL((gru_bf_fwd_r3_kernel<128, 2, 1>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gru_bf_fwd_r3_kernel<128, 2, 2>), grid, dim3(512), 0, st, p);
  } else if (H == 128) {
    if (nsplit == 3) hipLaunchKernelGGL((gru_bf_fwd_r3_kernel<128, 3, 0>), grid, dim3(512), 0, st, p);
    else if (nsplit == 2) hipLaunchKernelGGL((gru_bf_fwd_r3_kernel<128, 2, 0>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gru_bf_fwd_r3_kernel<128, 1, 0>), grid, dim3(512), 0, st, p);
  } else {
    if (nsplit == 3) hipLaunchKernelGGL((gru_bf_fwd_r3_kernel<64, 3, 0>), grid, dim3(256), 0, st, p);
    else if (nsplit == 2) hipLaunchKernelGGL((gru_bf_fwd_r3_kernel<64, 2, 0>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gru_bf_fwd_r3_kernel<64, 1, 0>), grid, dim3(256), 0, st, p);
  }
  SLU_CHECK_LAUNCH("gru_bf_fwd_r3_kernel");
  return SLU_OK;
}
