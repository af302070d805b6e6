// Input projection + recurrence of a TRAINABLE GRU layer in ONE launch (reference: torch.nn.GRU at models.py:232 / :262:
// gx = x W_ih^T + b_ih for all steps, then the T-step loop).  The two used to be two launches — the exact-fp32 GEMM
// (slu_gemm.hip), then the 4-sequence persistent recurrence (slu_gru.hip) — although step t only needs rows
// [t B, (t + 1) B) of gx: for the intent layer of the default path (T = 19, B = 64) the GEMM's 21 us and a graph edge sat in
// front of a 31 us recurrence on the critical path of every training step.
//
// Here the grid holds both roles.  Workgroups [0, nrec) run the recurrence (gru_seq_fwd4_kernel's arithmetic, bit for bit);
// workgroups [nrec, nrec + ngemm) each compute one 64 x 64 tile of gx (gemm_f32_kernel<true, true, 2>'s arithmetic, bit for
// bit) in TIME order — row tile by row tile from both ends of the sequence, because the reverse direction starts at t = T - 1.
// Hand-off (cdna_hip_programming.md, the counter form of the cross-workgroup publish): a producer writes its tile with
// write-through (sc1) stores, waits for them (vmcnt 0), the workgroup barriers, and one lane adds 1 to the row tile's counter
// (relaxed, agent scope); the consumer's lane 0 polls the counter of the step it will prefetch next with a relaxed agent-scope
// load issued at the top of a step and looked at at its end (never an acquire in the loop: that would drop the CU's L1 every
// time), and reads gx with sc1 loads a step ahead of its use.  Counters never reset: a launch expects (epoch + 1) * tiles_n per
// row tile, and the launch's last workgroup (ticket) advances the epoch word.
// A consumer that waits 2^22 polls (~seconds) for a tile traps: an error, never a silent wrong result or a hang.
#include "slu_common.h"
#include "slu_gemm_tile.h"
#include <cstdlib>

namespace slu {

__device__ __forceinline__ float pj_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float pj_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

struct ProjGruParams {
  // projection: gx (M x N) = x (M x K, row stride x_rs) W^T (W: N x K, row stride w_rs) + b_ih
  const float* x; long long x_rs;
  const float* w_ih; long long w_rs;
  const float* b_ih;
  float* gx;              // (T * B, D * 3H) scratch, written and read inside this launch
  int M, N, K;
  // recurrence (as GruFwdParams)
  const float* w_hh[2];
  const float* b_hh[2];
  float* out;             // (T, B, D * H)
  float* reserve;         // [D][T][NBT16][NW16][5][64][4] or null
  int T, B, D;
  // hand-off state (device memory owned by the caller, zero at first use, reused by every launch on one stream)
  int* flags;             // one counter per 64-row tile of gx
  int* epoch;             // launches completed on this state
  unsigned* ticket;       // workgroups of the current launch that have finished
  int nrec, tiles_n, n_rt;
};

template <int H>
__global__ void __launch_bounds__(H * 2)
gru_proj_fwd4_kernel(const ProjGruParams q, const int NBT16) {
  static_assert(H * 2 == GM_THREADS, "the projection tiles are written for 256 threads");
  constexpr int NW16 = H / 16;
  constexpr int KS = H / 2;
  constexpr int NQ = KS / 8;
  constexpr int LD = H + 16;
  constexpr int GEMM_LDS = (64 + 64) * GM_KPV;                 // floats: operand tiles, reused for the C tile
  constexpr int REC_LDS = 2 * 4 * LD;
  __shared__ __attribute__((aligned(16))) float smem[GEMM_LDS > REC_LDS ? GEMM_LDS : REC_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t gxr = __builtin_amdgcn_make_buffer_rsrc(q.gx, 0, (int)((long long)q.M * q.N * 4), 0x00020000);

  if ((int)blockIdx.x >= q.nrec) {
    // ------------------------------------------------------------------ producer: one 64 x 64 tile of gx
    const int g = (int)blockIdx.x - q.nrec;
    const int idx = g / q.tiles_n, nt = g - idx * q.tiles_n;
    // time order from both ends (D == 2): 0, last, 1, last - 1, ...
    const int rt = (q.D == 2) ? ((idx & 1) ? q.n_rt - 1 - (idx >> 1) : (idx >> 1)) : idx;
    float* const sA = smem;
    float* const sB = smem + 64 * GM_KPV;
    const int wm = w >> 1, wn = w & 1;
    const int m0 = rt * 64, n0 = nt * 64;
    const int ntiles = (q.K + GM_BK - 1) / GM_BK;
    const int i = lane & 15, kg = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ra[8], rb[8];
    load_tile<true, 2>(q.x, q.x_rs, 1, m0, q.M, 0, q.K, tid, ra);
    load_tile<true, 2>(q.w_ih, q.w_rs, 1, n0, q.N, 0, q.K, tid, rb);
    for (int t = 0; t < ntiles; ++t) {
      store_tile<true, 2>(sA, tid, ra);
      store_tile<true, 2>(sB, tid, rb);
      __syncthreads();
      if (t + 1 < ntiles) {
        const int k0 = (t + 1) * GM_BK;
        load_tile<true, 2>(q.x, q.x_rs, 1, m0, q.M, k0, q.K, tid, ra);
        load_tile<true, 2>(q.w_ih, q.w_rs, 1, n0, q.N, k0, q.K, tid, rb);
      }
      float af[2][8], bf[2][8];
#pragma unroll
      for (int a = 0; a < 2; ++a) load_frag<true>(sA, wm * 32 + a * 16 + i, kg, af[a]);
#pragma unroll
      for (int b = 0; b < 2; ++b) load_frag<true>(sB, wn * 32 + b * 16 + i, kg, bf[b]);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = mfma16(af[a][kk], bf[b][kk], acc[a][b]);
      __syncthreads();
    }
    // C tile through LDS, whole 256-byte rows, + bias (the arithmetic of gemm_f32_kernel's row-major epilogue)
    constexpr int LDC = 64 + 4;
    float* __restrict__ sC = sA;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sC[(wm * 32 + a * 16 + 4 * kg + r) * LDC + wn * 32 + b * 16 + i] = acc[a][b][r];
    __syncthreads();
    const int col = 4 * (tid & 15);
    const int n = n0 + col;                                    // N % 64 == 0 (launcher): whole float4s
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q.b_ih) { bv.x = q.b_ih[n]; bv.y = q.b_ih[n + 1]; bv.z = q.b_ih[n + 2]; bv.w = q.b_ih[n + 3]; }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int row = (tid >> 4) + 16 * h;
      const int m = m0 + row;
      if (m >= q.M) continue;
      const float4 v = *reinterpret_cast<const float4*>(&sC[row * LDC + col]);
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 o = {__float_as_uint(v.x + bv.x), __float_as_uint(v.y + bv.y), __float_as_uint(v.z + bv.z), __float_as_uint(v.w + bv.w)};
      __builtin_amdgcn_raw_buffer_store_b128(o, gxr, (int)(((long long)m * q.N + n) * 4), 0, 16);     // sc1: write-through
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(q.flags + rt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    // ------------------------------------------------------------------ consumer: the 4-sequence recurrence
    float (*hbuf)[4 * LD] = reinterpret_cast<float (*)[4 * LD]>(smem);
    const int NBT4 = q.nrec / q.D;
    const int dir = (int)blockIdx.x / NBT4;
    const int b0 = ((int)blockIdx.x - dir * NBT4) * 4;
    const int u = lane & 31, half = lane >> 5, blk = (lane >> 2) & 7, si = lane & 3;
    const int j = w * 32 + u;
    const int T = q.T, B = q.B, D = q.D;
    const int target = (*q.epoch + 1) * q.tiles_n;              // stable during the launch: the epoch moves at its very end

    float wr[KS], wz[KS], wn[KS];
    {
      const float* __restrict__ W = q.w_hh[dir];
      const float* pr = W + ((size_t)(0 * H + j)) * H + half * KS;
      const float* pz = W + ((size_t)(1 * H + j)) * H + half * KS;
      const float* pn = W + ((size_t)(2 * H + j)) * H + half * KS;
#pragma unroll
      for (int k = 0; k < KS; ++k) { wr[k] = pr[k]; wz[k] = pz[k]; wn[k] = pn[k]; }
    }
    const float bhr = q.b_hh[dir][j], bhz = q.b_hh[dir][H + j], bhn = q.b_hh[dir][2 * H + j];
    for (int x = tid; x < 2 * 4 * LD; x += H * 2) (&hbuf[0][0])[x] = 0.0f;   // h0 = 0
    float hprev[2] = {0.f, 0.f};
    bool rowok[2];
    int grow[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int b = b0 + 2 * half + e;
      rowok[e] = b < B;
      grow[e] = rowok[e] ? b : 0;
    }
    const size_t out_ts = (size_t)B * D * H;
    float* __restrict__ outd = q.out + (size_t)dir * H + j;
    const int gcol = dir * 3 * H + j;                           // this lane's r column of gx; z at + H, n at + 2H
    auto tstep = [&](int s) { return dir ? T - 1 - s : s; };
    // rows [t B + b0, t B + b0 + 4) of gx lie in ONE 64-row tile (b0 and 64 are multiples of 4)
    auto rtile = [&](int t) { return (t * B + b0) >> 6; };
    auto poll = [&](int s) -> int {                              // lane 0 only: the counter of step s's row tile (s < T)
      return __hip_atomic_load(q.flags + rtile(tstep(s)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto wait_for = [&](int s, int seen) {                       // lane 0 only
      int spins = 0;
      while (seen < target) {
        __builtin_amdgcn_s_sleep(1);
        seen = poll(s);
        if (++spins > (1 << 22)) __builtin_trap();
      }
    };
    auto gload = [&](int t, float (&r)[2], float (&z)[2], float (&n)[2]) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int off = (int)((((long long)t * B + grow[e]) * q.N + gcol) * 4);
        r[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gxr, off, 0, 16));                 // sc1
        z[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gxr, off + 4 * H, 0, 16));
        n[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gxr, off + 8 * H, 0, 16));
      }
    };
    // steps 0 and 1 must be there before their loads; step 2's counter is requested at the top of step 0 and looked at at its
    // end.  (A two-deep version — an unrolled pair of steps, gx and counters requested two steps ahead — needed 260 registers
    // and was slower: 67 vs 58 us for the intent layer; profiles/r04_ap_proj_gru_fused.txt.)
    if (tid == 0) {
      wait_for(0, poll(0));
      if (T > 1) wait_for(1, poll(1));
    }
    __syncthreads();
    float gr[2], gz[2], gn[2], g1r[2], g1z[2], g1n[2];
    gload(tstep(0), gr, gz, gn);
    if (T > 1) gload(tstep(1), g1r, g1z, g1n);
    else {
#pragma unroll
      for (int e = 0; e < 2; ++e) { g1r[e] = 0.f; g1z[e] = 0.f; g1n[e] = 0.f; }
    }
    const size_t rsv_lane = ((size_t)((j & 15) + 16 * ((b0 & 15) >> 2))) * 4 + 2 * half;
    const size_t rsv_wave = (size_t)(b0 >> 4) * NW16 + (j >> 4);

    for (int s = 0; s < T; ++s) {
      const int t = tstep(s);
      const int cur = s & 1;
      // the counter of step s + 2 (requested here, needed at the END of this step: the round trip hides behind the step)
      int seen = target;
      if (tid == 0 && s + 2 < T) seen = poll(s + 2);
      float af[NQ];
      {
        const float* __restrict__ hrow = &hbuf[cur][si * LD + half * KS + 4 * blk];
#pragma unroll
        for (int v = 0; v < NQ / 4; ++v) {
          const float4 x = *reinterpret_cast<const float4*>(hrow + 32 * v);
          af[4 * v + 0] = x.x; af[4 * v + 1] = x.y; af[4 * v + 2] = x.z; af[4 * v + 3] = x.w;
        }
      }
      f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = {0.f, 0.f, 0.f, 0.f}, an = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int qq_ = 0; qq_ < NQ; ++qq_) {
        const int kb = (qq_ / 4) * 32 + (qq_ % 4);
#define SLU_PJ_STEP(a)                                                              \
        ar = __builtin_amdgcn_mfma_f32_4x4x1f32(af[qq_], wr[kb + 4 * a], ar, 3, a, 0); \
        az = __builtin_amdgcn_mfma_f32_4x4x1f32(af[qq_], wz[kb + 4 * a], az, 3, a, 0); \
        an = __builtin_amdgcn_mfma_f32_4x4x1f32(af[qq_], wn[kb + 4 * a], an, 3, a, 0);
        SLU_PJ_STEP(0) SLU_PJ_STEP(1) SLU_PJ_STEP(2) SLU_PJ_STEP(3)
        SLU_PJ_STEP(4) SLU_PJ_STEP(5) SLU_PJ_STEP(6) SLU_PJ_STEP(7)
#undef SLU_PJ_STEP
      }
      float hr[2], hz[2], hq[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        auto sr = __builtin_amdgcn_permlane32_swap(__float_as_uint(ar[e]), __float_as_uint(ar[e + 2]), false, false);
        auto sz = __builtin_amdgcn_permlane32_swap(__float_as_uint(az[e]), __float_as_uint(az[e + 2]), false, false);
        auto sn = __builtin_amdgcn_permlane32_swap(__float_as_uint(an[e]), __float_as_uint(an[e + 2]), false, false);
        hr[e] = __uint_as_float(sr[0]) + __uint_as_float(sr[1]);
        hz[e] = __uint_as_float(sz[0]) + __uint_as_float(sz[1]);
        hq[e] = __uint_as_float(sn[0]) + __uint_as_float(sn[1]);
      }
      float rr[2], zz[2], nn[2], qv[2], hn[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        rr[e] = pj_sigmoid(gr[e] + (hr[e] + bhr));
        zz[e] = pj_sigmoid(gz[e] + (hz[e] + bhz));
        qv[e] = hq[e] + bhn;
        nn[e] = pj_tanh(gn[e] + rr[e] * qv[e]);
        hn[e] = (1.0f - zz[e]) * nn[e] + zz[e] * hprev[e];
      }
      float* __restrict__ hnext = &hbuf[cur ^ 1][0];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        hnext[(2 * half + e) * LD + j] = hn[e];
        if (rowok[e]) outd[(size_t)t * out_ts + (size_t)grow[e] * D * H] = hn[e];
      }
      if (q.reserve) {
        float* __restrict__ rs = q.reserve + ((((size_t)dir * T + t) * NBT16) * NW16 + rsv_wave) * (5 * 256) + rsv_lane;
        *reinterpret_cast<float2*>(rs + 0 * 256) = make_float2(rr[0], rr[1]);
        *reinterpret_cast<float2*>(rs + 1 * 256) = make_float2(zz[0], zz[1]);
        *reinterpret_cast<float2*>(rs + 2 * 256) = make_float2(nn[0], nn[1]);
        *reinterpret_cast<float2*>(rs + 3 * 256) = make_float2(qv[0], qv[1]);
        *reinterpret_cast<float2*>(rs + 4 * 256) = make_float2(hprev[0], hprev[1]);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) { hprev[e] = hn[e]; gr[e] = g1r[e]; gz[e] = g1z[e]; gn[e] = g1n[e]; }
      // step s + 2's tile: lane 0 makes sure of it BEFORE the barrier, everybody requests its gx right after
      if (tid == 0 && s + 2 < T) wait_for(s + 2, seen);
      __syncthreads();
      if (s + 2 < T) gload(tstep(s + 2), g1r, g1z, g1n);
    }
  }
  // the launch's last workgroup advances the epoch (every consumer has read it long ago: before its first step)
  __syncthreads();
  if (tid == 0) {
    const unsigned done = atomicAdd(q.ticket, 1u);
    if (done == gridDim.x - 1) {
      *q.epoch = *q.epoch + 1;
      *q.ticket = 0u;
    }
  }
}

}  // namespace slu

using namespace slu;

// Can slu_gru_proj_seq_fwd take this layer?  (H = 128, the 4-sequence recurrence geometry, whole 64-column tiles.)
extern "C" int slu_gru_proj_supported(int64_t T, int64_t B, int64_t I, int64_t H, int64_t D) {
  if (H != 128 || (D != 1 && D != 2) || T < 1 || B < 1 || I < 1) return 0;
  if ((D * 3 * H) % 64 != 0 || B % 4 != 0) return 0;
  if (cdiv(B, 16) * D >= 256) return 0;                       // the 16-sequence kernel's territory (gru_use_seq4)
  // the recurrence workgroups [0, nrec) SPIN on counters the projection workgroups behind them publish: both sets must
  // be resident together.  The smallest CU partition a stream of this package gets is 64 CUs at one workgroup per CU
  // (pipeline.cu_split: multiples of 16, the opt-in is refused below 64), so at most 32 consumers may be launched — the
  // other half of the slots is then always free for producers (B <= 64 bidirectional: the intent layer's shape)
  if (cdiv(B, 4) * D > 32) return 0;
  if (T * B * D * 3 * H * 4 >= (1LL << 31)) return 0;          // gx through one buffer descriptor
  return 1;
}

extern "C" int64_t slu_gru_proj_state_words(int64_t T, int64_t B) { return cdiv(T * B, 64) + 2; }

extern "C" int slu_gru_proj_seq_fwd(const float* x, int64_t x_rs, const float* w_ih, int64_t w_rs, const float* b_ih,
                                    float* gx_scratch, const float* w_hh_fwd, const float* w_hh_rev,
                                    const float* b_hh_fwd, const float* b_hh_rev, float* out, float* reserve,
                                    int64_t T, int64_t B, int64_t I, int64_t H, int64_t D,
                                    int32_t* state, int64_t state_words, void* stream) {
  SLU_REQUIRE(x && w_ih && gx_scratch && w_hh_fwd && b_hh_fwd && out && state, "slu_gru_proj_seq_fwd: null pointer");
  SLU_REQUIRE(D == 1 || (w_hh_rev && b_hh_rev), "slu_gru_proj_seq_fwd: reverse weights missing");
  if (!slu_gru_proj_supported(T, B, I, H, D))
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_gru_proj_seq_fwd: unsupported shape T=%lld B=%lld I=%lld H=%lld D=%lld",
             (long long)T, (long long)B, (long long)I, (long long)H, (long long)D);
  SLU_REQUIRE(state_words >= slu_gru_proj_state_words(T, B), "slu_gru_proj_seq_fwd: state buffer too small");
  SLU_REQUIRE(x_rs >= I && w_rs >= I, "slu_gru_proj_seq_fwd: bad row stride");
  ProjGruParams q;
  q.x = x; q.x_rs = x_rs; q.w_ih = w_ih; q.w_rs = w_rs; q.b_ih = b_ih; q.gx = gx_scratch;
  q.M = (int)(T * B); q.N = (int)(D * 3 * H); q.K = (int)I;
  q.w_hh[0] = w_hh_fwd; q.w_hh[1] = w_hh_rev; q.b_hh[0] = b_hh_fwd; q.b_hh[1] = b_hh_rev;
  q.out = out; q.reserve = reserve; q.T = (int)T; q.B = (int)B; q.D = (int)D;
  q.n_rt = (int)cdiv(T * B, 64);
  q.tiles_n = q.N / 64;
  q.nrec = (int)(cdiv(B, 4) * D);
  q.flags = state; q.epoch = state + q.n_rt; q.ticket = reinterpret_cast<unsigned*>(state + q.n_rt + 1);
  const int ngemm = q.n_rt * q.tiles_n;
  hipLaunchKernelGGL(gru_proj_fwd4_kernel<128>, dim3((unsigned)(q.nrec + ngemm)), dim3(256), 0, (hipStream_t)stream, q,
                     (int)cdiv(B, 16));
  SLU_CHECK_LAUNCH("gru_proj_fwd4_kernel");
  return SLU_OK;
}
