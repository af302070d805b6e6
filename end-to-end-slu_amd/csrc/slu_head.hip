// Fused intent head of the SLU model (reference models.py:709 final_classifier Linear, :112-123
// FinalPool max over time, :811-821 per-slot cross-entropy + accuracy, :839-844 per-slot argmax).
//
//   logits_t[t][b][v] = h[t][b][:] . W[v][:] + bias[v]        (T x B x V, never materialised in HBM)
//   logits[b][v]      = max_t logits_t[t][b][v]               (+ arg-max t for the backward pass)
//   loss              = sum_slots mean_b CE(logits[b][slot], y[b][slot]);  acc = mean_b [all slots right]
//
// One workgroup per utterance: its (T x C) feature rows and the (V x C) classifier are staged in LDS,
// every thread owns a few (t, v) dot products, the max over time and the per-slot softmax run on
// the first V threads.  V = 24 and T = 19 for the reference architecture: this is latency-bound
// glue, fused to replace ~35 small ATen launches per training step by 4.
#include "slu_common.h"
#include "slu_philox.h"

#include <algorithm>

namespace slu {

constexpr int HEAD_THREADS = 256;
constexpr int HEAD_MAX_SLOTS = 8;

struct HeadParams {
  const float* h;          // (T, B, C) time-major
  const float* W;          // (V, C)
  const float* bias;       // (V)
  const long long* y;      // (B, S) or null (inference: no loss)
  float* logits;           // (B, V)
  int* argmax_t;           // (B, V)
  long long* pred;         // (B, S)
  float* d_logits;         // (B, V) or null: d loss / d logits
  float* row_stats;        // (B, 2): per-utterance loss and all-slots-correct flag
  float* loss_acc;         // (2): batch loss and accuracy — written by the LAST workgroup when `ticket` is given
  double* epoch_sums;      // (2) or null: += B * (loss, acc), the epoch statistics of training.py:100-104
  unsigned int* ticket;    // zero-initialised device word (zero again afterwards) or null: separate reduce launch
  int T, B, C, V, S;
  int w_in_lds;            // 0: the classifier rows are read from L2 (T x C + V x C does not fit the LDS)
  int vec;                 // 1: C % 4 == 0 and 16-byte aligned h / W: float4 staging
  // fused Dropout of the layer below (nn.Dropout after the last intent GRU, models.py:700; vec only): h is the GRU's raw
  // output, element (t, b, c) is multiplied by its keep factor of the Philox stream (seed, offset [+ *offset_dev]) —
  // the mask slu_dropout_pool_fwd would draw — and the dropped rows go to h_drop (the backward pass reads them)
  float drop_p, drop_scale;
  unsigned long long seed, offset;
  const unsigned long long* offset_dev;
  float* h_drop;           // (T, B, C) or null
  int slot_begin[HEAD_MAX_SLOTS + 1];
};

// loss = sum_b row_loss / B (each slot's CE is a mean over the batch), acc = mean_b correct.  256 threads.
// COHERENT: the rows were written by other workgroups of the SAME launch (possibly on another XCD, i.e. behind
// another L2): read them with agent-scope loads.
template <bool COHERENT>
__device__ __forceinline__ void head_reduce_body(const float* row_stats, float* loss_acc, double* epoch_sums, int B) {
  __shared__ float r0[256], r1[256];
  float a = 0.0f, c = 0.0f;
  for (int b = threadIdx.x; b < B; b += 256) {
    if (COHERENT) {
      // one 8-byte agent-scope load per utterance: the granule its workgroup published with ONE 8-byte agent-scope store
      const unsigned long long g = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(row_stats) + b,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a += __uint_as_float((unsigned)(g & 0xffffffffull));
      c += __uint_as_float((unsigned)(g >> 32));
    } else {
      a += row_stats[2 * b]; c += row_stats[2 * b + 1];
    }
  }
  r0[threadIdx.x] = a; r1[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float loss = r0[0] / (float)B, acc = r1[0] / (float)B;
    loss_acc[0] = loss; loss_acc[1] = acc;
    if (epoch_sums) { epoch_sums[0] += (double)loss * (double)B; epoch_sums[1] += (double)acc * (double)B; }
  }
}

__global__ void __launch_bounds__(256)
head_reduce_kernel(const float* __restrict__ row_stats, float* __restrict__ loss_acc, double* epoch_sums, int B) {
  head_reduce_body<false>(row_stats, loss_acc, epoch_sums, B);
}

__global__ void __launch_bounds__(HEAD_THREADS)
head_fwd_kernel(const HeadParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sh = reinterpret_cast<float*>(smem);          // [T][C+1]
  float* sw = sh + (size_t)p.T * (p.C + 1);            // [V][C+1]
  float* sl = sw + (p.w_in_lds ? (size_t)p.V * (p.C + 1) : 0);   // [T][V] logits_t
  float* sm = sl + (size_t)p.T * p.V;                  // [V] pooled logits
  const int b = blockIdx.x, tid = threadIdx.x;
  const int T = p.T, C = p.C, V = p.V, LD = p.C + 1;
  // stage the T feature rows and the V classifier rows: ALL of a thread's loads are issued before its first LDS
  // store (48 in flight: the whole staging costs one memory round trip; on the training stream's small CU
  // partition, beside the look-ahead kernels' traffic, a round trip is 2-3 us and a load -> store chain per
  // batch of 4 or 8 elements cost 20-30 us here)
  if (p.vec) {
    // four consecutive channels per thread and load (float4): a quarter of the load instructions of the scalar path
    // below, and one Philox block per load when the dropout is fused
    const int C4 = C >> 2;
    const int nh = T * C4, nw = p.w_in_lds ? V * C4 : 0;
    const float thr = 1.0f - p.drop_p;
    const unsigned long long off = p.offset + (p.offset_dev ? *p.offset_dev : 0ull);
    constexpr int U = 13;
    for (int base = 0; base < nh + nw; base += HEAD_THREADS * U) {
      float4 v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int idx = base + j * HEAD_THREADS + tid;
        const bool ok = idx < nh + nw, isw = idx >= nh;
        const int e = ok ? (isw ? idx - nh : idx) : 0;
        const int row = e / C4, c = (e - row * C4) * 4;
        const float* src = (ok && isw) ? p.W + (size_t)row * C + c : p.h + ((size_t)row * p.B + b) * C + c;
        v[j] = *reinterpret_cast<const float4*>(src);
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int idx = base + j * HEAD_THREADS + tid;
        const bool ok = idx < nh + nw, isw = idx >= nh;
        const int e = ok ? (isw ? idx - nh : idx) : 0;
        const int row = e / C4, c = (e - row * C4) * 4;
        if (!ok) continue;
        float4 x = v[j];
        if (!isw && p.drop_p > 0.0f) {
          const size_t g = ((size_t)row * p.B + b) * C + c;                 // element index of the (T, B, C) tensor
          const float4 k = philox_keep4(p.seed, off, g, thr, p.drop_scale);
          x = make_float4(x.x * k.x, x.y * k.y, x.z * k.z, x.w * k.w);
          if (p.h_drop) *reinterpret_cast<float4*>(p.h_drop + g) = x;
        }
        float* dst = sh + (isw ? T + row : row) * LD + c;                   // sw = sh + T * LD; rows of C + 1 floats
        dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
      }
    }
  } else {
    const int nh = T * C, nw = p.w_in_lds ? V * C : 0;
    constexpr int U = 48;
    for (int base = 0; base < nh + nw; base += HEAD_THREADS * U) {
      float v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        // every lane loads unconditionally from a clamped, valid address (a branch around a load makes the
        // compiler wait for each load where it is issued)
        const int idx = base + j * HEAD_THREADS + tid;
        const bool ok = idx < nh + nw, isw = idx >= nh;
        const int e = ok ? (isw ? idx - nh : idx) : 0;
        const int row = e / C, c = e - row * C;
        const float* src = (ok && isw) ? p.W + e : p.h + ((size_t)row * p.B + b) * C + c;
        v[j] = *src;
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int idx = base + j * HEAD_THREADS + tid;
        const bool ok = idx < nh + nw, isw = idx >= nh;
        const int e = ok ? (isw ? idx - nh : idx) : 0;
        const int row = e / C, c = e - row * C;
        if (ok) sh[(isw ? T + row : row) * LD + c] = v[j];      // sw = sh + T * LD
      }
    }
  }
  __syncthreads();
  // logits_t (T x V) = h (T x C) W^T on the fp32 MFMA (16 x 16 tiles, exact fp32): wave w takes tiles w, w+4, ...
  // Rows / columns past T / V are clamped duplicates whose results are dropped.  (The previous formulation —
  // one thread per (t, v) dot product, 2 C LDS reads each — was LDS-latency bound: 30 us for 15 MFLOP.)
  {
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kg = lane >> 4;
    const int MT = (T + 15) >> 4, NT = (V + 15) >> 4;
    for (int tile = wave; tile < MT * NT; tile += HEAD_THREADS / 64) {
      const int mt = tile / NT, nt = tile - mt * NT;
      const float* a = sh + min(mt * 16 + i, T - 1) * LD + kg;
      const int vb = min(nt * 16 + i, V - 1);
      const float* w = (p.w_in_lds ? sw + vb * LD : p.W + (size_t)vb * C) + kg;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      int k0 = 0;
      for (; k0 + 31 < C; k0 += 32) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = a[k0 + 4 * u]; bv[u] = w[k0 + 4 * u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = mfma16(av[u], bv[u], acc);
      }
      for (; k0 < C; k0 += 4) {
        const bool ok = k0 + kg < C;
        acc = mfma16(ok ? a[k0] : 0.0f, ok ? w[k0] : 0.0f, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = mt * 16 + 4 * kg + r, v = nt * 16 + i;
        if (t < T && v < V) sl[t * V + v] = acc[r] + p.bias[v];
      }
    }
  }
  __syncthreads();
  if (tid < V) {
    float best = sl[tid];
    int arg = 0;
    for (int t = 1; t < T; ++t) {
      const float x = sl[t * V + tid];
      if (x > best) { best = x; arg = t; }          // first maximum wins, like torch.max
    }
    sm[tid] = best;
    p.logits[(size_t)b * V + tid] = best;
    p.argmax_t[(size_t)b * V + tid] = arg;
  }
  __syncthreads();
  // per-slot softmax / cross-entropy / arg-max: lane v of wave 0 handles classifier output v (each lane scans its own
  // slot, <= a few dozen values; one lane for everything was a ~100-expf serial chain), lane 0 then folds the slots
  __shared__ float s_part[HEAD_THREADS];               // per output: its slot's -log softmax[y] (on the slot's first lane)
  __shared__ int s_ok[HEAD_THREADS];
  if (tid < V) {
    int sl_ = 0;
    while (sl_ + 1 < p.S && tid >= p.slot_begin[sl_ + 1]) ++sl_;
    const int v0 = p.slot_begin[sl_], v1 = p.slot_begin[sl_ + 1];
    float mx = sm[v0];
    int am = v0;
    for (int v = v0 + 1; v < v1; ++v) if (sm[v] > mx) { mx = sm[v]; am = v; }       // first maximum wins
    if (tid == v0) p.pred[(size_t)b * p.S + sl_] = am - v0;
    s_part[tid] = 0.0f;
    s_ok[tid] = 1;
    if (p.y) {
      float den = 0.0f;
      for (int v = v0; v < v1; ++v) den += expf(sm[v] - mx);                          // same order for every lane
      const int yv = (int)p.y[(size_t)b * p.S + sl_];
      if (tid == v0) {
        s_part[tid] = logf(den) - (sm[v0 + yv] - mx);                                 // -log softmax[y]
        s_ok[tid] = (am - v0 == yv) ? 1 : 0;
      }
      if (p.d_logits) {
        const float inv = 1.0f / (den * (float)p.B);
        p.d_logits[(size_t)b * V + tid] = expf(sm[tid] - mx) * inv - ((tid - v0 == yv) ? 1.0f / (float)p.B : 0.0f);
      }
    }
  }
  __syncthreads();
  if (tid == 0 && p.y) {
    float loss = 0.0f;
    bool all_ok = true;
    for (int s2 = 0; s2 < p.S; ++s2) {                  // slot order: the summation order of the reference's loop
      loss += s_part[p.slot_begin[s2]];
      all_ok = all_ok && (s_ok[p.slot_begin[s2]] != 0);
    }
    // (loss, all-correct) as ONE naturally aligned 8-byte granule, written through (agent scope = sc1): the last
    // workgroup reads it with an 8-byte agent-scope load — 8-byte agent atomics on both sides need no fence (round 5:
    // the two __threadfence() of the ticket epilogue, a full L2 write-back + invalidate each, cost more than the reduce)
    const unsigned long long g = (unsigned long long)__float_as_uint(loss) |
                                 ((unsigned long long)__float_as_uint(all_ok ? 1.0f : 0.0f) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.row_stats) + b, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // batch loss / accuracy by the last workgroup to get here (same summation as head_reduce_kernel, whichever
  // workgroup runs it): saves the 1-workgroup reduce launch of every training step
  if (p.ticket && p.y) {
    __shared__ int s_last;
    if (tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the granule has left this CU before the ticket is drawn
      s_last = (__hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (s_last) {
      head_reduce_body<true>(p.row_stats, p.loss_acc, p.epoch_sums, p.B);
      if (tid == 0) *p.ticket = 0u;
    }
  }
}

// d_h[t][b][c] = g * sum_v [t == argmax_t[b][v]] d_logits[b][v] W[v][c]
// One workgroup per utterance; thread c owns column c of a (T x C) LDS accumulator and adds, for
// each classifier output v, d_logits[v] * W[v][c] into the row of v's arg-max time step.
struct HeadDrop {          // the fused dropout of slu_cls_maxpool_ce_fwd (p = 0: none)
  float p, scale;
  unsigned long long seed, offset;
  const unsigned long long* offset_dev;
};

__device__ __forceinline__ void
head_bwd_dh_body(char* smem, const int b, const float* __restrict__ d_logits, const int* __restrict__ argmax_t,
                 const float* __restrict__ W, const float* __restrict__ gscale,
                 float* __restrict__ d_h, int T, int B, int C, int V, const HeadDrop& dr) {
  float* acc = reinterpret_cast<float*>(smem);          // [T][C]
  __shared__ float s_dl[HEAD_THREADS];
  __shared__ int s_at[HEAD_THREADS];
  const int tid = threadIdx.x;
  const float g = gscale[0];
  if (tid < V) { s_dl[tid] = d_logits[(size_t)b * V + tid] * g; s_at[tid] = argmax_t[(size_t)b * V + tid]; }
  for (int idx = tid; idx < T * C; idx += HEAD_THREADS) acc[idx] = 0.0f;
  __syncthreads();
  for (int c = tid; c < C; c += HEAD_THREADS)
    for (int v = 0; v < V; ++v) acc[s_at[v] * C + c] = fmaf(s_dl[v], W[(size_t)v * C + c], acc[s_at[v] * C + c]);
  __syncthreads();
  if (dr.p > 0.0f) {
    // the gradient w.r.t. the GRU's RAW output: times the keep factors the forward pass applied (C % 4 == 0)
    const int C4 = C >> 2;
    const float thr = 1.0f - dr.p;
    const unsigned long long off = dr.offset + (dr.offset_dev ? *dr.offset_dev : 0ull);
    for (int idx = tid; idx < T * C4; idx += HEAD_THREADS) {
      const int t = idx / C4, c = (idx - t * C4) * 4;
      const size_t g = ((size_t)t * B + b) * C + c;
      const float4 k = philox_keep4(dr.seed, off, g, thr, dr.scale);
      const float* a = acc + t * C + c;
      *reinterpret_cast<float4*>(d_h + g) = make_float4(a[0] * k.x, a[1] * k.y, a[2] * k.z, a[3] * k.w);
    }
    return;
  }
  for (int idx = tid; idx < T * C; idx += HEAD_THREADS) {
    const int t = idx / C, c = idx - t * C;
    d_h[((size_t)t * B + b) * C + c] = acc[idx];
  }
}

// d_W[v][c] = g * sum_b d_logits[b][v] h[argmax_t[b][v]][b][c];   d_bias[v] = g * sum_b d_logits[b][v]
// One workgroup per classifier output v: wave w accumulates utterances b = w, w+4, ... for its 64-lane
// slices of C (fixed order -> deterministic), then the four partial rows are summed through LDS.
__device__ __forceinline__ void
head_bwd_dw_body(char* smem, const int v, const float* __restrict__ d_logits, const int* __restrict__ argmax_t,
                 const float* __restrict__ h, const float* __restrict__ gscale,
                 float* __restrict__ d_W, float* __restrict__ d_bias, int T, int B, int C, int V) {
  float* part = reinterpret_cast<float*>(smem);         // [4][C]
  float* s_dl = part + 4 * C;                           // [HEAD_THREADS] this output's d_logits, 256 utterances at a time
  int* s_at = reinterpret_cast<int*>(s_dl + HEAD_THREADS);
  __shared__ float s_b[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float g = gscale[0];
  float bsum = 0.0f;
  constexpr int KMAX = 16;                               // C <= 1024
  float acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = 0.0f;
  // The per-utterance scalars (d_logits, arg-max step) are fetched for 256 utterances at once, so that the row
  // loads below depend on LDS only and several utterances' rows are in flight (the dependent scalar -> row
  // chain per utterance cost 1.5 us x B/4).  Same per-wave summation order as before (b = w, w+4, ...).
  for (int bb = 0; bb < B; bb += HEAD_THREADS) {
    __syncthreads();
    if (bb + (int)threadIdx.x < B) {
      s_dl[threadIdx.x] = d_logits[(size_t)(bb + threadIdx.x) * V + v];
      s_at[threadIdx.x] = argmax_t[(size_t)(bb + threadIdx.x) * V + v];
    }
    __syncthreads();
    const int nb = min(HEAD_THREADS, B - bb);
    // 8 utterances' rows in flight per wave (8 x C/64 loads), accumulated in the fixed order b = w, w+4, ...
    for (int bl0 = w; bl0 < nb; bl0 += 32) {
      float hv[8][KMAX > 4 ? 4 : KMAX];
      if (C <= 256) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int bl = min(bl0 + 4 * u, nb - 1);
          const float* hr = h + ((size_t)s_at[bl] * B + (bb + bl)) * C;
#pragma unroll
          for (int k = 0; k < 4; ++k) hv[u][k] = hr[min(lane + 64 * k, C - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int bl = bl0 + 4 * u;
          if (bl < nb) {
            const float dl = s_dl[bl];
            bsum += dl;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (lane + 64 * k < C) acc[k] = fmaf(dl, hv[u][k], acc[k]);
          }
        }
      } else {
        for (int u = 0; u < 8; ++u) {
          const int bl = bl0 + 4 * u;
          if (bl >= nb) break;
          const float dl = s_dl[bl];
          const float* hr = h + ((size_t)s_at[bl] * B + (bb + bl)) * C;
          bsum += dl;
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            const int c = lane + 64 * k;
            if (c < C) acc[k] = fmaf(dl, hr[c], acc[k]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int c = lane + 64 * k;
    if (c < C) part[w * C + c] = acc[k];
  }
  if (lane == 0) s_b[w] = bsum;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += HEAD_THREADS)
    d_W[(size_t)v * C + c] = (((part[c] + part[C + c]) + part[2 * C + c]) + part[3 * C + c]) * g;
  if (threadIdx.x == 0) d_bias[v] = (((s_b[0] + s_b[1]) + s_b[2]) + s_b[3]) * g;
  (void)T;
}

// Both gradients of the head in ONE launch: workgroups [0, n_dh) = one utterance of d_h each, the rest = one
// classifier output of (d_W, d_bias) each (n_dh = 0 or B; d_W null: no such workgroups are launched).
__global__ void __launch_bounds__(HEAD_THREADS)
head_bwd_kernel(const float* __restrict__ d_logits, const int* __restrict__ argmax_t, const float* __restrict__ h,
                const float* __restrict__ W, const float* __restrict__ gscale, float* __restrict__ d_h,
                float* __restrict__ d_W, float* __restrict__ d_bias, int n_dh, int T, int B, int C, int V,
                const HeadDrop dr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < n_dh)
    head_bwd_dh_body(smem, blockIdx.x, d_logits, argmax_t, W, gscale, d_h, T, B, C, V, dr);
  else
    head_bwd_dw_body(smem, (int)blockIdx.x - n_dh, d_logits, argmax_t, h, gscale, d_W, d_bias, T, B, C, V);
}

}  // namespace slu

using namespace slu;

extern "C" int slu_cls_maxpool_ce_fwd(const float* h, const float* weight, const float* bias,
                                      const int64_t* y, const int64_t* values_per_slot,
                                      int64_t num_slots, float* logits, int32_t* argmax_t,
                                      int64_t* pred, float* d_logits, float* row_stats,
                                      float* loss_acc, double* epoch_sums, uint32_t* ticket, float drop_p,
                                      uint64_t drop_seed, uint64_t drop_offset, const uint64_t* drop_offset_dev,
                                      float* h_drop, int64_t T, int64_t B, int64_t C, void* stream) {
  SLU_REQUIRE(h && weight && bias && logits && argmax_t && pred && values_per_slot, "slu_cls_maxpool_ce_fwd: null pointer");
  SLU_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "slu_cls_maxpool_ce_fwd: dropout probability must be in [0, 1)");
  SLU_REQUIRE(num_slots >= 1 && num_slots <= HEAD_MAX_SLOTS, "slu_cls_maxpool_ce_fwd: 1..%d slots supported", HEAD_MAX_SLOTS);
  SLU_REQUIRE(!y || (row_stats && loss_acc), "slu_cls_maxpool_ce_fwd: row_stats / loss_acc required with labels");
  SLU_REQUIRE(T > 0 && B > 0 && C > 0, "slu_cls_maxpool_ce_fwd: non-positive size");
  HeadParams p;
  p.h = h; p.W = weight; p.bias = bias; p.y = (const long long*)y; p.logits = logits; p.argmax_t = argmax_t;
  p.pred = (long long*)pred; p.d_logits = d_logits; p.row_stats = row_stats;
  p.loss_acc = loss_acc; p.epoch_sums = y ? epoch_sums : nullptr; p.ticket = (unsigned int*)ticket;
  p.T = (int)T; p.B = (int)B; p.C = (int)C; p.S = (int)num_slots;
  p.vec = ((C & 3) == 0 && (((uintptr_t)h | (uintptr_t)weight | (uintptr_t)h_drop) & 15) == 0) ? 1 : 0;
  if (drop_p > 0.0f && !p.vec)
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_cls_maxpool_ce_fwd: the fused dropout needs C %% 4 == 0 (got %lld) and 16-byte aligned "
             "h / weight / h_drop", (long long)C);
  p.drop_p = drop_p; p.drop_scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
  p.seed = drop_seed; p.offset = drop_offset; p.offset_dev = (const unsigned long long*)drop_offset_dev;
  p.h_drop = drop_p > 0.0f ? h_drop : nullptr;
  int V = 0;
  for (int s = 0; s < num_slots; ++s) { p.slot_begin[s] = V; V += (int)values_per_slot[s]; }
  p.slot_begin[num_slots] = V;
  p.V = V;
  SLU_REQUIRE(V >= 1 && V <= HEAD_THREADS, "slu_cls_maxpool_ce_fwd: 1..%d classifier outputs supported", HEAD_THREADS);
  size_t lds = ((size_t)(T + V) * (C + 1) + (size_t)T * V + V) * sizeof(float);
  p.w_in_lds = 1;
  // the kernel also owns ~4 KB of STATIC LDS (slot partials, last-workgroup reduce): the dynamic part gets the rest
  const size_t lds_cap = 160 * 1024 - 4096;
  if (lds > lds_cap) {                         // wide features (H > 128): keep only the T x C rows in LDS
    p.w_in_lds = 0;
    lds = ((size_t)T * (C + 1) + (size_t)T * V + V) * sizeof(float);
  }
  if (lds > lds_cap) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_cls_maxpool_ce_fwd: T=%lld x C=%lld does not fit the LDS", (long long)T, (long long)C);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)head_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "slu_cls_maxpool_ce_fwd: %s", hipGetErrorString(e));
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)B), dim3(HEAD_THREADS), lds, st, p);
  SLU_CHECK_LAUNCH("head_fwd_kernel");
  if (y && !ticket) {
    hipLaunchKernelGGL(head_reduce_kernel, dim3(1), dim3(256), 0, st, (const float*)row_stats, loss_acc, epoch_sums, (int)B);
    SLU_CHECK_LAUNCH("head_reduce_kernel");
  }
  return SLU_OK;
}

extern "C" int slu_cls_maxpool_ce_bwd(const float* d_logits, const int32_t* argmax_t, const float* h,
                                      const float* weight, const float* grad_scale, float* d_h,
                                      float* d_weight, float* d_bias, float drop_p, uint64_t drop_seed,
                                      uint64_t drop_offset, const uint64_t* drop_offset_dev, int64_t T, int64_t B,
                                      int64_t C, int64_t V, void* stream) {
  SLU_REQUIRE(d_logits && argmax_t && h && weight && grad_scale, "slu_cls_maxpool_ce_bwd: null pointer");
  SLU_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "slu_cls_maxpool_ce_bwd: dropout probability must be in [0, 1)");
  if (drop_p > 0.0f && d_h && ((C & 3) || ((uintptr_t)d_h & 15)))
    SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_cls_maxpool_ce_bwd: the fused dropout needs C %% 4 == 0 and a 16-byte aligned d_h");
  HeadDrop dr;
  dr.p = drop_p; dr.scale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
  dr.seed = drop_seed; dr.offset = drop_offset; dr.offset_dev = (const unsigned long long*)drop_offset_dev;
  SLU_REQUIRE((d_weight == nullptr) == (d_bias == nullptr), "slu_cls_maxpool_ce_bwd: d_weight and d_bias go together");
  SLU_REQUIRE(C <= 1024, "slu_cls_maxpool_ce_bwd: C <= 1024");
  hipStream_t st = (hipStream_t)stream;
  size_t lds = 0;
  if (d_h) {
    SLU_REQUIRE(V <= HEAD_THREADS && T * C * 4 <= 160 * 1024 - 4096, "slu_cls_maxpool_ce_bwd: T*C too large for the LDS");
    lds = (size_t)T * C * sizeof(float);
  }
  if (d_weight) lds = std::max(lds, ((size_t)4 * C + 2 * HEAD_THREADS) * sizeof(float));
  if (!d_h && !d_weight) return SLU_OK;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)head_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "slu_cls_maxpool_ce_bwd: %s", hipGetErrorString(e));
  }
  const int n_dh = d_h ? (int)B : 0;
  hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)(n_dh + (d_weight ? V : 0))), dim3(HEAD_THREADS), lds, st,
                     d_logits, argmax_t, h, weight, grad_scale, d_h, d_weight, d_bias, n_dh, (int)T, (int)B, (int)C, (int)V, dr);
  SLU_CHECK_LAUNCH("head_bwd_kernel");
  return SLU_OK;
}
