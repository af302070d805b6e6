// slu_comm_ipc_*: the data-parallel step's gradient all-reduce as ONE hand-written kernel over peer-mapped device
// memory — the xGMI-native shape SURVEY §5 / §8(e) names — instead of a library collective.  The reference has no
// distributed code at all: new design.
//
// Why: the payload is 1.2 MB (frozen encoder) … 5.5 MB (everything trainable) per 0.15–2.7 ms step, i.e. the
// collective is LATENCY-bound.  xGMI is point to point (7 links per GPU), so a two-shot all-reduce can use all links at
// once with two flag hand-offs, where a ring pays 2 (N − 1) hops:
//   window of rank r (fine-grained device memory, exported with hipIpcGetMemHandle, mapped by every peer):
//       [flags 4 KiB | `in` staging (cap bytes) | `out` staging (cap bytes)]
//   1. every rank copies its bucket into its own `in` (local, written through), raises flag_in[r] on every
//      window (its own too: a workgroup reads what the rank's other workgroups wrote only behind the rank's own flag);
//   2. waits for all N flag_in; rank r then OWNS chunk r: it reads chunk r of every rank's `in` over the links
//      (N − 1 remote reads of payload / N each, all links busy at once), adds them IN RANK ORDER 0 … N−1 (one rank
//      computes each element, so the replicas receive bit-identical sums) and WRITES the sum into chunk r of every
//      rank's `out` (N − 1 remote writes, written through); raises flag_out[r] on every window;
//   3. waits for all N flag_out, copies its own `out` back into the bucket (local).
// The fp32 bucket and the 160-element float64 bucket of the Sinc parameters travel as TYPED SEGMENTS of the same
// payload: one launch, one collective per step whatever the trainable set.
// Flags hold a monotonically increasing epoch (never reset), the epoch counter lives in device memory: the launch has
// no per-call host argument and replays as a node of the step's hipGraph.  Every wait is bounded (about a minute): a peer that never
// arrives raises the window's status word instead of hanging the GPU (slu_comm_ipc_status).
// A rank's `in` may be overwritten by its next call only after every peer has read it: a peer raises flag_out AFTER
// its reads, and a rank leaves step 3 only after it has seen every peer's flag_out.  A rank's `out` is written by the
// peers of call k + 1 only after they saw its flag_in of call k + 1, which it raises after finishing call k.
#include "slu_common.h"
#include <string.h>

namespace slu {

constexpr int IPC_MAX_RANKS = 8;
constexpr long long IPC_FLAG_BYTES = 4096;
constexpr int IPC_WGS = 64;                 // all resident at once on any partition of this package (>= 16 CUs, 8 per CU)
// Polls of ~1 us each before a wait gives up and raises the status word: about a minute.  Not seconds: ranks that SHARE a
// GPU (the test set-up) were seen to be descheduled for more than two seconds now and then (status raised in 1 of ~10
// four-rank runs with a 2^21 limit; the longest wait of a call is kept in the window, slu_comm_ipc_max_wait).
constexpr unsigned IPC_SPIN_LIMIT = 1u << 26;
// ... which is the START-UP limit (a rank whose first matmul initialises rocBLAS arrives seconds late).  Once a job is
// running the waits are microseconds, and 64 workgroups spinning for a minute on every peer's training partition is a poor
// way to learn that a rank has died: slu_comm_ipc_set_spin_limit lowers the limit of a window (word 23; 0 = this default).

// window-relative offsets of the control words (each on a 64-byte line of its own)
//   flag_in[src]  at 64 * src            flag_out[src] at 64 * (8 + src)
//   epoch         at 64 * 16             arrive[0..2]  at 64 * (17 + k)           status at 64 * 20
__device__ __forceinline__ unsigned long long* ipc_word(unsigned char* win, int line) {
  return reinterpret_cast<unsigned long long*>(win + 64 * line);
}

struct IpcArgs {
  unsigned char* win[IPC_MAX_RANKS];        // every rank's window in THIS process' address space; win[rank] = own
  int rank, nranks;
  long long cap;                            // capacity of each staging area (bytes, multiple of 256)
  float* f32; long long n32;                // the fp32 bucket (reduced in place)
  double* f64; long long n64;               // the float64 bucket or null
};

// Staging traffic is SYSTEM-SCOPE on both sides: 16-byte buffer stores / loads with sc0 sc1 (aux 17: written through, never
// served from a stale cache line) — "sc0 sc1 stores and loads on both sides" needs no release / acquire fence, whatever
// memory type the exporting and the importing process map the window with.  (The first version used plain 16-byte
// accesses between system-scope release / acquire fences: its four fences cost 7 of its 16 us.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int IPC_SYS = 17;                                 // aux bits: sc0 (1) | sc1 (16) = system scope

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ipc_rsrc(unsigned char* win, long long bytes) {
  const unsigned long long b = (unsigned long long)win;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}

// wait until flag word `line + src` of the own window has reached `epoch`, for EVERY rank src (the own one too: the other
// workgroups of this rank publish behind it); threads 0 .. nranks-1 poll one flag each.  Returns with the workgroup
// synchronised.  No acquire: everything read behind the flags is read with system-scope loads.
__device__ __forceinline__ void ipc_wait_all(const IpcArgs& a, int line, unsigned long long epoch) {
  unsigned char* own = a.win[a.rank];
  if ((int)threadIdx.x < a.nranks) {
    unsigned long long* f = ipc_word(own, line + (int)threadIdx.x);
    unsigned spins = 0;
    const unsigned long long lim_ = __hip_atomic_load(ipc_word(own, 23), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned limit = lim_ ? (unsigned)lim_ : IPC_SPIN_LIMIT;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > limit) {
        __hip_atomic_store(ipc_word(own, 20), 1ull + (unsigned long long)threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
    if (spins > 4096)            // the longest wait so far, in polls (diagnostics: slu_comm_ipc_max_wait)
      __hip_atomic_fetch_max(ipc_word(own, 22), (unsigned long long)spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}

// every wave drains its (written-through) stores; the workgroup's lane 0 arrives on `arrive_line`; the LAST workgroup of the
// launch — every workgroup's stores have been acknowledged by then — raises flag `flag_line + rank` (= epoch) on every
// rank's window, the own one included.
__device__ __forceinline__ void ipc_publish(const IpcArgs& a, int arrive_line, int flag_line, unsigned long long epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long* cnt = ipc_word(a.win[a.rank], arrive_line);
    const unsigned long long old = __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned long long)gridDim.x - 1) {
      __hip_atomic_store(cnt, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int q = 0; q < a.nranks; ++q)
        __hip_atomic_store(ipc_word(a.win[q], flag_line + a.rank), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void __launch_bounds__(256)
allreduce_ipc_kernel(const IpcArgs a) {
  unsigned char* own = a.win[a.rank];
  const unsigned long long epoch =
      __hip_atomic_load(ipc_word(own, 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nthr = (long long)gridDim.x * 256;
  const long long u32 = (a.n32 + 3) / 4;                   // 16-byte units of the fp32 segment (tail padded with zeros)
  const long long u64 = (a.n64 + 1) / 2;                   // 16-byte units of the float64 segment
  const long long ua = u32 + u64;                          // the float64 units follow the fp32 units in a staging area
  const long long wbytes = IPC_FLAG_BYTES + 2 * a.cap;     // < 2^31 (checked by the launcher)
  const int IN = (int)IPC_FLAG_BYTES, OUT = (int)(IPC_FLAG_BYTES + a.cap);
  __amdgpu_buffer_rsrc_t rs[IPC_MAX_RANKS];
#pragma unroll
  for (int q = 0; q < IPC_MAX_RANKS; ++q) rs[q] = ipc_rsrc(a.win[q < a.nranks ? q : a.rank], wbytes);
  const __amdgpu_buffer_rsrc_t rown = ipc_rsrc(own, wbytes);

  // ---- 1. bucket -> own `in` ----
  for (long long u = tid; u < ua; u += nthr) {
    u32x4 o;
    if (u < u32) {
      if (4 * u + 3 < a.n32) {
        const float4 v = reinterpret_cast<const float4*>(a.f32)[u];
        o = u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
      } else {
        o = u32x4{4 * u + 0 < a.n32 ? __float_as_uint(a.f32[4 * u + 0]) : 0u, 4 * u + 1 < a.n32 ? __float_as_uint(a.f32[4 * u + 1]) : 0u,
                  4 * u + 2 < a.n32 ? __float_as_uint(a.f32[4 * u + 2]) : 0u, 0u};
      }
    } else {
      const long long e = 2 * (u - u32);
      const unsigned long long d0 = (unsigned long long)__double_as_longlong(a.f64[e]);
      const unsigned long long d1 = e + 1 < a.n64 ? (unsigned long long)__double_as_longlong(a.f64[e + 1]) : 0ull;
      o = u32x4{(unsigned)d0, (unsigned)(d0 >> 32), (unsigned)d1, (unsigned)(d1 >> 32)};
    }
    __builtin_amdgcn_raw_buffer_store_b128(o, rown, IN + (int)(16 * u), 0, IPC_SYS);
  }
  ipc_publish(a, 17, 0, epoch);
  ipc_wait_all(a, 0, epoch);

  // ---- 2. reduce this rank's chunk over all ranks (rank order), push the sum into every rank's `out` ----
  {
    const long long c0 = u32 * a.rank / a.nranks, c1 = u32 * (a.rank + 1) / a.nranks;
    for (long long u = c0 + tid; u < c1; u += nthr) {
      u32x4 v[IPC_MAX_RANKS];
#pragma unroll
      for (int q = 0; q < IPC_MAX_RANKS; ++q)
        if (q < a.nranks) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs[q], IN + (int)(16 * u), 0, IPC_SYS);
      float s0 = __uint_as_float(v[0][0]), s1 = __uint_as_float(v[0][1]), s2 = __uint_as_float(v[0][2]), s3 = __uint_as_float(v[0][3]);
#pragma unroll
      for (int q = 1; q < IPC_MAX_RANKS; ++q)
        if (q < a.nranks) {
          s0 += __uint_as_float(v[q][0]); s1 += __uint_as_float(v[q][1]);
          s2 += __uint_as_float(v[q][2]); s3 += __uint_as_float(v[q][3]);
        }
      const u32x4 o = {__float_as_uint(s0), __float_as_uint(s1), __float_as_uint(s2), __float_as_uint(s3)};
#pragma unroll
      for (int q = 0; q < IPC_MAX_RANKS; ++q)
        if (q < a.nranks) __builtin_amdgcn_raw_buffer_store_b128(o, rs[q], OUT + (int)(16 * u), 0, IPC_SYS);
    }
    // the float64 segment (160 values when the Sinc layer trains) is rank 0's
    if (a.rank == 0) {
      for (long long u = u32 + tid; u < ua; u += nthr) {
        double t0 = 0.0, t1 = 0.0;
        for (int q = 0; q < a.nranks; ++q) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs[q], IN + (int)(16 * u), 0, IPC_SYS);
          const double d0 = __longlong_as_double((long long)(((unsigned long long)v[1] << 32) | v[0]));
          const double d1 = __longlong_as_double((long long)(((unsigned long long)v[3] << 32) | v[2]));
          if (q == 0) { t0 = d0; t1 = d1; } else { t0 += d0; t1 += d1; }
        }
        const unsigned long long b0 = (unsigned long long)__double_as_longlong(t0), b1 = (unsigned long long)__double_as_longlong(t1);
        const u32x4 o = {(unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32)};
        for (int q = 0; q < a.nranks; ++q) __builtin_amdgcn_raw_buffer_store_b128(o, rs[q], OUT + (int)(16 * u), 0, IPC_SYS);
      }
    }
  }
  ipc_publish(a, 18, 8, epoch);
  ipc_wait_all(a, 8, epoch);

  // ---- 3. own `out` -> bucket ----
  for (long long u = tid; u < ua; u += nthr) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rown, OUT + (int)(16 * u), 0, IPC_SYS);
    if (u < u32) {
      if (4 * u + 3 < a.n32) {
        reinterpret_cast<float4*>(a.f32)[u] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
      } else {
        if (4 * u + 0 < a.n32) a.f32[4 * u + 0] = __uint_as_float(v[0]);
        if (4 * u + 1 < a.n32) a.f32[4 * u + 1] = __uint_as_float(v[1]);
        if (4 * u + 2 < a.n32) a.f32[4 * u + 2] = __uint_as_float(v[2]);
      }
    } else {
      const long long e = 2 * (u - u32);
      a.f64[e] = __longlong_as_double((long long)(((unsigned long long)v[1] << 32) | v[0]));
      if (e + 1 < a.n64) a.f64[e + 1] = __longlong_as_double((long long)(((unsigned long long)v[3] << 32) | v[2]));
    }
  }
  // the launch's last workgroup advances the epoch (every workgroup has read it by now)
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long* cnt = ipc_word(own, 19);
    const unsigned long long old = __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned long long)gridDim.x - 1) {
      __hip_atomic_store(cnt, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ipc_word(own, 16), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Touch every page of every window once (one system-scope load per 4 KiB, the sum parked in the own window's scratch
// line): peer windows are mapped lazily (hipIpcMemLazyEnablePeerAccess — the only mode hipIpcOpenMemHandle offers), so
// the FIRST access to a page of a peer's window can stall its wave for a long time while the mapping is established.
// Inside the all-reduce such a stall makes a rank miss its peers' bounded waits (seen: the first 1.21 MB call after a few
// tiny ones timed out now and then on ranks sharing a GPU).  This kernel has no flags and no waits: it just pays the
// first-touch cost up front, once per communicator.
__global__ void __launch_bounds__(256)
ipc_touch_kernel(const IpcArgs a, long long window_bytes) {
  unsigned acc = 0;
  const long long pages = (window_bytes + 4095) / 4096;
  for (int q = 0; q < a.nranks; ++q) {
    const __amdgpu_buffer_rsrc_t rs = ipc_rsrc(a.win[q], window_bytes);
    for (long long pg = (long long)blockIdx.x * 256 + threadIdx.x; pg < pages; pg += (long long)gridDim.x * 256)
      acc += __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(pg * 4096), 0, IPC_SYS);
  }
  if (acc == 0x9e3779b9u)        // never true for a zeroed / flag-only window: keeps the loads alive
    __hip_atomic_store(ipc_word(a.win[a.rank], 21), (unsigned long long)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace slu

using namespace slu;

extern "C" int64_t slu_comm_ipc_window_bytes(int64_t payload_bytes) {
  if (payload_bytes < 0) return 0;
  const long long cap = (payload_bytes + 32 + 255) / 256 * 256;      // + one padded 16-byte unit per typed segment
  return IPC_FLAG_BYTES + 2 * cap;
}

extern "C" int slu_comm_ipc_window_create(int64_t window_bytes, int64_t fine_grained, void** window_out, void* handle64) {
  SLU_REQUIRE(window_out && handle64 && window_bytes > IPC_FLAG_BYTES && (window_bytes - IPC_FLAG_BYTES) % 512 == 0,
              "slu_comm_ipc_window_create: bad argument (size from slu_comm_ipc_window_bytes)");
  void* p = nullptr;
  hipError_t e = fine_grained ? hipExtMallocWithFlags(&p, (size_t)window_bytes, hipDeviceMallocFinegrained)
                              : hipMalloc(&p, (size_t)window_bytes);
  if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "slu_comm_ipc_window_create: allocation of %lld bytes: %s", (long long)window_bytes, hipGetErrorString(e));
  e = hipMemset(p, 0, (size_t)window_bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    SLU_FAIL(SLU_ERR_HIP, "slu_comm_ipc_window_create: %s", hipGetErrorString(e));
  }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  memcpy(handle64, &h, 64);
  *window_out = p;
  return SLU_OK;
}

extern "C" int slu_comm_ipc_window_open(const void* handle64, void** window_out) {
  SLU_REQUIRE(handle64 && window_out, "slu_comm_ipc_window_open: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
  *window_out = p;
  return SLU_OK;
}

extern "C" int slu_comm_ipc_window_close(void* peer_window) {
  if (!peer_window) return SLU_OK;
  const hipError_t e = hipIpcCloseMemHandle(peer_window);
  if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "hipIpcCloseMemHandle: %s", hipGetErrorString(e));
  return SLU_OK;
}

extern "C" int slu_comm_ipc_window_destroy(void* own_window) {
  if (!own_window) return SLU_OK;
  const hipError_t e = hipFree(own_window);
  if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "hipFree: %s", hipGetErrorString(e));
  return SLU_OK;
}

extern "C" int slu_comm_allreduce_ipc(void* const* windows, int64_t rank, int64_t nranks, int64_t window_bytes,
                                      float* f32, int64_t n32, double* f64, int64_t n64, void* stream) {
  SLU_REQUIRE(windows && nranks >= 1 && nranks <= IPC_MAX_RANKS && rank >= 0 && rank < nranks,
              "slu_comm_allreduce_ipc: 1..%d ranks", IPC_MAX_RANKS);
  SLU_REQUIRE(n32 >= 0 && n64 >= 0 && n32 + n64 > 0 && (n32 == 0 || f32) && (n64 == 0 || f64),
              "slu_comm_allreduce_ipc: empty payload or null bucket");
  SLU_REQUIRE(((uintptr_t)f32 & 15) == 0 && ((uintptr_t)f64 & 15) == 0, "slu_comm_allreduce_ipc: buckets must be 16-byte aligned");
  IpcArgs a;
  a.cap = (window_bytes - IPC_FLAG_BYTES) / 2;
  SLU_REQUIRE(window_bytes < (1LL << 31), "slu_comm_allreduce_ipc: windows above 2 GiB are not addressable by one buffer descriptor");
  SLU_REQUIRE(window_bytes > IPC_FLAG_BYTES && 16 * ((n32 + 3) / 4) + 16 * ((n64 + 1) / 2) <= a.cap,
              "slu_comm_allreduce_ipc: payload of %lld + %lld elements exceeds the window's staging capacity (%lld bytes)",
              (long long)n32, (long long)n64, (long long)a.cap);
  for (int q = 0; q < IPC_MAX_RANKS; ++q) {
    a.win[q] = q < nranks ? (unsigned char*)windows[q] : nullptr;
    SLU_REQUIRE(q >= nranks || a.win[q], "slu_comm_allreduce_ipc: window of rank %d is null", q);
  }
  a.rank = (int)rank; a.nranks = (int)nranks;
  a.f32 = f32; a.n32 = n32; a.f64 = f64; a.n64 = n64;
  hipLaunchKernelGGL(allreduce_ipc_kernel, dim3(IPC_WGS), dim3(256), 0, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("allreduce_ipc_kernel");
  return SLU_OK;
}

// The longest wait of any call so far, in polls (~1 us each; 0 = none above 4096).  Synchronises the device.
extern "C" int slu_comm_ipc_max_wait(void* own_window, int64_t* polls_out) {
  SLU_REQUIRE(own_window && polls_out, "slu_comm_ipc_max_wait: null pointer");
  unsigned long long v = 0;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(&v, (unsigned char*)own_window + 64 * 22, 8, hipMemcpyDeviceToHost);
  if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "slu_comm_ipc_max_wait: %s", hipGetErrorString(e));
  *polls_out = (int64_t)v;
  return SLU_OK;
}

extern "C" int slu_comm_ipc_window_touch(void* const* windows, int64_t rank, int64_t nranks, int64_t window_bytes, void* stream) {
  SLU_REQUIRE(windows && nranks >= 1 && nranks <= IPC_MAX_RANKS && rank >= 0 && rank < nranks,
              "slu_comm_ipc_window_touch: 1..%d ranks", IPC_MAX_RANKS);
  SLU_REQUIRE(window_bytes > IPC_FLAG_BYTES && window_bytes < (1LL << 31), "slu_comm_ipc_window_touch: bad window size");
  IpcArgs a;
  a.cap = (window_bytes - IPC_FLAG_BYTES) / 2;
  for (int q = 0; q < IPC_MAX_RANKS; ++q) {
    a.win[q] = q < nranks ? (unsigned char*)windows[q] : nullptr;
    SLU_REQUIRE(q >= nranks || a.win[q], "slu_comm_ipc_window_touch: window of rank %d is null", q);
  }
  a.rank = (int)rank; a.nranks = (int)nranks;
  a.f32 = nullptr; a.n32 = 0; a.f64 = nullptr; a.n64 = 0;
  hipLaunchKernelGGL(ipc_touch_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, a, (long long)window_bytes);
  SLU_CHECK_LAUNCH("ipc_touch_kernel");
  return SLU_OK;
}

// Bound of every later wait of this rank's launches, in polls of ~1 - 2.5 us (0 = the start-up default, 2^26).  Synchronises
// the device (no launch may be reading the word while it changes).
extern "C" int slu_comm_ipc_set_spin_limit(void* own_window, int64_t polls) {
  SLU_REQUIRE(own_window && polls >= 0 && polls <= 0xffffffffLL, "slu_comm_ipc_set_spin_limit: bad argument");
  const unsigned long long v = (unsigned long long)polls;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy((unsigned char*)own_window + 64 * 23, &v, 8, hipMemcpyHostToDevice);
  if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "slu_comm_ipc_set_spin_limit: %s", hipGetErrorString(e));
  return SLU_OK;
}

// How many workgroups of the all-reduce kernel can be RESIDENT at once on `cus` compute units (occupancy of the kernel x
// cus) against the IPC_WGS = 64 it launches: its workgroups spin on flags that the LAST workgroup of the same launch raises,
// so all of them must be co-resident — the communicator refuses a partition on which they would not be.
extern "C" int slu_comm_ipc_resident_workgroups(int64_t cus, int64_t* resident_out, int64_t* launched_out) {
  SLU_REQUIRE(cus > 0 && resident_out && launched_out, "slu_comm_ipc_resident_workgroups: bad argument");
  int per_cu = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, allreduce_ipc_kernel, 256, 0);
  if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "hipOccupancyMaxActiveBlocksPerMultiprocessor: %s", hipGetErrorString(e));
  *resident_out = (int64_t)per_cu * cus;
  *launched_out = IPC_WGS;
  return SLU_OK;
}

// status word of the own window: 0 = every wait of every call so far was answered; 1 + q = a wait for rank q timed out
// (the results of that call are garbage, the ranks are out of step).  Synchronises the device.
extern "C" int slu_comm_ipc_status(void* own_window, int64_t* status_out) {
  SLU_REQUIRE(own_window && status_out, "slu_comm_ipc_status: null pointer");
  unsigned long long v = 0;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(&v, (unsigned char*)own_window + 64 * 20, 8, hipMemcpyDeviceToHost);
  if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "slu_comm_ipc_status: %s", hipGetErrorString(e));
  *status_out = (int64_t)v;
  return SLU_OK;
}
