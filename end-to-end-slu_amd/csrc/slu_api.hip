// Library-level entry points: version, error string, device check.
#include "slu_common.h"
#include <string.h>

namespace slu {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace slu

extern "C" int slu_version(void) { return SLU_ABI_VERSION; }

extern "C" const char* slu_last_error(void) { return slu::g_err; }

extern "C" const char* slu_device_arch(void) {
  static thread_local char arch[256];
  arch[0] = 0;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess) return arch;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return arch;
  strncpy(arch, prop.gcnArchName, sizeof(arch) - 1);
  arch[sizeof(arch) - 1] = 0;
  return arch;
}

extern "C" int slu_device_check(void) {
  const char* a = slu_device_arch();
  if (strncmp(a, "gfx950", 6) != 0)
    SLU_FAIL(SLU_ERR_DEVICE, "libslu_hip is built for gfx950 (MI355X) only; current device is '%s'", a);
  return SLU_OK;
}

// HIP stream restricted to compute units [first_cu, first_cu + n_cus) of the current device.  The KFD
// deals mask bit i to XCD (i mod 8), so a contiguous bit range is spread evenly over the 8 XCDs.
// Used by the look-ahead pipeline to give the latency-bound trainable part of the step a few CUs of
// its own while the frozen-prefix super-batches fill the rest.  The stream lives until process exit.
extern "C" int slu_stream_create_cu_range(int64_t first_cu, int64_t n_cus, void** stream_out) {
  SLU_REQUIRE(stream_out, "slu_stream_create_cu_range: null pointer");
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
    SLU_FAIL(SLU_ERR_DEVICE, "slu_stream_create_cu_range: no HIP device");
  const int64_t total = prop.multiProcessorCount;
  SLU_REQUIRE(first_cu >= 0 && n_cus > 0 && first_cu + n_cus <= total,
              "slu_stream_create_cu_range: range [%lld, %lld) outside the %lld CUs of the device",
              (long long)first_cu, (long long)(first_cu + n_cus), (long long)total);
  uint32_t mask[32] = {0};
  SLU_REQUIRE(total <= 32 * 32, "slu_stream_create_cu_range: device too large");
  for (int64_t i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)((total + 31) / 32), mask);
  if (e != hipSuccess) SLU_FAIL(SLU_ERR_HIP, "hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
  *stream_out = (void*)st;
  return SLU_OK;
}

// ---- fused staging of a captured step's inputs -----------------------------------------------------------
// A hipGraph-captured step reads its inputs from static buffers; refreshing them took one copy kernel per
// input plus a fill for the dropout-stream offset (3 launches of ~5 us on a 64-CU partition).  One launch:
// up to 4 strided 2-D byte copies (rows x row_bytes, 16-byte granules when aligned) and one int64 store.
namespace slu {
struct StageSeg { const unsigned char* src; unsigned char* dst; long long rows, row_bytes, src_stride; };
struct StageArgs { StageSeg seg[4]; int nseg; long long* set_ptr; long long set_value; };

__global__ void __launch_bounds__(256)
stage_inputs_kernel(const StageArgs a) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.set_ptr) *a.set_ptr = a.set_value;
  for (int k = 0; k < a.nseg; ++k) {
    const StageSeg sg = a.seg[k];
    const bool vec = ((sg.row_bytes | sg.src_stride | (long long)(uintptr_t)sg.src | (long long)(uintptr_t)sg.dst) & 15) == 0;
    if (vec) {
      const long long per_row = sg.row_bytes >> 4, total = sg.rows * per_row;
      for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / per_row, c = i - r * per_row;
        reinterpret_cast<uint4*>(sg.dst + r * sg.row_bytes)[c] = reinterpret_cast<const uint4*>(sg.src + r * sg.src_stride)[c];
      }
    } else {
      const long long total = sg.rows * sg.row_bytes;
      for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / sg.row_bytes, c = i - r * sg.row_bytes;
        sg.dst[r * sg.row_bytes + c] = sg.src[r * sg.src_stride + c];
      }
    }
  }
}
}  // namespace slu

// ---- a few 64-bit words from the host into device memory, by value in the kernel arguments (hipGraph-friendly:
// no staging buffer whose lifetime the caller would have to track) — the row-pointer table and the dropout-stream
// offset of a captured look-ahead super-batch ----
namespace slu {
struct StoreArgs { unsigned long long v[64]; unsigned long long* dst; int n; };
__global__ void store_u64_kernel(const StoreArgs a) {
  if ((int)threadIdx.x < a.n) a.dst[threadIdx.x] = a.v[threadIdx.x];
}
}  // namespace slu

extern "C" int slu_store_u64(uint64_t* dst, const uint64_t* values, int64_t count, void* stream) {
  SLU_REQUIRE(dst && values && count >= 1 && count <= 64, "slu_store_u64: 1..64 values");
  slu::StoreArgs a;
  for (int k = 0; k < (int)count; ++k) a.v[k] = values[k];
  a.dst = (unsigned long long*)dst; a.n = (int)count;
  hipLaunchKernelGGL(slu::store_u64_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("store_u64_kernel");
  return SLU_OK;
}

extern "C" int slu_stage_inputs(const void* const* src, void* const* dst, const int64_t* rows, const int64_t* row_bytes,
                                const int64_t* src_stride_bytes, int64_t count, int64_t* set_ptr, int64_t set_value,
                                void* stream) {
  SLU_REQUIRE(count >= 0 && count <= 4, "slu_stage_inputs: at most 4 segments");
  slu::StageArgs a;
  long long bytes = 0;
  for (int k = 0; k < (int)count; ++k) {
    SLU_REQUIRE(src[k] && dst[k] && rows[k] > 0 && row_bytes[k] > 0, "slu_stage_inputs: bad segment %d", k);
    a.seg[k].src = (const unsigned char*)src[k]; a.seg[k].dst = (unsigned char*)dst[k];
    a.seg[k].rows = rows[k]; a.seg[k].row_bytes = row_bytes[k]; a.seg[k].src_stride = src_stride_bytes[k];
    bytes += rows[k] * row_bytes[k];
  }
  a.nseg = (int)count; a.set_ptr = (long long*)set_ptr; a.set_value = set_value;
  long long blocks = (bytes / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(slu::stage_inputs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("stage_inputs_kernel");
  return SLU_OK;
}

// ---- several small device-to-device copies / in-place scalings in ONE launch (lists by value in the kernel
// arguments): packing the fresh gradients into the flat all-reduce bucket (slu_hip/dp.py) and applying an upstream
// scalar to the few gradients of a loss head ----
namespace slu {
constexpr int MULTI_MAX = 32;
struct MultiArgs { const unsigned char* src[MULTI_MAX]; unsigned char* dst[MULTI_MAX]; long long bytes[MULTI_MAX]; int n; };

__global__ void __launch_bounds__(256)
copy_multi_kernel(const MultiArgs a) {
  const int k = blockIdx.y;
  if (k >= a.n) return;
  const unsigned char* __restrict__ s = a.src[k];
  unsigned char* __restrict__ d = a.dst[k];
  const long long nb = a.bytes[k];
  const bool vec = (((long long)(uintptr_t)s | (long long)(uintptr_t)d | nb) & 15) == 0;
  if (vec) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (nb >> 4); i += (long long)gridDim.x * 256)
      reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
  } else {            // every tensor is a whole number of 4-byte words (fp32 / fp64 elements)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (nb >> 2); i += (long long)gridDim.x * 256)
      reinterpret_cast<unsigned*>(d)[i] = reinterpret_cast<const unsigned*>(s)[i];
  }
}

struct ScaleArgs { float* ptr[MULTI_MAX]; long long numel[MULTI_MAX]; int n; const float* g; };

__global__ void __launch_bounds__(256)
scale_multi_kernel(const ScaleArgs a) {
  const int k = blockIdx.y;
  if (k >= a.n) return;
  const float g = a.g[0];
  float* __restrict__ p = a.ptr[k];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.numel[k]; i += (long long)gridDim.x * 256) p[i] *= g;
}
}  // namespace slu

extern "C" int slu_multi_max(void) { return slu::MULTI_MAX; }

extern "C" int slu_copy_multi(const void* const* src, void* const* dst, const int64_t* nbytes, int64_t count, void* stream) {
  SLU_REQUIRE(src && dst && nbytes && count >= 1 && count <= slu::MULTI_MAX, "slu_copy_multi: 1..%d segments", slu::MULTI_MAX);
  slu::MultiArgs a;
  long long most = 0;
  for (int k = 0; k < (int)count; ++k) {
    SLU_REQUIRE(src[k] && dst[k] && nbytes[k] > 0 && nbytes[k] % 4 == 0 && (((uintptr_t)src[k] | (uintptr_t)dst[k]) & 3) == 0,
                "slu_copy_multi: segment %d must be a non-empty, 4-byte aligned run of whole words", k);
    a.src[k] = (const unsigned char*)src[k]; a.dst[k] = (unsigned char*)dst[k]; a.bytes[k] = nbytes[k];
    if (nbytes[k] > most) most = nbytes[k];
  }
  a.n = (int)count;
  long long bx = (most / 16 + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(slu::copy_multi_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("copy_multi_kernel");
  return SLU_OK;
}

extern "C" int slu_scale_multi(float* const* ptrs, const int64_t* numel, int64_t count, const float* scale_dev, void* stream) {
  SLU_REQUIRE(ptrs && numel && scale_dev && count >= 1 && count <= slu::MULTI_MAX, "slu_scale_multi: 1..%d tensors", slu::MULTI_MAX);
  slu::ScaleArgs a;
  long long most = 0;
  for (int k = 0; k < (int)count; ++k) {
    SLU_REQUIRE(ptrs[k] && numel[k] > 0, "slu_scale_multi: bad tensor %d", k);
    a.ptr[k] = ptrs[k]; a.numel[k] = numel[k];
    if (numel[k] > most) most = numel[k];
  }
  a.n = (int)count; a.g = scale_dev;
  long long bx = (most + 1023) / 1024;
  if (bx < 1) bx = 1;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(slu::scale_multi_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("scale_multi_kernel");
  return SLU_OK;
}

// ---- range words of up to MULTI_MAX fp32 tensors in ONE launch: words[k] = max(words[k], bit pattern of max |x| over
// tensor k) — integer maximum of the sign-stripped IEEE patterns, so NaN / infinity report above every finite value.  The
// host-side guard of the f16x2 split scheme checks a model's frozen weights with it once per weight version
// (slu_hip/guard.py): a tensor whose largest entry leaves fp16's comfortable range sends the frozen stages to bf16x3. ----
namespace slu {
struct AbsmaxArgs { const float* ptr[MULTI_MAX]; long long numel[MULTI_MAX]; int n; unsigned* words; };

__global__ void __launch_bounds__(256)
absmax_multi_kernel(const AbsmaxArgs a) {
  const int k = blockIdx.y;
  if (k >= a.n) return;
  const float* __restrict__ p = a.ptr[k];
  unsigned mx = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.numel[k]; i += (long long)gridDim.x * 256)
    mx = max(mx, __float_as_uint(p[i]) & 0x7fffffffu);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, d, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(a.words + k, mx);
}
}  // namespace slu

extern "C" int slu_absmax_multi(const float* const* ptrs, const int64_t* numel, int64_t count, uint32_t* words, void* stream) {
  SLU_REQUIRE(ptrs && numel && words && count >= 1 && count <= slu::MULTI_MAX, "slu_absmax_multi: 1..%d tensors", slu::MULTI_MAX);
  slu::AbsmaxArgs a;
  long long most = 0;
  for (int k = 0; k < (int)count; ++k) {
    SLU_REQUIRE(ptrs[k] && numel[k] > 0, "slu_absmax_multi: bad tensor %d", k);
    a.ptr[k] = ptrs[k]; a.numel[k] = numel[k];
    if (numel[k] > most) most = numel[k];
  }
  a.n = (int)count; a.words = words;
  long long bx = (most + 4095) / 4096;
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(slu::absmax_multi_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, a);
  SLU_CHECK_LAUNCH("absmax_multi_kernel");
  return SLU_OK;
}

// ---- PCM16 samples -> fp32 (sample * scale; scale = 1 / 32768 is the sox / soundfile convention the reference's loaders
// hand the model, data.py:273-293): for first blocks that run on the exact fp32 kernels — the split-precision first block
// reads the int16 samples itself (slu_wconv_fwd_bf16 in_pcm16) ----
namespace slu {
__global__ void __launch_bounds__(256)
pcm16_to_f32_kernel(const short* __restrict__ in, float* __restrict__ out, long long n, float scale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = (float)in[i] * scale;
}
}  // namespace slu

extern "C" int slu_pcm16_to_f32(const int16_t* in, float* out, int64_t n, float scale, void* stream) {
  SLU_REQUIRE(in && out && n > 0, "slu_pcm16_to_f32: null pointer / empty");
  long long bx = (n + 255) / 256;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(slu::pcm16_to_f32_kernel, dim3((unsigned)bx), dim3(256), 0, (hipStream_t)stream, in, out, (long long)n, scale);
  SLU_CHECK_LAUNCH("pcm16_to_f32_kernel");
  return SLU_OK;
}
