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
