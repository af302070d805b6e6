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
