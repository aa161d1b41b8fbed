// Library-level entry points: version, error string, device check.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "../../include/vg_kernels.h"

static thread_local char g_err[512] = "";

extern "C" void vg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vg_last_error(void) { return g_err; }

extern "C" int vg_version(void) { return 100; }  // 0.1.0

extern "C" int vg_init(int device) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    vg_set_error("vg_init: hipGetDeviceProperties(%d) failed: %s", device, hipGetErrorString(e));
    return VG_ERR_ARG;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    vg_set_error("vg_init: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return VG_ERR_UNSUPPORTED;
  }
  return prop.multiProcessorCount;
}
