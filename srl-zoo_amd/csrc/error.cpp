// error.cpp — thread-local error text + misc host entry points of the C ABI.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void srlz_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int srlz_hip_fail(hipError_t e, const char* what) {
  srlz_set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return SRLZ_ERR_HIP;
}

extern "C" int srlz_version(void) { return 100; }

extern "C" const char* srlz_last_error(void) { return g_err; }

extern "C" int srlz_device_cus(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
  return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
}
