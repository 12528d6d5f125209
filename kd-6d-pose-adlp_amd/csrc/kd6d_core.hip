// Error plumbing + ABI version of libkd6d.so.
#include <stdarg.h>

#include "kd6d_common.h"

namespace {
thread_local char g_err[512] = "";
}

void kd6d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* kd6d_last_error(void) { return g_err; }
extern "C" int kd6d_abi_version(void) { return KD6D_ABI_VERSION; }

// Number of compute units of the current device (host query, no sync).
extern "C" int kd6d_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}
