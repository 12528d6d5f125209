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

// Device timestamp marker: one lane stores the constant-rate (100 MHz) wall clock at the moment the launch runs.
// Lets a caller read phase boundaries of a replayed hipGraph whose streams really run concurrently (under
// rocprofv3 a graph replays one kernel at a time).
namespace {
__global__ void mark_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
}

extern "C" int kd6d_mark(unsigned long long* slot, void* stream) {
  KD6D_CHECK_ARG(slot != nullptr, "kd6d_mark: null slot");
  hipLaunchKernelGGL(mark_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), slot);
  KD6D_CHECK_LAUNCH("kd6d_mark");
  return KD6D_OK;
}
