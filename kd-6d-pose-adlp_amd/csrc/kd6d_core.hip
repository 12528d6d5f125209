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

// Kernel-selection options: see Kd6dOption in kd6d_common.h.  Plain loads / stores of aligned 64-bit words; the
// caller sets them between launches from the thread that launches.
namespace {
struct OptRow { const char* name; long long def; };
const OptRow kOptRows[KD6D_OPT_COUNT] = {
    {"conv.halo", -1},  {"conv.smallc", -1},       {"conv.splitk", -1}, {"conv.tile", -1},     {"wgrad.small", -1},
    {"bn.onepass", 1},  {"bn.onepass_max", 65536}, {"gn.onepass", 1},   {"sinkhorn.lanes", 1},
};
long long g_opt[KD6D_OPT_COUNT] = {-1, -1, -1, -1, -1, 1, 65536, 1, 1};
int opt_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < KD6D_OPT_COUNT; ++i)
    if (strcmp(name, kOptRows[i].name) == 0) return i;
  return -1;
}
}  // namespace

long long kd6d_opt(int id) { return g_opt[id]; }

extern "C" int kd6d_set_option(const char* name, long long value) {
  const int i = opt_index(name);
  KD6D_CHECK_ARG(i >= 0, "kd6d_set_option: unknown option '%s'", name ? name : "(null)");
  g_opt[i] = value;
  return KD6D_OK;
}

extern "C" int kd6d_get_option(const char* name, long long* value) {
  const int i = opt_index(name);
  KD6D_CHECK_ARG(i >= 0 && value, "kd6d_get_option: unknown option '%s'", name ? name : "(null)");
  *value = g_opt[i];
  return KD6D_OK;
}

extern "C" int kd6d_reset_options(void) {
  for (int i = 0; i < KD6D_OPT_COUNT; ++i) g_opt[i] = kOptRows[i].def;
  return KD6D_OK;
}

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
