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
    {"conv.halo_pairing", 1}, {"conv.fuse_norm", 3}, {"sinkhorn.dense_mfma", 1}, {"conv.halo_wide", 1},
    {"conv.smallc_wmax", 640}, {"sinkhorn.dense_screen", 1}, {"sinkhorn.dense_rows", -1},
};
int opt_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < KD6D_OPT_COUNT; ++i)
    if (strcmp(name, kOptRows[i].name) == 0) return i;
  return -1;
}

void ctx_defaults(kd6d_ctx* c) {
  for (int i = 0; i < KD6D_OPT_COUNT; ++i) c->opt[i] = kOptRows[i].def;
  c->pair = nullptr; c->pair_free = nullptr; c->timeouts = nullptr; c->owns_timeouts = false;
}

kd6d_ctx* default_ctx() {
  static kd6d_ctx ctx = []() { kd6d_ctx c; ctx_defaults(&c); return c; }();
  return &ctx;
}
thread_local kd6d_ctx* g_current = nullptr;
}  // namespace

namespace kd6d_detail { unsigned int* barrier_timeouts_device_ptr(); }      // norm_ops.hip: the default context's word

kd6d_ctx* kd6d_current_ctx() { return g_current ? g_current : default_ctx(); }

unsigned int* kd6d_ctx_timeouts_ptr() {
  kd6d_ctx* c = kd6d_current_ctx();
  if (!c->timeouts) c->timeouts = kd6d_detail::barrier_timeouts_device_ptr();     // default context: the library's symbol
  return c->timeouts;
}

long long kd6d_opt(int id) { return kd6d_current_ctx()->opt[id]; }

extern "C" int kd6d_ctx_create(kd6d_ctx** out) {
  KD6D_CHECK_ARG(out != nullptr, "kd6d_ctx_create: null output");
  kd6d_ctx* c = new kd6d_ctx;
  ctx_defaults(c);
  void* word = nullptr;
  if (hipMalloc(&word, sizeof(unsigned int)) != hipSuccess || hipMemset(word, 0, sizeof(unsigned int)) != hipSuccess) {
    if (word) (void)hipFree(word);
    delete c;
    kd6d_set_error("kd6d_ctx_create: could not allocate the barrier-timeout counter (no device?)");
    return KD6D_ERR_LAUNCH;
  }
  c->timeouts = reinterpret_cast<unsigned int*>(word);
  c->owns_timeouts = true;
  *out = c;
  return KD6D_OK;
}

extern "C" int kd6d_ctx_destroy(kd6d_ctx* c) {
  if (!c) return KD6D_OK;
  KD6D_CHECK_ARG(c != default_ctx(), "kd6d_ctx_destroy: the default context cannot be destroyed");
  if (g_current == c) g_current = nullptr;
  if (c->pair && c->pair_free) c->pair_free(c->pair);
  if (c->owns_timeouts && c->timeouts) (void)hipFree(c->timeouts);
  delete c;
  return KD6D_OK;
}

extern "C" int kd6d_ctx_make_current(kd6d_ctx* c) {
  g_current = (c == default_ctx()) ? nullptr : c;
  return KD6D_OK;
}

extern "C" kd6d_ctx* kd6d_ctx_current(void) { return kd6d_current_ctx(); }

extern "C" int kd6d_ctx_set_option(kd6d_ctx* c, const char* name, long long value) {
  const int i = opt_index(name);
  KD6D_CHECK_ARG(i >= 0, "kd6d_set_option: unknown option '%s'", name ? name : "(null)");
  (c ? c : kd6d_current_ctx())->opt[i] = value;
  return KD6D_OK;
}

extern "C" int kd6d_ctx_get_option(kd6d_ctx* c, const char* name, long long* value) {
  const int i = opt_index(name);
  KD6D_CHECK_ARG(i >= 0 && value, "kd6d_get_option: unknown option '%s'", name ? name : "(null)");
  *value = (c ? c : kd6d_current_ctx())->opt[i];
  return KD6D_OK;
}

extern "C" int kd6d_ctx_reset_options(kd6d_ctx* c) {
  c = c ? c : kd6d_current_ctx();
  for (int i = 0; i < KD6D_OPT_COUNT; ++i) c->opt[i] = kOptRows[i].def;
  return KD6D_OK;
}

extern "C" int kd6d_ctx_barrier_timeouts(kd6d_ctx* c) {
  kd6d_ctx* saved = g_current;
  if (c) g_current = (c == default_ctx()) ? nullptr : c;
  unsigned int* p = kd6d_ctx_timeouts_ptr();
  g_current = saved;
  unsigned int v = 0;
  if (!p || hipMemcpy(&v, p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)v;
}

// the context-free forms act on the calling thread's current context
extern "C" int kd6d_set_option(const char* name, long long value) { return kd6d_ctx_set_option(nullptr, name, value); }
extern "C" int kd6d_get_option(const char* name, long long* value) { return kd6d_ctx_get_option(nullptr, name, value); }
extern "C" int kd6d_reset_options(void) { return kd6d_ctx_reset_options(nullptr); }
extern "C" int kd6d_barrier_timeouts(void) { return kd6d_ctx_barrier_timeouts(nullptr); }

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

// Step prologue: every buffer a KD step accumulates into (the flat gradient bucket, the statistics arena, the dense
// head gradient, the loss-side slot arrays) zeroed by ONE launch, plus the BatchNorm step counters.  HBM-bound
// (~20 MB of 16-B stores per step at B = 16); it replaces five to six separate fills at the head of the step's
// dependency chain.
namespace {
struct ZeroArgs {
  unsigned long long ptr[KD6D_MAX_ZERO];
  long long granules_end[KD6D_MAX_ZERO];   // running end of each region in 16-B granules (tails included, rounded up)
  long long bytes[KD6D_MAX_ZERO];
  int n;
};

__global__ __launch_bounds__(256) void zero_regions_kernel(const ZeroArgs a, long long* counter, int n_counter) {
  const long long total = a.granules_end[a.n - 1];
  const long long stride = (long long)gridDim.x * 256;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    int r = 0;
    while (g >= a.granules_end[r]) ++r;
    const long long local = g - (r ? a.granules_end[r - 1] : 0);
    char* base = reinterpret_cast<char*>(a.ptr[r]);
    const long long off = local * 16;
    if (off + 16 <= a.bytes[r]) {
      *reinterpret_cast<u32x4_t*>(base + off) = u32x4_t{0u, 0u, 0u, 0u};
    } else {                                           // the region's last, partial granule: 4-B words
      for (long long b = off; b < a.bytes[r]; b += 4) *reinterpret_cast<unsigned*>(base + b) = 0u;
    }
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < n_counter) counter[threadIdx.x] += 1;
}
}  // namespace

extern "C" int kd6d_zero_regions(const kd6d_zero_list* list, long long* counter, int n_counter, void* stream) {
  KD6D_CHECK_ARG(list && list->n >= 0 && list->n <= KD6D_MAX_ZERO, "kd6d_zero_regions: bad list");
  KD6D_CHECK_ARG(n_counter >= 0 && n_counter <= 256 && (n_counter == 0 || counter), "kd6d_zero_regions: bad counters");
  ZeroArgs a;
  a.n = 0;
  long long end = 0;
  for (int i = 0; i < list->n; ++i) {
    if (list->bytes[i] == 0) continue;
    KD6D_CHECK_ARG(list->ptr[i] && list->bytes[i] > 0 && (list->bytes[i] & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(list->ptr[i]) & 15) == 0,
                   "kd6d_zero_regions: region %d must be 16-B aligned with a size that is a multiple of 4", i);
    end += (list->bytes[i] + 15) / 16;
    a.ptr[a.n] = reinterpret_cast<unsigned long long>(list->ptr[i]);
    a.bytes[a.n] = list->bytes[i];
    a.granules_end[a.n] = end;
    ++a.n;
  }
  if (a.n == 0 && n_counter == 0) return KD6D_OK;
  if (a.n == 0) { a.n = 1; a.ptr[0] = 0; a.bytes[0] = 0; a.granules_end[0] = 0; }
  long long blocks = (end + 256 * 8 - 1) / (256 * 8);      // 8 granules (128 B) per thread
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(zero_regions_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     a, counter, n_counter);
  KD6D_CHECK_LAUNCH("kd6d_zero_regions");
  return KD6D_OK;
}

// Uniform [0, 1) keys for the SSC sampling of losses/loss.py:224-228 (the reference draws torch.randperm per ground
// truth and level; a uniform key per cell and "the n_k smallest keys inside the mask" is the same distribution).
// Counter-based: key(i) = mix(seed, *counter, i) with the splitmix64 finaliser, so a replayed hipGraph draws fresh keys
// every step from the step counter the prologue increments -- torch's generator inside a captured graph costs a key
// kernel plus two seed / offset fills per replay.
namespace {
__global__ __launch_bounds__(256) void uniform_keys_kernel(float* __restrict__ out, long long n,
                                                           const long long* __restrict__ counter,
                                                           unsigned long long seed) {
  const unsigned long long step = (unsigned long long)counter[0];
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    unsigned long long z = seed + step * 0x9E3779B97F4A7C15ull + (unsigned long long)i * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    out[i] = (float)(z >> 40) * (1.0f / 16777216.0f);      // 24 random bits: every value is exact in fp32, < 1
  }
}
}  // namespace

extern "C" int kd6d_uniform_keys(float* out, int64_t n, const long long* counter, unsigned long long seed,
                                 void* stream) {
  KD6D_CHECK_ARG(out && counter && n > 0, "kd6d_uniform_keys: bad arguments");
  long long blocks = (n + 256 * 4 - 1) / (256 * 4);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(uniform_keys_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     out, (long long)n, counter, seed);
  KD6D_CHECK_LAUNCH("kd6d_uniform_keys");
  return KD6D_OK;
}
