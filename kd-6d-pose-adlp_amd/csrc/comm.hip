// Data-parallel exchange of the KD step over RCCL (xGMI), behind the C ABI.
//
// Replaces the reference's process-group plumbing for the step: libs/distributed.py:9-41 (rank / world /
// barrier helpers over gloo), train_kd.py:48-51 (init_process_group + barrier) and the DDP constructor's
// parameter broadcast of libs/train_libs.py:123-130 -- plus the per-step gradient all-reduce the reference
// lacks because it discards the DDP wrapper.
//
// One communicator per process (= per GPU).  The data path has exactly one collective per step, a mean
// all-reduce of the flat fp32 gradient bucket (9.2 / 33.9 MB): latency-bound on 7 x 153 GB/s xGMI links, so one
// bucket, in place, enqueued on the caller's stream right behind the last weight gradient.  librccl is resolved
// at run time (dlopen) so that a single-GPU host needs no RCCL at all and a process that already carries a copy
// (PyTorch ships one with the same SONAME) shares that instance instead of loading a second one.
#include <dlfcn.h>

#include "kd6d_common.h"

namespace {

// the slice of rccl.h this file needs (rccl/rccl.h: ncclUniqueId :43, ncclRedOp_t :447-453, ncclDataType_t :456-472)
struct UniqueId { char internal[128]; };
typedef void* Comm;
constexpr int kSum = 0, kAvg = 4;
constexpr int kUint8 = 1, kFloat32 = 7;

struct Api {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
};
Api g_api;

bool load_api() {
  if (g_api.handle) return true;
  void* h = nullptr;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  // a copy that is already mapped (same SONAME) is shared: two RCCL instances in one process would each claim
  // the device's IPC / proxy resources
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    kd6d_set_error("kd6d_comm: librccl not found (%s)", dlerror());
    return false;
  }
  Api a;
  a.handle = h;
#define KD6D_SYM(field, name)                                                        \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                     \
  if (!a.field) {                                                                    \
    kd6d_set_error("kd6d_comm: librccl lacks %s", name);                             \
    return false;                                                                    \
  }
  KD6D_SYM(GetUniqueId, "ncclGetUniqueId")
  KD6D_SYM(CommInitRank, "ncclCommInitRank")
  KD6D_SYM(CommDestroy, "ncclCommDestroy")
  KD6D_SYM(AllReduce, "ncclAllReduce")
  KD6D_SYM(Broadcast, "ncclBroadcast")
  KD6D_SYM(GetErrorString, "ncclGetErrorString")
  KD6D_SYM(GetVersion, "ncclGetVersion")
#undef KD6D_SYM
  g_api = a;
  return true;
}

#define KD6D_RCCL(call, what)                                                             \
  do {                                                                                    \
    const int r__ = (call);                                                               \
    if (r__ != 0) {                                                                       \
      kd6d_set_error("%s: RCCL error %d (%s)", what, r__, g_api.GetErrorString(r__));     \
      return KD6D_ERR_LAUNCH;                                                             \
    }                                                                                     \
  } while (0)

}  // namespace

struct kd6d_comm {
  Comm comm;
  int rank, world, device;
};

extern "C" int kd6d_comm_unique_id(void* id_out_host) {
  KD6D_CHECK_ARG(id_out_host != nullptr, "kd6d_comm_unique_id: null output");
  if (!load_api()) return KD6D_ERR_UNSUPPORTED;
  UniqueId id;
  KD6D_RCCL(g_api.GetUniqueId(&id), "kd6d_comm_unique_id");
  memcpy(id_out_host, id.internal, sizeof(id.internal));
  return KD6D_OK;
}

extern "C" int kd6d_comm_init(kd6d_comm** out, int rank, int world, const void* unique_id_host) {
  KD6D_CHECK_ARG(out && unique_id_host && world >= 1 && rank >= 0 && rank < world, "kd6d_comm_init: bad arguments");
  if (!load_api()) return KD6D_ERR_UNSUPPORTED;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    kd6d_set_error("kd6d_comm_init: no current HIP device");
    return KD6D_ERR_LAUNCH;
  }
  UniqueId id;
  memcpy(id.internal, unique_id_host, sizeof(id.internal));
  Comm c = nullptr;
  KD6D_RCCL(g_api.CommInitRank(&c, world, id, rank), "kd6d_comm_init");
  kd6d_comm* k = new kd6d_comm{c, rank, world, dev};
  *out = k;
  return KD6D_OK;
}

extern "C" int kd6d_comm_rank(const kd6d_comm* c) { return c ? c->rank : -1; }
extern "C" int kd6d_comm_world(const kd6d_comm* c) { return c ? c->world : -1; }

extern "C" int kd6d_comm_version(void) {
  if (!load_api()) return -1;
  int v = 0;
  return g_api.GetVersion(&v) == 0 ? v : -1;
}

extern "C" int kd6d_comm_allreduce(kd6d_comm* c, float* buf, int64_t n, int mean, void* stream) {
  KD6D_CHECK_ARG(c && buf && n > 0, "kd6d_comm_allreduce: bad arguments");
  KD6D_RCCL(g_api.AllReduce(buf, buf, (size_t)n, kFloat32, mean ? kAvg : kSum, c->comm,
                            reinterpret_cast<hipStream_t>(stream)), "kd6d_comm_allreduce");
  return KD6D_OK;
}

extern "C" int kd6d_comm_broadcast(kd6d_comm* c, void* buf, int64_t nbytes, int root, void* stream) {
  KD6D_CHECK_ARG(c && buf && nbytes > 0 && root >= 0 && root < c->world, "kd6d_comm_broadcast: bad arguments");
  KD6D_RCCL(g_api.Broadcast(buf, buf, (size_t)nbytes, kUint8, root, c->comm, reinterpret_cast<hipStream_t>(stream)),
            "kd6d_comm_broadcast");
  return KD6D_OK;
}

extern "C" int kd6d_comm_destroy(kd6d_comm* c) {
  if (!c) return KD6D_OK;
  int rc = KD6D_OK;
  if (c->comm && g_api.CommDestroy) {
    const int r = g_api.CommDestroy(c->comm);
    if (r != 0) {
      kd6d_set_error("kd6d_comm_destroy: RCCL error %d (%s)", r, g_api.GetErrorString(r));
      rc = KD6D_ERR_LAUNCH;
    }
  }
  delete c;
  return rc;
}
