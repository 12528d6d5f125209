// In-kernel barriers between workgroups of one launch (gfx950: 8 XCDs, per-XCD L2s that are not coherent with each
// other).  Shared by norm_ops.hip (one-launch BN / GN backward) and the convolution epilogue with a fused
// normalisation (conv_common.h: conv_epilogue_norm).
//
// What crosses workgroups is exchanged ONLY through device-scope atomics (partial sums, counters) and device-scope
// atomic loads afterwards: those are performed at the memory side, beyond the per-XCD L2s, so no L2 write-back /
// invalidate (what an agent-scope release / acquire fence costs on a multi-XCD part, for every workgroup) is needed.
// Partial sums are published with det_add_performed (kd6d_det.h; a RETURNING integer atomic: its result can only come
// back from where the add was performed, so the add is visible before the workgroup arrives; with returnless atomics one
// step in ~20 came out wrong); the workgroup-scope release and the __syncthreads order the arrival behind them.
//
// Every spin is bounded (~0.3 s): a barrier that cannot complete gives up, counts itself in the caller's timeout
// counter (kd6d_barrier_timeouts()) and lets the kernel drain -- wrong numbers instead of a hung GPU.
//
// RESIDENCY (why the waits complete).  A workgroup that waits holds its CU resources until the workgroups it waits for
// have ARRIVED; those may not have been dispatched yet.  Two facts about the hardware dispatcher are used: workgroups of
// a launch are dispatched in blockIdx order, and a workgroup that is not waiting on anything finishes in bounded time.
//   * window barrier (group_barrier / key barriers): a workgroup waits only for workgroups whose blockIdx lies within a
//     fixed window W of its own.  With the prefix [0, n) dispatched, every workgroup whose window lies inside the prefix
//     sees all its arrivals (arrival precedes waiting and is never blocked) and completes; only the < W workgroups at
//     the end of the prefix can be stuck.  Such a launch therefore pins at most W workgroups however the device is
//     shared.
//   * grid barrier: all `grid` workgroups must become resident together; the launch pins up to `grid` workgroups.
//   Progress needs a CU that can host the next workgroup of some launch.  Other streams' ordinary kernels release
//   their resources in bounded time, so only pinned workgroups count: the host (conv_fused_norm_ok / bn onepass
//   sizing) admits a barrier launch only if its pinnable set, PLUS the windows of the barrier launches that may run
//   beside it on other streams, fit in half of the device's LDS and wave slots -- then fewer than all CUs can be filled
//   by pinned workgroups of <= 80 KB / 8 waves each, and one with room for the next workgroup always exists.  A step
//   runs at most one grid-barrier launch at a time (they are all on the student's main stream); window launches may
//   run on the teacher's stream beside it.
#pragma once
#include <hip/hip_runtime.h>

namespace kd6d_detail {

constexpr unsigned kSpinLimit = 1u << 21;
constexpr unsigned kBarrierFan = 16;      // sub-counters of a grid barrier (KD6D_BARRIER_WORDS = 32 words per barrier)

__device__ __forceinline__ float load_device_scope(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one thread: arrive at `ctr`, wait until `need` have
__device__ __forceinline__ void arrive_and_wait(unsigned int* ctr, unsigned need, unsigned int* timeouts) {
  __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned it = 0;
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
    __builtin_amdgcn_s_sleep(2);
    if (++it > kSpinLimit) { atomicAdd(timeouts, 1u); break; }
  }
}

// Arrive at `ctr` and wait until `need` workgroups have.
__device__ __forceinline__ void group_barrier(unsigned int* ctr, unsigned need, unsigned int* timeouts) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) arrive_and_wait(ctr, need, timeouts);
  __syncthreads();
}

// All workgroups of the launch.  512 arrivals on one word would retire one after the other (~27 ns each, 14 us):
// they are spread over kBarrierFan sub-counters (ctr[1..]) whose last arrivers report to ctr[0], the word everyone
// polls.  ctr: KD6D_BARRIER_WORDS pre-zeroed words.
__device__ __forceinline__ void grid_barrier(unsigned int* ctr, unsigned bid, unsigned nblocks, unsigned int* timeouts) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned sub = bid % kBarrierFan;
    const unsigned in_sub = (nblocks - sub + kBarrierFan - 1) / kBarrierFan;
    const unsigned old = __hip_atomic_fetch_add(ctr + 1 + sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == in_sub) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned need = nblocks < kBarrierFan ? nblocks : kBarrierFan;
    unsigned it = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(2);
      if (++it > kSpinLimit) { atomicAdd(timeouts, 1u); break; }
    }
  }
  __syncthreads();
}

// device address of the library's timeout counter (norm_ops.hip); host side
unsigned int* barrier_timeouts_device_ptr();

}  // namespace kd6d_detail
