// Order-independent accumulation of fp32 addends (what makes two executions of the step BITWISE equal).
//
// A floating-point atomicAdd rounds after every addition, so the total depends on the order in which workgroups (and,
// inside a workgroup, waves) retire their atomics; on bf16 stores behind it a 1e-7 difference becomes an occasional
// 2^-9 flip and then a different max-pool / LeakyReLU / threshold decision (round 3: 15-30 of 150 gradient tensors moved
// by 2-21 % from run to run).  Integer addition is associative.  Every cross-wave / cross-workgroup reduction of the
// library therefore adds FIXED-POINT images of its fp32 partial sums with 64-bit integer atomics and converts back once:
//
//   accumulator  = two int64 words {lo, hi}, 16 bytes, zero-initialised
//   unit of lo   = 2^-E                     (E is a property of the reduction: KD6D_DET_ACT / KD6D_DET_GRAD below)
//   unit of hi   = 2^(47-E)
//   addend v     -> q = round-to-nearest(v * 2^E)            if it fits 47 bits: one atomic on lo
//                -> {q mod 2^47, q div 2^47} (sign applied)  otherwise: one atomic on each word
//   value        = hi * 2^(47-E) + lo * 2^-E                 (double, rounded to fp32 once)
//
// The image of an addend is a pure function of the addend, so the total is independent of the order of the atomics.
// |lo| addends are < 2^47: 2^15 of them per address cannot overflow (the launches of this library have <= 2^13
// workgroups per address).  A non-finite or absurdly large addend (>= 2^(109-E)) poisons the accumulator: hi += 2^47,
// and det_value returns NaN for |hi| >= 2^46 -- a NaN in a reduction stays visible, as with floating-point atomics.
// Resolution: KD6D_DET_ACT 2^-32 = 2.3e-10 absolute per addend (activation sums, |partial| ~ 1e-2 ... 1e4: finer than the
// fp32 rounding of the partial itself), KD6D_DET_GRAD 2^-52 = 2.2e-16 (gradient sums).
//
// det_split / det_value compile for the host as well (tests/test_det_accumulator.py builds them with g++).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KD6D_HD __host__ __device__ __forceinline__
#else
#define KD6D_HD static inline
#endif

#define KD6D_DET_ACT 32      // lo unit 2^-32: forward statistics (sums of activations and their squares), loss values
#define KD6D_DET_GRAD 52     // lo unit 2^-52: everything summed in the reverse sweep

namespace kd6d_detail {

struct det_words { long long lo, hi; };

// fixed-point image of v (fp32 bit pattern in, integer arithmetic only: no fp64 temporaries in the kernels' epilogues)
template <int E>
KD6D_HD det_words det_split(float v) {
  union { float f; unsigned u; } cv;
  cv.f = v;
  const unsigned bits = cv.u;
  const int e = (int)((bits >> 23) & 0xffu);
  const bool neg = (bits >> 31) != 0;
  det_words w;
  w.lo = 0; w.hi = 0;
  if (e == 0) return w;                                 // zero / fp32 subnormal (< 1.2e-38): nothing at either resolution
  const unsigned long long m = (unsigned long long)((bits & 0x7fffffu) | 0x800000u);   // v = m * 2^(e - 150)
  const int s = e - 150 + E;                            // q = m * 2^s in units of 2^-E
  if (e == 255 || s > 85) {                             // Inf / NaN / >= 2^(109 - E): poison
    w.hi = 1ll << 47;
    return w;
  }
  unsigned long long qlo, qhi = 0;
  if (s <= 0) {
    const int r = -s;
    qlo = r > 24 ? 0ull : (r == 0 ? m : ((m + (1ull << (r - 1))) >> r));    // round half up in magnitude
  } else if (s <= 23) {
    qlo = m << s;                                        // < 2^47
  } else if (s < 47) {
    qhi = m >> (47 - s);
    qlo = (m & ((1ull << (47 - s)) - 1ull)) << s;
  } else {
    qhi = m << (s - 47);                                 // s <= 85: < 2^62
    qlo = 0;
  }
  w.lo = neg ? -(long long)qlo : (long long)qlo;
  w.hi = neg ? -(long long)qhi : (long long)qhi;
  return w;
}

// Value of an accumulator as fp32: a pure function of the two words (every consumer of a statistic uses THIS function,
// so they all see the same bits), branch-free -- the normalisation kernels read 16-64 of them per thread and a
// data-dependent branch per read kept the loads from being issued together (round 4: +3 us on every 5-us kernel).
// fp32 arithmetic: each word is rounded to 24 bits (the precision of the result), power-of-two scalings are exact.
template <int E>
KD6D_HD float det_value(long long lo, long long hi) {
  const float v = __builtin_ldexpf((float)hi, 47 - E) + __builtin_ldexpf((float)lo, -E);
  union { unsigned u; float f; } nanv;
  nanv.u = 0x7fc00000u;
  return (hi >= (1ll << 46) || hi <= -(1ll << 46)) ? nanv.f : v;
}

#if defined(__HIPCC__)
// ---- device side -----------------------------------------------------------------------------------------------
// acc: two consecutive int64 words {lo, hi}.  Global memory, result not awaited.
template <int E>
__device__ __forceinline__ void det_add(long long* acc, float v) {
  const det_words w = det_split<E>(v);
  if (w.lo) atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)w.lo);
  if (w.hi) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + 1, (unsigned long long)w.hi);
}
__device__ __forceinline__ void det_add_words(long long* acc, long long lo, long long hi) {
  if (lo) atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)lo);
  if (hi) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + 1, (unsigned long long)hi);
}
// Device-scope adds whose RESULT the thread waits for (kd6d_barrier.h: a returning atomic is performed at the memory
// side before it comes back, so it is visible to every workgroup that sees this one arrive afterwards).
__device__ __forceinline__ void det_add_words_performed(long long* acc, long long lo, long long hi) {
  long long r0 = 0, r1 = 0;
  if (lo) r0 = __hip_atomic_fetch_add(acc, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (hi) r1 = __hip_atomic_fetch_add(acc + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::"v"(r0), "v"(r1));
}
template <int E>
__device__ __forceinline__ void det_add_performed(long long* acc, float v) {
  const det_words w = det_split<E>(v);
  det_add_words_performed(acc, w.lo, w.hi);
}
// LDS accumulator (two int64 words), workgroup scope
template <int E>
__device__ __forceinline__ void det_add_lds(long long* acc, float v) {
  const det_words w = det_split<E>(v);
  if (w.lo) atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)w.lo);
  if (w.hi) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + 1, (unsigned long long)w.hi);
}
// PLANAR layout (the gradient bucket's accumulators: word lo of element i at lo_ptr[i], word hi at lo_ptr[i + hi_off]):
// consecutive elements' lo words are contiguous, so the 64 atomics of a wave instruction cover 512 contiguous bytes
// (tools/microbench/atomic_rate.hip: interleaved {lo, hi} tiles flush 2-4x slower)
__device__ __forceinline__ void det_add_words_planar(long long* lo_ptr, long long hi_off, long long lo, long long hi) {
  if (lo) atomicAdd(reinterpret_cast<unsigned long long*>(lo_ptr), (unsigned long long)lo);
  if (hi) atomicAdd(reinterpret_cast<unsigned long long*>(lo_ptr + hi_off), (unsigned long long)hi);
}
template <int E>
__device__ __forceinline__ void det_add_planar(long long* lo_ptr, long long hi_off, float v) {
  const det_words w = det_split<E>(v);
  det_add_words_planar(lo_ptr, hi_off, w.lo, w.hi);
}
template <int E>
__device__ __forceinline__ float det_read(const long long* acc) {
  typedef long long ll2_t __attribute__((ext_vector_type(2)));
  const ll2_t w = *reinterpret_cast<const ll2_t*>(acc);        // one 16-byte load (accumulators are 16-byte aligned)
  return det_value<E>(w[0], w[1]);
}
// after an in-kernel barrier: device-scope loads (the per-XCD L2s are not coherent with each other)
__device__ __forceinline__ det_words det_load_device_scope(const long long* acc) {
  det_words w;
  w.lo = __hip_atomic_load(acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  w.hi = __hip_atomic_load(acc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return w;
}
// A launch that reduces to ONE fp32 scalar (a loss value, the squared gradient norm).  ws (kd6d_scalar_ws, 32 bytes,
// pre-zeroed): {lo, hi, arrivals}.  ONE thread per workgroup calls this with the workgroup's partial sum: the partial
// is added to the accumulator with returning atomics, the workgroup arrives, and the LAST arrival converts the total
// and writes *out (the running total if the workspace is reused without zeroing).  The arrival counter is left at 0.
template <int E>
__device__ __forceinline__ void det_scalar_arrive(long long* ws, float partial, unsigned nblocks, float* out) {
  det_add_performed<E>(ws, partial);
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  unsigned int* arrivals = reinterpret_cast<unsigned int*>(ws + 2);
  const unsigned old = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old + 1 == nblocks) {
    const det_words w = det_load_device_scope(ws);
    *out = det_value<E>(w.lo, w.hi);
    __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
#endif

}  // namespace kd6d_detail
