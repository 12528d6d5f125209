// Shared device/host helpers for the kd6d HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/kd6d.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// ---- error plumbing -------------------------------------------------------
void kd6d_set_error(const char* fmt, ...);

#define KD6D_CHECK_ARG(cond, ...)              \
  do {                                         \
    if (!(cond)) {                             \
      kd6d_set_error(__VA_ARGS__);             \
      return KD6D_ERR_ARG;                     \
    }                                          \
  } while (0)

#define KD6D_CHECK_LAUNCH(name)                                          \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      kd6d_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return KD6D_ERR_LAUNCH;                                            \
    }                                                                    \
  } while (0)

// ---- kernel-selection options (kd6d_set_option, include/kd6d.h) -------------------------------------------------
// One table instead of environment variables: the parity tests and the per-layer benches select a kernel family
// through the C ABI, inside one process; the product path never sets any of them.
enum Kd6dOption {
  KD6D_OPT_CONV_HALO = 0,      // -1 auto | 0 off | 1 256x128, 2 128x128 (4 waves), 3 128x128, 4 128x64, 5 128x32, 6 192x128, 9 64x64,
                               // 11-15 the two-per-CU twins: 128x128 on 4 / on 8 waves, 128x64, 64x64, 128x32
  KD6D_OPT_CONV_SMALLC,        // -1 auto | 0 off | 1 also below 2^17 pixels
  KD6D_OPT_CONV_SPLITK,        // -1 auto | 0 off | tile*100 + splits (tile 1 = 128x64, 2 = 64x64)
  KD6D_OPT_CONV_TILE,          // -1 auto | 0 register-staged kernel | 1 128x128, 2 128x64, 3 64x64 (LDS-DMA kernel)
  KD6D_OPT_WGRAD_SMALL,        // -1 auto | 0 off | 1 any size
  KD6D_OPT_BN_ONEPASS,         // 1 | 0: the two-launch BN backward even when the caller passes a barrier counter
  KD6D_OPT_BN_ONEPASS_MAX,     // largest x (16-B granules) on the one-launch BN backward
  KD6D_OPT_GN_ONEPASS,         // 1 | 0: the two-launch GN backward
  KD6D_OPT_SINKHORN_LANES,     // 1 | 0: every set on the general (one softmin after the other) path
  KD6D_OPT_CONV_HALO_PAIRING,  // 1 | 0: maps <= 32 wide keep the double-buffered (one workgroup per CU) halo tiles
  KD6D_OPT_CONV_FUSE_NORM,     // bit 0: GroupNorm, bit 1: BatchNorm geometries may take the fused launch (0: kd6d_conv2d_fwd_norm_fusable reports 0)
  KD6D_OPT_SINKHORN_DENSE_MFMA,  // dense OT, D = 16: 1 the matrix-pipe softmins where their cancellation error allows | 0 never | 2 always
                                 // | 3 as 1, but the gradient-carrying softmins of the last extrapolation keep the difference form
  KD6D_OPT_CONV_HALO_WIDE,     // 1 | 0: maps 65 ... 80 wide (480 x 640 full frames) stay off the halo-patch kernel
  KD6D_OPT_CONV_SMALLC_WMAX,   // widest map the resident-patch kernel takes (640; 256 = the limit of rounds 1-2)
  KD6D_OPT_SINKHORN_DENSE_SCREEN,  // dense OT, D = 16, passes below the matrix-pipe rule: 1 screened on the matrix pipe, exact pairs in the
                                   // difference form | 0 every pair in the difference form
  KD6D_OPT_SINKHORN_DENSE_ROWS,    // rows per workgroup of the dense matrix-pipe softmins: -1 = 128 from 8192 rows up, else 64 | 64 | 128
  KD6D_OPT_COUNT
};
long long kd6d_opt(int id);          // of the calling thread's current context

// ---- context (kd6d_ctx, include/kd6d.h): what used to be process-global state ------------------------------------
// Options, the pair bracket of conv_halo.hip and the barrier-timeout counter live in a context; every entry point of
// the library acts on the calling thread's CURRENT context (kd6d_ctx_make_current; none made current: the
// process-wide default context, which is what the Python host uses).
struct kd6d_ctx {
  long long opt[KD6D_OPT_COUNT];
  void* pair;                    // conv_halo.hip's PairState, created on first use
  void (*pair_free)(void*);
  unsigned int* timeouts;        // device word counting barrier waits that gave up
  bool owns_timeouts;
};
kd6d_ctx* kd6d_current_ctx();
unsigned int* kd6d_ctx_timeouts_ptr();      // device address of the current context's counter

// ---- scalar conversions ---------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 16-byte granule: 8 bf16 or 4 f32.
template <typename T> struct Granule;
template <> struct Granule<bf16_t> { static constexpr int N = 8; };
template <> struct Granule<float> { static constexpr int N = 4; };

template <typename T>
__device__ __forceinline__ void granule_to_f32(const u32x4_t& g, float* out);
template <>
__device__ __forceinline__ void granule_to_f32<float>(const u32x4_t& g, float* out) {
  out[0] = __uint_as_float(g.x); out[1] = __uint_as_float(g.y);
  out[2] = __uint_as_float(g.z); out[3] = __uint_as_float(g.w);
}
template <>
__device__ __forceinline__ void granule_to_f32<bf16_t>(const u32x4_t& g, float* out) {
  out[0] = __uint_as_float(g.x << 16); out[1] = __uint_as_float(g.x & 0xffff0000u);
  out[2] = __uint_as_float(g.y << 16); out[3] = __uint_as_float(g.y & 0xffff0000u);
  out[4] = __uint_as_float(g.z << 16); out[5] = __uint_as_float(g.z & 0xffff0000u);
  out[6] = __uint_as_float(g.w << 16); out[7] = __uint_as_float(g.w & 0xffff0000u);
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  bf16_t a = (bf16_t)lo, b = (bf16_t)hi;
  unsigned short ua = __builtin_bit_cast(unsigned short, a);
  unsigned short ub = __builtin_bit_cast(unsigned short, b);
  return (unsigned)ua | ((unsigned)ub << 16);
}

template <typename T>
__device__ __forceinline__ u32x4_t f32_to_granule(const float* in);
template <>
__device__ __forceinline__ u32x4_t f32_to_granule<float>(const float* in) {
  u32x4_t g;
  g.x = __float_as_uint(in[0]); g.y = __float_as_uint(in[1]);
  g.z = __float_as_uint(in[2]); g.w = __float_as_uint(in[3]);
  return g;
}
template <>
__device__ __forceinline__ u32x4_t f32_to_granule<bf16_t>(const float* in) {
  u32x4_t g;
  g.x = pack_bf16x2(in[0], in[1]); g.y = pack_bf16x2(in[2], in[3]);
  g.z = pack_bf16x2(in[4], in[5]); g.w = pack_bf16x2(in[6], in[7]);
  return g;
}

// ---- wave / block reductions (wave = 64 lanes) -----------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
