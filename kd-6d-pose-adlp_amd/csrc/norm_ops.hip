// Normalisation / pooling / resampling kernels of the KD step (NHWC, HBM-bound).
//
// Reference ops replaced:
//   BatchNorm2d (train mode) + LeakyReLU(0.1) of ConvBlock, backbone/common.py:316-324
//   GroupNorm(32) + ReLU of the PoseHead towers, models/model.py:395-417,438-451
//   MaxPool2d(2,2), backbone/darknet.py:94-97
//   F.interpolate(nearest, x2) + add of the FPN top-down path, models/model.py:75-78
//   F.relu between P6 and P7, models/model.py:101
// Every kernel reads/writes 16-B granules.  Reductions: fp32 inside a thread (fixed order), then -- across the threads
// of a workgroup and across workgroups -- 64-bit integer atomics on fixed-point accumulators (kd6d_det.h), so every
// statistic and every gradient sum is BITWISE reproducible from run to run whatever order the atomics retire in.
#include "kd6d_barrier.h"
#include "kd6d_common.h"
#include "kd6d_det.h"

namespace {

constexpr int kThreads = 256;
using kd6d_detail::det_add_lds;
using kd6d_detail::det_add_words;
using kd6d_detail::det_add_words_performed;
using kd6d_detail::det_add_words_planar;
using kd6d_detail::det_load_device_scope;
using kd6d_detail::det_read;
using kd6d_detail::det_value;
using kd6d_detail::det_words;
typedef long long acc_t;            // one word of an accumulator {lo, hi} (kd6d_acc = two of them)

// ---------------------------------------------------------------------------
// Per-channel reductions over rows.  Thread t owns channel granule t % (C/EG);
// requires (C/EG) | 256.  F(acc, granule index, values...) accumulates NACC sums.
// ---------------------------------------------------------------------------
// v += v of the lanes o to the right (mod 16) inside each DPP row of 16 lanes: VALU only.
template <int O>
__device__ __forceinline__ float row_ror_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + O, 0xF, 0xF, true));
}

// out[a]: `replicas` rows of C accumulators; this workgroup adds into row blockIdx.x % replicas (the rows are summed
// by the consumer), so an address sees 1/replicas of the serially retired atomics.  E: KD6D_DET_ACT / KD6D_DET_GRAD.
// Inside the workgroup the threads that own the same channel meet through LDS SLOTS and a fixed-order fp32 sum -- no
// LDS atomics: at most 16 partial sums per channel (narrow tensors: the lanes of a 16-lane row that own the same channel
// granule are summed with DPP rotations first), one fixed-point conversion per channel and workgroup.
// Dynamic LDS: kFlushLdsBytes.  PLANAR: out[a] is the lo plane of a gradient-bucket accumulator array (hi words at
// + hi_off, no replica rows).
constexpr int kFlushLdsBytes = 16384;      // 256 threads x NACC * EG (<= 16) floats
template <int NACC, int EG, int E, bool PERFORMED = false, bool PLANAR = false>
__device__ __forceinline__ void block_channel_flush(float (&acc)[NACC][EG], int C, int cg,
                                                    acc_t* const* out, int replicas = 1, long long hi_off = 0) {
  extern __shared__ acc_t lds_acc[];
  float* slot = reinterpret_cast<float*>(lds_acc);      // [partial][NACC * C]
  const int cgs = C / EG;
  const bool pre = cgs < 16 && (16 % cgs) == 0;
  int part, nparts;
  if (pre) {
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int e = 0; e < EG; ++e) {
        float v = acc[a][e];
        v = row_ror_add<8>(v);
        if (cgs <= 4) v = row_ror_add<4>(v);
        if (cgs <= 2) v = row_ror_add<2>(v);
        if (cgs <= 1) v = row_ror_add<1>(v);
        acc[a][e] = v;
      }
    part = threadIdx.x >> 4;                // one partial per 16-lane row: lanes 0 .. cgs-1 of the row hold it
    nparts = kThreads / 16;
  } else {
    part = threadIdx.x / cgs;               // thread t owns granule t % cgs (threads beyond (256 / cgs) * cgs idle)
    nparts = kThreads / cgs;
  }
  const int roff = replicas > 1 ? (int)(blockIdx.x % replicas) * C : 0;
  if (nparts > 16) {
    // odd channel counts (C / EG neither a divisor nor a multiple of 16, e.g. C = 40): up to 85 partials per channel --
    // LDS fixed-point accumulators instead of slots
    for (int i = threadIdx.x; i < NACC * C * 2; i += kThreads) lds_acc[i] = 0;
    __syncthreads();
    if (part < nparts) {
#pragma unroll 1
      for (int k = 0; k < NACC * EG; ++k) {
        float v = 0.f;
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
          for (int e = 0; e < EG; ++e) v = (k == a * EG + e) ? acc[a][e] : v;
        det_add_lds<E>(&lds_acc[((k / EG) * C + cg * EG + (k % EG)) * 2], v);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NACC * C; i += kThreads) {
      const int a = i / C, c = i - a * C;
      if (!out[a]) continue;
      if (PLANAR) det_add_words_planar(out[a] + c, hi_off, lds_acc[2 * i], lds_acc[2 * i + 1]);
      else if (PERFORMED) det_add_words_performed(out[a] + (size_t)(roff + c) * 2, lds_acc[2 * i], lds_acc[2 * i + 1]);
      else det_add_words(out[a] + (size_t)(roff + c) * 2, lds_acc[2 * i], lds_acc[2 * i + 1]);
    }
    return;
  }
  const bool writer = pre ? (int)(threadIdx.x & 15) < cgs : part < nparts;
  if (writer) {
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int e = 0; e < EG; ++e) slot[(part * NACC + a) * C + cg * EG + e] = acc[a][e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NACC * C; i += kThreads) {
    const int a = i / C, c = i - a * C;
    if (!out[a]) continue;
    float t = slot[a * C + c];
    for (int p = 1; p < nparts; ++p) t += slot[(p * NACC + a) * C + c];
    const det_words w = kd6d_detail::det_split<E>(t);
    if (PLANAR) {
      det_add_words_planar(out[a] + c, hi_off, w.lo, w.hi);
    } else {
      acc_t* o = out[a] + (size_t)(roff + c) * 2;
      if (PERFORMED) det_add_words_performed(o, w.lo, w.hi);
      else det_add_words(o, w.lo, w.hi);
    }
  }
}

// In-kernel barriers: kd6d_barrier.h (returning atomics, group_barrier, grid_barrier, the residency argument).  The
// kernels of this file count their give-ups here; the convolution epilogues add to the same word through its device
// address (barrier_timeouts_device_ptr).
__device__ unsigned int g_barrier_timeouts = 0;      // the default context's counter; other contexts own a word each
__device__ __forceinline__ void group_barrier(unsigned int* ctr, unsigned need, unsigned int* timeouts) {
  kd6d_detail::group_barrier(ctr, need, timeouts);
}
__device__ __forceinline__ void grid_barrier(unsigned int* ctr, unsigned nblocks, unsigned int* timeouts) {
  kd6d_detail::grid_barrier(ctr, blockIdx.x, nblocks, timeouts);
}

// Row-tiled variant of the ownership rule: thread t owns channel granule t % cgs of row t / cgs
// (threads beyond (256/cgs)*cgs idle), so any C with C/EG <= 256 works (e.g. C = 240 bias grads).
// GRAD: the column sums are a bias gradient -- class KD6D_DET_GRAD, planar accumulators (hi words at + hi_off)
template <typename T, bool GRAD = false>
__global__ __launch_bounds__(kThreads) void colstats_kernel(const T* __restrict__ x, long long rows,
                                                            int C, acc_t* sum, acc_t* sumsq, long long hi_off) {
  constexpr int EG = Granule<T>::N;
  const int cgs = C / EG;
  const int rpp = kThreads / cgs;            // rows per pass
  const int cg = threadIdx.x % cgs;
  const int rsub = threadIdx.x / cgs;
  float acc[2][EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  const u32x4_t* xg = reinterpret_cast<const u32x4_t*>(x);
  if (rsub < rpp) {
#pragma unroll 4
    for (long long r = (long long)blockIdx.x * rpp + rsub; r < rows; r += (long long)gridDim.x * rpp) {
      float v[EG];
      granule_to_f32<T>(xg[r * cgs + cg], v);
#pragma unroll
      for (int e = 0; e < EG; ++e) { acc[0][e] += v[e]; acc[1][e] += v[e] * v[e]; }
    }
  }
  acc_t* outs[2] = {sum, sumsq};
  if (GRAD) block_channel_flush<2, EG, KD6D_DET_GRAD, false, true>(acc, C, cg, outs, 1, hi_off);
  else block_channel_flush<2, EG, KD6D_DET_ACT>(acc, C, cg, outs);
}

// Granule g (EG consecutive channels) of a tensor stored as TX; TX = float with EG = 8 is the
// "fp32 pre-normalisation tensor, bf16 activations" case: the conv output that feeds BN / GN is
// kept in fp32 so that (x - mean) does not cancel bf16 rounding error.
template <typename TX, int EG>
__device__ __forceinline__ void load_x(const TX* __restrict__ base, long long g, float* v);
template <>
__device__ __forceinline__ void load_x<bf16_t, 8>(const bf16_t* __restrict__ base, long long g, float* v) {
  granule_to_f32<bf16_t>(reinterpret_cast<const u32x4_t*>(base)[g], v);
}
template <>
__device__ __forceinline__ void load_x<float, 4>(const float* __restrict__ base, long long g, float* v) {
  granule_to_f32<float>(reinterpret_cast<const u32x4_t*>(base)[g], v);
}
template <>
__device__ __forceinline__ void load_x<float, 8>(const float* __restrict__ base, long long g, float* v) {
  granule_to_f32<float>(reinterpret_cast<const u32x4_t*>(base)[2 * g], v);
  granule_to_f32<float>(reinterpret_cast<const u32x4_t*>(base)[2 * g + 1], v + 4);
}

// act(v * sc + sh) with an explicit fma: the pooled kernels below recompute it in the backward pass and have to
// reproduce the forward value bit for bit (the argmax of a 2x2 window must not depend on a contraction choice)
__device__ __forceinline__ float bn_act(float v, float sc, float sh, int act) {
  float t = __builtin_fmaf(v, sc, sh);
  if (act == KD6D_ACT_LEAKY) t = t > 0.f ? t : 0.1f * t;
  else if (act == KD6D_ACT_RELU) t = fmaxf(t, 0.f);
  return t;
}
__device__ __forceinline__ void bn_scale_shift(float mean, float invstd, float gamma, float beta, float& sc, float& sh) {
  sc = gamma * invstd;
  sh = __builtin_fmaf(-mean, sc, beta);
}

// Train-mode BatchNorm forward, per-channel scale / shift from the batch sums.  Reading an accumulator is two 64-bit
// integer -> float conversions (~25 instructions); a thread that converted the sums of its own 8 channels spent more on
// that than on its four granules (+1.5 us on 6-us launches).  One conversion per channel and WORKGROUP instead: thread c
// converts channel c and leaves {scale, shift} in LDS; block 0 also writes the saved / running statistics from there.
constexpr int kBnTabC = 1024;
template <int EG>
__device__ __forceinline__ void bn_fwd_scale_shift(
    float* s_tab, int C, int cg, float inv_rows, const acc_t* __restrict__ sum, const acc_t* __restrict__ sumsq,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, float unbias,
    float* running_mean, float* running_var, float* save_mean, float* save_invstd, float (&sc)[EG], float (&sh)[EG]) {
  const bool tab = C <= kBnTabC;
  for (int c = threadIdx.x; c < C; c += kThreads) {
    if (!tab && blockIdx.x != 0) break;
    const float m = det_read<KD6D_DET_ACT>(sum + 2 * c) * inv_rows;
    const float var = fmaxf(det_read<KD6D_DET_ACT>(sumsq + 2 * c) * inv_rows - m * m, 0.f);
    const float is = rsqrtf(var + eps);
    if (tab) bn_scale_shift(m, is, gamma[c], beta[c], s_tab[c], s_tab[kBnTabC + c]);
    if (blockIdx.x == 0) {
      if (save_mean) save_mean[c] = m;
      if (save_invstd) save_invstd[c] = is;
      if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
      if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * unbias;
    }
  }
  if (tab) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EG; ++e) { sc[e] = s_tab[cg * EG + e]; sh[e] = s_tab[kBnTabC + cg * EG + e]; }
  } else {
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      const int c = cg * EG + e;
      const float m = det_read<KD6D_DET_ACT>(sum + 2 * c) * inv_rows;
      const float var = fmaxf(det_read<KD6D_DET_ACT>(sumsq + 2 * c) * inv_rows - m * m, 0.f);
      bn_scale_shift(m, rsqrtf(var + eps), gamma[c], beta[c], sc[e], sh[e]);
    }
  }
}

// y = act(gamma * (x - mean) * invstd + beta), mean/var from the batch sums.
template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void bn_apply_fwd_kernel(
    const TX* __restrict__ x, T* __restrict__ y, long long ngran, int C, float inv_rows,
    const acc_t* __restrict__ sum, const acc_t* __restrict__ sumsq, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float unbias,
    float* running_mean, float* running_var, float* save_mean, float* save_invstd, int act) {
  constexpr int EG = Granule<T>::N;
  const int cgs = C / EG;
  const int cg = threadIdx.x % cgs;
  float sc[EG], sh[EG];
  __shared__ float s_tab[2 * kBnTabC];
  bn_fwd_scale_shift<EG>(s_tab, C, cg, inv_rows, sum, sumsq, gamma, beta, eps, momentum, unbias, running_mean,
                         running_var, save_mean, save_invstd, sc, sh);
  u32x4_t* yg = reinterpret_cast<u32x4_t*>(y);
#pragma unroll 4
  for (long long g = (long long)blockIdx.x * kThreads + threadIdx.x; g < ngran;
       g += (long long)gridDim.x * kThreads) {
    float v[EG];
    load_x<TX, EG>(x, g, v);
#pragma unroll
    for (int e = 0; e < EG; ++e) v[e] = bn_act(v[e], sc[e], sh[e], act);
    yg[g] = f32_to_granule<T>(v);
  }
}

// reduce pass of BN backward: sum(dy), sum(dy * xhat) per channel, dy = dz * act'(bn(x))
template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(
    const TX* __restrict__ x, const T* __restrict__ dz, long long ngran, int C,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int act, acc_t* sum_dy,
    acc_t* sum_dy_xhat, int replicas) {
  constexpr int EG = Granule<T>::N;
  const int cgs = C / EG;
  const int cg = threadIdx.x % cgs;
  float m[EG], is[EG], ga[EG], be[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    m[e] = mean[c]; is[e] = invstd[c]; ga[e] = gamma[c]; be[e] = beta[c];
  }
  float acc[2][EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  const u32x4_t* dg = reinterpret_cast<const u32x4_t*>(dz);
#pragma unroll 4
  for (long long g = (long long)blockIdx.x * kThreads + threadIdx.x; g < ngran;
       g += (long long)gridDim.x * kThreads) {
    float xv[EG], dv[EG];
    load_x<TX, EG>(x, g, xv);
    granule_to_f32<T>(dg[g], dv);
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      const float xh = (xv[e] - m[e]) * is[e];
      const float pre = xh * ga[e] + be[e];
      float d = dv[e];
      if (act == KD6D_ACT_LEAKY) d = pre > 0.f ? d : 0.1f * d;
      else if (act == KD6D_ACT_RELU) d = pre > 0.f ? d : 0.f;
      acc[0][e] += d;
      acc[1][e] += d * xh;
    }
  }
  acc_t* outs[2] = {sum_dy, sum_dy_xhat};
  block_channel_flush<2, EG, KD6D_DET_GRAD>(acc, C, cg, outs, replicas);
}

// total of `replicas` rows of C accumulators at channel c: the words are added as integers (exact), converted once
template <int E>
__device__ __forceinline__ float replica_sum(const acc_t* __restrict__ rows, int C, int replicas, int c) {
  acc_t lo = 0, hi = 0;
  for (int r = 0; r < replicas; ++r) {
    lo += rows[((size_t)r * C + c) * 2];
    hi += rows[((size_t)r * C + c) * 2 + 1];
  }
  return det_value<E>(lo, hi);
}

template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(
    const TX* __restrict__ x, const T* __restrict__ dz, T* __restrict__ dx, long long ngran, int C,
    float inv_rows, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int act,
    const acc_t* __restrict__ sum_dy, const acc_t* __restrict__ sum_dy_xhat, float* dgamma,
    float* dbeta, int replicas) {
  constexpr int EG = Granule<T>::N;
  extern __shared__ float tot[];             // 2 * C: replica rows of the two reductions, summed once per workgroup
  const int cgs = C / EG;
  const int cg = threadIdx.x % cgs;
  float m[EG], is[EG], ga[EG], be[EG], k1[EG], k2[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    m[e] = mean[c]; is[e] = invstd[c]; ga[e] = gamma[c]; be[e] = beta[c];
  }
  for (int i = threadIdx.x; i < 2 * C; i += kThreads)
    tot[i] = i < C ? replica_sum<KD6D_DET_GRAD>(sum_dy, C, replicas, i) : replica_sum<KD6D_DET_GRAD>(sum_dy_xhat, C, replicas, i - C);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    k1[e] = tot[c] * inv_rows;
    k2[e] = tot[C + c] * inv_rows;
  }
  if (blockIdx.x == 0) {                     // one writer per channel: plain adds, reproducible
    for (int c = threadIdx.x; c < C; c += kThreads) {
      if (dgamma) dgamma[c] += tot[C + c];
      if (dbeta) dbeta[c] += tot[c];
    }
  }
  const u32x4_t* dg = reinterpret_cast<const u32x4_t*>(dz);
  u32x4_t* og = reinterpret_cast<u32x4_t*>(dx);
#pragma unroll 4
  for (long long g = (long long)blockIdx.x * kThreads + threadIdx.x; g < ngran;
       g += (long long)gridDim.x * kThreads) {
    float xv[EG], dv[EG];
    load_x<TX, EG>(x, g, xv);
    granule_to_f32<T>(dg[g], dv);
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      const float xh = (xv[e] - m[e]) * is[e];
      const float pre = xh * ga[e] + be[e];
      float d = dv[e];
      if (act == KD6D_ACT_LEAKY) d = pre > 0.f ? d : 0.1f * d;
      else if (act == KD6D_ACT_RELU) d = pre > 0.f ? d : 0.f;
      dv[e] = ga[e] * is[e] * (d - k1[e] - xh * k2[e]);
    }
    og[g] = f32_to_granule<T>(dv);
  }
}

// ---------------------------------------------------------------------------
// BN (train) + activation + MaxPool2d(2,2) in one pass -- the last block of a darknet-tiny stage followed by
// the pool (backbone/darknet.py:94-97).  The activation tensor of such a layer is never written: forward stores
// only the pooled maxima; backward re-derives the four activations of each window from the fp32 conv output,
// finds the argmax again (first maximum in row-major window order, values rounded to T exactly as the separate
// bn -> maxpool pair sees them) and routes the pooled gradient there, so neither the activation nor its
// gradient ever touch HBM.  x: (B,H,W,C) TX; pooled tensors (B,H/2,W/2,C) T.  One work item = one pooled granule.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void round_as(float* v) {
  if constexpr (Granule<T>::N == 8) {       // bf16: through the storage format and back
    const u32x4_t q = f32_to_granule<T>(v);
    granule_to_f32<T>(q, v);
  }
}

struct PoolWin {
  int cg;
  long long g00;      // input granule index of the window's top-left pixel
  long long row;      // input granules per image row (W * cgs)
  int cgs;
};
__device__ __forceinline__ PoolWin pool_window(int item, int H, int W, int cgs) {
  const int Ho = H >> 1, Wo = W >> 1;
  PoolWin w;
  w.cg = item % cgs;
  int r = item / cgs;
  const int ox = r % Wo; r /= Wo;
  const int oy = r % Ho;
  const int b = r / Ho;
  w.cgs = cgs;
  w.row = (long long)W * cgs;
  w.g00 = ((long long)(b * H + oy * 2) * W + ox * 2) * cgs + w.cg;
  return w;
}
__device__ __forceinline__ long long pool_tap(const PoolWin& w, int k) {
  return w.g00 + (k >> 1) * w.row + (k & 1) * w.cgs;
}

template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void bn_pool_fwd_kernel(
    const TX* __restrict__ x, T* __restrict__ y, int items, int H, int W, int C, float inv_rows,
    const acc_t* __restrict__ sum, const acc_t* __restrict__ sumsq, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float unbias,
    float* running_mean, float* running_var, float* save_mean, float* save_invstd, int act) {
  constexpr int EG = Granule<T>::N;
  const int cgs = C / EG;
  const int cg = threadIdx.x % cgs;
  float sc[EG], sh[EG];
  __shared__ float s_tab[2 * kBnTabC];
  bn_fwd_scale_shift<EG>(s_tab, C, cg, inv_rows, sum, sumsq, gamma, beta, eps, momentum, unbias, running_mean,
                         running_var, save_mean, save_invstd, sc, sh);
  u32x4_t* yg = reinterpret_cast<u32x4_t*>(y);
  for (int item = blockIdx.x * kThreads + threadIdx.x; item < items; item += gridDim.x * kThreads) {
    const PoolWin w = pool_window(item, H, W, cgs);
    float v[4][EG];
#pragma unroll
    for (int k = 0; k < 4; ++k) load_x<TX, EG>(x, pool_tap(w, k), v[k]);
    float best[EG];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int e = 0; e < EG; ++e) v[k][e] = bn_act(v[k][e], sc[e], sh[e], act);
      round_as<T>(v[k]);
#pragma unroll
      for (int e = 0; e < EG; ++e) best[e] = (k == 0 || v[k][e] > best[e]) ? v[k][e] : best[e];
    }
    yg[item] = f32_to_granule<T>(best);
  }
}

// the window's activations again (forward formula, forward rounding) -> arg[e] = first maximum; xh[k][e] = xhat
template <typename T, typename TX, int EG>
__device__ __forceinline__ void pool_rederive(const TX* __restrict__ x, const PoolWin& w, const float* m,
                                              const float* is, const float* sc, const float* sh, int act,
                                              float (&xh)[4][EG], int (&arg)[EG]) {
  float best[EG];
#pragma unroll
  for (int k = 0; k < 4; ++k) load_x<TX, EG>(x, pool_tap(w, k), xh[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float a[EG];
#pragma unroll
    for (int e = 0; e < EG; ++e) a[e] = bn_act(xh[k][e], sc[e], sh[e], act);
    round_as<T>(a);
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      if (k == 0 || a[e] > best[e]) { best[e] = a[e]; arg[e] = k; }
      xh[k][e] = (xh[k][e] - m[e]) * is[e];
    }
  }
}

template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void bn_pool_bwd_reduce_kernel(
    const TX* __restrict__ x, const T* __restrict__ dy, int items, int H, int W, int C,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, int act, acc_t* sum_dy, acc_t* sum_dy_xhat, int replicas) {
  constexpr int EG = Granule<T>::N;
  const int cgs = C / EG;
  const int cg = threadIdx.x % cgs;
  float m[EG], is[EG], ga[EG], be[EG], sc[EG], sh[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    m[e] = mean[c]; is[e] = invstd[c]; ga[e] = gamma[c]; be[e] = beta[c];
    bn_scale_shift(m[e], is[e], ga[e], be[e], sc[e], sh[e]);
  }
  float acc[2][EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  const u32x4_t* dg = reinterpret_cast<const u32x4_t*>(dy);
  for (int item = blockIdx.x * kThreads + threadIdx.x; item < items; item += gridDim.x * kThreads) {
    const PoolWin w = pool_window(item, H, W, cgs);
    float xh[4][EG], dv[EG];
    int arg[EG];
    granule_to_f32<T>(dg[item], dv);
    pool_rederive<T, TX, EG>(x, w, m, is, sc, sh, act, xh, arg);
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      const float xa = arg[e] == 0 ? xh[0][e] : arg[e] == 1 ? xh[1][e] : arg[e] == 2 ? xh[2][e] : xh[3][e];
      const float pre = xa * ga[e] + be[e];
      float d = dv[e];
      if (act == KD6D_ACT_LEAKY) d = pre > 0.f ? d : 0.1f * d;
      else if (act == KD6D_ACT_RELU) d = pre > 0.f ? d : 0.f;
      acc[0][e] += d;
      acc[1][e] += d * xa;
    }
  }
  acc_t* outs[2] = {sum_dy, sum_dy_xhat};
  block_channel_flush<2, EG, KD6D_DET_GRAD>(acc, C, cg, outs, replicas);
}

template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void bn_pool_bwd_apply_kernel(
    const TX* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int items, int H, int W, int C,
    float inv_rows, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int act,
    const acc_t* __restrict__ sum_dy, const acc_t* __restrict__ sum_dy_xhat, float* dgamma, float* dbeta,
    int replicas) {
  constexpr int EG = Granule<T>::N;
  extern __shared__ float tot[];             // 2 * C
  const int cgs = C / EG;
  const int cg = threadIdx.x % cgs;
  float m[EG], is[EG], ga[EG], be[EG], sc[EG], sh[EG], k1[EG], k2[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    m[e] = mean[c]; is[e] = invstd[c]; ga[e] = gamma[c]; be[e] = beta[c];
    bn_scale_shift(m[e], is[e], ga[e], be[e], sc[e], sh[e]);
  }
  for (int i = threadIdx.x; i < 2 * C; i += kThreads)
    tot[i] = i < C ? replica_sum<KD6D_DET_GRAD>(sum_dy, C, replicas, i) : replica_sum<KD6D_DET_GRAD>(sum_dy_xhat, C, replicas, i - C);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    k1[e] = tot[c] * inv_rows;
    k2[e] = tot[C + c] * inv_rows;
  }
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += kThreads) {
      if (dgamma) dgamma[c] += tot[C + c];
      if (dbeta) dbeta[c] += tot[c];
    }
  }
  const u32x4_t* dg = reinterpret_cast<const u32x4_t*>(dy);
  u32x4_t* og = reinterpret_cast<u32x4_t*>(dx);
  for (int item = blockIdx.x * kThreads + threadIdx.x; item < items; item += gridDim.x * kThreads) {
    const PoolWin w = pool_window(item, H, W, cgs);
    float xh[4][EG], dv[EG];
    int arg[EG];
    granule_to_f32<T>(dg[item], dv);
    pool_rederive<T, TX, EG>(x, w, m, is, sc, sh, act, xh, arg);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o[EG];
#pragma unroll
      for (int e = 0; e < EG; ++e) {
        const float pre = xh[k][e] * ga[e] + be[e];
        float d = arg[e] == k ? dv[e] : 0.f;
        if (act == KD6D_ACT_LEAKY) d = pre > 0.f ? d : 0.1f * d;
        else if (act == KD6D_ACT_RELU) d = pre > 0.f ? d : 0.f;
        o[e] = ga[e] * is[e] * (d - k1[e] - xh[k][e] * k2[e]);
      }
      og[pool_tap(w, k)] = f32_to_granule<T>(o);
    }
  }
}

// ---------------------------------------------------------------------------
// BN backward in ONE launch: every workgroup keeps its slice (<= kBnHold granules per thread, raw x and dz) in
// registers, adds its partial sums into the replica rows, meets all other workgroups of the launch at an
// in-kernel barrier and finishes dx from the registers -- x and dz are read once and the second launch with its
// prologue goes.  The host only launches grids of at most 3/4 of the workgroups the device keeps resident
// (occupancy query; the kernels are held under 170 VGPRs -> 3 workgroups per CU, 768 on an MI355X, grid <= 512)
// and a step has one such kernel in flight at a time, so the barrier completes; group_barrier's spin limit is the
// safety net.  Larger tensors take the reduce + apply pair.
// ---------------------------------------------------------------------------
constexpr int kBnHold = 4;
constexpr int kBnOnepassBlocks = 512;

// totals of the replica rows -> tot[2C] (LDS floats), read with device-scope atomic loads (see group_barrier); the
// words of the rows are added as integers (exact) and converted once
__device__ __forceinline__ void replica_totals(const acc_t* sum_dy, const acc_t* sum_dy_xhat, int C, int replicas,
                                               float* tot) {
  for (int i = threadIdx.x; i < 2 * C; i += kThreads) {
    const acc_t* src = i < C ? sum_dy + 2 * i : sum_dy_xhat + 2 * (i - C);
    acc_t lo = 0, hi = 0;
    for (int r0 = 0; r0 < replicas; r0 += 8) {           // eight rows in flight: one memory round trip, not eight
      det_words v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        v[r].lo = 0; v[r].hi = 0;
        if (r0 + r < replicas) v[r] = det_load_device_scope(src + (size_t)(r0 + r) * C * 2);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) { lo += v[r].lo; hi += v[r].hi; }
    }
    tot[i] = det_value<KD6D_DET_GRAD>(lo, hi);
  }
  __syncthreads();
}

template <typename T, typename TX>
__global__ __launch_bounds__(kThreads, 3) void bn_bwd_onepass_kernel(
    const TX* __restrict__ x, const T* __restrict__ dz, T* __restrict__ dx, long long ngran, int per_thread, int C,
    float inv_rows, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int act, acc_t* sum_dy, acc_t* sum_dy_xhat,
    unsigned int* counter, float* dgamma, float* dbeta, int replicas, unsigned int* timeouts) {
  constexpr int EG = Granule<T>::N;
  extern __shared__ acc_t lds_acc[];         // 2 * C accumulators (block_channel_flush), then the totals as floats
  float* red = reinterpret_cast<float*>(lds_acc);
  const int cgs = C / EG;
  const int cg = threadIdx.x % cgs;
  float m[EG], is[EG], ga[EG], be[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    m[e] = mean[c]; is[e] = invstd[c]; ga[e] = gamma[c]; be[e] = beta[c];
  }
  float acc[2][EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  const u32x4_t* dg = reinterpret_cast<const u32x4_t*>(dz);
  u32x4_t* og = reinterpret_cast<u32x4_t*>(dx);
  const long long base = (long long)blockIdx.x * per_thread * kThreads + threadIdx.x;
  float hx[kBnHold][EG];
  u32x4_t hz[kBnHold];
#pragma unroll
  for (int i = 0; i < kBnHold; ++i) {
    const long long g = base + (long long)i * kThreads;
    const bool valid = i < per_thread && g < ngran;
    const long long gi = valid ? g : 0;
    load_x<TX, EG>(x, gi, hx[i]);
    hz[i] = dg[gi];
    if (!valid) hz[i] = u32x4_t{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int i = 0; i < kBnHold; ++i) {
    float dv[EG];
    granule_to_f32<T>(hz[i], dv);
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      const float xh = (hx[i][e] - m[e]) * is[e];
      const float pre = xh * ga[e] + be[e];
      float d = dv[e];
      if (act == KD6D_ACT_LEAKY) d = pre > 0.f ? d : 0.1f * d;
      else if (act == KD6D_ACT_RELU) d = pre > 0.f ? d : 0.f;
      acc[0][e] += d;
      acc[1][e] += d * xh;
    }
  }
  acc_t* outs[2] = {sum_dy, sum_dy_xhat};
  block_channel_flush<2, EG, KD6D_DET_GRAD, true>(acc, C, cg, outs, replicas);
  grid_barrier(counter, gridDim.x, timeouts);
  // only the RAW slice crosses the barrier: without this the compiler also keeps xhat and the masked gradient of
  // the first phase alive (twice the registers, half the resident workgroups)
#pragma unroll
  for (int i = 0; i < kBnHold; ++i) {
    asm volatile("" : "+v"(hz[i].x), "+v"(hz[i].y), "+v"(hz[i].z), "+v"(hz[i].w));
#pragma unroll
    for (int e = 0; e < EG; ++e) asm volatile("" : "+v"(hx[i][e]));
  }
  replica_totals(sum_dy, sum_dy_xhat, C, replicas, red);
  float k1[EG], k2[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    k1[e] = red[c] * inv_rows;
    k2[e] = red[C + c] * inv_rows;
  }
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += kThreads) {
      if (dgamma) dgamma[c] += red[C + c];
      if (dbeta) dbeta[c] += red[c];
    }
  }
#pragma unroll
  for (int i = 0; i < kBnHold; ++i) {
    const long long g = base + (long long)i * kThreads;
    if (i < per_thread && g < ngran) {
      float dv[EG];
      granule_to_f32<T>(hz[i], dv);
#pragma unroll
      for (int e = 0; e < EG; ++e) {
        const float xh = (hx[i][e] - m[e]) * is[e];
        const float pre = xh * ga[e] + be[e];
        float d = dv[e];
        if (act == KD6D_ACT_LEAKY) d = pre > 0.f ? d : 0.1f * d;
        else if (act == KD6D_ACT_RELU) d = pre > 0.f ? d : 0.f;
        dv[e] = ga[e] * is[e] * (d - k1[e] - xh * k2[e]);
      }
      og[g] = f32_to_granule<T>(dv);
    }
  }
}

// the pooled variant: <= kPoolHold windows per thread (xhat of the four taps, the pooled gradient, the argmax)
constexpr int kPoolHold = 2;
template <typename T, typename TX>
__global__ __launch_bounds__(kThreads, 3) void bn_pool_bwd_onepass_kernel(
    const TX* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int items, int per_thread, int H, int W,
    int C, float inv_rows, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int act, acc_t* sum_dy, acc_t* sum_dy_xhat,
    unsigned int* counter, float* dgamma, float* dbeta, int replicas, unsigned int* timeouts) {
  constexpr int EG = Granule<T>::N;
  extern __shared__ acc_t lds_acc[];         // 2 * C accumulators (block_channel_flush), then the totals as floats
  float* red = reinterpret_cast<float*>(lds_acc);
  const int cgs = C / EG;
  const int cg = threadIdx.x % cgs;
  float m[EG], is[EG], ga[EG], be[EG], sc[EG], sh[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    m[e] = mean[c]; is[e] = invstd[c]; ga[e] = gamma[c]; be[e] = beta[c];
    bn_scale_shift(m[e], is[e], ga[e], be[e], sc[e], sh[e]);
  }
  float acc[2][EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  const u32x4_t* dg = reinterpret_cast<const u32x4_t*>(dy);
  u32x4_t* og = reinterpret_cast<u32x4_t*>(dx);
  const int base = blockIdx.x * per_thread * kThreads + threadIdx.x;
  float hx[kPoolHold][4][EG];
  float hd[kPoolHold][EG];
  int harg[kPoolHold][EG];
#pragma unroll
  for (int i = 0; i < kPoolHold; ++i) {
    const int item = base + i * kThreads;
    const bool valid = i < per_thread && item < items;
    const PoolWin w = pool_window(valid ? item : 0, H, W, cgs);
    granule_to_f32<T>(dg[valid ? item : 0], hd[i]);
    pool_rederive<T, TX, EG>(x, w, m, is, sc, sh, act, hx[i], harg[i]);
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      const int a = harg[i][e];
      const float xa = a == 0 ? hx[i][0][e] : a == 1 ? hx[i][1][e] : a == 2 ? hx[i][2][e] : hx[i][3][e];
      const float pre = xa * ga[e] + be[e];
      float d = valid ? hd[i][e] : 0.f;
      if (act == KD6D_ACT_LEAKY) d = pre > 0.f ? d : 0.1f * d;
      else if (act == KD6D_ACT_RELU) d = pre > 0.f ? d : 0.f;
      hd[i][e] = d;                           // the activation's derivative is already applied
      acc[0][e] += d;
      acc[1][e] += d * xa;
    }
  }
  acc_t* outs[2] = {sum_dy, sum_dy_xhat};
  block_channel_flush<2, EG, KD6D_DET_GRAD, true>(acc, C, cg, outs, replicas);
  grid_barrier(counter, gridDim.x, timeouts);
  replica_totals(sum_dy, sum_dy_xhat, C, replicas, red);
  float k1[EG], k2[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    k1[e] = red[c] * inv_rows;
    k2[e] = red[C + c] * inv_rows;
  }
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += kThreads) {
      if (dgamma) dgamma[c] += red[C + c];
      if (dbeta) dbeta[c] += red[c];
    }
  }
#pragma unroll
  for (int i = 0; i < kPoolHold; ++i) {
    const int item = base + i * kThreads;
    if (i < per_thread && item < items) {
      const PoolWin w = pool_window(item, H, W, cgs);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float o[EG];
#pragma unroll
        for (int e = 0; e < EG; ++e) {
          const float d = harg[i][e] == k ? hd[i][e] : 0.f;
          o[e] = ga[e] * is[e] * (d - k1[e] - hx[i][k][e] * k2[e]);
        }
        og[pool_tap(w, k)] = f32_to_granule<T>(o);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// GroupNorm(G) + ReLU over multi-level NHWC tensors.  Statistics are kept as RAW sums
// stats[(seg*batch + b)*G + g] = {sum x, sum x^2}, two accumulators (kd6d_det.h; integer atomics from row-chunk
// workgroups, so a 32x32 level is reduced by 16 workgroups instead of one); consumers derive
// mean = s/n, rstd = rsqrt(max(q/n - mean^2, 0) + eps) with n = H*W*C/G.
// ---------------------------------------------------------------------------
// rows of one (level, sample) handled by a reduction workgroup (GnGeom::chunk_rows)
int gn_chunk_rows() { return 128; }

struct GnGeom {
  int nseg, batch, C, G;
  int row0[KD6D_MAX_SEG];
  int hw[KD6D_MAX_SEG];
  int nblk;                   // reduction workgroups in all
  int blk0[KD6D_MAX_SEG];     // first reduction workgroup of the level
  int cps[KD6D_MAX_SEG];      // reduction workgroups (row chunks) per sample
  int chunk_rows;
  unsigned int* timeouts;     // the launching context's counter of barrier waits that gave up
};

// reduction workgroup -> (seg, b, first row, row count)
__device__ __forceinline__ void gn_chunk(const GnGeom& gm, int blk, int& seg, int& b, int& r_begin, int& r_cnt) {
  seg = 0;
  int b0 = 0, cps = 1, hw = 0, row0 = 0;
#pragma unroll
  for (int s = 0; s < KD6D_MAX_SEG; ++s)
    if (s < gm.nseg && blk >= gm.blk0[s]) { seg = s; b0 = gm.blk0[s]; cps = gm.cps[s]; hw = gm.hw[s]; row0 = gm.row0[s]; }
  const int local = blk - b0;
  b = local / cps;
  const int chunk = local - b * cps;
  const int lo = chunk * gm.chunk_rows;
  r_cnt = min(gm.chunk_rows, hw - lo);
  r_begin = row0 + b * hw + lo;
}

// st: the {sum, sumsq} accumulator pair of one (level, image, group)
__device__ __forceinline__ void gn_mean_rstd(const acc_t* __restrict__ st, float inv_n, float eps, float& mu,
                                             float& rs) {
  mu = det_read<KD6D_DET_ACT>(st) * inv_n;
  rs = rsqrtf(fmaxf(det_read<KD6D_DET_ACT>(st + 2) * inv_n - mu * mu, 0.f) + eps);
}

template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void gn_stats_kernel(const TX* __restrict__ x, GnGeom gm,
                                                            acc_t* __restrict__ stats) {
  constexpr int EG = Granule<T>::N;
  __shared__ acc_t s_st[64 * 4];           // per group {sum, sumsq} accumulators
  int seg, b, r_begin, r_cnt;
  gn_chunk(gm, blockIdx.x, seg, b, r_begin, r_cnt);
  const int C = gm.C, G = gm.G, cpg = C / G;
  s_st[threadIdx.x] = 0;                   // kThreads == 64 * 4
  __syncthreads();
  const int cgs = C / EG;
  const long long ngran = (long long)r_cnt * cgs;
  const TX* xb = x + (size_t)r_begin * C;
  const int cg = threadIdx.x % cgs;  // (C/EG) | 256
  // a granule may straddle groups when cpg < EG (C=128,G=32,bf16: 2 groups per granule)
  float a1[EG], a2[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
#pragma unroll 4
  for (long long g = threadIdx.x; g < ngran; g += kThreads) {
    float v[EG];
    load_x<TX, EG>(xb, g, v);
#pragma unroll
    for (int e = 0; e < EG; ++e) { a1[e] += v[e]; a2[e] += v[e] * v[e]; }
  }
  // the channels of one group a lane owns are added in registers first (fixed order), then one LDS add per group
#pragma unroll
  for (int e0 = 0; e0 < EG; e0 += 4) {
    if ((cpg & 3) == 0) {
      const int grp = (cg * EG + e0) / cpg;
      det_add_lds<KD6D_DET_ACT>(&s_st[grp * 4], (a1[e0] + a1[e0 + 1]) + (a1[e0 + 2] + a1[e0 + 3]));
      det_add_lds<KD6D_DET_ACT>(&s_st[grp * 4 + 2], (a2[e0] + a2[e0 + 1]) + (a2[e0 + 2] + a2[e0 + 3]));
    } else {
#pragma unroll
      for (int e = e0; e < e0 + 4; ++e) {
        const int grp = (cg * EG + e) / cpg;
        det_add_lds<KD6D_DET_ACT>(&s_st[grp * 4], a1[e]);
        det_add_lds<KD6D_DET_ACT>(&s_st[grp * 4 + 2], a2[e]);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * G) {               // one {lo, hi} pair per thread
    acc_t* o = stats + ((size_t)(seg * gm.batch + b) * G) * 4 + threadIdx.x * 2;
    det_add_words(o, s_st[threadIdx.x * 2], s_st[threadIdx.x * 2 + 1]);
  }
}

__device__ __forceinline__ void gn_locate(const GnGeom& gm, long long row, int& seg, int& b, int& hw) {
  seg = 0; b = 0; hw = 1;
  int r0 = 0;
#pragma unroll
  for (int s = 0; s < KD6D_MAX_SEG; ++s)
    if (s < gm.nseg && row >= gm.row0[s]) { seg = s; r0 = gm.row0[s]; hw = gm.hw[s]; }
  b = (int)((row - r0) / hw);
}

template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void gn_relu_fwd_kernel(
    const TX* __restrict__ x, T* __restrict__ y, GnGeom gm, long long ngran, float eps,
    const acc_t* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta) {
  constexpr int EG = Granule<T>::N;
  const int C = gm.C, G = gm.G, cpg = C / G, cgs = C / EG;
  const int cg = threadIdx.x % cgs;
  float ga[EG], be[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) { ga[e] = gamma[cg * EG + e]; be[e] = beta[cg * EG + e]; }
  u32x4_t* yg = reinterpret_cast<u32x4_t*>(y);
  // A workgroup takes a CONTIGUOUS range of granules, kThreads at a time, so a thread's consecutive granules lie
  // kThreads / cgs rows apart and almost always in the same (level, image): the statistics of its one or two groups are
  // converted from their accumulators when that key changes, not per granule (4 conversions of ~25 instructions each
  // per granule made this kernel 1.7 us slower per launch than with fp32 statistics).
  const long long per = (((ngran + gridDim.x - 1) / gridDim.x + kThreads - 1) / kThreads) * kThreads;
  const long long g_begin = (long long)blockIdx.x * per;
  const long long g_end = g_begin + per < ngran ? g_begin + per : ngran;
  const int g0 = (cg * EG) / cpg;
  const bool two = (cpg % EG) != 0;       // a granule may touch two groups (cpg >= EG / 2); one when groups are whole granules
  int key = -1;
  float mu0 = 0.f, rs0 = 0.f, mu1 = 0.f, rs1 = 0.f;
#pragma unroll 4
  for (long long g = g_begin + threadIdx.x; g < g_end; g += kThreads) {
    const long long row = g / cgs;
    int seg, b, hw;
    gn_locate(gm, row, seg, b, hw);
    float v[EG];
    load_x<TX, EG>(x, g, v);
    const int k_sb = seg * gm.batch + b;
    if (k_sb != key) {
      key = k_sb;
      const float inv_n = 1.f / ((float)hw * (float)cpg);
      const acc_t* st = stats + ((size_t)k_sb * G) * 4;
      gn_mean_rstd(st + g0 * 4, inv_n, eps, mu0, rs0);
      if (two) gn_mean_rstd(st + min(g0 + 1, G - 1) * 4, inv_n, eps, mu1, rs1);
    }
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      const bool k = two && (cg * EG + e) / cpg != g0;
      const float t = (v[e] - (k ? mu1 : mu0)) * (k ? rs1 : rs0) * ga[e] + be[e];
      v[e] = fmaxf(t, 0.f);
    }
    yg[g] = f32_to_granule<T>(v);
  }
}

// backward reduce over row chunks:
//   gsum[(seg,b,g)] += {sum(dy*gamma), sum(dy*gamma*xhat)} (accumulators); dgamma / dbeta: accumulators of the gradient
//   bucket in its PLANAR layout (word lo of channel c at dgamma[c], hi at dgamma[c + acc_hi]; kd6d.h).
// Inside the workgroup the partial sums of a channel / a group meet in LDS accumulators (dynamic LDS: 2 * C of them).
template <int EG>
__device__ __forceinline__ void gn_bwd_block_sums(float (&a_dy)[EG], float (&a_dyx)[EG], const float (&ga)[EG], int C,
                                                  int cpg, int cgs, int cg, acc_t* red, acc_t* s_ab) {
  // lanes l, l + cgs, l + 2 cgs ... of a wave own the same channel granule (cgs = 16 or 32 here): fold them
  // with shuffles first (a fixed order) -- fewer same-address LDS atomics
  const bool fold = (cgs == 16 || cgs == 32);
  if (fold) {
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      a_dy[e] += __shfl_xor(a_dy[e], 32, 64);
      a_dyx[e] += __shfl_xor(a_dyx[e], 32, 64);
      if (cgs == 16) {
        a_dy[e] += __shfl_xor(a_dy[e], 16, 64);
        a_dyx[e] += __shfl_xor(a_dyx[e], 16, 64);
      }
    }
  }
  if (!fold || (int)(threadIdx.x & 63) < cgs) {
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      const int c = cg * EG + e;
      det_add_lds<KD6D_DET_GRAD>(&red[c * 2], a_dy[e]);
      det_add_lds<KD6D_DET_GRAD>(&red[(C + c) * 2], a_dyx[e]);
    }
    // group sums: add up the lane's channels of one group in registers first
#pragma unroll
    for (int e0 = 0; e0 < EG; e0 += 4) {        // cpg is 4 or a multiple of 4 here -> 4 aligned channels share a group
      const int c = cg * EG + e0;
      if ((cpg & 3) == 0) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int e = e0; e < e0 + 4; ++e) { sa += a_dy[e] * ga[e]; sb += a_dyx[e] * ga[e]; }
        det_add_lds<KD6D_DET_GRAD>(&s_ab[(c / cpg) * 4], sa);
        det_add_lds<KD6D_DET_GRAD>(&s_ab[(c / cpg) * 4 + 2], sb);
      } else {
#pragma unroll
        for (int e = e0; e < e0 + 4; ++e) {
          det_add_lds<KD6D_DET_GRAD>(&s_ab[((cg * EG + e) / cpg) * 4], a_dy[e] * ga[e]);
          det_add_lds<KD6D_DET_GRAD>(&s_ab[((cg * EG + e) / cpg) * 4 + 2], a_dyx[e] * ga[e]);
        }
      }
    }
  }
}

// the workgroup's per-channel sums (LDS accumulators) -> the gradient bucket's planar accumulators
__device__ __forceinline__ void gn_bwd_flush_param_grads(const acc_t* red, int C, acc_t* dgamma, acc_t* dbeta,
                                                         long long acc_hi) {
  for (int c = threadIdx.x; c < C; c += kThreads) {
    if (dbeta) det_add_words_planar(dbeta + c, acc_hi, red[c * 2], red[c * 2 + 1]);
    if (dgamma) det_add_words_planar(dgamma + c, acc_hi, red[(C + c) * 2], red[(C + c) * 2 + 1]);
  }
}

template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void gn_relu_bwd_reduce_kernel(
    const TX* __restrict__ x, const T* __restrict__ dz, GnGeom gm, float eps, const acc_t* __restrict__ stats,
    const float* __restrict__ gamma, const float* __restrict__ beta, acc_t* __restrict__ gsum,
    acc_t* dgamma, acc_t* dbeta, long long acc_hi) {
  constexpr int EG = Granule<T>::N;
  __shared__ acc_t s_ab[64 * 4];  // per group {sum dy*gamma, sum dy*gamma*xhat}
  extern __shared__ acc_t red[];  // 2*C accumulators
  int seg, b, r_begin, r_cnt;
  gn_chunk(gm, blockIdx.x, seg, b, r_begin, r_cnt);
  int hw = 1;
#pragma unroll
  for (int s = 0; s < KD6D_MAX_SEG; ++s)
    if (s == seg) hw = gm.hw[s];
  const int C = gm.C, G = gm.G, cpg = C / G, cgs = C / EG;
  const float inv_n = 1.f / ((float)hw * (float)cpg);
  s_ab[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < 4 * C; i += kThreads) red[i] = 0;
  __syncthreads();
  const int cg = threadIdx.x % cgs;
  const size_t sb = (size_t)(seg * gm.batch + b) * G;
  float ga[EG], be[EG], mu[EG], rs[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    ga[e] = gamma[c]; be[e] = beta[c];
    // (one conversion per group: the channels of a granule share one or two)
    if (e == 0 || c % cpg == 0) gn_mean_rstd(stats + (sb + c / cpg) * 4, inv_n, eps, mu[e], rs[e]);
    else { mu[e] = mu[e ? e - 1 : 0]; rs[e] = rs[e ? e - 1 : 0]; }
  }
  float a_dy[EG], a_dyx[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) { a_dy[e] = 0.f; a_dyx[e] = 0.f; }
  const size_t base = (size_t)r_begin * C;
  const TX* xb = x + base;
  const u32x4_t* dg = reinterpret_cast<const u32x4_t*>(dz + base);
  const long long ngran = (long long)r_cnt * cgs;
#pragma unroll 4
  for (long long g = threadIdx.x; g < ngran; g += kThreads) {
    float xv[EG], dv[EG];
    load_x<TX, EG>(xb, g, xv);
    granule_to_f32<T>(dg[g], dv);
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      const float xh = (xv[e] - mu[e]) * rs[e];
      const float pre = xh * ga[e] + be[e];
      const float d = pre > 0.f ? dv[e] : 0.f;
      a_dy[e] += d;
      a_dyx[e] += d * xh;
    }
  }
  gn_bwd_block_sums<EG>(a_dy, a_dyx, ga, C, cpg, cgs, cg, red, s_ab);
  __syncthreads();
  if (threadIdx.x < 2 * G) {
    acc_t* o = gsum + sb * 4 + threadIdx.x * 2;
    det_add_words(o, s_ab[threadIdx.x * 2], s_ab[threadIdx.x * 2 + 1]);
  }
  gn_bwd_flush_param_grads(red, C, dgamma, dbeta, acc_hi);
}

// GN+ReLU backward in ONE pass: the workgroup keeps its row chunk (<= kGnHold granules per thread) in registers,
// adds its partial sums to gsum, waits until the sibling workgroups of the same (level, image) -- at most
// hw / chunk_rows of them, consecutive block ids -- have done the same, and finishes dx from the registers: x and
// dz are read once instead of twice and the second launch goes.  Siblings are dispatched in order and the largest
// group is far smaller than the number of resident workgroups, so the wait always ends.
constexpr int kGnHold = 8;
template <typename T, typename TX>
__device__ __forceinline__ void gn_relu_bwd_onepass_body(
    const TX* __restrict__ x, const T* __restrict__ dz, T* __restrict__ dx, const GnGeom& gm, float eps,
    const acc_t* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
    acc_t* gsum, unsigned int* counters, acc_t* dgamma, acc_t* dbeta, long long acc_hi, int bid) {
  constexpr int EG = Granule<T>::N;
  __shared__ acc_t s_ab[64 * 4];
  extern __shared__ acc_t red[];  // 2*C accumulators
  int seg, b, r_begin, r_cnt;
  gn_chunk(gm, bid, seg, b, r_begin, r_cnt);
  int hw = 1;
#pragma unroll
  for (int s = 0; s < KD6D_MAX_SEG; ++s)
    if (s == seg) hw = gm.hw[s];
  const int C = gm.C, G = gm.G, cpg = C / G, cgs = C / EG;
  const float inv_n = 1.f / ((float)hw * (float)cpg);
  s_ab[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < 4 * C; i += kThreads) red[i] = 0;
  __syncthreads();
  const int cg = threadIdx.x % cgs;
  const size_t sb = (size_t)(seg * gm.batch + b) * G;
  float ga[EG], be[EG], mu[EG], rs[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) {
    const int c = cg * EG + e;
    ga[e] = gamma[c]; be[e] = beta[c];
    // (one conversion per group: the channels of a granule share one or two)
    if (e == 0 || c % cpg == 0) gn_mean_rstd(stats + (sb + c / cpg) * 4, inv_n, eps, mu[e], rs[e]);
    else { mu[e] = mu[e ? e - 1 : 0]; rs[e] = rs[e ? e - 1 : 0]; }
  }
  float a_dy[EG], a_dyx[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) { a_dy[e] = 0.f; a_dyx[e] = 0.f; }
  const size_t base = (size_t)r_begin * C;
  const TX* xb = x + base;
  const u32x4_t* dg = reinterpret_cast<const u32x4_t*>(dz + base);
  u32x4_t* og = reinterpret_cast<u32x4_t*>(dx + base);
  const int ngran = r_cnt * cgs;
  float hd[kGnHold][EG], hx[kGnHold][EG];     // masked dz, xhat
  // straight-line loads (index clamped, contribution masked): all of a thread's granules are in flight at once
#pragma unroll
  for (int i = 0; i < kGnHold; ++i) {
    const int g = threadIdx.x + i * kThreads;
    const bool valid = g < ngran;
    const int gi = valid ? g : ngran - 1;
    float xv[EG], dv[EG];
    load_x<TX, EG>(xb, gi, xv);
    granule_to_f32<T>(dg[gi], dv);
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      const float xh = (xv[e] - mu[e]) * rs[e];
      const float pre = xh * ga[e] + be[e];
      const float d = (valid && pre > 0.f) ? dv[e] : 0.f;
      hd[i][e] = d; hx[i][e] = xh;
      a_dy[e] += d;
      a_dyx[e] += d * xh;
    }
  }
  gn_bwd_block_sums<EG>(a_dy, a_dyx, ga, C, cpg, cgs, cg, red, s_ab);
  __syncthreads();
  if (threadIdx.x < 2 * G) {
    acc_t* o = gsum + sb * 4 + threadIdx.x * 2;
    det_add_words_performed(o, s_ab[threadIdx.x * 2], s_ab[threadIdx.x * 2 + 1]);
  }
  // (the dgamma / dbeta atomics -- every workgroup of the launch on the same C addresses -- wait until the end:
  //  the siblings do not need them and the barrier would otherwise sit behind that queue)
  group_barrier(counters + seg * gm.batch + b, (unsigned)((hw + gm.chunk_rows - 1) / gm.chunk_rows), gm.timeouts);
  float k1[EG], k2[EG];
#pragma unroll
  for (int e0 = 0; e0 < EG; e0 += 4) {           // 4 aligned channels share a group when cpg % 4 == 0
#pragma unroll
    for (int e = e0; e < e0 + 4; ++e) {
      if (e == e0 || (cpg & 3) != 0) {
        const acc_t* o = gsum + (sb + (cg * EG + e) / cpg) * 4;
        const det_words w1 = det_load_device_scope(o), w2 = det_load_device_scope(o + 2);
        k1[e] = det_value<KD6D_DET_GRAD>(w1.lo, w1.hi) * inv_n;
        k2[e] = det_value<KD6D_DET_GRAD>(w2.lo, w2.hi) * inv_n;
      } else {
        k1[e] = k1[e0];
        k2[e] = k2[e0];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kGnHold; ++i) {
    const int g = threadIdx.x + i * kThreads;
    if (g < ngran) {
      float o[EG];
#pragma unroll
      for (int e = 0; e < EG; ++e) o[e] = rs[e] * (hd[i][e] * ga[e] - k1[e] - hx[i][e] * k2[e]);
      og[g] = f32_to_granule<T>(o);
    }
  }
  gn_bwd_flush_param_grads(red, C, dgamma, dbeta, acc_hi);
}

template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void gn_relu_bwd_onepass_kernel(
    const TX* __restrict__ x, const T* __restrict__ dz, T* __restrict__ dx, GnGeom gm, float eps,
    const acc_t* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
    acc_t* gsum, unsigned int* counters, acc_t* dgamma, acc_t* dbeta, long long acc_hi) {
  gn_relu_bwd_onepass_body<T, TX>(x, dz, dx, gm, eps, stats, gamma, beta, gsum, counters, dgamma, dbeta, acc_hi, blockIdx.x);
}

// Two GroupNorm backwards of identical geometry (the cls and the pose tower layer of the head) as ONE launch:
// workgroups [0, gm.nblk) take the first tensor set, the rest the second.  One of them alone is 170 workgroups of
// 4 waves on 256 CUs and latency-bound; back to back on one stream the second waited for the first.
struct GnBwdSet {
  const void* x; const void* dz; void* dx;
  const acc_t* stats; const float* gamma; const float* beta;
  acc_t* gsum; unsigned int* counters; acc_t* dgamma; acc_t* dbeta;
};
template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void gn_relu_bwd_onepass_pair_kernel(GnBwdSet a, GnBwdSet b, GnGeom gm, float eps,
                                                                            long long acc_hi) {
  const bool first = (int)blockIdx.x < gm.nblk;
  const GnBwdSet& q = first ? a : b;
  gn_relu_bwd_onepass_body<T, TX>((const TX*)q.x, (const T*)q.dz, (T*)q.dx, gm, eps, q.stats, q.gamma, q.beta, q.gsum,
                                  q.counters, q.dgamma, q.dbeta, acc_hi, first ? blockIdx.x : blockIdx.x - gm.nblk);
}

template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void gn_relu_bwd_apply_kernel(
    const TX* __restrict__ x, const T* __restrict__ dz, T* __restrict__ dx, GnGeom gm,
    long long ngran, float eps, const acc_t* __restrict__ stats, const acc_t* __restrict__ gsum,
    const float* __restrict__ gamma, const float* __restrict__ beta) {
  constexpr int EG = Granule<T>::N;
  const int C = gm.C, G = gm.G, cpg = C / G, cgs = C / EG;
  const int cg = threadIdx.x % cgs;
  float ga[EG], be[EG];
#pragma unroll
  for (int e = 0; e < EG; ++e) { ga[e] = gamma[cg * EG + e]; be[e] = beta[cg * EG + e]; }
  const u32x4_t* dg = reinterpret_cast<const u32x4_t*>(dz);
  u32x4_t* og = reinterpret_cast<u32x4_t*>(dx);
#pragma unroll 4
  for (long long g = (long long)blockIdx.x * kThreads + threadIdx.x; g < ngran;
       g += (long long)gridDim.x * kThreads) {
    const long long row = g / cgs;
    int seg, b, hw;
    gn_locate(gm, row, seg, b, hw);
    const float inv_n = 1.f / ((float)hw * (float)cpg);
    const size_t sb = (size_t)(seg * gm.batch + b) * G;
    float xv[EG], dv[EG];
    load_x<TX, EG>(x, g, xv);
    granule_to_f32<T>(dg[g], dv);
    constexpr int NG = 2;
    const int g0 = (cg * EG) / cpg;
    float mu[NG], rs[NG], k1[NG], k2[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int gi = min(g0 + k, G - 1);
      mu[k] = rs[k] = k1[k] = k2[k] = 0.f;
      if (k && (cpg % EG) == 0) continue;         // groups are whole granules: the second one is never read
      gn_mean_rstd(stats + (sb + gi) * 4, inv_n, eps, mu[k], rs[k]);
      k1[k] = det_read<KD6D_DET_GRAD>(gsum + (sb + gi) * 4) * inv_n;
      k2[k] = det_read<KD6D_DET_GRAD>(gsum + (sb + gi) * 4 + 2) * inv_n;
    }
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      const int k = (cg * EG + e) / cpg - g0;
      const float m_ = k ? mu[1] : mu[0], r_ = k ? rs[1] : rs[0];
      const float xh = (xv[e] - m_) * r_;
      const float pre = xh * ga[e] + be[e];
      const float d = pre > 0.f ? dv[e] * ga[e] : 0.f;
      dv[e] = r_ * (d - (k ? k1[1] : k1[0]) - xh * (k ? k2[1] : k2[0]));
    }
    og[g] = f32_to_granule<T>(dv);
  }
}

// ---------------------------------------------------------------------------
// MaxPool 2x2/2, nearest x2 upsample + add, 2x2 sum-pool (its adjoint), ReLU.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void maxpool2_fwd_kernel(const T* __restrict__ x,
                                                                T* __restrict__ y, int B, int H,
                                                                int W, int C) {
  constexpr int EG = Granule<T>::N;
  const int Ho = H / 2, Wo = W / 2, cgs = C / EG;
  const long long total = (long long)B * Ho * Wo * cgs;
  const u32x4_t* xg = reinterpret_cast<const u32x4_t*>(x);
  u32x4_t* yg = reinterpret_cast<u32x4_t*>(y);
  for (long long g = (long long)blockIdx.x * kThreads + threadIdx.x; g < total;
       g += (long long)gridDim.x * kThreads) {
    const int cg = (int)(g % cgs);
    long long r = g / cgs;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float best[EG];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int iy = oy * 2 + (k >> 1), ix = ox * 2 + (k & 1);
      float v[EG];
      granule_to_f32<T>(xg[((long long)(b * H + iy) * W + ix) * cgs + cg], v);
#pragma unroll
      for (int e = 0; e < EG; ++e) best[e] = (k == 0 || v[e] > best[e]) ? v[e] : best[e];
    }
    yg[g] = f32_to_granule<T>(best);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void maxpool2_bwd_kernel(const T* __restrict__ x,
                                                                const T* __restrict__ dy,
                                                                T* __restrict__ dx, int B, int H,
                                                                int W, int C, int accumulate) {
  constexpr int EG = Granule<T>::N;
  const int Ho = H / 2, Wo = W / 2, cgs = C / EG;
  const long long total = (long long)B * Ho * Wo * cgs;
  const u32x4_t* xg = reinterpret_cast<const u32x4_t*>(x);
  const u32x4_t* dg = reinterpret_cast<const u32x4_t*>(dy);
  u32x4_t* og = reinterpret_cast<u32x4_t*>(dx);
  for (long long g = (long long)blockIdx.x * kThreads + threadIdx.x; g < total;
       g += (long long)gridDim.x * kThreads) {
    const int cg = (int)(g % cgs);
    long long r = g / cgs;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float v[4][EG], d[EG], best[EG];
    int arg[EG];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int iy = oy * 2 + (k >> 1), ix = ox * 2 + (k & 1);
      granule_to_f32<T>(xg[((long long)(b * H + iy) * W + ix) * cgs + cg], v[k]);
#pragma unroll
      for (int e = 0; e < EG; ++e)
        if (k == 0 || v[k][e] > best[e]) { best[e] = v[k][e]; arg[e] = k; }
    }
    granule_to_f32<T>(dg[g], d);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int iy = oy * 2 + (k >> 1), ix = ox * 2 + (k & 1);
      const long long o = ((long long)(b * H + iy) * W + ix) * cgs + cg;
      float out[EG];
      if (accumulate) granule_to_f32<T>(og[o], out);
#pragma unroll
      for (int e = 0; e < EG; ++e) {
        const float t = arg[e] == k ? d[e] : 0.f;
        out[e] = accumulate ? out[e] + t : t;
      }
      og[o] = f32_to_granule<T>(out);
    }
  }
}

// out[b,y,x,:] = fine[b,y,x,:] + coarse[b,y/2,x/2,:]   (fine grid H x W, coarse H/2 x W/2)
template <typename T>
__global__ __launch_bounds__(kThreads) void upsample2_add_kernel(const T* __restrict__ fine,
                                                                 const T* __restrict__ coarse,
                                                                 T* __restrict__ out, int B, int H,
                                                                 int W, int C) {
  constexpr int EG = Granule<T>::N;
  const int cgs = C / EG, Hc = H / 2, Wc = W / 2;
  const long long total = (long long)B * H * W * cgs;
  const u32x4_t* fg = reinterpret_cast<const u32x4_t*>(fine);
  const u32x4_t* cgp = reinterpret_cast<const u32x4_t*>(coarse);
  u32x4_t* og = reinterpret_cast<u32x4_t*>(out);
  for (long long g = (long long)blockIdx.x * kThreads + threadIdx.x; g < total;
       g += (long long)gridDim.x * kThreads) {
    const int cg = (int)(g % cgs);
    long long r = g / cgs;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    float a[EG], c[EG];
    granule_to_f32<T>(fg[g], a);
    granule_to_f32<T>(cgp[((long long)(b * Hc + (y >> 1)) * Wc + (x >> 1)) * cgs + cg], c);
#pragma unroll
    for (int e = 0; e < EG; ++e) a[e] += c[e];
    og[g] = f32_to_granule<T>(a);
  }
}

// dcoarse[b,y,x,:] (+)= sum of the 2x2 fine block of dfine
template <typename T>
__global__ __launch_bounds__(kThreads) void sumpool2_kernel(const T* __restrict__ dfine,
                                                            T* __restrict__ dcoarse, int B, int H,
                                                            int W, int C, int accumulate) {
  constexpr int EG = Granule<T>::N;
  const int cgs = C / EG, Hc = H / 2, Wc = W / 2;
  const long long total = (long long)B * Hc * Wc * cgs;
  const u32x4_t* fg = reinterpret_cast<const u32x4_t*>(dfine);
  u32x4_t* og = reinterpret_cast<u32x4_t*>(dcoarse);
  for (long long g = (long long)blockIdx.x * kThreads + threadIdx.x; g < total;
       g += (long long)gridDim.x * kThreads) {
    const int cg = (int)(g % cgs);
    long long r = g / cgs;
    const int x = (int)(r % Wc); r /= Wc;
    const int y = (int)(r % Hc);
    const int b = (int)(r / Hc);
    float s[EG];
    if (accumulate) granule_to_f32<T>(og[g], s);
    else {
#pragma unroll
      for (int e = 0; e < EG; ++e) s[e] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v[EG];
      granule_to_f32<T>(fg[((long long)(b * H + 2 * y + (k >> 1)) * W + 2 * x + (k & 1)) * cgs + cg], v);
#pragma unroll
      for (int e = 0; e < EG; ++e) s[e] += v[e];
    }
    og[g] = f32_to_granule<T>(s);
  }
}

// mode 0: y = relu(x);  mode 1: y = (x > 0) ? dy : 0   (ReLU backward with x = forward input)
// mode 2: y = x + dy (elementwise add)
template <typename T>
__global__ __launch_bounds__(kThreads) void eltwise_kernel(const T* __restrict__ x,
                                                           const T* __restrict__ dy,
                                                           T* __restrict__ y, long long ngran,
                                                           int mode) {
  constexpr int EG = Granule<T>::N;
  const u32x4_t* xg = reinterpret_cast<const u32x4_t*>(x);
  const u32x4_t* dg = reinterpret_cast<const u32x4_t*>(dy);
  u32x4_t* yg = reinterpret_cast<u32x4_t*>(y);
#pragma unroll 4
  for (long long g = (long long)blockIdx.x * kThreads + threadIdx.x; g < ngran;
       g += (long long)gridDim.x * kThreads) {
    float a[EG], d[EG];
    granule_to_f32<T>(xg[g], a);
    if (mode != 0) granule_to_f32<T>(dg[g], d);
#pragma unroll
    for (int e = 0; e < EG; ++e) {
      if (mode == 0) a[e] = fmaxf(a[e], 0.f);
      else if (mode == 1) a[e] = a[e] > 0.f ? d[e] : 0.f;
      else a[e] = a[e] + d[e];
    }
    yg[g] = f32_to_granule<T>(a);
  }
}

// NCHW fp32 image (B,Cimg,H,W) -> NHWC T (B,H,W,Cpad), zero padded channels
template <typename T>
__global__ __launch_bounds__(kThreads) void image_to_nhwc_kernel(const float* __restrict__ img,
                                                                 T* __restrict__ out, int B,
                                                                 int Cimg, int H, int W, int Cpad) {
  const long long total = (long long)B * H * W;
  for (long long p = (long long)blockIdx.x * kThreads + threadIdx.x; p < total;
       p += (long long)gridDim.x * kThreads) {
    const long long hw = (long long)H * W;
    const int b = (int)(p / hw);
    const long long r = p - (long long)b * hw;
    if (sizeof(T) == 2 && Cpad == 8) {      // the step's case: one 16-byte store per pixel instead of eight 2-byte ones
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = c < Cimg ? img[((long long)b * Cimg + c) * hw + r] : 0.f;
      u32x4_t g;
      g.x = pack_bf16x2(v[0], v[1]); g.y = pack_bf16x2(v[2], v[3]);
      g.z = pack_bf16x2(v[4], v[5]); g.w = pack_bf16x2(v[6], v[7]);
      *reinterpret_cast<u32x4_t*>(out + p * 8) = g;
      continue;
    }
    for (int c = 0; c < Cpad; ++c) {
      const float v = c < Cimg ? img[((long long)b * Cimg + c) * hw + r] : 0.f;
      out[p * Cpad + c] = from_f32<T>(v);
    }
  }
}

// Every workgroup of a per-channel reduction ends with one atomic per channel, and same-address atomics
// retire serially, ~27 ns each (measured: time grows linearly with the workgroup count, 512 -> 2048 = 18 -> 55 us
// on a 17-MB tensor).  colstats' plain load loop tolerates long per-thread walks, so it is capped at 128
// workgroups; the BatchNorm backward reduction is latency-bound per thread (8 granules at most) and keeps 512.
constexpr int kColstatsCap = 128;
constexpr int kBnBwdReduceCap = 512;

int grid_for(long long work_items) {
  long long b = (work_items + kThreads - 1) / kThreads;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

int check_channels(int dtype, int C, const char* who) {
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "%s: bad dtype %d", who, dtype);
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(C > 0 && C % eg == 0, "%s: C=%d must be a multiple of %d", who, C, eg);
  const int cgs = C / eg;
  KD6D_CHECK_ARG(cgs <= 256 && 256 % cgs == 0, "%s: C=%d: C/%d must divide 256", who, C, eg);
  return KD6D_OK;
}

bool fill_gn(const int32_t* level_hw, int nseg, int batch, int C, int G, GnGeom* gm, int chunk_rows = 0) {
  if (nseg < 1 || nseg > KD6D_MAX_SEG || batch < 1 || G < 1 || G > 64 || C % G) return false;
  gm->nseg = nseg; gm->batch = batch; gm->C = C; gm->G = G;
  gm->chunk_rows = chunk_rows > 0 ? chunk_rows : gn_chunk_rows();
  gm->timeouts = kd6d_ctx_timeouts_ptr();
  int row = 0, blk = 0;
  for (int s = 0; s < KD6D_MAX_SEG; ++s) {
    gm->row0[s] = row;
    gm->hw[s] = s < nseg ? level_hw[s] : 0;
    gm->blk0[s] = blk;
    gm->cps[s] = 1;
    if (s < nseg) {
      if (level_hw[s] <= 0) return false;
      row += batch * level_hw[s];
      gm->cps[s] = (level_hw[s] + gm->chunk_rows - 1) / gm->chunk_rows;
      blk += batch * gm->cps[s];
    }
  }
  gm->nblk = blk;
  return true;
}

// Workgroups of `kernel` the device keeps resident at once (occupancy x CUs); 0 if the query fails.  The kernels
// with an in-kernel barrier are only launched with grids well inside it.
template <typename K>
int resident_workgroups(K kernel, size_t lds) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kThreads, lds) != hipSuccess) return 0;
  return per_cu * kd6d_device_cu_count();
}
template <typename T, typename TX>
int bn_onepass_capacity(bool pooled) {
  static const int plain = resident_workgroups(bn_bwd_onepass_kernel<T, TX>, 16384);
  static const int pool = resident_workgroups(bn_pool_bwd_onepass_kernel<T, TX>, 16384);
  return pooled ? pool : plain;
}
template <typename T, typename TX>
int gn_onepass_capacity() {
  static const int v = resident_workgroups(gn_relu_bwd_onepass_kernel<T, TX>, 16384);
  return v;
}
// option bn.onepass = 0: the two-launch BN backward even when the caller passes a barrier counter
bool bn_onepass() { return kd6d_opt(KD6D_OPT_BN_ONEPASS) != 0; }
// Largest tensor (16-B granules of x) that takes the one-launch BN backward.  Measured on an MI355X
// (tools/bench_norm.py): up to ~128 workgroups the barrier is cheaper than a second launch (11-13 us against
// 15 us per layer); at 256-512 workgroups publishing and collecting the partial sums through device-scope
// returning atomics costs more than re-reading x and dz (27-29 us against 20 us), so those keep the pair.
// Option bn.onepass_max = <granules> overrides the 65536 (tests drive 512-workgroup grids through it).
long long bn_onepass_max_granules() { return kd6d_opt(KD6D_OPT_BN_ONEPASS_MAX); }
// option gn.onepass = 0: the two-launch GN backward (reduce, apply)
bool gn_onepass() { return kd6d_opt(KD6D_OPT_GN_ONEPASS) != 0; }

long long gn_rows(const GnGeom& gm) {
  long long r = 0;
  for (int s = 0; s < gm.nseg; ++s) r += (long long)gm.batch * gm.hw[s];
  return r;
}

}  // namespace

namespace kd6d_detail {
int colsum_grad_planar(int dtype, const void* x, int64_t rows, int C, long long* acc, long long acc_hi, void* stream);
int gn_stats_levels(int src_f32, const void* y, const int* row0, const int* hw, int nseg, int batch, int C, int G,
                    unsigned mask, long long* stats, hipStream_t st);
}

#define DISPATCH_T(dtype, expr_bf16, expr_f32) \
  do { if ((dtype) == KD6D_BF16) { expr_bf16; } else { expr_f32; } } while (0)

extern "C" int kd6d_colstats(int dtype, const void* x, int64_t rows, int C, kd6d_acc* sum_acc,
                             kd6d_acc* sumsq_acc, void* stream) {
  acc_t* sum = reinterpret_cast<acc_t*>(sum_acc);
  acc_t* sumsq = reinterpret_cast<acc_t*>(sumsq_acc);
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "kd6d_colstats: bad dtype %d", dtype);
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(C > 0 && C % eg == 0 && C / eg <= kThreads, "kd6d_colstats: C=%d must be a multiple of %d and <= %d",
                 C, eg, eg * kThreads);
  KD6D_CHECK_ARG(x && rows > 0, "kd6d_colstats: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rpp = kThreads / (C / eg);
  long long nb = (rows + (long long)rpp * 8 - 1) / ((long long)rpp * 8);
  if (nb > kColstatsCap) nb = kColstatsCap;
  if (nb < 1) nb = 1;
  const size_t lds = kFlushLdsBytes;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(colstats_kernel<bf16_t>, dim3((int)nb), dim3(kThreads), lds, st,
                                (const bf16_t*)x, (long long)rows, C, sum, sumsq, 0ll),
             hipLaunchKernelGGL(colstats_kernel<float>, dim3((int)nb), dim3(kThreads), lds, st,
                                (const float*)x, (long long)rows, C, sum, sumsq, 0ll));
  KD6D_CHECK_LAUNCH("kd6d_colstats");
  return KD6D_OK;
}

// conv_igemm.hip (stats_followup): the group statistics of the levels in `mask`, from the tensor the convolution stored.
// Unmasked levels get no workgroups (their row-chunk count is 0); the statistics index keeps the level's own number.
int kd6d_detail::gn_stats_levels(int src_f32, const void* y, const int* row0, const int* hw, int nseg, int batch, int C,
                                 int G, unsigned mask, long long* stats, hipStream_t st) {
  KD6D_CHECK_ARG(y && stats && nseg >= 1 && nseg <= KD6D_MAX_SEG && batch >= 1 && G >= 1 && G <= 64 && C % G == 0,
                 "gn_stats_levels: bad arguments");
  const int eg = src_f32 ? 4 : 8;
  KD6D_CHECK_ARG(C % eg == 0 && (C / eg) <= 256 && 256 % (C / eg) == 0 && (C / G) * 2 >= eg, "gn_stats_levels: C=%d G=%d", C, G);
  GnGeom gm;
  gm.nseg = nseg; gm.batch = batch; gm.C = C; gm.G = G;
  gm.chunk_rows = gn_chunk_rows();
  gm.timeouts = nullptr;
  int blk = 0;
  for (int s = 0; s < KD6D_MAX_SEG; ++s) {
    gm.row0[s] = s < nseg ? row0[s] : 0;
    gm.hw[s] = s < nseg ? hw[s] : 0;
    gm.blk0[s] = blk;
    gm.cps[s] = 1;
    if (s < nseg && ((mask >> s) & 1u)) {
      gm.cps[s] = (hw[s] + gm.chunk_rows - 1) / gm.chunk_rows;
      blk += batch * gm.cps[s];
    }
  }
  gm.nblk = blk;
  if (blk == 0) return KD6D_OK;
  // gn_chunk() picks the LAST level whose first workgroup is <= the workgroup id: levels without workgroups share their
  // first id with the next masked level and lose to it; trailing unmasked levels are moved out of range
  for (int s = KD6D_MAX_SEG - 1; s >= 0 && !(s < nseg && ((mask >> s) & 1u)); --s) gm.blk0[s] = blk;
  if (src_f32) hipLaunchKernelGGL((gn_stats_kernel<float, float>), dim3(blk), dim3(kThreads), 0, st, (const float*)y, gm, stats);
  else hipLaunchKernelGGL((gn_stats_kernel<bf16_t, bf16_t>), dim3(blk), dim3(kThreads), 0, st, (const bf16_t*)y, gm, stats);
  KD6D_CHECK_LAUNCH("gn_stats_levels");
  return KD6D_OK;
}

// conv_igemm.hip (exact-fp32 weight gradient): the bias gradient as a separate column-sum pass
int kd6d_detail::colsum_grad_planar(int dtype, const void* x, int64_t rows, int C, long long* acc, long long acc_hi,
                                    void* stream) {
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "colsum_grad_planar: bad dtype %d", dtype);
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(C > 0 && C % eg == 0 && C / eg <= kThreads && x && acc && rows > 0, "colsum_grad_planar: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rpp = kThreads / (C / eg);
  long long nb = (rows + (long long)rpp * 8 - 1) / ((long long)rpp * 8);
  if (nb > kColstatsCap) nb = kColstatsCap;
  if (nb < 1) nb = 1;
  const size_t lds = kFlushLdsBytes;
  acc_t* none = nullptr;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((colstats_kernel<bf16_t, true>), dim3((int)nb), dim3(kThreads), lds, st,
                                (const bf16_t*)x, (long long)rows, C, acc, none, acc_hi),
             hipLaunchKernelGGL((colstats_kernel<float, true>), dim3((int)nb), dim3(kThreads), lds, st,
                                (const float*)x, (long long)rows, C, acc, none, acc_hi));
  KD6D_CHECK_LAUNCH("colsum_grad_planar");
  return KD6D_OK;
}

#define DISPATCH_TTX(dtype, xf32, CALL)                                        \
  do {                                                                         \
    if ((dtype) == KD6D_F32) { typedef float T_; typedef float TX_; CALL; }     \
    else if (xf32) { typedef bf16_t T_; typedef float TX_; CALL; }              \
    else { typedef bf16_t T_; typedef bf16_t TX_; CALL; }                       \
  } while (0)

extern "C" int kd6d_bn_train_fwd(int dtype, int x_f32, const void* x, void* y, int64_t rows, int C,
                                 const kd6d_acc* sum_acc, const kd6d_acc* sumsq_acc, const float* gamma,
                                 const float* beta, float eps, float momentum, float* running_mean,
                                 float* running_var, float* save_mean, float* save_invstd, int act,
                                 void* stream) {
  int rc = check_channels(dtype, C, "kd6d_bn_train_fwd");
  if (rc) return rc;
  const acc_t* sum = reinterpret_cast<const acc_t*>(sum_acc);
  const acc_t* sumsq = reinterpret_cast<const acc_t*>(sumsq_acc);
  KD6D_CHECK_ARG(x && y && sum && sumsq && gamma && beta && rows > 0, "kd6d_bn_train_fwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  const long long ngran = rows * (C / eg);
  const float inv_rows = 1.f / (float)rows;
  const float unbias = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
  const int nb = grid_for((ngran + 3) / 4);
  DISPATCH_TTX(dtype, x_f32,
               hipLaunchKernelGGL((bn_apply_fwd_kernel<T_, TX_>), dim3(nb), dim3(kThreads), 0, st,
                                   (const TX_*)x, (T_*)y, ngran, C, inv_rows, sum, sumsq, gamma, beta, eps,
                                   momentum, unbias, running_mean, running_var, save_mean, save_invstd, act));
  KD6D_CHECK_LAUNCH("kd6d_bn_train_fwd");
  return KD6D_OK;
}

extern "C" int kd6d_bn_train_bwd_reduce(int dtype, int x_f32, const void* x, const void* dz, int64_t rows,
                                        int C, const float* mean, const float* invstd, const float* gamma,
                                        const float* beta, int act, kd6d_acc* sum_dy_acc, kd6d_acc* sum_dy_xhat_acc,
                                        int replicas, void* stream) {
  int rc = check_channels(dtype, C, "kd6d_bn_train_bwd_reduce");
  if (rc) return rc;
  acc_t* sum_dy = reinterpret_cast<acc_t*>(sum_dy_acc);
  acc_t* sum_dy_xhat = reinterpret_cast<acc_t*>(sum_dy_xhat_acc);
  KD6D_CHECK_ARG(replicas >= 1 && replicas <= 64, "kd6d_bn_train_bwd_reduce: replicas=%d outside [1,64]", replicas);
  KD6D_CHECK_ARG(x && dz && mean && invstd && gamma && beta && sum_dy && sum_dy_xhat && rows > 0,
                 "kd6d_bn_train_bwd_reduce: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  const long long ngran = rows * (C / eg);
  // every workgroup ends with one global atomic per channel and same-address atomics retire serially
  // (~25 ns each): 2 workgroups per CU keep the loads in flight without a 1000-deep atomic queue
  long long nb = (ngran + kThreads * 8 - 1) / (kThreads * 8);
  if (nb > kBnBwdReduceCap) nb = kBnBwdReduceCap;
  if (nb < 1) nb = 1;
  const size_t lds = kFlushLdsBytes;
  DISPATCH_TTX(dtype, x_f32,
               hipLaunchKernelGGL((bn_bwd_reduce_kernel<T_, TX_>), dim3((int)nb), dim3(kThreads), lds, st,
                                   (const TX_*)x, (const T_*)dz, ngran, C, mean, invstd, gamma, beta, act, sum_dy,
                                   sum_dy_xhat, replicas));
  KD6D_CHECK_LAUNCH("kd6d_bn_train_bwd_reduce");
  return KD6D_OK;
}

extern "C" int kd6d_bn_train_bwd_apply(int dtype, int x_f32, const void* x, const void* dz, void* dx,
                                       int64_t rows, int C, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, int act,
                                       const kd6d_acc* sum_dy_acc, const kd6d_acc* sum_dy_xhat_acc, float* dgamma,
                                       float* dbeta, int replicas, void* stream) {
  int rc = check_channels(dtype, C, "kd6d_bn_train_bwd_apply");
  if (rc) return rc;
  const acc_t* sum_dy = reinterpret_cast<const acc_t*>(sum_dy_acc);
  const acc_t* sum_dy_xhat = reinterpret_cast<const acc_t*>(sum_dy_xhat_acc);
  KD6D_CHECK_ARG(replicas >= 1 && replicas <= 64, "kd6d_bn_train_bwd_apply: replicas=%d outside [1,64]", replicas);
  KD6D_CHECK_ARG(x && dz && dx && mean && invstd && gamma && beta && sum_dy && sum_dy_xhat && rows > 0,
                 "kd6d_bn_train_bwd_apply: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  const long long ngran = rows * (C / eg);
  const int nb = grid_for((ngran + 3) / 4);
  const float inv_rows = 1.f / (float)rows;
  DISPATCH_TTX(dtype, x_f32,
               hipLaunchKernelGGL((bn_bwd_apply_kernel<T_, TX_>), dim3(nb), dim3(kThreads),
                                   (size_t)2 * C * sizeof(float), st,
                                   (const TX_*)x, (const T_*)dz, (T_*)dx, ngran, C, inv_rows, mean, invstd, gamma,
                                   beta, act, sum_dy, sum_dy_xhat, dgamma, dbeta, replicas));   /* LDS: the totals as floats */
  KD6D_CHECK_LAUNCH("kd6d_bn_train_bwd_apply");
  return KD6D_OK;
}

extern "C" int kd6d_bn_train_bwd(int dtype, int x_f32, const void* x, const void* dz, void* dx, int64_t rows, int C,
                                 const float* mean, const float* invstd, const float* gamma, const float* beta,
                                 int act, kd6d_acc* sum_dy_acc, kd6d_acc* sum_dy_xhat_acc, unsigned int* counter, float* dgamma,
                                 float* dbeta, int replicas, void* stream) {
  int rc = check_channels(dtype, C, "kd6d_bn_train_bwd");
  if (rc) return rc;
  acc_t* sum_dy = reinterpret_cast<acc_t*>(sum_dy_acc);
  acc_t* sum_dy_xhat = reinterpret_cast<acc_t*>(sum_dy_xhat_acc);
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  const long long ngran = rows * (C / eg);
  int cap = 0;
  DISPATCH_TTX(dtype, x_f32, cap = (bn_onepass_capacity<T_, TX_>(false)));
  cap = cap * 3 / 4;                                      // the whole grid must be resident for the barrier
  if (cap > kBnOnepassBlocks) cap = kBnOnepassBlocks;
  if (!(counter && bn_onepass() && rows > 0 && cap > 0 && ngran <= (long long)cap * kThreads * kBnHold &&
        ngran <= bn_onepass_max_granules())) {
    rc = kd6d_bn_train_bwd_reduce(dtype, x_f32, x, dz, rows, C, mean, invstd, gamma, beta, act, sum_dy_acc, sum_dy_xhat_acc,
                                  replicas, stream);
    if (rc) return rc;
    return kd6d_bn_train_bwd_apply(dtype, x_f32, x, dz, dx, rows, C, mean, invstd, gamma, beta, act, sum_dy_acc,
                                   sum_dy_xhat_acc, dgamma, dbeta, replicas, stream);
  }
  KD6D_CHECK_ARG(replicas >= 1 && replicas <= 64, "kd6d_bn_train_bwd: replicas=%d outside [1,64]", replicas);
  KD6D_CHECK_ARG(x && dz && dx && mean && invstd && gamma && beta && sum_dy && sum_dy_xhat,
                 "kd6d_bn_train_bwd: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  long long nb = (ngran + 2 * kThreads - 1) / (2 * kThreads);       // two granules per thread while the device has room
  if (nb > cap) nb = cap;
  const int per = (int)((ngran + nb * kThreads - 1) / (nb * kThreads));
  const float inv_rows = 1.f / (float)rows;
  DISPATCH_TTX(dtype, x_f32,
               hipLaunchKernelGGL((bn_bwd_onepass_kernel<T_, TX_>), dim3((int)nb), dim3(kThreads),
                                  (size_t)kFlushLdsBytes, st, (const TX_*)x, (const T_*)dz, (T_*)dx, ngran, per,
                                  C, inv_rows, mean, invstd, gamma, beta, act, sum_dy, sum_dy_xhat, counter, dgamma,
                                  dbeta, replicas, kd6d_ctx_timeouts_ptr()));
  KD6D_CHECK_LAUNCH("kd6d_bn_train_bwd");
  return KD6D_OK;
}

unsigned int* kd6d_detail::barrier_timeouts_device_ptr() {
  static unsigned int* ptr = []() {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_barrier_timeouts)) != hipSuccess) q = nullptr;
    return reinterpret_cast<unsigned int*>(q);
  }();
  return ptr;
}


static int check_pool(int dtype, int B, int H, int W, int C, const char* who, long long* items) {
  int rc = check_channels(dtype, C, who);
  if (rc) return rc;
  KD6D_CHECK_ARG(B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "%s: B=%d H=%d W=%d (H, W must be even)", who, B, H, W);
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  *items = (long long)B * (H / 2) * (W / 2) * (C / eg);
  KD6D_CHECK_ARG(*items * 4 < (1ll << 31), "%s: %lld granules exceed the 32-bit item index", who, *items * 4);
  return KD6D_OK;
}

extern "C" int kd6d_bn_pool_train_fwd(int dtype, int x_f32, const void* x, void* y, int B, int H, int W, int C,
                                      const kd6d_acc* sum_acc, const kd6d_acc* sumsq_acc, const float* gamma,
                                      const float* beta, float eps, float momentum, float* running_mean,
                                      float* running_var, float* save_mean, float* save_invstd, int act,
                                      void* stream) {
  const acc_t* sum = reinterpret_cast<const acc_t*>(sum_acc);
  const acc_t* sumsq = reinterpret_cast<const acc_t*>(sumsq_acc);
  long long items = 0;
  int rc = check_pool(dtype, B, H, W, C, "kd6d_bn_pool_train_fwd", &items);
  if (rc) return rc;
  KD6D_CHECK_ARG(x && y && sum && sumsq && gamma && beta, "kd6d_bn_pool_train_fwd: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long rows = (long long)B * H * W;
  const float inv_rows = 1.f / (float)rows;
  const float unbias = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
  const int nb = grid_for(items);
  DISPATCH_TTX(dtype, x_f32,
               hipLaunchKernelGGL((bn_pool_fwd_kernel<T_, TX_>), dim3(nb), dim3(kThreads), 0, st, (const TX_*)x,
                                   (T_*)y, (int)items, H, W, C, inv_rows, sum, sumsq, gamma, beta, eps, momentum,
                                   unbias, running_mean, running_var, save_mean, save_invstd, act));
  KD6D_CHECK_LAUNCH("kd6d_bn_pool_train_fwd");
  return KD6D_OK;
}

extern "C" int kd6d_bn_pool_train_bwd(int dtype, int x_f32, const void* x, const void* dy, void* dx, int B, int H,
                                      int W, int C, const float* mean, const float* invstd, const float* gamma,
                                      const float* beta, int act, kd6d_acc* sum_dy_acc, kd6d_acc* sum_dy_xhat_acc,
                                      unsigned int* counter, float* dgamma, float* dbeta, int replicas,
                                      void* stream) {
  acc_t* sum_dy = reinterpret_cast<acc_t*>(sum_dy_acc);
  acc_t* sum_dy_xhat = reinterpret_cast<acc_t*>(sum_dy_xhat_acc);
  long long items = 0;
  int rc = check_pool(dtype, B, H, W, C, "kd6d_bn_pool_train_bwd", &items);
  if (rc) return rc;
  KD6D_CHECK_ARG(replicas >= 1 && replicas <= 64, "kd6d_bn_pool_train_bwd: replicas=%d outside [1,64]", replicas);
  KD6D_CHECK_ARG(x && dy && dx && mean && invstd && gamma && beta && sum_dy && sum_dy_xhat,
                 "kd6d_bn_pool_train_bwd: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float inv_rows = 1.f / (float)((long long)B * H * W);
  long long nbr = (items + kThreads * 2 - 1) / (kThreads * 2);      // 8 input granules per thread, as the unpooled pass
  if (nbr > kBnBwdReduceCap) nbr = kBnBwdReduceCap;
  if (nbr < 1) nbr = 1;
  const int nba = grid_for(items);
  const size_t lds = kFlushLdsBytes;
  int cap = 0;
  DISPATCH_TTX(dtype, x_f32, cap = (bn_onepass_capacity<T_, TX_>(true)));
  cap = cap * 3 / 4;                                      // the whole grid must be resident for the barrier
  if (cap > kBnOnepassBlocks) cap = kBnOnepassBlocks;
  if (counter && bn_onepass() && cap > 0 && items <= (long long)cap * kThreads * kPoolHold &&
      items * 4 <= bn_onepass_max_granules()) {
    long long nb1 = (items + kThreads - 1) / kThreads;
    if (nb1 > cap) nb1 = cap;
    const int per = (int)((items + nb1 * kThreads - 1) / (nb1 * kThreads));
    DISPATCH_TTX(dtype, x_f32,
                 hipLaunchKernelGGL((bn_pool_bwd_onepass_kernel<T_, TX_>), dim3((int)nb1), dim3(kThreads), lds, st,
                                    (const TX_*)x, (const T_*)dy, (T_*)dx, (int)items, per, H, W, C, inv_rows, mean,
                                    invstd, gamma, beta, act, sum_dy, sum_dy_xhat, counter, dgamma, dbeta, replicas,
                                    kd6d_ctx_timeouts_ptr()));
    KD6D_CHECK_LAUNCH("kd6d_bn_pool_train_bwd");
    return KD6D_OK;
  }
  DISPATCH_TTX(dtype, x_f32, {
    hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<T_, TX_>), dim3((int)nbr), dim3(kThreads), lds, st,
                       (const TX_*)x, (const T_*)dy, (int)items, H, W, C, mean, invstd, gamma, beta, act, sum_dy,
                       sum_dy_xhat, replicas);
    hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<T_, TX_>), dim3(nba), dim3(kThreads), lds, st, (const TX_*)x,
                       (const T_*)dy, (T_*)dx, (int)items, H, W, C, inv_rows, mean, invstd, gamma, beta, act,
                       sum_dy, sum_dy_xhat, dgamma, dbeta, replicas);
  });
  KD6D_CHECK_LAUNCH("kd6d_bn_pool_train_bwd");
  return KD6D_OK;
}

extern "C" int kd6d_gn_relu_fwd(int dtype, int x_f32, const void* x, void* y, const int32_t* level_hw_host,
                                int nseg, int batch, int C, int groups, const float* gamma,
                                const float* beta, float eps, kd6d_acc* stats_acc, int flags, void* stream) {
  int rc = check_channels(dtype, C, "kd6d_gn_relu_fwd");
  if (rc) return rc;
  acc_t* stats = reinterpret_cast<acc_t*>(stats_acc);
  GnGeom gm;
  KD6D_CHECK_ARG(level_hw_host && fill_gn(level_hw_host, nseg, batch, C, groups, &gm),
                 "kd6d_gn_relu_fwd: bad level table / groups");
  KD6D_CHECK_ARG(x && y && gamma && beta && stats, "kd6d_gn_relu_fwd: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  const long long ngran = gn_rows(gm) * (C / eg);
  const int nb = grid_for((ngran + 3) / 4);
  KD6D_CHECK_ARG((C / groups) * 2 >= eg, "kd6d_gn_relu_fwd: C/groups=%d too small for %d-wide granules", C / groups, eg);
  const bool ready = flags & KD6D_GN_STATS_READY;
  if (!ready && !(flags & KD6D_GN_WS_ZEROED) &&
      hipMemsetAsync(stats, 0, sizeof(kd6d_acc) * 2 * (size_t)nseg * batch * groups, st) != hipSuccess) {
    kd6d_set_error("kd6d_gn_relu_fwd: memset failed");
    return KD6D_ERR_LAUNCH;
  }
  DISPATCH_TTX(dtype, x_f32, {
    if (!ready)
      hipLaunchKernelGGL((gn_stats_kernel<T_, TX_>), dim3(gm.nblk), dim3(kThreads), 0, st, (const TX_*)x, gm, stats);
    hipLaunchKernelGGL((gn_relu_fwd_kernel<T_, TX_>), dim3(nb), dim3(kThreads), 0, st, (const TX_*)x, (T_*)y, gm,
                       ngran, eps, stats, gamma, beta);
  });
  KD6D_CHECK_LAUNCH("kd6d_gn_relu_fwd");
  return KD6D_OK;
}

extern "C" int kd6d_gn_relu_bwd(int dtype, int x_f32, const void* x, const void* dz, void* dx,
                                const int32_t* level_hw_host, int nseg, int batch, int C, int groups,
                                const float* gamma, const float* beta, float eps, const kd6d_acc* stats_acc,
                                kd6d_acc* gsum_ws_acc, int64_t* dgamma_acc, int64_t* dbeta_acc, int64_t acc_hi_stride,
                                int flags, void* stream) {
  int rc = check_channels(dtype, C, "kd6d_gn_relu_bwd");
  if (rc) return rc;
  const acc_t* stats = reinterpret_cast<const acc_t*>(stats_acc);
  acc_t* gsum_ws = reinterpret_cast<acc_t*>(gsum_ws_acc);
  acc_t* dgamma = reinterpret_cast<acc_t*>(dgamma_acc);
  acc_t* dbeta = reinterpret_cast<acc_t*>(dbeta_acc);
  const long long acc_hi = (long long)acc_hi_stride;
  KD6D_CHECK_ARG((!dgamma && !dbeta) || acc_hi_stride != 0, "kd6d_gn_relu_bwd: acc_hi_stride = 0 with gradient accumulators");
  GnGeom gm;
  KD6D_CHECK_ARG(level_hw_host && fill_gn(level_hw_host, nseg, batch, C, groups, &gm),
                 "kd6d_gn_relu_bwd: bad level table / groups");
  KD6D_CHECK_ARG(x && dz && dx && gamma && beta && stats && gsum_ws, "kd6d_gn_relu_bwd: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  const long long ngran = gn_rows(gm) * (C / eg);
  const int nb = grid_for((ngran + 3) / 4);
  const size_t lds = kFlushLdsBytes;
  KD6D_CHECK_ARG((C / groups) * 2 >= eg, "kd6d_gn_relu_bwd: C/groups=%d too small for %d-wide granules", C / groups, eg);
  if (!(flags & KD6D_GN_WS_ZEROED) &&
      hipMemsetAsync(gsum_ws, 0, sizeof(kd6d_acc) * 2 * (size_t)nseg * batch * groups, st) != hipSuccess) {
    kd6d_set_error("kd6d_gn_relu_bwd: memset failed");
    return KD6D_ERR_LAUNCH;
  }
  if (gn_onepass()) {
    // row chunks small enough for the registers of one workgroup; the barrier counters sit behind the sums
    const int cgs = C / eg;
    int chunk = gn_chunk_rows();
    if (chunk * cgs > kGnHold * kThreads) chunk = kGnHold * kThreads / cgs;
    GnGeom g1;
    KD6D_CHECK_ARG(chunk >= 1 && fill_gn(level_hw_host, nseg, batch, C, groups, &g1, chunk), "kd6d_gn_relu_bwd: geometry");
    int gcap = 0, gmax = 1;
    DISPATCH_TTX(dtype, x_f32, gcap = (gn_onepass_capacity<T_, TX_>()));
    for (int s = 0; s < nseg; ++s) gmax = g1.cps[s] > gmax ? g1.cps[s] : gmax;
    KD6D_CHECK_ARG(gcap >= 2 * gmax, "kd6d_gn_relu_bwd: a level of %d row chunks does not fit the %d resident workgroups "
                   "(set option gn.onepass = 0)", gmax, gcap);
    unsigned int* counters = reinterpret_cast<unsigned int*>(gsum_ws + 4 * (size_t)nseg * batch * groups);
    if (!(flags & KD6D_GN_WS_ZEROED) &&
        hipMemsetAsync(counters, 0, sizeof(unsigned int) * (size_t)nseg * batch, st) != hipSuccess) {
      kd6d_set_error("kd6d_gn_relu_bwd: memset failed");
      return KD6D_ERR_LAUNCH;
    }
    DISPATCH_TTX(dtype, x_f32,
                 hipLaunchKernelGGL((gn_relu_bwd_onepass_kernel<T_, TX_>), dim3(g1.nblk), dim3(kThreads), lds, st,
                                    (const TX_*)x, (const T_*)dz, (T_*)dx, g1, eps, stats, gamma, beta, gsum_ws,
                                    counters, dgamma, dbeta, acc_hi));
    KD6D_CHECK_LAUNCH("kd6d_gn_relu_bwd");
    return KD6D_OK;
  }
  DISPATCH_TTX(dtype, x_f32, {
    hipLaunchKernelGGL((gn_relu_bwd_reduce_kernel<T_, TX_>), dim3(gm.nblk), dim3(kThreads), lds, st,
                       (const TX_*)x, (const T_*)dz, gm, eps, stats, gamma, beta, gsum_ws, dgamma, dbeta, acc_hi);
    hipLaunchKernelGGL((gn_relu_bwd_apply_kernel<T_, TX_>), dim3(nb), dim3(kThreads), 0, st, (const TX_*)x,
                       (const T_*)dz, (T_*)dx, gm, ngran, eps, stats, gsum_ws, gamma, beta);
  });
  KD6D_CHECK_LAUNCH("kd6d_gn_relu_bwd");
  return KD6D_OK;
}

extern "C" int kd6d_gn_relu_bwd_pair(int dtype, int x_f32, const kd6d_gn_item* a, const kd6d_gn_item* b,
                                     const int32_t* level_hw_host, int nseg, int batch, int C, int groups, float eps,
                                     int64_t acc_hi_stride, int flags, void* stream) {
  KD6D_CHECK_ARG(a && b, "kd6d_gn_relu_bwd_pair: null item");
  // the one-launch form exists for the in-kernel-barrier backward only; everything else goes out one by one
  int chunk = 0;
  GnGeom g1;
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  bool pair = gn_onepass() && check_channels(dtype, C, "kd6d_gn_relu_bwd_pair") == KD6D_OK && level_hw_host;
  if (pair) {
    const int cgs = C / eg;
    chunk = gn_chunk_rows();
    if (chunk * cgs > kGnHold * kThreads) chunk = kGnHold * kThreads / cgs;
    pair = chunk >= 1 && (C / groups) * 2 >= eg && fill_gn(level_hw_host, nseg, batch, C, groups, &g1, chunk);
  }
  if (pair) {
    int gcap = 0, gmax = 1;
    DISPATCH_TTX(dtype, x_f32, gcap = (gn_onepass_capacity<T_, TX_>()));
    for (int s = 0; s < nseg; ++s) gmax = g1.cps[s] > gmax ? g1.cps[s] : gmax;
    pair = gcap >= 4 * gmax;                 // two launches' worth of siblings resident at once
  }
  const kd6d_gn_item* it[2] = {a, b};
  if (!pair) {
    for (int i = 0; i < 2; ++i) {
      const int rc = kd6d_gn_relu_bwd(dtype, x_f32, it[i]->x, it[i]->dz, it[i]->dx, level_hw_host, nseg, batch, C, groups,
                                      it[i]->gamma, it[i]->beta, eps, it[i]->stats, it[i]->gsum_ws, it[i]->dgamma,
                                      it[i]->dbeta, acc_hi_stride, flags, stream);
      if (rc) return rc;
    }
    return KD6D_OK;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  GnBwdSet sets[2];
  for (int i = 0; i < 2; ++i) {
    KD6D_CHECK_ARG(it[i]->x && it[i]->dz && it[i]->dx && it[i]->gamma && it[i]->beta && it[i]->stats && it[i]->gsum_ws,
                   "kd6d_gn_relu_bwd_pair: null pointer in item %d", i);
    KD6D_CHECK_ARG((!it[i]->dgamma && !it[i]->dbeta) || acc_hi_stride != 0,
                   "kd6d_gn_relu_bwd_pair: acc_hi_stride = 0 with gradient accumulators");
    unsigned int* counters = reinterpret_cast<unsigned int*>(it[i]->gsum_ws + 2 * (size_t)nseg * batch * groups);   // kd6d_acc units
    if (!(flags & KD6D_GN_WS_ZEROED) &&
        hipMemsetAsync(it[i]->gsum_ws, 0, sizeof(kd6d_acc) * 2 * (size_t)nseg * batch * groups + sizeof(unsigned int) * (size_t)nseg * batch,
                       st) != hipSuccess) {
      kd6d_set_error("kd6d_gn_relu_bwd_pair: memset failed");
      return KD6D_ERR_LAUNCH;
    }
    sets[i] = GnBwdSet{it[i]->x, it[i]->dz, it[i]->dx, reinterpret_cast<const acc_t*>(it[i]->stats), it[i]->gamma,
                       it[i]->beta, reinterpret_cast<acc_t*>(it[i]->gsum_ws), counters,
                       reinterpret_cast<acc_t*>(it[i]->dgamma), reinterpret_cast<acc_t*>(it[i]->dbeta)};
  }
  const size_t lds = kFlushLdsBytes;
  DISPATCH_TTX(dtype, x_f32,
               hipLaunchKernelGGL((gn_relu_bwd_onepass_pair_kernel<T_, TX_>), dim3(2 * g1.nblk), dim3(kThreads), lds, st,
                                  sets[0], sets[1], g1, eps, (long long)acc_hi_stride));
  KD6D_CHECK_LAUNCH("kd6d_gn_relu_bwd_pair");
  return KD6D_OK;
}

extern "C" int kd6d_maxpool2_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C,
                                 void* stream) {
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "kd6d_maxpool2_fwd: bad dtype");
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(x && y && B > 0 && H >= 2 && W >= 2 && C % eg == 0, "kd6d_maxpool2_fwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long total = (long long)B * (H / 2) * (W / 2) * (C / eg);
  const int nb = grid_for(total);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16_t>, dim3(nb), dim3(kThreads), 0, st,
                                (const bf16_t*)x, (bf16_t*)y, B, H, W, C),
             hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(nb), dim3(kThreads), 0, st,
                                (const float*)x, (float*)y, B, H, W, C));
  KD6D_CHECK_LAUNCH("kd6d_maxpool2_fwd");
  return KD6D_OK;
}

extern "C" int kd6d_maxpool2_bwd(int dtype, const void* x, const void* dy, void* dx, int B, int H,
                                 int W, int C, int accumulate, void* stream) {
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "kd6d_maxpool2_bwd: bad dtype");
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(x && dy && dx && B > 0 && H >= 2 && W >= 2 && C % eg == 0 && H % 2 == 0 && W % 2 == 0,
                 "kd6d_maxpool2_bwd: bad arguments (H, W must be even)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long total = (long long)B * (H / 2) * (W / 2) * (C / eg);
  const int nb = grid_for(total);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(nb), dim3(kThreads), 0, st,
                                (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C, accumulate),
             hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(nb), dim3(kThreads), 0, st,
                                (const float*)x, (const float*)dy, (float*)dx, B, H, W, C, accumulate));
  KD6D_CHECK_LAUNCH("kd6d_maxpool2_bwd");
  return KD6D_OK;
}

extern "C" int kd6d_upsample2_add(int dtype, const void* fine, const void* coarse, void* out, int B,
                                  int H, int W, int C, void* stream) {
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "kd6d_upsample2_add: bad dtype");
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(fine && coarse && out && B > 0 && H % 2 == 0 && W % 2 == 0 && C % eg == 0,
                 "kd6d_upsample2_add: bad arguments (H, W must be even)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long total = (long long)B * H * W * (C / eg);
  const int nb = grid_for(total);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(upsample2_add_kernel<bf16_t>, dim3(nb), dim3(kThreads), 0, st,
                                (const bf16_t*)fine, (const bf16_t*)coarse, (bf16_t*)out, B, H, W, C),
             hipLaunchKernelGGL(upsample2_add_kernel<float>, dim3(nb), dim3(kThreads), 0, st,
                                (const float*)fine, (const float*)coarse, (float*)out, B, H, W, C));
  KD6D_CHECK_LAUNCH("kd6d_upsample2_add");
  return KD6D_OK;
}

extern "C" int kd6d_sumpool2(int dtype, const void* dfine, void* dcoarse, int B, int H, int W, int C,
                             int accumulate, void* stream) {
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "kd6d_sumpool2: bad dtype");
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(dfine && dcoarse && B > 0 && H % 2 == 0 && W % 2 == 0 && C % eg == 0,
                 "kd6d_sumpool2: bad arguments (H, W must be even)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long total = (long long)B * (H / 2) * (W / 2) * (C / eg);
  const int nb = grid_for(total);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(sumpool2_kernel<bf16_t>, dim3(nb), dim3(kThreads), 0, st,
                                (const bf16_t*)dfine, (bf16_t*)dcoarse, B, H, W, C, accumulate),
             hipLaunchKernelGGL(sumpool2_kernel<float>, dim3(nb), dim3(kThreads), 0, st,
                                (const float*)dfine, (float*)dcoarse, B, H, W, C, accumulate));
  KD6D_CHECK_LAUNCH("kd6d_sumpool2");
  return KD6D_OK;
}

extern "C" int kd6d_eltwise(int dtype, int mode, const void* x, const void* dy, void* y,
                            int64_t n_elems, void* stream) {
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "kd6d_eltwise: bad dtype");
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(x && y && n_elems > 0 && n_elems % eg == 0 && mode >= 0 && mode <= 2 &&
                     (mode == 0 || dy),
                 "kd6d_eltwise: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long ngran = n_elems / eg;
  const int nb = grid_for(ngran);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(eltwise_kernel<bf16_t>, dim3(nb), dim3(kThreads), 0, st,
                                (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)y, ngran, mode),
             hipLaunchKernelGGL(eltwise_kernel<float>, dim3(nb), dim3(kThreads), 0, st,
                                (const float*)x, (const float*)dy, (float*)y, ngran, mode));
  KD6D_CHECK_LAUNCH("kd6d_eltwise");
  return KD6D_OK;
}

extern "C" int kd6d_image_to_nhwc(int dtype, const float* img_nchw, void* out, int B, int Cimg, int H,
                                  int W, int Cpad, void* stream) {
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "kd6d_image_to_nhwc: bad dtype");
  KD6D_CHECK_ARG(img_nchw && out && B > 0 && Cimg > 0 && Cpad >= Cimg && H > 0 && W > 0,
                 "kd6d_image_to_nhwc: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nb = grid_for((long long)B * H * W);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(image_to_nhwc_kernel<bf16_t>, dim3(nb), dim3(kThreads), 0, st, img_nchw,
                                (bf16_t*)out, B, Cimg, H, W, Cpad),
             hipLaunchKernelGGL(image_to_nhwc_kernel<float>, dim3(nb), dim3(kThreads), 0, st, img_nchw,
                                (float*)out, B, Cimg, H, W, Cpad));
  KD6D_CHECK_LAUNCH("kd6d_image_to_nhwc");
  return KD6D_OK;
}
