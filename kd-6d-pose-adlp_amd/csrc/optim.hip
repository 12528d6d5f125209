// Fused optimiser side of the KD step over ONE flat fp32 parameter/gradient buffer.
//
// Replaces train_kd.py:138-139: nn.utils.clip_grad_norm_(model.parameters(), 1.0) followed by
// torch.optim.AdamW(lr, weight_decay=1e-4, eps=1e-8).step() (libs/train_libs.py:119): the
// reference walks 150 tensors with several launches each; here it is two launches
// (sum of squares, then clip + AdamW + optional bf16 shadow refresh), HBM-bound:
// 16 B read + 12 B (+2 B) written per parameter.
#include "kd6d_common.h"
#include "kd6d_det.h"

namespace {

constexpr int kT = 256;

// partial[b] = sum of squares of workgroup b's share (fixed order inside the workgroup); workgroup 0 clears the unused
// slots.  No atomics, no finishing pass: kd6d_clip_adamw adds the KD6D_SUMSQ_PARTS partials in a fixed order itself.
__global__ __launch_bounds__(kT) void sumsq_kernel(const float* __restrict__ x, long long n,
                                                   float* __restrict__ partial) {
  __shared__ float s_part[kT / 64];
  float acc = 0.f;
  const long long n4 = n >> 2;
  const f32x4_t* x4 = reinterpret_cast<const f32x4_t*>(x);
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n4; i += (long long)gridDim.x * kT) {
    const f32x4_t v = x4[i];
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0) {
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += kT) acc += x[i] * x[i];
    for (int i = gridDim.x + threadIdx.x; i < KD6D_SUMSQ_PARTS; i += kT) partial[i] = 0.f;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < kT / 64; ++w) s += s_part[w];
    partial[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kT) void clip_adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        long long n, const float* __restrict__ gnorm_parts,
                                                        float* __restrict__ gnorm_sq_out,
                                                        float max_norm, float lr, float beta1, float beta2,
                                                        float eps, float wd, float bc1, float bc2_sqrt,
                                                        const float* __restrict__ hyper,
                                                        bf16_t* __restrict__ shadow) {
  if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; }   // graph-replay path: device-resident schedule
  float coef = 1.f;
  if (gnorm_parts) {
    // the squared gradient norm from kd6d_sumsq's partials: every wave adds the same KD6D_SUMSQ_PARTS numbers in the same
    // order (two per lane, then the butterfly), so every thread of the launch scales by the same bits
    static_assert(KD6D_SUMSQ_PARTS == 128, "two partials per lane");
    const int lane = threadIdx.x & 63;
    const float g2 = wave_sum(gnorm_parts[lane] + gnorm_parts[lane + 64]);
    if (gnorm_sq_out && blockIdx.x == 0 && threadIdx.x == 0) gnorm_sq_out[0] = g2;
    if (max_norm > 0.f) {
      const float c = max_norm / (sqrtf(g2) + 1e-6f);
      coef = c < 1.f ? c : 1.f;
    }
  }
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n; i += (long long)gridDim.x * kT) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * mi / denom;
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (shadow) shadow[i] = (bf16_t)pi;
  }
}

__global__ void set_hyper_kernel(float* __restrict__ hyper, float lr, float bc1, float bc2_sqrt) {
  hyper[0] = lr; hyper[1] = bc1; hyper[2] = bc2_sqrt; hyper[3] = 0.f;
}

__global__ __launch_bounds__(kT) void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y,
                                                       long long n) {
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n; i += (long long)gridDim.x * kT)
    y[i] = (bf16_t)x[i];
}

// value of n interleaved accumulators -> fp32 (tests, debugging, stand-alone callers of the statistics kernels)
template <int E>
__global__ __launch_bounds__(kT) void acc_read_kernel(long long* __restrict__ acc, long long n, float* __restrict__ out,
                                                      int accumulate, int clear) {
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n; i += (long long)gridDim.x * kT) {
    const float v = kd6d_detail::det_value<E>(acc[2 * i], acc[2 * i + 1]);
    out[i] = accumulate ? out[i] + v : v;
    if (clear) { acc[2 * i] = 0; acc[2 * i + 1] = 0; }
  }
}

// End of the reverse sweep: every gradient that was summed across workgroups -> fp32 gradients, in one launch.
// desc: int64 quintuples {first element, element count, first workgroup, parts, slab address}.
//   parts == 0: the elements' PLANAR fixed-point accumulators (cleared for the next step); 1024 elements per workgroup.
//   parts >= 1: the partial images of a per-layer weight gradient, slab[part][element].  A workgroup owns 1024 / PG
//   elements, PG = kd6d_resolve_part_groups(parts) threads per element: thread (element, g) adds parts g, g + PG, ... into
//   four interleaved running sums (four loads in flight), the PG partial sums meet in LDS and are added in g order -- a
//   FIXED association (bitwise reproducible) with at most 4 dependent memory round trips per thread (a narrow layer split
//   512 ways summed by one thread per element took 0.5 ms: 512 exposed round trips; 32 in a row still 60 us).
__host__ __device__ inline int resolve_part_groups(int parts) {
  int pg = 1;
  while (pg < 32 && parts > 16 * pg) pg *= 2;
  return pg;
}

__global__ __launch_bounds__(kT) void grad_acc_resolve_kernel(const long long* __restrict__ desc, int n_regions,
                                                              long long* __restrict__ acc, long long hi_off,
                                                              float* __restrict__ grads) {
  __shared__ float part_sum[1024];
  int r = 0;
  for (int i = 1; i < n_regions; ++i)
    if ((long long)blockIdx.x >= desc[i * 5 + 2]) r = i;
  const long long first = desc[r * 5], count = desc[r * 5 + 1];
  const int parts = (int)desc[r * 5 + 3];
  if (parts == 0) {
    const long long base = ((long long)blockIdx.x - desc[r * 5 + 2]) * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long i = base + k * kT + threadIdx.x;
      if (i >= count) continue;
      long long* lo = acc + first + i;
      const long long l = lo[0], h = lo[hi_off];
      if (l | h) {
        grads[first + i] += kd6d_detail::det_value<KD6D_DET_GRAD>(l, h);
        lo[0] = 0;
        if (h) lo[hi_off] = 0;
      }
    }
    return;
  }
  const float* __restrict__ slab = reinterpret_cast<const float*>(desc[r * 5 + 4]);
  const int pg = resolve_part_groups(parts);
  const int epb = 1024 / pg;                               // elements per workgroup
  const long long base = ((long long)blockIdx.x - desc[r * 5 + 2]) * epb;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int slot = k * kT + threadIdx.x;                 // (g, j): consecutive threads -> consecutive elements
    const int g = slot / epb, j = slot - g * epb;
    const long long i = base + j;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (i < count) {
      for (int s2 = g; s2 < parts; s2 += 4 * pg) {
        const float a0 = slab[(size_t)s2 * count + i];
        const float a1 = s2 + pg < parts ? slab[(size_t)(s2 + pg) * count + i] : 0.f;
        const float a2 = s2 + 2 * pg < parts ? slab[(size_t)(s2 + 2 * pg) * count + i] : 0.f;
        const float a3 = s2 + 3 * pg < parts ? slab[(size_t)(s2 + 3 * pg) * count + i] : 0.f;
        t0 += a0; t1 += a1; t2 += a2; t3 += a3;
      }
    }
    part_sum[slot] = (t0 + t1) + (t2 + t3);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < epb; j += kT) {
    const long long i = base + j;
    if (i >= count) continue;
    float t = part_sum[j];
    for (int g = 1; g < pg; ++g) t += part_sum[g * epb + j];
    grads[first + i] += t;
  }
}

int blocks_for(long long n) {
  long long b = (n + kT * 4 - 1) / (kT * 4);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int kd6d_acc_read(kd6d_acc* acc, int64_t n, int kind, float* out, int accumulate, int clear, void* stream) {
  KD6D_CHECK_ARG(acc && out && n > 0 && (kind == KD6D_ACC_ACT || kind == KD6D_ACC_GRAD), "kd6d_acc_read: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  long long* words = reinterpret_cast<long long*>(acc);
  const int nb = blocks_for(n * 4);
  if (kind == KD6D_ACC_ACT)
    hipLaunchKernelGGL(acc_read_kernel<KD6D_DET_ACT>, dim3(nb), dim3(kT), 0, st, words, (long long)n, out, accumulate, clear);
  else
    hipLaunchKernelGGL(acc_read_kernel<KD6D_DET_GRAD>, dim3(nb), dim3(kT), 0, st, words, (long long)n, out, accumulate, clear);
  KD6D_CHECK_LAUNCH("kd6d_acc_read");
  return KD6D_OK;
}

extern "C" int kd6d_grad_acc_resolve_part_groups(int parts) { return parts <= 0 ? 1 : resolve_part_groups(parts); }

extern "C" int kd6d_grad_acc_resolve(const int64_t* desc_dev, int n_regions, int total_blocks, int64_t* acc,
                                     int64_t acc_hi_stride, float* grads, void* stream) {
  KD6D_CHECK_ARG(desc_dev && n_regions > 0 && total_blocks > 0 && acc && acc_hi_stride != 0 && grads,
                 "kd6d_grad_acc_resolve: bad arguments");
  static_assert(KD6D_ACC_ACT == KD6D_DET_ACT && KD6D_ACC_GRAD == KD6D_DET_GRAD, "kd6d.h / kd6d_det.h classes");
  hipLaunchKernelGGL(grad_acc_resolve_kernel, dim3(total_blocks), dim3(kT), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const long long*>(desc_dev), n_regions, reinterpret_cast<long long*>(acc),
                     (long long)acc_hi_stride, grads);
  KD6D_CHECK_LAUNCH("kd6d_grad_acc_resolve");
  return KD6D_OK;
}

extern "C" int kd6d_sumsq(const float* x, int64_t n, float* partials, void* stream) {
  KD6D_CHECK_ARG(x && partials && n > 0, "kd6d_sumsq: bad arguments");
  KD6D_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0, "kd6d_sumsq: x must be 16-byte aligned");
  // every workgroup ends with ONE atomic on the same address, and those retire serially (~13-27 ns each): 2048
  // workgroups spent 28 us on a 9-MB gradient bucket, 128 read it in a third of that
  int nb = blocks_for(n);
  if (nb > KD6D_SUMSQ_PARTS) nb = KD6D_SUMSQ_PARTS;
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(kT), 0, reinterpret_cast<hipStream_t>(stream), x,
                     (long long)n, partials);
  KD6D_CHECK_LAUNCH("kd6d_sumsq");
  return KD6D_OK;
}

extern "C" int kd6d_clip_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                               const float* gnorm_partials, float* gnorm_sq_out, double max_norm, double lr, double beta1,
                               double beta2, double eps, double weight_decay, int64_t step, const float* hyper_dev,
                               void* bf16_shadow, void* stream) {
  KD6D_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && (step >= 1 || hyper_dev),
                 "kd6d_clip_adamw: bad arguments");
  // bias corrections in double, like torch's python-float arithmetic
  const float bc1 = hyper_dev ? 1.f : (float)(1.0 - pow(beta1, (double)step));
  const float bc2_sqrt = hyper_dev ? 1.f : (float)sqrt(1.0 - pow(beta2, (double)step));
  hipLaunchKernelGGL(clip_adamw_kernel, dim3(blocks_for(n)), dim3(kT), 0, reinterpret_cast<hipStream_t>(stream),
                     param, grad, exp_avg, exp_avg_sq, (long long)n, gnorm_partials, gnorm_sq_out, (float)max_norm, (float)lr,
                     (float)beta1, (float)beta2, (float)eps, (float)weight_decay, bc1, bc2_sqrt, hyper_dev,
                     reinterpret_cast<bf16_t*>(bf16_shadow));
  KD6D_CHECK_LAUNCH("kd6d_clip_adamw");
  return KD6D_OK;
}

extern "C" int kd6d_set_hyper(float* hyper_dev, double lr, double beta1, double beta2, int64_t step, void* stream) {
  KD6D_CHECK_ARG(hyper_dev && step >= 1, "kd6d_set_hyper: bad arguments");
  const float bc1 = (float)(1.0 - pow(beta1, (double)step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  hipLaunchKernelGGL(set_hyper_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), hyper_dev,
                     (float)lr, bc1, bc2_sqrt);
  KD6D_CHECK_LAUNCH("kd6d_set_hyper");
  return KD6D_OK;
}

extern "C" int kd6d_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  KD6D_CHECK_ARG(x && y && n > 0, "kd6d_cast_f32_to_bf16: bad arguments");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks_for(n)), dim3(kT), 0, reinterpret_cast<hipStream_t>(stream), x,
                     reinterpret_cast<bf16_t*>(y), (long long)n);
  KD6D_CHECK_LAUNCH("kd6d_cast_f32_to_bf16");
  return KD6D_OK;
}
