// Implicit-GEMM convolution for gfx950 (forward, data-gradient, weight-gradient).
//
// Replaces the torch conv2d calls of the reference hot path:
//   backbone/common.py:316-324 (ConvBlock), backbone/darknet53.py:54-58 (DarkUnit
//   residual), models/model.py:64-83 / 97-103 (FPN convs), models/model.py:438-451
//   (PoseHead towers + cls_logits + pose_pred, run over ALL pyramid levels in one
//   launch through the segment table).
//
// Layout: activations NHWC (rows = pixels), weights KRSC w[cout][ky][kx][cin].
// GEMM view (forward):  D[channel n][pixel m] = sum_k W[n][k] * im2col(X)[m][k],
// k = (ky, kx, ci).  The MFMA "A" operand is the weight tile, the "B" operand the
// gathered pixel tile, so each lane ends up with 4 CONSECUTIVE CHANNELS of one
// pixel (8-B bf16 / 16-B fp32 stores, per-channel epilogue parameters as float4).
//
// LDS image: every tile row is 128 B = 8 granules of 16 B (64 bf16 / 32 fp32 along
// k), granule g of row r stored at g ^ (r & 7)  (conflict-free ds_read_b128 for the
// 16x16 fragment pattern: lanes 0-15 -> rows, lane>>4 -> granule).
//   bf16: v_mfma_f32_16x16x32_bf16, one granule per lane per 32-deep chunk.
//   fp32: v_mfma_f32_16x16x4_f32 x8 on a 32-deep chunk; lane group q holds
//         k = 8q..8q+7 (two granules) and MFMA j contracts {8q+j}: exact fp32.
#include "conv_common.h"

using namespace kd6d_detail;

namespace {

// XF = true (forward only): the gather source is the PREVIOUS ConvBlock's fp32 conv output and its train-mode BatchNorm
// + activation (backbone/common.py:316-324) is applied while the pixel operand is loaded -- scale / shift per channel
// from the batch sums that block's epilogue accumulated, the same fma + LeakyReLU as bn_apply_fwd_kernel, one rounding to
// T -- so the separate normalise launch and its (rows, C) round trip go without any wait inside a kernel.  The
// activation tensor the weight gradient needs is written on the way (centre tap, channel tile 0: every input pixel
// exactly once); workgroup 0 publishes save_mean / save_invstd and updates the running statistics.
// Waves per SIMD the register allocation must leave room for (second __launch_bounds__ argument on AMD): the tiles whose
// LDS lets 2 / 5 workgroups share a CU sit a few registers under that occupancy step, and without the hint the compiler
// spreads into AGPRs past it (round 4: the statistics epilogue grew by 8 VGPRs and 128 x 128 fell from 2 workgroups per CU
// to 1: 33 -> 52 us per launch).  0 = no constraint.
template <typename T, int BP, int BC, bool XF, bool NORM>
constexpr int igemm_min_waves() {
  if (XF || NORM) return 1;
  if (BP == 128 && BC == 128) return 2;
  if (BP == 64 && BC == 64 && sizeof(T) == 2) return 5;
  return 1;
}

template <typename T, int BP, int BC, int WP, int WC, int MODE, bool XF = false, bool NORM = false>
__global__ __launch_bounds__(256, (igemm_min_waves<T, BP, BC, XF, NORM>())) void conv_igemm_kernel(const ConvParams p) {
  constexpr int EG = Granule<T>::N;
  constexpr int BK = 8 * EG;
  constexpr int PI = BP / WP / 16;
  constexpr int CI = BC / WC / 16;
  constexpr int PR = (BP + 31) / 32;
  constexpr int CR = (BC + 31) / 32;
  constexpr int TILE_BYTES = (BP + BC) * 128;
  static_assert(WP * WC == 4, "4 waves");
  static_assert(PI >= 1 && CI >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wp = wave % WP;
  const int wc = wave / WP;

  const int wg = tile_of_workgroup<NORM || XF>(p, blockIdx.x, gridDim.x);
  const int tile_c = p.p_fastest ? wg / p.n_ptiles : wg % p.n_ctiles;
  const int tile_p = p.p_fastest ? wg % p.n_ptiles : wg / p.n_ctiles;
  const int m0 = tile_p * BP;
  const int n0 = tile_c * BC;

  // ---- loader state -------------------------------------------------------
  const int gcol = tid & 7;
  const int lrow = tid >> 3;  // 0..31
  int ry[PR], rx[PR], rbase[PR], rhw[PR];
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    const int prow = lrow + 32 * i;
    RowInfo ri = decode_row(p, m0 + prow);
    if (prow >= BP) { ri.src_h = 0; ri.src_w = 0; }
    if (MODE == MODE_FWD) {
      ry[i] = ri.y * p.stride - p.pad;
      rx[i] = ri.x * p.stride - p.pad;
    } else {
      ry[i] = ri.y + p.pad;
      rx[i] = ri.x + p.pad;
    }
    rbase[i] = ri.src_base;
    rhw[i] = (ri.src_h << 16) | ri.src_w;
  }

  const T* __restrict__ src = reinterpret_cast<const T*>(p.src);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(p.wgt);

  u32x4_t preg[PR], creg[CR];

  // XF: per-channel scale / shift of the previous block's BatchNorm in LDS, behind the two tile buffers
  float* const xf_tab = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);      // sc[C] | sh[C]
  if constexpr (XF) {
    const int R = p.xf_replicas > 1 ? p.xf_replicas : 1;
    for (int c = tid; c < p.C; c += 256) {
      long long l1 = 0, h1 = 0, l2 = 0, h2 = 0;      // replica rows: the accumulators' words added as integers
      for (int r = 0; r < R; ++r) {
        const long long* a1 = p.xf_sum + ((size_t)(2 * r) * p.C + c) * 2;
        const long long* a2 = p.xf_sum + ((size_t)(2 * r + 1) * p.C + c) * 2;
        l1 += a1[0]; h1 += a1[1]; l2 += a2[0]; h2 += a2[1];
      }
      const float s1 = det_value<KD6D_DET_ACT>(l1, h1), s2 = det_value<KD6D_DET_ACT>(l2, h2);
      const float mean = s1 * p.xf_inv_rows;
      const float var = fmaxf(s2 * p.xf_inv_rows - mean * mean, 0.f);
      const float is = rsqrtf(var + p.xf_eps);
      const float sc = p.xf_gamma[c] * is;
      xf_tab[c] = sc;
      xf_tab[p.C + c] = __builtin_fmaf(-mean, sc, p.xf_beta[c]);
      if (blockIdx.x == 0) {
        if (p.xf_save_mean) p.xf_save_mean[c] = mean;
        if (p.xf_save_invstd) p.xf_save_invstd[c] = is;
        if (p.xf_running_mean) p.xf_running_mean[c] = (1.f - p.xf_momentum) * p.xf_running_mean[c] + p.xf_momentum * mean;
        if (p.xf_running_var) p.xf_running_var[c] = (1.f - p.xf_momentum) * p.xf_running_var[c] + p.xf_momentum * var * p.xf_unbias;
      }
    }
    __syncthreads();
  }
  int center_tap = 0;
  if constexpr (XF) center_tap = (p.ks * p.ks) >> 1;

  auto issue_loads = [&](int kt) {
    const int kk = kt * BK + gcol * EG;
    const bool kvalid = kk < p.K;
    const int tap = kk / p.C;
    const int cc = kk - tap * p.C;
    const int ky = tap / p.ks;
    const int kx = tap - ky * p.ks;
    float xsc[XF ? EG : 1], xsh[XF ? EG : 1];
    if constexpr (XF) {
#pragma unroll
      for (int e = 0; e < EG; ++e) {
        xsc[e] = kvalid ? xf_tab[cc + e] : 0.f;
        xsh[e] = kvalid ? xf_tab[p.C + cc + e] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const int sh = rhw[i] >> 16, sw = rhw[i] & 0xffff;
      int sy, sx;
      bool ok = kvalid;
      if (MODE == MODE_FWD) {
        sy = ry[i] + ky;
        sx = rx[i] + kx;
      } else {
        const int ty = ry[i] - ky, tx = rx[i] - kx;
        ok = ok && ty >= 0 && tx >= 0;
        if (p.stride == 1) {
          sy = ty; sx = tx;
        } else {
          sy = ty / p.stride; sx = tx / p.stride;
          ok = ok && (sy * p.stride == ty) && (sx * p.stride == tx);
        }
      }
      ok = ok && (unsigned)sy < (unsigned)sh && (unsigned)sx < (unsigned)sw;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (ok) {
        const size_t off = (size_t)(rbase[i] + sy * sw + sx) * (size_t)p.C + (size_t)cc;
        if constexpr (XF) {
          const float* __restrict__ raw = reinterpret_cast<const float*>(p.src) + off;
          float f[EG];
#pragma unroll
          for (int e4 = 0; e4 < EG; e4 += 4) {
            const f32x4_t q = *reinterpret_cast<const f32x4_t*>(raw + e4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = __builtin_fmaf(q[e], xsc[e4 + e], xsh[e4 + e]);
              if (p.xf_act == KD6D_ACT_LEAKY) t = t > 0.f ? t : 0.1f * t;
              else if (p.xf_act == KD6D_ACT_RELU) t = fmaxf(t, 0.f);
              f[e4 + e] = t;
            }
          }
          v = f32_to_granule<T>(f);
          if (p.xf_z && tap == center_tap && tile_c == 0)
            *reinterpret_cast<u32x4_t*>(reinterpret_cast<T*>(p.xf_z) + off) = v;
        } else {
          v = *reinterpret_cast<const u32x4_t*>(src + off);
        }
      }
      preg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < CR; ++i) {
      const int crow = lrow + 32 * i;
      const int n = n0 + crow;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (kvalid && crow < BC && n < p.N) {
        v = *reinterpret_cast<const u32x4_t*>(wgt + (size_t)n * (size_t)p.K + (size_t)kk);
      }
      creg[i] = v;
    }
  };

  auto store_tiles = [&](int buf) {
    char* ptile = smem + buf * TILE_BYTES;
    char* ctile = ptile + BP * 128;
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const int prow = lrow + 32 * i;
      if (prow < BP) *reinterpret_cast<u32x4_t*>(ptile + lds_off(prow, gcol)) = preg[i];
    }
#pragma unroll
    for (int i = 0; i < CR; ++i) {
      const int crow = lrow + 32 * i;
      if (crow < BC) *reinterpret_cast<u32x4_t*>(ctile + lds_off(crow, gcol)) = creg[i];
    }
  };

  f32x4_t acc[CI][PI];
#pragma unroll
  for (int c = 0; c < CI; ++c)
#pragma unroll
    for (int q = 0; q < PI; ++q) acc[c][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  const int fr = lane & 15;
  const int fq = lane >> 4;

  issue_loads(0);
  store_tiles(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue_loads(kt + 1);

    const char* ptile = smem + buf * TILE_BYTES;
    const char* ctile = ptile + BP * 128;
#pragma unroll
    for (int ch = 0; ch < Frag<T>::CHUNKS; ++ch) {
      Frag<T> fa[CI], fb[PI];
#pragma unroll
      for (int c = 0; c < CI; ++c)
        load_frag<T>(ctile, wc * (BC / WC) + c * 16 + fr, ch, fq, fa[c]);
#pragma unroll
      for (int q = 0; q < PI; ++q)
        load_frag<T>(ptile, wp * (BP / WP) + q * 16 + fr, ch, fq, fb[q]);
#pragma unroll
      for (int c = 0; c < CI; ++c)
#pragma unroll
        for (int q = 0; q < PI; ++q) mma(fa[c], fb[q], acc[c][q]);
    }

    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  conv_epilogue_full<T, BP, BC, WP, WC, true, NORM || XF>(p, acc, m0, n0, wp, wc, lane, reinterpret_cast<float*>(smem), blockIdx.x,
                                        gridDim.x, tile_c);
}

// ---------------------------------------------------------------------------
// bf16 implicit GEMM with LDS-DMA staging (the kernel the step uses for N > 32).
//
// Same tile image and fragment reads as above, but the tiles travel global -> LDS directly
// (global_load_lds_dwordx4, no staging registers, no ds_write) into a ring of NSTAGE buffers so
// that NSTAGE-1 k-steps are in flight while one is multiplied: these layers are short-K / small-M
// and were bound by the global-load round trip of a 2-deep register pipeline, not by MFMA.
// One wave-instruction fills 8 rows x 128 B; lane l lands at slot l&7 of row l>>3, so it FETCHES
// the k-granule (l&7)^(l>>3) -- the XOR swizzle is applied to the source address (the LDS side of
// an LDS-DMA is linear).  Zero padding / tails fetch from a zero page.  One raw s_barrier per
// k-step; loads are retired with counted s_waitcnt vmcnt so they stay in flight across barriers.
// ---------------------------------------------------------------------------

template <int BP, int BC, int WP, int WC, int MODE, int NSTAGE, bool SPLIT = false>
__global__ __launch_bounds__(256, (SPLIT ? 1 : (BP == 128 && BC == 128) ? 2 : (BP == 64 && BC == 64) ? 5 : 1))
void conv_igemm_glds_kernel(const ConvParams p) {
  using T = bf16_t;
  constexpr int BK = 64;
  constexpr int PI = BP / WP / 16;
  constexpr int CI = BC / WC / 16;
  constexpr int PR = BP / 32;   // pixel rows fetched per lane per k-step (8 rows per wave-instruction)
  constexpr int CR = BC / 32;
  constexpr int L = PR + CR;    // LDS-DMA instructions per wave per k-step
  constexpr int STAGE = (BP + BC) * 128;
  static_assert(WP * WC == 4 && BP % 32 == 0 && BC % 32 == 0 && PI >= 1 && CI >= 1, "tile shape");
  static_assert(NSTAGE >= 3 && NSTAGE <= 6, "ring depth");
  static_assert((NSTAGE - 2) * L <= 63, "vmcnt range");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wp = wave % WP;
  const int wc = wave / WP;

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_c = p.p_fastest ? wg / p.n_ptiles : wg % p.n_ctiles;
  const int tile_p = p.p_fastest ? wg % p.n_ptiles : wg / p.n_ctiles;
  const int m0 = tile_p * BP;
  const int n0 = tile_c * BC;

  // ---- loader state: lane -> row (lane>>3) of each 8-row group, k-granule (lane&7)^(lane>>3) ----
  const int lrow = lane >> 3;
  const int gk = (lane & 7) ^ lrow;
  int ry[PR], rx[PR], rbase[PR], rhw[PR];
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    const int prow = 8 * (wave + 4 * i) + lrow;
    RowInfo ri = decode_row(p, m0 + prow);
    if (MODE == MODE_FWD) {
      ry[i] = ri.y * p.stride - p.pad;
      rx[i] = ri.x * p.stride - p.pad;
    } else {
      ry[i] = ri.y + p.pad;
      rx[i] = ri.x + p.pad;
    }
    rbase[i] = ri.src_base;
    rhw[i] = (ri.src_h << 16) | ri.src_w;
  }
  const T* __restrict__ src = reinterpret_cast<const T*>(p.src);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(p.wgt);
  const char* zero = reinterpret_cast<const char*>(kd6d_zero_page);
  int wofs[CR];   // element offset of the lane's weight row, or -1
#pragma unroll
  for (int i = 0; i < CR; ++i) {
    const int n = n0 + 8 * (wave + 4 * i) + lrow;
    wofs[i] = n < p.N ? n * p.K : -1;
  }

  // split-K: this workgroup contracts k-steps [kt0, kt0 + nk) and leaves an fp32 partial tile
  const int nk_all = (p.K + BK - 1) / BK;
  const int kt0 = SPLIT ? blockIdx.y * p.nk_split : 0;
  const int nk = SPLIT ? (kt0 + p.nk_split < nk_all ? p.nk_split : nk_all - kt0) : nk_all;

  // k-granule decode, advanced incrementally (one wrap per k-step at most when C >= 64)
  int kk = kt0 * BK + gk * 8;
  int cc, ky, kx;
  {
    const int tap = kk / p.C;
    cc = kk - tap * p.C;
    ky = tap / p.ks;
    kx = tap - ky * p.ks;
  }

  auto issue = [&](int stage) {
    char* base = smem + stage * STAGE + wave * 1024;
    const bool kvalid = kk < p.K;
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const int sh = rhw[i] >> 16, sw = rhw[i] & 0xffff;
      int sy, sx;
      bool ok = kvalid;
      if (MODE == MODE_FWD) {
        sy = ry[i] + ky;
        sx = rx[i] + kx;
      } else {
        const int ty = ry[i] - ky, tx = rx[i] - kx;
        ok = ok && ty >= 0 && tx >= 0;
        if (p.stride == 1) {
          sy = ty; sx = tx;
        } else {
          sy = ty / p.stride; sx = tx / p.stride;
          ok = ok && (sy * p.stride == ty) && (sx * p.stride == tx);
        }
      }
      ok = ok && (unsigned)sy < (unsigned)sh && (unsigned)sx < (unsigned)sw;
      const void* g = zero;
      if (ok) g = src + ((size_t)(rbase[i] + sy * sw + sx) * (size_t)p.C + (size_t)cc);
      glds16(g, base + i * 4096);
    }
#pragma unroll
    for (int i = 0; i < CR; ++i) {
      const void* g = zero;
      if (kvalid && wofs[i] >= 0) g = wgt + ((size_t)wofs[i] + (size_t)kk);
      glds16(g, base + BP * 128 + i * 4096);
    }
    // advance to the next k-step
    kk += BK;
    if (p.C >= BK) {
      cc += BK;
      if (cc >= p.C) {
        cc -= p.C;
        if (++kx == p.ks) { kx = 0; ++ky; }
      }
    } else {
      const int tap = kk / p.C;
      cc = kk - tap * p.C;
      ky = tap / p.ks;
      kx = tap - ky * p.ks;
    }
  };

  f32x4_t acc[CI][PI];
#pragma unroll
  for (int c = 0; c < CI; ++c)
#pragma unroll
    for (int q = 0; q < PI; ++q) acc[c][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;
  const int fq = lane >> 4;

#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue(s);

  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // retire this k-step's loads, keep the younger stages in flight
    {
      const int younger = nk - 1 - kt;          // k-steps issued after this one (capped by the ring)
      if (NSTAGE >= 6 && younger >= 4) wait_vmcnt<(NSTAGE >= 6 ? 4 : 0) * L>();
      else if (NSTAGE >= 5 && younger >= 3) wait_vmcnt<(NSTAGE >= 5 ? 3 : 0) * L>();
      else if (NSTAGE >= 4 && younger >= 2) wait_vmcnt<(NSTAGE >= 4 ? 2 : 0) * L>();
      else if (younger >= 1) wait_vmcnt<L>();
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NSTAGE - 1 < nk) {
      int st2 = stage + NSTAGE - 1;
      if (st2 >= NSTAGE) st2 -= NSTAGE;
      issue(st2);
    }
    const char* ptile = smem + stage * STAGE;
    const char* ctile = ptile + BP * 128;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      Frag<T> fa[CI], fb[PI];
#pragma unroll
      for (int c = 0; c < CI; ++c) load_frag<T>(ctile, wc * (BC / WC) + c * 16 + fr, ch, fq, fa[c]);
#pragma unroll
      for (int q = 0; q < PI; ++q) load_frag<T>(ptile, wp * (BP / WP) + q * 16 + fr, ch, fq, fb[q]);
#pragma unroll
      for (int c = 0; c < CI; ++c)
#pragma unroll
        for (int q = 0; q < PI; ++q) mma(fa[c], fb[q], acc[c][q]);
    }
    if (++stage == NSTAGE) stage = 0;
  }
  if (SPLIT) {      // raw fp32 partial tile; splitk_finalize_kernel sums the slabs and applies the epilogue
    float* slab = p.slab + (size_t)blockIdx.y * (size_t)p.M * (size_t)p.N;
#pragma unroll
    for (int q = 0; q < PI; ++q) {
      const int m = m0 + wp * (BP / WP) + q * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        const int n = n0 + wc * (BC / WC) + c * 16 + fq * 4;
        if (n < p.N) *reinterpret_cast<f32x4_t*>(slab + (size_t)m * p.N + n) = acc[c][q];
      }
    }
    return;
  }
  conv_epilogue_full<T, BP, BC, WP, WC>(p, acc, m0, n0, wp, wc, lane, reinterpret_cast<float*>(smem), blockIdx.x,
                                        gridDim.x, tile_c);
}

// out = epilogue(sum over splits of the fp32 partial tiles), 4 channels per thread.
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const ConvParams p, int nsplit) {
  const long long total = (long long)p.M * (p.N >> 2);
  const size_t stride = (size_t)p.M * (size_t)p.N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / (p.N >> 2));
    const int n = (int)(i - (long long)m * (p.N >> 2)) * 4;
    f32x4_t a = *reinterpret_cast<const f32x4_t*>(p.slab + (size_t)m * p.N + n);
    for (int s = 1; s < nsplit; ++s) {
      const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p.slab + s * stride + (size_t)m * p.N + n);
      a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
    }
    int drow = m, sg = 0;
#pragma unroll
    for (int q = 0; q < kMaxSeg; ++q)
      if (q < p.nseg && m >= p.seg[q].m_begin) { drow = p.seg[q].dst_row0 + (m - p.seg[q].m_begin); sg = q; }
    const float sscale = p.seg_scale ? p.seg_scale[sg] : 1.f;
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (p.ch_scale) {
      const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(p.ch_scale + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= s4[r];
    }
    if (p.ch_shift) {
      const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(p.ch_shift + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += s4[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] *= sscale;
      if (p.act == KD6D_ACT_LEAKY) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];
      else if (p.act == KD6D_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
    }
    const size_t o = (size_t)drow * (size_t)p.N + (size_t)n;
    if (p.residual) {
      if (p.out_f32) {
        const f32x4_t r4 = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p.residual) + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += r4[r];
      } else {
        const bf16_t* rp = reinterpret_cast<const bf16_t*>(p.residual) + o;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rp[r];
      }
    }
    if (p.out_f32) {
      *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.dst) + o) = f32x4_t{v[0], v[1], v[2], v[3]};
    } else {
      u32x2_t pk;
      pk.x = pack_bf16x2(v[0], v[1]);
      pk.y = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(p.dst) + o) = pk;
    }
  }
}


// ---------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1, bf16, C in {8, 16, 32}: the wide, shallow layers at the top of both
// backbones (up to a million pixels, K = 72..288).  They are bound by memory, not by MFMA: the
// generic kernels fetch every input pixel nine times in 16-B granules (one per tap), pad K to a
// multiple of 64 and pay a pixel decode per tile for two to five k-steps of work.
// Here a workgroup owns 256 consecutive pixels: their input rows plus a halo (level width + 1 on
// either side) and the WHOLE weight matrix land in LDS once (LDS-DMA, one wait), the im2col operand
// never exists: the MFMA pixel fragment of k-chunk c is a 16-B read at patch row
// (pixel + dy*W + dx) for the tap that lane's eight k-values belong to (k = tap*C + ci, so with
// C = 8 one 32-deep MFMA spans four taps), out-of-image taps and the K padding read a zero row.
// No pipeline inside the workgroup: several workgroups share a CU and overlap each other's phases.
// ---------------------------------------------------------------------------
template <int CG>
__device__ __forceinline__ int smallc_swz(int row) {
  return CG == 1 ? 0 : (CG == 2 ? ((row >> 3) & 1) : ((row >> 2) & 3));
}

template <int CG, int NB, int MODE>
__global__ __launch_bounds__(256, (NB == 1 ? 7 : 1)) void conv3x3_smallc_kernel(const ConvParams p, int halo, int total_rows,
                                                             int patch_bytes, int wbytes) {
  using T = bf16_t;
  constexpr int BP = 256;
  constexpr int BC = 16 * NB;
  constexpr int C = 8 * CG;
  constexpr int KG = 9 * CG;                 // 16-B granules of one weight row
  constexpr int NKC = (9 * C + 31) / 32;     // 32-deep k-chunks
  constexpr int WG = NKC * 4 + 1;            // 16-B slots per weight row in LDS (odd multiple of 16 B mod 128: no bank conflicts)
  constexpr int ROWB = 16 * CG;
  constexpr int PI = 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;
  char* const wlds = smem + patch_bytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_c = wg % p.n_ctiles;
  const int tile_p = wg / p.n_ctiles;
  const int m0 = tile_p * BP;
  const int n0 = tile_c * BC;
  const int patch_lo = m0 - halo;
  const int prows = BP + 2 * halo;           // real patch rows; row `prows` is the zero row
  const int zero_row = prows;

  const T* __restrict__ src = reinterpret_cast<const T*>(p.src);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(p.wgt);
  const char* zero = reinterpret_cast<const char*>(kd6d_zero_page);

  // ---- everything this workgroup reads, in one burst of LDS-DMA ----
  {
    const int pslots = patch_bytes >> 4;
    for (int s0 = wave * 64; s0 < pslots; s0 += 256) {
      const int s = s0 + lane;
      const int prow = s / CG;                     // CG is a power of two
      const int gs = s - prow * CG;
      const int row = patch_lo + prow;
      const void* g = zero;
      if (prow < prows && row >= 0 && row < total_rows)
        g = src + ((size_t)row * (size_t)C + (size_t)((gs ^ smallc_swz<CG>(prow)) * 8));
      glds16(g, patch + s0 * 16);
    }
    constexpr int wslots = (BC * WG + 63) / 64 * 64;
    for (int s0 = wave * 64; s0 < wslots; s0 += 256) {
      const int s = s0 + lane;
      const int n = s / WG;
      const int gs = s - n * WG;
      const void* g = zero;
      if (n < BC && n0 + n < p.N && gs < KG) g = wgt + ((size_t)(n0 + n) * (size_t)p.K + (size_t)(gs * 8));
      glds16(g, wlds + s0 * 16);
    }
  }

  // ---- pixel state, one decode per pixel of the tile (thread t <-> pixel m0 + t), shared through LDS ----
  const int fr = lane & 15;
  const int fq = lane >> 4;
  int* const pinfo = reinterpret_cast<int*>(wlds + wbytes);      // tap validity (9 bits) | level width << 16
  {
    const int m = m0 + tid;
    const RowInfo ri = decode_row(p, m);      // packed identically on both sides: src row == dst row == m
    // tap t = 3*ty + tx reads (y + dy, x + dx): three column bits, replicated into the valid rows
    const int lo_x = ri.x > 0 ? 1 : 0, hi_x = ri.x + 1 < ri.src_w ? 1 : 0;
    const int lo_y = ri.y > 0 ? 1 : 0, hi_y = ri.y + 1 < ri.src_h ? 1 : 0;
    const int cols = MODE == MODE_FWD ? (lo_x | 2 | (hi_x << 2)) : (hi_x | 2 | (lo_x << 2));
    const int r0 = MODE == MODE_FWD ? lo_y : hi_y, r2 = MODE == MODE_FWD ? hi_y : lo_y;
    const int mask = (r0 ? cols : 0) | (cols << 3) | (r2 ? (cols << 6) : 0);
    pinfo[tid] = (m < p.M ? mask : 0) | (ri.src_w << 16);
  }

  f32x4_t acc[NB][PI];
#pragma unroll
  for (int c = 0; c < NB; ++c)
#pragma unroll
    for (int q = 0; q < PI; ++q) acc[c][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  wait_vmcnt<0>();
  __syncthreads();

  int pbase[PI], pw[PI], pmask[PI];
#pragma unroll
  for (int q = 0; q < PI; ++q) {
    const int pl = wave * 64 + q * 16 + fr;
    const int info = pinfo[pl];
    pbase[q] = pl + halo;                     // patch row of the pixel itself
    pw[q] = info >> 16;
    pmask[q] = info & 0x1ff;
  }

#pragma unroll
  for (int ch = 0; ch < NKC; ++ch) {
    const int kg = ch * 4 + fq;               // this lane's 16-B k-granule: tap kg / CG, channel granule kg % CG
    const int tap = kg / CG;
    const int cgr = kg - tap * CG;
    const int ty = tap / 3;
    const int dy = MODE == MODE_FWD ? ty - 1 : 1 - ty;
    const int dx = MODE == MODE_FWD ? (tap - 3 * ty) - 1 : 1 - (tap - 3 * ty);
    Frag<T> fa[NB], fb[PI];
#pragma unroll
    for (int c = 0; c < NB; ++c)
      fa[c].v = *reinterpret_cast<const bf16x8_t*>(wlds + ((c * 16 + fr) * WG + kg) * 16);
#pragma unroll
    for (int q = 0; q < PI; ++q) {
      int r = pbase[q] + dy * pw[q] + dx;
      r = (tap < 9 && ((pmask[q] >> tap) & 1)) ? r : zero_row;
      fb[q].v = *reinterpret_cast<const bf16x8_t*>(patch + r * ROWB + ((cgr ^ smallc_swz<CG>(r)) << 4));
    }
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
      for (int q = 0; q < PI; ++q) mma(fa[c], fb[q], acc[c][q]);
  }
  conv_epilogue<T, BP, BC, 4, 1, false>(p, acc, m0, n0, wave, 0, lane, reinterpret_cast<float*>(smem));
}

// ---------------------------------------------------------------------------
// Weight gradient:  dW[n][j] += sum_m dY[m][n] * im2col(X)[m][j],  j = (tap, ci).
// Both operands are reduced along the pixel axis, which is the slow axis of NHWC,
// so tiles are staged TRANSPOSED into the same 128-B-row LDS image (row = channel,
// k = pixel) and consumed by the identical fragment reads.  The pixel axis is split over workgroups; every split leaves
// its partial dW image [cout][j] in the caller's SLAB with plain stores (slab[split][cout][j]) and the caller adds the
// splits in a fixed order (kd6d_grad_acc_resolve at the end of the reverse sweep): bitwise reproducible, and a plain
// store moves 4-5x the bytes per second of an atomic add.
// ---------------------------------------------------------------------------
struct WgradParams {
  int nseg, batch;
  int Cin, Cout;
  int ks, stride, pad;
  int J;        // ks*ks*Cin
  int M;        // total output pixels
  int n_jtiles;
  int m_chunk;  // pixels per split (multiple of the k-step)
  SegDev seg[kMaxSeg];
  const void* x;
  const void* dy;
  float* dw;          // slab: partial image of split s at dw + s * Cout * J (plain stores)
  long long slab_floats;
  int* plan_splits;   // host: non-null = dry run, the launch function reports its number of partial images here
  long long* dbias;   // optional: += column sums of dY (bias gradient), planar accumulators (hi word at + acc_hi),
  long long acc_hi;   //           added by the j-tile-0 workgroups
  int cu_budget;  // CUs this launch should aim to fill (callers that run several weight gradients side by side)
};

template <typename T>
__device__ __forceinline__ void lds_scatter_granule(char* tile, int ch0, int col, const u32x4_t& g);
template <>
__device__ __forceinline__ void lds_scatter_granule<bf16_t>(char* tile, int ch0, int col,
                                                            const u32x4_t& g) {
  const unsigned w[4] = {g.x, g.y, g.z, g.w};
  const int gran = col >> 3, sub = (col & 7) * 2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int row = ch0 + e;
    const unsigned short v = (unsigned short)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
    *reinterpret_cast<unsigned short*>(tile + lds_off(row, gran) + sub) = v;
  }
}
template <>
__device__ __forceinline__ void lds_scatter_granule<float>(char* tile, int ch0, int col,
                                                           const u32x4_t& g) {
  const unsigned w[4] = {g.x, g.y, g.z, g.w};
  const int gran = col >> 2, sub = (col & 3) * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    *reinterpret_cast<unsigned*>(tile + lds_off(ch0 + e, gran) + sub) = w[e];
  }
}

template <typename T, int BN, int BJ, int WN, int WJ>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int EG = Granule<T>::N;
  constexpr int BKM = 8 * EG;             // pixels per k-step
  constexpr int NI = BN / WN / 16;
  constexpr int JI = BJ / WJ / 16;
  constexpr int GN = BN / EG;             // granules per pixel row of the dY tile
  constexpr int GJ = BJ / EG;
  constexpr int LN = (BKM * GN + 255) / 256;
  constexpr int LJ = (BKM * GJ + 255) / 256;
  constexpr int TILE_BYTES = (BN + BJ) * 128;
  static_assert(WN * WJ == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % WN;
  const int wj = wave / WN;

  const int tile_j = blockIdx.x % p.n_jtiles;
  const int tile_n = blockIdx.x / p.n_jtiles;
  const int n0 = tile_n * BN;
  const int j0 = tile_j * BJ;
  const int m_lo = blockIdx.y * p.m_chunk;
  int m_hi = m_lo + p.m_chunk;
  if (m_hi > p.M) m_hi = p.M;

  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);

  // per-slot column decode for the X tile (fixed over the pixel loop)
  int jcc[LJ], jky[LJ], jkx[LJ], jpr[LJ], jch[LJ];
  bool jok[LJ];
#pragma unroll
  for (int i = 0; i < LJ; ++i) {
    const int idx = tid + 256 * i;
    const int gq = idx % GJ;
    jpr[i] = idx / GJ;
    jch[i] = gq * EG;
    const int j = j0 + gq * EG;
    jok[i] = (jpr[i] < BKM) && (j < p.J);
    const int tap = j / p.Cin;
    jcc[i] = j - tap * p.Cin;
    jky[i] = tap / p.ks;
    jkx[i] = tap - jky[i] * p.ks;
  }
  int npr[LN], nch[LN];
  bool nok[LN];
#pragma unroll
  for (int i = 0; i < LN; ++i) {
    const int idx = tid + 256 * i;
    const int gq = idx % GN;
    npr[i] = idx / GN;
    nch[i] = gq * EG;
    nok[i] = (npr[i] < BKM) && (n0 + gq * EG < p.Cout);
  }

  // reuse the row decoder through a ConvParams-shaped view
  auto decode = [&](int m, int& y, int& xq, int& sh, int& sw, int& sbase) {
    int mb = 0, hw = 1, dw = 1, s0 = 0;
    float ihw = 1.f, iw = 1.f;
    sh = 0; sw = 0;
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s) {
      if (s < p.nseg && m >= p.seg[s].m_begin) {
        mb = p.seg[s].m_begin; hw = p.seg[s].dst_hw; dw = p.seg[s].dst_w;
        sh = p.seg[s].src_h; sw = p.seg[s].src_w; s0 = p.seg[s].src_row0;
        ihw = p.seg[s].inv_hw; iw = p.seg[s].inv_w;
      }
    }
    const int local = m - mb;
    const int b = fast_div(local, hw, ihw);
    const int rem = local - b * hw;
    y = fast_div(rem, dw, iw);
    xq = rem - y * dw;
    sbase = s0 + b * sh * sw;
  };

  u32x4_t nreg[LN], jreg[LJ];

  auto issue_loads = [&](int mstep) {
#pragma unroll
    for (int i = 0; i < LN; ++i) {
      const int m = mstep + npr[i];
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (nok[i] && m < m_hi) {
        v = *reinterpret_cast<const u32x4_t*>(dy + (size_t)m * (size_t)p.Cout +
                                              (size_t)(n0 + nch[i]));
      }
      nreg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < LJ; ++i) {
      const int m = mstep + jpr[i];
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (jok[i] && m < m_hi) {
        int y, xq, sh, sw, sbase;
        decode(m, y, xq, sh, sw, sbase);
        const int sy = y * p.stride - p.pad + jky[i];
        const int sx = xq * p.stride - p.pad + jkx[i];
        if ((unsigned)sy < (unsigned)sh && (unsigned)sx < (unsigned)sw) {
          v = *reinterpret_cast<const u32x4_t*>(
              x + (size_t)(sbase + sy * sw + sx) * (size_t)p.Cin + (size_t)jcc[i]);
        }
      }
      jreg[i] = v;
    }
  };

  auto store_tiles = [&](int buf) {
    char* ntile = smem + buf * TILE_BYTES;
    char* jtile = ntile + BN * 128;
#pragma unroll
    for (int i = 0; i < LN; ++i)
      if (npr[i] < BKM && nch[i] < BN) lds_scatter_granule<T>(ntile, nch[i], npr[i], nreg[i]);
#pragma unroll
    for (int i = 0; i < LJ; ++i)
      if (jpr[i] < BKM && jch[i] < BJ) lds_scatter_granule<T>(jtile, jch[i], jpr[i], jreg[i]);
  };

  f32x4_t acc[NI][JI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < JI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int nsteps = (m_hi - m_lo + BKM - 1) / BKM;
  if (nsteps <= 0) return;

  issue_loads(m_lo);
  store_tiles(0);
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    if (st + 1 < nsteps) issue_loads(m_lo + (st + 1) * BKM);
    const char* ntile = smem + buf * TILE_BYTES;
    const char* jtile = ntile + BN * 128;
#pragma unroll
    for (int ch = 0; ch < Frag<T>::CHUNKS; ++ch) {
      Frag<T> fa[NI], fb[JI];
#pragma unroll
      for (int a = 0; a < NI; ++a) load_frag<T>(ntile, wn * (BN / WN) + a * 16 + fr, ch, fq, fa[a]);
#pragma unroll
      for (int b = 0; b < JI; ++b) load_frag<T>(jtile, wj * (BJ / WJ) + b * 16 + fr, ch, fq, fb[b]);
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < JI; ++b) mma(fa[a], fb[b], acc[a][b]);
    }
    if (st + 1 < nsteps) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // D rows = out channel n (4 per lane), D cols = j (lane&15): [n][j] fp32 image in LDS (the k-loop ended on a barrier),
  // then one rolled loop of fixed-point adds (the conversion inlined at every accumulator multiplied the build time)
  float* et = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int a = 0; a < NI; ++a) {
#pragma unroll
    for (int b = 0; b < JI; ++b) {
      const int jl = wj * (BJ / WJ) + b * 16 + fr;
      const int nl = wn * (BN / WN) + a * 16 + fq * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) et[(nl + r) * BJ + jl] = acc[a][b][r];
    }
  }
  __syncthreads();
  float* part = p.dw + (size_t)blockIdx.y * (size_t)p.Cout * (size_t)p.J;
  for (int i = tid; i < BN * BJ; i += 256) {
    const int nl = i / BJ, jl = i - nl * BJ;
    const int n = n0 + nl, j = j0 + jl;
    if (n < p.Cout && j < p.J) part[(size_t)n * (size_t)p.J + (size_t)j] = et[i];
  }
}

// ---------------------------------------------------------------------------
// Weight gradient, bf16, transposed-read form (the one the step uses).
//
// dW[n][j] = sum_m dY[m][n] * im2col(X)[m][j] contracts over pixels m, the slow axis of both NHWC
// operands.  Tiles are therefore staged in their NATURAL layout -- row = pixel, 256-B pitch, 16-B
// chunks swizzled by ch ^ (((row&3)<<2) | ((row>>2)&3)) -- with plain 16-B LDS stores, and the MFMA
// fragments (8 consecutive pixels of one channel per lane) come out of ds_read_b64_tr_b16, the
// gfx950 transposing LDS read: conflict-free for the two 4-row blocks a 32-lane half fetches.
// Split over pixel ranges; partial tiles are re-laid out through LDS so every atomic wave-instruction
// adds 64 consecutive elements of one dW row (512 contiguous bytes of the accumulators' lo plane).  The split count
// balances the k-loop against the atomic rate of the memory side (launch_wgrad_tr).
// ---------------------------------------------------------------------------
typedef short s16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int tr_off(int row, int ch) {
  return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int cb, int kc, int g, int q, int pp) {
  const int r0 = kc * 32 + 8 * g + q;
  const int ch = 2 * cb + (pp >> 1);
  const int sub = 8 * (pp & 1);
  typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + tr_off(r0, ch) + sub));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + tr_off(r0 + 4, ch) + sub));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int BN, int WN, int WJ>
__global__ __launch_bounds__(256) void conv_wgrad_tr_kernel(const WgradParams p) {
  constexpr int BJ = 128, BKM = 64;
  constexpr int NI = BN / WN / 16;
  constexpr int JI = BJ / WJ / 16;
  constexpr int GN = BN / 8, GJ = BJ / 8;   // 16-B granules per tile row
  constexpr int LN = (GN + 3) / 4, LJ = GJ / 4;
  constexpr int TILE = BKM * 256;
  constexpr int EP = BJ + 4;                // fp32 pitch of the epilogue image
  static_assert(WN * WJ == 4 && NI >= 1 && JI >= 1, "4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % WN;
  const int wj = wave / WN;

  const int tile_j = blockIdx.x % p.n_jtiles;
  const int tile_n = blockIdx.x / p.n_jtiles;
  const int n0 = tile_n * BN;
  const int j0 = tile_j * BJ;
  const int m_lo = blockIdx.y * p.m_chunk;
  int m_hi = m_lo + p.m_chunk;
  if (m_hi > p.M) m_hi = p.M;
  const int nsteps = (m_hi - m_lo + BKM - 1) / BKM;
  if (nsteps <= 0) return;

  const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(p.dy);

  // loader: thread -> pixel row tid>>2 of the k-step, granules (tid&3) + 4*i of that row
  const int prow = tid >> 2, sub = tid & 3;
  int jcc[LJ], jky[LJ], jkx[LJ];
  bool jok[LJ];
#pragma unroll
  for (int i = 0; i < LJ; ++i) {
    const int j = j0 + (sub + 4 * i) * 8;
    jok[i] = j < p.J;
    const int tap = j / p.Cin;
    jcc[i] = j - tap * p.Cin;
    jky[i] = tap / p.ks;
    jkx[i] = tap - jky[i] * p.ks;
  }
  bool nok[LN];
#pragma unroll
  for (int i = 0; i < LN; ++i) nok[i] = (sub + 4 * i < GN) && (n0 + (sub + 4 * i) * 8 < p.Cout);

  u32x4_t nreg[LN], jreg[LJ];
  const bool do_bias = p.dbias != nullptr && tile_j == 0;
  float bacc[LN][8];
#pragma unroll
  for (int i = 0; i < LN; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) bacc[i][e] = 0.f;

  auto issue_loads = [&](int mstep) {
    const int m = mstep + prow;
    const bool mv = m < m_hi;
    // pixel decode (level, image, y, x) once per k-step
    int mb = 0, hw = 1, dw = 1, sh = 0, sw = 0, s0 = 0;
    float ihw = 1.f, iw = 1.f;
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s) {
      if (s < p.nseg && m >= p.seg[s].m_begin) {
        mb = p.seg[s].m_begin; hw = p.seg[s].dst_hw; dw = p.seg[s].dst_w;
        sh = p.seg[s].src_h; sw = p.seg[s].src_w; s0 = p.seg[s].src_row0;
        ihw = p.seg[s].inv_hw; iw = p.seg[s].inv_w;
      }
    }
    const int local = m - mb;
    const int b = fast_div(local, hw, ihw);
    const int rem = local - b * hw;
    const int y = fast_div(rem, dw, iw);
    const int xq = rem - y * dw;
    const int sbase = s0 + b * sh * sw;
    const int by = y * p.stride - p.pad, bx = xq * p.stride - p.pad;
#pragma unroll
    for (int i = 0; i < LN; ++i) {
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (mv && nok[i])
        v = *reinterpret_cast<const u32x4_t*>(dy + (size_t)m * (size_t)p.Cout + (size_t)(n0 + (sub + 4 * i) * 8));
      nreg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < LJ; ++i) {
      u32x4_t v = {0u, 0u, 0u, 0u};
      const int sy = by + jky[i], sx = bx + jkx[i];
      if (mv && jok[i] && (unsigned)sy < (unsigned)sh && (unsigned)sx < (unsigned)sw)
        v = *reinterpret_cast<const u32x4_t*>(x + (size_t)(sbase + sy * sw + sx) * (size_t)p.Cin + (size_t)jcc[i]);
      jreg[i] = v;
    }
  };

  auto store_tiles = [&](int buf) {
    char* ntile = smem + buf * 2 * TILE;
    char* jtile = ntile + TILE;
#pragma unroll
    for (int i = 0; i < LN; ++i)
      if (sub + 4 * i < GN) *reinterpret_cast<u32x4_t*>(ntile + tr_off(prow, sub + 4 * i)) = nreg[i];
#pragma unroll
    for (int i = 0; i < LJ; ++i) *reinterpret_cast<u32x4_t*>(jtile + tr_off(prow, sub + 4 * i)) = jreg[i];
    if (do_bias) {          // every dY granule passes through here exactly once per n-tile
#pragma unroll
      for (int i = 0; i < LN; ++i) {
        float v[8];
        granule_to_f32<bf16_t>(nreg[i], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) bacc[i][e] += v[e];
      }
    }
  };

  f32x4_t acc[NI][JI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < JI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int tq = fr >> 2, tp = fr & 3;

  issue_loads(m_lo);
  store_tiles(0);
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    if (st + 1 < nsteps) issue_loads(m_lo + (st + 1) * BKM);
    const char* ntile = smem + buf * 2 * TILE;
    const char* jtile = ntile + TILE;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      bf16x8_t fa[NI], fb[JI];
#pragma unroll
      for (int a = 0; a < NI; ++a) fa[a] = tr_frag(ntile, wn * (BN / WN / 16) + a, kc, fq, tq, tp);
#pragma unroll
      for (int b = 0; b < JI; ++b) fb[b] = tr_frag(jtile, wj * (BJ / WJ / 16) + b, kc, fq, tq, tp);
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < JI; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (st + 1 < nsteps) store_tiles(buf ^ 1);
    __syncthreads();
  }

  if (p.dbias != nullptr && tile_j == 0) {
    // block-level reduction in LDS (the staging buffers are dead: the k-loop ended on a barrier), then
    // ONE atomic wave-instruction per 64 channels: the memory side retires ~one atomic instruction
    // per 50 ns per CU however few lanes it carries
    long long* bred = reinterpret_cast<long long*>(smem);     // BN accumulators {lo, hi}
    for (int i = tid; i < 2 * BN; i += 256) bred[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LN; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = bacc[i][e];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);   // lanes with equal lane&3: same channels
        if (lane < 4 && sub + 4 * i < GN) det_add_lds<KD6D_DET_GRAD>(&bred[((sub + 4 * i) * 8 + e) * 2], v);
      }
    __syncthreads();
    for (int i = tid; i < BN; i += 256)
      if (n0 + i < p.Cout) det_add_words_planar(p.dbias + n0 + i, p.acc_hi, bred[2 * i], bred[2 * i + 1]);
    __syncthreads();
  }
  // ---- epilogue: [n][j] fp32 image in LDS, then rows of 256 contiguous bytes per wave store ----
  float* et = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < JI; ++b) {
      const int nl = wn * (BN / WN) + a * 16 + fq * 4;
      const int jl = wj * (BJ / WJ) + b * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) et[(nl + r) * EP + jl] = acc[a][b][r];
    }
  __syncthreads();
  float* part = p.dw + (size_t)blockIdx.y * (size_t)p.Cout * (size_t)p.J;      // this split's partial image
  for (int nl = wave; nl < BN; nl += 4) {
    const int n = n0 + nl;
    if (n >= p.Cout) break;
#pragma unroll
    for (int h = 0; h < BJ / 64; ++h) {
      const int jl = h * 64 + lane;
      const int j = j0 + jl;
      if (j < p.J) part[(size_t)n * (size_t)p.J + (size_t)j] = et[nl * EP + jl];
    }
  }
}

// ---------------------------------------------------------------------------
// Weight gradient of the wide, shallow layers (Cin in {8,16,32}, Cout <= 64, 3x3/s1/p1 or 1x1, one level):
// dW is a few hundred to a few thousand numbers reduced over up to a million pixels.  The general kernel
// above pads J to 128 columns, decodes a pixel per thread per 64-pixel step and waits one global round trip
// per step: 16-45 us for 0.02-1.2 GFLOP.  Here a PERSISTENT workgroup walks tiles of R whole image rows:
//   * the dY rows and the X rows (+1 row above and below) of a tile land in LDS by LDS-DMA, double buffered,
//     in a ZERO-PADDED 2-D layout -- every image row carries one zero column left and right -- so that tap
//     (dy,dx) of position q is simply position q + dy*(W+2) + dx and no (pixel, tap) validity mask exists:
//     the pad positions of dY are zero and contribute nothing;
//   * the contraction runs over padded positions q; both MFMA operands (8 consecutive q of one channel per
//     lane) come out of the transposing LDS read, the X operand with the tap's row offset added per lane
//     (with C = 8 one 16-column block spans two taps, i.e. two offsets inside one read);
//   * the (n-block, j-block) accumulators are dealt round robin to the 4 waves and live in registers across
//     ALL tiles of the workgroup; one atomic flush at the end (a few hundred workgroups per address at most).
// ---------------------------------------------------------------------------
__device__ __forceinline__ bf16x8_t tr_read8(const char* base, int row, int pitch, int col_bytes) {
  typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(base + row * pitch + col_bytes));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(base + (row + 4) * pitch + col_bytes));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int CG, int NB, int KS>
__global__ __launch_bounds__(256) void conv_wgrad_small_kernel(const WgradParams p, int R, int tiles_per_img,
                                                               int ntiles, int buf_bytes, int prow) {
  constexpr int C = 8 * CG, J = KS * KS * C, JB = (J + 15) / 16, NBLK = NB * JB, MAXB = (NBLK + 3) / 4;
  constexpr int DYB = 32 * NB;        // LDS bytes of one dY position (Cout padded to 16 * NB channels)
  constexpr int PXB = 16 * CG;        // LDS bytes of one X position
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = p.seg[0].dst_w, H = p.seg[0].dst_h;
  const int Wp = KS == 3 ? W + 2 : W;
  const float inv_wp = 1.0f / (float)Wp;
  const int Q = R * Wp, Qpad = (Q + 31) & ~31;
  const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* __restrict__ dy = reinterpret_cast<const bf16_t*>(p.dy);
  const char* zero = reinterpret_cast<const char*>(kd6d_zero_page);

  auto issue = [&](int tile, char* b) {
    const int img = tile / tiles_per_img;
    const int y0 = (tile - img * tiles_per_img) * R;
    const size_t ibase = (size_t)img * H * W;
    for (int s0 = wave * 64; s0 < Qpad * 2 * NB; s0 += 256) {        // dY: 2 * NB 16-B slots per position
      const int s = s0 + lane;
      const int q = s / (2 * NB), g = s - q * (2 * NB);
      const int r = fast_div(q, Wp, inv_wp);
      const int xx = q - r * Wp - (KS == 3 ? 1 : 0), y = y0 + r;
      const void* src = zero;
      if (q < Q && (unsigned)xx < (unsigned)W && y < H && g * 8 < p.Cout)
        src = dy + ((ibase + (size_t)y * W + xx) * (size_t)p.Cout + (size_t)(g * 8));
      glds16(src, b + s0 * 16);
    }
    char* pb = b + Qpad * DYB;
    for (int s0 = wave * 64; s0 < prow * CG; s0 += 256) {            // X: CG slots per patch position
      const int s = s0 + lane;
      const int pr = s / CG, g = s - pr * CG;
      const int pi = pr - (KS == 3 ? 1 : 0);                         // patch row 0 is a spare zero row (tap -W-3)
      const int rr = pi >= 0 ? fast_div(pi, Wp, inv_wp) : 0;
      const int xx = pi - rr * Wp - (KS == 3 ? 1 : 0);
      const int y = y0 + rr - (KS == 3 ? 1 : 0);
      const void* src = zero;
      if (pi >= 0 && rr < R + (KS == 3 ? 2 : 0) && (unsigned)xx < (unsigned)W && (unsigned)y < (unsigned)H)
        src = x + ((ibase + (size_t)y * W + xx) * (size_t)C + (size_t)(g * 8));
      glds16(src, pb + s0 * 16);
    }
  };

  // ---- this wave's accumulator blocks: block b = nb * JB + jb for b = wave, wave + 4, ... ----
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  int a_col[MAXB], b_col[MAXB], b_off[MAXB];      // byte columns of the lane's 8-B piece; X row offset (or the zero rows)
  bool b_zero[MAXB];
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int blk = wave + 4 * i;
    const int nb = blk / JB, jb = blk - nb * JB;
    a_col[i] = nb * 32 + tp * 8;
    const int j0 = jb * 16 + tp * 4;                // this lane's 4 columns j0..j0+3 share a tap (C >= 8)
    const int tap = j0 / C, c = j0 - tap * C;
    b_col[i] = c * 2;
    b_zero[i] = j0 >= J;
    const int ky = tap / KS, kx = tap - ky * KS;
    b_off[i] = KS == 3 ? ky * Wp + kx : 0;          // LDS row of position q under tap (ky,kx): (q + ky*Wp + kx - 1) + 1
  }
  f32x4_t acc[MAXB];
#pragma unroll
  for (int i = 0; i < MAXB; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  int tile = blockIdx.x;
  if (tile < ntiles) issue(tile, smem);
  int cur = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    wait_vmcnt<0>();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x, smem + (cur ^ 1) * buf_bytes);
    const char* dyp = smem + cur * buf_bytes;
    const char* pat = dyp + Qpad * DYB;
    for (int qc = 0; qc < Qpad; qc += 32) {
      const int row = qc + 8 * fq + tq;
#pragma unroll
      for (int i = 0; i < MAXB; ++i) {
        if (wave + 4 * i < NBLK) {
          const bf16x8_t fa = tr_read8(dyp, row, DYB, a_col[i]);
          const bf16x8_t fb = tr_read8(pat, b_zero[i] ? prow - 8 + tq : row + b_off[i], PXB, b_zero[i] ? 0 : b_col[i]);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
        }
      }
    }
    cur ^= 1;
  }
  // ---- one flush per workgroup: [n][j] fp32 image in LDS (launch_wgrad_small sizes it), one rolled loop of adds ----
  constexpr int JP = JB * 16;
  __syncthreads();                       // the last tile's fragment reads are done
  float* et = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int blk = wave + 4 * i;
    if (blk >= NBLK) continue;
    const int nb = blk / JB, jb = blk - nb * JB;
#pragma unroll
    for (int r = 0; r < 4; ++r) et[(nb * 16 + fq * 4 + r) * JP + jb * 16 + fr] = acc[i][r];
  }
  __syncthreads();
  float* part = p.dw + (size_t)blockIdx.x * (size_t)p.Cout * J;      // one partial image per (persistent) workgroup
  for (int i = tid; i < NB * 16 * JP; i += 256) {
    const int n = i / JP, j = i - n * JP;
    if (n < p.Cout && j < J) part[(size_t)n * J + j] = et[i];
  }
}

// ---------------------------------------------------------------------------
// dgrad weight packing: wt[ci][ky][kx][co] <- w[co][ky][kx][ci], all layers at once.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const T* __restrict__ w, T* __restrict__ wt,
                                                         const int* __restrict__ desc, int n_layers) {
  __shared__ T tile[16][64 + 2];
  // find layer: desc[l*6+5] = first block of layer l (ascending)
  int l = 0;
  for (int i = 1; i < n_layers; ++i)
    if ((int)blockIdx.x >= desc[i * 6 + 5]) l = i;
  const int w_off = desc[l * 6 + 0], wt_off = desc[l * 6 + 1];
  const int cout = desc[l * 6 + 2], cin = desc[l * 6 + 3], ks = desc[l * 6 + 4];
  const int blk = blockIdx.x - desc[l * 6 + 5];
  const int taps = ks * ks;
  const int total = cout * taps * cin;
  if ((cin & 63) == 0 && (cout & 15) == 0) {
    // the layers that hold the weights (heads, FPN, the wide backbone stages): 16 (co) x 64 (ci) tiles of one tap through
    // LDS, read along ci and written along co -- both sides in contiguous pieces (element by element, read with a stride
    // of taps * cin, the launch took 35 us at the head of every step for 9 MB of traffic)
    const int n_blocks = (total + 2047) / 2048;           // what the caller's first_block table gives this layer
    const int ct = cout >> 4, it = cin >> 6;
    const int n_tiles = taps * ct * it;
    const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 4;      // load: row = co, 4 consecutive ci
    const int sr = threadIdx.x >> 2, sc = (threadIdx.x & 3) * 4;       // store: row = ci, 4 consecutive co
    for (int t = blk; t < n_tiles; t += n_blocks) {
      const int ci0 = (t % it) * 64, co0 = ((t / it) % ct) * 16, tap = t / (it * ct);
      const T* src = w + (size_t)w_off + ((size_t)(co0 + lr) * taps + tap) * cin + ci0 + lc;
#pragma unroll
      for (int k = 0; k < 4; ++k) tile[lr][lc + k] = src[k];
      __syncthreads();
      T* dst = wt + (size_t)wt_off + ((size_t)(ci0 + sr) * taps + tap) * cout + co0 + sc;
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[k] = tile[sc + k][sr];
      __syncthreads();
    }
    return;
  }
  // each thread produces one element of wt (co fastest => coalesced writes)
  for (int e = blk * 256 * 8 + threadIdx.x; e < min(total, (blk + 1) * 256 * 8); e += 256) {
    const int co = e % cout;
    const int rest = e / cout;
    const int tap = rest % taps;
    const int ci = rest / taps;
    wt[(size_t)wt_off + e] = w[(size_t)w_off + ((size_t)co * taps + tap) * cin + ci];
  }
}


template <typename T, int BP, int BC, int WP, int WC, int MODE, bool XF = false, bool NORM = false>
void launch_igemm_impl(const ConvParams& p, hipStream_t st) {
  ConvParams q = p;
  q.n_ctiles = (p.N + BC - 1) / BC;
  const int ptiles = (p.M + BP - 1) / BP;
  set_tile_order(q, ptiles, BP, BC);
  const size_t lds = (size_t)(BP + BC) * 128 * 2 + (XF ? (size_t)2 * p.C * sizeof(float) : 0);
  if (plan_only(ptiles * q.n_ctiles, 256, lds, NORM)) return;
  auto kern = conv_igemm_kernel<T, BP, BC, WP, WC, MODE, XF, NORM>;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_lds = lds;
  }
  hipLaunchKernelGGL(kern, dim3(ptiles * q.n_ctiles), dim3(256), lds, st, q);
}

// the fused-normalisation epilogue exists in the forward, non-XF variants only (kd6d_conv2d_fwd_norm)
template <typename T, int BP, int BC, int WP, int WC, int MODE, bool XF = false>
void launch_igemm(const ConvParams& p, hipStream_t st) {
  if constexpr (MODE == MODE_FWD && !XF) {
    // ... and so do the replica rows of the fused batch statistics (kd6d_conv2d_fwd_block)
    if (p.norm_dst || p.stats_replicas > 1) { launch_igemm_impl<T, BP, BC, WP, WC, MODE, false, true>(p, st); return; }
  }
  launch_igemm_impl<T, BP, BC, WP, WC, MODE, XF, false>(p, st);
}

template <int BP, int BC, int WP, int WC, int MODE, int NSTAGE>
void launch_glds(const ConvParams& p, hipStream_t st) {
  ConvParams q = p;
  q.n_ctiles = (p.N + BC - 1) / BC;
  const int ptiles = (p.M + BP - 1) / BP;
  set_tile_order(q, ptiles, BP, BC);
  const size_t lds = (size_t)(BP + BC) * 128 * NSTAGE;
  if (plan_only(ptiles * q.n_ctiles, 256, lds, false)) return;
  auto kern = conv_igemm_glds_kernel<BP, BC, WP, WC, MODE, NSTAGE>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(ptiles * q.n_ctiles), dim3(256), lds, st, q);
}



template <int CG, int NB, int MODE>
void launch_smallc(const ConvParams& p, int halo, int total_rows, hipStream_t st) {
  constexpr int BP = 256, BC = 16 * NB;
  constexpr int NKC = (9 * 8 * CG + 31) / 32, WG = NKC * 4 + 1;
  ConvParams q = p;
  q.n_ctiles = (p.N + BC - 1) / BC;
  const int ptiles = (p.M + BP - 1) / BP;
  q.n_ptiles = ptiles;
  // patch rows [m0 - halo, m0 + BP + halo) + the zero row, padded to whole 1-KB LDS-DMA bursts
  const int patch_bytes = ((BP + 2 * halo + 1) * 16 * CG + 1023) / 1024 * 1024;
  const size_t wbytes = (size_t)(BC * WG + 63) / 64 * 1024;
  size_t lds = (size_t)patch_bytes + wbytes + 256 * sizeof(int);
  const size_t epi = (size_t)2 * BC * sizeof(float) + 4096;      // statistics scratch of the epilogue
  if (lds < epi) lds = epi;
  auto kern = conv3x3_smallc_kernel<CG, NB, MODE>;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_lds = lds;
  }
  hipLaunchKernelGGL(kern, dim3(ptiles * q.n_ctiles), dim3(256), lds, st, q, halo, total_rows, patch_bytes, (int)wbytes);
}

// 3x3/s1/p1 layers with 8, 16 or 32 gather-source channels on maps up to 256 wide, both sides packed identically.
template <int MODE>
bool dispatch_smallc(const ConvParams& p, const kd6d_conv_geom* g, hipStream_t st) {
  const int force = (int)kd6d_opt(KD6D_OPT_CONV_SMALLC);
  if (force == 0) return false;
  if (p.ks != 3 || p.stride != 1 || p.pad != 1 || (p.C != 8 && p.C != 16 && p.C != 32) || (p.N & 3)) return false;
  if (p.stats && p.stats_groups > 0) return false;      // the group-statistics table wants the big staging buffers
  if (p.norm_dst || p.stats_replicas > 1) return false;   // no fused-normalisation epilogue / replica rows in this kernel
  // one burst per workgroup, no pipeline: pays once >= 2 workgroups per CU overlap each other (measured: the
  // 64x64-pixel layers and below are faster on the pipelined kernels)
  if (force < 0 && p.M < (1 << 17)) return false;
  int wmax = 0, rows = 0;
  for (int s = 0; s < g->nseg; ++s) {
    const kd6d_seg& q = g->seg[s];
    if (q.in_row0 != q.out_row0 || q.in_row0 != rows) return false;
    if (q.in_w > wmax) wmax = q.in_w;
    rows += g->batch * q.in_h * q.in_w;
  }
  // the patch is 256 pixels + a halo of (width + 1) rows on either side: up to 256-wide maps always (the size the kernel
  // was tuned on), wider ones (the 640- and 320-wide levels of full frames) while it stays within 32 KB, i.e. 8 or 16
  // channels -- read amplification (256 + 2 halo) / 256 grows to 6x at 640, all of it L2 hits, against 9 taps on the
  // generic kernels: 480 x 640 x 8 -> 32 forward 298 -> 165 us, 8 -> 8 forward / dgrad 266 / 289 -> 78 / 79,
  // 240 x 320 x 8 -> 16 71 / 90 -> 24 / 33; with 32 channels (57 KB, one or two workgroups per CU) 129 -> 216, excluded
  if (wmax > 256 && (wmax > (int)kd6d_opt(KD6D_OPT_CONV_SMALLC_WMAX) || (256 + 2 * (wmax + 1) + 1) * 2 * p.C > 32768)) return false;
  const int halo = wmax + 1;
  const int nb = p.N <= 16 ? 1 : (p.N <= 32 ? 2 : (p.N <= 64 ? 4 : 8));
#define KD6D_SMALLC_CASE(CG_, NB_) \
  if (p.C == 8 * CG_ && nb == NB_) { launch_smallc<CG_, NB_, MODE>(p, halo, rows, st); return true; }
  KD6D_SMALLC_CASE(1, 1) KD6D_SMALLC_CASE(1, 2) KD6D_SMALLC_CASE(1, 4) KD6D_SMALLC_CASE(1, 8)
  KD6D_SMALLC_CASE(2, 1) KD6D_SMALLC_CASE(2, 2) KD6D_SMALLC_CASE(2, 4) KD6D_SMALLC_CASE(2, 8)
  KD6D_SMALLC_CASE(4, 1) KD6D_SMALLC_CASE(4, 2) KD6D_SMALLC_CASE(4, 4) KD6D_SMALLC_CASE(4, 8)
#undef KD6D_SMALLC_CASE
  return false;
}

template <int BP, int BC, int WP, int WC, int MODE, int NSTAGE>
void launch_splitk(const ConvParams& p, int nsplit, hipStream_t st) {
  ConvParams q = p;
  q.n_ctiles = (p.N + BC - 1) / BC;
  const int ptiles = (p.M + BP - 1) / BP;
  set_tile_order(q, ptiles, BP, BC);
  const int nk_all = (p.K + 63) / 64;
  q.nk_split = (nk_all + nsplit - 1) / nsplit;
  nsplit = (nk_all + q.nk_split - 1) / q.nk_split;
  const size_t lds = (size_t)(BP + BC) * 128 * NSTAGE;
  auto kern = conv_igemm_glds_kernel<BP, BC, WP, WC, MODE, NSTAGE, true>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(ptiles * q.n_ctiles, nsplit), dim3(256), lds, st, q);
  const long long total = (long long)p.M * (p.N >> 2);
  int nb = (int)((total + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(splitk_finalize_kernel, dim3(nb), dim3(256), 0, st, q, nsplit);
}

// Split-K for the layers whose output yields too few tiles to fill 256 CUs while K is long (teacher
// stages 4/5, FPN top: M <= 4096, K = 2304..9216): partial tiles go to fp32 slabs in the caller's
// workspace (plain 16-B stores, no atomics), a small second launch sums them and applies the epilogue.
template <int MODE>
bool dispatch_splitk(const ConvParams& p, float* ws, size_t ws_bytes, hipStream_t st) {
  const int force = (int)kd6d_opt(KD6D_OPT_CONV_SPLITK);
  if (force == 0 || !ws || p.stats || (p.N & 3) || p.N <= 32) return false;
  const int nk = (p.K + 63) / 64;
  auto nblocks = [&](int bp, int bc) { return ((p.M + bp - 1) / bp) * ((p.N + bc - 1) / bc); };
  int tile = 0, ns = 0;
  if (force > 0) { tile = force / 100; ns = force % 100; }
  else if (nk >= 16 && nblocks(64, 64) <= 320) {
    tile = 2;
    ns = 768 / nblocks(64, 64);
    if (ns > nk / 6) ns = nk / 6;
    if (ns > 16) ns = 16;
  }
  if (tile == 0 || ns < 2) return false;
  if ((size_t)ns * p.M * p.N * sizeof(float) > ws_bytes) return false;
  ConvParams q = p;
  q.slab = ws;
  if (tile == 1) launch_splitk<128, 64, 2, 2, MODE, 3>(q, ns, st);
  else launch_splitk<64, 64, 2, 2, MODE, 3>(q, ns, st);
  return true;
}

// bf16, N > 32: LDS-DMA kernel.  Tile by how many workgroups the layer yields (256 CUs).
template <int MODE>
bool dispatch_glds(const ConvParams& p, hipStream_t st) {
  const int force = (int)kd6d_opt(KD6D_OPT_CONV_TILE);
  if (force == 0 || p.N <= 32) return false;
  // launches with a fused normalisation (kd6d_conv2d_fwd_norm) or replica rows of the batch statistics take the
  // register-staged kernel: this one is not compiled with that epilogue (and its 64-KB rings would not leave a
  // grid-barrier launch room to be resident at once)
  if (p.norm_dst || p.stats_replicas > 1) return false;
  const int N = p.N, M = p.M;
  auto nblocks = [&](int bp, int bc) { return ((M + bp - 1) / bp) * ((N + bc - 1) / bc); };
  // measured (tools/bench_conv.py): with enough workgroups the register-staged kernel is as fast or faster
  // (several workgroups per CU hide the load round trip); the layers with <= ~1 workgroup per CU and a long
  // K (teacher stages 4/5, FPN top) are bound by that round trip and gain from a deep LDS-DMA ring
  int pick = 0;
  if (nblocks(128, 64) < 384) pick = 3;
  else if (nblocks(128, 64) <= 640 && p.K >= 1024) pick = 2;       // stride-2 stage-3 entry: 28 -> 25 us
  // k-steps that straddle taps (source channels not a multiple of 64) pay a tap decode per step here; on the
  // large maps (dgrad of the student's cls / pose heads) the register-staged kernel is 15-25 % faster
  if (pick == 3 && (p.C & 63) && M > 16384) pick = 0;
  if (force > 0) pick = force;
  if (pick == 0) return false;
  if (pick == 1) launch_glds<128, 128, 2, 2, MODE, 3>(p, st);
  else if (pick == 2) launch_glds<128, 64, 2, 2, MODE, 3>(p, st);
  else if (pick == 3) launch_glds<64, 64, 2, 2, MODE, 4>(p, st);
  else launch_glds<64, 64, 2, 2, MODE, 6>(p, st);
  return true;
}

template <typename T, int MODE, bool XF = false>
void dispatch_igemm(const ConvParams& p, hipStream_t st) {
  const int N = p.N, M = p.M;
  auto nblocks = [&](int bp, int bc) { return ((M + bp - 1) / bp) * ((N + bc - 1) / bc); };
  if (N <= 16) {
    if (nblocks(256, 16) >= 512) launch_igemm<T, 256, 16, 4, 1, MODE, XF>(p, st);
    else launch_igemm<T, 64, 16, 4, 1, MODE, XF>(p, st);
  } else if (N <= 32) {
    if (nblocks(256, 32) >= 512) launch_igemm<T, 256, 32, 4, 1, MODE, XF>(p, st);
    else launch_igemm<T, 64, 32, 4, 1, MODE, XF>(p, st);
  } else if (N <= 64) {
    if (nblocks(128, 64) >= 384) launch_igemm<T, 128, 64, 2, 2, MODE, XF>(p, st);
    else launch_igemm<T, 64, 64, 2, 2, MODE, XF>(p, st);
  } else {
    if (nblocks(128, 128) >= 384) launch_igemm<T, 128, 128, 2, 2, MODE, XF>(p, st);
    else if (nblocks(128, 64) >= 384) launch_igemm<T, 128, 64, 2, 2, MODE, XF>(p, st);
    else launch_igemm<T, 64, 64, 2, 2, MODE, XF>(p, st);
  }
}

// dry run (p.plan_splits): report the number of partial images; else check the slab holds them (error flag otherwise)
thread_local bool g_wgrad_slab_short = false;
bool wgrad_slab_ok(const WgradParams& p, int parts) {
  if (p.plan_splits) { *p.plan_splits = parts; return false; }
  if ((long long)parts * p.Cout * p.J > p.slab_floats) { g_wgrad_slab_short = true; return false; }
  return true;
}

template <typename T, int BN, int BJ, int WN, int WJ>
void launch_wgrad(const WgradParams& p, hipStream_t st) {
  constexpr int BKM = 8 * Granule<T>::N;
  WgradParams q = p;
  q.n_jtiles = (p.J + BJ - 1) / BJ;
  const int ntiles = (p.Cout + BN - 1) / BN;
  const int tiles = q.n_jtiles * ntiles;
  const int steps_total = (p.M + BKM - 1) / BKM;
  int splits = (1024 + tiles - 1) / tiles;          // aim for ~1024 workgroups
  int max_splits = (steps_total + 3) / 4;           // at least 4 k-steps per split
  if (max_splits < 1) max_splits = 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int steps_per = (steps_total + splits - 1) / splits;
  q.m_chunk = steps_per * BKM;
  splits = (p.M + q.m_chunk - 1) / q.m_chunk;
  if (!wgrad_slab_ok(p, splits)) return;
  const size_t lds = (size_t)(BN + BJ) * 128 * 2;
  auto kern = conv_wgrad_kernel<T, BN, BJ, WN, WJ>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles, splits), dim3(256), lds, st, q);
}

template <int BN, int WN, int WJ>
void launch_wgrad_tr(const WgradParams& p, hipStream_t st) {
  constexpr int BJ = 128, BKM = 64;
  WgradParams q = p;
  q.n_jtiles = (p.J + BJ - 1) / BJ;
  const int ntiles = (p.Cout + BN - 1) / BN;
  const int tiles = q.n_jtiles * ntiles;
  const int steps_total = (p.M + BKM - 1) / BKM;
  // time ~ (steps/S) * t_step + S * |dW| / (flush rate), t_step ~ 1.6 us measured.  A split's partial image costs a plain
  // store here (~6 TB/s) and a read by kd6d_grad_acc_resolve at the end of the sweep (~4 TB/s): 2.4 TB/s together (the fp32
  // atomic flush of rounds 1-3: 1.3 TB/s)
  //   => S* = sqrt(steps * t_step * rate / |dW|); at most 2 workgroups per CU, because many
  //   workgroups adding into one small dW are contention-bound (measured: 2048 -> 512 = -25 %)
  // a caller that keeps several weight gradients in flight asks each for a fraction of the device: fewer,
  // longer splits -> proportionally fewer atomic tile flushes for the same k-loop work
  const double frac = (double)p.cu_budget / (double)cached_cu_count();
  const double dw_bytes = (double)p.Cout * (double)p.J * 4.0;
  int splits = (int)(frac * sqrt((double)steps_total * 3.8e6 / dw_bytes) + 0.5);
  if (splits > 512 / tiles) splits = 512 / tiles;
  if (splits > steps_total / 2) splits = steps_total / 2;
  if (splits < 1) splits = 1;
  const int steps_per = (steps_total + splits - 1) / splits;
  q.m_chunk = steps_per * BKM;
  splits = (p.M + q.m_chunk - 1) / q.m_chunk;
  if (!wgrad_slab_ok(p, splits)) return;
  const size_t stage = (size_t)2 * 2 * BKM * 256;
  const size_t epi = (size_t)BN * (BJ + 4) * 4;
  const size_t lds = stage > epi ? stage : epi;
  auto kern = conv_wgrad_tr_kernel<BN, WN, WJ>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles, splits), dim3(256), lds, st, q);
}

template <int CG, int NB, int KS>
void launch_wgrad_small(const WgradParams& p, int R, hipStream_t st) {
  const int W = p.seg[0].dst_w, H = p.seg[0].dst_h;
  const int Wp = KS == 3 ? W + 2 : W;
  const int Qpad = (R * Wp + 31) & ~31;
  int prow = Qpad + (KS == 3 ? 2 * Wp + 3 : 0) + 8;            // furthest tap of the last position + 8 zero rows
  const int unit = 64 / CG;                                    // whole 1-KB LDS-DMA bursts
  prow = (prow + unit - 1) / unit * unit;
  const int buf_bytes = Qpad * 32 * NB + prow * 16 * CG;
  const int tiles_per_img = (H + R - 1) / R;
  const int ntiles = p.batch * tiles_per_img;
  int grid = 2 * p.cu_budget;                                  // persistent: one flush per workgroup
  if (grid > ntiles) grid = ntiles;
  if (!wgrad_slab_ok(p, grid)) return;
  constexpr int JB_ = (KS * KS * 8 * CG + 15) / 16;
  const size_t image = (size_t)NB * 16 * JB_ * 16 * sizeof(float);      // the flush's [n][j] image
  const size_t lds = (size_t)2 * buf_bytes > image ? (size_t)2 * buf_bytes : image;
  auto kern = conv_wgrad_small_kernel<CG, NB, KS>;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_lds = lds;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p, R, tiles_per_img, ntiles, buf_bytes, prow);
}

// wide, shallow layers: Cin in {8,16,32}, Cout <= 64, 3x3/s1/p1 or 1x1/s1, one level, >= 2^15 pixels, no bias gradient
bool dispatch_wgrad_small(const WgradParams& p, const kd6d_conv_geom* g, hipStream_t st) {
  const int force = (int)kd6d_opt(KD6D_OPT_WGRAD_SMALL);
  if (force == 0 || p.dbias != nullptr || g->nseg != 1 || p.stride != 1) return false;
  if (!((p.ks == 3 && p.pad == 1) || (p.ks == 1 && p.pad == 0))) return false;
  if ((p.Cin != 8 && p.Cin != 16 && p.Cin != 32) || p.Cout > 64 || (p.Cout & 7)) return false;
  const kd6d_seg& q = g->seg[0];
  if (q.in_row0 != 0 || q.out_row0 != 0 || q.in_w > 256) return false;
  // measured (tools/bench_conv.py, B = 16): the 1x1 layers gain (64x64 map, 16 -> 8 channels: 16.2 -> 7.5 us); the
  // 3x3 layers do NOT -- a tile is one DMA round trip of ~3 us for ~0.3 us of MFMA work and the persistent grid
  // that keeps the atomic flush small also keeps too few round trips in flight (256x256x8->8: 46 -> 45 us,
  // 128x128x8->16: 26 -> 31, 64x64x8->64: 21 -> 20; more workgroups: slower, the flush serialises).  They stay
  // on the general kernel unless forced; a deeper DMA ring per workgroup is the open improvement.
  if (force < 0 && (p.M < (1 << 15) || p.ks != 1)) return false;
  const int W = q.in_w, H = q.in_h;
  int R = 512 / W;                     // ~512 positions per tile (256: 9.5 us on the 16 -> 8 layer, 512: 7.5)
  if (R < 1) R = 1;
  if (R > H) R = H;
  const int nb = (p.Cout + 15) / 16;
  const int cg = p.Cin / 8;
  auto lds_bytes = [&](int r) {        // as launch_wgrad_small sizes the two buffers
    const int wp = p.ks == 3 ? W + 2 : W;
    const int qpad = (r * wp + 31) & ~31;
    int prow = qpad + (p.ks == 3 ? 2 * wp + 3 : 0) + 8;
    const int unit = 64 / cg;
    prow = (prow + unit - 1) / unit * unit;
    return (size_t)2 * ((size_t)qpad * 32 * nb + (size_t)prow * 16 * cg);
  };
  while (R > 1 && lds_bytes(R) > 72 * 1024) R >>= 1;           // two workgroups per CU
  if (lds_bytes(R) > 144 * 1024) return false;
#define KD6D_WS_CASE(CG_, NB_, KS_) \
  if (cg == CG_ && nb == NB_ && p.ks == KS_) { launch_wgrad_small<CG_, NB_, KS_>(p, R, st); return true; }
  KD6D_WS_CASE(1, 1, 3) KD6D_WS_CASE(1, 2, 3) KD6D_WS_CASE(1, 4, 3)
  KD6D_WS_CASE(2, 1, 3) KD6D_WS_CASE(2, 2, 3) KD6D_WS_CASE(2, 4, 3)
  KD6D_WS_CASE(4, 1, 3) KD6D_WS_CASE(4, 2, 3) KD6D_WS_CASE(4, 4, 3)
  KD6D_WS_CASE(1, 1, 1) KD6D_WS_CASE(2, 1, 1) KD6D_WS_CASE(4, 1, 1)
  KD6D_WS_CASE(1, 2, 1) KD6D_WS_CASE(2, 2, 1) KD6D_WS_CASE(4, 2, 1)
  KD6D_WS_CASE(1, 4, 1) KD6D_WS_CASE(2, 4, 1) KD6D_WS_CASE(4, 4, 1)
#undef KD6D_WS_CASE
  return false;
}

void dispatch_wgrad_tr(const WgradParams& p, hipStream_t st) {
  if (p.Cout <= 16) launch_wgrad_tr<16, 1, 4>(p, st);
  else if (p.Cout <= 32) launch_wgrad_tr<32, 1, 4>(p, st);
  else if (p.Cout <= 64) launch_wgrad_tr<64, 1, 4>(p, st);
  else launch_wgrad_tr<128, 2, 2>(p, st);
}

template <typename T>
void dispatch_wgrad(const WgradParams& p, hipStream_t st) {
  if (p.Cout <= 16) launch_wgrad<T, 16, 128, 1, 4>(p, st);
  else if (p.Cout <= 32) launch_wgrad<T, 32, 128, 1, 4>(p, st);
  else if (p.Cout <= 64 || p.J <= 64) launch_wgrad<T, 64, 64, 2, 2>(p, st);
  else launch_wgrad<T, 128, 128, 2, 2>(p, st);
}

}  // namespace

thread_local kd6d_detail::LaunchPlan* kd6d_detail::g_launch_plan = nullptr;

namespace {

// forward dispatch: the per-layer kernel choice
void dispatch_fwd(const ConvParams& p, const kd6d_conv_geom* g, int dtype, void* workspace, int64_t workspace_bytes,
                  hipStream_t st) {
  if (dtype == KD6D_BF16) {
    if (!dispatch_smallc<MODE_FWD>(p, g, st) && !dispatch_halo_fwd(p, g, st) &&
        !dispatch_splitk<MODE_FWD>(p, reinterpret_cast<float*>(workspace), (size_t)(workspace_bytes > 0 ? workspace_bytes : 0), st) &&
        !dispatch_glds<MODE_FWD>(p, st))
      dispatch_igemm<bf16_t, MODE_FWD>(p, st);
  } else {
    dispatch_igemm<float, MODE_FWD>(p, st);
  }
}

int fwd_params(const kd6d_conv_geom* g, int dtype, const char* who, ConvParams& p) {
  int rc = check_geom(g, dtype, who);
  if (rc) return rc;
  memset(&p, 0, sizeof(p));
  p.nseg = g->nseg; p.batch = g->batch; p.C = g->cin; p.N = g->cout;
  p.ks = g->ksize; p.stride = g->stride; p.pad = g->pad;
  p.K = g->ksize * g->ksize * g->cin;
  KD6D_CHECK_ARG(fill_segs(g, false, p.seg, &p.M), "%s: grid too large", who);
  return KD6D_OK;
}

int set_stats(const kd6d_conv_geom* g, ConvParams& p, kd6d_acc* stats, int stats_groups, const char* who) {
  KD6D_CHECK_ARG(g->cout % 4 == 0 && stats_groups >= 0, "%s: fused statistics need cout %% 4 == 0", who);
  KD6D_CHECK_ARG(stats_groups == 0 || (g->cout % stats_groups == 0 && (g->cout / stats_groups) % 4 == 0 &&
                                       g->cout / stats_groups <= 8),
                 "%s: fused group statistics need 4 or 8 channels per group (cout=%d, groups=%d)", who, g->cout,
                 stats_groups);
  p.stats = reinterpret_cast<long long*>(stats); p.stats_groups = stats_groups;
  p.stats_skip = 0;
  if (stats_groups > 0) {
    p.stats_cpg_shift = (g->cout / stats_groups) == 8 ? 3 : 2;
    // the epilogue sums whole 16-row fragments that lie inside one image; a level where that does not hold is left to
    // stats_followup()
    for (int s2 = 0; s2 < p.nseg; ++s2)
      if ((p.seg[s2].dst_hw % 16) != 0 || (p.seg[s2].m_begin % 16) != 0) p.stats_skip |= 1 << s2;
  }
  return KD6D_OK;
}

// Would kd6d_conv2d_fwd_norm take the fused path?  Dry run of the dispatch with the fields that steer it set the way
// the real call sets them, then the residency rule of kd6d_barrier.h.
}  // namespace

// The levels the epilogue's group statistics skip (ConvParams::stats_skip): summed from the stored tensor by a small
// gn_stats launch behind the convolution (norm_ops.hip).  Also called by the pair bracket (conv_halo.hip) for the
// launches it deferred.
int kd6d_detail::stats_followup(const ConvParams& p, bool dst_f32, hipStream_t st) {
  if (!p.stats || p.stats_groups <= 0 || p.stats_skip == 0) return KD6D_OK;
  int row0[kMaxSeg] = {0}, hw[kMaxSeg] = {0};
  for (int s2 = 0; s2 < p.nseg; ++s2) { row0[s2] = p.seg[s2].dst_row0; hw[s2] = p.seg[s2].dst_hw; }
  return gn_stats_levels(dst_f32 ? 1 : 0, p.dst, row0, hw, p.nseg, p.batch, p.N, p.stats_groups, (unsigned)p.stats_skip,
                         p.stats, st);
}

namespace {
bool norm_fusable(const kd6d_conv_geom* g, int dtype, int kind, int groups, ConvParams& p, LaunchPlan& plan) {
  // option conv.fuse_norm: bit 0 = GroupNorm launches, bit 1 = BatchNorm launches
  if (((int)kd6d_opt(KD6D_OPT_CONV_FUSE_NORM) & (kind == KD6D_NORM_GROUP ? 1 : 2)) == 0) return false;
  if (fwd_params(g, dtype, "kd6d_conv2d_fwd_norm", p) != KD6D_OK) return false;
  static float dummy;
  static kd6d_acc dummy_acc;
  if (set_stats(g, p, &dummy_acc, kind == KD6D_NORM_GROUP ? groups : 0, "kd6d_conv2d_fwd_norm") != KD6D_OK) return false;
  if (kind == KD6D_NORM_GROUP && groups <= 0) return false;
  if (p.stats_skip) return false;       // a level whose statistics need the separate pass: nothing to wait for in-kernel
  p.norm_dst = &dummy;
  p.out_f32 = 1;
  g_launch_plan = &plan;
  dispatch_fwd(p, g, dtype, nullptr, 0, nullptr);
  g_launch_plan = nullptr;
  if (!plan.fused_epilogue || plan.grid <= 0) return false;
  const int ncu = cached_cu_count();
  const size_t lds = plan.lds > 0 ? plan.lds : 1;
  const int waves = plan.threads / 64;
  if (kind == KD6D_NORM_BATCH) {
    // grid barrier: every workgroup of the launch pinned until the last one has arrived.  Admit only launches that fit
    // in HALF of the device's LDS and wave slots (the other half is what window-barrier launches on other streams and
    // fragmentation may hold), at no more than two workgroups per CU
    if ((size_t)plan.grid * lds > (size_t)ncu * (160u << 10) / 2) return false;
    if (plan.grid * waves > ncu * 32 / 2) return false;
    if (plan.grid > 2 * ncu) return false;
  } else {
    // window barrier: with workgroup id == tile id a tile waits for the pixel tiles of its own (level, image) keys only
    int hw_max = 1;
    for (int s = 0; s < g->nseg; ++s) hw_max = hw_max > g->seg[s].out_h * g->seg[s].out_w ? hw_max : g->seg[s].out_h * g->seg[s].out_w;
    const int n_ctiles = (g->cout + 15) / 16;       // upper bound of the channel tiles
    if (n_ctiles > 16 * KD6D_NORM_MAX_CTILES) return false;
    (void)hw_max;
  }
  return true;
}

}  // namespace

extern "C" int kd6d_conv2d_fwd(const kd6d_conv_geom* g, int dtype, const void* x, const void* w,
                               void* y, const float* ch_scale, const float* ch_shift, int act,
                               const void* residual, const float* seg_scale, int out_f32,
                               kd6d_acc* stats, int stats_groups, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  ConvParams p;
  int rc = fwd_params(g, dtype, "kd6d_conv2d_fwd", p);
  if (rc) return rc;
  KD6D_CHECK_ARG(x && w && y, "kd6d_conv2d_fwd: null tensor pointer");
  KD6D_CHECK_ARG(act >= 0 && act <= 2, "kd6d_conv2d_fwd: bad act %d", act);
  p.src = x; p.wgt = w; p.dst = y;
  p.ch_scale = ch_scale; p.ch_shift = ch_shift; p.residual = residual; p.seg_scale = seg_scale;
  p.act = act; p.out_f32 = out_f32 ? 1 : 0;
  if (stats) {
    rc = set_stats(g, p, stats, stats_groups, "kd6d_conv2d_fwd");
    if (rc) return rc;
  }
  const int pending = kd6d_conv2d_pair_pending();
  dispatch_fwd(p, g, dtype, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
  KD6D_CHECK_LAUNCH("kd6d_conv2d_fwd");
  if (p.stats_skip && kd6d_conv2d_pair_pending() == pending) {       // launched (not recorded by a pair bracket)
    rc = stats_followup(p, p.out_f32 || dtype == KD6D_F32, reinterpret_cast<hipStream_t>(stream));
    if (rc) return rc;
  }
  return KD6D_OK;
}

extern "C" int kd6d_conv2d_fwd_block(const kd6d_conv_geom* g, int dtype, const void* x, const kd6d_bn_in* bn, void* z_out,
                                     const void* w, float* y_raw, kd6d_acc* stats, int stats_replicas, void* stream) {
  ConvParams p;
  int rc = fwd_params(g, dtype, "kd6d_conv2d_fwd_block", p);
  if (rc) return rc;
  KD6D_CHECK_ARG(x && w && y_raw, "kd6d_conv2d_fwd_block: null tensor pointer");
  KD6D_CHECK_ARG(stats_replicas >= 1 && stats_replicas <= 64, "kd6d_conv2d_fwd_block: stats_replicas=%d", stats_replicas);
  p.src = x; p.wgt = w; p.dst = y_raw;
  p.act = KD6D_ACT_NONE; p.out_f32 = 1;
  if (stats) {
    rc = set_stats(g, p, stats, 0, "kd6d_conv2d_fwd_block");
    if (rc) return rc;
    p.stats_replicas = stats_replicas;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!bn) {
    KD6D_CHECK_ARG(z_out == nullptr, "kd6d_conv2d_fwd_block: z_out without a BatchNorm to apply");
    dispatch_fwd(p, g, dtype, nullptr, 0, st);
  } else {
    KD6D_CHECK_ARG(bn->sums && bn->gamma && bn->beta && bn->replicas >= 1 && bn->act >= 0 && bn->act <= 2,
                   "kd6d_conv2d_fwd_block: bad BatchNorm description");
    KD6D_CHECK_ARG(g->stride == 1 && (g->ksize == 1 || g->ksize == 3) && g->pad == g->ksize / 2,
                   "kd6d_conv2d_fwd_block: BatchNorm-on-load needs a 1x1 or 3x3 stride-1 'same' convolution");
    long long rows_in = 0;
    for (int s = 0; s < g->nseg; ++s) rows_in += (long long)g->batch * g->seg[s].in_h * g->seg[s].in_w;
    p.xf_sum = reinterpret_cast<const long long*>(bn->sums); p.xf_replicas = bn->replicas; p.xf_gamma = bn->gamma; p.xf_beta = bn->beta;
    p.xf_eps = bn->eps; p.xf_momentum = bn->momentum; p.xf_act = bn->act;
    p.xf_inv_rows = 1.0f / (float)rows_in;
    p.xf_unbias = rows_in > 1 ? (float)rows_in / (float)(rows_in - 1) : 1.0f;
    p.xf_save_mean = bn->save_mean; p.xf_save_invstd = bn->save_invstd;
    p.xf_running_mean = bn->running_mean; p.xf_running_var = bn->running_var;
    p.xf_z = z_out;
    if (dtype == KD6D_BF16) dispatch_igemm<bf16_t, MODE_FWD, true>(p, st);
    else dispatch_igemm<float, MODE_FWD, true>(p, st);
  }
  KD6D_CHECK_LAUNCH("kd6d_conv2d_fwd_block");
  return KD6D_OK;
}

extern "C" int kd6d_conv2d_fwd_norm_fusable(const kd6d_conv_geom* g, int dtype, int kind, int groups) {
  if (!g || (kind != KD6D_NORM_GROUP && kind != KD6D_NORM_BATCH)) return 0;
  ConvParams p;
  LaunchPlan plan;
  return norm_fusable(g, dtype, kind, groups, p, plan) ? 1 : 0;
}

extern "C" int kd6d_conv2d_fwd_norm(const kd6d_conv_geom* g, int dtype, const void* x, const void* w, void* raw_out,
                                    const float* bias, const kd6d_conv_norm* norm, void* stream) {
  KD6D_CHECK_ARG(g && x && w && norm && norm->y && norm->gamma && norm->beta && norm->stats && norm->counters,
                 "kd6d_conv2d_fwd_norm: null pointer");
  KD6D_CHECK_ARG(norm->kind == KD6D_NORM_GROUP || norm->kind == KD6D_NORM_BATCH, "kd6d_conv2d_fwd_norm: bad kind %d", norm->kind);
  KD6D_CHECK_ARG(norm->act >= 0 && norm->act <= 2, "kd6d_conv2d_fwd_norm: bad act %d", norm->act);
  KD6D_CHECK_ARG(norm->kind != KD6D_NORM_BATCH || (norm->save_mean && norm->save_invstd),
                 "kd6d_conv2d_fwd_norm: BatchNorm needs save_mean / save_invstd");
  ConvParams p;
  LaunchPlan plan;
  if (!norm_fusable(g, dtype, norm->kind, norm->groups, p, plan)) {
    kd6d_set_error("kd6d_conv2d_fwd_norm: this geometry does not take the fused path (kd6d_conv2d_fwd_norm_fusable)");
    return KD6D_ERR_UNSUPPORTED;
  }
  unsigned int* timeouts = kd6d_ctx_timeouts_ptr();
  KD6D_CHECK_ARG(timeouts != nullptr, "kd6d_conv2d_fwd_norm: no barrier-timeout counter");
  p.src = x; p.wgt = w; p.dst = raw_out;
  p.ch_shift = bias; p.act = KD6D_ACT_NONE; p.out_f32 = 1;
  p.stats = reinterpret_cast<long long*>(norm->stats);
  p.norm_dst = norm->y; p.norm_gamma = norm->gamma; p.norm_beta = norm->beta;
  p.norm_ctr = norm->counters; p.norm_timeouts = timeouts;
  p.norm_eps = norm->eps; p.norm_act = norm->act;
  if (norm->kind == KD6D_NORM_BATCH) {
    p.stats_replicas = KD6D_BN_FUSED_REPLICAS;
    p.bn_inv_rows = 1.0f / (float)p.M;
    p.bn_momentum = norm->momentum;
    p.bn_unbias = p.M > 1 ? (float)p.M / (float)(p.M - 1) : 1.0f;
    p.bn_save_mean = norm->save_mean; p.bn_save_invstd = norm->save_invstd;
    p.bn_running_mean = norm->running_mean; p.bn_running_var = norm->running_var;
  } else {
    p.linear_tiles = 1;
  }
  dispatch_fwd(p, g, dtype, nullptr, 0, reinterpret_cast<hipStream_t>(stream));
  KD6D_CHECK_LAUNCH("kd6d_conv2d_fwd_norm");
  return KD6D_OK;
}

extern "C" int kd6d_conv2d_dgrad(const kd6d_conv_geom* g, int dtype, const void* dy, const void* wt,
                                 void* dx, int accumulate, void* stream) {
  int rc = check_geom(g, dtype, "kd6d_conv2d_dgrad");
  if (rc) return rc;
  KD6D_CHECK_ARG(dy && wt && dx, "kd6d_conv2d_dgrad: null tensor pointer");
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(g->cout % eg == 0, "kd6d_conv2d_dgrad: cout=%d must be a multiple of %d", g->cout, eg);
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.nseg = g->nseg; p.batch = g->batch; p.C = g->cout; p.N = g->cin;
  p.ks = g->ksize; p.stride = g->stride; p.pad = g->pad;
  p.K = g->ksize * g->ksize * g->cout;
  KD6D_CHECK_ARG(fill_segs(g, true, p.seg, &p.M), "kd6d_conv2d_dgrad: grid too large");
  p.src = dy; p.wgt = wt; p.dst = dx;
  p.residual = accumulate ? dx : nullptr;
  p.act = KD6D_ACT_NONE; p.out_f32 = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == KD6D_BF16) {
    if (!dispatch_smallc<MODE_DGRAD>(p, g, st) && !dispatch_halo_dgrad(p, g, st) && !dispatch_glds<MODE_DGRAD>(p, st))
      dispatch_igemm<bf16_t, MODE_DGRAD>(p, st);
  } else {
    dispatch_igemm<float, MODE_DGRAD>(p, st);
  }
  KD6D_CHECK_LAUNCH("kd6d_conv2d_dgrad");
  return KD6D_OK;
}

// norm_ops.hip: per-channel column sums of a (rows, C) tensor into PLANAR gradient accumulators
namespace kd6d_detail {
int colsum_grad_planar(int dtype, const void* x, int64_t rows, int C, long long* acc, long long acc_hi, void* stream);
}

namespace {
int wgrad_params(const kd6d_conv_geom* g, int dtype, int cu_budget, const char* who, WgradParams& p) {
  int rc = check_geom(g, dtype, who);
  if (rc) return rc;
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(g->cout % eg == 0, "%s: cout=%d must be a multiple of %d", who, g->cout, eg);
  memset(&p, 0, sizeof(p));
  p.nseg = g->nseg; p.batch = g->batch; p.Cin = g->cin; p.Cout = g->cout;
  p.ks = g->ksize; p.stride = g->stride; p.pad = g->pad;
  p.J = g->ksize * g->ksize * g->cin;
  KD6D_CHECK_ARG(fill_segs(g, false, p.seg, &p.M), "%s: grid too large", who);
  for (int s = 0; s < g->nseg; ++s)
    KD6D_CHECK_ARG(p.seg[s].dst_row0 == p.seg[s].m_begin, "%s: output levels must be packed back to back", who);
  const int ncu = cached_cu_count();
  KD6D_CHECK_ARG(cu_budget >= 0, "%s: cu_budget=%d", who, cu_budget);
  p.cu_budget = (cu_budget == 0 || cu_budget > ncu) ? ncu : cu_budget;
  return KD6D_OK;
}
void wgrad_dispatch(const WgradParams& p, const kd6d_conv_geom* g, int dtype, hipStream_t st) {
  if (dtype == KD6D_BF16) {
    if (!dispatch_wgrad_small(p, g, st)) dispatch_wgrad_tr(p, st);
  } else {
    dispatch_wgrad<float>(p, st);
  }
}
}  // namespace

extern "C" int kd6d_conv2d_wgrad_parts(const kd6d_conv_geom* g, int dtype, int with_bias, int cu_budget) {
  WgradParams p;
  int rc = wgrad_params(g, dtype, cu_budget, "kd6d_conv2d_wgrad_parts", p);
  if (rc) return rc;
  int parts = 0;
  p.plan_splits = &parts;
  // (the narrow-layer kernel takes no bias gradient: the same dispatch decision as the real call)
  static long long dummy_bias;
  p.dbias = with_bias ? &dummy_bias : nullptr;
  wgrad_dispatch(p, g, dtype, nullptr);
  KD6D_CHECK_ARG(parts >= 1, "kd6d_conv2d_wgrad_parts: no kernel takes this geometry");
  return parts;
}

extern "C" int kd6d_conv2d_wgrad(const kd6d_conv_geom* g, int dtype, const void* x, const void* dy,
                                 float* dw_slab, int64_t slab_floats, int64_t* dbias, int64_t acc_hi_stride,
                                 int cu_budget, void* stream) {
  WgradParams p;
  int rc = wgrad_params(g, dtype, cu_budget, "kd6d_conv2d_wgrad", p);
  if (rc) return rc;
  KD6D_CHECK_ARG(x && dy && dw_slab, "kd6d_conv2d_wgrad: null tensor pointer");
  KD6D_CHECK_ARG(!dbias || acc_hi_stride != 0, "kd6d_conv2d_wgrad: acc_hi_stride = 0 with a bias accumulator");
  p.x = x; p.dy = dy;
  p.dw = dw_slab; p.slab_floats = (long long)slab_floats;
  p.dbias = reinterpret_cast<long long*>(dbias); p.acc_hi = (long long)acc_hi_stride;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  g_wgrad_slab_short = false;
  wgrad_dispatch(p, g, dtype, st);
  KD6D_CHECK_ARG(!g_wgrad_slab_short, "kd6d_conv2d_wgrad: slab of %lld floats is too small (kd6d_conv2d_wgrad_parts x cout*k*k*cin)",
                 (long long)slab_floats);
  if (dtype != KD6D_BF16 && dbias) {           // exact-fp32 parity path: separate column-sum pass
    rc = kd6d_detail::colsum_grad_planar(dtype, dy, p.M, g->cout, p.dbias, p.acc_hi, stream);
    if (rc) return rc;
  }
  KD6D_CHECK_LAUNCH("kd6d_conv2d_wgrad");
  return KD6D_OK;
}

extern "C" int kd6d_pack_dgrad_weights(int dtype, const void* w_base, void* wt_base,
                                       const int32_t* desc_dev, int n_layers, int total_blocks,
                                       void* stream) {
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "kd6d_pack_dgrad_weights: bad dtype");
  KD6D_CHECK_ARG(w_base && wt_base && desc_dev && n_layers > 0 && total_blocks > 0,
                 "kd6d_pack_dgrad_weights: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == KD6D_BF16)
    hipLaunchKernelGGL(pack_dgrad_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, st,
                       reinterpret_cast<const bf16_t*>(w_base), reinterpret_cast<bf16_t*>(wt_base),
                       desc_dev, n_layers);
  else
    hipLaunchKernelGGL(pack_dgrad_kernel<float>, dim3(total_blocks), dim3(256), 0, st,
                       reinterpret_cast<const float*>(w_base), reinterpret_cast<float*>(wt_base),
                       desc_dev, n_layers);
  KD6D_CHECK_LAUNCH("kd6d_pack_dgrad_weights");
  return KD6D_OK;
}
