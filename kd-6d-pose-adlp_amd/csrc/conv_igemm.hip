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
#include <math.h>
#include <stdlib.h>

#include "kd6d_common.h"

namespace {

constexpr int kMaxSeg = KD6D_MAX_SEG;
enum { MODE_FWD = 0, MODE_DGRAD = 1 };

struct SegDev {
  int src_h, src_w;    // gather-source grid
  int dst_h, dst_w;    // destination grid (rows of the GEMM)
  int src_row0;        // first source row of the level
  int dst_row0;        // first destination row of the level
  int m_begin;         // first GEMM row index of the level
  int dst_hw;
};

struct ConvParams {
  int nseg, batch;
  int C;       // gather-source channels (k granularity)
  int N;       // result channels
  int ks, stride, pad;
  int K;       // ks*ks*C
  int M;       // total destination pixels
  int n_ctiles;
  SegDev seg[kMaxSeg];
  const void* src;
  const void* wgt;
  void* dst;
  const float* ch_scale;
  const float* ch_shift;
  const void* residual;
  const float* seg_scale;
  int act;
  int out_f32;
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  static constexpr int CHUNKS = 2;  // 32-deep chunks per 128-B row
  bf16x8_t v;
};
template <> struct Frag<float> {
  static constexpr int CHUNKS = 1;
  f32x4_t lo, hi;
};

__device__ __forceinline__ int lds_off(int row, int gran) {
  return row * 128 + ((gran ^ (row & 7)) << 4);
}

template <typename T>
__device__ __forceinline__ void load_frag(const char* tile, int row, int chunk, int q, Frag<T>& f);
template <>
__device__ __forceinline__ void load_frag<bf16_t>(const char* tile, int row, int chunk, int q,
                                                  Frag<bf16_t>& f) {
  f.v = *reinterpret_cast<const bf16x8_t*>(tile + lds_off(row, chunk * 4 + q));
}
template <>
__device__ __forceinline__ void load_frag<float>(const char* tile, int row, int /*chunk*/, int q,
                                                 Frag<float>& f) {
  f.lo = *reinterpret_cast<const f32x4_t*>(tile + lds_off(row, 2 * q));
  f.hi = *reinterpret_cast<const f32x4_t*>(tile + lds_off(row, 2 * q + 1));
}

__device__ __forceinline__ void mma(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x4_t& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(const Frag<float>& a, const Frag<float>& b, f32x4_t& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[0], b.lo[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[1], b.lo[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[2], b.lo[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[3], b.lo[3], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[0], b.hi[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[1], b.hi[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[2], b.hi[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[3], b.hi[3], acc, 0, 0, 0);
}

// XCD-aware, bijective remap of the linear workgroup id: consecutive remapped ids
// (which share an input pixel tile) land on one XCD / one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Decode GEMM row m -> (level fields) without dynamic indexing of the kernarg table.
struct RowInfo {
  int y, x;        // destination coordinates
  int src_h, src_w;
  int src_base;    // source row of (b, 0, 0)
  int dst_row;
  int seg;
};
__device__ __forceinline__ RowInfo decode_row(const ConvParams& p, int m) {
  RowInfo r;
  int mb = 0, hw = 1, dw = 1, sh = 0, sw = 0, s0 = 0, d0 = 0, sg = 0;
#pragma unroll
  for (int s = 0; s < kMaxSeg; ++s) {
    if (s < p.nseg && m >= p.seg[s].m_begin) {
      mb = p.seg[s].m_begin; hw = p.seg[s].dst_hw; dw = p.seg[s].dst_w;
      sh = p.seg[s].src_h; sw = p.seg[s].src_w; s0 = p.seg[s].src_row0;
      d0 = p.seg[s].dst_row0; sg = s;
    }
  }
  const int local = m - mb;
  const int b = local / hw;
  const int rem = local - b * hw;
  r.y = rem / dw;
  r.x = rem - r.y * dw;
  r.src_h = sh; r.src_w = sw;
  r.src_base = s0 + b * sh * sw;
  r.dst_row = d0 + local;
  r.seg = sg;
  if (m >= p.M) { r.src_h = 0; r.src_w = 0; }
  return r;
}

template <typename T, int BP, int BC, int WP, int WC, int MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
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

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_c = wg % p.n_ctiles;
  const int tile_p = wg / p.n_ctiles;
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

  auto issue_loads = [&](int kt) {
    const int kk = kt * BK + gcol * EG;
    const bool kvalid = kk < p.K;
    const int tap = kk / p.C;
    const int cc = kk - tap * p.C;
    const int ky = tap / p.ks;
    const int kx = tap - ky * p.ks;
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
        v = *reinterpret_cast<const u32x4_t*>(src + off);
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

  // ---- epilogue: lane owns pixel (lane&15), channels (lane>>4)*4 .. +3 -------
  const bool vec_ok = (p.N & 3) == 0;
#pragma unroll
  for (int q = 0; q < PI; ++q) {
    const int m = m0 + wp * (BP / WP) + q * 16 + fr;
    if (m >= p.M) continue;
    int drow = m, sg = 0;
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s) {
      if (s < p.nseg && m >= p.seg[s].m_begin) {
        drow = p.seg[s].dst_row0 + (m - p.seg[s].m_begin);
        sg = s;
      }
    }
    float sscale = 1.f;
    if (p.seg_scale) sscale = p.seg_scale[sg];
#pragma unroll
    for (int c = 0; c < CI; ++c) {
      const int n = n0 + wc * (BC / WC) + c * 16 + fq * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[c][q][0], acc[c][q][1], acc[c][q][2], acc[c][q][3]};
      const size_t o = (size_t)drow * (size_t)p.N + (size_t)n;
      if (vec_ok) {
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
        if (p.residual) {
          if (p.out_f32) {
            const f32x4_t r4 = *reinterpret_cast<const f32x4_t*>(
                reinterpret_cast<const float*>(p.residual) + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += r4[r];
          } else {
            const T* rp = reinterpret_cast<const T*>(p.residual) + o;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(rp[r]);
          }
        }
        if (p.out_f32) {
          *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.dst) + o) =
              f32x4_t{v[0], v[1], v[2], v[3]};
        } else if (sizeof(T) == 4) {
          *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.dst) + o) =
              f32x4_t{v[0], v[1], v[2], v[3]};
        } else {
          u32x2_t pk;
          pk.x = pack_bf16x2(v[0], v[1]);
          pk.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(p.dst) + o) = pk;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= p.N) continue;
          float t = v[r];
          if (p.ch_scale) t *= p.ch_scale[n + r];
          if (p.ch_shift) t += p.ch_shift[n + r];
          t *= sscale;
          if (p.act == KD6D_ACT_LEAKY) t = t > 0.f ? t : 0.1f * t;
          else if (p.act == KD6D_ACT_RELU) t = fmaxf(t, 0.f);
          if (p.residual) {
            t += p.out_f32 ? reinterpret_cast<const float*>(p.residual)[o + r]
                           : to_f32<T>(reinterpret_cast<const T*>(p.residual)[o + r]);
          }
          if (p.out_f32) reinterpret_cast<float*>(p.dst)[o + r] = t;
          else reinterpret_cast<T*>(p.dst)[o + r] = from_f32<T>(t);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Weight gradient:  dW[n][j] += sum_m dY[m][n] * im2col(X)[m][j],  j = (tap, ci).
// Both operands are reduced along the pixel axis, which is the slow axis of NHWC,
// so tiles are staged TRANSPOSED into the same 128-B-row LDS image (row = channel,
// k = pixel) and consumed by the identical fragment reads.  fp32 atomics into dW.
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
  float* dw;
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
    sh = 0; sw = 0;
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s) {
      if (s < p.nseg && m >= p.seg[s].m_begin) {
        mb = p.seg[s].m_begin; hw = p.seg[s].dst_hw; dw = p.seg[s].dst_w;
        sh = p.seg[s].src_h; sw = p.seg[s].src_w; s0 = p.seg[s].src_row0;
      }
    }
    const int local = m - mb;
    const int b = local / hw;
    const int rem = local - b * hw;
    y = rem / dw;
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

  // D rows = out channel n (4 per lane), D cols = j (lane&15)
#pragma unroll
  for (int a = 0; a < NI; ++a) {
#pragma unroll
    for (int b = 0; b < JI; ++b) {
      const int j = j0 + wj * (BJ / WJ) + b * 16 + fr;
      const int n = n0 + wn * (BN / WN) + a * 16 + fq * 4;
      if (j >= p.J) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r < p.Cout) atomicAdd(p.dw + (size_t)(n + r) * (size_t)p.J + (size_t)j, acc[a][b][r]);
      }
    }
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
// adds 256 contiguous bytes of one dW row.  The split count balances the k-loop against the
// ~1.3 TB/s fp32-atomic rate of the memory side (launch_wgrad_tr).
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

  auto issue_loads = [&](int mstep) {
    const int m = mstep + prow;
    const bool mv = m < m_hi;
    // pixel decode (level, image, y, x) once per k-step
    int mb = 0, hw = 1, dw = 1, sh = 0, sw = 0, s0 = 0;
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s) {
      if (s < p.nseg && m >= p.seg[s].m_begin) {
        mb = p.seg[s].m_begin; hw = p.seg[s].dst_hw; dw = p.seg[s].dst_w;
        sh = p.seg[s].src_h; sw = p.seg[s].src_w; s0 = p.seg[s].src_row0;
      }
    }
    const int local = m - mb;
    const int b = local / hw;
    const int rem = local - b * hw;
    const int y = rem / dw;
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

  // ---- epilogue: [n][j] fp32 image in LDS, then 256-B contiguous atomic rows ----
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
  for (int nl = wave; nl < BN; nl += 4) {
    const int n = n0 + nl;
    if (n >= p.Cout) break;
#pragma unroll
    for (int h = 0; h < BJ / 64; ++h) {
      const int jl = h * 64 + lane;
      const int j = j0 + jl;
      if (j < p.J) atomicAdd(p.dw + (size_t)n * (size_t)p.J + (size_t)j, et[nl * EP + jl]);
    }
  }
}

// ---------------------------------------------------------------------------
// dgrad weight packing: wt[ci][ky][kx][co] <- w[co][ky][kx][ci], all layers at once.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const T* __restrict__ w, T* __restrict__ wt,
                                                         const int* __restrict__ desc, int n_layers) {
  // find layer: desc[l*6+5] = first block of layer l (ascending)
  int l = 0;
  for (int i = 1; i < n_layers; ++i)
    if ((int)blockIdx.x >= desc[i * 6 + 5]) l = i;
  const int w_off = desc[l * 6 + 0], wt_off = desc[l * 6 + 1];
  const int cout = desc[l * 6 + 2], cin = desc[l * 6 + 3], ks = desc[l * 6 + 4];
  const int blk = blockIdx.x - desc[l * 6 + 5];
  const int taps = ks * ks;
  const int total = cout * taps * cin;
  // each thread produces one element of wt (co fastest => coalesced writes)
  for (int e = blk * 256 * 8 + threadIdx.x; e < min(total, (blk + 1) * 256 * 8); e += 256) {
    const int co = e % cout;
    const int rest = e / cout;
    const int tap = rest % taps;
    const int ci = rest / taps;
    wt[(size_t)wt_off + e] = w[(size_t)w_off + ((size_t)co * taps + tap) * cin + ci];
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
bool fill_segs(const kd6d_conv_geom* g, bool dgrad, SegDev* seg, int* M_out) {
  int m = 0;
  for (int s = 0; s < g->nseg; ++s) {
    const kd6d_seg& gs = g->seg[s];
    SegDev& d = seg[s];
    if (!dgrad) {
      d.src_h = gs.in_h; d.src_w = gs.in_w; d.dst_h = gs.out_h; d.dst_w = gs.out_w;
      d.src_row0 = gs.in_row0; d.dst_row0 = gs.out_row0;
    } else {
      d.src_h = gs.out_h; d.src_w = gs.out_w; d.dst_h = gs.in_h; d.dst_w = gs.in_w;
      d.src_row0 = gs.out_row0; d.dst_row0 = gs.in_row0;
    }
    d.dst_hw = d.dst_h * d.dst_w;
    d.m_begin = m;
    if (d.src_h > 32767 || d.src_w > 32767 || d.src_h < 0 || d.src_w < 0) return false;
    m += g->batch * d.dst_hw;
  }
  *M_out = m;
  return true;
}

int check_geom(const kd6d_conv_geom* g, int dtype, const char* who) {
  KD6D_CHECK_ARG(g != nullptr, "%s: null geometry", who);
  KD6D_CHECK_ARG(g->nseg >= 1 && g->nseg <= kMaxSeg, "%s: nseg=%d out of range", who, g->nseg);
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "%s: bad dtype %d", who, dtype);
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(g->cin > 0 && g->cin % eg == 0, "%s: cin=%d must be a multiple of %d", who, g->cin, eg);
  KD6D_CHECK_ARG(g->cout > 0, "%s: cout=%d", who, g->cout);
  KD6D_CHECK_ARG(g->ksize >= 1 && g->ksize <= 7 && g->stride >= 1 && g->stride <= 4 && g->pad >= 0,
                 "%s: bad ksize/stride/pad %d/%d/%d", who, g->ksize, g->stride, g->pad);
  KD6D_CHECK_ARG(g->batch >= 1, "%s: batch=%d", who, g->batch);
  for (int s = 0; s < g->nseg; ++s) {
    const kd6d_seg& q = g->seg[s];
    KD6D_CHECK_ARG(q.in_h > 0 && q.in_w > 0 && q.out_h > 0 && q.out_w > 0, "%s: empty level %d", who, s);
    KD6D_CHECK_ARG(q.out_h == (q.in_h + 2 * g->pad - g->ksize) / g->stride + 1 &&
                       q.out_w == (q.in_w + 2 * g->pad - g->ksize) / g->stride + 1,
                   "%s: level %d output grid %dx%d inconsistent with input %dx%d", who, s, q.out_h,
                   q.out_w, q.in_h, q.in_w);
  }
  return KD6D_OK;
}

template <typename T, int BP, int BC, int WP, int WC, int MODE>
void launch_igemm(const ConvParams& p, hipStream_t st) {
  ConvParams q = p;
  q.n_ctiles = (p.N + BC - 1) / BC;
  const int ptiles = (p.M + BP - 1) / BP;
  const size_t lds = (size_t)(BP + BC) * 128 * 2;
  auto kern = conv_igemm_kernel<T, BP, BC, WP, WC, MODE>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(ptiles * q.n_ctiles), dim3(256), lds, st, q);
}

template <typename T, int MODE>
void dispatch_igemm(const ConvParams& p, hipStream_t st) {
  const int N = p.N, M = p.M;
  auto nblocks = [&](int bp, int bc) { return ((M + bp - 1) / bp) * ((N + bc - 1) / bc); };
  if (N <= 16) {
    if (nblocks(256, 16) >= 512) launch_igemm<T, 256, 16, 4, 1, MODE>(p, st);
    else launch_igemm<T, 64, 16, 4, 1, MODE>(p, st);
  } else if (N <= 32) {
    if (nblocks(256, 32) >= 512) launch_igemm<T, 256, 32, 4, 1, MODE>(p, st);
    else launch_igemm<T, 64, 32, 4, 1, MODE>(p, st);
  } else if (N <= 64) {
    if (nblocks(128, 64) >= 384) launch_igemm<T, 128, 64, 2, 2, MODE>(p, st);
    else launch_igemm<T, 64, 64, 2, 2, MODE>(p, st);
  } else {
    if (nblocks(128, 128) >= 384) launch_igemm<T, 128, 128, 2, 2, MODE>(p, st);
    else if (nblocks(128, 64) >= 384) launch_igemm<T, 128, 64, 2, 2, MODE>(p, st);
    else launch_igemm<T, 64, 64, 2, 2, MODE>(p, st);
  }
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
  // time ~ (steps/S) * t_step + S * |dW| / (fp32 atomic rate 1.3 TB/s), t_step ~ 1.6 us measured
  //   => S* = sqrt(steps * t_step * rate / |dW|); at most 2 workgroups per CU, because many
  //   workgroups adding into one small dW are contention-bound (measured: 2048 -> 512 = -25 %)
  static const double scale = []() {
    const char* e = getenv("KD6D_WGRAD_SPLIT_SCALE");
    return e ? atof(e) : 1.0;
  }();
  const double dw_bytes = (double)p.Cout * (double)p.J * 4.0;
  int splits = (int)(scale * sqrt((double)steps_total * 2.08e6 / dw_bytes) + 0.5);
  if (splits > 512 / tiles) splits = 512 / tiles;
  if (splits > steps_total / 2) splits = steps_total / 2;
  if (splits < 1) splits = 1;
  const int steps_per = (steps_total + splits - 1) / splits;
  q.m_chunk = steps_per * BKM;
  splits = (p.M + q.m_chunk - 1) / q.m_chunk;
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

extern "C" int kd6d_conv2d_fwd(const kd6d_conv_geom* g, int dtype, const void* x, const void* w,
                               void* y, const float* ch_scale, const float* ch_shift, int act,
                               const void* residual, const float* seg_scale, int out_f32,
                               void* stream) {
  int rc = check_geom(g, dtype, "kd6d_conv2d_fwd");
  if (rc) return rc;
  KD6D_CHECK_ARG(x && w && y, "kd6d_conv2d_fwd: null tensor pointer");
  KD6D_CHECK_ARG(act >= 0 && act <= 2, "kd6d_conv2d_fwd: bad act %d", act);
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.nseg = g->nseg; p.batch = g->batch; p.C = g->cin; p.N = g->cout;
  p.ks = g->ksize; p.stride = g->stride; p.pad = g->pad;
  p.K = g->ksize * g->ksize * g->cin;
  KD6D_CHECK_ARG(fill_segs(g, false, p.seg, &p.M), "kd6d_conv2d_fwd: grid too large");
  p.src = x; p.wgt = w; p.dst = y;
  p.ch_scale = ch_scale; p.ch_shift = ch_shift; p.residual = residual; p.seg_scale = seg_scale;
  p.act = act; p.out_f32 = out_f32 ? 1 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == KD6D_BF16) dispatch_igemm<bf16_t, MODE_FWD>(p, st);
  else dispatch_igemm<float, MODE_FWD>(p, st);
  KD6D_CHECK_LAUNCH("kd6d_conv2d_fwd");
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
  if (dtype == KD6D_BF16) dispatch_igemm<bf16_t, MODE_DGRAD>(p, st);
  else dispatch_igemm<float, MODE_DGRAD>(p, st);
  KD6D_CHECK_LAUNCH("kd6d_conv2d_dgrad");
  return KD6D_OK;
}

extern "C" int kd6d_conv2d_wgrad(const kd6d_conv_geom* g, int dtype, const void* x, const void* dy,
                                 float* dw, void* stream) {
  int rc = check_geom(g, dtype, "kd6d_conv2d_wgrad");
  if (rc) return rc;
  KD6D_CHECK_ARG(x && dy && dw, "kd6d_conv2d_wgrad: null tensor pointer");
  const int eg = dtype == KD6D_BF16 ? 8 : 4;
  KD6D_CHECK_ARG(g->cout % eg == 0, "kd6d_conv2d_wgrad: cout=%d must be a multiple of %d", g->cout, eg);
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.nseg = g->nseg; p.batch = g->batch; p.Cin = g->cin; p.Cout = g->cout;
  p.ks = g->ksize; p.stride = g->stride; p.pad = g->pad;
  p.J = g->ksize * g->ksize * g->cin;
  KD6D_CHECK_ARG(fill_segs(g, false, p.seg, &p.M), "kd6d_conv2d_wgrad: grid too large");
  for (int s = 0; s < g->nseg; ++s)
    KD6D_CHECK_ARG(p.seg[s].dst_row0 == p.seg[s].m_begin,
                   "kd6d_conv2d_wgrad: output levels must be packed back to back");
  p.x = x; p.dy = dy; p.dw = dw;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == KD6D_BF16) dispatch_wgrad_tr(p, st);
  else dispatch_wgrad<float>(p, st);
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
