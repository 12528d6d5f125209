// Shared by the convolution kernels of libkd6d.so (conv_igemm.hip: register-staged / LDS-DMA / resident-patch /
// weight-gradient kernels; conv_halo.hip: the 3x3 halo-patch kernel): GEMM geometry, LDS tile image, fragment
// loads, the common epilogue with its fused normalisation statistics, LDS-DMA helpers and the host-side geometry
// checks.  Everything is static / inline: each translation unit gets its own copy.
#pragma once
#include <math.h>
#include <stdlib.h>

#include "kd6d_barrier.h"
#include "kd6d_common.h"
#include "kd6d_det.h"

namespace kd6d_detail {


constexpr int kMaxSeg = KD6D_MAX_SEG;
enum { MODE_FWD = 0, MODE_DGRAD = 1 };

struct SegDev {
  int src_h, src_w;    // gather-source grid
  int dst_h, dst_w;    // destination grid (rows of the GEMM)
  int src_row0;        // first source row of the level
  int dst_row0;        // first destination row of the level
  int m_begin;         // first GEMM row index of the level
  int dst_hw;
  float inv_hw, inv_w; // 1 / dst_hw, 1 / dst_w: row decode without integer division (GEMM rows < 2^24)
};

// floor(x / d) for 0 <= x < 2^24 from inv = 1.0f / d: float estimate, one correction step either way
__device__ __forceinline__ int fast_div(int x, int d, float inv) {
  int q = (int)((float)x * inv);
  q += ((q + 1) * d <= x) ? 1 : 0;
  q -= (q * d > x) ? 1 : 0;
  return q;
}

struct ConvParams {
  int nseg, batch;
  int C;       // gather-source channels (k granularity)
  int N;       // result channels
  int ks, stride, pad;
  int K;       // ks*ks*C
  int M;       // total destination pixels
  int n_ctiles;
  int n_ptiles;
  int p_fastest;   // workgroup order: pixel tiles fastest (weight tile shared inside an XCD) instead of channel tiles
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
  long long* stats;    // optional fused statistics of the stored values: accumulators {lo, hi} (kd6d_det.h; conv_epilogue_stats)
  int stats_groups;    // 0: per channel {sum[N], sumsq[N]} (BatchNorm); G > 0: {sum, sumsq} per (level, image, group)
  float* slab;         // split-K: fp32 partial tiles, slab[split][M][N] (kd6d_conv2d_fwd workspace)
  int nk_split;        // k-steps per split
  int stats_cpg_shift; // log2(channels per group): 2 or 3
  int stats_skip;      // group statistics: bit s set = level s is NOT summed by the epilogue (its rows do not fall into whole
                       // 16-row fragments of one image: H*W or its first row not a multiple of 16); the host sums those levels
                       // with a separate gn_stats launch behind the convolution (set_stats / stats_followup)
  // ---- normalisation + activation fused behind the convolution (conv_epilogue_norm; kd6d_conv2d_fwd_norm) ----
  void* norm_dst;              // non-null: once the statistics are complete across workgroups, act(norm(v)) goes here (T)
  const float* norm_gamma;
  const float* norm_beta;
  unsigned int* norm_ctr;      // pre-zeroed words: BatchNorm KD6D_BARRIER_WORDS; GroupNorm one per (level, image, channel tile)
  unsigned int* norm_timeouts; // the library's counter of barrier waits that gave up
  float norm_eps;
  int norm_act;
  int stats_replicas;          // BatchNorm, fused: rows {sum[N], sumsq[N]} of `stats` (workgroup b adds to row b % replicas)
  int linear_tiles;            // 1: workgroup id == tile id (no XCD remap): tiles that wait for each other are dispatched together
  float bn_inv_rows, bn_momentum, bn_unbias;
  float* bn_save_mean;
  float* bn_save_invstd;
  float* bn_running_mean;
  float* bn_running_var;
  // ---- train-mode BatchNorm + activation of the PREVIOUS block applied while this convolution loads its input
  // (conv_igemm_kernel<..., XF = true>; kd6d_conv2d_fwd_bn_in): src is that block's fp32 conv output ----
  const long long* xf_sum;     // xf_replicas rows of {sum[C], sumsq[C]} accumulators (what the previous launch's epilogue added up)
  const float* xf_gamma;
  const float* xf_beta;
  float* xf_save_mean;         // outputs for the backward pass of the previous block (written by workgroup 0)
  float* xf_save_invstd;
  float* xf_running_mean;      // updated in place by workgroup 0 (optional)
  float* xf_running_var;
  void* xf_z;                  // optional: the activation act(bn(src)) as T, (rows_in, C) -- what the weight gradient reads
  float xf_eps, xf_inv_rows, xf_momentum, xf_unbias;
  int xf_replicas, xf_act;
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

// (the kernels compiled without the fused-normalisation epilogue never read the fields behind stats_cpg_shift: their
//  scalar-register budget is the round-2 one -- two more SGPRs cost conv3x3_smallc_kernel<1,1> a wave per SIMD)
template <bool NORM>
__device__ __forceinline__ int tile_of_workgroup(const ConvParams& p, int bid, int nwg) {
  if constexpr (NORM) return p.linear_tiles ? bid : xcd_remap(bid, nwg);
  else return xcd_remap(bid, nwg);
}

// (level, image) key of GEMM row m: level * batch + image.  Monotone in m (levels are packed level-major, image-major).
__device__ __forceinline__ int stats_key(const ConvParams& p, int m, int* level = nullptr) {
  int mb = 0, hw = 1, sg = 0;
  float inv = 1.f;
#pragma unroll
  for (int s = 0; s < kMaxSeg; ++s)
    if (s < p.nseg && m >= p.seg[s].m_begin) {
      mb = p.seg[s].m_begin; hw = p.seg[s].dst_hw; sg = s; inv = p.seg[s].inv_hw;
    }
  if (level) *level = sg;
  return sg * p.batch + fast_div(m - mb, hw, inv);
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
  float ihw = 1.f, iw = 1.f;
#pragma unroll
  for (int s = 0; s < kMaxSeg; ++s) {
    if (s < p.nseg && m >= p.seg[s].m_begin) {
      mb = p.seg[s].m_begin; hw = p.seg[s].dst_hw; dw = p.seg[s].dst_w;
      sh = p.seg[s].src_h; sw = p.seg[s].src_w; s0 = p.seg[s].src_row0;
      d0 = p.seg[s].dst_row0; sg = s; ihw = p.seg[s].inv_hw; iw = p.seg[s].inv_w;
    }
  }
  const int local = m - mb;
  const int b = fast_div(local, hw, ihw);
  const int rem = local - b * hw;
  r.y = fast_div(rem, dw, iw);
  r.x = rem - r.y * dw;
  r.src_h = sh; r.src_w = sw;
  r.src_base = s0 + b * sh * sw;
  r.dst_row = d0 + local;
  r.seg = sg;
  if (m >= p.M) { r.src_h = 0; r.src_w = 0; }
  return r;
}

// Sum over the 16 lanes of a DPP row (lanes 16k..16k+15), result in every lane.  VALU-only (v_add_f32
// with dpp modifiers): the ds_bpermute form of __shfl_xor costs an LDS round trip per step.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// Fused normalisation statistics of a conv output (replaces a separate pass over the fp32 tensor).
// acc holds the FINAL values (what was stored).  Every cross-wave / cross-workgroup addition is either a fixed-order
// fp32 sum or an integer atomic on a fixed-point image (kd6d_det.h), so the statistics are BITWISE reproducible
// whatever order waves and workgroups retire in.  The fixed-point conversions sit in two short rolled loops -- not in
// the unrolled accumulator loops: inlined there (64-128 sites per kernel variant) they tripled the build time.
//   Per-channel mode (BatchNorm batch statistics): registers -> 16-lane DPP sum -> one LDS slot per (wave row, channel),
//   plain stores -> a thread per channel adds the WP slots in order -> one global integer atomic per channel and workgroup.
//   Group mode (GroupNorm): a 16-pixel fragment normally lies inside one (level, image): its 4-channel lane sums are
//   DPP-summed over the fragment's rows and staged (one entry per writer lane), a rolled loop adds the entries to the LDS
//   accumulator table [image of the tile][group of the tile]; one flush of the table per workgroup.  Levels whose rows do
//   not fall into whole fragments of one image (H*W not a multiple of 16: the 2 x 2 level of 256^2 crops, 15 x 20 and
//   4 x 5 of full frames) are skipped here (ConvParams::stats_skip) and summed by a small gn_stats launch the host
//   issues behind the convolution -- handling them in this epilogue cost the halo tiles 10-30 registers.
// `red_f32`: the dead staging buffers.
template <int BP, int BC, int WP, int WC, bool NORM = false>
__device__ __forceinline__ void conv_epilogue_stats(const ConvParams& p, f32x4_t (&acc)[BC / WC / 16][BP / WP / 16],
                                                    int m0, int n0, int wp, int wc, int lane, float* red_f32) {
  constexpr int PI = BP / WP / 16;
  constexpr int CI = BC / WC / 16;
  const int fr = lane & 15;
  const int fq = lane >> 4;
  if (p.stats_groups == 0) {
    float* slot = red_f32;                 // [WP][2][BC]: every slot is written exactly once
    __syncthreads();                       // staging buffers are dead from here on
#pragma unroll
    for (int c = 0; c < CI; ++c) {
      const int nl = wc * (BC / WC) + c * 16 + fq * 4;
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < PI; ++q) {
        const int m = m0 + wp * (BP / WP) + q * 16 + fr;
        const bool ok = m < p.M && n0 + nl < p.N;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = ok ? acc[c][q][r] : 0.f;
          s1[r] += v;
          s2[r] += v * v;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s1[r] = row16_sum(s1[r]);
        s2[r] = row16_sum(s2[r]);
      }
      if (fr == 0) {
        *reinterpret_cast<f32x4_t*>(slot + (wp * 2 + 0) * BC + nl) = f32x4_t{s1[0], s1[1], s1[2], s1[3]};
        *reinterpret_cast<f32x4_t*>(slot + (wp * 2 + 1) * BC + nl) = f32x4_t{s2[0], s2[1], s2[2], s2[3]};
      }
    }
    __syncthreads();
    // fused normalisation: replica rows (hundreds of workgroups adding into one address retire one after the other,
    // ~25 ns each) and RETURNING atomics (visible before this workgroup arrives at the barrier, kd6d_barrier.h)
    int rep = 0;
    if constexpr (NORM) rep = p.stats_replicas > 1 ? (int)(blockIdx.x % (unsigned)p.stats_replicas) : 0;
    for (int i = threadIdx.x; i < 2 * BC; i += blockDim.x) {
      const int which = i / BC, nl = i - which * BC;
      if (n0 + nl < p.N) {
        float t = slot[which * BC + nl];
#pragma unroll
        for (int w = 1; w < WP; ++w) t += slot[(w * 2 + which) * BC + nl];
        const det_words dw = det_split<KD6D_DET_ACT>(t);
        long long* o = p.stats + ((size_t)(rep * 2 + which) * p.N + n0 + nl) * 2;
        if (NORM && p.norm_dst) det_add_words_performed(o, dw.lo, dw.hi);
        else det_add_words(o, dw.lo, dw.hi);
      }
    }
    return;
  }
  // ---- group mode ----
  constexpr int NW = WP * WC;
  constexpr int NENT = NW * PI * CI * 4;   // staged entries: (wave, fragment q, channel tile c, lane quarter fq)
  const int G = p.stats_groups;
  const int cs = p.stats_cpg_shift;        // 4 or 8 channels per group: a lane's 4 aligned channels share one
  const int GT = BC >> cs;                 // groups touched by this channel tile
  const int m_last = (m0 + BP < p.M ? m0 + BP : p.M) - 1;
  const int key_lo = stats_key(p, m0);
  const int nkeys = stats_key(p, m_last) - key_lo + 1;
  const int tab = nkeys * GT * 2;          // accumulators {sum, sumsq} per (key, group): 16 bytes each
  // LDS image: accumulator table | staged entries | fragment keys
  long long* table = reinterpret_cast<long long*>(red_f32);
  f32x2_t* ent = reinterpret_cast<f32x2_t*>(table + 2 * tab);
  int* fragkey = reinterpret_cast<int*>(ent + NENT);
  const int wave = wp * WC + wc;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * tab; i += blockDim.x) table[i] = 0;
  // phase A (accumulators live): a fragment of 16 rows lies inside ONE image of a level the epilogue sums (the host
  // leaves the levels where that does not hold to a separate launch: ConvParams::stats_skip) or contributes nothing: its
  // 4-channel lane sums are DPP-summed over the rows and staged, one entry per 4-channel slice, plain stores
#pragma unroll
  for (int q = 0; q < PI; ++q) {
    const int mf = m0 + wp * (BP / WP) + q * 16;         // first row of the fragment (a multiple of 16)
    int level = 0;
    const int key0 = stats_key(p, mf < p.M ? mf : p.M - 1, &level) - key_lo;
    const bool live = mf < p.M && !((p.stats_skip >> level) & 1);
    if (lane == 0 && wc == 0) fragkey[wp * PI + q] = live ? key0 : -1;
    const bool mok = live && mf + fr < p.M;
#pragma unroll
    for (int c = 0; c < CI; ++c) {
      const int nl = wc * (BC / WC) + c * 16 + fq * 4;
      const bool ok = mok && n0 + nl < p.N;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = ok ? acc[c][q][r] : 0.f;
        s1 += v;
        s2 += v * v;
      }
      s1 = row16_sum(s1);
      s2 = row16_sum(s2);
      if (fr == 0) ent[((wave * PI + q) * CI + c) * 4 + fq] = f32x2_t{s1, s2};
    }
  }
  __syncthreads();
  // phase B (accumulators dead): the staged entries -> the accumulator table, integer LDS atomics (order-free)
  for (int e = threadIdx.x; e < NENT; e += blockDim.x) {
    const int efq = e & 3, ec = (e >> 2) % CI, eq = ((e >> 2) / CI) % PI, ew = (e >> 2) / (CI * PI);
    const int ewp = ew / WC, ewc = ew - ewp * WC;
    const int key = fragkey[ewp * PI + eq];
    if (key < 0) continue;
    const int nl = ewc * (BC / WC) + ec * 16 + efq * 4;
    const f32x2_t v = ent[e];
    long long* o2 = table + ((key * GT + (nl >> cs)) << 2);
    det_add_lds<KD6D_DET_ACT>(o2, v[0]);
    det_add_lds<KD6D_DET_ACT>(o2 + 2, v[1]);
  }
  __syncthreads();
  const int g_first = n0 >> cs;
  const int gts = 31 - __clz(GT * 2);      // GT * 2 is a power of two
  for (int i = threadIdx.x; i < tab; i += blockDim.x) {
    const int k = i >> gts, rem = i & (GT * 2 - 1);
    const int gl = rem >> 1;
    if (g_first + gl < G) {
      long long* o = p.stats + (((size_t)(key_lo + k) * G + g_first + gl) * 2 + (rem & 1)) * 2;
      if (NORM && p.norm_dst) det_add_words_performed(o, table[2 * i], table[2 * i + 1]);
      else det_add_words(o, table[2 * i], table[2 * i + 1]);
    }
  }
}

// Epilogue shared by the register-staged and the LDS-DMA kernels: lane owns pixel (lane&15),
// channels (lane>>4)*4 .. +3 of every 16x16 accumulator tile.
template <typename T, int BP, int BC, int WP, int WC, bool WIDE = true, bool NORM = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4_t (&acc)[BC / WC / 16][BP / WP / 16],
                                              int m0, int n0, int wp, int wc, int lane, float* smem_f32) {
  constexpr int PI = BP / WP / 16;
  constexpr int CI = BC / WC / 16;
  const int fr = lane & 15;
  const int fq = lane >> 4;
  const bool vec_ok = (p.N & 3) == 0;
  // per-channel epilogue parameters of this lane's channels, loaded ONCE (they do not depend on the pixel):
  // identity where absent, so the pixel loop below is branch-free in them
  constexpr bool HOIST = CI <= 4;
  f32x4_t hsc[HOIST ? CI : 1], hsh[HOIST ? CI : 1];
  if (HOIST && vec_ok) {
#pragma unroll
    for (int c = 0; c < CI; ++c) {
      const int n = n0 + wc * (BC / WC) + c * 16 + fq * 4;
      hsc[c] = f32x4_t{1.f, 1.f, 1.f, 1.f};
      hsh[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (n < p.N) {
        if (p.ch_scale) hsc[c] = *reinterpret_cast<const f32x4_t*>(p.ch_scale + n);
        if (p.ch_shift) hsh[c] = *reinterpret_cast<const f32x4_t*>(p.ch_shift + n);
      }
    }
  }
  // bf16 result, N % 8 == 0: lanes (fq, fq ^ 1) swap one of their two 4-channel groups of a channel-tile pair, so that
  // each holds 8 consecutive channels and stores 16 B (16 rows x 64 B per wave instruction instead of 16 x 32 B, half
  // the store instructions -- the epilogue of these launches is store-issue-bound); the residual is then read 16 B
  // wide as well.  Arithmetic and rounding are those of the 4-channel path below.
  // (WIDE = false: the resident-patch kernel, VALU-bound already -- measured 31 -> 36 us on the teacher's stage-1 3x3)
  if constexpr (WIDE && sizeof(T) == 2 && HOIST && (CI % 2 == 0)) {
    if (!p.out_f32 && (p.N & 7) == 0 && !(p.stats && p.residual)) {
      const bool odd = fq & 1;
      const int partner = (lane ^ 16) << 2;
#pragma unroll
      for (int q = 0; q < PI; ++q) {
        const int m = m0 + wp * (BP / WP) + q * 16 + fr;
        const bool row_ok = m < p.M;
        int drow = m, sg = 0;
#pragma unroll
        for (int s = 0; s < kMaxSeg; ++s) {
          if (s < p.nseg && m >= p.seg[s].m_begin) {
            drow = p.seg[s].dst_row0 + (m - p.seg[s].m_begin);
            sg = s;
          }
        }
        float sscale = 1.f;
        if (p.seg_scale && row_ok) sscale = p.seg_scale[sg];
#pragma unroll
        for (int cp = 0; cp < CI / 2; ++cp) {
          f32x4_t va0, va1;       // this lane's 4 channels of tiles 2cp and 2cp + 1
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t0 = (acc[2 * cp][q][r] * hsc[2 * cp][r] + hsh[2 * cp][r]) * sscale;
            float t1 = (acc[2 * cp + 1][q][r] * hsc[2 * cp + 1][r] + hsh[2 * cp + 1][r]) * sscale;
            if (p.act == KD6D_ACT_LEAKY) {
              t0 = t0 > 0.f ? t0 : 0.1f * t0;
              t1 = t1 > 0.f ? t1 : 0.1f * t1;
            } else if (p.act == KD6D_ACT_RELU) {
              t0 = fmaxf(t0, 0.f);
              t1 = fmaxf(t1, 0.f);
            }
            va0[r] = t0;
            va1[r] = t1;
          }
          if (p.stats) {          // statistics of the stored values, in this lane's ORIGINAL channels (no residual here)
            acc[2 * cp][q] = va0;
            acc[2 * cp + 1][q] = va1;
          }
          // even fq keeps tile 2cp and takes the partner's (fq + 1) group of it; odd fq keeps tile 2cp + 1
          const f32x4_t send = odd ? va0 : va1;
          f32x4_t got;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            got[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(send[r])));
          f32x4_t lo = odd ? got : va0, hi = odd ? va1 : got;
          const int n = n0 + wc * (BC / WC) + (2 * cp + (odd ? 1 : 0)) * 16 + (fq & ~1) * 4;
          if (!row_ok || n >= p.N) continue;
          const size_t o = (size_t)drow * (size_t)p.N + (size_t)n;
          if (p.residual) {
            const u32x4_t r8 = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(p.residual) + o);
            lo[0] += __uint_as_float(r8.x << 16); lo[1] += __uint_as_float(r8.x & 0xffff0000u);
            lo[2] += __uint_as_float(r8.y << 16); lo[3] += __uint_as_float(r8.y & 0xffff0000u);
            hi[0] += __uint_as_float(r8.z << 16); hi[1] += __uint_as_float(r8.z & 0xffff0000u);
            hi[2] += __uint_as_float(r8.w << 16); hi[3] += __uint_as_float(r8.w & 0xffff0000u);
          }
          u32x4_t pk;
          pk.x = pack_bf16x2(lo[0], lo[1]); pk.y = pack_bf16x2(lo[2], lo[3]);
          pk.z = pack_bf16x2(hi[0], hi[1]); pk.w = pack_bf16x2(hi[2], hi[3]);
          *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(p.dst) + o) = pk;
        }
      }
      if (p.stats) conv_epilogue_stats<BP, BC, WP, WC, NORM>(p, acc, m0, n0, wp, wc, lane, smem_f32);
      return;
    }
  }
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
        if (HOIST) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * hsc[c][r] + hsh[c][r];
        } else {
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
        if (p.stats) acc[c][q] = f32x4_t{v[0], v[1], v[2], v[3]};
        if (NORM && !p.dst) {
          // statistics only: the caller consumes the values from the accumulators (conv_epilogue_norm)
        } else if (p.out_f32) {
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
  if (p.stats) conv_epilogue_stats<BP, BC, WP, WC, NORM>(p, acc, m0, n0, wp, wc, lane, smem_f32);
}

// ---------------------------------------------------------------------------
// Normalisation + activation fused behind the convolution (replaces a separate bn_apply / gn_relu_fwd launch and the
// re-read of the fp32 pre-normalisation tensor; models/model.py:395-417 GroupNorm(32) + ReLU of the towers,
// backbone/common.py:316-324 train-mode BatchNorm + LeakyReLU).  Phase 1 is conv_epilogue: the pre-normalisation
// values (optionally stored as fp32 for the backward pass) and their statistics, added across workgroups with
// returning device-scope atomics.  Then the workgroup waits until the statistics it needs are complete --
//   BatchNorm: every workgroup of the launch (grid barrier);
//   GroupNorm: the pixel tiles that share one of this tile's (level, image) keys, per channel tile: one counter per
//              (key, channel tile), every tile arrives at each key it touches (kd6d_barrier.h: window barrier);
// -- reads the totals back with device-scope loads and finishes act(norm(v)) from the accumulators it still holds.
// ---------------------------------------------------------------------------
template <typename T, int BP, int BC, int WP, int WC>
__device__ __forceinline__ void conv_epilogue_norm(const ConvParams& p, f32x4_t (&acc)[BC / WC / 16][BP / WP / 16],
                                                   int m0, int n0, int wp, int wc, int lane, float* red, int bid,
                                                   int nwg, int tile_c) {
  constexpr int PI = BP / WP / 16;
  constexpr int CI = BC / WC / 16;
  const int fr = lane & 15;
  const int fq = lane >> 4;
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (p.stats_groups == 0) {
    // ---- BatchNorm (train): statistics over all rows of the launch ----
    grid_barrier(p.norm_ctr, (unsigned)bid, (unsigned)nwg, p.norm_timeouts);
    const int R = p.stats_replicas > 1 ? p.stats_replicas : 1;
    for (int i = tid; i < 2 * BC; i += nthr) {
      const int which = i / BC, nl = i - which * BC;
      float t = 0.f;
      if (n0 + nl < p.N) {
        long long lo = 0, hi = 0;           // the replica rows' words as integers (exact), converted once
        for (int r = 0; r < R; ++r) {
          const det_words w = det_load_device_scope(p.stats + ((size_t)(r * 2 + which) * p.N + n0 + nl) * 2);
          lo += w.lo; hi += w.hi;
        }
        t = det_value<KD6D_DET_ACT>(lo, hi);
      }
      red[i] = t;
    }
    __syncthreads();
    if (m0 == 0) {           // one workgroup per channel tile publishes what the backward pass and eval mode need
      for (int nl = tid; nl < BC; nl += nthr) {
        const int n = n0 + nl;
        if (n >= p.N) continue;
        const float mean = red[nl] * p.bn_inv_rows;
        const float var = fmaxf(red[BC + nl] * p.bn_inv_rows - mean * mean, 0.f);
        if (p.bn_save_mean) p.bn_save_mean[n] = mean;
        if (p.bn_save_invstd) p.bn_save_invstd[n] = rsqrtf(var + p.norm_eps);
        if (p.bn_running_mean) p.bn_running_mean[n] = (1.f - p.bn_momentum) * p.bn_running_mean[n] + p.bn_momentum * mean;
        if (p.bn_running_var) p.bn_running_var[n] = (1.f - p.bn_momentum) * p.bn_running_var[n] + p.bn_momentum * var * p.bn_unbias;
      }
    }
#pragma unroll
    for (int c = 0; c < CI; ++c) {
      const int nl = wc * (BC / WC) + c * 16 + fq * 4;
      const int n = n0 + nl;
      if (n >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float mean = red[nl + r] * p.bn_inv_rows;
        const float var = fmaxf(red[BC + nl + r] * p.bn_inv_rows - mean * mean, 0.f);
        const float sc = p.norm_gamma[n + r] * rsqrtf(var + p.norm_eps);
        const float sh = __builtin_fmaf(-mean, sc, p.norm_beta[n + r]);
#pragma unroll
        for (int q = 0; q < PI; ++q) {
          float t = __builtin_fmaf(acc[c][q][r], sc, sh);
          if (p.norm_act == KD6D_ACT_LEAKY) t = t > 0.f ? t : 0.1f * t;
          else if (p.norm_act == KD6D_ACT_RELU) t = fmaxf(t, 0.f);
          acc[c][q][r] = t;
        }
      }
    }
  } else {
    // ---- GroupNorm: statistics per (level, image, group) ----
    const int G = p.stats_groups, cs = p.stats_cpg_shift;
    const int GT = BC >> cs;
    const int gts = 31 - __clz(GT);
    const int m_last = (m0 + BP < p.M ? m0 + BP : p.M) - 1;
    const int key_lo = stats_key(p, m0);
    const int nkeys = stats_key(p, m_last) - key_lo + 1;
    // rows [ks, ke] of key k in GEMM-row space and its pixel count
    auto key_rows = [&](int key, int& ks, int& hw) {
      int sg = 0;
#pragma unroll
      for (int s = 1; s < kMaxSeg; ++s)
        if (s < p.nseg && key >= s * p.batch) sg = s;
      int mb = 0;
      hw = 1;
#pragma unroll
      for (int s = 0; s < kMaxSeg; ++s)
        if (s == sg) { mb = p.seg[s].m_begin; hw = p.seg[s].dst_hw; }
      ks = mb + (key - sg * p.batch) * hw;
    };
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();            // every thread's statistics are performed (conv_epilogue_stats) before the arrivals
    for (int k = tid; k < nkeys; k += nthr) {
      int ks, hw;
      key_rows(key_lo + k, ks, hw);
      const unsigned need = (unsigned)((ks + hw - 1) / BP - ks / BP + 1);
      arrive_and_wait(p.norm_ctr + (size_t)(key_lo + k) * p.n_ctiles + tile_c, need, p.norm_timeouts);
    }
    __syncthreads();
    const int tab = nkeys << gts;
    const int g_first = n0 >> cs;
    for (int i = tid; i < tab; i += nthr) {
      const int k = i >> gts, gl = i & (GT - 1);
      float mu = 0.f, rs = 0.f;
      if (g_first + gl < G) {
        int ks, hw;
        key_rows(key_lo + k, ks, hw);
        const long long* st = p.stats + ((size_t)(key_lo + k) * G + g_first + gl) * 4;
        const det_words w1 = det_load_device_scope(st), w2 = det_load_device_scope(st + 2);
        const float s1 = det_value<KD6D_DET_ACT>(w1.lo, w1.hi), s2 = det_value<KD6D_DET_ACT>(w2.lo, w2.hi);
        const float inv_n = 1.f / ((float)hw * (float)(1 << cs));
        mu = s1 * inv_n;
        rs = rsqrtf(fmaxf(s2 * inv_n - mu * mu, 0.f) + p.norm_eps);
      }
      red[2 * i] = mu;
      red[2 * i + 1] = rs;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PI; ++q) {
      const int m = m0 + wp * (BP / WP) + q * 16 + fr;
      const int key = m < p.M ? stats_key(p, m) - key_lo : 0;
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        const int nl = wc * (BC / WC) + c * 16 + fq * 4;
        const int n = n0 + nl;
        if (n >= p.N) continue;
        const float* mr = red + (((key << gts) + (nl >> cs)) << 1);
        const float mu = mr[0], rs = mr[1];
        const f32x4_t ga = *reinterpret_cast<const f32x4_t*>(p.norm_gamma + n);
        const f32x4_t be = *reinterpret_cast<const f32x4_t*>(p.norm_beta + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = (acc[c][q][r] - mu) * rs * ga[r] + be[r];
          if (p.norm_act == KD6D_ACT_RELU) t = fmaxf(t, 0.f);
          else if (p.norm_act == KD6D_ACT_LEAKY) t = t > 0.f ? t : 0.1f * t;
          acc[c][q][r] = t;
        }
      }
    }
  }
  // ---- store the activation: 16 B per lane (fp32: 4 channels; bf16: 8 channels through the lane-pair swap of
  // conv_epilogue, 4 channels = 8 B where a wave holds an odd number of channel tiles) ----
  const bool odd = fq & 1;
  const int partner = (lane ^ 16) << 2;
#pragma unroll
  for (int q = 0; q < PI; ++q) {
    const int m = m0 + wp * (BP / WP) + q * 16 + fr;
    const bool row_ok = m < p.M;
    int drow = m;
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s)
      if (s < p.nseg && m >= p.seg[s].m_begin) drow = p.seg[s].dst_row0 + (m - p.seg[s].m_begin);
    if constexpr (sizeof(T) == 2 && (CI % 2 == 0)) {
      if ((p.N & 7) == 0) {
#pragma unroll
        for (int cp = 0; cp < CI / 2; ++cp) {
          const f32x4_t va0 = acc[2 * cp][q], va1 = acc[2 * cp + 1][q];
          const f32x4_t send = odd ? va0 : va1;
          f32x4_t got;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            got[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(send[r])));
          const f32x4_t lo = odd ? got : va0, hi = odd ? va1 : got;
          const int n = n0 + wc * (BC / WC) + (2 * cp + (odd ? 1 : 0)) * 16 + (fq & ~1) * 4;
          if (!row_ok || n >= p.N) continue;
          u32x4_t pk;
          pk.x = pack_bf16x2(lo[0], lo[1]); pk.y = pack_bf16x2(lo[2], lo[3]);
          pk.z = pack_bf16x2(hi[0], hi[1]); pk.w = pack_bf16x2(hi[2], hi[3]);
          *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(p.norm_dst) + (size_t)drow * (size_t)p.N + (size_t)n) = pk;
        }
        continue;
      }
    }
    if (!row_ok) continue;
#pragma unroll
    for (int c = 0; c < CI; ++c) {
      const int n = n0 + wc * (BC / WC) + c * 16 + fq * 4;
      if (n >= p.N) continue;
      const size_t o = (size_t)drow * (size_t)p.N + (size_t)n;
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.norm_dst) + o) = acc[c][q];
      } else {
        u32x2_t pk;
        pk.x = pack_bf16x2(acc[c][q][0], acc[c][q][1]);
        pk.y = pack_bf16x2(acc[c][q][2], acc[c][q][3]);
        *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(p.norm_dst) + o) = pk;
      }
    }
  }
}

// What the kernels call: conv_epilogue, then -- in the kernel variants compiled with NORM -- the fused normalisation when
// the launch carries one.  NORM is a template parameter, not a run-time branch: the first version compiled the second
// phase into every convolution kernel and its registers cost the whole step 5 % (5400-5508 -> 5141-5156 images/s on one
// box, full frames 1530 -> 1065) although no launch used it.
template <typename T, int BP, int BC, int WP, int WC, bool WIDE = true, bool NORM = false>
__device__ __forceinline__ void conv_epilogue_full(const ConvParams& p, f32x4_t (&acc)[BC / WC / 16][BP / WP / 16],
                                                   int m0, int n0, int wp, int wc, int lane, float* smem_f32, int bid,
                                                   int nwg, int tile_c) {
  conv_epilogue<T, BP, BC, WP, WC, WIDE, NORM>(p, acc, m0, n0, wp, wc, lane, smem_f32);
  if constexpr (NORM) {
    if (p.norm_dst) conv_epilogue_norm<T, BP, BC, WP, WC>(p, acc, m0, n0, wp, wc, lane, smem_f32, bid, nwg, tile_c);
  }
}

static __device__ const uint4 kd6d_zero_page[4] = {};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline bool fill_segs(const kd6d_conv_geom* g, bool dgrad, SegDev* seg, int* M_out) {
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
    d.inv_hw = 1.0f / (float)d.dst_hw;
    d.inv_w = 1.0f / (float)d.dst_w;
    d.m_begin = m;
    if (d.src_h > 32767 || d.src_w > 32767 || d.src_h < 0 || d.src_w < 0) return false;
    if ((long long)m + (long long)g->batch * d.dst_hw >= (1ll << 24)) return false;   // fast_div range
    m += g->batch * d.dst_hw;
  }
  *M_out = m;
  return true;
}

static inline int check_geom(const kd6d_conv_geom* g, int dtype, const char* who) {
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

// Workgroup order.  After the XCD remap an XCD runs a CONTIGUOUS range of ~1/8 of the tile ids, so the
// fastest-varying tile index decides which operand that XCD's 4 MiB L2 can keep: channel tiles fastest
// -> the XCD touches few pixel tiles but ALL weights; pixel tiles fastest -> few weight tiles but many
// pixels.  Pick the order with the smaller per-XCD footprint (small-M / wide-N layers: weights).
static inline void set_tile_order(ConvParams& q, int ptiles, int BP, int BC) {
  q.n_ptiles = ptiles;
  const double tiles = (double)ptiles * q.n_ctiles;
  const double per_xcd = tiles / 8.0;
  const double w_tile = (double)BC * q.K * 2.0, x_tile = (double)BP * q.C * 2.0 * (q.ks > 1 ? 1.5 : 1.0);
  // channel tiles fastest: an XCD spans per_xcd / n_ctiles pixel tiles (>= 1) and min(per_xcd, n_ctiles) weight tiles
  auto foot = [&](double n_fast, double t_fast, double t_slow) {
    const double fast = per_xcd < n_fast ? per_xcd : n_fast;
    const double slow = per_xcd / n_fast < 1.0 ? 1.0 : per_xcd / n_fast;
    return fast * t_fast + slow * t_slow;
  };
  const double c_fast = foot(q.n_ctiles, w_tile, x_tile);
  const double p_fast = foot(ptiles, x_tile, w_tile);
  q.p_fastest = p_fast < c_fast ? 1 : 0;
}

static inline int cached_cu_count() {       // one device per process
  static const int n = []() { const int c = kd6d_device_cu_count(); return c > 0 ? c : 256; }();
  return n;
}

// Dry run of the forward dispatch (kd6d_conv2d_fwd_norm_fusable): while g_launch_plan is set, the launch functions
// record what they would launch instead of launching it.
struct LaunchPlan {
  int grid = 0, threads = 0;
  size_t lds = 0;
  bool fused_epilogue = false;     // the kernel ends in conv_epilogue_full
};
extern thread_local LaunchPlan* g_launch_plan;
static inline bool plan_only(int grid, int threads, size_t lds, bool fused_epilogue) {
  if (!g_launch_plan) return false;
  g_launch_plan->grid = grid; g_launch_plan->threads = threads; g_launch_plan->lds = lds;
  g_launch_plan->fused_epilogue = fused_epilogue;
  return true;
}

// conv_igemm.hip: group statistics of the levels the epilogue skips (ConvParams::stats_skip), from the stored tensor
int stats_followup(const ConvParams& p, bool dst_f32, hipStream_t st);
// norm_ops.hip: {sum, sum of squares} per (level, image, group) of the levels in `mask` of a packed NHWC tensor
int gn_stats_levels(int src_f32, const void* y, const int* row0, const int* hw, int nseg, int batch, int C, int G,
                    unsigned mask, long long* stats, hipStream_t st);
// conv_halo.hip: the 3x3 / stride 1 halo-patch kernel takes the layer (returns false: not its shape / too few tiles)
bool dispatch_halo_fwd(const ConvParams& p, const kd6d_conv_geom* g, hipStream_t st);
bool dispatch_halo_dgrad(const ConvParams& p, const kd6d_conv_geom* g, hipStream_t st);

}  // namespace kd6d_detail
