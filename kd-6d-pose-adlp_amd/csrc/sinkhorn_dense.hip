// Dense Sinkhorn divergence: thousands of points per set, D-dimensional local predictions
// (BASELINE config 5: ZebraPose-style 16-D code predictions over a 128x128 cell grid,
// N = M = 16384).  Same algorithm as sinkhorn.hip / SURVEY.md App. B -- geomloss 0.2.4
// SamplesLoss("sinkhorn", p=2, debias=True, reach) with epsilon scaling, symmetrised updates and a
// detached last extrapolation that carries the gradient -- but the N x M cost matrices (4 x 1 GiB at
// this size) are NEVER materialised: every softmin is an online logsumexp over column tiles staged
// in LDS ("flash" form), so a pass streams (N+M)(D+2) floats from HBM and is bound by the fp32 VALU
// + v_exp_f32 rate (~40 lane-ops per pair), not by memory.  The reference itself cannot run this
// size: geomloss switches to the KeOps backend above 5000^2 pairs and pykeops is not among its
// requirements.
//
// Cost is evaluated from coordinate DIFFERENCES, 0.5*|r - c|^2, not from the |r|^2 - 2 r.c + |c|^2
// expansion: at blur = 1e-3 (eps = 1e-6) the expansion cancels catastrophically in fp32 for exactly
// the near pairs that dominate the softmin, so a matrix-core dot product would be fast and wrong.
//
// One launch = the four softmins of one epsilon step (blockIdx.y selects x<-x, y<-y, y<-x, x<-y);
// thread = one row with its D coordinates in registers; 4 columns per inner step; exponentials in the
// exp2 domain (v_exp_f32).  Potentials are double-buffered so all four updates read the old values.
#include <math.h>

#include "kd6d_common.h"

namespace {

constexpr int kThreadsD = 512;    // 2 waves per SIMD: a lone wave issues one VALU op per 4+ clk
constexpr int kRows = 256;        // rows per workgroup of the default form: kSplit = 2 threads per row, each takes its share
constexpr int kSplit = 2;         // of every column tile, merged at the end; the gradient form of the last extrapolation
                                  // runs on two of the four softmins only and takes SPLIT = 4 (128 rows per workgroup) so
                                  // that it still yields one workgroup per CU at N = 16384
constexpr int kTile = 128;        // columns staged per LDS tile (kTile / kSplit per thread)
constexpr float kNegLogD = -100000.f;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr double kMfmaEpsRel = 1.5e-4;   // eps / diameter^2 from which the matrix-pipe softmin is taken (measured: tests)

struct DenseArgs {
  const float* x; const float* y;          // (N,D), (M,D)
  const float* la; const float* lb;        // log weights (N), (M)
  const float* pot_old;                    // [a_x (N) | b_x (N) | b_y (M) | a_y (M)]
  float* pot_new;
  int N, M;
  int mode;                                // 0: init (h = log w), 1: symmetrised update, 2: last extrapolation
  float eps, lam;                          // epsilon, 1/(1+eps/rho) (1 if balanced)
  float* grad_xx; float* grad_xy;          // mode 2: softmax-weighted sums of (x_i - c_j), (N,D) each
  float screen_th;                         // dense_softmin_screen_kernel: exponent distance from the running maximum beyond which a pair is dropped
  int which[4];                            // the softmins of this launch, one per blockIdx.y: 0 a_x (x<-x), 1 b_y (y<-y),
                                           // 2 a_y (y<-x), 3 b_x (x<-y)
};

// potentials layout helpers
__device__ __forceinline__ int off_ax(const DenseArgs&) { return 0; }
__device__ __forceinline__ int off_bx(const DenseArgs& a) { return a.N; }
__device__ __forceinline__ int off_by(const DenseArgs& a) { return 2 * a.N; }
__device__ __forceinline__ int off_ay(const DenseArgs& a) { return 2 * a.N + a.M; }

template <int D, bool GRAD, int SPLIT = kSplit>
__global__ __launch_bounds__(kThreadsD) void dense_softmin_kernel(const DenseArgs a) {
  constexpr int P = (D + 4) & ~3;          // LDS pitch: D coords + h (+ pad), a multiple of 16 bytes
  constexpr int kRows = kThreadsD / SPLIT;
  constexpr int kSplit = SPLIT;
  constexpr int MW = GRAD ? D + 2 : 2;     // floats of a partial (m, s, g)
  __shared__ __attribute__((aligned(16))) float tile[2][kTile * P];
  __shared__ float mrg[(SPLIT - 1) * kRows * MW];       // partials of the other column shares

  // which of the four softmins: rows / columns / column potential / output slot
  const int which = a.which[blockIdx.y];   // 0: a_x (x<-x)  1: b_y (y<-y)  2: a_y (y<-x)  3: b_x (x<-y)
  const bool rows_x = (which == 0 || which == 3);
  const bool cols_x = (which == 0 || which == 2);
  const float* R = rows_x ? a.x : a.y;
  const float* Cc = cols_x ? a.x : a.y;
  const int nr = rows_x ? a.N : a.M;
  const int nc = cols_x ? a.N : a.M;
  const float* lw = cols_x ? a.la : a.lb;
  // column potential of the update (App. B): a_x uses a_x, b_y uses b_y, a_y uses b_x, b_x uses a_y
  const int cpot = which == 0 ? off_ax(a) : which == 1 ? off_by(a) : which == 2 ? off_bx(a) : off_ay(a);
  const int opot = which == 0 ? off_ax(a) : which == 1 ? off_by(a) : which == 2 ? off_ay(a) : off_bx(a);

  const int lrow = threadIdx.x % kRows;
  const int part = threadIdx.x / kRows;    // which share of every column tile this thread reduces
  const int row = blockIdx.x * kRows + lrow;
  if (blockIdx.x * kRows >= nr) return;    // uniform per workgroup
  const bool rok = row < nr;
  float r[D];
#pragma unroll
  for (int d = 0; d < D; ++d) r[d] = rok ? R[(size_t)row * D + d] : 0.f;

  const float inv_eps = 1.f / a.eps;
  const float k2 = 0.5f * inv_eps * kLog2e;      // exp2-domain scale of the squared distance
  float m = -INFINITY, s = 0.f;
  float g[GRAD ? D : 1];
#pragma unroll
  for (int d = 0; d < (GRAD ? D : 1); ++d) g[d] = 0.f;

  auto stage = [&](int buf, int c0) {
    // kTile columns x (D coords + h): one float per thread-iteration, coalesced on the coordinate array
    for (int i = threadIdx.x; i < kTile * D; i += kThreadsD) {
      const int c = i / D, d = i - c * D;
      tile[buf][c * P + d] = (c0 + c < nc) ? Cc[(size_t)(c0 + c) * D + d] : 0.f;
    }
    for (int c = threadIdx.x; c < kTile; c += kThreadsD) {
      float h = -INFINITY;                 // padding columns never contribute
      if (c0 + c < nc) {
        h = lw[c0 + c];
        if (a.mode != 0) h += a.pot_old[cpot + c0 + c] * inv_eps;
        h *= kLog2e;
      }
      tile[buf][c * P + D] = h;
    }
  };

  const int ntiles = (nc + kTile - 1) / kTile;
  stage(0, 0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) stage(buf ^ 1, (t + 1) * kTile);
    const float* T = tile[buf] + part * (kTile / kSplit) * P;
#pragma unroll 1
    for (int c = 0; c < kTile / kSplit; c += 4) {
      float v[4], dd[4][GRAD ? D : 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* col = T + (c + u) * P;
        float d2 = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < D; d4 += 4) {
          const f32x4_t cv = *reinterpret_cast<const f32x4_t*>(col + d4);     // LDS broadcast
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (d4 + e < D) {
              const float df = r[d4 + e] - cv[e];
              d2 += df * df;
              if (GRAD) dd[u][d4 + e] = df;
            }
          }
        }
        v[u] = col[D] - d2 * k2;
      }
      const float mn = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
      if (mn > -INFINITY) {
        const float sc = __builtin_amdgcn_exp2f(m - mn);
        s *= sc;
        if (GRAD) {
#pragma unroll
          for (int d = 0; d < D; ++d) g[d] *= sc;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float e = __builtin_amdgcn_exp2f(v[u] - mn);
          s += e;
          if (GRAD) {
#pragma unroll
            for (int d = 0; d < D; ++d) g[d] += e * dd[u][d];
          }
        }
        m = mn;
      }
    }
    __syncthreads();
  }
  // merge the column shares of a row: logsumexp of the partial (max, sum) pairs
  if (part != 0) {
    float* o = mrg + ((part - 1) * kRows + lrow) * MW;
    o[0] = m;
    o[1] = s;
    if (GRAD) {
#pragma unroll
      for (int d = 0; d < D; ++d) o[2 + d] = g[d];
    }
  }
  __syncthreads();
  if (part != 0 || !rok) return;
#pragma unroll
  for (int q = 1; q < SPLIT; ++q) {
    const float* o = mrg + ((q - 1) * kRows + lrow) * MW;
    const float m2 = o[0], s2 = o[1];
    const float mn = fmaxf(m, m2);
    const float c1 = m > -INFINITY ? __builtin_amdgcn_exp2f(m - mn) : 0.f;
    const float c2 = m2 > -INFINITY ? __builtin_amdgcn_exp2f(m2 - mn) : 0.f;
    s = s * c1 + s2 * c2;
    if (GRAD) {
#pragma unroll
      for (int d = 0; d < D; ++d) g[d] = g[d] * c1 + o[2 + d] * c2;
    }
    m = mn;
  }
  // softmin = -eps * logsumexp;  back from the exp2 domain
  const float lse = (m + __builtin_amdgcn_logf(s)) * kLn2;     // v_log_f32 = log2
  const float val = -a.lam * a.eps * lse;
  if (a.mode == 1) a.pot_new[opot + row] = 0.5f * (a.pot_old[opot + row] + val);
  else a.pot_new[opot + row] = val;
  if (GRAD && (which == 0 || which == 3)) {
    float* go = (which == 0 ? a.grad_xx : a.grad_xy) + (size_t)row * D;
    const float inv_s = 1.f / s;
#pragma unroll
    for (int d = 0; d < D; ++d) go[d] = g[d] * inv_s;
  }
}

// ---------------------------------------------------------------------------
// The same softmin on the matrix pipe (D = 16, no gradient): C_ij / eps = k (|r_i|^2 + |c_j|^2 - 2 r_i.c_j), so in the
// exp2 domain  v_ij = H_j + (2 k2 r_i).c_j - k2 |r_i|^2  with  H_j = h_j log2e - k2 |c_j|^2.  The row term is constant
// along j and is added after the logsumexp; the VALU is left with the online logsumexp, ~4 lane-ops per pair instead
// of ~40.  The inner products run on the bf16 matrix pipe at fp32 accuracy: every fp32 operand is split into three
// bf16 pieces (hi + mid + lo = the 24-bit value exactly) and the six products of order <= 2^-16 -- hh, hm, mh, hl, lh,
// mm -- are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (K = 16 = the code dimension: one instruction per
// product), 6 x 32 cycles per 32 x 32 block.  (v_mfma_f32_32x32x2_f32 gives the same numbers but runs at the fp32
// VECTOR rate and, measured, does not overlap the logsumexp: 8 x 64 cycles + the VALU work, 0.45 ms per launch; the
// first version of this kernel.)  The columns' pieces do not depend on epsilon and are prepared once per call
// (dense_split_kernel); the rows' operand (2 k2)(r - centre) is split by the lane that owns it at the start of a launch.
// A = 32 columns (from LDS), B = 32 rows (registers), so a lane ends with 16 columns of ONE row: the running
// (max, sum) of a row lives in the two lanes l, l + 32 and in the two waves that share the row group.
// The expansion cancels: its absolute error is ~2^-22 (|r|^2 + |c|^2) whatever the distance, i.e. ~k2 2^-22 S in the
// exponent.  Points are centred (S = spread^2 instead of |p|^2) and the host takes this kernel only while that stays
// below the fp32 noise the difference form has anyway (eps >= kMfmaEpsRel * diameter^2; the last, gradient-carrying
// extrapolation and the small-eps steps keep the difference form above).
// ---------------------------------------------------------------------------
constexpr int kMRows = 64;         // rows per workgroup: 2 row groups of 32; the other two waves take the other column half
constexpr int kMTile = 128;        // columns per LDS tile: 4 blocks of 32, two per wave
constexpr int kSplitBytes = 96;    // prepared point: 3 pieces x 2 k-halves x 8 bf16
constexpr int kMPitchB = 112;      // LDS bytes per column: 96 used; 28 dwords -> conflict-free ds_read_b128 over 16 columns

typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split3(float v, bf16_t& h, bf16_t& m, bf16_t& l) {
  h = (bf16_t)v;
  const float r1 = v - (float)h;
  m = (bf16_t)r1;
  l = (bf16_t)(r1 - (float)m);
}

typedef _Float16 f16_t;
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
constexpr int kYtBytes = 64;            // per point: 16 dimensions x (hi, lo) fp16
constexpr float kLoScale = 2048.f;      // the lo pieces travel scaled by 2^11 (kept out of fp16's subnormal range)

// centred points -> MFMA-ready pieces [point][piece h,m,l][k half][8 bf16] + |p - centre|^2, once per call.
// `yt`: the same centred point as two fp16 pieces (hi, lo * 2^11) in the layout the gradient kernel's SECOND product
// reads as its A operand -- per block of 32 points [k-step s][lane half h][row m = piece * 16 + dimension][8 points]:
// the 8 points of (s, h) are the ones whose weights a lane of half h holds in accumulators 8 s .. 8 s + 7 of the first
// product (point 16 s + 8 (e / 4) + 4 h + e % 4 at position e).  Points n .. the next multiple of 128 are written as zeros.
__global__ __launch_bounds__(256) void dense_split_kernel(const float* __restrict__ p, int n, const float* __restrict__ sums,
                                                          float inv_count, char* __restrict__ out, float* __restrict__ n2,
                                                          char* __restrict__ yt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (yt != nullptr && i < ((n + 127) & ~127)) {
    const int jj = i & 31, st = jj >> 4, rem = jj & 15, e = 4 * (rem >> 3) + (rem & 3), hf = (rem >> 2) & 1;
    f16_t* base = reinterpret_cast<f16_t*>(yt + (size_t)(i >> 5) * (32 * kYtBytes)) + ((st * 2 + hf) * 32) * 8 + e;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      f16_t hi = (f16_t)0.f, lo = (f16_t)0.f;
      if (i < n) {
        const float v = p[(size_t)i * 16 + k] - sums[k] * inv_count;
        hi = (f16_t)v;
        lo = (f16_t)((v - (float)hi) * kLoScale);
      }
      base[k * 8] = hi;
      base[(16 + k) * 8] = lo;
    }
  }
  if (i >= n) return;
  float center[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) center[k] = sums[k] * inv_count;
  bf16_t h[16], m[16], l[16];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float v = p[(size_t)i * 16 + k] - center[k];
    s += v * v;
    split3(v, h[k], m[k], l[k]);
  }
  n2[i] = s;
  bf16x8_t* o = reinterpret_cast<bf16x8_t*>(out + (size_t)i * kSplitBytes);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    bf16x8_t vh, vm, vl;
#pragma unroll
    for (int e = 0; e < 8; ++e) { vh[e] = h[8 * half + e]; vm[e] = m[8 * half + e]; vl[e] = l[8 * half + e]; }
    o[0 * 2 + half] = vh; o[1 * 2 + half] = vm; o[2 * 2 + half] = vl;
  }
}

struct DenseSplit {
  const char* xs; const char* ys;      // prepared points (kSplitBytes each)
  const float* xn2; const float* yn2;  // |p - centre|^2
  const char* xt; const char* yt;      // the same points as the A operand of the gradient's second product (kYtBytes each)
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// online logsumexp over a lane's 2 x 16 columns of its row (padding columns are -inf).  Written for the vector pipe,
// which bounds this kernel: the maximum as ONE chain (v_max3_f32 takes two new values per instruction; a tree of
// fmaxf pairs made the compiler quiet every leaf with a v_max_f32 x, x first: 26 instructions per 16 values instead
// of 8), differences and sums two columns per instruction (v_pk_add_f32), one rescale of the running sum per tile.
__device__ __forceinline__ void dense_lse_update(const f32x16_t& a0, const f32x16_t& a1, float& m, float& s) {
  float mn = m;
#pragma unroll
  for (int u = 0; u < 8; ++u) mn = fmaxf(fmaxf(mn, a0[2 * u]), a0[2 * u + 1]);
#pragma unroll
  for (int u = 0; u < 8; ++u) mn = fmaxf(fmaxf(mn, a1[2 * u]), a1[2 * u + 1]);
  const f32x2_t mn2 = {mn, mn};
  f32x2_t add = {0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const f32x2_t d0 = f32x2_t{a0[2 * u], a0[2 * u + 1]} - mn2;
    const f32x2_t d1 = f32x2_t{a1[2 * u], a1[2 * u + 1]} - mn2;
    add += f32x2_t{__builtin_amdgcn_exp2f(d0[0]), __builtin_amdgcn_exp2f(d0[1])};
    add += f32x2_t{__builtin_amdgcn_exp2f(d1[0]), __builtin_amdgcn_exp2f(d1[1])};
  }
  s = s * __builtin_amdgcn_exp2f(m - mn) + (add[0] + add[1]);
  m = mn;
}

template <int RG>
__global__ __launch_bounds__(RG * 128) void dense_softmin_mfma_kernel(const DenseArgs a, const DenseSplit sp) {
  __shared__ __attribute__((aligned(16))) char cs[2][kMTile * kMPitchB];
  __shared__ __attribute__((aligned(16))) float hs[2][kMTile];
  constexpr int T = RG * 128, ROWS = RG * 32;        // threads, rows per workgroup: RG row groups x 2 column halves
  __shared__ float mrg[ROWS * 2];
  const int which = a.which[blockIdx.y];
  const bool rows_x = (which == 0 || which == 3);
  const bool cols_x = (which == 0 || which == 2);
  const char* Rs = rows_x ? sp.xs : sp.ys;
  const float* Rn2 = rows_x ? sp.xn2 : sp.yn2;
  const char* Cs = cols_x ? sp.xs : sp.ys;
  const float* Cn2 = cols_x ? sp.xn2 : sp.yn2;
  const int nr = rows_x ? a.N : a.M;
  const int nc = cols_x ? a.N : a.M;
  const float* lw = cols_x ? a.la : a.lb;
  const int cpot = which == 0 ? off_ax(a) : which == 1 ? off_by(a) : which == 2 ? off_bx(a) : off_ay(a);
  const int opot = which == 0 ? off_ax(a) : which == 1 ? off_by(a) : which == 2 ? off_ay(a) : off_bx(a);
  if (blockIdx.x * ROWS >= nr) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rgrp = wave % RG;                     // which 32 rows of the workgroup
  const int chalf = wave / RG;                    // which two of the four 32-column blocks of every tile
  const int half = lane >> 5;                     // which 8 of the 16 dimensions this lane feeds
  const int lrow = rgrp * 32 + (lane & 31);
  const int row = blockIdx.x * ROWS + lrow;
  const bool rok = row < nr;
  const float inv_eps = 1.f / a.eps;
  const float k2 = 0.5f * inv_eps * kLog2e;
  // B operand: (2 k2)(r - centre) for k = 8 half .. 8 half + 7, rebuilt from the prepared pieces (their sum is the fp32
  // value exactly) and split again after the scaling
  bf16x8_t bh, bm_, bl;
  {
    const bf16x8_t* rp = reinterpret_cast<const bf16x8_t*>(Rs + (size_t)(rok ? row : 0) * kSplitBytes);
    const bf16x8_t ph = rp[0 * 2 + half], pm = rp[1 * 2 + half], pl = rp[2 * 2 + half];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = rok ? ((float)ph[e] + (float)pm[e]) + (float)pl[e] : 0.f;
      bf16_t h, m, l;
      split3(2.f * k2 * v, h, m, l);
      bh[e] = h; bm_[e] = m; bl[e] = l;
    }
  }
  const float rr = rok ? Rn2[row] : 0.f;

  // Column tiles travel global -> registers -> LDS: fetch() issues the loads of tile t + 1 before tile t is multiplied,
  // put() stores them behind it (the first version loaded, waited and stored chunk by chunk inside the tile loop: three
  // exposed L2 round trips per tile and workgroup).  A tile is 128 x 96 contiguous bytes = 768 chunks of 16 B, three per
  // thread; chunk i is piece q = i % 6 of column i / 6.  H_j comes from this launch's potentials (threads 0 .. 127).
  constexpr int NCH = (768 + T - 1) / T;            // 16-byte chunks of the pieces per thread (the last may be partial)
  int ldst[NCH], scol[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int i = tid + T * j;
    scol[j] = i < 768 ? i / 6 : (1 << 28);          // a chunk beyond the tile: never in range
    ldst[j] = i < 768 ? scol[j] * kMPitchB + (i - scol[j] * 6) * 16 : 0;
  }
  // TWO tiles of loads are in flight: a tile's compute (~0.2 us per wave) is far shorter than an L2 / HBM round trip, and
  // with one tile of look-ahead every iteration waited for its loads (285 us per launch against an 82-us matrix-pipe
  // floor); the register sets alternate, the tile loop is unrolled by two so that both stay in registers
  struct Stage { u32x4_t sv[NCH]; float lw, pot, n2; };    // lw, pot, n2: the three inputs of H_j as loaded; combined in put()
  Stage stA, stB;
  const bool with_pot = a.mode != 0;
  auto fetch = [&](int c0, Stage& S) {                    // c0 >= nc: nothing is read, the tile is all padding
    const char* g = Cs + (size_t)c0 * kSplitBytes + (size_t)tid * 16;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      S.sv[j] = u32x4_t{0u, 0u, 0u, 0u};
      if (c0 + scol[j] < nc) S.sv[j] = *reinterpret_cast<const u32x4_t*>(g + j * (T * 16));
    }
    S.lw = -INFINITY; S.pot = 0.f; S.n2 = 0.f;
    if (tid < kMTile) {
      const bool ok = c0 + tid < nc;
      const int c = ok ? c0 + tid : 0;
      S.lw = ok ? lw[c] : -INFINITY;
      S.pot = with_pot ? a.pot_old[cpot + c] : 0.f;
      S.n2 = Cn2[c];
    }
  };
  auto put = [&](int buf, const Stage& S) {
#pragma unroll
    for (int j = 0; j < NCH; ++j)
      if (tid + T * j < 768) *reinterpret_cast<u32x4_t*>(&cs[buf][ldst[j]]) = S.sv[j];
    if (tid < kMTile) hs[buf][tid] = (S.lw + S.pot * inv_eps) * kLog2e - k2 * S.n2;     // -inf for a padding column
  };

  auto block = [&](int buf, int blk) {
    f32x16_t acc;
    const float* hb = &hs[buf][blk * 32];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4_t h4 = *reinterpret_cast<const f32x4_t*>(hb + 4 * half + 8 * u);
      acc[4 * u + 0] = h4[0]; acc[4 * u + 1] = h4[1]; acc[4 * u + 2] = h4[2]; acc[4 * u + 3] = h4[3];
    }
    const char* col = &cs[buf][(blk * 32 + (lane & 31)) * kMPitchB + half * 16];
    const bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(col);
    const bf16x8_t am = *reinterpret_cast<const bf16x8_t*>(col + 32);
    const bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(col + 64);
    // smallest products first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm_, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm_, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    return acc;
  };

  float m = -1e30f, s = 0.f;
  const int ntiles = (nc + kMTile - 1) / kMTile;
  // one tile: this wave's two blocks -- both MFMA chains are issued before the logsumexp, so the matrix pipe works on the
  // second block while the VALU starts on the first; the stores of the tile after next (and their vmcnt wait) stay behind
  // the logsumexp: the empty asm gives the pure arithmetic a position in the instruction stream the barrier can hold
  auto tile = [&](int buf) {
    __builtin_amdgcn_sched_barrier(0);
    const f32x16_t acc0 = block(buf, 2 * chalf);
    const f32x16_t acc1 = block(buf, 2 * chalf + 1);
    dense_lse_update(acc0, acc1, m, s);
    __asm__ volatile("" : "+v"(m), "+v"(s));
    __builtin_amdgcn_sched_barrier(0);
  };
  fetch(0, stA);
  put(0, stA);
  fetch(kMTile, stB);
  __syncthreads();
  // (an odd tile count runs one all-padding tile: H = -inf everywhere, no effect on (m, s); a loop body without branches
  // is one scheduling region)
  for (int t = 0; t < ntiles; t += 2) {
    fetch((t + 2) * kMTile, stA);
    tile(0);
    put(1, stB);                 // tile t + 1; every wave finished reading buffer 1 before the previous barrier
    __syncthreads();
    fetch((t + 3) * kMTile, stB);
    tile(1);
    put(0, stA);                 // tile t + 2
    __syncthreads();
  }
  // the two lanes of a row, then the two column halves (waves w and w + 2)
  {
    const float m2 = __shfl_xor(m, 32, 64), s2 = __shfl_xor(s, 32, 64);
    const float mn = fmaxf(m, m2);
    s = s * __builtin_amdgcn_exp2f(m - mn) + s2 * __builtin_amdgcn_exp2f(m2 - mn);
    m = mn;
  }
  if (chalf == 1 && half == 0) { mrg[2 * lrow] = m; mrg[2 * lrow + 1] = s; }
  __syncthreads();
  if (chalf != 0 || half != 0 || !rok) return;
  {
    const float m2 = mrg[2 * lrow], s2 = mrg[2 * lrow + 1];
    const float mn = fmaxf(m, m2);
    s = s * __builtin_amdgcn_exp2f(m - mn) + s2 * __builtin_amdgcn_exp2f(m2 - mn);
    m = mn;
  }
  const float lse = (m + __builtin_amdgcn_logf(s) - k2 * rr) * kLn2;
  const float val = -a.lam * a.eps * lse;
  if (a.mode == 1) a.pot_new[opot + row] = 0.5f * (a.pot_old[opot + row] + val);
  else a.pot_new[opot + row] = val;
}

// ---------------------------------------------------------------------------
// Small epsilon (below the matrix-pipe rule above): SCREENING.  The expansion's value is only good to
// ~k2 2^-22 (|r|^2 + |c|^2) in the exponent -- +-1 at eps = 1e-6 -- so it cannot BE the softmin's term; but a pair whose
// approximate exponent lies more than TH = 40 + (that error bound) below the running maximum of its row contributes less
// than 2^-40 of the sum whatever its exact value, and at small epsilon that is almost every pair: exponents are spread
// over k2 |r - c|^2 ~ 10^3 ... 10^6 units.  So: the six-product block on the matrix pipe as above, ONE maximum chain and
// one compare per lane and tile on the vector pipe, and only lanes that hold a pair within TH of their running maximum
// evaluate those pairs EXACTLY -- the difference form of dense_softmin_kernel, from raw fp32 coordinates staged beside
// the pieces: the same arithmetic, so the result equals the difference form's up to the neglected < 2^-40 terms.  The
// (max, sum[, gradient sums]) of the exact values are what the kernel keeps; the approximate values never enter them.
// A lane's own maximum is always within TH of itself, so every lane with a finite column has a term.
// GRAD: the softmax-weighted difference sums of the last extrapolation (rows of x only), as dense_softmin_kernel<.., true>.
// ---------------------------------------------------------------------------
template <bool GRAD, int RG>
__global__ __launch_bounds__(RG * 128) void dense_softmin_screen_kernel(const DenseArgs a, const DenseSplit sp) {
  constexpr int D = 16;
  __shared__ __attribute__((aligned(16))) char cs[2][kMTile * kMPitchB];
  __shared__ __attribute__((aligned(16))) float raw[2][kMTile * D];
  __shared__ __attribute__((aligned(16))) float hs[2][kMTile];       // approximate: h log2e - k2 |c - centre|^2
  __shared__ float hx[2][kMTile];                                     // exact: h log2e
  constexpr int T = RG * 128, ROWS = RG * 32;        // threads, rows per workgroup: RG row groups x 2 column halves
  __shared__ float mrg[ROWS * (GRAD ? 2 + D : 2)];
  const int which = a.which[blockIdx.y];
  const bool rows_x = (which == 0 || which == 3);
  const bool cols_x = (which == 0 || which == 2);
  const char* Rs = rows_x ? sp.xs : sp.ys;
  const char* Cs = cols_x ? sp.xs : sp.ys;
  const float* Cn2 = cols_x ? sp.xn2 : sp.yn2;
  const float* Rraw = rows_x ? a.x : a.y;
  const float* Craw = cols_x ? a.x : a.y;
  const int nr = rows_x ? a.N : a.M;
  const int nc = cols_x ? a.N : a.M;
  const float* lw = cols_x ? a.la : a.lb;
  const int cpot = which == 0 ? off_ax(a) : which == 1 ? off_by(a) : which == 2 ? off_bx(a) : off_ay(a);
  const int opot = which == 0 ? off_ax(a) : which == 1 ? off_by(a) : which == 2 ? off_ay(a) : off_bx(a);
  if (blockIdx.x * ROWS >= nr) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rgrp = wave % RG, chalf = wave / RG, half = lane >> 5;
  const int lrow = rgrp * 32 + (lane & 31);
  const int row = blockIdx.x * ROWS + lrow;
  const bool rok = row < nr;
  const float inv_eps = 1.f / a.eps;
  const float k2 = 0.5f * inv_eps * kLog2e;
  bf16x8_t bh, bm_, bl;
  {
    const bf16x8_t* rp = reinterpret_cast<const bf16x8_t*>(Rs + (size_t)(rok ? row : 0) * kSplitBytes);
    const bf16x8_t ph = rp[0 * 2 + half], pm = rp[1 * 2 + half], pl = rp[2 * 2 + half];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = rok ? ((float)ph[e] + (float)pm[e]) + (float)pl[e] : 0.f;
      bf16_t h, m, l;
      split3(2.f * k2 * v, h, m, l);
      bh[e] = h; bm_[e] = m; bl[e] = l;
    }
  }
  float r[D];
#pragma unroll
  for (int d4 = 0; d4 < D; d4 += 4) {
    const f32x4_t rv = *reinterpret_cast<const f32x4_t*>(Rraw + (size_t)(rok ? row : 0) * D + d4);
    r[d4] = rv[0]; r[d4 + 1] = rv[1]; r[d4 + 2] = rv[2]; r[d4 + 3] = rv[3];
  }

  // staging as in dense_softmin_mfma_kernel, plus the raw coordinates: 128 x 64 bytes = 512 chunks of 16 B, two per thread
  constexpr int NCH = (768 + T - 1) / T;            // 16-byte chunks of the pieces per thread (the last may be partial)
  int ldst[NCH], scol[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int i = tid + T * j;
    scol[j] = i < 768 ? i / 6 : (1 << 28);          // a chunk beyond the tile: never in range
    ldst[j] = i < 768 ? scol[j] * kMPitchB + (i - scol[j] * 6) * 16 : 0;
  }
  constexpr int NRW = (512 + T - 1) / T;
  struct Stage { u32x4_t sv[NCH]; u32x4_t rw[NRW]; float lw, pot, n2; };     // two tiles of loads in flight, as above
  Stage stA, stB;
  const bool with_pot = a.mode != 0;
  auto fetch = [&](int c0, Stage& S) {                    // c0 >= nc: nothing is read, the tile is all padding
    const char* g = Cs + (size_t)c0 * kSplitBytes + (size_t)tid * 16;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      S.sv[j] = u32x4_t{0u, 0u, 0u, 0u};
      if (c0 + scol[j] < nc) S.sv[j] = *reinterpret_cast<const u32x4_t*>(g + j * (T * 16));
    }
    const char* gr = reinterpret_cast<const char*>(Craw) + (size_t)c0 * (D * 4) + (size_t)tid * 16;
#pragma unroll
    for (int j = 0; j < NRW; ++j) {
      S.rw[j] = u32x4_t{0u, 0u, 0u, 0u};
      if (tid + T * j < 512 && c0 + (tid + T * j) / 4 < nc) S.rw[j] = *reinterpret_cast<const u32x4_t*>(gr + j * (T * 16));
    }
    S.lw = -INFINITY; S.pot = 0.f; S.n2 = 0.f;
    if (tid < kMTile) {
      const bool ok = c0 + tid < nc;
      const int c = ok ? c0 + tid : 0;
      S.lw = ok ? lw[c] : -INFINITY;
      S.pot = with_pot ? a.pot_old[cpot + c] : 0.f;
      S.n2 = Cn2[c];
    }
  };
  auto put = [&](int buf, const Stage& S) {
#pragma unroll
    for (int j = 0; j < NCH; ++j)
      if (tid + T * j < 768) *reinterpret_cast<u32x4_t*>(&cs[buf][ldst[j]]) = S.sv[j];
#pragma unroll
    for (int j = 0; j < NRW; ++j)
      if (tid + T * j < 512) *reinterpret_cast<u32x4_t*>(&raw[buf][(tid + T * j) * 4]) = S.rw[j];
    if (tid < kMTile) {
      const float h = (S.lw + S.pot * inv_eps) * kLog2e;          // -inf for a padding column
      hx[buf][tid] = h;
      hs[buf][tid] = h - k2 * S.n2;
    }
  };
  auto block = [&](int buf, int blk) {
    f32x16_t acc;
    const float* hb = &hs[buf][blk * 32];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4_t h4 = *reinterpret_cast<const f32x4_t*>(hb + 4 * half + 8 * u);
      acc[4 * u + 0] = h4[0]; acc[4 * u + 1] = h4[1]; acc[4 * u + 2] = h4[2]; acc[4 * u + 3] = h4[3];
    }
    const char* col = &cs[buf][(blk * 32 + (lane & 31)) * kMPitchB + half * 16];
    const bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(col);
    const bf16x8_t am = *reinterpret_cast<const bf16x8_t*>(col + 32);
    const bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(col + 64);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm_, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm_, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    return acc;
  };

  float mt = -1e30f;                       // running maximum of the APPROXIMATE exponents of this lane's columns
  float m = -1e30f, s = 0.f;               // exact (max, sum) over the pairs that passed the screen
  float g[GRAD ? D : 1];
#pragma unroll
  for (int d = 0; d < (GRAD ? D : 1); ++d) g[d] = 0.f;
  const float th = a.screen_th;

  // one pair, exactly: column `c` of the staged tile (the arithmetic of dense_softmin_kernel)
  auto exact = [&](int buf, int c) {
    const float* cr = &raw[buf][c * D];
    float d2 = 0.f, dd[GRAD ? D : 1];
#pragma unroll
    for (int d4 = 0; d4 < D; d4 += 4) {
      const f32x4_t cv = *reinterpret_cast<const f32x4_t*>(cr + d4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float df = r[d4 + e] - cv[e];
        d2 += df * df;
        if (GRAD) dd[d4 + e] = df;
      }
    }
    const float v = hx[buf][c] - d2 * k2;
    const float mn = fmaxf(m, v);
    const float sc = __builtin_amdgcn_exp2f(m - mn), e = __builtin_amdgcn_exp2f(v - mn);
    s = s * sc + e;
    if (GRAD) {
#pragma unroll
      for (int d = 0; d < D; ++d) g[d] = g[d] * sc + e * dd[d];
    }
    m = mn;
  };

  const int ntiles = (nc + kMTile - 1) / kMTile;
  auto tile = [&](int buf) {
    const f32x16_t acc0 = block(buf, 2 * chalf);
    const f32x16_t acc1 = block(buf, 2 * chalf + 1);
    float tm = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) tm = fmaxf(fmaxf(tm, acc0[2 * u]), acc0[2 * u + 1]);
#pragma unroll
    for (int u = 0; u < 8; ++u) tm = fmaxf(fmaxf(tm, acc1[2 * u]), acc1[2 * u + 1]);
    mt = fmaxf(mt, tm);
    const float thr = mt - th;
    if (tm >= thr) {                       // rare at small epsilon; lanes diverge here
      unsigned mask = 0u;
#pragma unroll
      for (int u = 0; u < 16; ++u) mask |= (acc0[u] >= thr ? 1u : 0u) << u;
#pragma unroll
      for (int u = 0; u < 16; ++u) mask |= (acc1[u] >= thr ? 1u : 0u) << (16 + u);
      while (mask) {
        const int u = __builtin_ctz(mask);
        mask &= mask - 1u;
        const int uu = u & 15;
        // accumulator 4 q + r of a block <-> its column 8 q + 4 half + r
        exact(buf, (2 * chalf + (u >> 4)) * 32 + 8 * (uu >> 2) + 4 * half + (uu & 3));
      }
    }
  };
  fetch(0, stA);
  put(0, stA);
  fetch(kMTile, stB);
  __syncthreads();
  for (int t = 0; t < ntiles; t += 2) {        // (an odd tile count runs one all-padding tile: no column passes the screen)
    fetch((t + 2) * kMTile, stA);
    tile(0);
    put(1, stB);
    __syncthreads();
    fetch((t + 3) * kMTile, stB);
    tile(1);
    put(0, stA);
    __syncthreads();
  }
  // the two lanes of a row, then the two column halves (waves w and w + 2)
  {
    const float m2 = __shfl_xor(m, 32, 64), s2 = __shfl_xor(s, 32, 64);
    const float mn = fmaxf(m, m2);
    const float c1 = __builtin_amdgcn_exp2f(m - mn), c2 = __builtin_amdgcn_exp2f(m2 - mn);
    s = s * c1 + s2 * c2;
    if (GRAD) {
#pragma unroll
      for (int d = 0; d < D; ++d) g[d] = g[d] * c1 + __shfl_xor(g[d], 32, 64) * c2;
    }
    m = mn;
  }
  constexpr int MW = GRAD ? 2 + D : 2;
  if (chalf == 1 && half == 0) {
    mrg[MW * lrow] = m; mrg[MW * lrow + 1] = s;
    if (GRAD) {
#pragma unroll
      for (int d = 0; d < D; ++d) mrg[MW * lrow + 2 + d] = g[d];
    }
  }
  __syncthreads();
  if (chalf != 0 || half != 0 || !rok) return;
  {
    const float m2 = mrg[MW * lrow], s2 = mrg[MW * lrow + 1];
    const float mn = fmaxf(m, m2);
    const float c1 = __builtin_amdgcn_exp2f(m - mn), c2 = __builtin_amdgcn_exp2f(m2 - mn);
    s = s * c1 + s2 * c2;
    if (GRAD) {
#pragma unroll
      for (int d = 0; d < D; ++d) g[d] = g[d] * c1 + mrg[MW * lrow + 2 + d] * c2;
    }
    m = mn;
  }
  const float lse = (m + __builtin_amdgcn_logf(s)) * kLn2;
  const float val = -a.lam * a.eps * lse;
  if (a.mode == 1) a.pot_new[opot + row] = 0.5f * (a.pot_old[opot + row] + val);
  else a.pot_new[opot + row] = val;
  if (GRAD) {
    float* go = (which == 0 ? a.grad_xx : a.grad_xy) + (size_t)row * D;
    const float inv_s = 1.f / s;
#pragma unroll
    for (int d = 0; d < D; ++d) go[d] = g[d] * inv_s;
  }
}

// ---------------------------------------------------------------------------
// The gradient-carrying softmins of the last extrapolation (rows of x: a_x over x, b_x over y) on the matrix pipe as well:
// grad_i = sum_j w_ij (x_i - c_j) = x_i - (sum_j P_ij c_j) / (sum_j P_ij),  P_ij = exp2(v_ij - m_i)  (centred points: the
// centre cancels).  The first product is the gradient-free kernel's; the second, O^T[piece * 16 + d][i] = sum_j C^T P^T, takes
// the accumulators of the first AS its B operand: a lane holds, for row i = lane % 32, the 16 columns 8 q + 4 h + r of a
// block, and accumulators 8 s .. 8 s + 7 are exactly the 8 k-values of lane half h in k-step s once the A operand lists the
// columns in that order (dense_split_kernel's `yt`).  Both operands in two fp16 pieces, the lo ones scaled by 2^11: P = P_hi +
// P_lo / 2^11 into two accumulators, C = C_hi + C_lo / 2^11 as rows 0-15 / 16-31 of A -- four v_mfma_f32_32x32x16_f16 per block
// behind the six of the inner products, every term of order >= 2^-22 kept.  One pass: the running maximum is shared by the two
// lanes of a row every tile (the second product sums over both lanes' columns, so they must scale alike), and O is rescaled with
// the running sum when it moves.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float lane_pair_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }

__global__ __launch_bounds__(256) void dense_softmin_mfma_grad_kernel(const DenseArgs a, const DenseSplit sp) {
  __shared__ __attribute__((aligned(16))) char cs[2][kMTile * kMPitchB];
  __shared__ __attribute__((aligned(16))) float hs[2][kMTile];
  __shared__ __attribute__((aligned(16))) char yts[2][kMTile * kYtBytes];
  const int which = a.which[blockIdx.y];          // 0: a_x (x <- x) or 3: b_x (x <- y): rows are points of x
  const bool cols_x = which == 0;
  const char* Rs = sp.xs;
  const float* Rn2 = sp.xn2;
  const char* Cs = cols_x ? sp.xs : sp.ys;
  const char* Ct = cols_x ? sp.xt : sp.yt;
  const float* Cn2 = cols_x ? sp.xn2 : sp.yn2;
  const int nr = a.N;
  const int nc = cols_x ? a.N : a.M;
  const float* lw = cols_x ? a.la : a.lb;
  const int cpot = which == 0 ? off_ax(a) : off_ay(a);
  const int opot = which == 0 ? off_ax(a) : off_bx(a);
  if (blockIdx.x * kMRows >= nr) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rgrp = wave & 1, chalf = wave >> 1, half = lane >> 5;
  const int lrow = rgrp * 32 + (lane & 31);
  const int row = blockIdx.x * kMRows + lrow;
  const bool rok = row < nr;
  const float inv_eps = 1.f / a.eps;
  const float k2 = 0.5f * inv_eps * kLog2e;
  bf16x8_t bh, bm_, bl;
  {
    const bf16x8_t* rp = reinterpret_cast<const bf16x8_t*>(Rs + (size_t)(rok ? row : 0) * kSplitBytes);
    const bf16x8_t ph = rp[0 * 2 + half], pm = rp[1 * 2 + half], pl = rp[2 * 2 + half];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = rok ? ((float)ph[e] + (float)pm[e]) + (float)pl[e] : 0.f;
      bf16_t h, m, l;
      split3(2.f * k2 * v, h, m, l);
      bh[e] = h; bm_[e] = m; bl[e] = l;
    }
  }
  const float rr = rok ? Rn2[row] : 0.f;

  int ldst[3], scol[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = tid + 256 * j;
    scol[j] = i / 6;
    ldst[j] = scol[j] * kMPitchB + (i - scol[j] * 6) * 16;
  }
  u32x4_t sv[3], tv[2];
  float s_lw = 0.f, s_pot = 0.f, s_n2 = 0.f;
  const bool with_pot = a.mode != 0;
  auto fetch = [&](int c0) {
    const char* g = Cs + (size_t)c0 * kSplitBytes + (size_t)tid * 16;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      sv[j] = u32x4_t{0u, 0u, 0u, 0u};
      if (c0 + scol[j] < nc) sv[j] = *reinterpret_cast<const u32x4_t*>(g + j * 4096);
    }
    const char* gt = Ct + (size_t)c0 * kYtBytes + (size_t)tid * 16;      // padded to a multiple of 128 points, zeros
#pragma unroll
    for (int j = 0; j < 2; ++j) tv[j] = *reinterpret_cast<const u32x4_t*>(gt + j * 4096);
    if (tid < kMTile) {
      const bool ok = c0 + tid < nc;
      const int c = ok ? c0 + tid : 0;
      s_lw = ok ? lw[c] : -INFINITY;
      s_pot = with_pot ? a.pot_old[cpot + c] : 0.f;
      s_n2 = Cn2[c];
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4_t*>(&cs[buf][ldst[j]]) = sv[j];
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4_t*>(&yts[buf][tid * 16 + j * 4096]) = tv[j];
    if (tid < kMTile) hs[buf][tid] = (s_lw + s_pot * inv_eps) * kLog2e - k2 * s_n2;
  };
  auto block = [&](int buf, int blk) {
    f32x16_t acc;
    const float* hb = &hs[buf][blk * 32];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4_t h4 = *reinterpret_cast<const f32x4_t*>(hb + 4 * half + 8 * u);
      acc[4 * u + 0] = h4[0]; acc[4 * u + 1] = h4[1]; acc[4 * u + 2] = h4[2]; acc[4 * u + 3] = h4[3];
    }
    const char* col = &cs[buf][(blk * 32 + (lane & 31)) * kMPitchB + half * 16];
    const bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(col);
    const bf16x8_t am = *reinterpret_cast<const bf16x8_t*>(col + 32);
    const bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(col + 64);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm_, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm_, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    return acc;
  };

  f32x16_t o1, o2;                     // sum_j P_hi C^T  and  2^11 sum_j P_lo C^T  (rows: 8 q + 4 half + r of [C_hi; 2^11 C_lo])
#pragma unroll
  for (int u = 0; u < 16; ++u) { o1[u] = 0.f; o2[u] = 0.f; }
  // P of one block (in place of its accumulators) -> two fp16 operands per k-step -> four products
  auto second = [&](int buf, int blk, const f32x16_t& pv) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      f16x8_t phi, plo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pf = pv[8 * st + e];
        const f16_t h = (f16_t)pf;
        phi[e] = h;
        plo[e] = (f16_t)((pf - (float)h) * kLoScale);
      }
      const f16x8_t ay = *reinterpret_cast<const f16x8_t*>(&yts[buf][blk * (32 * kYtBytes) + ((st * 2 + half) * 32 + (lane & 31)) * 16]);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ay, phi, o1, 0, 0, 0);
      o2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ay, plo, o2, 0, 0, 0);
    }
  };

  float m = -1e30f, s = 0.f;
  const int ntiles = (nc + kMTile - 1) / kMTile;
  fetch(0);
  put(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    fetch((t + 1 < ntiles ? t + 1 : t) * kMTile);
    __builtin_amdgcn_sched_barrier(0);
    f32x16_t acc0 = block(buf, 2 * chalf);
    f32x16_t acc1 = block(buf, 2 * chalf + 1);
    float mn = m;
#pragma unroll
    for (int u = 0; u < 8; ++u) mn = fmaxf(fmaxf(mn, acc0[2 * u]), acc0[2 * u + 1]);
#pragma unroll
    for (int u = 0; u < 8; ++u) mn = fmaxf(fmaxf(mn, acc1[2 * u]), acc1[2 * u + 1]);
    mn = lane_pair_max(mn);
    const float sc = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    const f32x2_t mn2 = {mn, mn};
    f32x2_t add = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x2_t d0 = f32x2_t{acc0[2 * u], acc0[2 * u + 1]} - mn2;
      const f32x2_t d1 = f32x2_t{acc1[2 * u], acc1[2 * u + 1]} - mn2;
      acc0[2 * u] = __builtin_amdgcn_exp2f(d0[0]); acc0[2 * u + 1] = __builtin_amdgcn_exp2f(d0[1]);
      acc1[2 * u] = __builtin_amdgcn_exp2f(d1[0]); acc1[2 * u + 1] = __builtin_amdgcn_exp2f(d1[1]);
      add += f32x2_t{acc0[2 * u], acc0[2 * u + 1]};
      add += f32x2_t{acc1[2 * u], acc1[2 * u + 1]};
    }
    s = s * sc + (add[0] + add[1]);
#pragma unroll
    for (int u = 0; u < 16; ++u) { o1[u] *= sc; o2[u] *= sc; }
    second(buf, 2 * chalf, acc0);
    second(buf, 2 * chalf + 1, acc1);
    __asm__ volatile("" : "+v"(m), "+v"(s));
    __builtin_amdgcn_sched_barrier(0);
    put(buf ^ 1);
    __syncthreads();
  }
  // the two lanes of a row scaled alike all along: their sums add.  Then the two column halves (waves w and w + 2): through
  // LDS (the tile buffers are free now), with a rescale to the common maximum.
  s += __shfl_xor(s, 32, 64);
  float* mrg = reinterpret_cast<float*>(&cs[0][0]);                  // [rgrp][lane][34]
  float* mine = mrg + ((size_t)rgrp * 64 + lane) * 34;
  if (chalf == 1) {
    mine[0] = m; mine[1] = s;
#pragma unroll
    for (int u = 0; u < 16; ++u) { mine[2 + u] = o1[u]; mine[18 + u] = o2[u]; }
  }
  __syncthreads();
  if (chalf != 0 || !rok) return;
  {
    const float m2 = mine[0], s2 = mine[1];
    const float mn = fmaxf(m, m2);
    const float c1 = __builtin_amdgcn_exp2f(m - mn), c2 = __builtin_amdgcn_exp2f(m2 - mn);
    s = s * c1 + s2 * c2;
#pragma unroll
    for (int u = 0; u < 16; ++u) { o1[u] = o1[u] * c1 + mine[2 + u] * c2; o2[u] = o2[u] * c1 + mine[18 + u] * c2; }
    m = mn;
  }
  if (half == 0) {
    const float lse = (m + __builtin_amdgcn_logf(s) - k2 * rr) * kLn2;
    a.pot_new[opot + row] = -a.lam * a.eps * lse;                    // (mode 2: the last extrapolation is not averaged)
  }
  // this lane's 8 dimensions: d = 8 q + 4 half + r for q = 0, 1 (accumulator 4 q + r = the hi rows, 8 + 4 q + r = the lo rows)
  const float inv_s = 1.f / s;
  const float il = 1.f / kLoScale;
  float* go = (which == 0 ? a.grad_xx : a.grad_xy) + (size_t)row * 16;
  const bf16x8_t* rp = reinterpret_cast<const bf16x8_t*>(Rs + (size_t)row * kSplitBytes);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const bf16x8_t ph = rp[0 * 2 + q], pm = rp[1 * 2 + q], pl = rp[2 * 2 + q];      // dimensions 8 q .. 8 q + 7 of x_i - centre
    f32x4_t g4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = 4 * half + r;
      const float xi = ((float)ph[e] + (float)pm[e]) + (float)pl[e];
      const float bary = (o1[4 * q + r] + o1[8 + 4 * q + r] * il) + (o2[4 * q + r] + o2[8 + 4 * q + r] * il) * il;
      g4[r] = xi - bary * inv_s;
    }
    *reinterpret_cast<f32x4_t*>(go + 8 * q + 4 * half) = g4;
  }
}

// per-dimension SUM of both point sets (D <= 16, 256 % D == 0): the centre the matrix-pipe softmin subtracts is
// sums / (N + M) (dense_split_kernel).  Two stages, both with a FIXED summation order -- kCenterWgs partial sums per set,
// then one small workgroup adds them up: the centre cancels mathematically, but its last bits decide how every point
// splits into pieces, and at blur 0.001 the gradient turns a 1e-7 difference there into 2e-3 (float atomics made two
// runs on the same inputs differ by that much).  (As ONE workgroup the whole reduction took 0.34 ms of a 5.3-ms image.)
constexpr int kCenterWgs = 64;
__global__ __launch_bounds__(256) void dense_center_kernel(const float* __restrict__ p, long long n_floats, int D,
                                                           float* __restrict__ partials) {
  __shared__ float part[256];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_floats; i += (long long)gridDim.x * 256) acc += p[i];
  part[threadIdx.x] = acc;               // thread t only ever sees dimension t % D: the stride is a multiple of D
  __syncthreads();
  if ((int)threadIdx.x < D) {
    float t = 0.f;
    for (int j = threadIdx.x; j < 256; j += D) t += part[j];
    partials[blockIdx.x * 16 + threadIdx.x] = t;
  }
}

__global__ void dense_center_finalize_kernel(const float* __restrict__ partials, int n_partials, int D, float* __restrict__ sums) {
  if ((int)threadIdx.x >= D) return;
  float t = 0.f;
  for (int j = 0; j < n_partials; ++j) t += partials[j * 16 + threadIdx.x];
  sums[threadIdx.x] = t;
}

// log weights (with geomloss' -1e5 for non-positive weights)
__global__ void dense_logw_kernel(const float* __restrict__ w, float* __restrict__ lw, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) lw[i] = w[i] > 0.f ? logf(w[i]) : kNegLogD;
}

// debiased cost, dS/dalpha and dS/dx from the final potentials (App. B "value" block + autograd rules)
template <int D>
__global__ __launch_bounds__(256) void dense_finalize_kernel(
    const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ pot, int N, int M,
    float eps, float lam, float rho, const float* __restrict__ grad_xx, const float* __restrict__ grad_xy,
    float* __restrict__ loss_parts, float* __restrict__ gx_out, float* __restrict__ galpha_out) {
  __shared__ float s_part[4];
  const bool unb = rho > 0.f;
  const float w_unb = rho + 0.5f * eps, inv_rho = unb ? 1.f / rho : 0.f;
  const float* ax = pot; const float* bx = pot + N; const float* by = pot + 2 * N; const float* ay = pot + 2 * N + M;
  float part = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const float a = alpha[i];
    float dS;
    float cxx, cxy;          // coefficients of the two softmax-weighted difference sums
    if (unb) {
      const float ea = expf(-ax[i] * inv_rho), eb = expf(-bx[i] * inv_rho);
      dS = w_unb * (ea - eb);
      const float c = -a * w_unb * inv_rho * lam;
      cxx = c * ea; cxy = -c * eb;
    } else {
      dS = bx[i] - ax[i];
      cxx = -a; cxy = a;
    }
    part += a * dS;
    galpha_out[i] = dS;
#pragma unroll
    for (int d = 0; d < D; ++d)
      gx_out[(size_t)i * D + d] = cxx * grad_xx[(size_t)i * D + d] + cxy * grad_xy[(size_t)i * D + d];
  }
  for (int j = blockIdx.x * 256 + threadIdx.x; j < M; j += gridDim.x * 256) {
    const float w = beta[j];
    if (unb) part += w * w_unb * (expf(-by[j] * inv_rho) - expf(-ay[j] * inv_rho));
    else part += w * (ay[j] - by[j]);
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
  __syncthreads();
  // one partial per workgroup, added in block order by dense_loss_sum_kernel: the value does not depend on the order
  // in which the workgroups retire (include/kd6d.h, "reproducible reductions")
  if (threadIdx.x == 0) loss_parts[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

__global__ __launch_bounds__(64) void dense_loss_sum_kernel(const float* __restrict__ parts, int n, float* __restrict__ loss) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += parts[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) *loss = s;
}

// coordinate-wise min / max over all points of both sets -> box diagonal (geomloss max_diameter)
__global__ __launch_bounds__(256) void dense_minmax_kernel(const float* __restrict__ p, long long n, int D,
                                                           float* __restrict__ mn, float* __restrict__ mx) {
  // thread t owns coordinate t % D (256 % D == 0 for D in {2,4,8,16})
  const int d = threadIdx.x % D;
  float lo = INFINITY, hi = -INFINITY;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n * D; i += (long long)gridDim.x * 256) {
    const float v = p[i];
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  // float atomics on the bit pattern need sign care; all reductions here go through int CAS-free min/max of
  // ordered ints: map float -> monotone int
  auto ord = [](float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; };
  atomicMin(reinterpret_cast<int*>(mn) + d, ord(lo));
  atomicMax(reinterpret_cast<int*>(mx) + d, ord(hi));
}

__global__ void dense_diam_kernel(const float* mn, const float* mx, int D, float* diam) {
  if (threadIdx.x == 0) {
    auto unord = [](int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); };
    float s = 0.f;
    for (int d = 0; d < D; ++d) {
      const float e = unord(reinterpret_cast<const int*>(mx)[d]) - unord(reinterpret_cast<const int*>(mn)[d]);
      s += e * e;
    }
    diam[0] = sqrtf(s);
  }
}

template <int D>
int run_dense(const float* x, const float* alpha, const float* y, const float* beta, int N, int M, float blur,
              float scaling, float reach, double diameter, float* ws, float* loss, float* gx, float* galpha,
              hipStream_t st) {
  // workspace: la (N) | lb (M) | pot A (2N+2M) | pot B (2N+2M) | grad_xx (N*D) | grad_xy (N*D)
  float* la = ws;
  float* lb = la + N;
  float* potA = lb + M;
  float* potB = potA + 2 * (size_t)(N + M);
  float* gxx = potB + 2 * (size_t)(N + M);
  float* gxy = gxx + (size_t)N * D;
  float* center = gxy + (size_t)N * D;            // 16 floats (of the 64-float pad)
  float* n2 = center + 64;                        // (N + M) floats, then the prepared points: (N + M) x 96 B
  char* splits = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(n2 + (size_t)(N + M)) + 15) & ~(uintptr_t)15);
  const bool use_mfma = D == 16 && kd6d_opt(KD6D_OPT_SINKHORN_DENSE_MFMA) != 0;
  DenseSplit sp;
  sp.xs = splits; sp.ys = splits + (size_t)N * kSplitBytes; sp.xn2 = n2; sp.yn2 = n2 + N;
  const size_t n128 = ((size_t)N + 127) & ~(size_t)127, m128 = ((size_t)M + 127) & ~(size_t)127;
  char* yt0 = splits + ((size_t)N + M) * kSplitBytes;          // 16-byte aligned: kSplitBytes is a multiple of 16
  sp.xt = yt0; sp.yt = yt0 + n128 * kYtBytes;
  if (use_mfma) {
    // (the partial sums borrow the head of the A-operand array: dense_split_kernel fills it afterwards; >= 4096 floats)
    float* partials = reinterpret_cast<float*>(yt0);
    hipLaunchKernelGGL(dense_center_kernel, dim3(kCenterWgs), dim3(256), 0, st, x, (long long)N * D, D, partials);
    hipLaunchKernelGGL(dense_center_kernel, dim3(kCenterWgs), dim3(256), 0, st, y, (long long)M * D, D, partials + kCenterWgs * 16);
    hipLaunchKernelGGL(dense_center_finalize_kernel, dim3(1), dim3(64), 0, st, (const float*)partials, 2 * kCenterWgs, D, center);
    const float inv_count = 1.0f / (float)((long long)N + M);
    hipLaunchKernelGGL(dense_split_kernel, dim3((unsigned)(n128 / 256 + 1)), dim3(256), 0, st, x, N, (const float*)center, inv_count,
                       splits, n2, yt0);
    hipLaunchKernelGGL(dense_split_kernel, dim3((unsigned)(m128 / 256 + 1)), dim3(256), 0, st, y, M, (const float*)center, inv_count,
                       splits + (size_t)N * kSplitBytes, n2 + N, yt0 + n128 * kYtBytes);
  }
  hipLaunchKernelGGL(dense_logw_kernel, dim3((N + 255) / 256), dim3(256), 0, st, alpha, la, N);
  hipLaunchKernelGGL(dense_logw_kernel, dim3((M + 255) / 256), dim3(256), 0, st, beta, lb, M);
  // epsilon schedule in double, like the reference's python floats (geomloss epsilon_schedule, p = 2)
  if (diameter < 1e-12) diameter = 1e-12;
  const double e_start = 2.0 * log(diameter), e_stop = 2.0 * log((double)blur), e_step = 2.0 * log((double)scaling);
  int n_ar = 0;
  if (e_step < 0.0 && e_start > e_stop) n_ar = (int)ceil((e_stop - e_start) / e_step);
  if (n_ar < 0) n_ar = 0;
  if (n_ar > 4096) n_ar = 4096;
  const int n_eps = n_ar + 2;
  const double rho = reach > 0.f ? (double)reach * (double)reach : -1.0;
  auto eps_at = [&](int i) -> double {
    if (i == 0) return diameter * diameter;
    if (i <= n_ar) return exp(e_start + (double)(i - 1) * e_step);
    return (double)blur * (double)blur;
  };
  const int nmax = N > M ? N : M;
  const dim3 grid((nmax + kRows - 1) / kRows, 4);
  DenseArgs a;
  a.x = x; a.y = y; a.la = la; a.lb = lb; a.N = N; a.M = M; a.grad_xx = gxx; a.grad_xy = gxy;
  float* cur = potA;
  float* nxt = potB;
  // the matrix-pipe kernel while the cancellation of its |r|^2 + |c|^2 - 2 r.c form stays below ~1e-4 in the exponent
  // (header of dense_softmin_mfma_kernel): eps >= kMfmaEpsRel * diameter^2; option sinkhorn.dense_mfma: 0 = never,
  // 1 = by that rule, 2 = every gradient-free pass (tests: how wrong it gets)
  const double mfma_eps_min = kd6d_opt(KD6D_OPT_SINKHORN_DENSE_MFMA) == 2 ? 0.0 : kMfmaEpsRel * diameter * diameter;
  // rows per workgroup of the matrix-pipe softmins: every workgroup streams ALL columns (108-172 bytes each) through LDS,
  // so the launch moves (rows / rows-per-workgroup) x columns x bytes from L2 -- 1.8-2.9 GB at N = M = 16384 with 64
  // rows, which is what bounds it (6-9 TB/s); 128 rows (8 waves) halve that.  Option sinkhorn.dense_rows: -1 = 128 from
  // 8192 rows up | 64 | 128
  const long long rows_opt = kd6d_opt(KD6D_OPT_SINKHORN_DENSE_ROWS);
  const bool rg4 = rows_opt == 128 || (rows_opt != 64 && nmax >= 8192);
  const int mrows = rg4 ? 128 : kMRows;
  const dim3 grid_m((nmax + mrows - 1) / mrows, 4);
  const dim3 block_m(rg4 ? 512 : 256);
  auto launch = [&](int mode, double eps, bool grad) {
    a.pot_old = cur; a.pot_new = nxt; a.mode = mode; a.eps = (float)eps;
    a.lam = rho > 0.0 ? (float)(1.0 / (1.0 + eps / rho)) : 1.f;
    const bool mfma_ok = use_mfma && eps >= mfma_eps_min;
    // below the rule: the matrix pipe screens, the difference form evaluates what passes (dense_softmin_screen_kernel).
    // Threshold: 40 + twice the bound of the approximate exponent's error, k2 2^-21 S with S <= diameter^2 (centred points)
    const bool screen_ok = use_mfma && !mfma_ok && kd6d_opt(KD6D_OPT_SINKHORN_DENSE_SCREEN) != 0;
    a.screen_th = (float)(40.0 + (0.5 / eps) * 1.4426950408889634 * diameter * diameter * (1.0 / 1048576.0));
    auto set_which = [&](int w0, int w1, int w2, int w3) { a.which[0] = w0; a.which[1] = w1; a.which[2] = w2; a.which[3] = w3; };
    if (!grad) {
      set_which(0, 1, 2, 3);
      if (mfma_ok) {
        if constexpr (D == 16) {
          if (rg4) hipLaunchKernelGGL(dense_softmin_mfma_kernel<4>, grid_m, block_m, 0, st, a, sp);
          else hipLaunchKernelGGL(dense_softmin_mfma_kernel<2>, grid_m, block_m, 0, st, a, sp);
        }
      } else if (screen_ok) {
        if constexpr (D == 16) {
          if (rg4) hipLaunchKernelGGL((dense_softmin_screen_kernel<false, 4>), grid_m, block_m, 0, st, a, sp);
          else hipLaunchKernelGGL((dense_softmin_screen_kernel<false, 2>), grid_m, block_m, 0, st, a, sp);
        }
      } else {
        hipLaunchKernelGGL((dense_softmin_kernel<D, false>), grid, dim3(kThreadsD), 0, st, a);
      }
    } else {
      // the last extrapolation: only the two softmins over the rows of x (a_x, b_x) carry a gradient -- difference form,
      // with the softmax-weighted sums; the other two (b_y, a_y) are plain potentials
      set_which(0, 3, 0, 0);
      if (mfma_ok && kd6d_opt(KD6D_OPT_SINKHORN_DENSE_MFMA) != 3) {
        if constexpr (D == 16)
          hipLaunchKernelGGL(dense_softmin_mfma_grad_kernel, dim3((N + kMRows - 1) / kMRows, 2), dim3(256), 0, st, a, sp);
      } else if (screen_ok) {
        if constexpr (D == 16) {
          if (rg4) hipLaunchKernelGGL((dense_softmin_screen_kernel<true, 4>), dim3((N + mrows - 1) / mrows, 2), block_m, 0, st, a, sp);
          else hipLaunchKernelGGL((dense_softmin_screen_kernel<true, 2>), dim3((N + mrows - 1) / mrows, 2), block_m, 0, st, a, sp);
        }
      } else {
        hipLaunchKernelGGL((dense_softmin_kernel<D, true, 4>), dim3((nmax + 127) / 128, 2), dim3(kThreadsD), 0, st, a);
      }
      set_which(1, 2, 0, 0);
      if (mfma_ok) {
        if constexpr (D == 16) {
          if (rg4) hipLaunchKernelGGL(dense_softmin_mfma_kernel<4>, dim3(grid_m.x, 2), block_m, 0, st, a, sp);
          else hipLaunchKernelGGL(dense_softmin_mfma_kernel<2>, dim3(grid_m.x, 2), block_m, 0, st, a, sp);
        }
      } else if (screen_ok) {
        if constexpr (D == 16) {
          if (rg4) hipLaunchKernelGGL((dense_softmin_screen_kernel<false, 4>), dim3(grid_m.x, 2), block_m, 0, st, a, sp);
          else hipLaunchKernelGGL((dense_softmin_screen_kernel<false, 2>), dim3(grid_m.x, 2), block_m, 0, st, a, sp);
        }
      } else {
        hipLaunchKernelGGL((dense_softmin_kernel<D, false>), dim3(grid.x, 2), dim3(kThreadsD), 0, st, a);
      }
    }
    float* t = cur; cur = nxt; nxt = t;
  };
  launch(0, eps_at(0), false);
  double eps = eps_at(0);
  for (int it = 0; it < n_eps; ++it) {
    eps = eps_at(it);
    launch(1, eps, false);
  }
  launch(2, eps, true);
  const float lam = rho > 0.0 ? (float)(1.0 / (1.0 + eps / rho)) : 1.f;
  int nb = (nmax + 255) / 256;
  if (nb > 512) nb = 512;
  // the other potential buffer (2 (N + M) floats >= nb) is free after the last pass: it takes the per-workgroup partials
  hipLaunchKernelGGL(dense_finalize_kernel<D>, dim3(nb), dim3(256), 0, st, alpha, beta, cur, N, M, (float)eps, lam,
                     (float)rho, gxx, gxy, nxt, gx, galpha);
  hipLaunchKernelGGL(dense_loss_sum_kernel, dim3(1), dim3(64), 0, st, (const float*)nxt, nb, loss);
  return KD6D_OK;
}

}  // namespace

extern "C" int64_t kd6d_sinkhorn_dense_workspace_floats(int N, int M, int D) {
  // log weights, two potential sets, two gradient partials, centre (+ pad), and for D = 16 the matrix-pipe softmin's
  // prepared points: |p|^2 and 96 bytes of bf16 pieces per point
  // ... and 64 bytes of fp16 pieces per point (sets padded to multiples of 128 points) for the gradient's second product
  return (int64_t)N + M + 4 * ((int64_t)N + M) + 2 * (int64_t)N * D + 64 +
         (D == 16 ? 25 * ((int64_t)N + M) + 16 + 16 * ((((int64_t)N + 127) & ~127ll) + (((int64_t)M + 127) & ~127ll)) : 0);
}

extern "C" int kd6d_sinkhorn_dense_diameter(const float* x, const float* y, int N, int M, int D, float* scratch64,
                                            float* diam_out, void* stream) {
  KD6D_CHECK_ARG(x && y && scratch64 && diam_out && N > 0 && M > 0, "kd6d_sinkhorn_dense_diameter: bad arguments");
  KD6D_CHECK_ARG(D == 2 || D == 4 || D == 8 || D == 16, "kd6d_sinkhorn_dense_diameter: D=%d (supported: 2, 4, 8, 16)", D);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // ordered-int encodings of +inf / -inf
  int init[64];
  for (int i = 0; i < 32; ++i) { init[i] = 0x7f800000; init[32 + i] = (int)0xff800000 ^ 0x7fffffff; }
  if (hipMemcpyAsync(scratch64, init, sizeof(init), hipMemcpyHostToDevice, st) != hipSuccess) {
    kd6d_set_error("kd6d_sinkhorn_dense_diameter: memcpy failed");
    return KD6D_ERR_LAUNCH;
  }
  int nb = (int)(((long long)N * D + 1023) / 1024);
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(dense_minmax_kernel, dim3(nb), dim3(256), 0, st, x, (long long)N, D, scratch64, scratch64 + 32);
  nb = (int)(((long long)M * D + 1023) / 1024);
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(dense_minmax_kernel, dim3(nb), dim3(256), 0, st, y, (long long)M, D, scratch64, scratch64 + 32);
  hipLaunchKernelGGL(dense_diam_kernel, dim3(1), dim3(64), 0, st, scratch64, scratch64 + 32, D, diam_out);
  KD6D_CHECK_LAUNCH("kd6d_sinkhorn_dense_diameter");
  return KD6D_OK;
}

extern "C" int kd6d_sinkhorn_dense_fwd_bwd(const float* x, const float* alpha, const float* y, const float* beta,
                                           int N, int M, int D, float p, float blur, float scaling, float reach,
                                           double diameter, float* workspace, int64_t workspace_floats,
                                           float* loss, float* grad_x, float* grad_alpha, void* stream) {
  KD6D_CHECK_ARG(x && alpha && y && beta && workspace && loss && grad_x && grad_alpha,
                 "kd6d_sinkhorn_dense_fwd_bwd: null pointer");
  KD6D_CHECK_ARG(N > 0 && M > 0, "kd6d_sinkhorn_dense_fwd_bwd: empty set");
  if (p != 2.0f) {
    kd6d_set_error("kd6d_sinkhorn_dense_fwd_bwd: only p=2 is implemented (got %g)", (double)p);
    return KD6D_ERR_UNSUPPORTED;
  }
  KD6D_CHECK_ARG(blur > 0.f && scaling > 0.f && scaling < 1.f && diameter > 0.0,
                 "kd6d_sinkhorn_dense_fwd_bwd: need blur>0, 0<scaling<1 and the diameter of the point cloud "
                 "(kd6d_sinkhorn_dense_diameter, or the caller's bound as with geomloss' diameter= argument)");
  KD6D_CHECK_ARG(workspace_floats >= kd6d_sinkhorn_dense_workspace_floats(N, M, D),
                 "kd6d_sinkhorn_dense_fwd_bwd: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc;
  switch (D) {
    case 2: rc = run_dense<2>(x, alpha, y, beta, N, M, blur, scaling, reach, diameter, workspace, loss, grad_x, grad_alpha, st); break;
    case 4: rc = run_dense<4>(x, alpha, y, beta, N, M, blur, scaling, reach, diameter, workspace, loss, grad_x, grad_alpha, st); break;
    case 8: rc = run_dense<8>(x, alpha, y, beta, N, M, blur, scaling, reach, diameter, workspace, loss, grad_x, grad_alpha, st); break;
    case 16: rc = run_dense<16>(x, alpha, y, beta, N, M, blur, scaling, reach, diameter, workspace, loss, grad_x, grad_alpha, st); break;
    default:
      kd6d_set_error("kd6d_sinkhorn_dense_fwd_bwd: D=%d (supported: 2, 4, 8, 16)", D);
      return KD6D_ERR_UNSUPPORTED;
  }
  if (rc) { kd6d_set_error("kd6d_sinkhorn_dense_fwd_bwd: launch failed"); return rc; }
  KD6D_CHECK_LAUNCH("kd6d_sinkhorn_dense_fwd_bwd");
  return KD6D_OK;
}
