// Grouped weight gradient (bf16): the dW of MANY convolution layers in one launch.
//
// Replaces the weight-gradient half of torch's conv2d autograd for the head and FPN layers of the student
// (models/model.py:64-83, 438-451 via backbone/common.py:316-324 semantics: fp32 dW, bias gradient = column sums
// of dY) -- in the reference one cuDNN/MIOpen call per layer, here one launch per network section.
//
// Why grouped.  dW[n][j] = sum over pixels of dY[m][n] * im2col(X)[m][j] is a GEMM whose long axis (pixels,
// 21760 for the head at B = 16) is the CONTRACTION: one layer alone has 3..30 output tiles for 256 CUs, so it
// must be split ~30x along the pixels and every split flushes a partial tile (#workgroups x tile bytes of
// traffic per layer -- two thirds of the old per-layer launch was that flush).  With all layers of a section in
// one work list there are enough tiles that a workgroup keeps a long k-loop (hundreds of 64-pixel steps) and the
// flush is paid once per section.
//
// Workgroup tile.  128 (or 16) result channels n  x  [3 taps of one kernel row ky] x 128 input channels: the
// three taps (ky, 0..2) read the SAME staged X rows at row offsets 0, 1, 2 and the same dY tile, so a k-step
// moves 16 KB of dY + 17 KB of X for 3 x the MFMA work of a one-tap tile (the per-CU L2 fill rate, not the
// matrix pipe, bounds a 128 x 128 one-tap tile).
//
// k-space = PADDED pixel positions.  Every image row of W pixels is W + 2 positions, the two extra ones holding
// zeros (fetched from a zero page), so tap kx of position q is position q + kx - 1 with no per-pixel validity
// mask anywhere in the k-loop; rows above / below the image are zero-page fetches decided by the loader, which
// computes a source address per lane anyway.  Tiles land in LDS in their natural layout (row = position) by
// LDS-DMA (2-stage ring, counted vmcnt, one raw s_barrier per step) with the chunk
// swizzle applied on the source side; MFMA fragments (8 consecutive positions of one channel per lane) come
// out of ds_read_b64_tr_b16.  A = X^T fragment, B = dY fragment, so a lane ends with 4 consecutive j of one n:
// 16-B stores of the fp32 partial tile into a slab; a second small kernel sums the slabs of a tile's splits and
// adds them into dW / dbias (plain stores: 4-5x the byte rate of fp32 atomics, and bitwise reproducible).
// The bias gradient is one extra MFMA per dY fragment against an all-ones operand.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "kd6d_common.h"

namespace {

constexpr int kMaxSeg = KD6D_MAX_SEG;
constexpr int KSTEP = 64;                    // k-positions per step
constexpr int kThreads = 512;
constexpr int kSlabTile = 128 * 384;         // floats of the largest partial tile
constexpr int kSlabStride = kSlabTile + 128; // + the bias partial
constexpr unsigned OOB = 0x80000000u;        // buffer offset beyond every tensor: the fetch returns zeros

struct GLevel {
  int H, W, PW, P;        // PW = W + 2*padx padded row length, P = batch*H*PW positions
  int in_row0, out_row0;
  int step0;              // first k-step of the level inside a tile's step space
  float inv_pw, inv_h;
  int pad_;
};

struct GProblem {
  const bf16_t* x;
  const bf16_t* dy;
  float* dw;
  float* dbias;
  int Cin, Cout, ks, J;
  int nseg, steps, x_bytes, dy_bytes;
  GLevel lv[kMaxSeg];
};

struct GTile {
  int problem, n0, c0, ky;
  int variant, first_wg, nsplit, want_bias;
};

struct GWork {
  int tile, step_lo, step_hi, pad_;
};

struct GRBlock {
  int tile, chunk;
};

struct GHeader {
  int n_problems, n_tiles, n_work, n_rblocks;
  int off_problems, off_tiles, off_work, off_rblocks;
  int pad0, pad1, pad2, pad3;
};

__device__ const uint4 kd6d_wg_zero_page[16] = {};

__device__ __forceinline__ int fdiv(int x, int d, float inv) {
  int q = (int)((float)x * inv);
  q += ((q + 1) * d <= x) ? 1 : 0;
  q -= (q * d > x) ? 1 : 0;
  return q;
}

template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// chunk swizzle of a tile row (16-B chunks), by row pitch: the 8 rows {r..r+3, r+8..r+11} a 32-lane half of
// ds_read_b64_tr_b16 touches land on 8 different 32-B bank groups for EVERY r (tap offsets shift r by 0, 1, 2)
template <int PITCH> __device__ __forceinline__ int swz(int row) {
  if (PITCH == 256) return ((row & 3) << 2) | ((row >> 2) & 3);
  if (PITCH == 128) return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;
  return 0;
}

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

// BN result channels x [NT taps x CJ input channels], 8 waves as WN (n) x WJ (j), NS-stage LDS ring
template <int BN, int WN, int WJ, int CJ, int NT, int NS>
__device__ __forceinline__ void wgrad_body(const GProblem& p, const GTile& t, const GWork& w, float* __restrict__ slab,
                                           char* smem) {
  constexpr int NI = BN / WN / 16;                 // dY fragments per wave
  constexpr int CW = CJ / WJ;                      // input channels per wave and tap
  constexpr int JI = CW / 16;
  constexpr int DP = BN * 2, XP = CJ * 2;          // row pitches in bytes
  constexpr int DRPI = 1024 / DP, XRPI = 1024 / XP;// rows per LDS-DMA wave-instruction
  constexpr int DCPR = DP / 16, XCPR = XP / 16;    // chunks per row
  constexpr int PADX = NT == 3 ? 1 : 0;
  constexpr int XROWS = KSTEP + 2 * PADX;
  constexpr int DI = KSTEP / DRPI;                 // LDS-DMA instructions of the dY tile / of the X tile per step
  constexpr int XI = (XROWS + XRPI - 1) / XRPI;
  constexpr int DS = (DI + 7) / 8, XS = (XI + 7) / 8;   // slots per wave: instruction id = wave + 8 * slot
  constexpr int DBYTES = KSTEP * DP;
  constexpr int STAGE = DBYTES + XI * 1024;
  static_assert(WN * WJ == 8 && NI >= 1 && JI >= 1 && KSTEP % DRPI == 0, "tile shape");
  static_assert(NS >= 2 && NS <= 4 && NS * STAGE <= 160 * 1024, "LDS ring");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform on purpose: scalar branches / addresses
  const int wn = wave % WN, wj = wave / WN;
  const int n0 = t.n0, c0 = t.c0, ky = t.ky;
  const int pady = p.ks >> 1;

  // ---- loader.  Waves 0-3 fetch the dY tile, waves 4-7 the X tile: piece (= one 1-KB LDS-DMA instruction) j of a
  // wave is tile piece (wave & 3) + 4 j, so a wave runs ONE kind of loop with no per-piece role test.  A lane always
  // fetches the same tile row and 16-B chunk, so its position advances by exactly KSTEP per step: (padded x, map
  // row y, byte offset) live in registers and move with adds and one conditional wrap -- integer multiplies run at
  // quarter rate, a division-per-row loader cost 2.5 us per step against 0.7 us of MFMA work, and every piece
  // costs its wave ~100 cycles of issue on top, so the instruction count around a piece is what bounds this
  // kernel.  Fetches go through buffer descriptors: a lane whose position is padding, outside the image or beyond
  // the level gets an out-of-range offset, which the hardware answers with zeros.
  // Every wave of a role issues NB pieces unconditionally; the TI % 4 left-over pieces go to the first waves of the
  // role as ONE extra piece with its own state (conditional pieces inside the unrolled loop made the compiler
  // shuffle the whole carried state through v_mov at every piece).
  constexpr int DNB = DI / 4, DEX = DI % 4, XNB = XI / 4, XEX = XI % 4;
  constexpr int NBMAX = DNB > XNB ? DNB : XNB;
  const bool is_x = wave >= 4;
  const int lw = wave & 3;
  const int nb = is_x ? XNB : DNB;
  const bool has_extra = lw < (is_x ? XEX : DEX);
  const int my_n = nb + (has_extra ? 1 : 0);                       // pieces this wave issues per step
  const __amdgpu_buffer_rsrc_t rs = is_x ? __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000)
                                         : __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
  int s_xp[NBMAX + 1], s_y[NBMAX + 1], s_rel[NBMAX + 1];        // [NBMAX] = the extra piece
  unsigned s_off[NBMAX + 1];
  bool chok = true;
  int cur_level = -1, lv_W = 1, lv_PW = 1, lv_H = 1, lv_P = 0, lv_q0 = 0, lv_next = 0x7fffffff;
  int lv_r64 = 0, lv_dy64 = 0, lv_adv = 0, lv_wrap = 0;
  const int C2 = (is_x ? p.Cin : p.Cout) * 2;                    // bytes per pixel row of this wave's tensor
  const int rpi = is_x ? XRPI : DRPI, cpr = is_x ? XCPR : DCPR;
  const int kyo = is_x ? ky - pady : 0;                          // map-row shift of the fetched rows
  const int ex_id = 4 * nb + lw;                                 // tile piece of the extra one

  auto enter_level = [&](int step) {        // (re)initialise the carried state at a level boundary: divisions only here
    int l = 0;
#pragma unroll
    for (int s = 1; s < kMaxSeg; ++s)
      if (s < p.nseg && step >= p.lv[s].step0) l = s;
    const GLevel& L = p.lv[l];
    cur_level = l;
    lv_W = L.W; lv_PW = L.PW; lv_H = L.H; lv_P = L.P;
    lv_q0 = (step - L.step0) * KSTEP;
    lv_next = (l + 1 < p.nseg) ? p.lv[l + 1].step0 : 0x7fffffff;
    const int d64 = KSTEP / L.PW;
    lv_r64 = KSTEP - d64 * L.PW;
    lv_dy64 = d64 % L.H;
    lv_adv = (d64 * L.W + lv_r64) * C2;
    lv_wrap = (L.W - L.PW) * C2;
#pragma unroll
    for (int s = 0; s <= NBMAX; ++s) {
      const int id = s < NBMAX ? lw + 4 * s : ex_id;
      const int row = id * rpi + lane / cpr;
      const int csrc = (lane % cpr) ^ (is_x ? swz<XP>(row) : swz<DP>(row));
      const int rel = is_x ? row - PADX : row;           // position relative to the step's first one
      const int qq = lv_q0 + rel + L.PW;                 // >= 0 (the X tile starts one position early)
      const int irp = fdiv(qq, L.PW, L.inv_pw);          // = map-row index + 1
      const int ir = irp - 1;
      s_rel[s] = rel;
      s_xp[s] = qq - irp * L.PW;
      const int yq = ir + L.H;                           // >= 0
      s_y[s] = yq - fdiv(yq, L.H, L.inv_h) * L.H;
      const int srow = is_x ? L.in_row0 + (ir + kyo) * L.W : L.out_row0 + ir * L.W;
      s_off[s] = (unsigned)((srow + s_xp[s] - PADX) * C2 + ((is_x ? c0 : n0) + csrc * 8) * 2);
      if (s == 0) chok = is_x || (n0 + csrc * 8 < p.Cout);     // the chunk is the same for every piece of a lane
    }
  };

  // one piece: fetch, then advance the lane's position by KSTEP.  EDGE: the step touches the first / last
  // positions of the level (range checks needed); NEEDY: rows above / below the image exist (X tile, ky off-centre)
  auto piece = [&](char* lds, int& xp_, int& y_, unsigned& off_, int rel_, auto EDGE, auto NEEDY) {
    bool ok = chok & ((unsigned)(xp_ - PADX) < (unsigned)lv_W);
    if (decltype(NEEDY)::value) ok = ok & ((unsigned)(y_ + kyo) < (unsigned)lv_H);
    if (decltype(EDGE)::value) ok = ok & (lv_q0 + rel_ >= 0) & (lv_q0 + rel_ < lv_P);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, ok ? off_ : OOB, 0, 0, 0);
    const unsigned xp = (unsigned)(xp_ + lv_r64);
    const bool wrap = xp >= (unsigned)lv_PW;
    xp_ = (int)(wrap ? xp - lv_PW : xp);
    off_ += (unsigned)(lv_adv + (wrap ? lv_wrap : 0));
    if (decltype(NEEDY)::value) {
      const unsigned y = (unsigned)(y_ + lv_dy64 + (wrap ? 1 : 0));
      y_ = (int)(y >= (unsigned)lv_H ? y - lv_H : y);
    }
  };
  auto issue_impl = [&](char* tbase, auto EDGE, auto NEEDY) {
    if (is_x) {
#pragma unroll
      for (int s = 0; s < XNB; ++s) piece(tbase + (lw + 4 * s) * 1024, s_xp[s], s_y[s], s_off[s], s_rel[s], EDGE, NEEDY);
    } else {
#pragma unroll
      for (int s = 0; s < DNB; ++s) piece(tbase + (lw + 4 * s) * 1024, s_xp[s], s_y[s], s_off[s], s_rel[s], EDGE, NEEDY);
    }
    if (has_extra) piece(tbase + ex_id * 1024, s_xp[NBMAX], s_y[NBMAX], s_off[NBMAX], s_rel[NBMAX], EDGE, NEEDY);
  };
  const bool need_y = kyo != 0;
  auto issue = [&](int stage, int step) {
    if (cur_level < 0 || step >= lv_next) enter_level(step);
    char* tbase = smem + stage * STAGE + (is_x ? DBYTES : 0);
    const bool edge = (lv_q0 < PADX) | (lv_q0 + KSTEP + PADX > lv_P);
    if (edge) {                             // rare: one general instance (it also tracks y)
      issue_impl(tbase, std::true_type{}, std::true_type{});
    } else if (need_y) {
      issue_impl(tbase, std::false_type{}, std::true_type{});
    } else {
      issue_impl(tbase, std::false_type{}, std::false_type{});
    }
    lv_q0 += KSTEP;
  };

  f32x4_t acc[NT][JI][NI];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < JI; ++b)
#pragma unroll
      for (int c = 0; c < NI; ++c) acc[a][b][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x4_t bacc[NI];
#pragma unroll
  for (int c = 0; c < NI; ++c) bacc[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = t.want_bias != 0 && wj == 0;
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

  // ---- fragment addresses: the lane-dependent part (row, swizzled chunk) once, outside the loop.  swz() depends on
  // row & 15 only, so the 32-row second half of a step is the same address + 32 rows.
  typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
  const int fr_ = lane & 15, g_ = lane >> 4, q_ = fr_ >> 2, pp_ = fr_ & 3;
  int ad[NI][2], ax[NT][JI][2];
#pragma unroll
  for (int c = 0; c < NI; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = 8 * g_ + q_ + 4 * h;
      ad[c][h] = r * DP + (((2 * (wn * NI + c) + (pp_ >> 1)) ^ swz<DP>(r)) << 4) + 8 * (pp_ & 1);
    }
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < JI; ++b)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = a + 8 * g_ + q_ + 4 * h;
        ax[a][b][h] = DBYTES + r * XP + (((2 * (wj * JI + b) + (pp_ >> 1)) ^ swz<XP>(r)) << 4) + 8 * (pp_ & 1);
      }
  auto frag = [&](int addr_lo, int addr_hi, int imm) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(smem + addr_lo + imm));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(smem + addr_hi + imm));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  const int nsteps = w.step_hi - w.step_lo;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nsteps) issue(s, w.step_lo + s);
  int stage = 0;
  for (int i = 0; i < nsteps; ++i) {
    // retire this step's fetches, leave the younger steps' in flight (vmcnt counts in issue order)
    const int younger = nsteps - 1 - i;
    const int keep = (younger >= NS - 2 ? NS - 2 : younger) * my_n;
    if (keep >= 10) wait_vm<10>();
    else if (keep >= 8) wait_vm<8>();
    else if (keep >= 6) wait_vm<6>();
    else if (keep >= 5) wait_vm<5>();
    else if (keep >= 4) wait_vm<4>();
    else if (keep >= 3) wait_vm<3>();
    else if (keep >= 2) wait_vm<2>();
    else if (keep >= 1) wait_vm<1>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (i + NS - 1 < nsteps) {
      int s2 = stage + NS - 1;
      if (s2 >= NS) s2 -= NS;
      issue(s2, w.step_lo + i + NS - 1);
    }
    {
      const int sb = stage * STAGE;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        bf16x8_t fd[NI];
#pragma unroll
        for (int c = 0; c < NI; ++c) fd[c] = frag(ad[c][0] + sb, ad[c][1] + sb, kc * 32 * DP);
        if (do_bias) {
#pragma unroll
          for (int c = 0; c < NI; ++c) bacc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fd[c], bacc[c], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          bf16x8_t fx[JI];
#pragma unroll
          for (int b = 0; b < JI; ++b) fx[b] = frag(ax[a][b][0] + sb, ax[a][b][1] + sb, kc * 32 * XP);
#pragma unroll
          for (int b = 0; b < JI; ++b)
#pragma unroll
            for (int c = 0; c < NI; ++c)
              acc[a][b][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[b], fd[c], acc[a][b][c], 0, 0, 0);
        }
      }
    }
    if (++stage == NS) stage = 0;
  }

  // ---- partial tile -> slab[n][tap][ci]: a lane holds 4 consecutive ci of one n ---------------------
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < JI; ++b)
#pragma unroll
      for (int c = 0; c < NI; ++c) {
        const int n = wn * (BN / WN) + c * 16 + fr;
        const int jl = a * CJ + wj * CW + b * 16 + fq * 4;
        *reinterpret_cast<f32x4_t*>(slab + (size_t)n * (NT * CJ) + jl) = acc[a][b][c];
      }
  if (do_bias && fq == 0) {
#pragma unroll
    for (int c = 0; c < NI; ++c) slab[kSlabTile + wn * (BN / WN) + c * 16 + fr] = bacc[c][0];
  }
}

enum { V_128x3 = 0, V_16x3 = 1, V_128x1 = 2, V_16x1 = 3 };

__global__ __launch_bounds__(kThreads) void wgrad_group_kernel(const char* __restrict__ plan, float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GHeader* h = reinterpret_cast<const GHeader*>(plan);
  const GWork w = reinterpret_cast<const GWork*>(plan + h->off_work)[blockIdx.x];
  const GTile t = reinterpret_cast<const GTile*>(plan + h->off_tiles)[w.tile];
  const GProblem& p = reinterpret_cast<const GProblem*>(plan + h->off_problems)[t.problem];
  float* my = slab + (size_t)blockIdx.x * kSlabStride;
  switch (t.variant) {
    case V_128x3: wgrad_body<128, 2, 4, 128, 3, 2>(p, t, w, my, smem); break;
    case V_16x3: wgrad_body<16, 1, 8, 128, 3, 2>(p, t, w, my, smem); break;
    case V_128x1: wgrad_body<128, 2, 4, 128, 1, 2>(p, t, w, my, smem); break;
    default: wgrad_body<16, 1, 8, 128, 1, 2>(p, t, w, my, smem); break;
  }
}

// dW += sum over a tile's splits of the partial tiles; dbias likewise.  One block = 1024 floats of one tile.
__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(const char* __restrict__ plan, const float* __restrict__ slab) {
  const GHeader* h = reinterpret_cast<const GHeader*>(plan);
  const GRBlock rb = reinterpret_cast<const GRBlock*>(plan + h->off_rblocks)[blockIdx.x];
  const GTile t = reinterpret_cast<const GTile*>(plan + h->off_tiles)[rb.tile];
  const GProblem& p = reinterpret_cast<const GProblem*>(plan + h->off_problems)[t.problem];
  const int BN = (t.variant == V_128x3 || t.variant == V_128x1) ? 128 : 16;
  const int NT = (t.variant == V_128x3 || t.variant == V_16x3) ? 3 : 1;
  const int width = NT * 128;
  const float* base = slab + (size_t)t.first_wg * kSlabStride;
  if (rb.chunk < 0) {                       // bias partials
    const int n = threadIdx.x;
    if (n < BN && t.n0 + n < p.Cout && p.dbias != nullptr) {
      float s = 0.f;
      for (int k = 0; k < t.nsplit; ++k) s += base[(size_t)k * kSlabStride + kSlabTile + n];
      p.dbias[t.n0 + n] += s;
    }
    return;
  }
  const int idx = rb.chunk * 1024 + threadIdx.x * 4;
  if (idx >= BN * width) return;
  const int n = idx / width, jl = idx - n * width;
  if (t.n0 + n >= p.Cout) return;
  f32x4_t s = *reinterpret_cast<const f32x4_t*>(base + idx);
  for (int k = 1; k < t.nsplit; ++k) s += *reinterpret_cast<const f32x4_t*>(base + (size_t)k * kSlabStride + idx);
  const int tap = jl >> 7, ci = jl & 127;
  float* dst = p.dw + (size_t)(t.n0 + n) * (size_t)p.J + (size_t)((t.ky * p.ks + tap) * p.Cin + t.c0 + ci);
  f32x4_t d = *reinterpret_cast<f32x4_t*>(dst);
  d += s;
  *reinterpret_cast<f32x4_t*>(dst) = d;
}

bool item_supported(const kd6d_conv_geom* g, int dtype) {
  if (g == nullptr || dtype != KD6D_BF16) return false;
  if (g->stride != 1 || !(g->ksize == 1 || g->ksize == 3) || g->pad != g->ksize / 2) return false;
  if (g->cin % 128 != 0 || g->cout % 8 != 0 || g->cout < 8) return false;
  if (g->nseg < 1 || g->nseg > kMaxSeg || g->batch < 1) return false;
  for (int s = 0; s < g->nseg; ++s) {
    const kd6d_seg& q = g->seg[s];
    if (q.in_h != q.out_h || q.in_w != q.out_w || q.in_h < 1 || q.in_w < 1) return false;
    if ((long long)g->batch * q.in_h * (q.in_w + 2) >= (1ll << 24)) return false;      // fdiv range
    const long long rows = (long long)q.in_row0 + (long long)g->batch * q.in_h * q.in_w;
    if (rows * (g->cin > g->cout ? g->cin : g->cout) * 2 >= (1ll << 31)) return false;   // 32-bit byte offsets
  }
  return true;
}

// two stages = 67 KB (2, 3 and 4 stages measured the same launch time: the loop waits on the per-CU miss window, not
// on ring depth), so two of these workgroups -- or one and a two-per-CU halo tile -- share a CU
constexpr int kLdsBytes = 2 * (KSTEP * 256 + 17 * 1024);

}  // namespace

extern "C" int kd6d_wgrad_group_supported(const kd6d_conv_geom* g, int dtype) { return item_supported(g, dtype) ? 1 : 0; }

extern "C" int64_t kd6d_wgrad_group_plan(const kd6d_wgrad_item* items, int n_items, int dtype, int n_workgroups,
                                         void* plan_host, int64_t plan_capacity, int32_t* info) {
  if (!items || n_items < 1 || !info || n_workgroups < 1) {
    kd6d_set_error("kd6d_wgrad_group_plan: bad arguments");
    return KD6D_ERR_ARG;
  }
  std::vector<GProblem> probs(n_items);
  std::vector<GTile> tiles;
  std::vector<double> cost;                      // relative cost of one k-step of the tile
  for (int i = 0; i < n_items; ++i) {
    const kd6d_conv_geom* g = &items[i].geom;
    if (!item_supported(g, dtype) || !items[i].x || !items[i].dy || !items[i].dw) {
      kd6d_set_error("kd6d_wgrad_group_plan: item %d is not a layer the grouped kernel takes "
                     "(bf16, stride 1, 1x1 or 3x3 'same', cin %% 128 == 0)", i);
      return KD6D_ERR_UNSUPPORTED;
    }
    GProblem& p = probs[i];
    memset(&p, 0, sizeof(p));
    p.x = reinterpret_cast<const bf16_t*>(items[i].x);
    p.dy = reinterpret_cast<const bf16_t*>(items[i].dy);
    p.dw = items[i].dw;
    p.dbias = items[i].dbias;
    p.Cin = g->cin; p.Cout = g->cout; p.ks = g->ksize; p.J = g->ksize * g->ksize * g->cin;
    p.nseg = g->nseg;
    {
      long long rin = 0, rout = 0;
      for (int q = 0; q < g->nseg; ++q) {
        const long long a = (long long)g->seg[q].in_row0 + (long long)g->batch * g->seg[q].in_h * g->seg[q].in_w;
        const long long b = (long long)g->seg[q].out_row0 + (long long)g->batch * g->seg[q].out_h * g->seg[q].out_w;
        rin = a > rin ? a : rin; rout = b > rout ? b : rout;
      }
      p.x_bytes = (int)(rin * g->cin * 2); p.dy_bytes = (int)(rout * g->cout * 2);
    }
    const int padx = g->ksize / 2;
    int step = 0;
    for (int s = 0; s < g->nseg; ++s) {
      GLevel& L = p.lv[s];
      L.H = g->seg[s].in_h; L.W = g->seg[s].in_w; L.PW = L.W + 2 * padx;
      L.P = g->batch * L.H * L.PW;
      L.in_row0 = g->seg[s].in_row0; L.out_row0 = g->seg[s].out_row0;
      L.step0 = step;
      L.inv_pw = 1.0f / (float)L.PW; L.inv_h = 1.0f / (float)L.H;
      step += (L.P + KSTEP - 1) / KSTEP;
    }
    p.steps = step;
    const bool wide = g->cout > 16;
    const int bn = wide ? 128 : 16;
    const int variant = g->ksize == 3 ? (wide ? V_128x3 : V_16x3) : (wide ? V_128x1 : V_16x1);
    for (int n0 = 0; n0 < g->cout; n0 += bn)
      for (int c0 = 0; c0 < g->cin; c0 += 128)
        for (int ky = 0; ky < g->ksize; ++ky) {
          GTile t;
          t.problem = i; t.n0 = n0; t.c0 = c0; t.ky = ky; t.variant = variant;
          t.first_wg = 0; t.nsplit = 1;
          t.want_bias = (items[i].dbias != nullptr && c0 == 0 && ky == g->ksize / 2) ? 1 : 0;
          tiles.push_back(t);
          const double mfma = (wide ? 8.0 : 1.0) * (g->ksize == 3 ? 3.0 : 1.0);
          cost.push_back(mfma + 3.0);             // + the fixed part of a step (fetch issue, barrier)
        }
  }
  double total = 0.0;
  for (size_t k = 0; k < tiles.size(); ++k) total += cost[k] * probs[tiles[k].problem].steps;
  const double per_wg = total / (double)n_workgroups;
  // Work order.  Workgroup ids are dealt round-robin over the 8 XCDs (observed; speed only), and the grid is about
  // one workgroup per CU, so everything runs at once.  The tiles of one layer that differ only in ky / n0 stream
  // the same X rows (shifted by a map row) and dY rows: their splits are laid out as runs of 8 consecutive ids,
  // split s of every such tile on XCD s % 8, so each byte leaves HBM once and is re-read from that XCD's L2.
  std::vector<GWork> work;
  std::vector<size_t> order(tiles.size());
  for (size_t k = 0; k < tiles.size(); ++k) order[k] = k;
  for (size_t k = 0; k < tiles.size(); ++k) {
    const int steps = probs[tiles[k].problem].steps;
    int ns = (int)(cost[k] * steps / per_wg + 0.5);
    if (ns >= 6) ns = (ns + 3) / 8 * 8 < 8 ? 8 : (ns + 3) / 8 * 8;
    if (ns > (steps + 3) / 4) ns = (steps + 3) / 4;      // at least 4 steps per split
    if (ns < 1) ns = 1;
    tiles[k].nsplit = ns;
  }
  // tiles with 8-aligned runs first (their alignment must not be disturbed by the short ones)
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
    return (tiles[a].nsplit % 8 == 0) > (tiles[b].nsplit % 8 == 0);
  });
  for (size_t oi = 0; oi < order.size(); ++oi) {
    const size_t k = order[oi];
    const int steps = probs[tiles[k].problem].steps;
    const int ns = tiles[k].nsplit;
    tiles[k].first_wg = (int)work.size();
    for (int s = 0; s < ns; ++s) {
      GWork w;
      w.tile = (int)k;
      w.step_lo = (int)((long long)steps * s / ns);
      w.step_hi = (int)((long long)steps * (s + 1) / ns);
      w.pad_ = 0;
      work.push_back(w);
    }
  }
  std::vector<GRBlock> rblocks;
  for (size_t k = 0; k < tiles.size(); ++k) {
    const int v = tiles[k].variant;
    const int floats = ((v == V_128x3 || v == V_128x1) ? 128 : 16) * ((v == V_128x3 || v == V_16x3) ? 384 : 128);
    for (int c = 0; c < (floats + 1023) / 1024; ++c) rblocks.push_back(GRBlock{(int)k, c});
    if (tiles[k].want_bias) rblocks.push_back(GRBlock{(int)k, -1});
  }
  GHeader h;
  memset(&h, 0, sizeof(h));
  h.n_problems = n_items; h.n_tiles = (int)tiles.size(); h.n_work = (int)work.size(); h.n_rblocks = (int)rblocks.size();
  auto align = [](size_t v) { return (v + 63) / 64 * 64; };
  size_t off = align(sizeof(GHeader));
  h.off_problems = (int)off; off = align(off + sizeof(GProblem) * probs.size());
  h.off_tiles = (int)off; off = align(off + sizeof(GTile) * tiles.size());
  h.off_work = (int)off; off = align(off + sizeof(GWork) * work.size());
  h.off_rblocks = (int)off; off = align(off + sizeof(GRBlock) * rblocks.size());
  info[0] = h.n_work; info[1] = h.n_rblocks;
  const long long slab_floats = (long long)h.n_work * kSlabStride;
  info[2] = (int32_t)(slab_floats & 0x7fffffff); info[3] = (int32_t)(slab_floats >> 31);
  if (plan_host == nullptr) return (int64_t)off;                // size query
  if ((int64_t)off > plan_capacity) {
    kd6d_set_error("kd6d_wgrad_group_plan: plan needs %lld bytes, buffer has %lld", (long long)off, (long long)plan_capacity);
    return KD6D_ERR_ARG;
  }
  char* out = reinterpret_cast<char*>(plan_host);
  memset(out, 0, off);
  memcpy(out, &h, sizeof(h));
  memcpy(out + h.off_problems, probs.data(), sizeof(GProblem) * probs.size());
  memcpy(out + h.off_tiles, tiles.data(), sizeof(GTile) * tiles.size());
  memcpy(out + h.off_work, work.data(), sizeof(GWork) * work.size());
  memcpy(out + h.off_rblocks, rblocks.data(), sizeof(GRBlock) * rblocks.size());
  return (int64_t)off;
}

extern "C" int kd6d_wgrad_group_launch(const void* plan_dev, int n_workgroups, int n_reduce_blocks, float* slab_dev,
                                       void* stream) {
  KD6D_CHECK_ARG(plan_dev && slab_dev && n_workgroups > 0 && n_reduce_blocks > 0, "kd6d_wgrad_group_launch: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_group_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_group_kernel, dim3(n_workgroups), dim3(kThreads), kLdsBytes, st,
                     reinterpret_cast<const char*>(plan_dev), slab_dev);
  KD6D_CHECK_LAUNCH("kd6d_wgrad_group_launch");
  hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(n_reduce_blocks), dim3(256), 0, st,
                     reinterpret_cast<const char*>(plan_dev), slab_dev);
  KD6D_CHECK_LAUNCH("kd6d_wgrad_group_launch (reduce)");
  return KD6D_OK;
}
