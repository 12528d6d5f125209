// 3x3 / stride 1 / pad 1 "halo patch" implicit-GEMM convolution (forward and data gradient) and the pair bracket
// that issues two such convolutions of identical geometry as one launch.  Split from conv_igemm.hip; the kernel
// description is above halo_tile.
#include "conv_common.h"

using namespace kd6d_detail;

namespace {

// ---------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1, bf16, C % 64 == 0: "halo patch" implicit GEMM.
//
// The generic kernels fetch the im2col operand tap by tap, i.e. every input pixel 9 times, and
// at these layer sizes they are bound by the ~28 B/clk a CU can pull from its XCD's L2, not by
// MFMA.  Here a workgroup keeps the input pixels of its tile PLUS a halo (packed rows
// [m0 - halo, m0 + BP + halo), halo = max level width + 1) resident in LDS for one 64-channel
// chunk and builds all 9 taps from it: the pixel operand is fetched once instead of 9 times, so a
// k-step streams only the weight tile (BC x 128 B).  The MFMA pixel fragment of tap (dy,dx) is a
// plain ds_read_b128 at patch row (m - patch_lo) + dy*W + dx; out-of-image taps read a zero row.
// k order is (chunk, tap, ci) instead of (tap, ci): only the fp32 summation order changes.
// Everything travels by LDS-DMA: weights through a 3-deep ring (BC/8/waves instructions per wave
// and k-step), the next chunk's patch double-buffered behind the current one; counted vmcnt,
// one raw s_barrier per k-step.  Levels of a multi-level (head) launch may share a tile.
// ---------------------------------------------------------------------------
//
// HMAX: largest halo (level width + 1) the LDS patch is sized for, 65 or 33.  PDB: the next chunk's patch is
// double-buffered behind the current one; without it the workgroup stops at a chunk boundary until the new patch
// has landed -- in exchange a 128x128 tile fits 76 KB of LDS, so TWO workgroups (of this or of another stream's
// launch) share a CU and one's prologue, chunk stalls and epilogue run under the other's k-loop.
template <int BP, int BC, int WP, int WC, int MODE, int HMAX = 65, bool PDB = true, bool NORM = false>
__device__ __forceinline__ void halo_tile(const ConvParams& p, int halo, int total_rows, int bid, int nwg) {
  using T = bf16_t;
  constexpr int NW = WP * WC;
  constexpr int PI = BP / WP / 16;
  constexpr int CI = BC / WC / 16;
  constexpr int PSLOT = (BP + 2 * HMAX + 7) / 8 + 1; // 8-row groups of a patch (+1: the zero row lives in the last)
  constexpr int PL = (PSLOT + NW - 1) / NW;          // patch LDS-DMA instructions per wave per chunk
  // double-buffered: every wave issues the same PL pieces per chunk (the counted vmcnt waits rely on it), slots past
  // PSLOT are padding; single buffer: only the PSLOT slots exist (its loads are followed by vmcnt(0))
  constexpr int PATCH_BYTES = (PDB ? PL * NW : PSLOT) * 1024;
  constexpr int ZERO_ROW = (PDB ? PL * NW : PSLOT) * 8 - 1;   // never a real patch row: always filled from the zero page
  constexpr int WL = BC / 8 / NW;                    // weight LDS-DMA instructions per wave per k-step
  constexpr int WSTAGE = BC * 128;
  static_assert(BC % (8 * NW) == 0 && PI >= 1 && CI >= 1, "tile shape");
  static_assert(WL + PL <= 63, "vmcnt range");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wring = smem;                      // 3 weight stages first: their fragment reads use immediate offsets
  char* const pbuf = smem + 3 * WSTAGE;          // 2 patch buffers (PDB) or 1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wp = wave % WP;
  const int wc = wave / WP;

  const int wg = tile_of_workgroup<NORM>(p, bid, nwg);
  const int tile_c = p.p_fastest ? wg / p.n_ptiles : wg % p.n_ctiles;
  const int tile_p = p.p_fastest ? wg % p.n_ptiles : wg / p.n_ctiles;
  const int m0 = tile_p * BP;
  const int n0 = tile_c * BC;
  const int patch_lo = m0 - halo;

  const int lrow = lane >> 3;
  const int gk = (lane & 7) ^ lrow;
  const int fr = lane & 15;
  const int fq = lane >> 4;

  const T* __restrict__ src = reinterpret_cast<const T*>(p.src);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(p.wgt);
  const char* zero = reinterpret_cast<const char*>(kd6d_zero_page);

  // ---- fragment addresses, once per tile.  Pixel fragment of (q, tap): byte offset (from the start of LDS) of the
  // lane's 16 B of patch row (m - patch_lo) + dy*W + dx in patch buffer 0, k-granule fq; out-of-image taps point at
  // the zero row.  The second 32-deep half of a k-step is the same address ^ 64 (granule 4 + fq), the other patch
  // buffer + PATCH_BYTES (toggled in place once per chunk): the k-loop itself computes no addresses -- it used to
  // issue 3.3 vector instructions per MFMA for them, more issue cycles than the matrix pipe's.
  int pa[PI][9];
#pragma unroll
  for (int q = 0; q < PI; ++q) {
    const int m = m0 + wp * (BP / WP) + q * 16 + fr;
    const RowInfo ri = decode_row(p, m);      // packed identically on both sides: src row == dst row == m
    const int base = m - patch_lo;
    const bool in = m < p.M;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = MODE == MODE_FWD ? tap / 3 - 1 : 1 - tap / 3;
      const int dx = MODE == MODE_FWD ? tap % 3 - 1 : 1 - tap % 3;
      const bool ok = in && (unsigned)(ri.y + dy) < (unsigned)ri.src_h && (unsigned)(ri.x + dx) < (unsigned)ri.src_w;
      const int r = ok ? base + dy * ri.src_w + dx : ZERO_ROW;
      pa[q][tap] = 3 * WSTAGE + r * 128 + ((fq ^ (r & 7)) << 4);
    }
  }
  int wa[CI];          // weight fragment (c): offset inside a ring stage, k-granule fq (second half: ^ 64)
#pragma unroll
  for (int c = 0; c < CI; ++c) {
    const int row = wc * (BC / WC) + c * 16 + fr;
    wa[c] = row * 128 + ((fq ^ (row & 7)) << 4);
  }

  // ---- loaders ----
  int wofs[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    const int n = n0 + 8 * (wave + NW * i) + lrow;
    wofs[i] = n < p.N ? n * p.K + gk * 8 : -1;
  }
  auto issue_w = [&](int stage, int chunk, int tap) {
    char* base = wring + stage * WSTAGE + wave * 1024;
    const int kk0 = tap * p.C + chunk * 64;
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const void* g = zero;
      if (wofs[i] >= 0) g = wgt + ((size_t)wofs[i] + (size_t)kk0);
      glds16(g, base + i * NW * 1024);
    }
  };
  // the next chunk's patch goes out a few pieces per k-step over taps 0..6 (all PL pieces at once put ~PL x 100
  // cycles of LDS-DMA issue in front of one k-step's 400 cycles of MFMA work)
  constexpr int PPT = (PL + 6) / 7;
  auto issue_patch = [&](int buf, int chunk, int i0, int i1) {
    char* base = pbuf + buf * PATCH_BYTES + wave * 1024;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const int slot = wave + NW * i;
      if (!PDB && slot >= PSLOT) continue;
      const int prow = 8 * slot + lrow;
      const int row = patch_lo + prow;
      const void* g = zero;
      if (prow < BP + 2 * halo && row >= 0 && row < total_rows)
        g = src + ((size_t)row * (size_t)p.C + (size_t)(chunk * 64 + gk * 8));
      glds16(g, base + i * NW * 1024);
    }
  };
  auto pieces_at = [](int tap) constexpr {       // patch pieces issued at k-step `tap` of a chunk
    if (tap < 0 || tap > 6) return 0;
    const int lo = tap * PPT, hi = (tap + 1) * PPT;
    return (hi < PL ? hi : PL) - (lo < PL ? lo : PL);
  };

  f32x4_t acc[CI][PI];
#pragma unroll
  for (int c = 0; c < CI; ++c)
#pragma unroll
    for (int q = 0; q < PI; ++q) acc[c][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nchunk = p.C >> 6;
  const int nk = nchunk * 9;

  // prologue: queue = [PATCH(0), W(0), W(1)]
  issue_patch(0, 0, 0, PL);
  issue_w(0, 0, 0);
  issue_w(1, 0, 1);
  int c2 = 0, t2 = 2;          // (chunk, tap) of W(kt+2)

  // Software pipeline across the barrier: a k-step's second 32-deep half is multiplied at the TOP of the next
  // step, from registers, right behind the barrier -- it covers the LDS latency of that step's first fragment
  // reads; the LDS-DMA pieces go out at the end of a step, behind 12 queued MFMAs per wave (a piece costs its wave
  // ~100+ cycles of issue; straight behind the barrier both waves of a SIMD paid that with an empty matrix pipe).
  Frag<T> ga[CI], gb[PI];
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const bool more_patch = PDB && chunk + 1 < nchunk;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kt = chunk * 9 + tap;
      // retire W(kt).  Younger than it: W(kt+1) and the patch pieces of the two previous k-steps (issue order
      // inside a step is W first, then the pieces)
      if (kt + 1 >= nk) {
        wait_vmcnt<0>();
      } else if (more_patch) {
        switch (pieces_at(tap - 1) + pieces_at(tap - 2)) {
          case 0: wait_vmcnt<WL>(); break;
          case 1: wait_vmcnt<WL + 1>(); break;
          case 2: wait_vmcnt<WL + 2>(); break;
          case 3: wait_vmcnt<WL + 3>(); break;
          case 4: wait_vmcnt<WL + 4>(); break;
          default: wait_vmcnt<WL>(); break;
        }
      } else {
        wait_vmcnt<WL>();
      }
      __builtin_amdgcn_s_barrier();
      const int wsoff = (tap % 3) * WSTAGE;      // ring slot of W(kt): 9 % 3 == 0, so it is tap % 3 in every chunk
      Frag<T> fa[CI], fb[PI];
#pragma unroll
      for (int c = 0; c < CI; ++c) fa[c].v = *reinterpret_cast<const bf16x8_t*>(smem + wsoff + wa[c]);
#pragma unroll
      for (int q = 0; q < PI; ++q) fb[q].v = *reinterpret_cast<const bf16x8_t*>(smem + pa[q][tap]);
      if (tap > 0 || chunk > 0) {               // second half of the previous k-step
#pragma unroll
        for (int c = 0; c < CI; ++c)
#pragma unroll
          for (int q = 0; q < PI; ++q) mma(ga[c], gb[q], acc[c][q]);
      }
#pragma unroll
      for (int c = 0; c < CI; ++c)
#pragma unroll
        for (int q = 0; q < PI; ++q) mma(fa[c], fb[q], acc[c][q]);
#pragma unroll
      for (int c = 0; c < CI; ++c) ga[c].v = *reinterpret_cast<const bf16x8_t*>(smem + wsoff + (wa[c] ^ 64));
#pragma unroll
      for (int q = 0; q < PI; ++q) gb[q].v = *reinterpret_cast<const bf16x8_t*>(smem + (pa[q][tap] ^ 64));
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 2 < nk) {
        issue_w((tap + 2) % 3, c2, t2);
        if (++t2 == 9) { t2 = 0; ++c2; }
      }
      if (more_patch && pieces_at(tap) > 0)
        issue_patch((chunk + 1) & 1, chunk + 1, tap * PPT, (tap + 1) * PPT < PL ? (tap + 1) * PPT : PL);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PDB) {     // the next chunk reads the other patch buffer
      const int flip = (chunk & 1) ? -PATCH_BYTES : PATCH_BYTES;
#pragma unroll
      for (int q = 0; q < PI; ++q)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) pa[q][tap] += flip;
    } else if (chunk + 1 < nchunk) {
      // single buffer: every wave's fragment reads of this chunk are in registers (lgkmcnt 0) before the patch is
      // overwritten; the new one is complete (vmcnt 0; the next k-step's barrier publishes it) before it is read
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue_patch(0, chunk + 1, 0, PL);
      wait_vmcnt<0>();
    }
  }
#pragma unroll
  for (int c = 0; c < CI; ++c)
#pragma unroll
    for (int q = 0; q < PI; ++q) mma(ga[c], gb[q], acc[c][q]);
  __syncthreads();             // the epilogue reuses the LDS the last k-step may still be reading
  conv_epilogue_full<T, BP, BC, WP, WC, true, NORM>(p, acc, m0, n0, wp, wc, lane, reinterpret_cast<float*>(smem), bid, nwg, tile_c);
}

// second launch bound = waves per SIMD the register budget has to leave room for: the single-buffer variants are
// built to run two workgroups per CU
template <int BP, int BC, int WP, int WC, int MODE, int HMAX = 65, bool PDB = true, bool NORM = false>
__global__ __launch_bounds__(WP* WC * 64, PDB ? 1 : WP * WC / 2) void conv3x3_halo_kernel(const ConvParams p, int halo, int total_rows) {
  halo_tile<BP, BC, WP, WC, MODE, HMAX, PDB, NORM>(p, halo, total_rows, blockIdx.x, gridDim.x);
}

// Two convolutions of identical geometry (the cls and the pose tower layer of the head: different tensors and
// weights, same shapes) as ONE launch: workgroups [0, tiles_a) run `pa`, the rest `pb`.  A student tower layer
// alone is 170 tiles of 128x128 on 256 CUs, one 128-KB workgroup per CU -- a second stream cannot use the idle
// third; as a pair the two layers are 228 tiles of 192x128, one full round for both.
template <int BP, int BC, int WP, int WC, int MODE, int HMAX = 65, bool PDB = true, bool NORM = false>
__global__ __launch_bounds__(WP* WC * 64, PDB ? 1 : WP * WC / 2) void conv3x3_halo_pair_kernel(const ConvParams pa, const ConvParams pb,
                                                                        int halo, int total_rows, int tiles_a) {
  if ((int)blockIdx.x < tiles_a) halo_tile<BP, BC, WP, WC, MODE, HMAX, PDB, NORM>(pa, halo, total_rows, blockIdx.x, tiles_a);
  else halo_tile<BP, BC, WP, WC, MODE, HMAX, PDB, NORM>(pb, halo, total_rows, blockIdx.x - tiles_a, gridDim.x - tiles_a);
}

// kd6d_conv2d_pair_begin / _end: between the two calls, halo-kernel launches are recorded instead of issued; two
// recorded launches of the same kernel variant and geometry go out as one conv3x3_halo_pair_kernel launch.
struct HaloRecord {
  ConvParams q;
  int halo, total_rows, tiles;
  hipStream_t st;
  void (*single)(const HaloRecord&);
  void (*pair)(const HaloRecord&, const HaloRecord&);
};
struct PairState {
  bool active = false;
  int count = 0;
  HaloRecord rec[2];
};
// the pair bracket belongs to the calling thread's current context (kd6d_ctx)
PairState& pair_state() {
  kd6d_ctx* c = kd6d_current_ctx();
  if (!c->pair) {
    c->pair = new PairState;
    c->pair_free = [](void* q) { delete static_cast<PairState*>(q); };
  }
  return *static_cast<PairState*>(c->pair);
}
#define g_pair (pair_state())

template <int BP, int BC, int WP, int WC, int MODE, int HMAX, bool PDB>
size_t halo_lds() {
  constexpr int NW = WP * WC;
  constexpr int PSLOT = (BP + 2 * HMAX + 7) / 8 + 1;
  constexpr int PL = (PSLOT + NW - 1) / NW;
  return (PDB ? (size_t)2 * PL * NW : (size_t)PSLOT) * 1024 + (size_t)3 * BC * 128;
}

template <int BP, int BC, int WP, int WC, int MODE, int HMAX = 65, bool PDB = true, bool NORM = false>
void halo_issue_single(const HaloRecord& r) {
  const size_t lds = halo_lds<BP, BC, WP, WC, MODE, HMAX, PDB>();
  auto kern = conv3x3_halo_kernel<BP, BC, WP, WC, MODE, HMAX, PDB, NORM>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(r.tiles), dim3(WP * WC * 64), lds, r.st, r.q, r.halo, r.total_rows);
}

template <int BP, int BC, int WP, int WC, int MODE, int HMAX = 65, bool PDB = true, bool NORM = false>
void halo_issue_pair(const HaloRecord& a, const HaloRecord& b) {
  const size_t lds = halo_lds<BP, BC, WP, WC, MODE, HMAX, PDB>();
  auto kern = conv3x3_halo_pair_kernel<BP, BC, WP, WC, MODE, HMAX, PDB, NORM>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles + b.tiles), dim3(WP * WC * 64), lds, a.st, a.q, b.q, a.halo, a.total_rows,
                     a.tiles);
}

template <int BP, int BC, int WP, int WC, int MODE, int HMAX = 65, bool PDB = true, bool NORM = false>
void launch_halo(const ConvParams& p, int halo, int total_rows, hipStream_t st) {
  HaloRecord r;
  r.q = p;
  r.q.n_ctiles = (p.N + BC - 1) / BC;
  const int ptiles = (p.M + BP - 1) / BP;
  set_tile_order(r.q, ptiles, BP, BC);
  r.halo = halo; r.total_rows = total_rows; r.tiles = ptiles * r.q.n_ctiles; r.st = st;
  r.single = &halo_issue_single<BP, BC, WP, WC, MODE, HMAX, PDB, NORM>;
  r.pair = &halo_issue_pair<BP, BC, WP, WC, MODE, HMAX, PDB, NORM>;
  if (plan_only(r.tiles, WP * WC * 64, halo_lds<BP, BC, WP, WC, MODE, HMAX, PDB>(), NORM)) return;
  if (g_pair.active && g_pair.count < 2) { g_pair.rec[g_pair.count++] = r; return; }
  r.single(r);
}

// 3x3/s1/p1 layers with C % 64 == 0 on maps at most 80 wide, both sides packed identically.
template <int MODE>
bool dispatch_halo(const ConvParams& p, const kd6d_conv_geom* g, hipStream_t st) {
  const int force = (int)kd6d_opt(KD6D_OPT_CONV_HALO);
  if (force == 0) return false;
  if (p.ks != 3 || p.stride != 1 || p.pad != 1 || (p.C & 63) || (p.N & 3)) return false;
  if (p.N < 64 && p.N > 32) return false;
  if (!p.norm_dst && p.stats_replicas > 1) return false;      // replica rows of the batch statistics: register-staged kernel
  int wmax = 0, rows = 0;
  for (int s = 0; s < g->nseg; ++s) {
    const kd6d_seg& q = g->seg[s];
    if (q.in_row0 != q.out_row0 || q.in_row0 != rows) return false;
    if (q.in_w > wmax) wmax = q.in_w;
    rows += g->batch * q.in_h * q.in_w;
  }
  if (wmax > (kd6d_opt(KD6D_OPT_CONV_HALO_WIDE) != 0 ? 80 : 64)) return false;      // (80: the 60 x 80 level of 480 x 640 full frames)
  const int halo = wmax + 1;
  // measured on the step's layers (tools/bench_conv.py), all variants with 8 waves (2 per SIMD: with 4 waves
  // the same 128x128 tile is 25-40 % slower, one wave per SIMD cannot hide the LDS-DMA / fragment latency):
  //   256x128 once it yields >= 150 workgroups (teacher head, stage 2);
  //   128x128 from >= 160 workgroups (teacher stage 3, FPN 32x32 level, student head towers fwd + dgrad);
  //   128x64  from >= 64 workgroups (teacher stage 4, student FPN 32x32 level) -- ahead of split-K;
  //   192x128 / 64x64 / 128x32: the tile-count corner cases below;
  // below that the layer goes to split-K / the generic kernels.
  // inside a kd6d_conv2d_pair_begin/_end bracket the launch shares the device with its twin: count tiles twice
  const int pf = g_pair.active ? 2 : 1;
  const int pt128 = pf * ((p.M + 127) / 128);
  int pick = 0;
  const int ct128 = (p.N + 127) / 128;
  const int ncu = cached_cu_count();
  const int ct64 = (p.N + 63) / 64;
  // few result channels (cls logits, dgrad into the narrow student stages): 128 x 32, or 64 x 64 on small maps
  if (p.N <= 32) pick = pt128 <= ncu / 2 ? 9 : 5;
  // 128 x 64 tiles would occupy at most half of the CUs: 64 x 64 (FPN 16x16 level, stage 5, student FPN)
  else if (pt128 * ct64 <= ncu / 2 && pf * ((p.M + 63) / 64) * ct64 >= 64) pick = 9;
  // 192 x 128 where it turns 256-pixel tiles that leave a third of the CUs idle into one full round (teacher head
  // towers: 172 tiles of 256 pixels on 256 CUs -> 228 tiles of 192)
  else if (pf * ((p.M + 255) / 256) * ct128 >= 150 && pf * ((p.M + 255) / 256) * ct128 <= (3 * ncu) / 4 &&
           pf * ((p.M + 191) / 192) * ct128 <= ncu) pick = 6;
  else if (pf * ((p.M + 255) / 256) * ((p.N + 127) / 128) >= 150) pick = 1;
  else if (pt128 * ((p.N + 127) / 128) >= 160) pick = 3;
  else if (pt128 * ((p.N + 63) / 64) >= 64) pick = 4;
  // maps up to 32 wide (halo <= 33): the single-patch-buffer twin of the picked tile, 38-74 KB of LDS instead of
  // 104-144, so that two workgroups -- of this launch or of the other stream's -- share a CU.  Alone on the device a
  // twin is as fast as its original or up to 40 % slower (chunk-boundary stalls, no partner to cover them); inside
  // the step the pairs give +4 % (4889-4918 -> 5097 images/s, interleaved runs; profiles/r02_halo_pairing.md)
  if (kd6d_opt(KD6D_OPT_CONV_HALO_PAIRING) != 0 && halo <= 33) {
    // (96 x 128 tiles for the 342-tile tower shape -- 456 tiles on 512 slots instead of 86 CUs carrying two tiles of 128 x
    //  128 and 170 one -- were built and measured in round 3: 5064-5099 against 5184-5192 images/s, interleaved; removed)
    // (256 x 128, one workgroup per CU, is what the counts above pick from ~40 000 rows on -- the teacher's towers over
    //  the 32 images of a grouped pass: 2.25 us per image against 1.75 for the twins, 1.91 for 192 x 128; tools/bench_conv.py
    //  --batch 32 --opt conv.halo=N)
    if (pick == 3 || pick == 6 || pick == 1) pick = 12;
    else if (pick == 4) pick = 13;
    else if (pick == 9) pick = 14;
    else if (pick == 5) pick = 15;
  }
  if (force > 0 && (force < 10 || halo <= 33)) pick = force;      // 11..15: the twins, maps <= 32 wide only
  if (pick == 0) return false;
  if (p.norm_dst) {
    // a fused normalisation behind the convolution (kd6d_conv2d_fwd_norm): compiled into the 128 x 128 forward tiles only
    if constexpr (MODE == MODE_FWD) {
      if (halo > 65) launch_halo<128, 128, 4, 2, MODE, 81, true, true>(p, halo, rows, st);
      else if (halo > 33 || kd6d_opt(KD6D_OPT_CONV_HALO_PAIRING) == 0) launch_halo<128, 128, 4, 2, MODE, 65, true, true>(p, halo, rows, st);
      else launch_halo<128, 128, 4, 2, MODE, 33, false, true>(p, halo, rows, st);
      return true;
    }
    return false;
  }
  if (pick == 11) { launch_halo<128, 128, 2, 2, MODE, 33, false>(p, halo, rows, st); return true; }
  if (pick == 12) { launch_halo<128, 128, 4, 2, MODE, 33, false>(p, halo, rows, st); return true; }
  if (pick == 13) { launch_halo<128, 64, 4, 2, MODE, 33, false>(p, halo, rows, st); return true; }
  if (pick == 14) { launch_halo<64, 64, 4, 2, MODE, 33, false>(p, halo, rows, st); return true; }
  if (pick == 15) { launch_halo<128, 32, 4, 1, MODE, 33, false>(p, halo, rows, st); return true; }
  if (halo > 65) {
    // maps 65 ... 80 wide: the same tiles with the patch sized for a halo of 81 rows (the 256 x 128 tile then takes
    // exactly the CU's 160 KB)
    if (pick == 1) launch_halo<256, 128, 4, 2, MODE, 81>(p, halo, rows, st);
    else if (pick == 3 || pick == 6) launch_halo<128, 128, 4, 2, MODE, 81>(p, halo, rows, st);
    else if (pick == 4) launch_halo<128, 64, 4, 2, MODE, 81>(p, halo, rows, st);
    else if (pick == 5) launch_halo<128, 32, 4, 1, MODE, 81>(p, halo, rows, st);
    else if (pick == 9) launch_halo<64, 64, 4, 2, MODE, 81>(p, halo, rows, st);
    else launch_halo<128, 128, 4, 2, MODE, 81>(p, halo, rows, st);
    return true;
  }
  if (pick == 1) launch_halo<256, 128, 4, 2, MODE>(p, halo, rows, st);
  else if (pick == 3) launch_halo<128, 128, 4, 2, MODE>(p, halo, rows, st);      // 8 waves on the 128x128 tile
  else if (pick == 4) launch_halo<128, 64, 4, 2, MODE>(p, halo, rows, st);
  else if (pick == 5) launch_halo<128, 32, 4, 1, MODE>(p, halo, rows, st);
  else if (pick == 6) launch_halo<192, 128, 4, 2, MODE>(p, halo, rows, st);
  else if (pick == 9) launch_halo<64, 64, 4, 2, MODE>(p, halo, rows, st);
  else launch_halo<128, 128, 2, 2, MODE>(p, halo, rows, st);
  return true;
}

}  // namespace

bool kd6d_detail::dispatch_halo_fwd(const ConvParams& p, const kd6d_conv_geom* g, hipStream_t st) {
  return dispatch_halo<MODE_FWD>(p, g, st);
}
bool kd6d_detail::dispatch_halo_dgrad(const ConvParams& p, const kd6d_conv_geom* g, hipStream_t st) {
  return dispatch_halo<MODE_DGRAD>(p, g, st);
}

#undef g_pair

namespace {
struct CtxScope {           // run an entry point under an explicit context
  kd6d_ctx* saved;
  explicit CtxScope(kd6d_ctx* c) : saved(kd6d_ctx_current()) { if (c) kd6d_ctx_make_current(c); }
  ~CtxScope() { kd6d_ctx_make_current(saved); }
};
}  // namespace
#define g_pair (pair_state())

extern "C" int kd6d_ctx_conv2d_pair_begin(kd6d_ctx* c) { CtxScope s(c); return kd6d_conv2d_pair_begin(); }
extern "C" int kd6d_ctx_conv2d_pair_end(kd6d_ctx* c) { CtxScope s(c); return kd6d_conv2d_pair_end(); }
extern "C" int kd6d_ctx_conv2d_pair_pending(kd6d_ctx* c) { CtxScope s(c); return kd6d_conv2d_pair_pending(); }

extern "C" int kd6d_conv2d_pair_begin(void) {
  KD6D_CHECK_ARG(!g_pair.active, "kd6d_conv2d_pair_begin: already inside a pair bracket");
  g_pair.active = true;
  g_pair.count = 0;
  return KD6D_OK;
}

extern "C" int kd6d_conv2d_pair_pending(void) { return g_pair.active ? g_pair.count : 0; }

extern "C" int kd6d_conv2d_pair_end(void) {
  KD6D_CHECK_ARG(g_pair.active, "kd6d_conv2d_pair_end: no pair bracket open");
  g_pair.active = false;
  const int n = g_pair.count;
  g_pair.count = 0;
  const HaloRecord& a = g_pair.rec[0];
  const HaloRecord& b = g_pair.rec[1];
  if (n == 2 && a.pair == b.pair && a.halo == b.halo && a.total_rows == b.total_rows && a.st == b.st) {
    a.pair(a, b);
  } else {
    for (int i = 0; i < n; ++i) g_pair.rec[i].single(g_pair.rec[i]);
  }
  KD6D_CHECK_LAUNCH("kd6d_conv2d_pair_end");
  for (int i = 0; i < n; ++i) {      // group statistics of the levels the epilogue leaves to a separate pass (bf16 kernel)
    const int rc = kd6d_detail::stats_followup(g_pair.rec[i].q, g_pair.rec[i].q.out_f32 != 0, g_pair.rec[i].st);
    if (rc) return rc;
  }
  return KD6D_OK;
}
