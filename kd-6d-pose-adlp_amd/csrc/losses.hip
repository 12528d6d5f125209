// Loss-side kernels of the KD step: teacher knowledge extraction, SSC target assignment,
// focal loss, object-space keypoint loss, and the scatter of loss gradients back into the
// head's logit-gradient tensors.  The reference runs these as Python loops with one or more
// device->host syncs per image; here each is one launch for the batch and nothing syncs.
//
//   teacher_select   <- postprocess/postprocess_kd.py:22-203 (PnP gate treated as true)
//   ssc_assign       <- losses/loss.py:164-268 (random pick = n smallest of caller-supplied keys)
//   focal fwd/bwd    <- losses/loss.py:20-40
//   student_points   <- losses/kd_loss.py:40-71,152 + models/model.py:144-166 (decode + reg loss)
//   loss_backward    <- autograd of kd_loss.py:47-86 and loss_libs.py:8-12 (chain into logits)
//
// Logit layout: cls (rows,16) fp32 [15 classes + 1 pad], reg (rows,240) fp32, rows packed
// level-major / image / row-major (the conv kernels' packed NHWC order).
#include "kd6d_common.h"
#include "kd6d_det.h"

namespace {

constexpr int kT = 256;

struct Levels {
  int n, batch;
  int h[KD6D_MAX_SEG], w[KD6D_MAX_SEG], row0[KD6D_MAX_SEG];
  float stride[KD6D_MAX_SEG], size[KD6D_MAX_SEG];
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// (value, index) arg-max over the workgroup; ties -> smallest index.  All threads get the result.
__device__ __forceinline__ void block_argmax(float& v, int& idx, float* s_v, int* s_i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = v; s_i[threadIdx.x >> 6] = idx; }
  __syncthreads();
  v = s_v[0]; idx = s_i[0];
  for (int w = 1; w < kT / 64; ++w)
    if (s_v[w] > v || (s_v[w] == v && s_i[w] < idx)) { v = s_v[w]; idx = s_i[w]; }
}

__device__ __forceinline__ void level_fields(const Levels& L, int l, int& h, int& w, int& row0,
                                             float& stride, float& size) {
  h = 0; w = 0; row0 = 0; stride = 1.f; size = 1.f;
#pragma unroll
  for (int s = 0; s < KD6D_MAX_SEG; ++s)
    if (s == l) { h = L.h[s]; w = L.w[s]; row0 = L.row0[s]; stride = L.stride[s]; size = L.size[s]; }
}

__device__ __forceinline__ void inv2x2(const float* bt, float* ai) {
  const float a = bt[0], b = bt[1], c = bt[3], d = bt[4];
  const float det = a * d - b * c;
  ai[0] = d / det; ai[1] = -b / det; ai[2] = -c / det; ai[3] = a / det;
}

// ------------------------------------------------------------------------------------
// Teacher knowledge extraction: one workgroup per image.
// out: t_cnt[b]; t_kp[(b*cap + slot)*16 + k*2 + {0,1}] full-frame px; t_score[(b*cap+slot)*8 + k]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void teacher_select_kernel(
    const float* __restrict__ cls, const float* __restrict__ reg, Levels L,
    const float* __restrict__ bbox_trans, float th, float positive_num, float positive_lambda, int cap,
    float frame_w, float frame_h, int* __restrict__ t_cnt, float* __restrict__ t_kp,
    float* __restrict__ t_score, int* __restrict__ t_row, float* __restrict__ t_kp_norm,
    float* __restrict__ t_beta, const int* __restrict__ class_filter, const int* __restrict__ n_gt) {
  extern __shared__ float s_all[];      // candidate scores of the current class, all levels of this image
  __shared__ float s_v[kT / 64];
  __shared__ int s_i[kT / 64];
  __shared__ unsigned s_mask;
  __shared__ int s_nk[KD6D_MAX_SEG];
  __shared__ int s_total;
  __shared__ int s_pick_row[64];
  __shared__ float s_pick_cx[64], s_pick_cy[64], s_pick_sz[64], s_pick_bv[64];
  const int b = blockIdx.x;
  // knowledge extraction: one workgroup per image, the first class that emits cells (postprocess_kd.py:86-90);
  // pose candidates (class_filter != null): workgroup (b, g) handles exactly the class of ground-truth slot g
  // (postprocess.py:118-121 keeps only labels present in target.class_ids) and writes output block b*MAX_GT + g
  const int ob = class_filter ? b * KD6D_MAX_GT + (int)blockIdx.y : b;
  int c_lo = 0, c_hi = 15;
  if (class_filter) {
    const int g = blockIdx.y;
    const int cf = g < n_gt[b] ? class_filter[b * KD6D_MAX_GT + g] : -1;
    if (cf < 0 || cf >= 15) {
      if (threadIdx.x == 0) t_cnt[ob] = 0;
      return;
    }
    c_lo = cf; c_hi = cf + 1;
  }
  int off[KD6D_MAX_SEG];                // first slot of level l in s_all
  {
    int o = 0;
#pragma unroll
    for (int l = 0; l < KD6D_MAX_SEG; ++l) { off[l] = o; o += l < L.n ? L.h[l] * L.w[l] : 0; }
  }

  // classes with at least one candidate cell
  if (threadIdx.x == 0) s_mask = 0u;
  __syncthreads();
  unsigned mine = 0u;
  for (int l = 0; l < L.n; ++l) {
    int h, w, row0; float st, sz;
    level_fields(L, l, h, w, row0, st, sz);
    const int hw = h * w;
    const f32x4_t* rows4 = reinterpret_cast<const f32x4_t*>(cls + (size_t)(row0 + b * hw) * 16);
#pragma unroll 2
    for (int cell = threadIdx.x; cell < hw; cell += kT) {      // one 64-B logit row per thread: 4 independent loads
      f32x4_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = rows4[cell * 4 + u];
#pragma unroll
      for (int c = 0; c < 15; ++c)
        if (sigmoidf_(v[c >> 2][c & 3]) > th) mine |= 1u << c;
    }
  }
  if (mine) atomicOr(&s_mask, mine);
  __syncthreads();
  const unsigned cmask = s_mask;

  float ai[4];
  inv2x2(bbox_trans + b * 6, ai);
  const float tx = bbox_trans[b * 6 + 2], ty = bbox_trans[b * 6 + 5];

  int emitted = 0;
  for (int c = c_lo; c < c_hi && emitted == 0; ++c) {
    if (!((cmask >> c) & 1u)) continue;
    // candidate scores of this class for every cell of the image, ONE pass over the logits; the arg-max and
    // top-n loops below only touch LDS (they used to re-read the logits level by level: 15 dependent global
    // round trips per image)
    __syncthreads();
    for (int l = 0; l < L.n; ++l) {
      int h, w, row0; float st, sz;
      level_fields(L, l, h, w, row0, st, sz);
      const int hw = h * w;
      int lo = 0;
#pragma unroll
      for (int q = 0; q < KD6D_MAX_SEG; ++q) if (q == l) lo = off[q];
      for (int cell = threadIdx.x; cell < hw; cell += kT) {
        const float p = sigmoidf_(cls[(size_t)(row0 + b * hw + cell) * 16 + c]);
        s_all[lo + cell] = p > th ? sqrtf(p) : -1.f;
      }
    }
    __syncthreads();
    // ---- most confident cell per level -> reference box size (postprocess_kd.py:121-141) ----
    float box_conf = 0.f, box_size = 0.f;
    for (int l = 0; l < L.n; ++l) {
      int h, w, row0; float st, sz;
      level_fields(L, l, h, w, row0, st, sz);
      const int hw = h * w;
      int lo = 0;
#pragma unroll
      for (int q = 0; q < KD6D_MAX_SEG; ++q) if (q == l) lo = off[q];
      float bv = -1.f; int bi = 0x7fffffff;
      for (int cell = threadIdx.x; cell < hw; cell += kT) {
        const float s = s_all[lo + cell];
        if (s > 0.f && (s > bv || (s == bv && cell < bi))) { bv = s; bi = cell; }
      }
      block_argmax(bv, bi, s_v, s_i);
      if (bv > 0.f && bv > box_conf) {
        box_conf = bv;
        const float* r = reg + (size_t)(row0 + b * hw + bi) * 240 + c * 16;
        const float cx = (float)(bi % w) * st + st * 0.5f, cy = (float)(bi / w) * st + st * 0.5f;
        float mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
        for (int k = 0; k < 8; ++k) {
          const float px = r[k] * sz + cx, py = r[8 + k] * sz + cy;
          mnx = fminf(mnx, px); mxx = fmaxf(mxx, px); mny = fminf(mny, py); mxy = fmaxf(mxy, py);
        }
        const float size = fmaxf(mxx - mnx, mxy - mny);
        if (size > box_size) box_size = size;
      }
    }
    // ---- cells per level: int(P * w_l / sum w + .5) over ALL anchor sizes (:143-146) ----
    if (threadIdx.x == 0) {
      float wl[KD6D_MAX_SEG], sum = 0.f;
      for (int l = 0; l < KD6D_MAX_SEG; ++l) {
        const float sz = L.size[l] > 0.f ? L.size[l] : 32.f * (float)(1 << l);
        const float dk = log2f(box_size / sz);
        wl[l] = expf(-positive_lambda * dk * dk);
        sum += wl[l];
      }
      for (int l = 0; l < KD6D_MAX_SEG; ++l) s_nk[l] = (int)(positive_num * wl[l] / sum + 0.5f);
      s_total = 0;
    }
    __syncthreads();
    // ---- top-n_l per level, score descending -------------------------------------------
    for (int l = 0; l < L.n; ++l) {
      int h, w, row0; float st, sz;
      level_fields(L, l, h, w, row0, st, sz);
      const int hw = h * w;
      const int want = s_nk[l];
      if (want <= 0) continue;
      int lo = 0;
#pragma unroll
      for (int q = 0; q < KD6D_MAX_SEG; ++q) if (q == l) lo = off[q];
      float* const s_cand = s_all + lo;       // every pick scans the level's scores in LDS and is struck out
      for (int t = 0; t < want; ++t) {
        float bv = -1.f; int bi = 0x7fffffff;
        for (int cell = threadIdx.x; cell < hw; cell += kT) {
          const float sc = s_cand[cell];
          if (sc > bv) { bv = sc; bi = cell; }      // ascending cells per thread: ties keep the smaller index
        }
        block_argmax(bv, bi, s_v, s_i);
        if (bv <= 0.f) break;          // fewer candidates than n_l
        if (threadIdx.x == 0) {
          s_cand[bi] = -1.f;
          const int slot = s_total;
          if (slot < cap) {            // decoded after the last pick, all slots in parallel
            s_pick_row[slot] = row0 + b * hw + bi;
            s_pick_cx[slot] = (float)(bi % w) * st + st * 0.5f;
            s_pick_cy[slot] = (float)(bi / w) * st + st * 0.5f;
            s_pick_sz[slot] = sz;
            s_pick_bv[slot] = bv;
          }
          s_total = slot + 1;
        }
        __syncthreads();
      }
    }
    // ---- decode the picked cells: slot x keypoint in parallel (one global round trip instead of one per pick) ----
    {
      const int n = s_total < cap ? s_total : cap;
      for (int idx = threadIdx.x; idx < n * 8; idx += kT) {
        const int slot = idx >> 3, k = idx & 7;
        const int row = s_pick_row[slot];
        const float sz = s_pick_sz[slot], bv = s_pick_bv[slot];
        const float* r = reg + (size_t)row * 240 + c * 16;
        const float px = r[k] * sz + s_pick_cx[slot] - tx, py = r[8 + k] * sz + s_pick_cy[slot] - ty;
        const size_t o = (size_t)(ob * cap + slot);
        const float fx = ai[0] * px + ai[1] * py, fy = ai[2] * px + ai[3] * py;
        t_kp[o * 16 + k * 2 + 0] = fx;
        t_kp[o * 16 + k * 2 + 1] = fy;
        t_score[o * 8 + k] = bv;
        if (t_kp_norm) {     // OT inputs (loss_libs.py:8-12 normalisation, kd_loss.py:82 weight = score^2)
          t_kp_norm[o * 16 + k * 2 + 0] = fx / frame_w;
          t_kp_norm[o * 16 + k * 2 + 1] = fy / frame_h;
          t_beta[o * 8 + k] = bv * bv;
        }
        if (k == 0 && t_row) t_row[o] = row;
      }
    }
    emitted = s_total;
    __syncthreads();
  }
  if (threadIdx.x == 0) t_cnt[ob] = emitted < cap ? emitted : cap;
}

// ------------------------------------------------------------------------------------
// SSC target assignment: one workgroup per image.
// ------------------------------------------------------------------------------------
constexpr int kMaxGt = 4;

__global__ __launch_bounds__(kT) void ssc_assign_kernel(
    Levels L, const float* __restrict__ mask, int mh, int mw, const float* __restrict__ kp3d,
    const float* __restrict__ Kmat, const int* __restrict__ class_ids, const int* __restrict__ n_gt,
    const float* __restrict__ rot, const float* __restrict__ trans,
    const float* __restrict__ bbox_trans, const float* __restrict__ keys, float positive_num,
    float positive_lambda, int cap, int* __restrict__ labels, int* __restrict__ pos_cnt,
    int* __restrict__ pos_row, int* __restrict__ pos_gt) {
  extern __shared__ float s_tab[];      // per cell of this image: mask value at the anchor centre | negated key
  __shared__ float s_v[kT / 64];
  __shared__ int s_i[kT / 64];
  __shared__ int s_has[kMaxGt];
  __shared__ float s_span[kMaxGt];
  __shared__ int s_total;
  __shared__ int s_sel_row[64];
  __shared__ int s_sel_gt[64];
  const int b = blockIdx.x;
  int G = n_gt[b];
  if (G > kMaxGt) G = kMaxGt;
  const float* m = mask + (size_t)b * mh * mw;
  int off[KD6D_MAX_SEG], ncell = 0;     // first slot of level l in the tables
#pragma unroll
  for (int l = 0; l < KD6D_MAX_SEG; ++l) { off[l] = ncell; ncell += l < L.n ? L.h[l] * L.w[l] : 0; }
  float* const s_mv = s_tab;            // mask value under the cell's anchor centre (loss.py:193-198)
  float* const s_nkey = s_tab + ncell;  // -key of the cell, later -inf once struck out / not a candidate
  // one gather pass over mask + keys; everything below reads these tables (it used to re-gather per level)
  for (int l = 0; l < L.n; ++l) {
    int h, w, row0; float st, sz;
    level_fields(L, l, h, w, row0, st, sz);
    const int hw = h * w;
    int lo = 0;
#pragma unroll
    for (int q = 0; q < KD6D_MAX_SEG; ++q) if (q == l) lo = off[q];
    for (int cell = threadIdx.x; cell < hw; cell += kT) {
      const float cx = (float)(cell % w) * st + st * 0.5f, cy = (float)(cell / w) * st + st * 0.5f;
      const int ix = (int)fminf(fmaxf(cx, 0.f), (float)(mw - 1));
      const int iy = (int)fminf(fmaxf(cy, 0.f), (float)(mh - 1));
      s_mv[lo + cell] = m[iy * mw + ix];
      s_nkey[lo + cell] = -keys[row0 + b * hw + cell];
    }
  }

  if (threadIdx.x < kMaxGt) s_has[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_total = 0;
  __syncthreads();
  // which instances are present in the mask (poses.py:269-278)
  {
    int has[kMaxGt] = {0, 0, 0, 0};
    const int n4 = ((mh * mw) & 3) == 0 && ((reinterpret_cast<uintptr_t>(m) & 15) == 0) ? (mh * mw) >> 2 : 0;
    const f32x4_t* m4 = reinterpret_cast<const f32x4_t*>(m);
    // one workgroup streams its image's whole mask (256 KB at 256x256): 16 independent 16-B loads in flight per
    // thread, or the scan is a chain of ~64 memory round trips and half of this kernel's time
    constexpr int kMlp = 16;
    int i0 = threadIdx.x;
    for (; i0 + (kMlp - 1) * kT < n4; i0 += kMlp * kT) {
      f32x4_t v[kMlp];
#pragma unroll
      for (int u = 0; u < kMlp; ++u) v[u] = m4[i0 + u * kT];
#pragma unroll
      for (int u = 0; u < kMlp; ++u)
#pragma unroll
        for (int g = 0; g < kMaxGt; ++g)
          has[g] |= (v[u][0] == (float)(g + 1)) | (v[u][1] == (float)(g + 1)) | (v[u][2] == (float)(g + 1)) |
                    (v[u][3] == (float)(g + 1));
    }
    for (int i = i0; i < n4; i += kT) {
      const f32x4_t v = m4[i];
#pragma unroll
      for (int g = 0; g < kMaxGt; ++g)
        has[g] |= (v[0] == (float)(g + 1)) | (v[1] == (float)(g + 1)) | (v[2] == (float)(g + 1)) | (v[3] == (float)(g + 1));
    }
    for (int i = n4 * 4 + threadIdx.x; i < mh * mw; i += kT) {
      const float v = m[i];
#pragma unroll
      for (int g = 0; g < kMaxGt; ++g) has[g] |= (v == (float)(g + 1));
    }
#pragma unroll
    for (int g = 0; g < kMaxGt; ++g)
      if (has[g]) atomicOr(&s_has[g], 1);
  }
  __syncthreads();
  // projected 3D-bbox span in crop coordinates (poses.py:280-300, boxlist.py:229-239)
  if (threadIdx.x < G) {
    const int g = threadIdx.x;
    float span = 1.f;  // box [0,0,0,0] -> span 1
    if (s_has[g]) {
      const int c = class_ids[b * kMaxGt + g];
      const float* X = kp3d + ((size_t)b * 15 + c) * 24;
      const float* R = rot + ((size_t)b * kMaxGt + g) * 9;
      const float* T = trans + ((size_t)b * kMaxGt + g) * 3;
      const float* Kk = Kmat + (size_t)b * 9;
      const float* bt = bbox_trans + (size_t)b * 6;
      float mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
      for (int k = 0; k < 8; ++k) {
        float cam[3];
        for (int r = 0; r < 3; ++r)
          cam[r] = R[r * 3 + 0] * X[k * 3 + 0] + R[r * 3 + 1] * X[k * 3 + 1] + R[r * 3 + 2] * X[k * 3 + 2] + T[r];
        float pr[3];
        for (int r = 0; r < 3; ++r) pr[r] = Kk[r * 3 + 0] * cam[0] + Kk[r * 3 + 1] * cam[1] + Kk[r * 3 + 2] * cam[2];
        const float u = pr[0] / (pr[2] + 1e-8f), v = pr[1] / (pr[2] + 1e-8f);
        const float x = bt[0] * u + bt[1] * v + bt[2], y = bt[3] * u + bt[4] * v + bt[5];
        mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
      }
      span = fmaxf(mxx - mnx + 1.f, mxy - mny + 1.f);
    }
    s_span[g] = span;
  }
  __syncthreads();

  // labels pass 1: -1 for every in-mask cell, 0 elsewhere (selected cells overwritten below)
  for (int l = 0; l < L.n; ++l) {
    int h, w, row0; float st, sz;
    level_fields(L, l, h, w, row0, st, sz);
    const int hw = h * w;
    int lo = 0;
#pragma unroll
    for (int q = 0; q < KD6D_MAX_SEG; ++q) if (q == l) lo = off[q];
    for (int cell = threadIdx.x; cell < hw; cell += kT) {
      const float v = s_mv[lo + cell];
      int in_any = 0;
      for (int g = 0; g < G; ++g) in_any |= (v == (float)(g + 1));
      labels[row0 + b * hw + cell] = in_any ? -1 : 0;
    }
  }
  __syncthreads();

  // per level / per gt: the n_k in-mask cells with the smallest random keys (loss.py:217-231)
  for (int l = 0; l < L.n; ++l) {
    int h, w, row0; float st, sz;
    level_fields(L, l, h, w, row0, st, sz);
    const int hw = h * w;
    for (int g = 0; g < G; ++g) {
      float sum = 0.f, wl = 0.f;
      for (int q = 0; q < L.n; ++q) {
        const float dk = fabsf(log2f(s_span[g] / L.size[q]));
        const float e = expf(-positive_lambda * dk * dk);
        sum += e;
        if (q == l) wl = e;
      }
      const int want = (int)(positive_num * wl / sum + 0.5f);
      if (want <= 0) continue;
      int lo = 0;
#pragma unroll
      for (int q = 0; q < KD6D_MAX_SEG; ++q) if (q == l) lo = off[q];
      const float gid = (float)(g + 1);
      // arg-max of the negated keys over the in-mask cells of this (level, instance) = smallest key; a picked cell
      // is struck out of the key table (a cell belongs to one instance, so this cannot hide it from another)
      for (int t = 0; t < want; ++t) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int cell = threadIdx.x; cell < hw; cell += kT) {
          const float sc = s_mv[lo + cell] == gid ? s_nkey[lo + cell] : -INFINITY;
          if (sc > bv) { bv = sc; bi = cell; }
        }
        block_argmax(bv, bi, s_v, s_i);
        if (bi == 0x7fffffff) break;   // fewer in-mask cells than n_k
        if (threadIdx.x == 0) {
          const int slot = s_total;
          if (slot < 64) { s_sel_row[slot] = row0 + b * hw + bi; s_sel_gt[slot] = g; }
          s_total = slot + 1;
          s_nkey[lo + bi] = -INFINITY;
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
  // emit positives in ascending row order (the reference's nonzero() order) and their labels
  if (threadIdx.x == 0) {
    int n = s_total < 64 ? s_total : 64;
    for (int i = 1; i < n; ++i) {
      const int r = s_sel_row[i], g = s_sel_gt[i];
      int j = i - 1;
      while (j >= 0 && s_sel_row[j] > r) { s_sel_row[j + 1] = s_sel_row[j]; s_sel_gt[j + 1] = s_sel_gt[j]; --j; }
      s_sel_row[j + 1] = r; s_sel_gt[j + 1] = g;
    }
    if (n > cap) n = cap;
    pos_cnt[b] = n;
    for (int i = 0; i < n; ++i) {
      pos_row[b * cap + i] = s_sel_row[i];
      pos_gt[b * cap + i] = s_sel_gt[i];
      labels[s_sel_row[i]] = class_ids[b * kMaxGt + s_sel_gt[i]] + 1;
    }
  }
}

// ------------------------------------------------------------------------------------
// Sigmoid focal loss (sum) and its gradient.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void focal_fwd_kernel(const float* __restrict__ cls,
                                                       const int* __restrict__ labels, int rows,
                                                       float gamma, float alpha, float* loss, long long* loss_ws) {
  __shared__ float s_part[kT / 64];
  float acc = 0.f;
  const long long total = (long long)rows * 15;
  for (long long e = (long long)blockIdx.x * kT + threadIdx.x; e < total; e += (long long)gridDim.x * kT) {
    const int row = (int)(e / 15), c = (int)(e - (long long)row * 15);
    const int t = labels[row];
    if (t < 0) continue;
    float p = sigmoidf_(cls[(size_t)row * 16 + c]);
    p = fminf(fmaxf(p, 1e-4f), 1.f - 1e-4f);
    if (t == c + 1) acc += -alpha * powf(1.f - p, gamma) * logf(p);
    else acc += -(1.f - alpha) * powf(p, gamma) * logf(1.f - p);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < kT / 64; ++w) s += s_part[w];
    kd6d_detail::det_scalar_arrive<KD6D_DET_ACT>(loss_ws, s, gridDim.x, loss);
  }
}

template <typename T>
__global__ __launch_bounds__(kT) void focal_bwd_kernel(const float* __restrict__ cls,
                                                       const int* __restrict__ labels, int rows,
                                                       float gamma, float alpha,
                                                       const float* __restrict__ weight,
                                                       T* __restrict__ dcls) {
  const float wgt = weight[0];
  const long long total = (long long)rows * 16;
  for (long long e = (long long)blockIdx.x * kT + threadIdx.x; e < total; e += (long long)gridDim.x * kT) {
    const int row = (int)(e >> 4), c = (int)(e & 15);
    const int t = labels[row];
    float g = 0.f;
    if (t >= 0 && c < 15) {
      const float p0 = sigmoidf_(cls[e]);
      if (p0 >= 1e-4f && p0 <= 1.f - 1e-4f) {
        const float p = p0, q = 1.f - p0;
        float dLdp;
        if (t == c + 1) dLdp = alpha * (gamma * powf(q, gamma - 1.f) * logf(p) - powf(q, gamma) / p);
        else dLdp = -(1.f - alpha) * (gamma * powf(p, gamma - 1.f) * logf(q) - powf(p, gamma) / q);
        g = dLdp * p * q * wgt;
      }
    }
    dcls[e] = from_f32<T>(g);
  }
}

// ------------------------------------------------------------------------------------
// Student local predictions of the positive cells + object-space loss.
// One workgroup per image, thread = (slot, keypoint).
// ------------------------------------------------------------------------------------
struct StudentArgs {
  const float* cls; const float* reg;
  const int* pos_cnt; const int* pos_row; const int* pos_gt;
  const int* class_ids; const float* kp3d; const float* rot; const float* trans;
  const float* bbox_trans; const float* diameters;
  float kinv[9];
  float frame_w, frame_h;
  int cap;
  float* xs; float* alpha; float* g_reg_xy; float* loss_reg; long long* loss_reg_ws; int* s_start;
};

__device__ __forceinline__ void locate_row(const Levels& L, int row, int& l, int& cell, int& w,
                                           float& st, float& sz) {
  l = 0; int r0 = 0, hw = 1; w = 1; st = 1.f; sz = 1.f;
#pragma unroll
  for (int s = 0; s < KD6D_MAX_SEG; ++s)
    if (s < L.n && row >= L.row0[s]) { l = s; r0 = L.row0[s]; hw = L.h[s] * L.w[s]; w = L.w[s]; st = L.stride[s]; sz = L.size[s]; }
  cell = (row - r0) % hw;
}

__global__ __launch_bounds__(kT) void student_points_kernel(Levels L, StudentArgs a) {
  __shared__ float s_part[kT / 64];
  const int b = blockIdx.x;
  const int n = a.pos_cnt[b];
  if (threadIdx.x == 0) a.s_start[b] = b * a.cap;
  float ai[4];
  inv2x2(a.bbox_trans + b * 6, ai);
  const float tx = a.bbox_trans[b * 6 + 2], ty = a.bbox_trans[b * 6 + 5];
  float acc = 0.f;
  for (int e = threadIdx.x; e < n * 8; e += kT) {
    const int slot = e >> 3, k = e & 7;
    const int row = a.pos_row[b * a.cap + slot];
    const int g = a.pos_gt[b * a.cap + slot];
    const int c = a.class_ids[b * kMaxGt + g];
    int l, cell, w; float st, sz;
    locate_row(L, row, l, cell, w, st, sz);
    const float cx = (float)(cell % w) * st + st * 0.5f, cy = (float)(cell / w) * st + st * 0.5f;
    const float* r = a.reg + (size_t)row * 240 + c * 16;
    const float px = r[k] * sz + cx - tx, py = r[8 + k] * sz + cy - ty;
    const float x = ai[0] * px + ai[1] * py, y = ai[2] * px + ai[3] * py;   // full-frame px
    const size_t o = (size_t)(b * a.cap + slot) * 8 + k;
    a.xs[o * 2 + 0] = x / a.frame_w;
    a.xs[o * 2 + 1] = y / a.frame_h;
    float p = sigmoidf_(a.cls[(size_t)row * 16 + c]);
    a.alpha[o] = fminf(fmaxf(p, 1e-3f), 1.f - 1e-3f);
    // ---- object-space loss (kd_loss.py:57-71) ----
    const float* X3 = a.kp3d + ((size_t)b * 15 + c) * 24 + k * 3;
    const float* R = a.rot + ((size_t)b * kMaxGt + g) * 9;
    const float* T = a.trans + ((size_t)b * kMaxGt + g) * 3;
    float X[3];
    for (int i = 0; i < 3; ++i) X[i] = R[i * 3] * X3[0] + R[i * 3 + 1] * X3[1] + R[i * 3 + 2] * X3[2] + T[i];
    float bb[3];
    for (int i = 0; i < 3; ++i) bb[i] = a.kinv[i * 3] * x + a.kinv[i * 3 + 1] * y + a.kinv[i * 3 + 2];
    const float bX = bb[0] * X[0] + bb[1] * X[1] + bb[2] * X[2];
    const float b2 = bb[0] * bb[0] + bb[1] * bb[1] + bb[2] * bb[2];
    const float s = bX / b2;
    const float d = a.diameters[c];
    float gp[3], lsum = 0.f;
    for (int i = 0; i < 3; ++i) {
      const float diff = 50.f * (bb[i] * s - X[i]) / d;
      const float ad = fabsf(diff);
      lsum += ad < 1.f ? 0.5f * diff * diff : ad - 0.5f;
      const float dl = ad < 1.f ? diff : (diff > 0.f ? 1.f : -1.f);
      gp[i] = dl / (24.f * d);     // d(loss_cell)/d(proj_i): (50/d) * (1/24) * (1/50)
    }
    acc += lsum / (24.f * 50.f);
    // proj = b * s ; s = (b.X)/(b.b)
    const float gpb = gp[0] * bb[0] + gp[1] * bb[1] + gp[2] * bb[2];
    float gb[3];
    for (int i = 0; i < 3; ++i) gb[i] = s * gp[i] + gpb * (X[i] / b2 - 2.f * bX * bb[i] / (b2 * b2));
    a.g_reg_xy[o * 2 + 0] = gb[0] * a.kinv[0] + gb[1] * a.kinv[3] + gb[2] * a.kinv[6];
    a.g_reg_xy[o * 2 + 1] = gb[0] * a.kinv[1] + gb[1] * a.kinv[4] + gb[2] * a.kinv[7];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < kT / 64; ++w) s += s_part[w];
    kd6d_detail::det_scalar_arrive<KD6D_DET_ACT>(a.loss_reg_ws, s, gridDim.x, a.loss_reg);
  }
}

// loss_kd = mean over valid images of loss_img (kd_loss.py:99-103)
__global__ void kd_mean_kernel(const float* __restrict__ loss_img, const int* __restrict__ valid_img,
                               int n, float* loss_kd, int* n_valid) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f; int v = 0;
    for (int i = 0; i < n; ++i)
      if (valid_img[i] > 0) { s += loss_img[i]; ++v; }
    *loss_kd = v > 0 ? s / (float)v : 0.f;
    *n_valid = v;
  }
}

// ------------------------------------------------------------------------------------
// Chain d(loss)/d(points, alpha) into the logit-gradient tensors of the positive rows.
// ------------------------------------------------------------------------------------
struct BackwardArgs {
  const float* cls; const float* reg;
  const int* pos_cnt; const int* pos_row; const int* pos_gt; const int* class_ids;
  const float* bbox_trans;
  const float* g_reg_xy; const float* g_kd_xs; const float* g_kd_alpha;
  const int* n_valid; const int* valid_img;
  const float* weights;      // {w_cls, w_reg, w_kd} upstream gradients of the three losses
  const float* seg_scale;    // PoseHead.scales (per level), may be null
  long long* dseg_scale;     // gradient of the scales: planar accumulators of the gradient bucket (kd6d.h), may be null
  long long acc_hi;
  float frame_w, frame_h;
  int cap; int detach_alpha;
  void* dcls; void* dreg;
};

template <typename T>
__global__ __launch_bounds__(kT) void loss_backward_kernel(Levels L, BackwardArgs a) {
  const int b = blockIdx.x;
  const int n = a.pos_cnt[b];
  const float w_reg = a.weights[1];
  const int nv = a.n_valid[0];
  const float w_kd = (nv > 0 && a.valid_img[b] > 0) ? a.weights[2] / (float)nv : 0.f;
  float ai[4];
  inv2x2(a.bbox_trans + b * 6, ai);
  T* dcls = reinterpret_cast<T*>(a.dcls);
  T* dreg = reinterpret_cast<T*>(a.dreg);
  for (int e = threadIdx.x; e < n * 8; e += kT) {
    const int slot = e >> 3, k = e & 7;
    const int row = a.pos_row[b * a.cap + slot];
    const int g = a.pos_gt[b * a.cap + slot];
    const int c = a.class_ids[b * kMaxGt + g];
    int l, cell, w; float st, sz;
    locate_row(L, row, l, cell, w, st, sz);
    const size_t o = (size_t)(b * a.cap + slot) * 8 + k;
    const float gx = w_reg * a.g_reg_xy[o * 2 + 0] + w_kd * a.g_kd_xs[o * 2 + 0] / a.frame_w;
    const float gy = w_reg * a.g_reg_xy[o * 2 + 1] + w_kd * a.g_kd_xs[o * 2 + 1] / a.frame_h;
    // [x;y] = Ainv ([px;py] - t)  =>  d/dp = Ainv^T g ; p = pred * size + centre
    float dpx = (ai[0] * gx + ai[2] * gy) * sz;
    float dpy = (ai[1] * gx + ai[3] * gy) * sz;
    float sc = 1.f;
    if (a.seg_scale) {
#pragma unroll
      for (int s = 0; s < KD6D_MAX_SEG; ++s)
        if (s == l) sc = a.seg_scale[s];
      if (a.dseg_scale) {
        const float* r = a.reg + (size_t)row * 240 + c * 16;
        // d/dscale = sum grad * raw = sum grad * out / scale
        kd6d_detail::det_add_planar<KD6D_DET_GRAD>(a.dseg_scale + l, a.acc_hi, (dpx * r[k] + dpy * r[8 + k]) / sc);
      }
    }
    dreg[(size_t)row * 240 + c * 16 + k] = from_f32<T>(dpx * sc);
    dreg[(size_t)row * 240 + c * 16 + 8 + k] = from_f32<T>(dpy * sc);
  }
  if (!a.detach_alpha) {
    for (int slot = threadIdx.x; slot < n; slot += kT) {
      const int row = a.pos_row[b * a.cap + slot];
      const int g = a.pos_gt[b * a.cap + slot];
      const int c = a.class_ids[b * kMaxGt + g];
      float ga = 0.f;
      for (int k = 0; k < 8; ++k) ga += a.g_kd_alpha[(size_t)(b * a.cap + slot) * 8 + k];
      const float p = sigmoidf_(a.cls[(size_t)row * 16 + c]);
      if (p >= 1e-3f && p <= 1.f - 1e-3f) {
        const size_t o = (size_t)row * 16 + c;
        dcls[o] = from_f32<T>(to_f32<T>(dcls[o]) + w_kd * ga * p * (1.f - p));
      }
    }
  }
}

bool fill_levels(const kd6d_levels* lv, Levels* L) {
  if (!lv || lv->n < 1 || lv->n > KD6D_MAX_SEG || lv->batch < 1) return false;
  memset(L, 0, sizeof(*L));
  L->n = lv->n; L->batch = lv->batch;
  int row = 0;
  for (int s = 0; s < KD6D_MAX_SEG; ++s) {
    L->size[s] = lv->anchor_size[s];
    L->stride[s] = lv->anchor_stride[s];
    if (s < lv->n) {
      if (lv->h[s] <= 0 || lv->w[s] <= 0 || lv->anchor_size[s] <= 0.f || lv->anchor_stride[s] <= 0.f) return false;
      L->h[s] = lv->h[s]; L->w[s] = lv->w[s]; L->row0[s] = row;
      row += lv->batch * lv->h[s] * lv->w[s];
    }
  }
  return true;
}

}  // namespace

static size_t total_cells(const Levels& L) {
  size_t t = 0;
  for (int l = 0; l < L.n; ++l) t += (size_t)L.h[l] * (size_t)L.w[l];
  return t > 0 ? t : 1;
}

extern "C" int kd6d_teacher_select(const kd6d_levels* levels, const float* cls, const float* reg,
                                   const float* bbox_trans, float threshold, float positive_num,
                                   float positive_lambda, int cap, float frame_w, float frame_h,
                                   int32_t* t_cnt, float* t_kp, float* t_score, int32_t* t_row,
                                   float* t_kp_norm, float* t_beta, void* stream) {
  Levels L;
  KD6D_CHECK_ARG(fill_levels(levels, &L), "kd6d_teacher_select: bad level table");
  KD6D_CHECK_ARG(cls && reg && bbox_trans && t_cnt && t_kp && t_score && t_row && t_kp_norm && t_beta && cap > 0 &&
                     frame_w > 0.f && frame_h > 0.f,
                 "kd6d_teacher_select: bad arguments");
  KD6D_CHECK_ARG(threshold > 0.f && threshold < 1.f, "kd6d_teacher_select: threshold must be in (0,1)");
  KD6D_CHECK_ARG(cap <= 64, "kd6d_teacher_select: cap=%d exceeds 64", cap);
  KD6D_CHECK_ARG(total_cells(L) * sizeof(float) <= 60 * 1024, "kd6d_teacher_select: %zu cells per image exceed the LDS table",
                 total_cells(L));
  hipLaunchKernelGGL(teacher_select_kernel, dim3(L.batch), dim3(kT), total_cells(L) * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream),
                     cls, reg, L, bbox_trans, threshold, positive_num, positive_lambda, cap, frame_w, frame_h,
                     t_cnt, t_kp, t_score, t_row, t_kp_norm, t_beta, (const int*)nullptr, (const int*)nullptr);
  KD6D_CHECK_LAUNCH("kd6d_teacher_select");
  return KD6D_OK;
}

extern "C" int kd6d_pose_candidates(const kd6d_levels* levels, const float* cls, const float* reg,
                                    const float* bbox_trans, const int32_t* class_ids, const int32_t* n_gt,
                                    float threshold, float positive_num, float positive_lambda, int cap,
                                    int32_t* cnt, float* kp, float* score, void* stream) {
  Levels L;
  KD6D_CHECK_ARG(fill_levels(levels, &L), "kd6d_pose_candidates: bad level table");
  KD6D_CHECK_ARG(cls && reg && bbox_trans && class_ids && n_gt && cnt && kp && score && cap > 0 && cap <= 64,
                 "kd6d_pose_candidates: bad arguments (cap must be in 1..64)");
  KD6D_CHECK_ARG(threshold > 0.f && threshold < 1.f, "kd6d_pose_candidates: threshold must be in (0,1)");
  KD6D_CHECK_ARG(total_cells(L) * sizeof(float) <= 60 * 1024, "kd6d_pose_candidates: %zu cells per image exceed the LDS table",
                 total_cells(L));
  hipLaunchKernelGGL(teacher_select_kernel, dim3(L.batch, KD6D_MAX_GT), dim3(kT), total_cells(L) * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream),
                     cls, reg, L, bbox_trans, threshold, positive_num, positive_lambda, cap, 1.f, 1.f,
                     cnt, kp, score, (int*)nullptr, (float*)nullptr, (float*)nullptr, class_ids, n_gt);
  KD6D_CHECK_LAUNCH("kd6d_pose_candidates");
  return KD6D_OK;
}

extern "C" int kd6d_ssc_assign(const kd6d_levels* levels, const float* mask, int mask_h, int mask_w,
                               const float* kp3d, const float* K, const int32_t* class_ids,
                               const int32_t* n_gt, const float* rot, const float* trans,
                               const float* bbox_trans, const float* keys, float positive_num,
                               float positive_lambda, int cap, int32_t* labels, int32_t* pos_cnt,
                               int32_t* pos_row, int32_t* pos_gt, void* stream) {
  Levels L;
  KD6D_CHECK_ARG(fill_levels(levels, &L), "kd6d_ssc_assign: bad level table");
  KD6D_CHECK_ARG(mask && kp3d && K && class_ids && n_gt && rot && trans && bbox_trans && keys && labels &&
                     pos_cnt && pos_row && pos_gt && cap > 0 && cap <= 64 && mask_h > 0 && mask_w > 0,
                 "kd6d_ssc_assign: bad arguments (cap must be in 1..64)");
  KD6D_CHECK_ARG(total_cells(L) * 2 * sizeof(float) <= 60 * 1024, "kd6d_ssc_assign: %zu cells per image exceed the LDS tables",
                 total_cells(L));
  hipLaunchKernelGGL(ssc_assign_kernel, dim3(L.batch), dim3(kT), total_cells(L) * 2 * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), L,
                     mask, mask_h, mask_w, kp3d, K, class_ids, n_gt, rot, trans, bbox_trans, keys,
                     positive_num, positive_lambda, cap, labels, pos_cnt, pos_row, pos_gt);
  KD6D_CHECK_LAUNCH("kd6d_ssc_assign");
  return KD6D_OK;
}

extern "C" int kd6d_focal_fwd(const float* cls, const int32_t* labels, int rows, float gamma, float alpha,
                              float* loss, kd6d_scalar_ws* loss_ws, void* stream) {
  KD6D_CHECK_ARG(cls && labels && loss && loss_ws && rows > 0, "kd6d_focal_fwd: bad arguments");
  int nb = (rows * 15 + kT - 1) / kT;
  if (nb > 128) nb = 128;        // one same-address atomic per workgroup, retired serially: 1024 of them were the kernel
  hipLaunchKernelGGL(focal_fwd_kernel, dim3(nb), dim3(kT), 0, reinterpret_cast<hipStream_t>(stream), cls, labels,
                     rows, gamma, alpha, loss, reinterpret_cast<long long*>(loss_ws));
  KD6D_CHECK_LAUNCH("kd6d_focal_fwd");
  return KD6D_OK;
}

extern "C" int kd6d_focal_bwd(int dtype, const float* cls, const int32_t* labels, int rows, float gamma,
                              float alpha, const float* weight, void* dcls, void* stream) {
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "kd6d_focal_bwd: bad dtype");
  KD6D_CHECK_ARG(cls && labels && weight && dcls && rows > 0, "kd6d_focal_bwd: bad arguments");
  int nb = (rows * 16 + kT - 1) / kT;
  if (nb > 2048) nb = 2048;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == KD6D_BF16)
    hipLaunchKernelGGL(focal_bwd_kernel<bf16_t>, dim3(nb), dim3(kT), 0, st, cls, labels, rows, gamma, alpha,
                       weight, reinterpret_cast<bf16_t*>(dcls));
  else
    hipLaunchKernelGGL(focal_bwd_kernel<float>, dim3(nb), dim3(kT), 0, st, cls, labels, rows, gamma, alpha,
                       weight, reinterpret_cast<float*>(dcls));
  KD6D_CHECK_LAUNCH("kd6d_focal_bwd");
  return KD6D_OK;
}

extern "C" int kd6d_student_points(const kd6d_levels* levels, const float* cls, const float* reg,
                                   const int32_t* pos_cnt, const int32_t* pos_row, const int32_t* pos_gt,
                                   const int32_t* class_ids, const float* kp3d, const float* rot,
                                   const float* trans, const float* bbox_trans, const float* diameters,
                                   const float* kinv_host, float frame_w, float frame_h, int cap, float* xs,
                                   float* alpha, float* g_reg_xy, float* loss_reg, kd6d_scalar_ws* loss_reg_ws,
                                   int32_t* s_start, void* stream) {
  Levels L;
  KD6D_CHECK_ARG(fill_levels(levels, &L), "kd6d_student_points: bad level table");
  KD6D_CHECK_ARG(cls && reg && pos_cnt && pos_row && pos_gt && class_ids && kp3d && rot && trans &&
                     bbox_trans && diameters && kinv_host && xs && alpha && g_reg_xy && loss_reg && loss_reg_ws && s_start,
                 "kd6d_student_points: null pointer");
  StudentArgs a;
  a.cls = cls; a.reg = reg; a.pos_cnt = pos_cnt; a.pos_row = pos_row; a.pos_gt = pos_gt;
  a.class_ids = class_ids; a.kp3d = kp3d; a.rot = rot; a.trans = trans; a.bbox_trans = bbox_trans;
  a.diameters = diameters;
  for (int i = 0; i < 9; ++i) a.kinv[i] = kinv_host[i];
  a.frame_w = frame_w; a.frame_h = frame_h; a.cap = cap;
  a.xs = xs; a.alpha = alpha; a.g_reg_xy = g_reg_xy; a.loss_reg = loss_reg; a.loss_reg_ws = reinterpret_cast<long long*>(loss_reg_ws); a.s_start = s_start;
  hipLaunchKernelGGL(student_points_kernel, dim3(L.batch), dim3(kT), 0, reinterpret_cast<hipStream_t>(stream), L, a);
  KD6D_CHECK_LAUNCH("kd6d_student_points");
  return KD6D_OK;
}

extern "C" int kd6d_kd_mean(const float* loss_img, const int32_t* valid_img, int n_images, float* loss_kd,
                            int32_t* n_valid, void* stream) {
  KD6D_CHECK_ARG(loss_img && valid_img && loss_kd && n_valid && n_images > 0, "kd6d_kd_mean: bad arguments");
  hipLaunchKernelGGL(kd_mean_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), loss_img,
                     valid_img, n_images, loss_kd, n_valid);
  KD6D_CHECK_LAUNCH("kd6d_kd_mean");
  return KD6D_OK;
}

extern "C" int kd6d_loss_backward(const kd6d_levels* levels, int dtype, const float* cls, const float* reg,
                                  const int32_t* pos_cnt, const int32_t* pos_row, const int32_t* pos_gt,
                                  const int32_t* class_ids, const float* bbox_trans, const float* g_reg_xy,
                                  const float* g_kd_xs, const float* g_kd_alpha, const int32_t* n_valid,
                                  const int32_t* valid_img, const float* weights, const float* seg_scale,
                                  int64_t* dseg_scale_acc, int64_t acc_hi_stride, float frame_w, float frame_h, int cap,
                                  int detach_alpha, void* dcls, void* dreg, void* stream) {
  Levels L;
  KD6D_CHECK_ARG(fill_levels(levels, &L), "kd6d_loss_backward: bad level table");
  KD6D_CHECK_ARG(dtype == KD6D_BF16 || dtype == KD6D_F32, "kd6d_loss_backward: bad dtype");
  KD6D_CHECK_ARG(cls && reg && pos_cnt && pos_row && pos_gt && class_ids && bbox_trans && g_reg_xy && g_kd_xs &&
                     g_kd_alpha && n_valid && valid_img && weights && dcls && dreg,
                 "kd6d_loss_backward: null pointer");
  BackwardArgs a;
  a.cls = cls; a.reg = reg; a.pos_cnt = pos_cnt; a.pos_row = pos_row; a.pos_gt = pos_gt; a.class_ids = class_ids;
  a.bbox_trans = bbox_trans; a.g_reg_xy = g_reg_xy; a.g_kd_xs = g_kd_xs; a.g_kd_alpha = g_kd_alpha;
  a.n_valid = n_valid; a.valid_img = valid_img; a.weights = weights; a.seg_scale = seg_scale;
  KD6D_CHECK_ARG(!dseg_scale_acc || acc_hi_stride != 0, "kd6d_loss_backward: acc_hi_stride = 0 with gradient accumulators");
  a.dseg_scale = reinterpret_cast<long long*>(dseg_scale_acc); a.acc_hi = (long long)acc_hi_stride; a.frame_w = frame_w; a.frame_h = frame_h; a.cap = cap;
  a.detach_alpha = detach_alpha; a.dcls = dcls; a.dreg = dreg;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == KD6D_BF16) hipLaunchKernelGGL(loss_backward_kernel<bf16_t>, dim3(L.batch), dim3(kT), 0, st, L, a);
  else hipLaunchKernelGGL(loss_backward_kernel<float>, dim3(L.batch), dim3(kT), 0, st, L, a);
  KD6D_CHECK_LAUNCH("kd6d_loss_backward");
  return KD6D_OK;
}
