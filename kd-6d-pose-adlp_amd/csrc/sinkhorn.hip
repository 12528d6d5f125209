// Debiased (optionally unbalanced) Sinkhorn divergence between small weighted point
// sets, forward value AND gradient in one launch for the whole batch.
//
// Replaces geomloss.SamplesLoss("sinkhorn", p=2, blur, scaling, reach) as it is called
// from the reference: losses/kd_loss.py:26-30 (construction), losses/loss_libs.py:22-51
// (per-image call with batch dim 8 = the 8 keypoints) and the autograd pass that
// loss.backward() (train_kd.py:137) runs through geomloss' "last extrapolation".
// geomloss 0.2.4 is NOT vendored by the reference; the algorithm restated here is
// SURVEY.md App. B (epsilon-scaling loop, symmetrised updates, detached final
// extrapolation, debiased cost with the (rho + eps/2) unbalanced weight).
//
// Mapping: one workgroup per image, 8 waves = the 8 keypoint problems of the image
// (they share the epsilon schedule: the diameter is taken over all 8*(N+M) points).
// Lane i owns row i of every softmin; column data and potentials live in the wave's
// private LDS slice and are read as broadcasts.  ~400 tiny launches + one .item() sync
// per image in the reference become 1 launch per step.
#include "kd6d_common.h"

namespace {

constexpr int kCap = 128;     // max points per set per image in this kernel
constexpr int kWaves = 8;     // keypoints per image
constexpr float kNegLog = -100000.f;

struct WaveLds {
  float px[kCap], py[kCap], la[kCap];   // student points, log weights
  float qx[kCap], qy[kCap], lb[kCap];   // teacher points, log weights
  float ax[kCap], bx[kCap];             // potentials living on x
  float by[kCap], ay[kCap];             // potentials living on y
  float hx1[kCap], hx2[kCap];           // la + a_x/eps, la + b_x/eps
  float hy1[kCap], hy2[kCap];           // lb + b_y/eps, lb + a_y/eps
};

// logsumexp_j( h_j - 0.5*|r - c_j|^2 * inv_eps ); optionally softmax-weighted sum of (r - c_j).
// One pass (running max with rescaling) over 4 columns at a time: the three 16-byte LDS broadcasts of a
// chunk are issued together, so the ~100-cycle LDS latency is paid once per 4 columns instead of 3x per
// column -- this loop is a pure dependency chain and was the whole cost of the kernel.
template <bool GRAD>
__device__ __forceinline__ float lse_row(float rx, float ry, const float* cx, const float* cy,
                                         const float* h, int n, float inv_eps, float& gx,
                                         float& gy) {
  float m = -INFINITY, s = 0.f, sx = 0.f, sy = 0.f;
  int j = 0;
  for (; j + 4 <= n; j += 4) {
    const f32x4_t X = *reinterpret_cast<const f32x4_t*>(cx + j);
    const f32x4_t Y = *reinterpret_cast<const f32x4_t*>(cy + j);
    const f32x4_t H = *reinterpret_cast<const f32x4_t*>(h + j);
    float dx[4], dy[4], v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      dx[t] = rx - X[t]; dy[t] = ry - Y[t];
      v[t] = H[t] - 0.5f * (dx[t] * dx[t] + dy[t] * dy[t]) * inv_eps;
    }
    const float mn = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    const float sc = expf(m - mn);            // 0 on the first chunk (m = -inf), 1 when the max did not move
    s *= sc;
    if (GRAD) { sx *= sc; sy *= sc; }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float e = expf(v[t] - mn);
      s += e;
      if (GRAD) { sx += e * dx[t]; sy += e * dy[t]; }
    }
    m = mn;
  }
  for (; j < n; ++j) {
    const float dx = rx - cx[j], dy = ry - cy[j];
    const float v = h[j] - 0.5f * (dx * dx + dy * dy) * inv_eps;
    const float mn = fmaxf(m, v);
    const float sc = expf(m - mn), e = expf(v - mn);
    s = s * sc + e;
    if (GRAD) { sx = sx * sc + e * dx; sy = sy * sc + e * dy; }
    m = mn;
  }
  if (GRAD) { gx = sx / s; gy = sy / s; }
  return m + logf(s);
}

__global__ __launch_bounds__(64 * kWaves) void sinkhorn_small_kernel(
    const float* __restrict__ xs, const float* __restrict__ alpha, const int* __restrict__ s_start,
    const int* __restrict__ s_cnt, const float* __restrict__ yt, const float* __restrict__ beta,
    const int* __restrict__ t_start, const int* __restrict__ t_cnt,
    float blur, float scaling, float reach, float* __restrict__ loss_img, int* __restrict__ valid_img,
    float* __restrict__ loss_kp, float* __restrict__ gx_out, float* __restrict__ galpha_out, int slow_path) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  WaveLds* all = reinterpret_cast<WaveLds*>(smem_raw);
  float* red = reinterpret_cast<float*>(smem_raw + sizeof(WaveLds) * kWaves);  // [kWaves][5]

  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int s0 = s_start[b], N = s_cnt[b];
  const int t0 = t_start[b], M = t_cnt[b];
  if (N <= 0 || M <= 0) {          // reference: image skipped (loss_libs.py:25-28)
    if (threadIdx.x == 0) { loss_img[b] = 0.f; valid_img[b] = 0; }
    if (loss_kp && threadIdx.x < kWaves) loss_kp[b * kWaves + threadIdx.x] = 0.f;
    return;
  }
  if (N > kCap || M > kCap) {      // caller must route larger sets to the dense kernel
    if (threadIdx.x == 0) { loss_img[b] = 0.f; valid_img[b] = -1; }
    if (loss_kp && threadIdx.x < kWaves) loss_kp[b * kWaves + threadIdx.x] = 0.f;
    return;
  }
  WaveLds& L = all[wave];
  const int k = wave;  // keypoint

  float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
  for (int i = lane; i < N; i += 64) {
    const float x = xs[((size_t)(s0 + i) * 8 + k) * 2 + 0];
    const float y = xs[((size_t)(s0 + i) * 8 + k) * 2 + 1];
    const float a = alpha[(size_t)(s0 + i) * 8 + k];
    L.px[i] = x; L.py[i] = y;
    L.la[i] = a > 0.f ? logf(a) : kNegLog;
    mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
  }
  for (int j = lane; j < M; j += 64) {
    const float x = yt[((size_t)(t0 + j) * 8 + k) * 2 + 0];
    const float y = yt[((size_t)(t0 + j) * 8 + k) * 2 + 1];
    const float w = beta[(size_t)(t0 + j) * 8 + k];
    L.qx[j] = x; L.qy[j] = y;
    L.lb[j] = w > 0.f ? logf(w) : kNegLog;
    mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
  }
  mnx = -wave_max(-mnx); mny = -wave_max(-mny); mxx = wave_max(mxx); mxy = wave_max(mxy);
  if (lane == 0) {
    red[wave * 5 + 0] = mnx; red[wave * 5 + 1] = mny; red[wave * 5 + 2] = mxx; red[wave * 5 + 3] = mxy;
  }
  __syncthreads();
  for (int w = 0; w < kWaves; ++w) {
    mnx = fminf(mnx, red[w * 5 + 0]); mny = fminf(mny, red[w * 5 + 1]);
    mxx = fmaxf(mxx, red[w * 5 + 2]); mxy = fmaxf(mxy, red[w * 5 + 3]);
  }
  const float ddx = mxx - mnx, ddy = mxy - mny;
  float diam_f = sqrtf(ddx * ddx + ddy * ddy);
  diam_f = fmaxf(diam_f, 1e-12f);

  // epsilon schedule (geomloss epsilon_schedule, p = 2), evaluated in double like the
  // reference's python floats:  [d^2] + exp(arange(2 ln d, 2 ln blur, 2 ln scaling)) + [blur^2]
  const double d = (double)diam_f;
  const double e_start = 2.0 * log(d), e_stop = 2.0 * log((double)blur), e_step = 2.0 * log((double)scaling);
  int n_ar = 0;
  if (e_step < 0.0 && e_start > e_stop) n_ar = (int)ceil((e_stop - e_start) / e_step);
  if (n_ar < 0) n_ar = 0;
  if (n_ar > 4096) n_ar = 4096;
  const int n_eps = n_ar + 2;
  const bool unbalanced = reach > 0.f;
  const double rho = (double)reach * (double)reach;

  auto eps_at = [&](int idx) -> double {
    if (idx == 0) return d * d;
    if (idx <= n_ar) return exp(e_start + (double)(idx - 1) * e_step);
    return (double)blur * (double)blur;
  };
  // the fp64 schedule once per workgroup (lane i evaluates step i) instead of once per thread and step
  constexpr int kSched = 128;
  __shared__ float s_feps[kSched], s_inv[kSched], s_lam[kSched];
  for (int i = threadIdx.x; i < n_eps && i < kSched; i += blockDim.x) {
    const double e = eps_at(i);
    s_feps[i] = (float)e;
    s_inv[i] = (float)(1.0 / e);
    s_lam[i] = unbalanced ? (float)(1.0 / (1.0 + e / rho)) : 1.f;
  }
  __syncthreads();

  // ---- small sets (N, M <= 16: the KD step has ~10 cells per image): the four softmins of an update run side by
  // side on the four 16-lane rows of the wave instead of one after the other on 10 of its 64 lanes.  Same
  // lse_row, same column order, same update formulas as the general path below: identical gradients (the loss
  // value differs in the last bits, its ~20 terms are summed in another order).
  if (N <= 16 && M <= 16 && !slow_path) {
    const int grp = lane >> 4, idx = lane & 15;
    const bool on_x = grp < 2;                        // rows of x (a_x, b_x) / rows of y (b_y, a_y)
    const bool from_x = grp == 0 || grp == 3;         // softmin over the x points / over the y points
    const int n_rows = on_x ? N : M, n_cols = from_x ? N : M;
    const bool act = idx < n_rows;
    const float* cxs = from_x ? L.px : L.qx;
    const float* cys = from_x ? L.py : L.qy;
    const float* logw = on_x ? L.la : L.lb;
    float* pot = grp == 0 ? L.ax : grp == 1 ? L.bx : grp == 2 ? L.by : L.ay;          // potential this row owns
    float* hself = grp == 0 ? L.hx1 : grp == 1 ? L.hx2 : grp == 2 ? L.hy1 : L.hy2;    // log weight + own potential / eps
    const float* hcol = grp == 0 ? L.hx1 : grp == 1 ? L.hy2 : grp == 2 ? L.hy1 : L.hx2;
    const float rx = act ? (on_x ? L.px[idx] : L.qx[idx]) : 0.f;
    const float ry = act ? (on_x ? L.py[idx] : L.qy[idx]) : 0.f;
    float g0, g1;
    {
      const double eps = eps_at(0);
      const float lam = unbalanced ? (float)(1.0 / (1.0 + eps / rho)) : 1.f;
      const float feps = (float)eps, inv = (float)(1.0 / eps);
      if (act) pot[idx] = -lam * feps * lse_row<false>(rx, ry, cxs, cys, from_x ? L.la : L.lb, n_cols, inv, g0, g1);
    }
    __syncthreads();
    for (int it = 0; it < n_eps; ++it) {
      float lam, feps, inv;
      if (it < kSched) {
        lam = s_lam[it]; feps = s_feps[it]; inv = s_inv[it];
      } else {
        const double e = eps_at(it);
        lam = unbalanced ? (float)(1.0 / (1.0 + e / rho)) : 1.f;
        feps = (float)e; inv = (float)(1.0 / e);
      }
      if (act) hself[idx] = logw[idx] + pot[idx] * inv;
      __syncthreads();
      if (act) {
        const float t = -lam * feps * lse_row<false>(rx, ry, cxs, cys, hcol, n_cols, inv, g0, g1);
        pot[idx] = 0.5f * (pot[idx] + t);
      }
      __syncthreads();
    }
    const double eps = eps_at(n_eps - 1);
    const float lam = unbalanced ? (float)(1.0 / (1.0 + eps / rho)) : 1.f;
    const float feps = (float)eps, inv = (float)(1.0 / eps);
    if (act) hself[idx] = logw[idx] + pot[idx] * inv;
    __syncthreads();
    const float w_unb = (float)(rho + 0.5 * eps);
    const float inv_rho = unbalanced ? (float)(1.0 / rho) : 0.f;
    float gr0 = 0.f, gr1 = 0.f;
    float val = 0.f;
    if (act) {
      if (on_x) val = -lam * feps * lse_row<true>(rx, ry, cxs, cys, hcol, n_cols, inv, gr0, gr1);
      else val = -lam * feps * lse_row<false>(rx, ry, cxs, cys, hcol, n_cols, inv, g0, g1);
    }
    // lanes idx (a_x / b_y) and idx + 16 (b_x / a_y) hold the two halves of a row's result
    const float o_val = __shfl_xor(val, 16, 64), o_g0 = __shfl_xor(gr0, 16, 64), o_g1 = __shfl_xor(gr1, 16, 64);
    float part = 0.f;
    if (grp == 0 && act) {
      const float a_x = val, b_x = o_val, gxx0 = gr0, gxx1 = gr1, gxy0 = o_g0, gxy1 = o_g1;
      const float a = alpha[(size_t)(s0 + idx) * 8 + k];
      float dS_da, gx0, gx1;
      if (unbalanced) {
        const float ea = expf(-a_x * inv_rho), eb = expf(-b_x * inv_rho);
        dS_da = w_unb * (ea - eb);
        const float c = -a * w_unb * inv_rho * lam;
        gx0 = c * (ea * gxx0 - eb * gxy0);
        gx1 = c * (ea * gxx1 - eb * gxy1);
      } else {
        dS_da = b_x - a_x;
        gx0 = a * (gxy0 - gxx0);
        gx1 = a * (gxy1 - gxx1);
      }
      part = a * dS_da;
      gx_out[((size_t)(s0 + idx) * 8 + k) * 2 + 0] = gx0;
      gx_out[((size_t)(s0 + idx) * 8 + k) * 2 + 1] = gx1;
      galpha_out[(size_t)(s0 + idx) * 8 + k] = dS_da;
    } else if (grp == 2 && act) {
      const float b_y = val, a_y = o_val;
      const float w = beta[(size_t)(t0 + idx) * 8 + k];
      if (unbalanced) part = w * w_unb * (expf(-b_y * inv_rho) - expf(-a_y * inv_rho));
      else part = w * (a_y - b_y);
    }
    part = wave_sum(part);
    if (lane == 0) { red[wave * 5 + 4] = part; if (loss_kp) loss_kp[b * kWaves + wave] = part; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = 0.f;
      for (int w = 0; w < kWaves; ++w) tot += red[w * 5 + 4];
      loss_img[b] = tot;
      valid_img[b] = 1;
    }
    return;
  }

  // ---- initialisation at eps_0 -------------------------------------------------
  {
    const double eps = eps_at(0);
    const float lam = unbalanced ? (float)(1.0 / (1.0 + eps / rho)) : 1.f;
    const float feps = (float)eps, inv = (float)(1.0 / eps);
    float g0, g1;
    for (int i = lane; i < N; i += 64) {
      const float rx = L.px[i], ry = L.py[i];
      L.ax[i] = -lam * feps * lse_row<false>(rx, ry, L.px, L.py, L.la, N, inv, g0, g1);
      L.bx[i] = -lam * feps * lse_row<false>(rx, ry, L.qx, L.qy, L.lb, M, inv, g0, g1);
    }
    for (int j = lane; j < M; j += 64) {
      const float rx = L.qx[j], ry = L.qy[j];
      L.by[j] = -lam * feps * lse_row<false>(rx, ry, L.qx, L.qy, L.lb, M, inv, g0, g1);
      L.ay[j] = -lam * feps * lse_row<false>(rx, ry, L.px, L.py, L.la, N, inv, g0, g1);
    }
  }
  __syncthreads();

  // ---- epsilon-scaling loop (no gradient) ----------------------------------------
  for (int it = 0; it < n_eps; ++it) {
    float lam, feps, inv;
    if (it < kSched) {
      lam = s_lam[it]; feps = s_feps[it]; inv = s_inv[it];
    } else {
      const double e = eps_at(it);
      lam = unbalanced ? (float)(1.0 / (1.0 + e / rho)) : 1.f;
      feps = (float)e; inv = (float)(1.0 / e);
    }
    for (int i = lane; i < N; i += 64) {
      L.hx1[i] = L.la[i] + L.ax[i] * inv;
      L.hx2[i] = L.la[i] + L.bx[i] * inv;
    }
    for (int j = lane; j < M; j += 64) {
      L.hy1[j] = L.lb[j] + L.by[j] * inv;
      L.hy2[j] = L.lb[j] + L.ay[j] * inv;
    }
    __syncthreads();
    float g0, g1;
    for (int i = lane; i < N; i += 64) {
      const float rx = L.px[i], ry = L.py[i];
      const float at_x = -lam * feps * lse_row<false>(rx, ry, L.px, L.py, L.hx1, N, inv, g0, g1);
      const float bt_x = -lam * feps * lse_row<false>(rx, ry, L.qx, L.qy, L.hy2, M, inv, g0, g1);
      L.ax[i] = 0.5f * (L.ax[i] + at_x);
      L.bx[i] = 0.5f * (L.bx[i] + bt_x);
    }
    for (int j = lane; j < M; j += 64) {
      const float rx = L.qx[j], ry = L.qy[j];
      const float bt_y = -lam * feps * lse_row<false>(rx, ry, L.qx, L.qy, L.hy1, M, inv, g0, g1);
      const float at_y = -lam * feps * lse_row<false>(rx, ry, L.px, L.py, L.hx2, N, inv, g0, g1);
      L.by[j] = 0.5f * (L.by[j] + bt_y);
      L.ay[j] = 0.5f * (L.ay[j] + at_y);
    }
    __syncthreads();
  }

  // ---- last extrapolation (carries the gradient) + debiased cost -------------------
  const double eps = eps_at(n_eps - 1);
  const float lam = unbalanced ? (float)(1.0 / (1.0 + eps / rho)) : 1.f;
  const float feps = (float)eps, inv = (float)(1.0 / eps);
  for (int i = lane; i < N; i += 64) {
    L.hx1[i] = L.la[i] + L.ax[i] * inv;
    L.hx2[i] = L.la[i] + L.bx[i] * inv;
  }
  for (int j = lane; j < M; j += 64) {
    L.hy1[j] = L.lb[j] + L.by[j] * inv;
    L.hy2[j] = L.lb[j] + L.ay[j] * inv;
  }
  __syncthreads();
  const float w_unb = (float)(rho + 0.5 * eps);
  const float inv_rho = unbalanced ? (float)(1.0 / rho) : 0.f;
  float part = 0.f;
  for (int i = lane; i < N; i += 64) {
    const float rx = L.px[i], ry = L.py[i];
    float gxx0, gxx1, gxy0, gxy1;
    const float a_x = -lam * feps * lse_row<true>(rx, ry, L.px, L.py, L.hx1, N, inv, gxx0, gxx1);
    const float b_x = -lam * feps * lse_row<true>(rx, ry, L.qx, L.qy, L.hy2, M, inv, gxy0, gxy1);
    const float a = alpha[(size_t)(s0 + i) * 8 + k];
    float dS_da, gx0, gx1;
    if (unbalanced) {
      const float ea = expf(-a_x * inv_rho), eb = expf(-b_x * inv_rho);
      dS_da = w_unb * (ea - eb);
      const float c = -a * w_unb * inv_rho * lam;
      gx0 = c * (ea * gxx0 - eb * gxy0);
      gx1 = c * (ea * gxx1 - eb * gxy1);
    } else {
      dS_da = b_x - a_x;
      gx0 = a * (gxy0 - gxx0);
      gx1 = a * (gxy1 - gxx1);
    }
    part += a * dS_da;
    gx_out[((size_t)(s0 + i) * 8 + k) * 2 + 0] = gx0;
    gx_out[((size_t)(s0 + i) * 8 + k) * 2 + 1] = gx1;
    galpha_out[(size_t)(s0 + i) * 8 + k] = dS_da;
  }
  for (int j = lane; j < M; j += 64) {
    const float rx = L.qx[j], ry = L.qy[j];
    float g0, g1;
    const float b_y = -lam * feps * lse_row<false>(rx, ry, L.qx, L.qy, L.hy1, M, inv, g0, g1);
    const float a_y = -lam * feps * lse_row<false>(rx, ry, L.px, L.py, L.hx2, N, inv, g0, g1);
    const float w = beta[(size_t)(t0 + j) * 8 + k];
    if (unbalanced) part += w * w_unb * (expf(-b_y * inv_rho) - expf(-a_y * inv_rho));
    else part += w * (a_y - b_y);
  }
  part = wave_sum(part);
  if (lane == 0) { red[wave * 5 + 4] = part; if (loss_kp) loss_kp[b * kWaves + wave] = part; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < kWaves; ++w) tot += red[w * 5 + 4];
    loss_img[b] = tot;
    valid_img[b] = 1;
  }
}

}  // namespace

extern "C" int kd6d_sinkhorn_div_fwd_bwd(const float* xs, const float* alpha, const int32_t* s_start,
                                         const int32_t* s_cnt, const float* yt, const float* beta,
                                         const int32_t* t_start, const int32_t* t_cnt, int n_images, float p, float blur, float scaling,
                                         float reach, float* loss_img, int32_t* valid_img, float* loss_kp,
                                         float* grad_xs, float* grad_alpha, void* stream) {
  KD6D_CHECK_ARG(xs && alpha && s_start && s_cnt && yt && beta && t_start && t_cnt && loss_img && valid_img && grad_xs &&
                     grad_alpha,
                 "kd6d_sinkhorn_div_fwd_bwd: null pointer");
  KD6D_CHECK_ARG(n_images > 0, "kd6d_sinkhorn_div_fwd_bwd: n_images=%d", n_images);
  if (p != 2.0f) {
    kd6d_set_error("kd6d_sinkhorn_div_fwd_bwd: only p=2 is implemented (got %g)", (double)p);
    return KD6D_ERR_UNSUPPORTED;
  }
  KD6D_CHECK_ARG(blur > 0.f && scaling > 0.f && scaling < 1.f,
                 "kd6d_sinkhorn_div_fwd_bwd: need blur>0 and 0<scaling<1");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = sizeof(WaveLds) * kWaves + sizeof(float) * kWaves * 5;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sinkhorn_small_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  // option sinkhorn.lanes = 0: every set on the general (one softmin after the other) path (tests)
  const int slow = kd6d_opt(KD6D_OPT_SINKHORN_LANES) == 0;
  hipLaunchKernelGGL(sinkhorn_small_kernel, dim3(n_images), dim3(64 * kWaves), lds, st, xs, alpha,
                     s_start, s_cnt, yt, beta, t_start, t_cnt, blur, scaling, reach, loss_img, valid_img, loss_kp,
                     grad_xs, grad_alpha, slow);
  KD6D_CHECK_LAUNCH("kd6d_sinkhorn_div_fwd_bwd");
  return KD6D_OK;
}

extern "C" int kd6d_sinkhorn_max_points(void) { return kCap; }
