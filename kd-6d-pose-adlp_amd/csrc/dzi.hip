// Dynamic-Zoom-In front-end on the GPU: decoded uint8 BGR frame + instance mask -> the normalised
// 256x256 crop the KD step consumes (the stage right before the hot path).
//
// Replaces, per image, the reference's CPU chain
//   libs/transform.py:299-308  Normalize (BGR->RGB, /255, -mean, /std) + ToTensor
//   libs/dzi_libs.py:142-210   get_affine_transform (rot = 0) -> bbox_trans
//   libs/dzi_libs.py:55-95     cv2.warpAffine INTER_LINEAR on the float32 image, INTER_NEAREST on the mask
// with one launch for the whole batch: HBM-bound, every output pixel reads 4 x 3 bytes of the frame
// (served by L2: neighbouring outputs share their taps) and writes 3 floats + 1 mask float.
// The box jitter (aug_bbox_DZI: three random numbers per image) stays on the host.
//
// Arithmetic follows cv2.warpAffine's fixed-point scheme (imgwarp.cpp): inverse matrix in double,
// source coordinates in 1/1024 px rounded to 1/32 px (bilinear) or whole pixels (nearest), float32
// weights (1-fy)(1-fx) ... from the 32-entry table, constant-0 border, products summed in tap order.
// The normalisation is a 3x256 lookup table built by the host in float64, so every tap value is the
// float32 the reference's float64 Normalize produces.
#include <math.h>

#include "kd6d_common.h"

namespace {

constexpr int AB_BITS = 10, INTER_BITS = 5;
constexpr int AB_SCALE = 1 << AB_BITS, INTER_TAB = 1 << INTER_BITS;

struct DziImage {
  double m00, m01, m10, m11, b1, b2;   // inverse map  src = [m00 m01; m10 m11] dst + [b1 b2]
};

// get_affine_transform(center, scale, 0, out_res): three float32 point pairs, forward matrix in double.
// rot = 0 makes it a pure scale + shift; the general 3-point solve is kept so the float32 rounding of
// the points enters exactly as in the reference.
__device__ void forward_affine(float cx, float cy, float scale, int out_res, double* M) {
  float src[3][2], dst[3][2];
  src[0][0] = cx; src[0][1] = cy;
  float hs = scale * -0.5f;                 // exact (power of two)
  src[1][0] = cx + 0.f; src[1][1] = cy + hs;
  const float half = (float)out_res * 0.5f;
  dst[0][0] = half; dst[0][1] = half;
  dst[1][0] = half + 0.f; dst[1][1] = half + (float)out_res * -0.5f;
  for (int k = 0; k < 2; ++k) {
    float (*p)[2] = k == 0 ? src : dst;
    const float d0 = p[0][0] - p[1][0], d1 = p[0][1] - p[1][1];
    p[2][0] = p[1][0] + -d1;
    p[2][1] = p[1][1] + d0;
  }
  // solve [x y 1] * [a b c]^T = u for the two output coordinates (Cramer, double)
  const double x0 = src[0][0], y0 = src[0][1], x1 = src[1][0], y1 = src[1][1], x2 = src[2][0], y2 = src[2][1];
  const double det = x0 * (y1 - y2) - y0 * (x1 - x2) + (x1 * y2 - x2 * y1);
  for (int r = 0; r < 2; ++r) {
    const double u0 = dst[0][r], u1 = dst[1][r], u2 = dst[2][r];
    M[r * 3 + 0] = (u0 * (y1 - y2) - y0 * (u1 - u2) + (u1 * y2 - u2 * y1)) / det;
    M[r * 3 + 1] = (x0 * (u1 - u2) - u0 * (x1 - x2) + (x1 * u2 - x2 * u1)) / det;
    M[r * 3 + 2] = (x0 * (y1 * u2 - y2 * u1) - y0 * (x1 * u2 - x2 * u1) + u0 * (x1 * y2 - x2 * y1)) / det;
  }
}

__device__ __forceinline__ long long round_half_even(double v) { return (long long)rint(v); }

__global__ __launch_bounds__(256) void dzi_crop_kernel(
    const unsigned char* __restrict__ frames, const float* __restrict__ masks, int H, int W,
    const float* __restrict__ center_scale, const float* __restrict__ lut, int out_res,
    float* __restrict__ images, float* __restrict__ masks_out, float* __restrict__ bbox_trans,
    float* __restrict__ bbox_scale) {
#pragma clang fp contract(off)   // products are rounded before they are summed (reference: plain float expression)
  __shared__ DziImage s_im;
  __shared__ float s_lut[3 * 256];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = lut[i];
  if (threadIdx.x == 0) {
    double M[6];
    const float cx = center_scale[b * 3 + 0], cy = center_scale[b * 3 + 1], sc = center_scale[b * 3 + 2];
    forward_affine(cx, cy, sc, out_res, M);
    if (blockIdx.x == 0) {
      for (int i = 0; i < 6; ++i) bbox_trans[b * 6 + i] = (float)M[i];
      bbox_scale[b] = (float)((double)out_res / (double)sc);
    }
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    s_im.m00 = A11; s_im.m01 = -M[1] * D; s_im.m10 = -M[3] * D; s_im.m11 = A22;
    s_im.b1 = -s_im.m00 * M[2] - s_im.m01 * M[5];
    s_im.b2 = -s_im.m10 * M[2] - s_im.m11 * M[5];
  }
  __syncthreads();
  const unsigned char* fr = frames + (size_t)b * H * W * 3;
  const float* mk = masks ? masks + (size_t)b * H * W : nullptr;
  const int npix = out_res * out_res;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
    const int y = p / out_res, x = p - y * out_res;
    const long long ad = round_half_even(s_im.m00 * (double)x * AB_SCALE);
    const long long bd = round_half_even(s_im.m10 * (double)x * AB_SCALE);
    const long long X0 = round_half_even((s_im.m01 * (double)y + s_im.b1) * AB_SCALE);
    const long long Y0 = round_half_even((s_im.m11 * (double)y + s_im.b2) * AB_SCALE);
    // ---- image: bilinear, coordinates rounded to 1/32 px ----
    {
      const int rd = AB_SCALE / INTER_TAB / 2;
      const long long X = (X0 + rd + ad) >> (AB_BITS - INTER_BITS), Y = (Y0 + rd + bd) >> (AB_BITS - INTER_BITS);
      const long long sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
      const float fx = (float)(X & (INTER_TAB - 1)) / (float)INTER_TAB, fy = (float)(Y & (INTER_TAB - 1)) / (float)INTER_TAB;
      const float w[4] = {(1.f - fy) * (1.f - fx), (1.f - fy) * fx, fy * (1.f - fx), fy * fx};   // exact: multiples of 1/1024
      float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long long yy = sy + (k >> 1), xx = sx + (k & 1);
        float px[3] = {0.f, 0.f, 0.f};
        if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
          const unsigned char* q = fr + ((size_t)yy * W + (size_t)xx) * 3;
          px[0] = s_lut[0 * 256 + q[2]];      // R <- byte 2 of BGR
          px[1] = s_lut[1 * 256 + q[1]];
          px[2] = s_lut[2 * 256 + q[0]];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float pr = px[c] * w[k];
          asm volatile("" : "+v"(pr));        // the product is rounded to fp32 before it is summed: no FMA contraction
          acc[c] = acc[c] + pr;
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) images[((size_t)b * 3 + c) * npix + p] = acc[c];
    }
    // ---- mask: nearest ----
    if (mk) {
      const int rd = AB_SCALE / 2;
      const long long X = (X0 + rd + ad) >> AB_BITS, Y = (Y0 + rd + bd) >> AB_BITS;
      float v = 0.f;
      if (X >= 0 && X < W && Y >= 0 && Y < H) v = mk[(size_t)Y * W + (size_t)X];
      masks_out[(size_t)b * npix + p] = v;
    }
  }
}

}  // namespace

extern "C" int kd6d_dzi_crop(const uint8_t* frames_bgr, const float* masks, int B, int H, int W,
                             const float* center_scale, const float* lut_rgb, int out_res, float* images_nchw,
                             float* masks_out, float* bbox_trans, float* bbox_scale, void* stream) {
  KD6D_CHECK_ARG(frames_bgr && center_scale && lut_rgb && images_nchw && bbox_trans && bbox_scale,
                 "kd6d_dzi_crop: null pointer");
  KD6D_CHECK_ARG((masks == nullptr) == (masks_out == nullptr), "kd6d_dzi_crop: masks and masks_out go together");
  KD6D_CHECK_ARG(B > 0 && H > 0 && W > 0 && out_res > 0 && H <= 16384 && W <= 16384 && out_res <= 4096,
                 "kd6d_dzi_crop: bad sizes B=%d H=%d W=%d out=%d", B, H, W, out_res);
  int nb = (out_res * out_res + 255) / 256;
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(dzi_crop_kernel, dim3(nb, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), frames_bgr,
                     masks, H, W, center_scale, lut_rgb, out_res, images_nchw, masks_out, bbox_trans, bbox_scale);
  KD6D_CHECK_LAUNCH("kd6d_dzi_crop");
  return KD6D_OK;
}
