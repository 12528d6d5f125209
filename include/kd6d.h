/*
 * kd6d.h -- C ABI of libkd6d.so, the MI355X (gfx950) implementation of the
 * teacher->student KD training step of GUOShuxuan/kd-6d-pose-adlp.
 *
 * The reference has no FFI layer (it is 100 % Python/PyTorch); its boundary is
 * the Python call surface of train_kd.py:94-140 -> models/model_kd.py:55-95 ->
 * losses/kd_loss.py:111-160.  Each entry point below names the reference
 * op group (file:line) whose arithmetic it replaces.  The Python host
 * (kd-6d-pose-adlp_amd/kd6d) binds these with ctypes; INTEGRATION.md shows the
 * stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (kd6d_last_error());
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch
 *     allocations) unless the name ends in _host; nothing is allocated or
 *     freed here and no call synchronises the device;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - activations are NHWC ("rows" = pixels, channels contiguous); a tensor
 *     may hold several pyramid levels back to back ("segments");
 *   - dtype: KD6D_BF16 (bf16 storage, fp32 accumulate on MFMA 16x16x32) or
 *     KD6D_F32 (fp32 storage, exact-fp32 MFMA 16x16x4).
 */
#ifndef KD6D_H
#define KD6D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KD6D_ABI_VERSION 1

enum { KD6D_BF16 = 0, KD6D_F32 = 1 };
enum { KD6D_ACT_NONE = 0, KD6D_ACT_LEAKY = 1, KD6D_ACT_RELU = 2 };
enum {
  KD6D_OK = 0,
  KD6D_ERR_ARG = -1,
  KD6D_ERR_LAUNCH = -2,
  KD6D_ERR_UNSUPPORTED = -3
};

#define KD6D_MAX_SEG 5

/* One pyramid level of a (possibly multi-level) convolution.
 * in_*  : input feature map grid,  out_* : output grid (forward sense).
 * row0  : index of the level's first pixel row inside the packed tensor. */
typedef struct kd6d_seg {
  int32_t in_h, in_w;
  int32_t out_h, out_w;
  int32_t in_row0;
  int32_t out_row0;
} kd6d_seg;

/* Forward-sense geometry of a conv layer; the same struct drives fwd, dgrad
 * and wgrad.  cin/cout are the STORED channel counts (multiples of 8). */
typedef struct kd6d_conv_geom {
  int32_t nseg;
  int32_t batch;
  int32_t cin, cout;
  int32_t ksize, stride, pad;
  int32_t reserved;
  kd6d_seg seg[KD6D_MAX_SEG];
} kd6d_conv_geom;

const char* kd6d_last_error(void);
int kd6d_abi_version(void);

/* ---- convolution: replaces torch conv2d inside backbone/common.py:316-324
 * (ConvBlock), models/model.py:64-83,97-103 (FPN) and :438-451 (PoseHead).
 * Implicit GEMM on MFMA, weights KRSC: w[cout][ky][kx][cin].
 *   y = act((conv(x) * ch_scale[c] + ch_shift[c]) * seg_scale[level]) + residual
 * ch_scale/ch_shift/seg_scale/residual may be NULL.  out_f32 != 0 writes fp32
 * regardless of dtype. */
int kd6d_conv2d_fwd(const kd6d_conv_geom* g, int dtype, const void* x,
                    const void* w, void* y, const float* ch_scale,
                    const float* ch_shift, int act, const void* residual,
                    const float* seg_scale, int out_f32, void* stream);

/* dx (+)= conv_transpose(dy, w).  wt is the dgrad packing wt[cin][ky][kx][cout]
 * produced by kd6d_pack_dgrad_weights.  accumulate != 0 adds into dx. */
int kd6d_conv2d_dgrad(const kd6d_conv_geom* g, int dtype, const void* dy,
                      const void* wt, void* dx, int accumulate, void* stream);

/* dw[cout][ky][kx][cin] += sum_pixels dy (x) x   (fp32 atomics, dw pre-zeroed
 * or holding a running sum). */
int kd6d_conv2d_wgrad(const kd6d_conv_geom* g, int dtype, const void* x,
                      const void* dy, float* dw, void* stream);

/* wt[cin][ky][kx][cout] <- w[cout][ky][kx][cin] for n_layers layers in one
 * launch.  desc_dev: int32[n_layers*6] = {w_off, wt_off, cout, cin, ksize,
 * first_block}; offsets in elements of the w / wt base arrays. */
int kd6d_pack_dgrad_weights(int dtype, const void* w_base, void* wt_base,
                            const int32_t* desc_dev, int n_layers,
                            int total_blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KD6D_H */
