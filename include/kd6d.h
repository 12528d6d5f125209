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

#define KD6D_ABI_VERSION 8

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

/* ---- reproducible reductions (ABI v8) -------------------------------------------------------------------------
 * Every sum this library forms ACROSS workgroups -- normalisation statistics, the sums of the normalisation
 * backwards, weight / bias / scale gradients, loss values, the gradient norm -- is accumulated with 64-bit INTEGER
 * atomics on a fixed-point image of the fp32 addends (csrc/kd6d_det.h), never with floating-point atomics: integer
 * addition is associative, so two executions of the same launch sequence on the same inputs give BITWISE identical
 * results whatever order the hardware retires the atomics in (the reference's CPU step is deterministic too:
 * train_kd.py:137-140 is one single-threaded backward).
 *   kd6d_acc: one accumulator, two words; value = hi * 2^(47-E) + lo * 2^-E with E = KD6D_ACC_ACT for statistics of
 *   activations and loss values, KD6D_ACC_GRAD for everything summed in the reverse sweep.  Zero-initialised by the
 *   caller (the engine zeroes its whole statistics arena once per step), only ever added to; a non-finite addend
 *   poisons it (the value reads as NaN).  Arrays of kd6d_acc are INTERLEAVED {lo, hi} per element.
 *   Small gradient outputs (dbias of kd6d_conv2d_wgrad, dgamma / dbeta of kd6d_gn_relu_bwd, dseg_scale of
 *   kd6d_loss_backward) use the PLANAR layout instead: `int64_t* acc` + `acc_hi_stride`, word lo of element i at
 *   acc[i], word hi at acc[i + acc_hi_stride].  The engine keeps ONE such accumulator image of its flat gradient bucket
 *   (lo plane then hi plane).  Per-layer WEIGHT gradients use no atomics at all: every pixel split of the launch
 *   stores its partial dW image into a caller-owned slab (plain stores, 4-5x the byte rate of atomic adds) and the
 *   splits are added in a fixed order.  kd6d_grad_acc_resolve turns both kinds into fp32 gradients, one launch at the
 *   end of the reverse sweep.
 * kd6d_acc_read: out[i] (=, or += when accumulate != 0) value(acc[i]) for n interleaved accumulators of class
 * `kind`; clear != 0 zeroes them afterwards.  kd6d_grad_acc_resolve: desc_dev holds n_regions int64 quintuples
 * {first element, element count, first workgroup, parts, slab address}, regions in ascending workgroup order,
 * total_blocks workgroups of 1024 elements each.  parts == 0: grads[e] += value(planar accumulator of element e, class
 * KD6D_ACC_GRAD), accumulator cleared.  parts >= 1: grads[first + i] += the sum over the `parts` partial images
 * slab[part][i] of `count` floats each (what kd6d_conv2d_wgrad wrote), in a fixed association: PG =
 * kd6d_grad_acc_resolve_part_groups(parts) interleaved partial sums (parts g, g + PG, ... in order), then those in g
 * order.  Workgroups per region: ceil(count / 1024) for parts == 0, ceil(count / (1024 / PG)) otherwise. */
typedef struct kd6d_acc { int64_t lo, hi; } kd6d_acc;
/* Workspace of a launch that reduces to ONE fp32 scalar (kd6d_focal_fwd, kd6d_student_points' loss_reg):
 * 32 bytes, pre-zeroed; the launch's LAST workgroup converts the fixed-point total and WRITES the scalar (the running
 * total if the workspace is reused without zeroing -- the "+=" of earlier ABI versions); arrivals is left at 0. */
typedef struct kd6d_scalar_ws { int64_t lo, hi; uint32_t arrivals; uint32_t reserved[3]; } kd6d_scalar_ws;
#define KD6D_ACC_ACT 32
#define KD6D_ACC_GRAD 52
int kd6d_acc_read(kd6d_acc* acc, int64_t n, int kind, float* out, int accumulate, int clear, void* stream);
int kd6d_grad_acc_resolve_part_groups(int parts);
int kd6d_grad_acc_resolve(const int64_t* desc_dev, int n_regions, int total_blocks, int64_t* acc,
                          int64_t acc_hi_stride, float* grads, void* stream);

/* Pair bracket: between _begin and _end, convolutions that resolve to the 3x3 "halo patch" kernel (forward or
 * data gradient) are recorded instead of launched; _end issues two recorded launches of the same kernel variant,
 * geometry and stream as ONE launch (workgroups of both convolutions in one grid), anything else one by one, in
 * call order.  For two independent convolutions of identical shape that each fill only part of the device -- the
 * cls and the pose tower layer of the head (models/model.py:438-451).  Everything else called inside the bracket
 * launches immediately; the recorded convolutions run at _end, so nothing inside the bracket may depend on them.
 * Thread-local; not nestable. */
int kd6d_conv2d_pair_begin(void);
int kd6d_conv2d_pair_end(void);
int kd6d_conv2d_pair_pending(void);   /* launches recorded so far in the open bracket (0 outside one) */

/* ---- convolution: replaces torch conv2d inside backbone/common.py:316-324
 * (ConvBlock), models/model.py:64-83,97-103 (FPN) and :438-451 (PoseHead).
 * Implicit GEMM on MFMA, weights KRSC: w[cout][ky][kx][cin].
 *   y = act((conv(x) * ch_scale[c] + ch_shift[c]) * seg_scale[level]) + residual
 * ch_scale/ch_shift/seg_scale/residual may be NULL.  out_f32 != 0 writes fp32
 * regardless of dtype.
 * stats (optional, caller-zeroed accumulators, class KD6D_ACC_ACT): statistics of the stored y accumulated by the
 * epilogue, so that the normalisation that follows needs no separate reduction pass:
 *   stats_groups == 0: {sum[cout], sumsq[cout]}           (BatchNorm batch statistics, = kd6d_colstats)
 *   stats_groups  > 0: {sum, sumsq} per (level, image, group), the layout kd6d_gn_relu_fwd consumes.
 * workspace (optional device scratch, any contents): lets layers with few output tiles and a long K run
 * split-K (fp32 partial slabs + a finalize launch); without it they run as one pass. */
int kd6d_conv2d_fwd(const kd6d_conv_geom* g, int dtype, const void* x,
                    const void* w, void* y, const float* ch_scale,
                    const float* ch_shift, int act, const void* residual,
                    const float* seg_scale, int out_f32, kd6d_acc* stats, int stats_groups,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* Convolution (+ bias) with the normalisation and activation that follow it fused into its epilogue: one launch
 * instead of conv -> statistics -> normalise, and the fp32 pre-normalisation tensor is not re-read (eval-mode callers
 * do not even store it).  Replaces, of the reference: models/model.py:395-417,438-451 (tower Conv2d -> GroupNorm(32) ->
 * ReLU) and backbone/common.py:316-324 in train mode (Conv2d -> BatchNorm2d(batch statistics) -> LeakyReLU(0.1)).
 * The workgroups of the launch exchange their partial statistics through device-scope atomics and wait for the ones
 * they need at an in-kernel barrier (GroupNorm: the tiles of the same (level, image); BatchNorm: the whole launch);
 * every wait is bounded and counted by kd6d_barrier_timeouts().
 *   kind      KD6D_NORM_GROUP | KD6D_NORM_BATCH;  groups: GroupNorm groups (4 or 8 channels per group)
 *   y         (rows_out, cout) in `dtype`: act(norm(conv(x) + bias))
 *   raw_out   optional (rows_out, cout) fp32: conv(x) + bias, what the backward pass of the normalisation reads
 *   stats     pre-zeroed accumulators: GROUP nseg*batch*groups*2 (the layout kd6d_gn_relu_bwd consumes);
 *             BATCH KD6D_BN_FUSED_REPLICAS*2*cout (private layout)
 *   counters  pre-zeroed 32-bit words: GROUP nseg*batch*KD6D_NORM_MAX_CTILES; BATCH KD6D_BARRIER_WORDS
 *   BATCH only: momentum, running_mean / running_var (updated in place), save_mean / save_invstd (outputs for
 *   kd6d_bn_train_bwd); all optional except the save pair.
 * kd6d_conv2d_fwd_norm_fusable() tells whether this geometry takes the fused path (1) or not (0: the kernel the layer
 * would run on has no fused epilogue, or -- BatchNorm -- its workgroups cannot all be resident at once on half of the
 * device; call kd6d_conv2d_fwd + kd6d_bn_train_fwd / kd6d_gn_relu_fwd instead).  kd6d_conv2d_fwd_norm returns
 * KD6D_ERR_UNSUPPORTED in that case.  Option "conv.fuse_norm" = 0 makes every geometry report 0. */
#define KD6D_NORM_GROUP 1
#define KD6D_NORM_BATCH 2
#define KD6D_BN_FUSED_REPLICAS 8
#define KD6D_NORM_MAX_CTILES 8
typedef struct kd6d_conv_norm {
  int32_t kind;
  int32_t groups;
  int32_t act;
  float eps;
  float momentum;
  const float* gamma;
  const float* beta;
  void* y;
  kd6d_acc* stats;
  unsigned int* counters;
  float* running_mean;
  float* running_var;
  float* save_mean;
  float* save_invstd;
} kd6d_conv_norm;
int kd6d_conv2d_fwd_norm_fusable(const kd6d_conv_geom* g, int dtype, int kind, int groups);
int kd6d_conv2d_fwd_norm(const kd6d_conv_geom* g, int dtype, const void* x, const void* w, void* raw_out,
                         const float* bias, const kd6d_conv_norm* norm, void* stream);

/* The convolution of a train-mode ConvBlock (backbone/common.py:316-324: Conv2d(no bias) -> BatchNorm2d -> LeakyReLU):
 * y_raw (rows_out, cout) fp32 = conv(input, w), with the batch sums of y_raw accumulated into `stats` (pre-zeroed,
 * stats_replicas rows of {sum[cout], sumsq[cout]}; workgroup b adds to row b % stats_replicas, consumers add the rows).
 * The input is either x (rows_in, cin) in `dtype` (bn == NULL), or -- bn != NULL -- the PREVIOUS block's fp32 conv
 * output, whose BatchNorm + activation is applied while it is loaded (no separate normalise launch, no wait inside the
 * kernel): bn->sums are that block's `stats` rows, save_mean / save_invstd (for kd6d_bn_train_bwd of that block) and the
 * running statistics are written by this launch, and z_out (optional, (rows_in, cin) in `dtype`) receives the
 * activation the weight gradient of THIS convolution reads.  bn != NULL needs a 1x1 or 3x3 stride-1 'same' convolution. */
typedef struct kd6d_bn_in {
  const kd6d_acc* sums;
  int32_t replicas;
  int32_t act;
  float eps;
  float momentum;
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* save_mean;
  float* save_invstd;
} kd6d_bn_in;
int kd6d_conv2d_fwd_block(const kd6d_conv_geom* g, int dtype, const void* x, const kd6d_bn_in* bn, void* z_out,
                          const void* w, float* y_raw, kd6d_acc* stats, int stats_replicas, void* stream);

/* dx (+)= conv_transpose(dy, w).  wt is the dgrad packing wt[cin][ky][kx][cout]
 * produced by kd6d_pack_dgrad_weights.  accumulate != 0 adds into dx. */
int kd6d_conv2d_dgrad(const kd6d_conv_geom* g, int dtype, const void* dy,
                      const void* wt, void* dx, int accumulate, void* stream);

/* Weight gradient dw[cout][ky][kx][cin] = sum_pixels dy (x) x as PARTIAL IMAGES: the pixel axis is split over
 * workgroups and split s stores its partial dW, all cout*k*k*cin floats, at dw_slab + s * cout*k*k*cin (plain stores;
 * every element of every part is written).  kd6d_conv2d_wgrad_parts() tells how many parts the launch writes for a
 * geometry, dtype, bias flag and cu_budget (>= 1; a deterministic function of its arguments and the device);
 * slab_floats is checked against it.  The caller adds the parts in order (kd6d_grad_acc_resolve; "reproducible
 * reductions").  dbias_acc (optional): += sum_pixels dy, the bias gradient of the same layer, taken from the dY tiles the
 * kernel stages anyway, into PLANAR accumulators (class KD6D_ACC_GRAD, stride acc_hi_stride).
 * cu_budget: how many compute units this launch should aim to fill (0 = the whole device): a caller that keeps k
 * weight gradients in flight on k streams passes CUs/k -- same k-loop work, 1/k of the partial images. */
int kd6d_conv2d_wgrad_parts(const kd6d_conv_geom* g, int dtype, int with_bias, int cu_budget);
int kd6d_conv2d_wgrad(const kd6d_conv_geom* g, int dtype, const void* x, const void* dy, float* dw_slab,
                      int64_t slab_floats, int64_t* dbias_acc, int64_t acc_hi_stride, int cu_budget, void* stream);

/* wt[cin][ky][kx][cout] <- w[cout][ky][kx][cin] for n_layers layers in one
 * launch.  desc_dev: int32[n_layers*6] = {w_off, wt_off, cout, cin, ksize,
 * first_block}; offsets in elements of the w / wt base arrays. */
/* Grouped weight gradient (bf16): the dW (+ bias gradients) of several layers as ONE launch pair -- a work list of
 * (layer, 128- or 16-channel row block of dW, kernel row ky, 128 input channels, pixel range) items sized so that
 * `n_workgroups` workgroups run equally long, each leaving an fp32 partial tile in `slab`, and a reduction launch
 * that adds the partial tiles into dw / dbias (+=, plain stores: bitwise reproducible).  Takes layers with
 * stride 1, 1x1 or 3x3 "same" padding, cin %% 128 == 0 (kd6d_wgrad_group_supported); anything else stays on
 * kd6d_conv2d_wgrad.  kd6d_wgrad_group_plan builds the work list in HOST memory (plan_host = NULL: size query);
 * the caller copies it to the device once and replays kd6d_wgrad_group_launch with it (the plan holds the items'
 * device pointers).  info[0] = workgroups, info[1] = reduction blocks, info[2] + (info[3] << 31) = slab floats. */
typedef struct kd6d_wgrad_item {
  kd6d_conv_geom geom;
  const void* x;      /* input activations (rows_in, cin), bf16 */
  const void* dy;     /* output gradient (rows_out, cout), bf16 */
  float* dw;          /* (cout, k, k, cin) fp32, accumulated into */
  float* dbias;       /* (cout) fp32 or NULL */
} kd6d_wgrad_item;
int kd6d_wgrad_group_supported(const kd6d_conv_geom* g, int dtype);
int64_t kd6d_wgrad_group_plan(const kd6d_wgrad_item* items, int n_items, int dtype, int n_workgroups, void* plan_host,
                              int64_t plan_capacity, int32_t* info);
int kd6d_wgrad_group_launch(const void* plan_dev, int n_workgroups, int n_reduce_blocks, float* slab_dev, void* stream);

int kd6d_pack_dgrad_weights(int dtype, const void* w_base, void* wt_base,
                            const int32_t* desc_dev, int n_layers,
                            int total_blocks, void* stream);


/* Pyramid description for the loss-side kernels: level l has an h[l] x w[l] grid of cells,
 * anchor stride / size as in configs/ape.yaml:3-4; rows are packed level-major, then image. */
typedef struct kd6d_levels {
  int32_t n;
  int32_t batch;
  int32_t h[KD6D_MAX_SEG];
  int32_t w[KD6D_MAX_SEG];
  float anchor_stride[KD6D_MAX_SEG];
  float anchor_size[KD6D_MAX_SEG];   /* all 5 entries are read by teacher_select (postprocess_kd.py:143) */
} kd6d_levels;

#define KD6D_MAX_GT 4   /* instances per image handled by the assignment kernel */

int kd6d_device_cu_count(void);

/* Kernel-selection options.  The dispatch rules inside the library are measured defaults; the parity tests and the
 * per-layer benches pin one kernel family for a call through this table (per context, see kd6d_ctx below; set between
 * launches by the launching thread).  The product path sets none of them.  Names and values:
 *   conv.halo      -1 auto | 0 off | 1 256x128, 2 128x128 (4 waves), 3 128x128, 4 128x64, 5 128x32, 6 192x128, 9 64x64,
 *                  11-15 the two-workgroups-per-CU twins (maps <= 32 wide): 128x128 on 4 / 8 waves, 128x64, 64x64, 128x32
 *   conv.halo_pairing  1 | 0 keep the one-workgroup-per-CU halo tiles on maps <= 32 wide
 *   conv.halo_wide     1 | 0 maps 65 ... 80 wide (the 60 x 80 level of 480 x 640 full frames) stay off the halo-patch kernel
 *   conv.smallc    -1 auto | 0 off | 1 the resident-patch kernel also below 2^17 pixels
 *   conv.splitk    -1 auto | 0 off | tile*100 + splits (tile 1 = 128x64, 2 = 64x64)
 *   conv.tile      -1 auto | 0 register-staged kernel | 1 128x128, 2 128x64, 3 64x64 (LDS-DMA kernel)
 *   wgrad.small    -1 auto | 0 off | 1 the narrow-layer weight-gradient kernel at any size
 *   bn.onepass      1 | 0 two-launch BatchNorm backward     bn.onepass_max  largest x in 16-B granules (65536)
 *   gn.onepass      1 | 0 two-launch GroupNorm backward     sinkhorn.lanes  1 | 0 general path for every point set
 *   conv.fuse_norm  bit 0: GroupNorm, bit 1: BatchNorm geometries may take kd6d_conv2d_fwd_norm (3 | 0: fusable() = 0)
 *   sinkhorn.dense_mfma  dense OT, D = 16: 1 softmin passes on the matrix pipe (inner products at fp32 accuracy from
 *                  bf16 pieces; the gradient-carrying pass's weighted sums as a second product) while
 *                  eps >= 1.5e-4 diameter^2 | 0 never | 2 every pass (error studies) | 3 as 1, the gradient-carrying
 *                  softmins in the difference form
 *   conv.smallc_wmax  widest map the resident-patch kernel (3x3, 8-32 input channels) takes: 640 | 256 (rounds 1-2)
 *   sinkhorn.dense_screen  dense OT, D = 16, passes with eps below the matrix-pipe rule: 1 approximate exponents on the
 *                  matrix pipe screen the pairs, those within 40 (+ the error bound) of a row's running maximum are
 *                  evaluated exactly in the difference form | 0 every pair in the difference form
 *   sinkhorn.dense_rows  rows per workgroup of the dense matrix-pipe softmins (each workgroup streams all columns through
 *                  LDS): -1 = 128 from 8192 rows up, 64 below | 64 | 128
 * Unknown names return KD6D_ERR_ARG. */
/* Context: the library's mutable state -- the option table, the pair bracket of kd6d_conv2d_pair_begin/_end and the
 * counter of in-kernel barrier waits that gave up -- lives in a kd6d_ctx.  Every entry point of this header acts on the
 * CALLING THREAD'S CURRENT context: the one set with kd6d_ctx_make_current(), else the process-wide default context
 * (what a host that never creates one gets; it cannot be destroyed).  Two models in one process that must not share
 * kernel-selection state each create a context and make it current around their calls (or use the kd6d_ctx_* forms,
 * which take the context explicitly; ctx == NULL means "the current one").  Contexts are not thread-safe: one thread
 * at a time per context.  The RCCL communicator (kd6d_comm) is its own handle and is passed explicitly already. */
typedef struct kd6d_ctx kd6d_ctx;
int kd6d_ctx_create(kd6d_ctx** out);               /* options at their defaults, no bracket open, counter 0; needs a device */
int kd6d_ctx_destroy(kd6d_ctx* ctx);
int kd6d_ctx_make_current(kd6d_ctx* ctx);          /* NULL: back to the default context */
kd6d_ctx* kd6d_ctx_current(void);
int kd6d_ctx_set_option(kd6d_ctx* ctx, const char* name, long long value);
int kd6d_ctx_get_option(kd6d_ctx* ctx, const char* name, long long* value);
int kd6d_ctx_reset_options(kd6d_ctx* ctx);
int kd6d_ctx_barrier_timeouts(kd6d_ctx* ctx);      /* of launches issued under that context */
int kd6d_ctx_conv2d_pair_begin(kd6d_ctx* ctx);
int kd6d_ctx_conv2d_pair_end(kd6d_ctx* ctx);
int kd6d_ctx_conv2d_pair_pending(kd6d_ctx* ctx);

int kd6d_set_option(const char* name, long long value);
int kd6d_get_option(const char* name, long long* value);
int kd6d_reset_options(void);

/* Step prologue: zero up to KD6D_MAX_ZERO device regions (16-B aligned, sizes multiples of 4 bytes; a size of 0
 * skips the entry) and add 1 to each of n_counter (<= 256) int64 counters, in ONE launch.  Stands where the
 * reference's step has optimizer.zero_grad() (train_kd.py:104) and BatchNorm's num_batches_tracked += 1; here the
 * same launch also clears the statistics arena, the dense head gradient and the loss-side slot arrays. */
#define KD6D_MAX_ZERO 8
typedef struct kd6d_zero_list {
  int32_t n;
  int32_t pad_;
  void* ptr[KD6D_MAX_ZERO];
  int64_t bytes[KD6D_MAX_ZERO];
} kd6d_zero_list;
int kd6d_zero_regions(const kd6d_zero_list* list, long long* counter, int n_counter, void* stream);

/* n uniform [0, 1) fp32 keys for kd6d_ssc_assign (the random in-mask cells of losses/loss.py:224-228), generated on
 * the device from (seed, *counter, index): *counter is a device int64 the caller advances every step (the step
 * prologue's counters), so a captured launch draws new keys at every replay.  Not torch's generator: the sequence is
 * reproducible from the seed and the step count alone. */
int kd6d_uniform_keys(float* out, int64_t n, const long long* counter, unsigned long long seed, void* stream);

/* Timing aid: stores the device's 100-MHz wall clock into *slot when the launch executes on `stream`
 * (phase boundaries inside a replayed hipGraph; tools/step_timeline.py). */
int kd6d_mark(unsigned long long* slot, void* stream);

/* ---- normalisation / pooling (HBM-bound, 16-B granules) ------------------------------------
 * BatchNorm2d(train)+LeakyReLU of ConvBlock (backbone/common.py:316-324): batch statistics by
 * kd6d_colstats (per-channel sum / sum of squares into pre-zeroed accumulators, class KD6D_ACC_ACT), then
 * kd6d_bn_train_fwd normalises, updates running stats (momentum 0.1, unbiased var) and saves
 * mean / invstd for the backward pair.  x_f32 != 0: the pre-normalisation tensor x is fp32 while
 * activations/gradients are `dtype` (keeps (x - mean) free of bf16 cancellation error).
 * Backward pair: sum_dy / sum_dy_xhat are `replicas` (1..64) rows of C accumulators (KD6D_ACC_GRAD) each, pre-zeroed; the
 * reduction's workgroups spread their per-channel atomics over the rows (same-address atomics retire
 * serially, ~27 ns each, and this kernel needs hundreds of workgroups), the apply kernel sums the rows. */
int kd6d_colstats(int dtype, const void* x, int64_t rows, int C, kd6d_acc* sum, kd6d_acc* sumsq, void* stream);
int kd6d_bn_train_fwd(int dtype, int x_f32, const void* x, void* y, int64_t rows, int C, const kd6d_acc* sum,
                      const kd6d_acc* sumsq, const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                      int act, void* stream);
int kd6d_bn_train_bwd_reduce(int dtype, int x_f32, const void* x, const void* dz, int64_t rows, int C,
                             const float* mean, const float* invstd, const float* gamma, const float* beta,
                             int act, kd6d_acc* sum_dy, kd6d_acc* sum_dy_xhat, int replicas, void* stream);
int kd6d_bn_train_bwd_apply(int dtype, int x_f32, const void* x, const void* dz, void* dx, int64_t rows, int C,
                            const float* mean, const float* invstd, const float* gamma, const float* beta,
                            int act, const kd6d_acc* sum_dy, const kd6d_acc* sum_dy_xhat, float* dgamma,
                            float* dbeta, int replicas, void* stream);

/* BN(train) + activation + MaxPool2d(2,2) in one pass: the last ConvBlock of a darknet-tiny stage and the pool
 * behind it (backbone/darknet.py:94-97).  x: (B,H,W,C) conv output; y / dy: the POOLED tensors (B,H/2,W/2,C);
 * dx: gradient of x.  The un-pooled activation and its gradient are never stored: the backward re-derives each
 * 2x2 window from x with the forward's arithmetic and rounding and sends dy to the first maximum in row-major
 * window order (MaxPool2d's rule).  Statistics, saved mean / invstd, replicas and workspaces as in the unpooled
 * entry points above; kd6d_bn_pool_train_bwd runs the reduction and the apply pass. */
int kd6d_bn_pool_train_fwd(int dtype, int x_f32, const void* x, void* y, int B, int H, int W, int C,
                           const kd6d_acc* sum, const kd6d_acc* sumsq, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, float* save_mean,
                           float* save_invstd, int act, void* stream);
int kd6d_bn_pool_train_bwd(int dtype, int x_f32, const void* x, const void* dy, void* dx, int B, int H, int W,
                           int C, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           int act, kd6d_acc* sum_dy, kd6d_acc* sum_dy_xhat, unsigned int* counter, float* dgamma,
                           float* dbeta, int replicas, void* stream);

#define KD6D_BARRIER_WORDS 32
/* BN backward as ONE launch (also kd6d_bn_pool_train_bwd, and kd6d_gn_relu_bwd below): when `counter`
 * (KD6D_BARRIER_WORDS pre-zeroed 32-bit words per call, e.g. in the per-step zeroed arena beside sum_dy) is given and the tensor fits
 * the registers of one resident grid (512 workgroups x 4 granules per thread), every workgroup keeps its slice of x / dz in registers,
 * adds its partial sums, meets the others at an in-kernel barrier on `counter` and finishes dx -- one read of x
 * and dz instead of two, one launch instead of two.  By default only tensors of up to 65536 granules take it
 * (<= 128 workgroups: there the barrier is cheaper than a launch; beyond, exchanging the sums costs more than the
 * second read -- option bn.onepass_max = <granules> moves the limit).  counter == NULL, a larger tensor or
 * option bn.onepass = 0: the reduce + apply pair above.  kd6d_barrier_timeouts(): number of barrier waits that gave up (must stay 0). */
int kd6d_bn_train_bwd(int dtype, int x_f32, const void* x, const void* dz, void* dx, int64_t rows, int C,
                      const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                      kd6d_acc* sum_dy, kd6d_acc* sum_dy_xhat, unsigned int* counter, float* dgamma, float* dbeta,
                      int replicas, void* stream);
int kd6d_barrier_timeouts(void);

/* GroupNorm(groups)+ReLU of the PoseHead towers (models/model.py:395-417) over a multi-level
 * tensor; level_hw_host[l] = H*W of level l (HOST array).  stats: 2 accumulators (KD6D_ACC_ACT) per
 * (level, image, group) = RAW sums {sum x, sum x^2} (mean/rstd are derived by the consumers, which is
 * why the backward takes eps too); gsum_ws (backward): 2*nseg*batch*groups accumulators (KD6D_ACC_GRAD) followed by
 * nseg*batch 32-bit barrier counters -- the backward is ONE launch whose workgroups of a (level, image) meet at
 * an in-kernel barrier (see kd6d_bn_train_bwd; option gn.onepass = 0: the reduce + apply pair).
 * flags: bit 0 (KD6D_GN_STATS_READY) -- stats were already accumulated by kd6d_conv2d_fwd(..., stats,
 * groups): skip the reduction pass; bit 1 (KD6D_GN_WS_ZEROED) -- the caller zeroed stats (fwd, when not
 * ready) / gsum_ws (bwd) itself (e.g. one memset of a whole scratch arena per step): skip the memset node.
 * dgamma / dbeta (backward, optional): PLANAR gradient accumulators with stride acc_hi_stride (see "reproducible
 * reductions").  Requires C/groups >= granule/2 (a 16-B granule spans <= 2 groups). */
#define KD6D_GN_STATS_READY 1
#define KD6D_GN_WS_ZEROED 2
int kd6d_gn_relu_fwd(int dtype, int x_f32, const void* x, void* y, const int32_t* level_hw_host, int nseg, int batch,
                     int C, int groups, const float* gamma, const float* beta, float eps, kd6d_acc* stats,
                     int flags, void* stream);
int kd6d_gn_relu_bwd(int dtype, int x_f32, const void* x, const void* dz, void* dx, const int32_t* level_hw_host,
                     int nseg, int batch, int C, int groups, const float* gamma, const float* beta, float eps,
                     const kd6d_acc* stats, kd6d_acc* gsum_ws, int64_t* dgamma_acc, int64_t* dbeta_acc,
                     int64_t acc_hi_stride, int flags, void* stream);
/* Two GroupNorm+ReLU backwards of identical geometry (the cls and the pose tower layer of PoseHead,
 * models/model.py:438-451) as ONE launch of the in-kernel-barrier form: alone each is 170 four-wave workgroups on
 * 256 CUs.  Same arguments as kd6d_gn_relu_bwd, the per-tensor ones in an item each; falls back to two launches
 * where the one-launch form does not apply (gn.onepass = 0, too few resident workgroups). */
typedef struct kd6d_gn_item {
  const void* x;
  const void* dz;
  void* dx;
  const float* gamma;
  const float* beta;
  const kd6d_acc* stats;
  kd6d_acc* gsum_ws;
  int64_t* dgamma;      /* planar gradient accumulators (acc_hi_stride of the call) */
  int64_t* dbeta;
} kd6d_gn_item;
int kd6d_gn_relu_bwd_pair(int dtype, int x_f32, const kd6d_gn_item* a, const kd6d_gn_item* b,
                          const int32_t* level_hw_host, int nseg, int batch, int C, int groups, float eps,
                          int64_t acc_hi_stride, int flags, void* stream);

/* MaxPool2d(2,2) (backbone/darknet.py:94-97), nearest-x2 upsample + add (models/model.py:75-78)
 * and its adjoint, ReLU / ReLU-backward / add (mode 0/1/2), NCHW fp32 image -> padded NHWC. */
int kd6d_maxpool2_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, void* stream);
int kd6d_maxpool2_bwd(int dtype, const void* x, const void* dy, void* dx, int B, int H, int W, int C,
                      int accumulate, void* stream);
int kd6d_upsample2_add(int dtype, const void* fine, const void* coarse, void* out, int B, int H, int W,
                       int C, void* stream);
int kd6d_sumpool2(int dtype, const void* dfine, void* dcoarse, int B, int H, int W, int C, int accumulate,
                  void* stream);
int kd6d_eltwise(int dtype, int mode, const void* x, const void* dy, void* y, int64_t n_elems, void* stream);
int kd6d_image_to_nhwc(int dtype, const float* img_nchw, void* out, int B, int Cimg, int H, int W, int Cpad,
                       void* stream);

/* ---- optimal-transport KD loss: replaces geomloss.SamplesLoss("sinkhorn", p=2, blur, scaling,
 * reach) as called from losses/loss_libs.py:22-51 (one call per image, batch dim = 8 keypoints)
 * and its autograd.  Image b uses student points xs[(s_start[b]+i)*8+k][2] (i < s_cnt[b]) with
 * weights alpha[(..)*8+k] and teacher points/weights likewise; reach <= 0 means balanced.
 * Outputs: loss_img[b] = sum_k S_k (0 when a set is empty: valid_img[b] = 0; -1 = set larger than
 * kd6d_sinkhorn_max_points()), loss_kp (optional, n_images*8) = the eight S_k themselves (what SamplesLoss returns
 * for a batch of 8 problems), grad_xs / grad_alpha = d loss_img / d xs, alpha. */
int kd6d_sinkhorn_div_fwd_bwd(const float* xs, const float* alpha, const int32_t* s_start,
                              const int32_t* s_cnt, const float* yt, const float* beta,
                              const int32_t* t_start, const int32_t* t_cnt, int n_images, float p, float blur,
                              float scaling, float reach, float* loss_img, int32_t* valid_img, float* loss_kp,
                              float* grad_xs, float* grad_alpha, void* stream);
int kd6d_sinkhorn_max_points(void);

/* ---- loss-side kernels (cls logits (rows,16) fp32 [15 classes + pad], reg logits (rows,240)) --
 * kd6d_teacher_select  <- postprocess/postprocess_kd.py:22-203 (PnP gate treated as true):
 *   slot layout: image b owns slots [b*cap, b*cap + t_cnt[b]); t_kp (slots,8,2) full-frame px,
 *   t_score (slots,8) = sqrt(sigmoid), t_row = packed row of the chosen cell; t_kp_norm =
 *   t_kp / (frame_w, frame_h) and t_beta = t_score^2 are the OT inputs (loss_libs.py:8-12,
 *   kd_loss.py:82).
 * kd6d_ssc_assign      <- losses/loss.py:164-268; per-image inputs are padded to KD6D_MAX_GT
 *   instances; keys (rows) are caller-supplied uniform randoms (the n smallest in-mask keys per
 *   level are the reference's randperm(...)[:n]).  labels (rows): -1 ignore, 0 bg, c+1.
 * kd6d_focal_fwd/bwd   <- losses/loss.py:20-40 (sum reduction into *loss through loss_ws, see kd6d_scalar_ws; bwd
 *   writes ALL of dcls).
 * kd6d_student_points  <- losses/kd_loss.py:40-71,152: decoded full-frame keypoints of the
 *   positive cells (normalised by frame_w/h into xs), OT weights alpha, loss_reg (through loss_reg_ws) and its
 *   gradient w.r.t. the full-frame points.
 * kd6d_kd_mean         <- kd_loss.py:99-103.
 * kd6d_loss_backward   <- autograd of the above into dcls (+=) / dreg (positive rows only); dseg_scale_acc (optional):
 *   the gradient of PoseHead.scales, PLANAR gradient accumulators (one per level) with stride acc_hi_stride. */
int kd6d_teacher_select(const kd6d_levels* levels, const float* cls, const float* reg,
                        const float* bbox_trans, float threshold, float positive_num, float positive_lambda,
                        int cap, float frame_w, float frame_h, int32_t* t_cnt, float* t_kp, float* t_score,
                        int32_t* t_row, float* t_kp_norm, float* t_beta, void* stream);

/* Pose candidates of the evaluation path (postprocess/postprocess.py:22-121, up to the PnP solver): the
 * same per-level top-n rule as kd6d_teacher_select, run for EVERY ground-truth slot g < n_gt[b] of image b on
 * the class class_ids[b*KD6D_MAX_GT + g] (the reference keeps only labels present in target.class_ids).
 * Output block o = b*KD6D_MAX_GT + g: cnt[o] cells, kp[(o*cap + i)*16 + k*2 + {0,1}] full-frame pixels of
 * keypoint k, score[(o*cap + i)*8 + k] = sqrt(sigmoid).  The solver (EPnP-RANSAC) runs on the host. */
int kd6d_pose_candidates(const kd6d_levels* levels, const float* cls, const float* reg,
                         const float* bbox_trans, const int32_t* class_ids, const int32_t* n_gt,
                         float threshold, float positive_num, float positive_lambda, int cap,
                         int32_t* cnt, float* kp, float* score, void* stream);
int kd6d_ssc_assign(const kd6d_levels* levels, const float* mask, int mask_h, int mask_w, const float* kp3d,
                    const float* K, const int32_t* class_ids, const int32_t* n_gt, const float* rot,
                    const float* trans, const float* bbox_trans, const float* keys, float positive_num,
                    float positive_lambda, int cap, int32_t* labels, int32_t* pos_cnt, int32_t* pos_row,
                    int32_t* pos_gt, void* stream);
int kd6d_focal_fwd(const float* cls, const int32_t* labels, int rows, float gamma, float alpha, float* loss,
                   kd6d_scalar_ws* loss_ws, void* stream);
int kd6d_focal_bwd(int dtype, const float* cls, const int32_t* labels, int rows, float gamma, float alpha,
                   const float* weight, void* dcls, void* stream);
int kd6d_student_points(const kd6d_levels* levels, const float* cls, const float* reg, const int32_t* pos_cnt,
                        const int32_t* pos_row, const int32_t* pos_gt, const int32_t* class_ids,
                        const float* kp3d, const float* rot, const float* trans, const float* bbox_trans,
                        const float* diameters, const float* kinv_host, float frame_w, float frame_h, int cap,
                        float* xs, float* alpha, float* g_reg_xy, float* loss_reg, kd6d_scalar_ws* loss_reg_ws,
                        int32_t* s_start, void* stream);
int kd6d_kd_mean(const float* loss_img, const int32_t* valid_img, int n_images, float* loss_kd,
                 int32_t* n_valid, void* stream);
int kd6d_loss_backward(const kd6d_levels* levels, int dtype, const float* cls, const float* reg,
                       const int32_t* pos_cnt, const int32_t* pos_row, const int32_t* pos_gt,
                       const int32_t* class_ids, const float* bbox_trans, const float* g_reg_xy,
                       const float* g_kd_xs, const float* g_kd_alpha, const int32_t* n_valid,
                       const int32_t* valid_img, const float* weights, const float* seg_scale,
                       int64_t* dseg_scale_acc, int64_t acc_hi_stride, float frame_w, float frame_h, int cap,
                       int detach_alpha, void* dcls, void* dreg, void* stream);

/* ---- dense optimal transport (BASELINE config 5: D-dimensional local predictions over a whole cell grid,
 * thousands of points per set; the reference would need geomloss' KeOps backend above 5000^2 pairs).  Same
 * divergence and gradients as kd6d_sinkhorn_div_fwd_bwd for ONE problem: x (N,D) with weights alpha (N)
 * against y (M,D) with beta (M), D in {2,4,8,16}; cost matrices are never materialised (online logsumexp over
 * LDS-staged column tiles).  `diameter` > 0 is the box diagonal of all points (geomloss' diameter= argument);
 * kd6d_sinkhorn_dense_diameter computes it on the device (scratch64: 64 floats).  workspace:
 * kd6d_sinkhorn_dense_workspace_floats(N, M, D) floats, any contents.  Outputs: loss (1), grad_x (N,D),
 * grad_alpha (N). */
int64_t kd6d_sinkhorn_dense_workspace_floats(int N, int M, int D);
int kd6d_sinkhorn_dense_diameter(const float* x, const float* y, int N, int M, int D, float* scratch64,
                                 float* diam_out, void* stream);
int kd6d_sinkhorn_dense_fwd_bwd(const float* x, const float* alpha, const float* y, const float* beta, int N, int M,
                                int D, float p, float blur, float scaling, float reach, double diameter,
                                float* workspace, int64_t workspace_floats, float* loss, float* grad_x,
                                float* grad_alpha, void* stream);

/* ---- Dynamic-Zoom-In front-end (the stage before the hot path): replaces
 * transform.py:299-308 (Normalize, ToTensor) + dzi_libs.py:55-95,142-210 (affine from the jittered box,
 * cv2.warpAffine bilinear on the image / nearest on the mask) for a whole batch.
 * frames_bgr (B,H,W,3) uint8 as decoded; masks (B,H,W) float or NULL; center_scale (B,3) = {cx, cy, box side}
 * (host: aug_bbox_DZI); lut_rgb (3,256) = float32((v/255 - mean_c)/std_c), RGB order.
 * Out: images_nchw (B,3,out,out) fp32, masks_out (B,out,out), bbox_trans (B,2,3), bbox_scale (B) = out/side. */
int kd6d_dzi_crop(const uint8_t* frames_bgr, const float* masks, int B, int H, int W, const float* center_scale,
                  const float* lut_rgb, int out_res, float* images_nchw, float* masks_out, float* bbox_trans,
                  float* bbox_scale, void* stream);

/* ---- optimiser: replaces clip_grad_norm_ + AdamW.step of train_kd.py:138-139 on one flat buffer.
 * kd6d_sumsq writes KD6D_SUMSQ_PARTS partial sums of x^2 (one per workgroup, unused slots zeroed; no atomics);
 * kd6d_clip_adamw adds them in a fixed order (reproducible), writes the total to *gnorm_sq_out (optional) and applies
 * g *= min(1, max_norm/(sqrt(total)+1e-6)) (gnorm_partials == NULL: no clipping) then the decoupled-weight-decay Adam update
 * (torch.optim.AdamW semantics, step counted from 1) and refreshes the bf16 shadow if given.
 * hyper_dev (optional, 4 floats on the device: lr, 1-beta1^t, sqrt(1-beta2^t), and a fourth that kd6d_set_hyper
 * sets to 0 -- the host passes it as gnorm_sq_out) overrides lr/step:
 * it lets the launch sit inside a captured hipGraph while the OneCycle schedule of
 * libs/train_libs.py:120 keeps advancing on the host; kd6d_set_hyper writes it (values travel in
 * the kernel arguments, so the host may run ahead of the device). */
#define KD6D_SUMSQ_PARTS 128
int kd6d_sumsq(const float* x, int64_t n, float* partials, void* stream);
int kd6d_clip_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                    const float* gnorm_partials, float* gnorm_sq_out, double max_norm, double lr, double beta1,
                    double beta2, double eps, double weight_decay, int64_t step, const float* hyper_dev,
                    void* bf16_shadow, void* stream);
int kd6d_set_hyper(float* hyper_dev, double lr, double beta1, double beta2, int64_t step, void* stream);
int kd6d_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);

/* ---- data-parallel exchange over RCCL / xGMI: replaces the reference's process-group plumbing for the step --
 * libs/distributed.py:9-41 (gloo rank / world / barrier helpers), train_kd.py:48-51 (init_process_group + barrier),
 * the DDP constructor's one-time parameter broadcast (libs/train_libs.py:123-130) -- and adds the per-step gradient
 * all-reduce the reference lacks (its DDP wrapper is discarded, libs/train_libs.py:130).
 * One communicator per process (= per GPU, the current HIP device at kd6d_comm_init).  Rendezvous: rank 0 calls
 * kd6d_comm_unique_id (128 bytes, host memory) and hands the id to the other ranks by any out-of-band channel
 * (the Python host uses the torch.distributed store); every rank then calls kd6d_comm_init (collective).
 * kd6d_comm_allreduce: in place, fp32, sum or mean over ranks, asynchronous on `stream` (stream-ordered behind
 * the caller's last weight gradient: no host synchronisation).  kd6d_comm_broadcast: in place, raw bytes from
 * `root`.  librccl is resolved at run time (a copy already mapped into the process is shared); without it these
 * return KD6D_ERR_UNSUPPORTED and everything else in this header still works. */
typedef struct kd6d_comm kd6d_comm;
int kd6d_comm_unique_id(void* id_out_host /* 128 bytes */);
int kd6d_comm_init(kd6d_comm** comm, int rank, int world, const void* unique_id_host /* 128 bytes */);
int kd6d_comm_rank(const kd6d_comm* comm);
int kd6d_comm_world(const kd6d_comm* comm);
int kd6d_comm_version(void);      /* RCCL version code (e.g. 22703), -1 without librccl */
int kd6d_comm_allreduce(kd6d_comm* comm, float* buf, int64_t n, int mean, void* stream);
int kd6d_comm_broadcast(kd6d_comm* comm, void* buf, int64_t nbytes, int root, void* stream);
int kd6d_comm_destroy(kd6d_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* KD6D_H */
