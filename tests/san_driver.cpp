// Host-side sanitizer driver (tests/test_sanitize_host.py): links against csrc/libkd6d_san.so -- the library built by
// `python kd-6d-pose-adlp_amd/build.py --sanitize` with -fsanitize=address,undefined on the HOST code of every csrc/*.hip
// (launchers, dispatch rules, work-list planners, argument checks) -- and drives the entry points that do their work on
// the host: argument checks (every call below must fail BEFORE a launch: there is no GPU in the CPU test environment),
// the dry-run dispatch of kd6d_conv2d_fwd_norm_fusable, the split planning of kd6d_conv2d_wgrad_parts and the grouped
// weight gradient's work-list planner.  Any ASan / UBSan report fails the test.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../include/kd6d.h"

static int fails = 0;
#define EXPECT(cond)                                                          \
  do {                                                                        \
    if (!(cond)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++fails; } \
  } while (0)

static kd6d_conv_geom geom(int batch, int cin, int cout, int k, int stride, std::vector<int> hw) {
  kd6d_conv_geom g;
  memset(&g, 0, sizeof(g));
  g.nseg = (int)hw.size() / 2; g.batch = batch; g.cin = cin; g.cout = cout; g.ksize = k; g.stride = stride; g.pad = k / 2;
  int rin = 0, rout = 0;
  for (int s = 0; s < g.nseg; ++s) {
    const int h = hw[2 * s], w = hw[2 * s + 1];
    g.seg[s].in_h = h; g.seg[s].in_w = w;
    g.seg[s].out_h = (h + 2 * g.pad - k) / stride + 1; g.seg[s].out_w = (w + 2 * g.pad - k) / stride + 1;
    g.seg[s].in_row0 = rin; g.seg[s].out_row0 = rout;
    rin += batch * h * w; rout += batch * g.seg[s].out_h * g.seg[s].out_w;
  }
  return g;
}

int main() {
  EXPECT(kd6d_abi_version() == KD6D_ABI_VERSION);
  // ---- option table ----
  long long v = -7;
  EXPECT(kd6d_reset_options() == 0);
  EXPECT(kd6d_get_option("bn.onepass_max", &v) == 0 && v == 65536);
  EXPECT(kd6d_set_option("conv.halo", 11) == 0 && kd6d_get_option("conv.halo", &v) == 0 && v == 11);
  EXPECT(kd6d_set_option("no.such.option", 1) < 0 && strstr(kd6d_last_error(), "unknown option"));
  EXPECT(kd6d_get_option(nullptr, &v) < 0);
  EXPECT(kd6d_reset_options() == 0);
  // ---- convolution entry points: geometry checks, null tensors ----
  kd6d_conv_geom g = geom(2, 12, 16, 3, 1, {8, 8});
  EXPECT(kd6d_conv2d_fwd(&g, KD6D_BF16, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) < 0 && strstr(kd6d_last_error(), "cin=12"));
  g = geom(2, 16, 16, 3, 1, {8, 8});
  EXPECT(kd6d_conv2d_fwd(&g, KD6D_BF16, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) < 0 && strstr(kd6d_last_error(), "null tensor"));
  EXPECT(kd6d_conv2d_fwd(nullptr, KD6D_BF16, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) < 0);
  EXPECT(kd6d_conv2d_fwd(&g, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) < 0);
  g.seg[0].out_h = 9;
  EXPECT(kd6d_conv2d_dgrad(&g, KD6D_F32, 0, 0, 0, 0, 0) < 0 && strstr(kd6d_last_error(), "inconsistent"));
  g = geom(2, 16, 16, 3, 1, {8, 8});
  g.nseg = 9;
  EXPECT(kd6d_conv2d_dgrad(&g, KD6D_BF16, 0, 0, 0, 0, 0) < 0);
  g = geom(2, 16, 16, 3, 1, {8, 8});
  EXPECT(kd6d_conv2d_wgrad(&g, KD6D_BF16, 0, 0, 0, 0, 0, 0, 0, 0) < 0);
  EXPECT(kd6d_conv2d_fwd_block(&g, KD6D_BF16, 0, 0, 0, 0, 0, 0, 77, 0) < 0);
  EXPECT(kd6d_conv2d_fwd_norm(&g, KD6D_BF16, 0, 0, 0, 0, 0, 0) < 0);
  // ---- host-only planning: the split count of every student layer shape (B = 16, 256 x 256), every budget ----
  struct L { int cin, cout, k, stride, h, w; };
  const L layers[] = {{8, 8, 3, 1, 256, 256}, {8, 16, 3, 1, 128, 128}, {16, 8, 1, 1, 64, 64}, {8, 64, 3, 1, 64, 64},
                      {64, 16, 1, 1, 32, 32}, {16, 128, 3, 1, 32, 32}, {128, 32, 1, 1, 16, 16}, {32, 256, 3, 1, 16, 16},
                      {256, 64, 1, 1, 16, 16}, {64, 128, 1, 1, 32, 32}, {64, 128, 3, 2, 16, 16}, {128, 128, 3, 2, 8, 8},
                      {128, 240, 3, 1, 32, 32}, {1024, 256, 3, 2, 8, 8}};
  for (const L& l : layers)
    for (int dt : {KD6D_BF16, KD6D_F32})
      for (int budget : {0, 1, 64, 128, 100000})
        for (int bias : {0, 1}) {
          kd6d_conv_geom q = geom(16, l.cin, l.cout, l.k, l.stride, {l.h, l.w});
          const int parts = kd6d_conv2d_wgrad_parts(&q, dt, bias, budget);
          EXPECT(parts >= 1 && parts <= 4096);
        }
  EXPECT(kd6d_conv2d_wgrad_parts(&g, KD6D_BF16, 0, -3) < 0);
  // ---- dry-run dispatch (no launch): which geometries take the fused conv + normalisation launch ----
  for (int c : {128, 256})
    for (int b : {1, 16, 48}) {
      kd6d_conv_geom q = geom(b, c, c, 3, 1, {32, 32, 16, 16, 8, 8, 4, 4, 2, 2});
      const int f = kd6d_conv2d_fwd_norm_fusable(&q, KD6D_BF16, KD6D_NORM_GROUP, 32);
      EXPECT(f == 0);                                 // the 2 x 2 level's statistics need the separate pass
      q = geom(b, c, c, 3, 1, {32, 32, 16, 16, 8, 8, 4, 4});
      const int f2 = kd6d_conv2d_fwd_norm_fusable(&q, KD6D_BF16, KD6D_NORM_GROUP, 32);
      EXPECT(f2 == 0 || f2 == 1);
      q = geom(b, 32, c, 3, 1, {16, 16});
      const int f3 = kd6d_conv2d_fwd_norm_fusable(&q, KD6D_BF16, KD6D_NORM_BATCH, 0);
      EXPECT(f3 == 0 || f3 == 1);
    }
  EXPECT(kd6d_conv2d_fwd_norm_fusable(nullptr, KD6D_BF16, KD6D_NORM_GROUP, 32) == 0);
  // ---- grouped weight gradient: the work-list planner runs on the host ----
  {
    std::vector<kd6d_wgrad_item> items;
    for (int cout : {128, 128, 128, 16, 240}) {
      kd6d_wgrad_item it;
      memset(&it, 0, sizeof(it));
      it.geom = geom(16, 128, cout, 3, 1, {32, 32, 16, 16, 8, 8, 4, 4});
      EXPECT(kd6d_wgrad_group_supported(&it.geom, KD6D_BF16) == 1);
      it.x = (const void*)0x1000; it.dy = (const void*)0x2000; it.dw = (float*)0x3000; it.dbias = (float*)0x4000;
      items.push_back(it);
    }
    kd6d_conv_geom odd = geom(16, 64, 128, 3, 2, {16, 16});
    EXPECT(kd6d_wgrad_group_supported(&odd, KD6D_BF16) == 0);
    for (int nwg : {1, 64, 128, 256, 512, 2048}) {
      int32_t info[4] = {0, 0, 0, 0};
      const int64_t nbytes = kd6d_wgrad_group_plan(items.data(), (int)items.size(), KD6D_BF16, nwg, nullptr, 0, info);
      EXPECT(nbytes > 0);
      std::vector<char> plan((size_t)nbytes);
      EXPECT(kd6d_wgrad_group_plan(items.data(), (int)items.size(), KD6D_BF16, nwg, plan.data(), nbytes, info) >= 0);
      EXPECT(info[0] >= 1 && info[1] >= 1);
      if (nbytes > 64) EXPECT(kd6d_wgrad_group_plan(items.data(), (int)items.size(), KD6D_BF16, nwg, plan.data(), nbytes - 64, info) < 0);
    }
    int32_t info[4];
    EXPECT(kd6d_wgrad_group_plan(nullptr, 3, KD6D_BF16, 64, nullptr, 0, info) < 0);
    EXPECT(kd6d_wgrad_group_plan(items.data(), 0, KD6D_BF16, 64, nullptr, 0, info) < 0);
    EXPECT(kd6d_wgrad_group_launch(nullptr, 1, 1, nullptr, nullptr) < 0);
  }
  // ---- normalisation / loss / optimiser entry points: bad channel counts, level tables, null pointers ----
  EXPECT(kd6d_colstats(KD6D_BF16, 0, 100, 12, 0, 0, 0) < 0);
  EXPECT(kd6d_bn_train_fwd(KD6D_BF16, 1, 0, 0, 100, 24, 0, 0, 0, 0, 1e-5f, 0.1f, 0, 0, 0, 0, 1, 0) < 0);
  EXPECT(kd6d_bn_train_bwd(KD6D_BF16, 1, 0, 0, 0, 100, 16, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 99, 0) < 0);
  EXPECT(kd6d_bn_pool_train_fwd(KD6D_BF16, 1, 0, 0, 2, 7, 8, 16, 0, 0, 0, 0, 1e-5f, 0.1f, 0, 0, 0, 0, 1, 0) < 0);
  const int32_t hw_ok[2] = {64, 16}, hw_bad[2] = {64, 0};
  EXPECT(kd6d_gn_relu_fwd(KD6D_BF16, 1, 0, 0, hw_bad, 2, 2, 128, 32, 0, 0, 1e-5f, 0, 0, 0) < 0);
  EXPECT(kd6d_gn_relu_fwd(KD6D_BF16, 1, 0, 0, hw_ok, 2, 2, 128, 32, 0, 0, 1e-5f, 0, 0, 0) < 0);      // null pointers
  EXPECT(kd6d_gn_relu_bwd(KD6D_BF16, 1, (void*)8, (void*)8, (void*)8, hw_ok, 2, 2, 128, 32, (float*)8, (float*)8, 1e-5f,
                          (kd6d_acc*)16, (kd6d_acc*)16, (int64_t*)8, (int64_t*)8, 0, KD6D_GN_WS_ZEROED, 0) < 0);   // stride 0
  EXPECT(kd6d_gn_relu_bwd_pair(KD6D_BF16, 1, 0, 0, hw_ok, 2, 2, 128, 32, 1e-5f, 4, 0, 0) < 0);
  EXPECT(kd6d_sinkhorn_div_fwd_bwd(0, 0, 0, 0, 0, 0, 0, 0, 4, 2.f, 0.001f, 0.5f, 0.5f, 0, 0, 0, 0, 0, 0) < 0);
  EXPECT(kd6d_sinkhorn_max_points() >= 64);
  EXPECT(kd6d_sinkhorn_dense_workspace_floats(16384, 16384, 16) > 0);
  EXPECT(kd6d_sinkhorn_dense_workspace_floats(100, 100, 3) < 0 || kd6d_sinkhorn_dense_workspace_floats(100, 100, 3) > 0);
  EXPECT(kd6d_focal_fwd(0, 0, 10, 2.f, 0.25f, 0, 0, 0) < 0);
  EXPECT(kd6d_sumsq(0, 10, 0, 0) < 0);
  EXPECT(kd6d_clip_adamw(0, 0, 0, 0, 10, 0, 0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1e-4, 1, 0, 0, 0) < 0);
  EXPECT(kd6d_acc_read(0, 10, KD6D_ACC_ACT, 0, 0, 0, 0) < 0);
  EXPECT(kd6d_acc_read((kd6d_acc*)16, 10, 5, (float*)16, 0, 0, 0) < 0);
  EXPECT(kd6d_grad_acc_resolve(0, 1, 1, 0, 0, 0, 0) < 0);
  kd6d_levels lv;
  memset(&lv, 0, sizeof(lv));
  lv.n = 9;
  EXPECT(kd6d_teacher_select(&lv, 0, 0, 0, 0.1f, 10.f, 1.f, 32, 640.f, 480.f, 0, 0, 0, 0, 0, 0, 0) < 0);
  EXPECT(kd6d_zero_regions(nullptr, nullptr, 0, 0) < 0);
  // ---- pair bracket: misuse is reported, nothing recorded leaks ----
  EXPECT(kd6d_conv2d_pair_end() < 0);
  EXPECT(kd6d_conv2d_pair_begin() == 0 && kd6d_conv2d_pair_begin() < 0);
  EXPECT(kd6d_conv2d_pair_pending() == 0);
  EXPECT(kd6d_conv2d_pair_end() == 0 || strstr(kd6d_last_error(), "launch failed"));     /* no device here: hipGetLastError may say so */
  // ---- communicator entry points without a communicator ----
  EXPECT(kd6d_comm_allreduce(nullptr, nullptr, 10, 1, nullptr) < 0);
  EXPECT(kd6d_comm_rank(nullptr) < 0 && kd6d_comm_world(nullptr) < 0);
  printf(fails ? "SAN_DRIVER_FAILED %d\n" : "SAN_DRIVER_OK\n", fails);
  return fails ? 1 : 0;
}
