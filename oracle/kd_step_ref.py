"""TEST INFRASTRUCTURE ONLY -- CPU (pure torch fp32) restatement of the reference KD step.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module;
the product path (kd-6d-pose-adlp_amd/kd6d) never does and fails loudly without libkd6d.so.

What is restated (reference file:line -> function here):
  backbone/darknet53.py:61-161,180-183 + backbone/common.py:250-324 -> DarkNet53Ref / ConvBlockRef
  backbone/darknet.py:48-135,163-171                                 -> DarkNetTinyRef
  models/model.py:40-103 (FPN, FPNTopP6P7)                          -> FPNRef
  models/model.py:370-451 (PoseHead, Scale)                         -> PoseHeadRef
  models/model.py:229-347 (AnchorGenerator)                         -> anchor_centers / anchors_for
  models/model.py:144-166 (TargetCoder.decode)                      -> decode_points
  postprocess/postprocess_kd.py:22-203 (teacher knowledge)          -> teacher_select
  losses/loss.py:164-268 (prepare_targets, SSC)                     -> ssc_assign
  losses/loss.py:20-40 (SigmoidFocalLoss)                           -> focal_loss_sum
  losses/kd_loss.py:40-109 (KDObjectSpaceLoss) + loss_libs.py:1-51  -> object_space_and_kd_loss
  geomloss SamplesLoss (absent; PARITY UNPINNED)     -> sinkhorn_divergence_torch
  losses/kd_loss.py:111-160 (KDPoseLoss.__call__)                   -> kd_pose_loss
  train_kd.py:104-140 + libs/train_libs.py:117-120                  -> KDStepRef

state_dict key names follow the reference so a reference state_dict loads
into these modules unchanged; tests/golden/make_golden.py checks this file against the
imported reference on identical inputs and writes the golden fixtures.

Known deliberate deviations (all documented in DESIGN.md):
  * the cv2.solvePnPRansac gate of postprocess_kd.py:187-202 is treated as always-true;
  * student OT weights are gathered per cell (pred_cls[i, cls_i]) instead of the reference's
    broadcast of pred_cls[..., unique(cls)] (kd_loss.py:43,83), identical for single-class batches;
  * a batch with no positive cell yields reg = kd = 0 (the reference returns sum(pred_reg)).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

INF = 100000000

# --------------------------------------------------------------------------------------
# bf16-storage emulation (test infrastructure for the bf16 mode of the HIP path)
# --------------------------------------------------------------------------------------
# The HIP path in bf16 mode computes every convolution with bf16 operands and fp32 accumulation and STORES
# activations / activation gradients as bf16; tensors feeding a normalisation, the logits, all parameters of the
# epilogues (bias, BN / GN gain and shift), the losses, the weight gradients and the optimiser are fp32.
# `with bf16_storage():` makes this restatement round at the same storage points -- fp32 arithmetic everywhere, values
# passed through bf16 where the engine stores bf16 (straight-through in the other direction):
#   _st(x)   value AND its gradient are stored bf16 (activations: conv inputs);
#   _stg(x)  only the gradient is (fp32 pre-normalisation tensors and logits, whose gradients the reverse sweep stores);
#   _w(w)    the bf16 shadow of a convolution weight (gradient reaches the fp32 master unrounded).
# With it off (default) the three are identities and the module is the plain fp32 restatement pinned by the goldens.
# Why it exists: the gradients of the student's first layers are so ill-conditioned at initialisation that rounding
# only the INPUT IMAGE to bf16 moves them by 40-70 % (tests/bf16_sensitivity.py, profiles/r03_bf16_sensitivity.md),
# so bf16-vs-fp32 deviations of those tensors measure the number format, not the kernels; against this emulation the
# kernels are held to the summation-order level.
_EMU = False


class bf16_storage:
    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global _EMU
        self.old, _EMU = _EMU, self.on
        return self

    def __exit__(self, *exc):
        global _EMU
        _EMU = self.old
        return False


def _r(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _Store(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd):
        return _r(x) if fwd else x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g), None


def _st(x):
    return _Store.apply(x, True) if _EMU else x


def _stg(x):
    return _Store.apply(x, False) if (_EMU and x.requires_grad) else x


def _w(w):
    return w + (_r(w) - w).detach() if _EMU else w


def _conv(m, x):
    """nn.Conv2d `m` applied with the bf16 shadow of its weight under emulation (bias stays fp32)."""
    if not _EMU:
        return m(x)
    return F.conv2d(x, _w(m.weight), m.bias, m.stride, m.padding)


# --------------------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------------------


class ConvBlockRef(nn.Module):
    """conv(no bias) + BN(eps 1e-5) + LeakyReLU(0.1)  (backbone/common.py:250-324)."""

    def __init__(self, cin, cout, k, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=1e-5)

    def forward(self, x, store=True):
        # engine: raw conv output fp32 (its gradient `draw` is stored bf16), activation stored bf16
        y = F.leaky_relu(self.bn(_stg(_conv(self.conv, x))), 0.1)
        return _st(y) if store else y


class DarkUnitRef(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = ConvBlockRef(cin, cout // 2, 1)
        self.conv2 = ConvBlockRef(cout // 2, cout, 3)

    def forward(self, x):
        # engine: the residual add is part of conv2's epilogue, one bf16 store behind it
        return _st(self.conv2(self.conv1(x), store=False) + x)


class DarkNet53Ref(nn.Module):
    def __init__(self):
        super().__init__()
        self.features = nn.Sequential()
        self.features.add_module("init_block", ConvBlockRef(3, 32, 3))
        cin = 32
        for i, (c, n) in enumerate(zip([64, 128, 256, 512, 1024], [2, 3, 9, 9, 5])):
            stage = nn.Sequential()
            for j in range(n):
                stage.add_module("unit%d" % (j + 1), ConvBlockRef(cin, c, 3, 2) if j == 0 else DarkUnitRef(cin, c))
                cin = c
            self.features.add_module("stage%d" % (i + 1), stage)
        self.output = nn.Linear(1024, 1000)   # registered, never used (darknet53.py:145-147)

    def forward(self, x):
        f = self.features
        o0 = f.init_block(_st(x))
        o1 = f.stage1(o0); o2 = f.stage2(o1); o3 = f.stage3(o2); o4 = f.stage4(o3); o5 = f.stage5(o4)
        return [o1, o2, o3, o4, o5]


TINY_CHANNELS = {
    "darknet_tiny": [[16], [32], [16, 128, 16, 128], [32, 256, 32, 256], [64, 512, 64, 512, 128]],
    "darknet_tiny_h": [[8], [16], [8, 64, 8, 64], [16, 128, 16, 128], [32, 256, 32, 256, 64]],
}


class DarkNetTinyRef(nn.Module):
    def __init__(self, arch):
        super().__init__()
        chans = TINY_CHANNELS[arch]
        self.features = nn.Sequential()
        cin = 3
        for i, per_stage in enumerate(chans):
            stage = nn.Sequential()
            for j, c in enumerate(per_stage):
                # darknet.py:92: pointwise iff the stage has >1 unit and the unit index is odd (1-based)
                pointwise = len(per_stage) > 1 and not (((j + 1) % 2 == 1) ^ True)
                stage.add_module("unit%d" % (j + 1), ConvBlockRef(cin, c, 1 if pointwise else 3))
                cin = c
            if i != len(chans) - 1:
                stage.add_module("pool%d" % (i + 1), nn.MaxPool2d(2, 2))
            self.features.add_module("stage%d" % (i + 1), stage)
        self.output = nn.Sequential()
        self.output.add_module("final_conv", nn.Conv2d(cin, 1000, 1))   # unused

    def forward(self, x):
        f = self.features
        x = _st(x)                                      # the packed NHWC input image is bf16 in bf16 mode
        o1 = f.stage1(x); o2 = f.stage2(o1); o3 = f.stage3(o2)
        o4 = f.stage5(f.stage4(o3))
        return [o1, o2, o3, o4]


class _P6P7(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.p6 = nn.Conv2d(cin, cout, 3, 2, 1)
        self.p7 = nn.Conv2d(cout, cout, 3, 2, 1)


class FPNRef(nn.Module):
    def __init__(self, in_channels, out_channel):
        super().__init__()
        self.inner_convs = nn.ModuleList()
        self.out_convs = nn.ModuleList()
        for c in in_channels:
            self.inner_convs.append(None if c == 0 else nn.Conv2d(c, out_channel, 1))
            self.out_convs.append(None if c == 0 else nn.Conv2d(out_channel, out_channel, 3, padding=1))
        self.top_blocks = _P6P7(in_channels[-1], out_channel)

    def forward(self, inputs):
        inner = _st(_conv(self.inner_convs[-1], inputs[-1]))
        outs = [_st(_conv(self.out_convs[-1], inner))]
        for feat, ic, oc in zip(inputs[:-1][::-1], list(self.inner_convs)[:-1][::-1],
                                list(self.out_convs)[:-1][::-1]):
            if ic is None:
                continue
            inner = _st(_st(_conv(ic, feat)) + F.interpolate(inner, scale_factor=2, mode="nearest"))
            outs.insert(0, _st(_conv(oc, inner)))
        p6 = _st(_conv(self.top_blocks.p6, inputs[-1]))            # use_p5=True: raw last backbone map
        p7 = _st(_conv(self.top_blocks.p7, _st(F.relu(p6))))
        return outs + [p6, p7]


class _Scale(nn.Module):
    def __init__(self):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor([1.0]))


class PoseHeadRef(nn.Module):
    def __init__(self, c, n_class=16, n_conv=4, prior=0.01):
        super().__init__()
        def tower():
            mods = []
            for _ in range(n_conv):
                mods += [nn.Conv2d(c, c, 3, padding=1), nn.GroupNorm(32, c), nn.ReLU()]
            return nn.Sequential(*mods)
        self.cls_tower = tower()
        self.pose_tower = tower()
        self.cls_logits = nn.Conv2d(c, n_class - 1, 3, padding=1)
        self.pose_pred = nn.Conv2d(c, (n_class - 1) * 16, 3, padding=1)
        self.scales = nn.ModuleList([_Scale() for _ in range(5)])
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.01)
                nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.cls_logits.bias, -math.log((1 - prior) / prior))

    def forward(self, feats):
        cls, reg = [], []
        for l, f in enumerate(feats):
            if not _EMU:
                cls.append(self.cls_logits(self.cls_tower(f)))
                reg.append(self.pose_pred(self.pose_tower(f)) * self.scales[l].scale)
                continue
            outs = []
            for tower, final in ((self.cls_tower, self.cls_logits), (self.pose_tower, self.pose_pred)):
                x = f
                for i in range(0, len(tower), 3):      # conv -> fp32 pre-GN tensor -> GroupNorm + ReLU -> bf16 store
                    x = _st(tower[i + 2](tower[i + 1](_stg(_conv(tower[i], x)))))
                outs.append(_stg(_conv(final, x)))     # fp32 logits; their gradients enter the sweep as bf16
            cls.append(outs[0])
            reg.append(outs[1] * self.scales[l].scale)
        return cls, reg


class _AnchorBuffers(nn.Module):
    def __init__(self, sizes, strides):
        super().__init__()
        self.cell_anchors = nn.Module()
        for i, (s, st) in enumerate(zip(sizes, strides)):
            c = st / 2.0
            self.cell_anchors.register_buffer(
                str(i), torch.tensor([[c - 0.5 * (s - 1), c - 0.5 * (s - 1), c + 0.5 * (s - 1), c + 0.5 * (s - 1)]],
                                     dtype=torch.float32))


BACKBONE_CFG = {
    # arch: (feat_channels, out_channel)   arguments/argument.py:59-68
    "darknet53": ([0, 0, 256, 512, 1024], 256),
    "darknet_tiny": ([0, 0, 128, 128], 256),
    "darknet_tiny_h": ([0, 0, 64, 64], 128),
}
ANCHOR_SIZES = [32, 64, 128, 256, 512]
ANCHOR_STRIDES = [8, 16, 32, 64, 128]


class PoseNetRef(nn.Module):
    """backbone + FPN + head with the reference's sub-module names (models/model.py:455-487)."""

    def __init__(self, arch):
        super().__init__()
        self.arch = arch
        feat, oc = BACKBONE_CFG[arch]
        self.backbone = DarkNet53Ref() if arch == "darknet53" else DarkNetTinyRef(arch)
        self.fpn = FPNRef(feat, oc)
        self.head = PoseHeadRef(oc)
        self.anchor_generator = _AnchorBuffers(ANCHOR_SIZES, ANCHOR_STRIDES)
        for m in self.backbone.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight)
        for m in self.fpn.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)

    def forward(self, images):
        feats = self.fpn(self.backbone(images))
        return self.head(feats)       # (cls list, reg list), NCHW per level


def seeded_state_dict(module, seed):
    """Deterministic weights from numpy default_rng(seed) (never stored in fixtures).
    conv/linear weights ~ N(0, 1/sqrt(fan_in)) * 1.4, BN/GN gamma ~ U(0.5,1.5), biases small,
    running_mean ~ N(0,0.1), running_var ~ U(0.5,1.5)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for k, v in module.state_dict().items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros_like(v)
        elif "cell_anchors" in k or k.endswith(".scale"):
            sd[k] = v.clone()
        elif k.endswith("running_mean"):
            sd[k] = torch.from_numpy(rng.normal(0, 0.1, shp).astype(np.float32))
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif v.dim() >= 2:
            fan_in = int(np.prod(shp[1:]))
            sd[k] = torch.from_numpy((rng.normal(0, 1.0, shp) * 1.4 / math.sqrt(fan_in)).astype(np.float32))
        elif k.endswith("weight"):     # BN / GN gamma
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        else:                          # biases / beta
            sd[k] = torch.from_numpy(rng.normal(0, 0.05, shp).astype(np.float32))
    return sd


# --------------------------------------------------------------------------------------
# anchors, decoding
# --------------------------------------------------------------------------------------


def level_shapes(h, w, n_levels):
    """Feature-map grids of the pyramid for an (h, w) network input."""
    shapes = []
    hh, ww = h // 8, w // 8
    for _ in range(n_levels):
        shapes.append((hh, ww))
        hh, ww = (hh + 1) // 2, (ww + 1) // 2     # 3x3 s2 p1 conv and /2 pooling agree on even sizes
    return shapes


def anchor_centers(shapes, strides=ANCHOR_STRIDES):
    """(cells,2) centres, (cells,) sizes, (cells,) level ids; row-major per level (model.py:229-281)."""
    cs, ss, ls = [], [], []
    for l, (h, w) in enumerate(shapes):
        st = strides[l]
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32),
                                indexing="ij")
        cs.append(torch.stack([xs.reshape(-1) * st + st / 2.0, ys.reshape(-1) * st + st / 2.0], 1))
        ss.append(torch.full((h * w,), float(ANCHOR_SIZES[l])))
        ls.append(torch.full((h * w,), l, dtype=torch.long))
    return torch.cat(cs), torch.cat(ss), torch.cat(ls)


def decode_points(pred16, centers, sizes, bbox_trans=None):
    """TargetCoder.decode (model.py:144-166): pred16 (n,16) = 8 x-offsets then 8 y-offsets.
    Returns (n,8,2) points; with bbox_trans (n,2,3) they are mapped back to the full frame."""
    px = pred16[:, :8] * sizes[:, None] + centers[:, 0:1]
    py = pred16[:, 8:] * sizes[:, None] + centers[:, 1:2]
    if bbox_trans is not None:
        A = bbox_trans[:, :, :2]
        t = bbox_trans[:, :, 2]
        Ainv = torch.inverse(A)
        dx, dy = px - t[:, 0:1], py - t[:, 1:2]
        px, py = Ainv[:, 0, 0:1] * dx + Ainv[:, 0, 1:2] * dy, Ainv[:, 1, 0:1] * dx + Ainv[:, 1, 1:2] * dy
    return torch.stack([px, py], -1)


def flatten_levels(per_level):
    """list of (B,C,H,W) -> (B, cells, C) image-major / level / row-major (loss.py:62-96)."""
    return torch.cat([t.permute(0, 2, 3, 1).reshape(t.shape[0], -1, t.shape[1]) for t in per_level], 1)


def level_weights(size, n_levels, positive_num=10, positive_lambda=1.0):
    """n_k per level: int(10 * w_k / sum w + .5), w_k = exp(-lambda * log2(size/S_k)^2)."""
    lv = torch.tensor(ANCHOR_SIZES[:n_levels], dtype=torch.float32)
    dk = torch.log2(size / lv)
    nk = torch.exp(-positive_lambda * dk * dk)
    nk = positive_num * nk / nk.sum()
    return (nk + 0.5).int()


# --------------------------------------------------------------------------------------
# teacher knowledge extraction
# --------------------------------------------------------------------------------------


def teacher_select(cls_levels, reg_levels, bbox_trans, th=0.1, positive_num=10, positive_lambda=1.0):
    """postprocess_kd.py:22-203 with the PnP gate always true.
    cls_levels/reg_levels: lists of (B,15,H,W)/(B,240,H,W); bbox_trans (B,2,3).
    Returns per-image lists: scores (n,8) [= sqrt(sigmoid)], kps (n,8,2) full-frame."""
    B = cls_levels[0].shape[0]
    shapes = [tuple(t.shape[-2:]) for t in cls_levels]
    L = len(shapes)
    out_scores, out_kps = [], []
    for b in range(B):
        per_level = []
        for l in range(L):
            h, w = shapes[l]
            st, sz = ANCHOR_STRIDES[l], float(ANCHOR_SIZES[l])
            sc = torch.sigmoid(cls_levels[l][b].permute(1, 2, 0).reshape(h * w, -1))
            rg = reg_levels[l][b].permute(1, 2, 0).reshape(h * w, -1, 16)
            loc, cl = torch.nonzero(sc > th, as_tuple=True)
            if loc.numel() == 0:
                per_level.append(None)
                continue
            cx = (loc % w).float() * st + st / 2.0
            cy = (loc // w).float() * st + st / 2.0
            det = decode_points(rg[loc, cl], torch.stack([cx, cy], 1), torch.full((loc.numel(),), sz))
            per_level.append((det, cl + 1, torch.sqrt(sc[loc, cl])))
        got = [p for p in per_level if p is not None]
        result = None
        if got:
            labels = torch.unique(torch.cat([p[1] for p in got]))
            for lb in labels:
                box_size = torch.tensor(0.0)
                box_conf = torch.tensor(0.0)
                dets, scs = [None] * L, [None] * L
                for l, p in enumerate(per_level):
                    if p is None:
                        continue
                    m = p[1] == lb
                    dets[l], scs[l] = p[0][m], p[2][m]
                    if scs[l].numel() > 0:
                        i = torch.argmax(scs[l])
                        if scs[l][i] > box_conf:
                            box_conf = scs[l][i]
                            k = dets[l][i]
                            size = torch.maximum(k[:, 0].max() - k[:, 0].min(), k[:, 1].max() - k[:, 1].min())
                            if size > box_size:
                                box_size = size
                nk = level_weights(box_size, len(ANCHOR_SIZES), positive_num, positive_lambda)
                sel_d, sel_s = [], []
                for l in range(L):
                    if scs[l] is None:
                        continue
                    n = min(int(scs[l].numel()), int(nk[l]))
                    if n > 0:
                        s, idx = scs[l].topk(n)
                        sel_d.append(dets[l][idx]); sel_s.append(s)
                if not sel_s:
                    continue
                d = torch.cat(sel_d); s = torch.cat(sel_s)
                A = bbox_trans[b][:, :2]; t = bbox_trans[b][:, 2]
                d = (d - t) @ torch.inverse(A).T
                result = (s[:, None].expand(-1, 8).contiguous(), d)
                break            # only the first class result per image is kept (postprocess_kd.py:86-90)
        if result is None:
            out_scores.append(torch.zeros(0, 8)); out_kps.append(torch.zeros(0, 8, 2))
        else:
            out_scores.append(result[0]); out_kps.append(result[1])
    return out_scores, out_kps


# --------------------------------------------------------------------------------------
# SSC target assignment
# --------------------------------------------------------------------------------------


def project_box(target):
    """PoseAnnot.to_object_boxlist (libs/poses.py:264-304) -> (G,4) xyxy in crop coordinates."""
    boxes = []
    for i in range(len(target["class_ids"])):
        if not bool((target["mask"] == (i + 1)).any()):
            boxes.append([0.0, 0.0, 0.0, 0.0]); continue
        kp = target["keypoints_3d"][target["class_ids"][i]]
        reps = target["K"] @ (target["rotations"][i] @ kp.t() + target["translations"][i].reshape(3, 1))
        xs = reps[0] / (reps[2] + 1e-8); ys = reps[1] / (reps[2] + 1e-8)
        bt = target["bbox_trans"]
        xs, ys = bt[0, 0] * xs + bt[0, 1] * ys + bt[0, 2], bt[1, 0] * xs + bt[1, 1] * ys + bt[1, 2]
        boxes.append([float(xs.min()), float(ys.min()), float(xs.max()), float(ys.max())])
    return torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4)


def ssc_assign(targets, shapes, positive_num=10, positive_lambda=1.0, choose=None):
    """loss.py:164-268.  choose(valid_pos, n, image, level, gt) -> indices into valid_pos; default
    is the reference's torch.randperm(len(valid_pos))[:n] (same global RNG call order).
    Returns labels (B, cells) int64 {-1,0,c+1}, gt index per cell (B, cells), aux_3D (B,cells,8,3)."""
    centers, _, lvl = anchor_centers(shapes)
    L = len(shapes)
    counts = [h * w for (h, w) in shapes]
    labels_all, gt_all, aux_all = [], [], []
    for im, t in enumerate(targets):
        G = len(t["class_ids"])
        boxes = project_box(t)
        span = torch.maximum(boxes[:, 2] - boxes[:, 0] + 1, boxes[:, 3] - boxes[:, 1] + 1)
        H, W = t["mask"].shape
        cx = centers[:, 0].clamp(0, W - 1).long(); cy = centers[:, 1].clamp(0, H - 1).long()
        at = t["mask"][cy, cx]
        in_mask = torch.stack([(at == (g + 1)) for g in range(G)], 1).long()         # (cells, G)
        lv = torch.tensor(ANCHOR_SIZES[:L], dtype=torch.float32)
        dk = torch.log2(span.view(1, -1) / lv.view(-1, 1)).abs()
        nk = torch.exp(-positive_lambda * dk * dk)
        nk = (positive_num * nk / nk.sum(0, keepdim=True) + 0.5).int()              # (L, G)
        cand = [[] for _ in range(G)]
        start = 0
        for l in range(L):
            end = start + counts[l]
            for g in range(G):
                vp = in_mask[start:end, g].nonzero().view(-1)
                n = min(int(nk[l][g]), len(vp))
                ridx = torch.randperm(len(vp))[:n] if choose is None else choose(vp, n, im, l, g)
                cand[g].append(vp[ridx] + start)
            start = end
        roi = torch.full_like(in_mask, -INF)
        for g in range(G):
            roi[torch.cat(cand[g]), g] = 1
        val, gidx = roi.max(dim=1)
        lab = (t["class_ids"] + 1)[gidx].clone()
        lab[val == -INF] = 0
        vis = in_mask.max(dim=1)[0]
        lab[(vis == 1) & (lab == 0)] = -1
        kp = t["keypoints_3d"][t["class_ids"][gidx]]                                  # (cells,8,3)
        aux = torch.bmm(t["rotations"][gidx], kp.transpose(1, 2)) + t["translations"][gidx].reshape(-1, 3, 1)
        labels_all.append(lab); gt_all.append(gidx); aux_all.append(aux.transpose(1, 2))
    return torch.stack(labels_all), torch.stack(gt_all), torch.stack(aux_all)


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------


def focal_loss_sum(logits, labels, gamma=2.0, alpha=0.25, eps=1e-4):
    """loss.py:20-40; logits (V,15), labels (V,) in {0, c+1}; SUM reduction."""
    ids = torch.arange(1, logits.shape[1] + 1, dtype=labels.dtype).unsqueeze(0)
    t = labels.unsqueeze(1)
    p = torch.clamp(torch.sigmoid(logits), eps, 1 - eps)
    term1 = (1 - p) ** gamma * torch.log(p)
    term2 = p ** gamma * torch.log(1 - p)
    loss = -(t == ids).float() * alpha * term1 - ((t != ids) * (t >= 0)).float() * (1 - alpha) * term2
    return loss.sum()


def _softmin_t(eps, C, h):
    return -eps * (h.unsqueeze(1) - C / eps).logsumexp(2)


def sinkhorn_divergence_torch(alpha, x, beta, y, blur=0.001, scaling=0.5, reach=0.5):
    """geomloss 0.2.4 SamplesLoss('sinkhorn', p=2) tensorized path, SURVEY App. B, under torch
    autograd (fp32): alpha (B,N), x (B,N,D), beta (B,M), y (B,M,D) -> (B,).  PARITY UNPINNED."""
    from oracle.sinkhorn_ref import epsilon_schedule
    D = x.shape[-1]
    xd, yd = x.detach(), y.detach()
    pts = torch.cat([xd.reshape(-1, D), yd.reshape(-1, D)], 0)
    diameter = max(float((pts.max(0)[0] - pts.min(0)[0]).norm()), 1e-12)
    eps_s = epsilon_schedule(2, diameter, blur, scaling)
    rho = None if reach is None else reach ** 2
    lam = (lambda e: 1.0) if rho is None else (lambda e: 1.0 / (1.0 + e / rho))

    def cost(u, v):
        # geomloss squared_distances: |u|^2 - 2 u.v + |v|^2, halved
        return ((u * u).sum(-1).unsqueeze(2) - 2 * torch.matmul(u, v.permute(0, 2, 1))
                + (v * v).sum(-1).unsqueeze(1)) / 2

    def logw(a):
        l = a.detach().log()
        l[a.detach() <= 0] = -100000
        return l

    a_log, b_log = logw(alpha), logw(beta)
    C_xx, C_yy, C_xy, C_yx = cost(x, xd), cost(y, yd), cost(x, yd), cost(y, xd)
    with torch.no_grad():
        eps = eps_s[0]; l = lam(eps)
        a_x = l * _softmin_t(eps, C_xx, a_log); b_y = l * _softmin_t(eps, C_yy, b_log)
        a_y = l * _softmin_t(eps, C_yx, a_log); b_x = l * _softmin_t(eps, C_xy, b_log)
        for eps in eps_s:
            l = lam(eps)
            at_x = l * _softmin_t(eps, C_xx, a_log + a_x / eps); bt_y = l * _softmin_t(eps, C_yy, b_log + b_y / eps)
            at_y = l * _softmin_t(eps, C_yx, a_log + b_x / eps); bt_x = l * _softmin_t(eps, C_xy, b_log + a_y / eps)
            a_x, b_y = 0.5 * (a_x + at_x), 0.5 * (b_y + bt_y)
            a_y, b_x = 0.5 * (a_y + at_y), 0.5 * (b_x + bt_x)
    l = lam(eps)
    a_x2 = l * _softmin_t(eps, C_xx, (a_log + a_x / eps).detach())
    b_y2 = l * _softmin_t(eps, C_yy, (b_log + b_y / eps).detach())
    a_y2 = l * _softmin_t(eps, C_yx, (a_log + b_x / eps).detach())
    b_x2 = l * _softmin_t(eps, C_xy, (b_log + a_y / eps).detach())
    if rho is None:
        return (alpha * (b_x2 - a_x2)).sum(1) + (beta * (a_y2 - b_y2)).sum(1)
    w = rho + eps / 2
    return ((alpha * w * ((-a_x2 / rho).exp() - (-b_x2 / rho).exp())).sum(1)
            + (beta * w * ((-b_y2 / rho).exp() - (-a_y2 / rho).exp())).sum(1))


def kd_pose_loss(cls_levels, reg_levels, targets, teacher, K, diameters, labels, gt_idx, aux3d,
                 kd=None, ot_fn=None, frame_wh=(640.0, 480.0)):
    """KDPoseLoss.__call__ (kd_loss.py:111-160) + KDObjectSpaceLoss (:40-109) + kd_loss_2d.
    labels/gt_idx/aux3d come from ssc_assign.  teacher = (scores list, kps list) or None.
    Returns dict(loss_cls, loss_reg, loss_kd) plus bookkeeping (pos_per_img, student points)."""
    kd = kd or dict(blur=0.001, scaling=0.5, reach=0.5, weighted=True, detach=False)
    ot_fn = ot_fn or sinkhorn_divergence_torch
    B = labels.shape[0]
    shapes = [tuple(t.shape[-2:]) for t in cls_levels]
    centers, sizes, _ = anchor_centers(shapes)
    cells = centers.shape[0]
    cls_f = flatten_levels(cls_levels).reshape(B * cells, -1)
    reg_f = flatten_levels(reg_levels).reshape(B * cells, -1)
    lab = labels.reshape(-1)
    pos = torch.nonzero(lab > 0).squeeze(1)
    valid = torch.nonzero(lab >= 0).squeeze(1)
    loss_cls = focal_loss_sum(cls_f[valid], lab[valid])
    out = dict(loss_cls=loss_cls, pos_per_img=[int((labels[b] > 0).sum()) for b in range(B)])
    if pos.numel() == 0:
        z = cls_f.sum() * 0.0
        out.update(loss_reg=z, loss_kd=z)
        return out
    img_of = pos // cells
    cell_of = pos % cells
    cls_lab = lab[pos] - 1
    bt = torch.stack([t["bbox_trans"] for t in targets])[img_of]                      # (P,2,3)
    pred16 = reg_f[pos].view(pos.numel(), -1, 16)[torch.arange(pos.numel()), cls_lab]
    pts = decode_points(pred16, centers[cell_of], sizes[cell_of], bt)                 # (P,8,2) full frame
    X = aux3d.reshape(B * cells, 8, 3)[pos]                                           # camera frame
    d = torch.as_tensor(diameters, dtype=torch.float32)[cls_lab].view(-1, 1, 1)
    Kinv = torch.inverse(torch.as_tensor(K, dtype=torch.float32).view(3, 3))
    hom = torch.cat([pts, torch.ones_like(pts[..., :1])], -1)                        # (P,8,3)
    b = hom @ Kinv.T
    proj = b * ((b * X).sum(-1, keepdim=True) / (b * b).sum(-1, keepdim=True))        # (b b^T / b^T b) X
    l1 = F.smooth_l1_loss(50.0 * proj / d, 50.0 * X / d, reduction="none").reshape(pos.numel(), -1).mean(1) / 50.0
    out["loss_reg"] = l1.sum()
    out["student_pts"] = pts
    # ---- KD ------------------------------------------------------------------------------
    if teacher is None:
        out["loss_kd"] = cls_f.sum() * 0.0
        return out
    t_scores, t_kps = teacher
    prob = torch.clamp(torch.sigmoid(cls_f[pos]), min=10e-4, max=1 - 10e-4)
    a_cell = prob[torch.arange(pos.numel()), cls_lab]
    if kd.get("detach", False):
        a_cell = a_cell.detach()
    w, h = frame_wh
    scale = torch.tensor([w, h], dtype=torch.float32)
    losses = []
    s0 = 0
    for bi in range(B):
        n = out["pos_per_img"][bi]
        m = t_scores[bi].shape[0]
        if n == 0 or m == 0:
            s0 += n
            continue
        xs = (pts[s0:s0 + n] / scale).transpose(0, 1).contiguous()                   # (8,n,2)
        ys = (t_kps[bi] / scale).transpose(0, 1).contiguous()
        if kd.get("weighted", True):
            al = a_cell[s0:s0 + n].view(1, n).expand(8, n).contiguous()
            be = t_scores[bi].pow(2).transpose(0, 1).contiguous()
        else:
            al = torch.full((8, n), 1.0 / n); be = torch.full((8, m), 1.0 / m)
        losses.append(ot_fn(al, xs, be, ys, kd["blur"], kd["scaling"], kd["reach"]).sum())
        s0 += n
    out["loss_kd"] = sum(losses) / len(losses) if losses else cls_f.sum() * 0.0
    out["alpha_cell"] = a_cell
    return out


# --------------------------------------------------------------------------------------
# the step
# --------------------------------------------------------------------------------------


class KDStepRef:
    """train_kd.py:104-140: teacher no-grad fwd -> select -> student fwd -> losses -> backward ->
    clip(1.0) -> AdamW(lr=base_lr/n_gpu, wd 1e-4, eps 1e-8) -> OneCycleLR(linear, pct .05)."""

    def __init__(self, student_arch="darknet_tiny_h", teacher_arch="darknet53", K=None, diameters=None,
                 kd_weight=5.0, base_lr=1e-3, max_iter=10000, n_gpu=1, w_cls=0.1, w_reg=1.0,
                 kd=None, student_seed=1, teacher_seed=2, teacher_cls_bias=None, emulate_bf16=False):
        # emulate_bf16: both networks run under bf16_storage() (values rounded where the HIP path's bf16 mode stores
        # bf16; see the top of this file) -- the yardstick for the bf16 parity tests, never a product path
        self.emulate_bf16 = bool(emulate_bf16)
        self.student = PoseNetRef(student_arch)
        self.student.load_state_dict(seeded_state_dict(self.student, student_seed))
        self.teacher = None
        if teacher_arch:
            self.teacher = PoseNetRef(teacher_arch)
            sd = seeded_state_dict(self.teacher, teacher_seed)
            if teacher_cls_bias is not None:
                sd["head.cls_logits.bias"] = torch.as_tensor(teacher_cls_bias, dtype=torch.float32)
            self.teacher.load_state_dict(sd)
            self.teacher.eval()
        self.student.train()
        self.K, self.diameters = K, diameters
        self.kd_weight, self.w_cls, self.w_reg = kd_weight, w_cls, w_reg
        self.kd = kd or dict(blur=0.001, scaling=0.5, reach=0.5, weighted=True, detach=False)
        lr = base_lr / n_gpu
        self.opt = torch.optim.AdamW(self.student.parameters(), lr=lr, weight_decay=1e-4, eps=1e-8)
        self.sched = torch.optim.lr_scheduler.OneCycleLR(self.opt, lr, max_iter + 100, pct_start=0.05,
                                                         cycle_momentum=False, anneal_strategy="linear")

    def teacher_knowledge(self, images, targets):
        with torch.no_grad(), bf16_storage(self.emulate_bf16):
            cls_t, reg_t = self.teacher(images)
        bt = torch.stack([t["bbox_trans"] for t in targets])
        return teacher_select(cls_t, reg_t, bt), (cls_t, reg_t)

    def forward_backward(self, images, targets, choose=None):
        """train_kd.py:104-137: teacher, student, losses, weighting, backward.  Leaves the UNCLIPPED gradients in
        p.grad (what one data-parallel rank contributes to the mean all-reduce)."""
        self.student.zero_grad()
        teacher = None
        extras = {}
        if self.teacher is not None:
            teacher, t_logits = self.teacher_knowledge(images, targets)
            extras["teacher_logits"] = t_logits
            extras["teacher"] = teacher
        with bf16_storage(self.emulate_bf16):
            cls_s, reg_s = self.student(images)
        shapes = [tuple(t.shape[-2:]) for t in cls_s]
        labels, gt_idx, aux = ssc_assign(targets, shapes, choose=choose)
        out = kd_pose_loss(cls_s, reg_s, targets, teacher, self.K, self.diameters, labels, gt_idx, aux, self.kd)
        loss = out["loss_cls"] * self.w_cls + out["loss_reg"] * self.w_reg
        if self.kd_weight > 0:
            loss = loss + out["loss_kd"] * self.kd_weight
        loss.backward()
        extras.update(student_logits=(cls_s, reg_s), labels=labels, out=out)
        return out, extras

    def optimizer_step(self):
        """train_kd.py:138-140 on whatever p.grad holds: clip_grad_norm_(1.0), AdamW, OneCycle.  Returns the
        pre-clip global gradient norm."""
        gn = nn.utils.clip_grad_norm_(self.student.parameters(), 1.0)
        self.opt.step()
        self.sched.step()
        return float(gn)

    def step(self, images, targets, choose=None, return_extras=False):
        out, extras = self.forward_backward(images, targets, choose)
        gn = self.optimizer_step()
        res = dict(loss_cls=float(out["loss_cls"]), loss_reg=float(out["loss_reg"]),
                   loss_kd=float(out["loss_kd"]), grad_norm=gn)
        if return_extras:
            return res, extras
        return res
