"""`SamplesLoss` and `KDPoseLoss` with the reference's signatures, on the kd6d kernels.

Reference: losses/kd_loss.py:13-38 (constructor, the geomloss object it builds), :40-109 (KDObjectSpaceLoss),
:111-161 (__call__); geomloss 0.2.4 `SamplesLoss("sinkhorn")` (third party, absent from the reference tree: the
arithmetic follows SURVEY.md App. B / oracle/sinkhorn_ref.py, parity unpinned at that boundary).
"""
import os

import torch

from .. import ops
from ..kd_losses import KDLoss, PackedTargets, TeacherKnowledge

_DENSE_DIMS = (2, 4, 8, 16)


class _SinkhornBatchFn(torch.autograd.Function):
    """(alpha (B,N), x (B,N,D), beta (B,M), y (B,M,D)) -> (B,) divergences, one epsilon schedule for the whole batch
    (geomloss measures the diameter over all B*(N+M) points).  Gradients reach alpha and x."""

    @staticmethod
    def forward(ctx, alpha, x, beta, y, p, blur, scaling, reach):
        B, N, D = x.shape
        M = y.shape[1]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        xd, yd = x.detach().to(torch.float32), y.detach().to(torch.float32)
        ad, bd = alpha.detach().to(torch.float32), beta.detach().to(torch.float32)
        cap = ops.lib.kd6d_sinkhorn_max_points()
        if D == 2 and B <= 8 and N <= cap and M <= cap:
            # one workgroup, one wave per problem; a batch shorter than 8 is filled with copies of problem 0 (same
            # points: the joint diameter does not move), whose results are dropped
            idx = torch.arange(8, device=dev) % B if B == 8 else torch.cat(
                [torch.arange(B, device=dev), torch.zeros(8 - B, dtype=torch.long, device=dev)])
            xs = xd[idx].permute(1, 0, 2).contiguous()            # (N,8,2)
            yt = yd[idx].permute(1, 0, 2).contiguous()
            al = ad[idx].t().contiguous()                         # (N,8)
            be = bd[idx].t().contiguous()
            i32 = dict(dtype=torch.int32, device=dev)
            zero = torch.zeros(1, **i32)
            loss_kp = torch.empty(1, 8, **f32)
            _, valid, gx, ga = ops.sinkhorn_div(xs, al, zero, torch.full((1,), N, **i32), yt, be, zero,
                                                torch.full((1,), M, **i32), 1, p, blur, scaling, reach, loss_kp=loss_kp)
            out = loss_kp[0, :B].clone()
            gx, ga = gx.permute(1, 0, 2)[:B].contiguous(), ga.t()[:B].contiguous()
        elif D in _DENSE_DIMS:
            pts = torch.cat([xd.reshape(-1, D), yd.reshape(-1, D)], 0)
            diameter = float((pts.max(0)[0] - pts.min(0)[0]).norm())          # the reference's .item() sync
            out = torch.empty(B, **f32)
            gx, ga = torch.empty(B, N, D, **f32), torch.empty(B, N, **f32)
            for b in range(B):
                l, g1, g2 = ops.sinkhorn_dense(xd[b].contiguous(), ad[b].contiguous(), yd[b].contiguous(),
                                               bd[b].contiguous(), blur, scaling, reach, diameter, p)
                out[b:b + 1].copy_(l)
                gx[b].copy_(g1)
                ga[b].copy_(g2)
        else:
            raise NotImplementedError("SamplesLoss on the HIP path: D=2 with <= %d points per set, or D in %s for "
                                      "larger sets (got D=%d)" % (cap, _DENSE_DIMS, D))
        ctx.save_for_backward(gx, ga)
        ctx.dtypes = (alpha.dtype, x.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        gx, ga = ctx.saved_tensors
        adt, xdt = ctx.dtypes
        return (g[:, None] * ga).to(adt), (g[:, None, None] * gx).to(xdt), None, None, None, None, None, None


class SamplesLoss:
    """geomloss.SamplesLoss(loss, p, blur, scaling, reach) as the reference builds it (losses/kd_loss.py:26-30):
    `L(alpha, x, beta, y) -> (B,)` for alpha (B,N), x (B,N,D), beta (B,M), y (B,M,D); `L(x, y)` uses uniform weights
    1/N, 1/M.  Debiased (unbalanced when reach is set) Sinkhorn divergence, cost |x-y|^2/2, epsilon-scaling from
    diameter^2 to blur^2; gradients flow into alpha and x (the student side of the KD loss), y and beta are
    constants (the reference feeds the teacher's no-grad outputs there)."""

    def __init__(self, loss="sinkhorn", p=2, blur=0.05, scaling=0.5, reach=None, **unsupported):
        if loss != "sinkhorn":
            raise NotImplementedError("only loss='sinkhorn' is implemented on the HIP path (got %r)" % (loss,))
        if float(p) != 2.0:
            raise NotImplementedError("only p=2 is implemented on the HIP path (got %r)" % (p,))
        if unsupported:
            raise NotImplementedError("SamplesLoss options not implemented on the HIP path: %s" % sorted(unsupported))
        self.loss, self.p, self.blur, self.scaling, self.reach = loss, float(p), float(blur), float(scaling), reach

    def __call__(self, *args):
        if len(args) == 2:
            x, y = args
            alpha = x.new_full(x.shape[:-1], 1.0 / x.shape[-2])
            beta = y.new_full(y.shape[:-1], 1.0 / y.shape[-2])
        elif len(args) == 4:
            alpha, x, beta, y = args
        else:
            raise TypeError("SamplesLoss: call with (x, y) or (alpha, x, beta, y)")
        squeeze = x.dim() == 2
        if squeeze:
            alpha, x, beta, y = alpha[None], x[None], beta[None], y[None]
        if y.requires_grad or beta.requires_grad:
            raise NotImplementedError("gradients w.r.t. the second measure (beta, y) are not computed on the HIP path")
        if not x.is_cuda:
            raise RuntimeError("kd6d SamplesLoss runs on the GPU only (no CPU fallback)")
        out = _SinkhornBatchFn.apply(alpha, x, beta, y, self.p, self.blur, self.scaling, self.reach)
        return out[0] if squeeze else out


class _ReferenceTeacher:
    """The reference's pred_t dict (models/model_kd.py:83-92: post_kp_2d (sum M,8,2) full-frame pixels,
    post_kp_cls (sum M,8) = sqrt(sigmoid), post_pos_per_img [M_i]) as the slot arrays the kernels read."""

    def __init__(self, pred_t, frame_wh, device):
        cnt = [int(c) for c in pred_t["post_pos_per_img"]]
        kp = pred_t["post_kp_2d"].detach().to(device=device, dtype=torch.float32).reshape(-1, 8, 2)
        sc = pred_t["post_kp_cls"].detach().to(device=device, dtype=torch.float32).reshape(-1, 8)
        assert kp.shape[0] == sum(cnt) == sc.shape[0], "pred_t: counts do not match the cell arrays"
        scale = torch.tensor(frame_wh, dtype=torch.float32, device=device)
        self.t_kp_norm = (kp / scale).contiguous() if kp.numel() else kp.new_zeros(1, 8, 2)
        self.t_beta = sc.pow(2).contiguous() if sc.numel() else sc.new_zeros(1, 8)      # kd_loss.py:82
        c = torch.tensor(cnt, dtype=torch.int32)
        self.t_cnt = c.to(device)
        self.t_start = (torch.cumsum(c, 0) - c).to(torch.int32).to(device)


class _KDPoseLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_packed, reg_packed, owner, levels, batch, tgt, teacher, keys):
        ev = owner.impl
        losses = ev.forward(cls_packed, reg_packed, levels, batch, tgt, teacher, keys=keys, seg_scale=None)
        ctx.owner, ctx.state = owner, ev.ctx
        return losses[0].clone(), losses[1].clone(), losses[2].clone()

    @staticmethod
    def backward(ctx, g_cls, g_reg, g_kd):
        ev = ctx.owner.impl
        ev.ctx = ctx.state
        cls_p, reg_p = ctx.state["cls"], ctx.state["reg"]
        w = torch.stack([g.to(torch.float32).reshape(()) for g in (g_cls, g_reg, g_kd)]).contiguous()
        dcls = torch.empty_like(cls_p)
        dreg = torch.zeros_like(reg_p)
        ev.backward(w, torch.float32, dcls, dreg)
        return dcls, dreg, None, None, None, None, None, None


class KDPoseLoss:
    """losses/kd_loss.py:13-161.  `KDPoseLoss(...)(pred_cls, pred_reg, targets, anchors, pred_t)` with pred_cls /
    pred_reg the per-level (B,15,H,W) / (B,240,H,W) head outputs -> [cls_loss, reg_loss, kd_loss] (focal SUM over the
    non-ignored cells, object-space SmoothL1 SUM over the positives, mean over images of the Sinkhorn divergences
    between the student's and the teacher's weighted keypoint votes), differentiable w.r.t. every tensor in pred_cls
    and pred_reg.

    `anchors` is accepted and ignored (one square anchor per cell: centres and sizes are closed forms of
    anchor_sizes / anchor_strides inside the kernels, models/model.py:229-281); `target_coder` and `top_k` are kept
    for signature compatibility (POINT / 3D targets only).  `pred_t` is the teacher forward's return value: the
    reference's dict or kd6d's TeacherKnowledge.  The positive subset of the SSC assignment is drawn from
    `self.keys` (uniform randoms, one per cell in packed order) when set, else from torch's generator.
    Deviations from the reference, as everywhere in this build: OT weights are gathered per
    cell (mixed-class batches work), a batch without positives gives reg = kd = 0."""

    def __init__(self, gamma, alpha, anchor_sizes, anchor_strides, positive_type, positive_num, positive_lambda,
                 top_k, internal_K, diameters, target_coder, cfg_kd=None):
        if positive_type != "SSC":
            raise NotImplementedError("the HIP path implements POSITIVE_TYPE=SSC (got %r)" % (positive_type,))
        ttype = getattr(target_coder, "target_type", "3D")
        if ttype != "3D":
            raise NotImplementedError("the HIP path implements LOSS_REG_TYPE=3D (got %r)" % (ttype,))
        self.anchor_sizes, self.anchor_strides = list(anchor_sizes), list(anchor_strides)
        self.cfg_kd = cfg_kd
        self.impl = KDLoss(internal_K, diameters, gamma, alpha, positive_num, positive_lambda,
                           cfg_kd if cfg_kd and "GTYPE" in cfg_kd else None)
        self.impl.anchor_sizes, self.impl.anchor_strides = self.anchor_sizes, self.anchor_strides
        if cfg_kd is not None:
            self.kd_loss = SamplesLoss(cfg_kd["GTYPE"], p=cfg_kd["GP"], blur=cfg_kd["GBLUR"],
                                       scaling=cfg_kd["SCALING"], reach=cfg_kd["REACH"])
            self.weighted_ot = cfg_kd["WEIGHTED_OT"]
            self.wot_detach = cfg_kd["DETACH"]
            if "vis_dir" in cfg_kd:
                self.vis_dir = cfg_kd["vis_dir"] + "/vis"
                os.makedirs(self.vis_dir, exist_ok=True)
        self.step = 0
        self.keys = None
        self.h, self.w = 480, 640            # kd_loss.py:116-117: the full frame, "not 256"

    @property
    def pos_per_img(self):
        """Positive cells per image of the last call (reads the device: one sync, like the reference's .item())."""
        return self.impl.ctx["pos_cnt"].cpu().tolist()

    @staticmethod
    def _pack(per_level, cpad):
        rows = [t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in per_level]      # level-major, image, (y, x)
        x = torch.cat(rows, 0).to(torch.float32)
        if x.shape[1] < cpad:
            x = torch.nn.functional.pad(x, (0, cpad - x.shape[1]))
        return x.contiguous()

    def __call__(self, pred_cls, pred_reg, targets, anchors, pred_t):
        dev = pred_cls[0].device
        if dev.type != "cuda":
            raise RuntimeError("kd6d KDPoseLoss runs on the GPU only (no CPU fallback)")
        batch = pred_cls[0].shape[0]
        levels = [tuple(t.shape[-2:]) for t in pred_cls]
        tgt = targets if isinstance(targets, PackedTargets) else PackedTargets(targets, dev)
        tgt.frame_wh = (float(self.w), float(self.h))
        if pred_t is None or isinstance(pred_t, TeacherKnowledge):
            teacher = pred_t
        else:
            teacher = _ReferenceTeacher(pred_t, tgt.frame_wh, dev)
        cls_p = self._pack(pred_cls, 16)
        reg_p = self._pack(pred_reg, pred_reg[0].shape[1])
        out = _KDPoseLossFn.apply(cls_p, reg_p, self, levels, batch, tgt, teacher, self.keys)
        self.step += 1
        return [out[0], out[1], out[2]]
