"""`kd_loss_2d` with the reference's signature (losses/loss_libs.py:1-51).

pred_xy (P*8, 2) / target_xy (M*8, 2): full-frame keypoint votes of the student's positive cells and of the
teacher's selected cells, image after image; pred_cls (P, 8) / target_cls (M, 8) their OT weights (or None);
pos_per_img / pos_per_img_t the cells per image.  Returns one scalar per image that has both sets non-empty:
the sum over the 8 keypoints of kd_loss(alpha, x, beta, y).

With kd6d's SamplesLoss (weighted, level 'point', dim 2, sets within kd6d_sinkhorn_max_points) all images go out as
ONE kd6d_sinkhorn_div_fwd_bwd launch; any other callable is applied image by image exactly as the reference does.
"""
import torch

from .. import ops
from .kd_loss import SamplesLoss


class _PackedSinkhornFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xs, alpha, yt, beta, s_cnt, t_cnt, kd):
        dev = xs.device
        i32 = dict(dtype=torch.int32, device=dev)
        sc, tc = torch.tensor(s_cnt, dtype=torch.int32), torch.tensor(t_cnt, dtype=torch.int32)
        s_start, t_start = (torch.cumsum(sc, 0) - sc).to(**i32), (torch.cumsum(tc, 0) - tc).to(**i32)
        loss, valid, gx, ga = ops.sinkhorn_div(xs.detach().contiguous(), alpha.detach().contiguous(), s_start, sc.to(**i32),
                                               yt.detach().contiguous(), beta.detach().contiguous(), t_start, tc.to(**i32),
                                               len(s_cnt), kd.p, kd.blur, kd.scaling, kd.reach)
        img_of = torch.repeat_interleave(torch.arange(len(s_cnt)), sc.long()).to(dev)
        ctx.save_for_backward(gx, ga, img_of)
        return loss

    @staticmethod
    def backward(ctx, g):
        gx, ga, img_of = ctx.saved_tensors
        gi = g[img_of]
        return gi[:, None, None] * gx, gi[:, None] * ga, None, None, None, None, None


def kd_loss_2d(pred_xy, target_xy, pred_cls, target_cls, w, h, level, kd_loss, dim, pos_per_img=None,
               pos_per_img_t=None, normalize=True):
    if dim == 2 and normalize:
        # in place, like the reference (loss_libs.py:8-12): the caller's tensors are in frame units afterwards
        scale = pred_xy.new_tensor([float(w), float(h)])
        pred_xy.div_(scale)
        target_xy.div_(scale)
    pos_per_img = [int(n) for n in pos_per_img]
    pos_per_img_t = [int(n) for n in pos_per_img_t]
    xs = pred_xy.view(-1, 8, dim)
    yt = target_xy.view(-1, 8, dim)
    live = [i for i, (n, m) in enumerate(zip(pos_per_img, pos_per_img_t)) if n > 0 and m > 0]
    if not live:
        return []
    cap = ops.lib.kd6d_sinkhorn_max_points()
    if (isinstance(kd_loss, SamplesLoss) and level == "point" and dim == 2 and target_cls is not None and xs.is_cuda
            and max(pos_per_img) <= cap and max(pos_per_img_t) <= cap):
        per_img = _PackedSinkhornFn.apply(xs, pred_cls, yt, target_cls, pos_per_img, pos_per_img_t, kd_loss)
        return [per_img[i] for i in live]
    if level != "point":
        raise NotImplementedError("kd_loss_2d: only --glevel point exists (losses/loss_libs.py:39)")
    losses, s0, t0 = [], 0, 0
    for n, m in zip(pos_per_img, pos_per_img_t):
        if n > 0 and m > 0:
            x_s = xs[s0:s0 + n].transpose(0, 1).contiguous()           # (8, n, dim): one problem per keypoint
            y_t = yt[t0:t0 + m].transpose(0, 1).contiguous()
            if target_cls is not None:
                a_s = pred_cls[s0:s0 + n].transpose(0, 1).contiguous()
                b_t = target_cls[t0:t0 + m].transpose(0, 1).contiguous()
                losses.append(kd_loss(a_s, x_s, b_t, y_t).sum())
            else:
                losses.append(kd_loss(x_s, y_t).sum())
        s0, t0 = s0 + n, t0 + m
    return losses
