"""Evaluation-time pose inference: drop-in for postprocess/postprocess.py:12-202 of the reference.

The reference walks the levels in Python, filters candidates per label, picks the per-level top-n cells by the
same n_k rule as training (`positive_num * exp(-lambda * log2(size/S_k)^2)`), maps their keypoints back to the
full frame and calls cv2's EPnP-RANSAC per object.  Here the selection is ONE launch for the batch
(`kd6d_pose_candidates`: a workgroup per (image, ground-truth slot), logits read once into LDS), one device->host
copy brings the <= cap cells per object over, and the solver is kd6d/libs/pnp.py (parity unpinned at the cv2
boundary).  Returned per image, as in the reference: a list of `[score, class id, R (3,3), T (3,1), xy2d (n,8,2)]`
for every ground-truth class that produced a valid pose.
"""
import ctypes

import numpy as np
import torch

from .. import ops
from .._lib import MAX_GT, check, lib
from ..kd_losses import CAP, make_levels
from ..libs.evaluate import pose_symmetry_handling
from ..libs.pnp import solve_pnp_ransac


def pose_candidates(cls, reg, levels, batch, bbox_trans, class_ids, n_gt, th, positive_num, positive_lambda, cap=CAP):
    """-> cnt (batch*MAX_GT,) int32, kp (batch*MAX_GT*cap, 8, 2) full-frame px, score (batch*MAX_GT*cap, 8)."""
    dev = cls.device
    lv = make_levels(batch, levels)
    n = batch * MAX_GT * cap
    kp = torch.zeros(n, 8, 2, dtype=torch.float32, device=dev)
    score = torch.zeros(n, 8, dtype=torch.float32, device=dev)
    cnt = torch.zeros(batch * MAX_GT, dtype=torch.int32, device=dev)
    check(lib.kd6d_pose_candidates(ctypes.byref(lv), ops._ptr(cls), ops._ptr(reg), ops._ptr(bbox_trans),
                                   ops._ptr(class_ids), ops._ptr(n_gt), th, float(positive_num),
                                   float(positive_lambda), cap, ops._ptr(cnt), ops._ptr(kp), ops._ptr(score),
                                   ops._stream()), "kd6d_pose_candidates")
    return cnt, kp, score


class PostProcessor:
    def __init__(self, inference_th, positive_num, positive_lambda, sym_types=None, cap=CAP, reproj_err=5.0):
        self.inference_th = inference_th
        self.positive_num = positive_num
        self.positive_lambda = positive_lambda
        self.sym_types = sym_types or {}
        self.cap = cap
        self.reproj_err = reproj_err

    def forward(self, cls, reg, levels, batch, tgt):
        """cls (rows,16) / reg (rows,240) packed logits of the eval forward, tgt: PackedTargets of the batch."""
        cnt, kp, score = pose_candidates(cls, reg, levels, batch, tgt.bbox_trans, tgt.class_ids, tgt.n_gt,
                                         self.inference_th, self.positive_num, self.positive_lambda, self.cap)
        cnt = cnt.cpu().numpy().reshape(batch, MAX_GT)              # the one synchronising copy of the path
        kp = kp.cpu().numpy().reshape(batch, MAX_GT, self.cap, 8, 2)
        score = score.cpu().numpy().reshape(batch, MAX_GT, self.cap, 8)
        class_ids = tgt.class_ids.cpu().numpy().reshape(batch, MAX_GT)
        n_gt = tgt.n_gt.cpu().numpy()
        K = tgt.K.cpu().numpy()
        kp3d = tgt.kp3d.cpu().numpy()
        results = []
        for b in range(batch):
            found = {}
            for g in range(int(n_gt[b])):
                c, n = int(class_ids[b, g]), int(cnt[b, g])
                if n == 0 or c in found:
                    continue
                xy2d = kp[b, g, :n]                                              # (n, 8, 2)
                xyz = np.tile(kp3d[b, c], (n, 1))                                # (n*8, 3), keypoints_3d[clsId].repeat
                ok, R, T, _ = solve_pnp_ransac(xyz, xy2d.reshape(-1, 2), K[b], reproj_err=self.reproj_err)
                if not ok or np.isnan(R.sum()) or np.isnan(T.sum()):
                    continue
                key = "cls_" + str(c)
                if key in self.sym_types:
                    R = pose_symmetry_handling(R, self.sym_types[key])
                found[c] = [float(score[b, g, :n].max()), c, R, T, torch.from_numpy(xy2d.copy())]
            # the reference iterates torch.unique(labels): ascending class id
            results.append([found[c] for c in sorted(found)])
        return results

    __call__ = forward
