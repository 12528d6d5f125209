"""Validation loop of the evaluation path: libs/eval_libs.py:44-110 of the reference without its dataset side.

`valid()` runs the model in eval mode over a loader of (images, targets, meta_infos), keeps the most confident
pose per image (`new_p[0][:-1]`: score, class, R, T -- the 2D points are dropped as in the reference) and scores
the collection with `evaluate_pose_predictions`.  Not rebuilt here (the BOP reader): loading
meshes / 3D boxes from disk (`load_bop_meshes`, `load_bbox_3d`) and `remap_predictions` (re-solving the pose for
the original camera matrix of a resized frame) -- meshes are passed in, and predictions are scored in the
internal camera frame they were solved in.
"""
import numpy as np
import torch

from .distributed import get_rank
from .evaluate import evaluate_pose_predictions


class _Mesh:
    def __init__(self, vertices):
        self.vertices = np.asarray(vertices)


@torch.no_grad()
def valid(cfg, steps, loader, model, device, meshes, logger=None):
    """meshes: per class id an (n,3) vertex array (or an object with `.vertices`).  -> the 6-tuple of
    evaluate_pose_predictions on rank 0, None elsewhere."""
    was_training = model.training
    model.eval()
    preds = {}
    for images, targets, meta_infos in loader:
        if hasattr(images, "to"):
            images = images.to(device)
        pred, _ = model(images, targets=targets)
        for m, p in zip(meta_infos, pred):
            best = [list(p[0][:-1])] if len(p) else []
            preds[m["path"]] = {"meta": m, "pred": best}
    model.train(was_training)
    if get_rank() != 0:
        return None
    ms = [m if hasattr(m, "vertices") else _Mesh(m) for m in meshes]
    out = evaluate_pose_predictions(preds, cfg["DATASETS"]["N_CLASS"], ms, cfg["DATASETS"]["MESH_DIAMETERS"],
                                    cfg["DATASETS"].get("SYMMETRY_TYPES", {}))
    if logger is not None:            # eval_libs.py:112-146 of the reference: per class, then the mean over classes seen
        all_adi, all_rep, n_valid = {}, {}, 0
        for i, (adi, rep) in enumerate(zip(out[0], out[2])):
            if adi:
                logger.add_scalars("ADI/class_%02d" % i, adi, steps)
                logger.add_scalars("REP/class_%02d" % i, rep, steps)
                for k, v in adi.items():
                    all_adi[k] = all_adi.get(k, 0.0) + v
                for k, v in rep.items():
                    all_rep[k] = all_rep.get(k, 0.0) + v
                n_valid += 1
        if n_valid:
            logger.add_scalars("ADI/all_class", {k: v / n_valid for k, v in all_adi.items()}, steps)
            logger.add_scalars("REP/all_class", {k: v / n_valid for k, v in all_rep.items()}, steps)
    return out
