"""Validation loop of the evaluation path: libs/eval_libs.py:44-110 of the reference without its dataset side.

`valid()` runs the model in eval mode over a loader of (images, targets, meta_infos), keeps the most confident
pose per image (`new_p[0][:-1]`: score, class, R, T -- the 2D points are dropped as in the reference) and scores
the collection with `evaluate_pose_predictions`.  As in eval_libs.py:69-76 every prediction is first re-solved for the
frame's own camera (`remap_predictions`: identity when the frame carries the internal camera, which is what the
synthetic loaders and frames stored at the internal resolution do).  Meshes are passed in by the caller
(kd6d.libs.train_libs.dataset_meshes / the synthetic loaders).
"""
import numpy as np
import torch

from .distributed import get_rank
from .evaluate import evaluate_pose_predictions, remap_predictions


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
            if len(p) and "K" in m and cfg.get("INPUT", {}).get("INTERNAL_K") is not None and \
                    not np.allclose(np.asarray(m["K"], np.float64).reshape(3, 3),
                                    np.asarray(cfg["INPUT"]["INTERNAL_K"], np.float64).reshape(3, 3)):
                kp3d = targets.kp3d[0].cpu().numpy() if hasattr(targets, "kp3d") else None     # (n_class, 8, 3) box corners
                if kp3d is not None:            # eval_libs.py:71-76: poses solved for the internal camera -> the frame's own
                    p = remap_predictions(cfg["INPUT"]["INTERNAL_K"], cfg["INPUT"].get("INTERNAL_WIDTH"),
                                          cfg["INPUT"].get("INTERNAL_HEIGHT"), kp3d, m, p)
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
