"""Seeded LINEMOD-shaped synthetic batches (no dataset, no network): SURVEY.md 8(d).

Geometry follows the reference's data path: frames live in a 640x480 full frame with the
intrinsics of configs/ape.yaml:20; the network sees a 256x256 Dynamic-Zoom-In crop
(libs/dzi_libs.py:12,55-95) described by the 2x3 affine `bbox_trans` (full frame -> crop).
"""
import math

import numpy as np
import torch

from .libs.poses import ImageList, PoseAnnot

INTERNAL_K = [572.4114, 0, 325.2611, 0, 573.57043, 242.04899, 0, 0, 1]
MESH_DIAMETERS = [104.26, 250.85, 167.49, 177.43, 204.83, 154.63, 129.85, 264.12, 110.83, 164.65, 178.35,
                  145.61, 279.04, 287.24, 213.25]
LINEMOD_CLASSES = [0, 1, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14]


def cube_keypoints(diameters=MESH_DIAMETERS):
    """(n_class, 8, 3): corners of the cube whose diagonal is the mesh diameter."""
    signs = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float32)
    e = np.asarray(diameters, np.float32) / (2.0 * math.sqrt(3.0))
    return signs[None] * e[:, None, None]


def make_batch(batch, seed, crop=256, mixed_classes=False, full_frame=False, class_id=0, class_offset=0):
    """Returns (ImageList on CPU, list[PoseAnnot] on CPU).  mixed_classes: image i shows LINEMOD class
    (class_offset + i) mod 13 (a rank passes its first global image index as class_offset)."""
    rng = np.random.default_rng(seed)
    K = np.asarray(INTERNAL_K, np.float32).reshape(3, 3)
    kp3d = cube_keypoints()
    H, W = (480, 640) if full_frame else (crop, crop)
    imgs = rng.standard_normal((batch, 3, H, W), dtype=np.float32)
    targets = []
    for i in range(batch):
        c = LINEMOD_CLASSES[(class_offset + i) % len(LINEMOD_CLASSES)] if mixed_classes else class_id
        q, r = np.linalg.qr(rng.standard_normal((3, 3)))
        q = q * np.sign(np.diag(r))[None, :]
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        R = q.astype(np.float32)
        T = np.array([rng.normal(0, 60), rng.normal(0, 40), 900 + rng.normal(0, 80)], np.float32).reshape(3, 1)
        cam = R @ kp3d[c].T + T
        uv = K @ cam
        u, v = uv[0] / uv[2], uv[1] / uv[2]
        ext = max(u.max() - u.min(), v.max() - v.min())
        cx, cy = 0.5 * (u.max() + u.min()), 0.5 * (v.max() + v.min())
        if full_frame:
            s, tx, ty = 1.0, 0.0, 0.0
        else:
            s = crop / (1.5 * ext)                       # DZI box = 1.5 x max extent (dzi_libs.py:107)
            tx, ty = crop / 2.0 - s * cx, crop / 2.0 - s * cy
        bbox_trans = np.array([[s, 0, tx], [0, s, ty]], np.float32)
        mask = np.zeros((H, W), np.float32)
        hw = s * (u.max() - u.min()) / 3.0                # rectangle 2/3 of the projected extent
        hh = s * (v.max() - v.min()) / 3.0
        mx, my = s * cx + tx, s * cy + ty
        x0, x1 = int(max(0, round(mx - hw))), int(min(W, round(mx + hw)))
        y0, y1 = int(max(0, round(my - hh))), int(min(H, round(my + hh)))
        mask[y0:y1, x0:x1] = 1.0
        targets.append(PoseAnnot(torch.from_numpy(kp3d.copy()), torch.from_numpy(K.copy()), torch.from_numpy(mask),
                                 torch.tensor([c], dtype=torch.long), torch.from_numpy(R[None].copy()),
                                 torch.from_numpy(T[None].copy()), W, H,
                                 torch.tensor(float(s)), torch.from_numpy(bbox_trans)))
    return ImageList(torch.from_numpy(imgs), [(H, W)] * batch), targets
