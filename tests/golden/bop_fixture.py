"""A tiny BOP-format tree written on the fly (shared by tests/test_dataset.py and tests/golden/make_golden_bop.py)."""
import json
import os
import struct

import numpy as np


def write_tree(root):
    from PIL import Image
    rng = np.random.default_rng(4)
    scene = os.path.join(root, "train", "000001")
    os.makedirs(os.path.join(scene, "rgb")); os.makedirs(os.path.join(scene, "mask_visib"))
    models = os.path.join(root, "models") + "/"
    os.makedirs(models)
    H, W = 20, 24
    rgb3 = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    Image.fromarray(rgb3, "RGB").save(os.path.join(scene, "rgb", "000003.png"))
    grey16 = rng.integers(0, 65536, (H, W), dtype=np.uint16)
    Image.fromarray(grey16).save(os.path.join(scene, "rgb", "000007.png"))
    rgba = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    rgba[:, :, 3] = np.where(rng.random((H, W)) < 0.3, 0, 255)
    Image.fromarray(rgba, "RGBA").save(os.path.join(scene, "rgb", "000009.png"))
    cam, gt = {}, {}
    for im_id, objs in ((3, [1, 9, 5]), (7, [5]), (9, [9])):
        cam[str(im_id)] = {"cam_K": [572.4, 0, 12.0 + im_id, 0, 573.6, 10.0, 0, 0, 1], "depth_scale": 1.0}
        gt[str(im_id)] = []
        for k, oid in enumerate(objs):
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            gt[str(im_id)].append({"cam_R_m2c": q.reshape(-1).tolist(), "cam_t_m2c": [float(rng.normal(0, 30)), float(rng.normal(0, 30)), 800.0 + k],
                                   "obj_id": oid})
            m = np.zeros((H, W), np.uint8)
            m[2 + 3 * k:9 + 3 * k, 1 + 4 * k:12 + 4 * k] = 255
            Image.fromarray(m, "L").save(os.path.join(scene, "mask_visib", "%06d_%06d.png" % (im_id, k)))
    json.dump(cam, open(os.path.join(scene, "scene_camera.json"), "w"))
    json.dump(gt, open(os.path.join(scene, "scene_gt.json"), "w"))
    v1 = rng.normal(0, 40, (17, 3))
    with open(models + "obj_000001.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment test\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "element face 1\nproperty list uchar int vertex_indices\nend_header\n" % len(v1))
        for r in v1:
            f.write("%.6f %.6f %.6f\n" % tuple(r))
        f.write("3 0 1 2\n")
    v5 = rng.normal(0, 25, (9, 3)).astype(np.float32)
    with open(models + "obj_000005.ply", "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "property float nx\nproperty float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\n"
                 "property uchar blue\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n" % len(v5)).encode())
        for r in v5:
            f.write(struct.pack("<6f3B", r[0], r[1], r[2], 0.0, 0.0, 1.0, 10, 20, 30))
        f.write(struct.pack("<B3i", 3, 0, 1, 2))
    boxes = []
    for v in (v1, v5.astype(np.float64)):
        lo, hi = v.min(0), v.max(0)
        boxes.append([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    json.dump(boxes, open(os.path.join(root, "bbox.json"), "w"))
    lst = os.path.join(root, "train", "list.txt")
    with open(lst, "w") as f:
        f.write("000001/rgb/000003.png\n000001/rgb/000007.png\n" + os.path.join(scene, "rgb", "000009.png") + "\n")
    return dict(list_file=lst, models=models, bbox=os.path.join(root, "bbox.json"), scene=scene, rgb3=rgb3, grey16=grey16,
                rgba=rgba, v1=v1, v5=v5.astype(np.float64), gt=gt, cam=cam, H=H, W=W)
