"""BOP / LINEMOD reader: libs/dataset.py:27-183 and libs/utils.py:43-61,238-301 of the
reference without OpenCV and trimesh.

What it covers: the image list file, `scene_camera.json` / `scene_gt.json` / `mask_visib/<img>_<k>.png`
annotations of a BOP scene (`get_single_bop_annotation`), the object-id -> class-id table and mesh vertices from
`obj_XXXXXX.ply` (`load_bop_meshes`; ASCII and binary_little_endian PLY, vertices only), the 3D-box json
(`load_bbox_3d`), and the image normalisations of `getitem_dzi` (16-bit -> 8-bit, grey -> 3 channels, alpha ->
white background).  Frames come back as BGR uint8 (the reference's cv2 order) so that the GPU front-end
(`kd6d/libs/dzi_libs.dzi_batch`, csrc/dzi.hip) can crop + normalise them; `collate_frames` stacks a batch of
equal-size frames and derives the DZI boxes from the projected 3D boxes like `dzi_libs.py:142-210`.

Not rebuilt: the CPU augmentation pipeline of libs/transform.py (resize / colour jitter / occlusion with cv2 and
imgaug) -- frames must already have the internal resolution -- and `remap_predictions`.
Decoding uses Pillow; **parity unpinned** against cv2.imread for exotic PNG variants (checked: 8-bit RGB / RGBA /
grey and 16-bit grey, the formats BOP ships).
"""
import json
import os
import random

import numpy as np
import torch

from .poses import PoseAnnot


def load_json_cached(path, mem_cache=None):
    if mem_cache is not None and path in mem_cache:
        return mem_cache[path]
    with open(path, "r") as f:
        data = json.load(f)
    if mem_cache is not None:
        mem_cache[path] = data
    return data


def load_image_cached(path, mem_cache=None):
    """-> numpy array like cv2.imread(path, IMREAD_UNCHANGED): (H,W) grey, (H,W,3) BGR or (H,W,4) BGRA; None if unreadable."""
    if mem_cache is not None and path in mem_cache:
        return mem_cache[path]
    try:
        from PIL import Image
        with Image.open(path) as im:
            if im.mode in ("I;16", "I;16B", "I"):
                arr = np.asarray(im).astype(np.uint16)
            elif im.mode == "L":
                arr = np.asarray(im)
            elif im.mode == "RGBA":
                arr = np.asarray(im)[:, :, [2, 1, 0, 3]]
            else:
                arr = np.asarray(im.convert("RGB"))[:, :, ::-1]
        arr = np.ascontiguousarray(arr)
    except Exception:
        return None
    if mem_cache is not None:
        mem_cache[path] = arr
    return arr


def load_ply_vertices(path):
    """Vertex positions (n,3) float64 of a PLY file (ascii or binary_little_endian; x,y,z must be the first three
    vertex properties, as in the BOP models)."""
    sizes = {"char": 1, "uchar": 1, "int8": 1, "uint8": 1, "short": 2, "ushort": 2, "int16": 2, "uint16": 2,
             "int": 4, "uint": 4, "int32": 4, "uint32": 4, "float": 4, "float32": 4, "double": 8, "float64": 8}
    codes = {"char": "i1", "uchar": "u1", "int8": "i1", "uint8": "u1", "short": "<i2", "ushort": "<u2", "int16": "<i2",
             "uint16": "<u2", "int": "<i4", "uint": "<u4", "int32": "<i4", "uint32": "<u4", "float": "<f4",
             "float32": "<f4", "double": "<f8", "float64": "<f8"}
    with open(path, "rb") as f:
        fmt, n_vert, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: no end_header" % path)
            tok = line.decode("ascii", "replace").strip().split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vert = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("%s: list property inside the vertex element" % path)
                props.append((tok[2], tok[1]))
            elif tok[0] == "end_header":
                break
        if [p[0] for p in props[:3]] != ["x", "y", "z"]:
            raise ValueError("%s: vertex properties must start with x y z" % path)
        if fmt == "ascii":
            rows = [f.readline().split()[:3] for _ in range(n_vert)]
            return np.asarray(rows, np.float64)
        if fmt != "binary_little_endian":
            raise ValueError("%s: unsupported PLY format %s" % (path, fmt))
        dt = np.dtype([(name, codes[t]) for name, t in props])
        assert dt.itemsize == sum(sizes[t] for _, t in props)
        data = np.frombuffer(f.read(n_vert * dt.itemsize), dtype=dt, count=n_vert)
        return np.stack([data["x"], data["y"], data["z"]], 1).astype(np.float64)


class Mesh:
    def __init__(self, vertices):
        self.vertices = vertices


def load_bop_meshes(model_path):
    """-> (meshes sorted by file name, {str(obj_id): class id}); obj id = the digits of obj_XXXXXX.ply."""
    files = sorted(f for f in os.listdir(model_path) if f.endswith(".ply"))
    meshes, table = [], {}
    for i, name in enumerate(files):
        table[str(int(os.path.splitext(name)[0][4:]))] = i
        meshes.append(Mesh(load_ply_vertices(os.path.join(model_path, name))))
    return meshes, table


def load_bbox_3d(json_file):
    with open(json_file, "r") as f:
        return json.load(f)


def get_single_bop_annotation(img_path, objID_2_clsID, mem_cache=None):
    """-> K (3,3), merged instance mask (H,W) uint8 (0 background, i+1 = i-th kept object), class ids, rotations,
    translations.  Objects whose id is not in the table are skipped (and leave no mask)."""
    img_path = img_path.strip()
    gt_dir, sub, name = img_path.rsplit("/", 2)
    assert sub == "rgb", "BOP layout: <scene>/rgb/<image>"
    base, _ = os.path.splitext(name)
    cam = load_json_cached(gt_dir + "/scene_camera.json", mem_cache)
    gt = load_json_cached(gt_dir + "/scene_gt.json", mem_cache)
    key = str(int(base))
    cam_a = cam[key] if key in cam else cam[base]
    poses = gt[key] if key in gt else gt[base]
    K = np.array(cam_a["cam_K"]).reshape(3, 3)
    class_ids, rotations, translations, merged = [], [], [], None
    inst = 1
    for i, p in enumerate(poses):
        mask = load_image_cached(gt_dir + "/mask_visib/" + ("%s_%06d.png" % (base, i)), mem_cache)
        if merged is None:
            merged = np.zeros(mask.shape[:2], np.uint8)
        obj_id = str(p["obj_id"])
        if obj_id not in objID_2_clsID:
            continue
        class_ids.append(objID_2_clsID[obj_id])
        rotations.append(np.array(p["cam_R_m2c"]).reshape(3, 3))
        translations.append(np.array(p["cam_t_m2c"]).reshape(3, 1))
        merged[mask == 255] = inst
        inst += 1
    return K, merged, class_ids, rotations, translations


def normalise_frame(img):
    """The image clean-up of BOP_Dataset.getitem_dzi (dataset.py:131-144): 16-bit -> 8-bit (scale 255/65535,
    rounded, saturating like cv2.convertScaleAbs), grey -> 3 channels, alpha == 0 -> white background."""
    if img.dtype == np.uint16:
        img = np.clip(np.rint(img.astype(np.float64) * (255.0 / 65535.0)), 0, 255).astype(np.uint8)
    if img.ndim == 2:
        img = np.repeat(img.reshape(img.shape[0], img.shape[1], 1), 3, axis=2)
    elif img.shape[2] == 4:
        img = img.copy()
        img[:, :, 0:3][img[:, :, 3] == 0] = 255
    return img


class BOP_Dataset(torch.utils.data.Dataset):
    """Items: (frame BGR uint8 (H,W,3) tensor, PoseAnnot on the full frame, meta_info) -- the raw sample of
    dataset.py:71-103 before the transform; the crop + normalisation happen on the GPU (`collate_frames` +
    `dzi_libs.dzi_batch`)."""

    def __init__(self, image_list_file, mesh_dir, bbox_json, symmetry_types=None, training=True, mem_cache=None):
        data_dir = os.path.split(image_list_file)[0]
        with open(image_list_file, "r") as f:
            files = [ln.strip() for ln in f.readlines() if ln.strip()]
        self.img_files = [p if p.startswith("/") else data_dir + "/" + p for p in files]
        if training:
            random.shuffle(self.img_files)
        self.meshes, self.objID_2_clsID = load_bop_meshes(mesh_dir)
        self.bbox_3d = torch.tensor(load_bbox_3d(bbox_json), dtype=torch.float32)      # (n_class, 8, 3)
        self.symmetry_types = symmetry_types
        self.training = training
        self.cache = mem_cache

    def __len__(self):
        return len(self.img_files)

    def __getitem__(self, index):
        item = self.getitem1(index)
        while item is None:
            item = self.getitem1(random.randint(0, len(self.img_files) - 1))
        return item

    def getitem1(self, index):
        path = self.img_files[index]
        img = load_image_cached(path, self.cache)
        if img is None:
            print("image %s not found" % path)
            return None
        img = normalise_frame(img)
        h, w = img.shape[:2]
        K, mask, class_ids, rotations, translations = get_single_bop_annotation(path, self.objID_2_clsID, self.cache)
        if self.training and len(class_ids) == 0:
            return None
        meta = {"path": path, "K": K, "width": w, "height": h, "class_ids": class_ids, "rotations": rotations,
                "translations": translations}
        target = PoseAnnot(self.bbox_3d, torch.tensor(K, dtype=torch.float32), torch.from_numpy(mask.astype(np.float32)),
                           torch.tensor(class_ids, dtype=torch.long),
                           torch.tensor(np.asarray(rotations, np.float32).reshape(-1, 3, 3)),
                           torch.tensor(np.asarray(translations, np.float32).reshape(-1, 3, 1)), w, h)
        return torch.from_numpy(np.ascontiguousarray(img[:, :, :3])), target, meta


def projected_box(target, g=0):
    """xyxy box of the projected 3D-box corners of instance g (the box DZI jitters, dzi_libs.py:142-155)."""
    c = int(target.class_ids[g])
    X = target.keypoints_3d[c].numpy().astype(np.float64)
    cam = target.rotations[g].numpy().astype(np.float64) @ X.T + target.translations[g].numpy().reshape(3, 1)
    uv = target.K.numpy().astype(np.float64) @ cam
    u, v = uv[0] / uv[2], uv[1] / uv[2]
    return np.array([u.min(), v.min(), u.max(), v.max()])


def collate_frames(batch):
    """list of dataset items (equal frame size) -> (frames (B,H,W,3) uint8, masks (B,H,W) float32, targets, metas)."""
    frames = torch.stack([b[0] for b in batch]).contiguous()
    masks = torch.stack([b[1].mask for b in batch]).contiguous()
    return frames, masks, [b[1] for b in batch], [b[2] for b in batch]
