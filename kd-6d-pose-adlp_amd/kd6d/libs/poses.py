"""Batch containers crossing the hot-path boundary.

Mirrors the *type* surface the reference step consumes: `PoseAnnot` (libs/poses.py:21-37,
236-259: fields + .to(device)) and `ImageList` (libs/dataset.py:185-194).  Only what
train_kd.py:104-110 and losses/loss.py:164-268 touch is kept; the cv2-based transforms and
drawing helpers of the reference are data-pipeline code and out of scope.
"""
import torch


class PoseAnnot(object):
    """6D pose annotations of one (cropped) image.

    keypoints_3d (n_class,8,3) 3D-bbox corners of every class, K (3,3), mask (H,W) float
    instance ids (0 background, i+1 instance i), class_ids (G,) int64, rotations (G,3,3),
    translations (G,3,1), width/height of the crop, bbox_scale (), bbox_trans (2,3) affine
    full-frame -> crop.
    """

    def __init__(self, bbox_3d, K, mask, class_ids, rotations, translations, width, height,
                 bbox_scale=None, bbox_trans=None):
        self.keypoints_3d = bbox_3d
        self.K = K
        self.mask = mask
        self.class_ids = class_ids
        self.rotations = rotations
        self.translations = translations
        self.width = width
        self.height = height
        self.bbox_scale = bbox_scale
        self.bbox_trans = bbox_trans

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return PoseAnnot(mv(self.keypoints_3d), mv(self.K), mv(self.mask), mv(self.class_ids),
                         mv(self.rotations), mv(self.translations), self.width, self.height,
                         mv(self.bbox_scale), mv(self.bbox_trans))

    def __len__(self):
        return len(self.class_ids)

    def as_dict(self):
        return dict(keypoints_3d=self.keypoints_3d, K=self.K, mask=self.mask, class_ids=self.class_ids,
                    rotations=self.rotations, translations=self.translations, width=self.width,
                    height=self.height, bbox_scale=self.bbox_scale, bbox_trans=self.bbox_trans)


class ImageList:
    def __init__(self, tensors, sizes):
        self.tensors = tensors
        self.sizes = sizes

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.sizes)
