"""Dynamic Zoom-In on the GPU -- the reference's libs/dzi_libs.py + the Normalize/ToTensor transforms, batched.

`aug_bbox_DZI` keeps the reference's name and arithmetic (three random numbers per image, host side);
`dzi_batch` runs Normalize + the affine crop of image and mask for a whole batch in one kd6d_dzi_crop launch and
returns what `dzi_train` / `dzi_test` attach to the targets (mask, bbox_trans, bbox_scale, 256x256 image).
"""
import numpy as np
import torch

from .. import ops
from ..ops import check, lib

_DZI_PAD_SCALE = 1.5
_DZI_SCALE_RATIO = 0.25
_DZI_SHIFT_RATIO = 0.25
_INPUT_RES = 256


def aug_bbox_DZI(bbox_xyxy, im_H, im_W, rng=np.random):
    """dzi_libs.py:14-53, 'uniform' type: jittered square box -> (center, scale)."""
    x1, y1, x2, y2 = [float(v) for v in bbox_xyxy]
    cx, cy, bw, bh = 0.5 * (x1 + x2), 0.5 * (y1 + y2), x2 - x1, y2 - y1
    scale_ratio = 1 + _DZI_SCALE_RATIO * (2 * rng.random_sample() - 1)
    shift_ratio = _DZI_SHIFT_RATIO * (2 * rng.random_sample(2) - 1)
    center = np.array([cx + bw * shift_ratio[0], cy + bh * shift_ratio[1]])
    scale = max(y2 - y1, x2 - x1) * scale_ratio * _DZI_PAD_SCALE
    return center, min(scale, max(im_H, im_W)) * 1.0


def test_bbox_DZI(bbox_xyxy, im_H, im_W):
    """dzi_libs.py:98-108 (dzi_test): centred box, no jitter."""
    x1, y1, x2, y2 = [float(v) for v in bbox_xyxy]
    center = np.array([0.5 * (x1 + x2), 0.5 * (y1 + y2)])
    scale = max(max(y2 - y1, 1), max(x2 - x1, 1)) * _DZI_PAD_SCALE
    return center, min(scale, max(im_H, im_W)) * 1.0


def normalize_lut(mean, std, device):
    v = np.arange(256, dtype=np.float64)[None, :] / 255.0
    lut = ((v - np.asarray(mean, np.float64)[:, None]) / np.asarray(std, np.float64)[:, None]).astype(np.float32)
    return torch.from_numpy(lut).to(device).contiguous()


def dzi_batch(frames_bgr, masks, centers, scales, lut, input_res=_INPUT_RES):
    """frames_bgr (B,H,W,3) uint8 device tensor, masks (B,H,W) float32 or None, centers (B,2) / scales (B,)
    host arrays -> images (B,3,R,R) fp32, masks (B,R,R), bbox_trans (B,2,3), bbox_scale (B)."""
    assert frames_bgr.dtype == torch.uint8 and frames_bgr.is_cuda and frames_bgr.is_contiguous()
    B, H, W, C = frames_bgr.shape
    assert C == 3
    dev = frames_bgr.device
    cs = torch.tensor(np.concatenate([np.asarray(centers, np.float32).reshape(B, 2),
                                      np.asarray(scales, np.float32).reshape(B, 1)], 1), device=dev)
    images = torch.empty(B, 3, input_res, input_res, dtype=torch.float32, device=dev)
    masks_out = torch.empty(B, input_res, input_res, dtype=torch.float32, device=dev) if masks is not None else None
    trans = torch.empty(B, 2, 3, dtype=torch.float32, device=dev)
    bscale = torch.empty(B, dtype=torch.float32, device=dev)
    check(lib.kd6d_dzi_crop(ops._ptr(frames_bgr), ops._ptr(masks), B, H, W, ops._ptr(cs), ops._ptr(lut), input_res,
                            ops._ptr(images), ops._ptr(masks_out), ops._ptr(trans), ops._ptr(bscale), ops._stream()),
          "kd6d_dzi_crop")
    return images, masks_out, trans, bscale
