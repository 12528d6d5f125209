"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the Dynamic-Zoom-In front-end.

Restates, in numpy, what the reference does between the decoded frame and the network input:
  libs/transform.py:299-308   Normalize: BGR->RGB, /255, -mean, /std (float64, then .float())
  libs/dzi_libs.py:14-53      aug_bbox_DZI: jittered square box (centre, scale) from the object box
  libs/dzi_libs.py:142-210    crop_resize_by_warp_affine / get_affine_transform: 3-point affine (rot = 0)
  libs/dzi_libs.py:55-95      dzi_train: cv2.warpAffine INTER_LINEAR on the float32 image, INTER_NEAREST on the mask

PARITY UNPINNED at the cv2 boundary: OpenCV is not installed in this image and the reference holds no
fixture of a warped crop, so `warp_affine` below restates cv2.warpAffine's published fixed-point scheme
(imgwarp.cpp, WarpAffineInvoker + remapBilinear): inverse matrix in double, source coordinates in 1/1024 px
(AB_BITS = 10) rounded to 1/32 px (INTER_BITS = 5, round_delta 16) for INTER_LINEAR or to whole pixels
(round_delta 512) for INTER_NEAREST, float32 interpolation weights from the 32x32 table, constant-0 border.
Pinned only by its own known answers (identity, integer shifts, borders) in tests/test_dzi_oracle.py.
"""
import numpy as np

AB_BITS, INTER_BITS = 10, 5
AB_SCALE, INTER_TAB = 1 << AB_BITS, 1 << INTER_BITS


def normalize_lut(mean, std):
    """lut[c][v] = float32((v/255 - mean[c]) / std[c]) evaluated in float64 like transform.py:303-307 (RGB order)."""
    v = np.arange(256, dtype=np.float64)[None, :] / 255.0
    return ((v - np.asarray(mean, np.float64)[:, None]) / np.asarray(std, np.float64)[:, None]).astype(np.float32)


def aug_bbox_dzi(bbox_xyxy, im_h, im_w, rng, scale_ratio=0.25, shift_ratio=0.25, pad_scale=1.5, train=True):
    """dzi_libs.py:14-53 ('uniform') / :104-108 (test).  rng: numpy RandomState-like with random_sample()."""
    x1, y1, x2, y2 = [float(v) for v in bbox_xyxy]
    cx, cy, bw, bh = 0.5 * (x1 + x2), 0.5 * (y1 + y2), x2 - x1, y2 - y1
    if train:
        sr = 1 + scale_ratio * (2 * rng.random_sample() - 1)
        sh = shift_ratio * (2 * rng.random_sample(2) - 1)
        center = np.array([cx + bw * sh[0], cy + bh * sh[1]])
        scale = max(bh, bw) * sr * pad_scale
    else:
        center = np.array([cx, cy])
        scale = max(max(bh, 1), max(bw, 1)) * pad_scale
    return center, min(scale, max(im_h, im_w)) * 1.0


def affine_from_box(center, scale, out_res):
    """get_affine_transform(center, scale, rot=0, out_res): the three float32 point pairs, solved in float64."""
    c = np.asarray(center, np.float32)
    sw = np.float32(scale)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = c
    src[1] = c + np.array([0.0, sw * np.float32(-0.5)], np.float32)
    dst[0] = [out_res * 0.5, out_res * 0.5]
    dst[1] = np.array([out_res * 0.5, out_res * 0.5], np.float32) + np.array([0, out_res * -0.5], np.float32)
    for p in (src, dst):
        d = p[0] - p[1]
        p[2] = p[1] + np.array([-d[1], d[0]], np.float32)
    A = np.concatenate([src.astype(np.float64), np.ones((3, 1))], 1)          # [x y 1] M^T = dst
    return np.linalg.solve(A, dst.astype(np.float64)).T                         # (2,3) float64


def _round_int(v):
    return np.rint(v).astype(np.int64)          # saturate_cast<int>(double): round half to even


def warp_affine(img, M, out_res, nearest=False):
    """cv2.warpAffine(img, M, (out_res, out_res), flags=INTER_NEAREST|INTER_LINEAR), float32 HxW[xC] image."""
    img = np.asarray(img, np.float32)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W, C = img.shape
    M = np.asarray(M, np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    m00, m01, m10, m11 = A11, -M[0, 1] * D, -M[1, 0] * D, A22
    b1 = -m00 * M[0, 2] - m01 * M[1, 2]
    b2 = -m10 * M[0, 2] - m11 * M[1, 2]
    xs = np.arange(out_res, dtype=np.float64)
    adelta = _round_int(m00 * xs * AB_SCALE)
    bdelta = _round_int(m10 * xs * AB_SCALE)
    rd = AB_SCALE // 2 if nearest else AB_SCALE // INTER_TAB // 2
    out = np.zeros((out_res, out_res, C), np.float32)
    tab = (np.arange(INTER_TAB, dtype=np.float32) / np.float32(INTER_TAB))
    for y in range(out_res):
        X0 = _round_int((m01 * y + b1) * AB_SCALE) + rd
        Y0 = _round_int((m11 * y + b2) * AB_SCALE) + rd
        if nearest:
            X = (X0 + adelta) >> AB_BITS
            Y = (Y0 + bdelta) >> AB_BITS
            ok = (X >= 0) & (X < W) & (Y >= 0) & (Y < H)
            out[y][ok] = img[Y[ok], X[ok]]
            continue
        X = (X0 + adelta) >> (AB_BITS - INTER_BITS)
        Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS)
        sx, sy = X >> INTER_BITS, Y >> INTER_BITS
        fx, fy = tab[X & (INTER_TAB - 1)], tab[Y & (INTER_TAB - 1)]
        w = [(np.float32(1) - fy) * (np.float32(1) - fx), (np.float32(1) - fy) * fx, fy * (np.float32(1) - fx), fy * fx]
        acc = np.zeros((out_res, C), np.float32)
        for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            yy, xx = sy + dy, sx + dx
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            px = np.zeros((out_res, C), np.float32)
            px[ok] = img[yy[ok], xx[ok]]
            acc = acc + px * w[k][:, None]
        out[y] = acc
    return out[:, :, 0] if squeeze else out


def dzi_crop(frame_bgr_u8, mask, center, scale, mean, std, out_res=256):
    """Frame (H,W,3) uint8 BGR + mask (H,W) -> image (3,out,out) float32 normalised RGB, mask (out,out),
    bbox_trans (2,3) float32, bbox_scale float32  (= Normalize + ToTensor + dzi_train without the box jitter)."""
    lut = normalize_lut(mean, std)
    rgb = frame_bgr_u8[:, :, ::-1]
    img = np.stack([lut[c][rgb[:, :, c]] for c in range(3)], -1)               # (H,W,3) float32
    M = affine_from_box(center, scale, out_res)
    roi = warp_affine(img, M, out_res, nearest=False)
    roi_mask = warp_affine(np.asarray(mask, np.float32), M, out_res, nearest=True)
    return roi.transpose(2, 0, 1).copy(), roi_mask, M.astype(np.float32), np.float32(out_res / scale)
