"""CPU: known answers of the DZI oracle (oracle/dzi_ref.py).  cv2 is absent, so the fixed-point warp restatement
is pinned by what any correct warp must do: identity, integer shifts, exact down-scaling taps, zero border,
and agreement with plain float bilinear interpolation up to the 1/32-pixel coordinate quantisation."""
import numpy as np
import pytest

from oracle import dzi_ref as Z

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _frame(h, w, seed=0):
    r = np.random.default_rng(seed)
    return r.integers(0, 256, (h, w, 3), dtype=np.uint8), (r.random((h, w)) > 0.6).astype(np.float32)


def test_normalize_lut_matches_float64_normalize():
    lut = Z.normalize_lut(MEAN, STD)
    v = np.array([0, 1, 127, 255])
    for c in range(3):
        want = ((v / 255.0) - MEAN[c]) / STD[c]
        assert np.array_equal(lut[c][v], want.astype(np.float32))


def test_identity_crop_reproduces_the_normalised_frame():
    fr, mk = _frame(64, 64)
    img, m, M, s = Z.dzi_crop(fr, mk, (32.0, 32.0), 64.0, MEAN, STD, out_res=64)
    np.testing.assert_allclose(M, [[1, 0, 0], [0, 1, 0]], atol=1e-6)
    lut = Z.normalize_lut(MEAN, STD)
    want = np.stack([lut[c][fr[:, :, 2 - c]] for c in range(3)])
    assert np.array_equal(img, want) and np.array_equal(m, mk) and s == 1.0


def test_integer_shift_and_zero_border():
    fr, mk = _frame(48, 80, 1)
    # box of side 32 centred at (50, 20) on an 80x48 frame, output 32: pure shift by (-34, -4)
    img, m, M, _ = Z.dzi_crop(fr, mk, (50.0, 20.0), 32.0, MEAN, STD, out_res=32)
    lut = Z.normalize_lut(MEAN, STD)
    full = np.stack([lut[c][fr[:, :, 2 - c]] for c in range(3)])
    assert np.array_equal(img, full[:, 4:36, 34:66]) and np.array_equal(m, mk[4:36, 34:66])
    # a box hanging over the frame edge: outside taps contribute 0 (constant border)
    img2, m2, _, _ = Z.dzi_crop(fr, mk, (70.0, 40.0), 32.0, MEAN, STD, out_res=32)
    assert np.array_equal(img2[:, :24, :26], full[:, 24:48, 54:80]) and np.all(img2[:, 24:, :] == 0) and np.all(img2[:, :, 26:] == 0)
    assert np.all(m2[24:, :] == 0) and np.array_equal(m2[:24, :26], mk[24:48, 54:80])


def test_bilinear_matches_float_interpolation_within_quantisation():
    fr, mk = _frame(60, 90, 2)
    c, sc, R = (41.3, 29.7), 47.0, 64
    img, m, M, s = Z.dzi_crop(fr, mk, c, sc, MEAN, STD, out_res=R)
    assert s == pytest.approx(R / sc)
    lut = Z.normalize_lut(MEAN, STD)
    full = np.stack([lut[ch][fr[:, :, 2 - ch]] for ch in range(3)]).astype(np.float64)
    # plain double-precision bilinear at the exact inverse-mapped coordinates
    ys, xs = np.mgrid[0:R, 0:R].astype(np.float64)
    sx = (xs - M[0, 2]) / M[0, 0]; sy = (ys - M[1, 2]) / M[1, 1]
    x0 = np.floor(sx).astype(int); y0 = np.floor(sy).astype(int); fx = sx - x0; fy = sy - y0
    def tap(yy, xx):
        ok = (xx >= 0) & (xx < 90) & (yy >= 0) & (yy < 60)
        out = np.zeros((3, R, R)); out[:, ok] = full[:, yy[ok], xx[ok]]; return out
    want = tap(y0, x0) * (1 - fy) * (1 - fx) + tap(y0, x0 + 1) * (1 - fy) * fx + tap(y0 + 1, x0) * fy * (1 - fx) + tap(y0 + 1, x0 + 1) * fy * fx
    # coordinates are quantised to 1/32 px: error <= (1/64) * |gradient| per axis; pixel values span ~ +-2.6
    assert np.abs(img - want).max() <= 2 * (1 / 64) * 2 * 5.3 + 1e-5
    assert np.abs(img - want).mean() <= 0.03


def test_box_jitter_ranges():
    rng = np.random.RandomState(0)
    for _ in range(50):
        c, s = Z.aug_bbox_dzi([100, 80, 180, 140], 480, 640, rng)
        assert 140 - 20 <= c[0] <= 140 + 20 and 110 - 15 <= c[1] <= 110 + 15
        assert 80 * 0.75 * 1.5 <= s <= 80 * 1.25 * 1.5
    c, s = Z.aug_bbox_dzi([100, 80, 180, 140], 480, 640, rng, train=False)
    assert tuple(c) == (140.0, 110.0) and s == 120.0
