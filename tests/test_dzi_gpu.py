"""GPU: kd6d_dzi_crop (through the C ABI) vs oracle/dzi_ref.py -- bit-exact images, masks, bbox_trans/scale:
the kernel restates the same fixed-point arithmetic, products rounded before summation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


@pytest.mark.parametrize("shape", [(480, 640, 256), (60, 90, 64), (33, 47, 32)])
def test_dzi_crop_matches_oracle_bit_exact(gpu_device, shape):
    from kd6d.libs import dzi_libs as Dz
    from oracle import dzi_ref as Z
    H, W, R = shape
    r = np.random.default_rng(H)
    B = 4
    frames = r.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    masks = (r.integers(0, 3, (B, H, W)) - 1).astype(np.float32)          # -1 occluded, 0 bg, 1 instance
    rng = np.random.RandomState(1)
    boxes = [[W * 0.3, H * 0.25, W * 0.7, H * 0.8], [2, 3, W * 0.4, H * 0.5], [W * 0.6, H * 0.5, W - 1, H - 1],
             [W * 0.1, H * 0.1, W * 0.95, H * 0.9]]
    cs = [Z.aug_bbox_dzi(b, H, W, rng) for b in boxes]                      # incl. boxes hanging over the frame
    lut = Dz.normalize_lut(MEAN, STD, gpu_device)
    img, msk, tr, sc = Dz.dzi_batch(torch.from_numpy(frames).to(gpu_device), torch.from_numpy(masks).to(gpu_device),
                                    [c for c, _ in cs], [s for _, s in cs], lut, input_res=R)
    torch.cuda.synchronize()
    for b in range(B):
        wi, wm, wt, ws = Z.dzi_crop(frames[b], masks[b], cs[b][0], cs[b][1], MEAN, STD, out_res=R)
        assert np.array_equal(img[b].cpu().numpy(), wi), "image %d" % b
        assert np.array_equal(msk[b].cpu().numpy(), wm), "mask %d" % b
        np.testing.assert_allclose(tr[b].cpu().numpy(), wt, rtol=1e-6, atol=1e-5)
        assert float(sc[b]) == pytest.approx(float(ws), rel=1e-6)


def test_dzi_crop_feeds_the_step_inputs(gpu_device):
    """Shapes/dtypes are what PoseModuleKD and PackedTargets take: (B,3,256,256) fp32, mask (B,256,256), (B,2,3)."""
    from kd6d.libs import dzi_libs as Dz
    B, H, W = 2, 480, 640
    frames = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=gpu_device)
    masks = torch.zeros(B, H, W, device=gpu_device)
    c, s = Dz.test_bbox_DZI([200, 150, 330, 260], H, W)
    img, msk, tr, sc = Dz.dzi_batch(frames, masks, [c, c], [s, s], Dz.normalize_lut(MEAN, STD, gpu_device))
    assert img.shape == (B, 3, 256, 256) and img.dtype == torch.float32 and msk.shape == (B, 256, 256)
    assert tr.shape == (B, 2, 3) and float(tr[0, 0, 0]) == pytest.approx(256 / s)
    assert bool(torch.isfinite(img).all())


def test_bop_reader_feeds_the_gpu_front_end(gpu_device, tmp_path):
    """BOP tree -> BOP_Dataset -> collate_frames -> DZI boxes from the projected 3D boxes -> kd6d_dzi_crop ->
    PackedTargets: the reader's output is what the GPU front-end and the step take."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from bop_fixture import write_tree
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs import dataset as D
    from kd6d.libs import dzi_libs as Dz
    from kd6d.libs.poses import PoseAnnot
    t = write_tree(str(tmp_path))
    ds = D.BOP_Dataset(t["list_file"], t["models"], t["bbox"], training=False)
    frames, masks, targets, metas = D.collate_frames([ds[0], ds[1]])
    H, W = frames.shape[1:3]
    cs = [Dz.test_bbox_DZI(D.projected_box(tg, 0), H, W) for tg in targets]
    img, msk, tr, sc = Dz.dzi_batch(frames.to(gpu_device), masks.to(gpu_device), [c for c, _ in cs], [s for _, s in cs],
                                    Dz.normalize_lut(MEAN, STD, gpu_device), input_res=64)
    assert img.shape == (2, 3, 64, 64) and bool(torch.isfinite(img).all()) and msk.shape == (2, 64, 64)
    crops = [PoseAnnot(tg.keypoints_3d, tg.K, msk[i].cpu(), tg.class_ids, tg.rotations, tg.translations, 64, 64,
                       sc[i].cpu(), tr[i].cpu()) for i, tg in enumerate(targets)]
    packed = PackedTargets(crops, gpu_device)
    assert packed.mask.shape == (2, 64, 64) and packed.bbox_trans.shape == (2, 2, 3) and int(packed.n_gt[0]) == 2
