"""BOP reader: a tiny BOP tree written on the fly (tests/golden/bop_fixture.py), checked against
what was written and against the annotation golden captured from the imported reference
(tests/golden/bop_annotation.npz, make_golden_bop.py)."""
import os
import sys

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, G)


@pytest.fixture()
def tree(tmp_path):
    from bop_fixture import write_tree
    return write_tree(str(tmp_path))


def test_image_and_ply_loaders(tree):
    from kd6d.libs import dataset as D
    a = D.load_image_cached(os.path.join(tree["scene"], "rgb", "000003.png"))
    assert a.dtype == np.uint8 and np.array_equal(a, tree["rgb3"][:, :, ::-1])            # BGR like cv2.imread
    g = D.load_image_cached(os.path.join(tree["scene"], "rgb", "000007.png"))
    assert g.dtype == np.uint16 and np.array_equal(g, tree["grey16"])
    n = D.normalise_frame(g)
    assert n.shape == (tree["H"], tree["W"], 3) and n.dtype == np.uint8
    assert np.array_equal(n[:, :, 0], np.rint(tree["grey16"] * (255.0 / 65535.0)).astype(np.uint8))
    r = D.normalise_frame(D.load_image_cached(os.path.join(tree["scene"], "rgb", "000009.png")))
    back = tree["rgba"][:, :, 3] == 0
    assert (r[:, :, :3][back] == 255).all() and np.array_equal(r[:, :, :3][~back], tree["rgba"][:, :, [2, 1, 0]][~back])
    assert D.load_image_cached(os.path.join(tree["scene"], "rgb", "missing.png")) is None
    meshes, table = D.load_bop_meshes(tree["models"])
    assert table == {"1": 0, "5": 1}
    np.testing.assert_allclose(meshes[0].vertices, tree["v1"], atol=1e-6)                  # ascii
    np.testing.assert_allclose(meshes[1].vertices, tree["v5"], atol=0)                     # binary, extra properties + faces


def test_annotation_matches_reference_golden(tree):
    from kd6d.libs import dataset as D
    z = np.load(os.path.join(G, "bop_annotation.npz"))
    table = {"1": 0, "5": 1}
    for name in ("000003", "000007", "000009"):
        K, m, ids, Rs, Ts = D.get_single_bop_annotation(os.path.join(tree["scene"], "rgb", name + ".png"), table, {})
        np.testing.assert_array_equal(np.asarray(K, np.float64), z[name + "_K"])
        np.testing.assert_array_equal(m, z[name + "_mask"])
        np.testing.assert_array_equal(np.asarray(ids, np.int64), z[name + "_ids"])
        np.testing.assert_array_equal(np.asarray(Rs, np.float64).reshape(-1, 3, 3), z[name + "_R"])
        np.testing.assert_array_equal(np.asarray(Ts, np.float64).reshape(-1, 3, 1), z[name + "_T"])
    # image 3 holds objects (1, 9, 5): 9 is unknown -> skipped without consuming an instance id
    K, m, ids, Rs, Ts = D.get_single_bop_annotation(os.path.join(tree["scene"], "rgb", "000003.png"), table)
    assert ids == [0, 1] and set(np.unique(m)) == {0, 1, 2}


def test_dataset_items_and_collate(tree):
    from kd6d.libs import dataset as D
    ds = D.BOP_Dataset(tree["list_file"], tree["models"], tree["bbox"], training=False)
    assert len(ds) == 3
    frame, target, meta = ds[0]
    assert frame.dtype == torch.uint8 and tuple(frame.shape) == (tree["H"], tree["W"], 3)
    assert meta["path"].endswith("000001/rgb/000003.png") and meta["width"] == tree["W"] and meta["height"] == tree["H"]
    assert target.keypoints_3d.shape == (2, 8, 3) and target.class_ids.tolist() == [0, 1]
    assert target.rotations.shape == (2, 3, 3) and target.translations.shape == (2, 3, 1) and target.mask.shape == (tree["H"], tree["W"])
    np.testing.assert_allclose(target.K.numpy(), np.array(tree["cam"]["3"]["cam_K"]).reshape(3, 3), rtol=1e-6)
    box = D.projected_box(target, 0)
    assert box[2] > box[0] and box[3] > box[1]
    frames, masks, targets, metas = D.collate_frames([ds[0], ds[1]])
    assert tuple(frames.shape) == (2, tree["H"], tree["W"], 3) and masks.dtype == torch.float32 and len(targets) == 2
    # training mode drops images without a known object: image 9 only holds the unknown object id 9
    dt = D.BOP_Dataset(tree["list_file"], tree["models"], tree["bbox"], training=True)
    idx9 = [i for i, p in enumerate(dt.img_files) if p.endswith("000009.png")][0]
    assert dt.getitem1(idx9) is None and dt[idx9] is not None
