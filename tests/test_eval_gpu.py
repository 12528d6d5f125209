"""Evaluation path on the GPU: the candidate cells kd6d_pose_candidates hands to the solver ==
what the imported reference's PostProcessor.forward hands to cv2.solvePnPRansac (tests/golden/eval_candidates.npz),
and the eval-mode forward of PoseModuleKD recovers the pose those keypoints encode."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup(dev, precision="fp32"):
    from test_step_gpu import build
    from kd6d.synthetic import make_batch
    z = np.load(os.path.join(G, "eval_candidates.npz"))
    model = build("darknet53", precision, int(z["model_seed"]), dev, cls_bias=z["teacher_cls_bias"]).eval()
    images, targets = make_batch(int(z["batch"]), int(z["seed"]), crop=int(z["crop"]))
    images.tensors = images.tensors.to(dev)
    return z, model, images, targets


def test_candidates_match_reference_golden(gpu_device):
    from kd6d.kd_losses import PackedTargets
    from kd6d.postprocess.postprocess import pose_candidates
    from kd6d._lib import MAX_GT
    from kd6d.kd_losses import CAP
    z, model, images, targets = _setup(gpu_device)
    B = int(z["batch"])
    with torch.no_grad():
        cls, reg = model.net.forward(images.tensors)
    tgt = PackedTargets(targets, gpu_device)
    cnt, kp, score = pose_candidates(cls, reg, model.net.levels, B, tgt.bbox_trans, tgt.class_ids, tgt.n_gt,
                                     model.inference_th, model.positive_num, model.positive_lambda)
    cnt = cnt.cpu().numpy().reshape(B, MAX_GT)
    kp = kp.cpu().numpy().reshape(B, MAX_GT, CAP, 8, 2)
    score = score.cpu().numpy().reshape(B, MAX_GT, CAP, 8)
    for b in range(B):
        assert int(z["n_results"][b]) == int((cnt[b] > 0).sum())
        for j in range(int(z["n_results"][b])):
            ref = z["img%d_%d_xy2d" % (b, j)]                      # (n, 8, 2)
            g = [g for g in range(MAX_GT) if cnt[b, g] > 0][j]
            assert int(tgt.class_ids.cpu().reshape(B, MAX_GT)[b, g]) == int(z["img%d_%d_cls" % (b, j)])
            n = int(cnt[b, g])
            assert n == ref.shape[0]
            # same cells (the reference concatenates levels in order, top-k descending inside a level: compare as sets)
            a = kp[b, g, :n].reshape(n, -1); r = ref.reshape(n, -1)
            a = a[np.lexsort(a.T[::-1])]; r = r[np.lexsort(r.T[::-1])]
            np.testing.assert_allclose(a, r, rtol=1e-4, atol=0.3)            # fp32 logits through 50+ layers, pixels
            np.testing.assert_allclose(score[b, g, :n].max(), float(z["img%d_%d_score" % (b, j)]), rtol=2e-3)


def test_eval_forward_is_consistent(gpu_device):
    """A randomly initialised network predicts keypoints no pose explains, so the solver may (correctly) reject an
    object; whatever comes back must be well formed."""
    z, model, images, targets = _setup(gpu_device)
    with torch.no_grad():
        pred, extra = model(images, targets)
    assert extra == {} and len(pred) == int(z["batch"])
    for b, res in enumerate(pred):
        assert len(res) <= int(z["n_results"][b])
        for score, cid, R, T, xy2d in res:
            assert 0.0 < score <= 1.0 and cid == int(targets[b].class_ids[0])
            assert R.shape == (3, 3) and T.shape == (3, 1) and abs(np.linalg.det(R) - 1) < 1e-3
            assert xy2d.shape[1:] == (8, 2)


def test_postprocessor_recovers_encoded_pose(gpu_device):
    """Logits that encode the projected 3D-box corners of a known pose (with pixel noise and one corrupted cell) ->
    PostProcessor.forward returns that pose and the ADI of prediction vs truth is tiny."""
    from kd6d import engine
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.evaluate import compute_pose_diff
    from kd6d.postprocess import PostProcessor
    from kd6d.synthetic import make_batch
    dev = gpu_device
    B, crop = 2, 256
    _, targets = make_batch(B, 77, crop=crop)
    levels = [(crop // s, crop // s) for s in engine.ANCHOR_STRIDES]
    rows = B * sum(h * w for h, w in levels)
    cls = torch.full((rows, 16), -10.0)
    reg = torch.zeros(rows, 240)
    rng = np.random.default_rng(3)
    row0 = 0
    picked = {b: 0 for b in range(B)}
    for li, (h, w) in enumerate(levels):
        st, sz = float(engine.ANCHOR_STRIDES[li]), float(engine.ANCHOR_SIZES[li])
        for b in range(B):
            t = targets[b]
            c = int(t.class_ids[0])
            Kb, R, T = t.K.numpy().astype(np.float64), t.rotations[0].numpy().astype(np.float64), t.translations[0].numpy().reshape(3, 1)
            cam = R @ t.keypoints_3d[c].numpy().T.astype(np.float64) + T
            uv = (Kb @ cam)[:2] / (Kb @ cam)[2]                       # full-frame pixels (2, 8)
            bt = t.bbox_trans.numpy().astype(np.float64)
            p = bt[:, :2] @ uv + bt[:, 2:3]                           # crop pixels
            ctr = p.mean(1)
            ix, iy = int(np.clip(ctr[0] // st, 0, w - 1)), int(np.clip(ctr[1] // st, 0, h - 1))
            for (dx, dy) in ((0, 0), (1, 0), (0, 1)):
                x, y = min(ix + dx, w - 1), min(iy + dy, h - 1)
                cell = y * w + x
                r = row0 + b * h * w + cell
                if cls[r, c] > 0:
                    continue
                cx, cy = x * st + st * 0.5, y * st + st * 0.5
                q = p + rng.normal(0, 0.7, p.shape)
                if li == 1 and (dx, dy) == (1, 0):
                    q = q + rng.normal(0, 40, p.shape)                # one grossly wrong cell
                cls[r, c] = 3.0 - 0.1 * picked[b]
                reg[r, c * 16:c * 16 + 8] = torch.from_numpy((q[0] - cx) / sz)
                reg[r, c * 16 + 8:c * 16 + 16] = torch.from_numpy((q[1] - cy) / sz)
                picked[b] += 1
        row0 += B * h * w
    pp = PostProcessor(0.1, 10, 1.0)
    res = pp(cls.to(dev), reg.to(dev), levels, B, PackedTargets(targets, dev))
    assert len(res) == B
    for b in range(B):
        assert len(res[b]) == 1
        score, cid, R, T, xy2d = res[b][0]
        t = targets[b]
        assert cid == int(t.class_ids[0]) and score > 0.9 and xy2d.shape[0] >= 3
        mesh = t.keypoints_3d[cid].numpy().astype(np.float64)
        e3, e2 = compute_pose_diff(mesh, t.K.numpy().astype(np.float64), t.rotations[0].numpy().astype(np.float64),
                                   t.translations[0].numpy().reshape(3, 1).astype(np.float64), R.astype(np.float64),
                                   T.astype(np.float64))
        diam = float(np.linalg.norm(mesh.max(0) - mesh.min(0)))
        assert e3 / diam < 0.1 and e2 < 5.0, (e3 / diam, e2)         # ADI-0.1d and REP-5px hold


def test_valid_loop_plumbing(gpu_device):
    """libs/eval_libs.valid over two synthetic batches: every image is scored (a missing pose counts as ADI 1.0)."""
    from kd6d.libs.eval_libs import valid
    from kd6d.synthetic import make_batch
    z, model, images, targets = _setup(gpu_device)
    cfg = model.cfg
    loader = []
    for i in range(2):
        im, tg = make_batch(2, 50 + i, crop=int(z["crop"]))
        im.tensors = im.tensors.to(gpu_device)
        metas = [{"path": "b%d_%d" % (i, j), "K": t.K.numpy(), "class_ids": [int(c) for c in t.class_ids],
                  "rotations": [r.numpy() for r in t.rotations], "translations": [x.numpy().reshape(3, 1) for x in t.translations]}
                 for j, t in enumerate(tg)]
        loader.append((im, tg, metas))
    meshes = [targets[0].keypoints_3d[c].numpy() for c in range(cfg["DATASETS"]["N_CLASS"] - 1)]
    out = valid(cfg, 0, loader, model, gpu_device, meshes)
    adi, auc, rep, adi_d, rep_d, rng = out
    assert len(adi) == cfg["DATASETS"]["N_CLASS"] - 1 and rng[0] < rng[1]
    assert set(adi[0].keys()) == {"ADI.05d", "ADI.10d", "ADI.20d", "ADI.50d"} and 0.0 <= adi[0]["ADI.50d"] <= 100.0
    assert sum(len(e) > 0 for e in adi_d) >= 1
