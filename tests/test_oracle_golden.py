"""CPU: the oracle (oracle/kd_step_ref.py) reproduces the golden vectors captured from the
imported reference by tests/golden/make_golden.py (which also asserted agreement at capture time)."""
import os

import numpy as np
import pytest
import torch

from kd6d.synthetic import INTERNAL_K, MESH_DIAMETERS, make_batch
from oracle import kd_step_ref as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_pieces_focal_and_decode():
    z = np.load(os.path.join(G, "pieces.npz"))
    logits = torch.from_numpy(z["focal_logits"]).requires_grad_(True)
    labels = torch.from_numpy(z["focal_labels"])
    keep = labels >= 0
    loss = O.focal_loss_sum(logits[keep], labels[keep])
    loss.backward()
    assert float(loss) == pytest.approx(float(z["focal_loss"]), rel=1e-6)
    np.testing.assert_allclose(logits.grad.numpy(), z["focal_grad"], rtol=1e-5, atol=1e-7)
    c, s, _ = O.anchor_centers([(4, 4), (2, 2), (1, 1)])
    anc = torch.from_numpy(z["anchors"])
    torch.testing.assert_close((anc[:, :2] + anc[:, 2:]) / 2, c)
    torch.testing.assert_close(anc[:, 2] - anc[:, 0] + 1, s)
    preds = torch.from_numpy(z["dec_preds"])
    bt = torch.from_numpy(z["dec_bt"]).repeat(preds.shape[0], 1, 1)
    plain = O.decode_points(preds, c, s).permute(0, 2, 1).reshape(-1, 16)
    aff = O.decode_points(preds, c, s, bt).permute(0, 2, 1).reshape(-1, 16)
    np.testing.assert_allclose(plain.numpy(), z["dec_plain"], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(aff.numpy(), z["dec_affine"], rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("name", ["step_tinyh_b2_128", "step_tiny_b2_128"])
def test_full_step_matches_reference_capture(name):
    z = np.load(os.path.join(G, name + ".npz"))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    images, targets = make_batch(int(z["batch"]), int(z["seed"]), crop=int(z["crop"]))
    step = O.KDStepRef(str(z["student_arch"]), "darknet53", K=INTERNAL_K, diameters=MESH_DIAMETERS,
                       kd_weight=5.0, teacher_cls_bias=z["teacher_cls_bias"])
    torch.manual_seed(int(z["rng_seed"]))
    res, ex = step.step(images.tensors, [t.as_dict() for t in targets], return_extras=True)
    assert res["loss_cls"] == pytest.approx(float(z["loss_cls"]), rel=2e-4)
    assert res["loss_reg"] == pytest.approx(float(z["loss_reg"]), rel=2e-4)
    assert res["loss_kd"] == pytest.approx(float(z["loss_kd"]), rel=2e-4)
    assert res["grad_norm"] == pytest.approx(float(z["grad_norm"]), rel=1e-3)
    assert ex["out"]["pos_per_img"] == z["pos_per_img"].tolist()
    assert np.array_equal(ex["labels"].numpy().astype(np.int8), z["labels"])
    scores, kps = ex["teacher"]
    assert [s.shape[0] for s in scores] == z["teacher_counts"].tolist()
    np.testing.assert_allclose(torch.cat(kps).numpy(), z["teacher_kp"], rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(torch.cat(scores).numpy(), z["teacher_cls"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ex["out"]["student_pts"].detach().numpy(), z["student_pts"], rtol=1e-4, atol=1e-2)
