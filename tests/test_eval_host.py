"""Host side of the evaluation path: the PnP-RANSAC solver against known answers (parity
unpinned at the cv2 boundary) and the accuracy metrics against golden vectors captured from the imported
reference (tests/golden/eval_metrics.npz, tests/golden/make_golden_eval.py)."""
import json
import os
import sys

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, G)

K = np.array([[572.4, 0, 325.3], [0, 573.6, 242.0], [0, 0, 1.0]])


def _box(d=100.0):
    h = d / np.sqrt(3) / 2 * np.array([1.0, 1.2, 0.8])
    return np.array([[sx * h[0], sy * h[1], sz * h[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])


def _pose(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    return q, np.array([rng.normal(0, 60), rng.normal(0, 40), 900 + rng.normal(0, 80)])


def _errors(R, T, Q, Tt):
    ang = np.degrees(np.arccos(np.clip((np.trace(R.T @ Q) - 1) / 2, -1, 1)))
    return ang, float(np.linalg.norm(T.reshape(3) - Tt))


@pytest.mark.parametrize("noise,outliers,max_ang,max_t,max_bad", [(0.0, 0.0, 0.05, 0.05, 0), (1.0, 0.0, 1.5, 12.0, 0),
                                                                 (0.5, 0.4, 3.0, 25.0, 2)])
def test_pnp_ransac_known_answers(noise, outliers, max_ang, max_t, max_bad):
    from kd6d.libs import pnp
    rng = np.random.default_rng(5)
    bad = 0
    for _ in range(16):
        Q, Tt = _pose(rng)
        xyz = np.tile(_box(), (10, 1))
        uv, _ = pnp.project(K, Q, Tt, xyz)
        uv = uv + rng.normal(0, noise, uv.shape) if noise else uv
        o = rng.random(len(uv)) < outliers
        uv[o] += rng.normal(0, 60, (int(o.sum()), 2))
        ok, R, T, inl = pnp.solve_pnp_ransac(xyz, uv, K)
        if not ok:
            bad += 1
            continue
        ang, te = _errors(R, T, Q, Tt)
        bad += int(ang > max_ang or te > max_t)
        assert R.shape == (3, 3) and T.shape == (3, 1) and abs(np.linalg.det(R) - 1) < 1e-4
        if outliers:
            assert len(inl) <= (~o).sum() + 3
    assert bad <= max_bad


def test_pnp_degenerate_input_reports_failure():
    from kd6d.libs import pnp
    assert pnp.solve_pnp_ransac(np.zeros((16, 3)), np.zeros((16, 2)), K)[0] is False        # one distinct 3D point
    assert pnp.solve_pnp_ransac(_box()[:4], np.zeros((4, 2)), K)[0] is False                # fewer than 6 points
    bad = np.tile(_box(), (2, 1)); uv = np.full((16, 2), np.nan)
    assert pnp.solve_pnp_ransac(bad, uv, K)[0] is False
    r = np.array([0.3, -1.1, 0.7])
    np.testing.assert_allclose(pnp.rotvec(pnp.rodrigues(r)), r, atol=1e-10)


def test_metrics_against_reference_golden():
    from kd6d.libs import evaluate as E
    from make_golden_eval import _Mesh, eval_inputs
    z = np.load(os.path.join(G, "eval_metrics.npz"))
    meshes, diam, Kk, preds = eval_inputs(int(z["seed"]))
    for row in z["pose_diff"]:
        i, mi, sym = int(row[0]), int(row[1]), bool(row[2])
        it = preds["img%02d" % i]
        R1, T1 = it["meta"]["rotations"][0], it["meta"]["translations"][0]
        R2, T2 = row[7:16].reshape(3, 3), row[16:19].reshape(3, 1)
        np.random.seed(100 + i)
        e3, e2 = E.compute_pose_diff(meshes[mi], Kk, R1, T1, R2, T2, isSym=sym)
        er, et = E.compute_pose_diff_speed(R1, T1, R2, T2)
        np.testing.assert_allclose([e3, e2, er, et], row[3:7], rtol=1e-9, atol=1e-9)
    errs = z["auc_in"]
    got = [E.evalute_auc_metric(errs, 100), E.evalute_auc_metric(errs[:7], 50), E.evalute_auc_metric([], 100)]
    np.testing.assert_allclose(got, z["auc"], rtol=1e-12)
    q = np.stack([E.rotation2quaternion(np.asarray(it["meta"]["rotations"][0])) for it in preds.values()])
    np.testing.assert_allclose(q, z["quat"], rtol=1e-12, atol=1e-12)
    np.random.seed(7)
    res = E.evaluate_pose_predictions(preds, 4, [_Mesh(m) for m in meshes], diam, {"cls_2": ["Z", 0]})
    ref = json.loads(str(z["evaluate_json"]))
    mine = json.loads(json.dumps([res[0], res[1], res[2], res[3], res[4], [float(x) for x in res[5]]], sort_keys=True))

    def close(a, b):
        if isinstance(a, dict):
            assert a.keys() == b.keys(), (a.keys(), b.keys())
            for k in a:
                close(a[k], b[k])
        elif isinstance(a, list):
            assert len(a) == len(b)
            for x, y in zip(a, b):
                close(x, y)
        else:
            assert abs(a - b) <= 1e-9 * max(1.0, abs(b)), (a, b)
    close(mine, ref)


def test_symmetry_handling_known_answers():
    from scipy.spatial.transform import Rotation
    from kd6d.libs.evaluate import pose_symmetry_handling
    R = Rotation.from_euler("zyx", [0.9, 0.3, -0.2]).as_matrix()
    assert pose_symmetry_handling(R, []) is R
    # continuous symmetry about Z: the angle about Z is removed, the other two survive
    Rz = pose_symmetry_handling(R, ["Z", 0])
    np.testing.assert_allclose(Rotation.from_matrix(Rz).as_euler("zyx"), [0.0, 0.3, -0.2], atol=1e-6)
    np.testing.assert_allclose(pose_symmetry_handling(Rz, ["Z", 0]), Rz, atol=1e-6)             # idempotent
    # 180-degree symmetry about X folds the X angle into (-pi, pi) mod pi
    R2 = Rotation.from_euler("xyz", [2.5, 0.1, 0.4]).as_matrix()
    a = Rotation.from_matrix(pose_symmetry_handling(R2, ["X", 180])).as_euler("xyz")
    np.testing.assert_allclose(a, [np.fmod(2.5, np.pi), 0.1, 0.4], atol=1e-6)
    with pytest.raises(ValueError):
        pose_symmetry_handling(R, ["Q", 0])


def test_remap_predictions_known_answers():
    """libs/evaluate.py:174-197 / libs/utils.py:504-526: a pose solved for the internal camera, re-solved for the frame's
    own camera through the homography K * K_int^-1.  Known answers: the identity when both cameras agree; for a camera
    with another focal length / principal point the remapped pose must reproject the object's points (through the new
    camera) exactly where the homography puts the old projections; K * K_int^-1 maps a pixel to the pixel of the SAME
    ray in the other camera, so the pose itself does not change (focal length x 1.25: same R, same depth)."""
    from kd6d.libs.evaluate import remap_predictions
    from kd6d.libs.pnp import project, remap_pose, rodrigues
    rng = np.random.default_rng(4)
    K_int = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1.0]])
    kp = (rng.uniform(-1, 1, (15, 8, 3)) * 60.0)
    R = rodrigues(np.array([0.3, -0.7, 0.2]))
    T = np.array([[25.0], [-40.0], [900.0]])
    pred = [[0.9, 3, R, T, np.zeros((8, 2))]]
    same = remap_predictions(K_int.reshape(-1).tolist(), 640, 480, kp, {"K": K_int}, pred)
    np.testing.assert_allclose(same[0][2], R, atol=1e-8)
    np.testing.assert_allclose(same[0][3], T, atol=1e-5)
    assert same[0][0] == 0.9 and same[0][1] == 3
    # focal length x 1.25, same principal point: depth scales with the focal length
    K2 = K_int.copy()
    K2[0, 0] *= 1.25; K2[1, 1] *= 1.25
    out = remap_predictions(K_int.reshape(-1).tolist(), 640, 480, kp, {"K": K2}, pred)
    np.testing.assert_allclose(out[0][2], R, atol=2e-3)
    np.testing.assert_allclose(out[0][3][2, 0], 900.0, rtol=2e-3)       # K2 * K_int^-1 maps u -> the same ray: same pose
    # a different camera altogether: reprojection consistency
    K3 = np.array([[610.0, 0, 300.0], [0, 590.0, 260.0], [0, 0, 1.0]])
    newR, newT, err = remap_pose(K_int, R, T, kp[3], K3, K3 @ np.linalg.inv(K_int))
    assert err < 1e-3
    uv_old, _ = project(K_int, R, T.reshape(3), kp[3])
    h = (K3 @ np.linalg.inv(K_int)) @ np.concatenate([uv_old, np.ones((8, 1))], 1).T
    uv_new, z = project(K3, newR, newT.reshape(3), kp[3])
    np.testing.assert_allclose(uv_new, (h[:2] / h[2]).T, atol=1e-3)
    assert np.all(z > 0)
