"""Perspective-n-point with RANSAC for the evaluation path (host side, numpy).

The reference calls `cv2.solvePnPRansac(xyz, uv, K, None, flags=cv2.SOLVEPNP_EPNP, reprojectionError=5.0)`
(postprocess/postprocess.py:187) on the 8 x n keypoint correspondences of the cells picked for one object.
OpenCV is not a dependency of this package, so the solver is restated here: **parity unpinned at the cv2
boundary** (different minimal solver, different random sampling); it is pinned by known-answer tests instead
(tests/test_eval_host.py: exact recovery from clean projections, sub-pixel recovery under noise, recovery with
40 % gross outliers, degenerate input reports failure).

Solver: RANSAC over direct linear transforms of up to 8 distinct 3D points (corners of a 3D box: never coplanar),
loose-then-tight consensus, Gauss-Newton on (rotation vector, t) over the consensus set.
"""
import numpy as np


def rodrigues(rvec):
    """Rotation vector -> matrix (the cv2.Rodrigues convention)."""
    r = np.asarray(rvec, np.float64).reshape(3)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def rotvec(R):
    """Matrix -> rotation vector."""
    R = np.asarray(R, np.float64)
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-10:
        return np.zeros(3)
    if np.pi - th < 1e-6:       # near pi: axis from the symmetric part
        A = (R + np.eye(3)) / 2
        k = np.sqrt(np.clip(np.diag(A), 0, None))
        i = int(np.argmax(k))
        k = A[:, i] / max(k[i], 1e-12)
        return th * k / np.linalg.norm(k)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(th))
    return th * w


def project(K, R, T, xyz):
    cam = xyz @ R.T + T.reshape(1, 3)
    uvw = cam @ K.T
    return uvw[:, :2] / (uvw[:, 2:3] + 1e-12), cam[:, 2]


def _dlt(Kinv, xyz, uv):
    """Pose from >= 6 correspondences by the direct linear transform on normalised image coordinates."""
    n = xyz.shape[0]
    xn = (np.concatenate([uv, np.ones((n, 1))], 1) @ Kinv.T)
    X = np.concatenate([xyz, np.ones((n, 1))], 1)
    A = np.zeros((2 * n, 12))
    A[0::2, 0:4] = X
    A[0::2, 8:12] = -xn[:, 0:1] * X
    A[1::2, 4:8] = X
    A[1::2, 8:12] = -xn[:, 1:2] * X
    try:
        _, _, Vt = np.linalg.svd(A)
    except np.linalg.LinAlgError:
        return None
    P = Vt[-1].reshape(3, 4)
    M = P[:, :3]
    if not np.all(np.isfinite(P)) or abs(np.linalg.det(M)) < 1e-18:
        return None
    if np.linalg.det(M) < 0:
        P = -P
        M = -M
    U, S, Vt2 = np.linalg.svd(M)
    R = U @ Vt2
    if np.linalg.det(R) < 0:
        return None
    scale = S.mean()
    T = P[:, 3] / scale
    return R, T


def _refine(K, R, T, xyz, uv, iters=10):
    """Gauss-Newton on the reprojection error over (rotation vector, translation)."""
    r = rotvec(R)
    t = np.asarray(T, np.float64).reshape(3).copy()
    fx, fy = K[0, 0], K[1, 1]
    for _ in range(iters):
        Rm = rodrigues(r)
        cam = xyz @ Rm.T + t
        z = cam[:, 2]
        if np.any(np.abs(z) < 1e-9):
            break
        pred = np.stack([fx * cam[:, 0] / z + K[0, 2], fy * cam[:, 1] / z + K[1, 2]], 1)
        res = (pred - uv).reshape(-1)
        # d(cam)/d(r) = -[R X]_x (left perturbation), d(cam)/d(t) = I
        RX = xyz @ Rm.T
        J = np.zeros((2 * xyz.shape[0], 6))
        du = np.stack([fx / z, np.zeros_like(z), -fx * cam[:, 0] / z ** 2], 1)
        dv = np.stack([np.zeros_like(z), fy / z, -fy * cam[:, 1] / z ** 2], 1)
        skew = np.zeros((xyz.shape[0], 3, 3))
        skew[:, 0, 1] = RX[:, 2]; skew[:, 0, 2] = -RX[:, 1]
        skew[:, 1, 0] = -RX[:, 2]; skew[:, 1, 2] = RX[:, 0]
        skew[:, 2, 0] = RX[:, 1]; skew[:, 2, 1] = -RX[:, 0]
        J[0::2, 0:3] = np.einsum("ni,nij->nj", du, skew)
        J[1::2, 0:3] = np.einsum("ni,nij->nj", dv, skew)
        J[0::2, 3:6] = du
        J[1::2, 3:6] = dv
        try:
            step = np.linalg.lstsq(J, -res, rcond=None)[0]
        except np.linalg.LinAlgError:
            break
        Rm = rodrigues(step[0:3]) @ Rm
        r = rotvec(Rm)
        t = t + step[3:6]
        if np.linalg.norm(step) < 1e-10:
            break
    return rodrigues(r), t


def solve_pnp_ransac(xyz, uv, K, reproj_err=5.0, iters=300, seed=0):
    """-> (ok, R (3,3), T (3,1), inlier index array).  xyz (n,3), uv (n,2) pixels, K (3,3).

    The correspondences of this path are n_cells observations of the same few 3D points (8 box corners), so a
    minimal sample takes ONE observation of each of up to 8 distinct 3D points (a sample with a repeated 3D point
    is degenerate for the DLT).  Hypotheses are scored with a loose threshold (3 x reproj_err: an 8-point DLT of
    noisy pixels is a rough model), the best are re-fitted on their loose consensus set, refined by Gauss-Newton and
    ranked by the number of inliers at reproj_err."""
    xyz = np.asarray(xyz, np.float64).reshape(-1, 3)
    uv = np.asarray(uv, np.float64).reshape(-1, 2)
    K = np.asarray(K, np.float64).reshape(3, 3)
    n = xyz.shape[0]
    if n < 6 or uv.shape[0] != n or not (np.all(np.isfinite(xyz)) and np.all(np.isfinite(uv))):
        return False, None, None, None
    _, corner = np.unique(np.round(xyz, 6), axis=0, return_inverse=True)
    corner = corner.reshape(-1)
    ncorner = int(corner.max()) + 1
    if ncorner < 6:
        return False, None, None, None
    obs = [np.nonzero(corner == c)[0] for c in range(ncorner)]
    m = min(8, ncorner)
    Kinv = np.linalg.inv(K)
    rng = np.random.default_rng(seed)

    def tight(R, T):
        pred, z = project(K, R, T, xyz)
        return (np.linalg.norm(pred - uv, axis=1) < reproj_err) & (z > 0)

    best = (0, None, None, None)          # tight inlier count, R, T, mask
    best_loose = 0
    for _ in range(iters):
        cs = rng.choice(ncorner, m, replace=False)
        idx = np.array([obs[c][rng.integers(len(obs[c]))] for c in cs])
        fit = _dlt(Kinv, xyz[idx], uv[idx])
        if fit is None:
            continue
        pred, z = project(K, fit[0], fit[1], xyz)
        loose = (np.linalg.norm(pred - uv, axis=1) < 3.0 * reproj_err) & (z > 0)
        nl = int(loose.sum())
        if nl < 6 or nl < 0.8 * best_loose:
            continue
        best_loose = max(best_loose, nl)
        if len(np.unique(corner[loose])) < 6:
            continue
        fit2 = _dlt(Kinv, xyz[loose], uv[loose])
        if fit2 is None:
            continue
        R, T = _refine(K, fit2[0], fit2[1], xyz[loose], uv[loose], iters=4)
        inl = tight(R, T)
        cnt = int(inl.sum())
        if cnt >= 6 and len(np.unique(corner[inl])) >= 6:
            R, T = _refine(K, R, T, xyz[inl], uv[inl], iters=6)
            inl = tight(R, T)
            cnt = int(inl.sum())
        if cnt > best[0]:
            best = (cnt, R, T, inl)
            if cnt == n:
                break
    cnt, R, T, inl = best
    if cnt < 6 or not (np.all(np.isfinite(R)) and np.all(np.isfinite(T))):
        return False, None, None, None
    return True, R.astype(np.float32), T.reshape(3, 1).astype(np.float32), np.nonzero(inl)[0]


def solve_pnp(xyz, uv, K):
    """cv2.solvePnP without RANSAC for clean correspondences (libs/utils.py:511 uses SOLVEPNP_EPNP): direct linear
    transform + Gauss-Newton.  -> (ok, R (3,3), T (3,1))."""
    xyz = np.asarray(xyz, np.float64).reshape(-1, 3)
    uv = np.asarray(uv, np.float64).reshape(-1, 2)
    K = np.asarray(K, np.float64).reshape(3, 3)
    if xyz.shape[0] < 6 or uv.shape[0] != xyz.shape[0] or not (np.all(np.isfinite(xyz)) and np.all(np.isfinite(uv))):
        return False, None, None
    fit = _dlt(np.linalg.inv(K), xyz, uv)
    if fit is None:
        return False, None, None
    R, T = _refine(K, fit[0], fit[1], xyz, uv, iters=20)
    if not (np.all(np.isfinite(R)) and np.all(np.isfinite(T))):
        return False, None, None
    return True, R, np.asarray(T, np.float64).reshape(3, 1)


def remap_pose(srcK, srcR, srcT, pt3d, dstK, transM):
    """libs/utils.py:504-526: the pose that, seen through dstK, projects the object's 3D points where (srcR, srcT) seen
    through srcK and mapped by the homography transM put them.  -> (newR, newT (3,1), mean reprojection difference in
    pixels), or (srcR, srcT, -1) when no pose is found."""
    pt3d = np.asarray(pt3d, np.float64).reshape(-1, 3)
    srcR = np.asarray(srcR, np.float64).reshape(3, 3)
    srcT = np.asarray(srcT, np.float64).reshape(3, 1)
    dstK = np.asarray(dstK, np.float64).reshape(3, 3)
    pts = np.asarray(transM, np.float64) @ (np.asarray(srcK, np.float64).reshape(3, 3) @ (srcR @ pt3d.T + srcT))
    xy2d = np.stack([pts[0] / (pts[2] + 1e-8), pts[1] / (pts[2] + 1e-8)], 1)
    ok, newR, newT = solve_pnp(pt3d, xy2d, dstK)
    if not ok:
        print("Error in pose remapping!")
        return srcR, srcT, -1
    new = dstK @ (newR @ pt3d.T + newT)
    new_xy = np.stack([new[0] / (new[2] + 1e-8), new[1] / (new[2] + 1e-8)], 1)
    return newR, newT, float(np.linalg.norm(xy2d - new_xy, axis=1).mean())
