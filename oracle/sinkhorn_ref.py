"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the optimal-transport KD loss.

PARITY UNPINNED at the geomloss boundary: the reference calls the third-party package
geomloss==0.2.4 (requirements.txt:45; constructed losses/kd_loss.py:26-30, invoked
losses/loss_libs.py:47,49), which is neither vendored in /root/reference nor installed
in this image, and the reference has no test or golden vector touching it.  This file
restates the published algorithm of geomloss 0.2.4's tensorized
SamplesLoss("sinkhorn", p=2, debias=True) in float64 numpy and is pinned
only by analytic known-answer tests (tests/test_sinkhorn_oracle.py): S(a,a)=0, symmetry,
the assignment-cost limit, the large-blur limit and finite-difference gradients.

Nothing outside tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this module.
"""
import numpy as np

NEG_LOG = -100000.0


def max_diameter(x, y):
    """geomloss.sinkhorn_samples.max_diameter: box diagonal over all points of the call."""
    pts = np.concatenate([x.reshape(-1, x.shape[-1]), y.reshape(-1, y.shape[-1])], 0)
    return float(np.linalg.norm(pts.max(0) - pts.min(0)))


def epsilon_schedule(p, diameter, blur, scaling):
    """geomloss.sinkhorn_divergence.epsilon_schedule."""
    return ([diameter ** p]
            + [float(np.exp(e)) for e in np.arange(p * np.log(diameter), p * np.log(blur), p * np.log(scaling))]
            + [blur ** p])


def _lse(v, axis):
    m = v.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(v - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def _softmin(eps, C, h):
    # C: (B,R,Cc), h: (B,Cc) -> (B,R)
    return -eps * _lse(h[:, None, :] - C / eps, axis=2)


def _cost(x, y):
    d = x[:, :, None, :] - y[:, None, :, :]
    return 0.5 * (d * d).sum(-1)


def sinkhorn_divergence(alpha, x, beta, y, blur=0.001, scaling=0.5, reach=0.5, p=2, diameter=None,
                        with_grad=False, dtype=np.float64):
    """Debiased (unbalanced if reach is not None) Sinkhorn divergence, batch form.

    alpha (B,N), x (B,N,D), beta (B,M), y (B,M,D) -> S (B,)
    with_grad: also returns dS/dx (B,N,D) and dS/dalpha (B,N) exactly as autograd produces
    them in geomloss (gradient only through the last extrapolation and the <alpha, .> factor).
    """
    assert p == 2
    alpha = np.asarray(alpha, dtype); beta = np.asarray(beta, dtype)
    x = np.asarray(x, dtype); y = np.asarray(y, dtype)
    if diameter is None:
        diameter = max_diameter(x, y)
    diameter = max(diameter, 1e-12)
    eps_s = epsilon_schedule(p, diameter, blur, scaling)
    rho = None if reach is None else reach ** p
    lam = (lambda e: 1.0) if rho is None else (lambda e: 1.0 / (1.0 + e / rho))

    with np.errstate(divide="ignore"):
        a_log = np.where(alpha > 0, np.log(np.where(alpha > 0, alpha, 1.0)), NEG_LOG)
        b_log = np.where(beta > 0, np.log(np.where(beta > 0, beta, 1.0)), NEG_LOG)
    C_xx, C_yy, C_xy, C_yx = _cost(x, x), _cost(y, y), _cost(x, y), _cost(y, x)

    eps = eps_s[0]
    l = lam(eps)
    a_x = l * _softmin(eps, C_xx, a_log)
    b_y = l * _softmin(eps, C_yy, b_log)
    a_y = l * _softmin(eps, C_yx, a_log)
    b_x = l * _softmin(eps, C_xy, b_log)
    for eps in eps_s:
        l = lam(eps)
        at_x = l * _softmin(eps, C_xx, a_log + a_x / eps)
        bt_y = l * _softmin(eps, C_yy, b_log + b_y / eps)
        at_y = l * _softmin(eps, C_yx, a_log + b_x / eps)
        bt_x = l * _softmin(eps, C_xy, b_log + a_y / eps)
        a_x, b_y = 0.5 * (a_x + at_x), 0.5 * (b_y + bt_y)
        a_y, b_x = 0.5 * (a_y + at_y), 0.5 * (b_x + bt_x)
    # last extrapolation, all four from the old values
    l = lam(eps)
    h_xx, h_yy = a_log + a_x / eps, b_log + b_y / eps
    h_yx, h_xy = a_log + b_x / eps, b_log + a_y / eps
    a_x_f = l * _softmin(eps, C_xx, h_xx)
    b_y_f = l * _softmin(eps, C_yy, h_yy)
    a_y_f = l * _softmin(eps, C_yx, h_yx)
    b_x_f = l * _softmin(eps, C_xy, h_xy)

    if rho is None:
        fa = b_x_f - a_x_f
        S = (alpha * fa).sum(1) + (beta * (a_y_f - b_y_f)).sum(1)
    else:
        w = rho + eps / 2.0
        ea, eb = np.exp(-a_x_f / rho), np.exp(-b_x_f / rho)
        fa = w * (ea - eb)
        S = (alpha * fa).sum(1) + (beta * w * (np.exp(-b_y_f / rho) - np.exp(-a_y_f / rho))).sum(1)
    if not with_grad:
        return S

    def softmax_grad(C, h, r, c):
        v = h[:, None, :] - C / eps
        v = v - v.max(axis=2, keepdims=True)
        pij = np.exp(v)
        pij /= pij.sum(axis=2, keepdims=True)
        diff = r[:, :, None, :] - c[:, None, :, :]
        return (pij[..., None] * diff).sum(2)          # (B,R,D) = d softmin / d r

    g_xx = softmax_grad(C_xx, h_xx, x, x)
    g_xy = softmax_grad(C_xy, h_xy, x, y)
    if rho is None:
        gx = alpha[..., None] * l * (g_xy - g_xx)
    else:
        gx = alpha[..., None] * w * (-1.0 / rho) * l * (ea[..., None] * g_xx - eb[..., None] * g_xy)
    return S, gx, fa


def kd_loss_images(xs, alpha, s_off, yt, beta, t_off, blur=0.001, scaling=0.5, reach=0.5, dtype=np.float64):
    """Per-image sum over the 8 keypoint problems; restates losses/loss_libs.py:22-51 around the
    OT call: xs (P,8,2) normalised student points, alpha (P,8), yt (M,8,2), beta (M,8), offsets
    (B+1).  Returns loss (B), valid (B), dS/dxs (P,8,2), dS/dalpha (P,8)."""
    B = len(s_off) - 1
    loss = np.zeros(B); valid = np.zeros(B, np.int32)
    gx = np.zeros(xs.shape, np.float64); ga = np.zeros(alpha.shape, np.float64)
    for i in range(B):
        s0, s1, t0, t1 = s_off[i], s_off[i + 1], t_off[i], t_off[i + 1]
        if s1 == s0 or t1 == t0:
            continue
        xi = np.transpose(xs[s0:s1], (1, 0, 2))       # (8,N,2)
        ai = np.transpose(alpha[s0:s1], (1, 0))
        yi = np.transpose(yt[t0:t1], (1, 0, 2))
        bi = np.transpose(beta[t0:t1], (1, 0))
        S, g, fa = sinkhorn_divergence(ai, xi, bi, yi, blur, scaling, reach, with_grad=True, dtype=dtype)
        loss[i] = S.sum(); valid[i] = 1
        gx[s0:s1] = np.transpose(g, (1, 0, 2))
        ga[s0:s1] = np.transpose(fa, (1, 0))
    return loss, valid, gx, ga
