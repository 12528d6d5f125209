"""Analytic known-answer tests that pin oracle/sinkhorn_ref.py (geomloss is absent: SURVEY 8c)."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from oracle.sinkhorn_ref import epsilon_schedule, kd_loss_images, sinkhorn_divergence


def _rand(seed, B=3, N=7, M=9, D=2):
    r = np.random.default_rng(seed)
    return (r.uniform(0.1, 1.0, (B, N)), r.uniform(0, 1, (B, N, D)),
            r.uniform(0.1, 1.0, (B, M)), r.uniform(0, 1, (B, M, D)))


def test_schedule_matches_formula():
    e = epsilon_schedule(2, 0.7, 0.001, 0.5)
    assert e[0] == pytest.approx(0.49) and e[-1] == pytest.approx(1e-6)
    assert all(e[i] >= e[i + 1] for i in range(1, len(e) - 1))
    assert 11 <= len(e) <= 14


@pytest.mark.parametrize("reach", [None, 0.5])
def test_self_divergence_zero(reach):
    a, x, _, _ = _rand(0)
    S, gx, fa = sinkhorn_divergence(a, x, a, x, reach=reach, with_grad=True)
    assert np.abs(S).max() < 1e-10
    assert np.abs(gx).max() < 1e-8 and np.abs(fa).max() < 1e-10


@pytest.mark.parametrize("reach", [None, 0.5])
def test_symmetry(reach):
    a, x, b, y = _rand(1)
    S1 = sinkhorn_divergence(a, x, b, y, reach=reach, diameter=1.5)
    S2 = sinkhorn_divergence(b, y, a, x, reach=reach, diameter=1.5)
    np.testing.assert_allclose(S1, S2, rtol=1e-9, atol=1e-12)


def test_balanced_small_blur_is_assignment_cost():
    r = np.random.default_rng(2)
    N = 6
    x = r.uniform(0, 1, (1, N, 2)); y = r.uniform(0, 1, (1, N, 2))
    w = np.full((1, N), 1.0 / N)
    S = sinkhorn_divergence(w, x, w, y, blur=1e-3, scaling=0.9, reach=None)  # converged loop
    C = 0.5 * ((x[0][:, None] - y[0][None]) ** 2).sum(-1)
    ri, ci = linear_sum_assignment(C)
    assert S[0] == pytest.approx(C[ri, ci].sum() / N, rel=1e-5)


def test_large_blur_limit_is_energy_distance():
    a, x, b, y = _rand(3, B=1)
    a /= a.sum(); b /= b.sum()
    S = sinkhorn_divergence(a, x, b, y, blur=50.0, reach=None, diameter=60.0)
    pts = np.concatenate([x[0], y[0]]); m = np.concatenate([a[0], -b[0]])
    C = 0.5 * ((pts[:, None] - pts[None]) ** 2).sum(-1)
    assert S[0] == pytest.approx(-0.5 * m @ C @ m, rel=1e-3)


@pytest.mark.parametrize("reach", [None, 0.5])
def test_gradients_match_finite_differences_of_last_extrapolation(reach):
    # autograd in geomloss differentiates only the final extrapolation with the inner
    # potentials frozen.  With a converged loop that is the envelope-theorem gradient
    # (times (rho+eps/2)/(rho+eps) in the unbalanced case, because UnbalancedWeight.backward
    # is not an autograd hook), so central differences of S must agree.
    a, x, b, y = _rand(4, B=1, N=5, M=6)
    if reach is None:  # balanced OT needs equal masses to converge
        a /= a.sum(); b /= b.sum()
    blur = 0.05
    kw = dict(blur=blur, scaling=0.9999 if reach else 0.999, reach=reach, diameter=2.0)
    fac = 1.0 if reach is None else (reach ** 2 + blur ** 2 / 2) / (reach ** 2 + blur ** 2)
    S, gx, fa = sinkhorn_divergence(a, x, b, y, with_grad=True, **kw)
    h = 1e-6
    for (i, d) in [(0, 0), (2, 1), (4, 0)]:
        xp, xm = x.copy(), x.copy()
        xp[0, i, d] += h; xm[0, i, d] -= h
        fd = (sinkhorn_divergence(a, xp, b, y, **kw)[0] - sinkhorn_divergence(a, xm, b, y, **kw)[0]) / (2 * h)
        assert gx[0, i, d] == pytest.approx(fac * fd, rel=1e-2, abs=1e-7)
    if reach is None:
        return
    for i in (1, 3):
        ap, am = a.copy(), a.copy()
        ap[0, i] += h; am[0, i] -= h
        fd = (sinkhorn_divergence(ap, x, b, y, **kw)[0] - sinkhorn_divergence(am, x, b, y, **kw)[0]) / (2 * h)
        assert fa[0, i] == pytest.approx(fd, rel=1e-2, abs=1e-7)


def test_image_loop_skips_empty_sets():
    r = np.random.default_rng(5)
    xs = r.uniform(0, 1, (5, 8, 2)); al = r.uniform(0.1, 0.9, (5, 8))
    yt = r.uniform(0, 1, (4, 8, 2)); be = r.uniform(0.1, 0.9, (4, 8))
    loss, valid, gx, ga = kd_loss_images(xs, al, [0, 3, 3, 5], yt, be, [0, 2, 4, 4])
    assert valid.tolist() == [1, 0, 0] and loss[1] == 0 and loss[2] == 0
    assert np.abs(gx[3:]).max() == 0 and np.abs(gx[:3]).max() > 0
