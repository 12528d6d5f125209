"""The loss-level drop-in surface (kd6d.losses: SamplesLoss, kd_loss_2d, KDPoseLoss with the signatures of
losses/kd_loss.py:13-161 and losses/loss_libs.py:1-51 of the reference) against the CPU oracle, values and
gradients, through torch autograd the way a torch training loop would use them."""
import numpy as np
import pytest
import torch

from test_step_gpu import ref_to_packed_rows

pytestmark = pytest.mark.gpu


def _problem(B, N, M, D, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, N, D, generator=g) * 0.2 + 0.4
    y = x[:, torch.randperm(N, generator=g)[:M] % N] + 0.01 * torch.randn(B, M, D, generator=g) if M <= N else \
        torch.rand(B, M, D, generator=g) * 0.2 + 0.4
    a = torch.rand(B, N, generator=g) * 0.9 + 0.05
    b = torch.rand(B, M, generator=g) * 0.9 + 0.05
    return a, x, b, y


@pytest.mark.parametrize("B,N,M,reach", [(8, 10, 9, 0.5), (8, 10, 10, None), (3, 7, 12, 0.5), (1, 40, 33, 0.5)])
def test_samples_loss_small_sets_vs_oracle(gpu_device, B, N, M, reach):
    """SamplesLoss("sinkhorn", p=2, blur, scaling, reach)(alpha, x, beta, y) -> (B,): the call of
    losses/kd_loss.py:26-30 / loss_libs.py:47 (batch = the 8 keypoints of an image; shorter batches are padded)."""
    from kd6d.losses import SamplesLoss
    from oracle import kd_step_ref as O
    a, x, b, y = _problem(B, N, M, 2, 5 + B)
    xr, ar = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
    ref = O.sinkhorn_divergence_torch(ar, xr, b, y, blur=0.001, scaling=0.5, reach=reach)
    w = torch.linspace(0.5, 1.5, B)
    (ref * w).sum().backward()
    dev = gpu_device
    xg, ag = x.to(dev).requires_grad_(True), a.to(dev).requires_grad_(True)
    L = SamplesLoss("sinkhorn", p=2, blur=0.001, scaling=0.5, reach=reach)
    got = L(ag, xg, b.to(dev), y.to(dev))
    assert got.shape == (B,)
    (got * w.to(dev)).sum().backward()
    torch.testing.assert_close(got.detach().cpu(), ref.detach(), rtol=2e-4, atol=1e-7)
    torch.testing.assert_close(ag.grad.cpu(), ar.grad, rtol=2e-3, atol=1e-6)
    scale = float(xr.grad.abs().max())
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=5e-3, atol=5e-3 * scale)
    # (x, y) form: uniform weights
    got_u = L(x.to(dev), y.to(dev))
    ref_u = O.sinkhorn_divergence_torch(torch.full((B, N), 1.0 / N), x, torch.full((B, M), 1.0 / M), y, 0.001, 0.5, reach)
    torch.testing.assert_close(got_u.cpu(), ref_u, rtol=2e-4, atol=1e-7)
    with pytest.raises(NotImplementedError):
        SamplesLoss("gaussian")
    with pytest.raises(NotImplementedError):
        L(ag, xg, b.to(dev), y.to(dev).requires_grad_(True))


def test_samples_loss_large_sets_take_the_dense_kernel(gpu_device):
    """Above kd6d_sinkhorn_max_points (or D > 2) the same call runs the dense kernel per batch row with the joint
    diameter: D = 16, 300 x 280 points, blur 0.05, vs the fp64 oracle."""
    from kd6d.losses import SamplesLoss
    from oracle.sinkhorn_ref import sinkhorn_divergence
    B, N, M, D = 2, 300, 280, 16
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(B, N, D, generator=g), torch.rand(B, M, D, generator=g)
    a, b = torch.rand(B, N, generator=g) + 0.1, torch.rand(B, M, generator=g) + 0.1
    pts = torch.cat([x.reshape(-1, D), y.reshape(-1, D)])
    diam = float((pts.max(0)[0] - pts.min(0)[0]).norm())
    dev = gpu_device
    xg, ag = x.to(dev).requires_grad_(True), a.to(dev).requires_grad_(True)
    got = SamplesLoss("sinkhorn", p=2, blur=0.05, scaling=0.5, reach=0.5)(ag, xg, b.to(dev), y.to(dev))
    got.sum().backward()
    for i in range(B):
        S, gx, fa = sinkhorn_divergence(a[i:i + 1].numpy().astype(np.float64), x[i:i + 1].numpy().astype(np.float64),
                                        b[i:i + 1].numpy().astype(np.float64), y[i:i + 1].numpy().astype(np.float64),
                                        blur=0.05, scaling=0.5, reach=0.5, diameter=diam, with_grad=True)
        S, gx, fa = S[0], gx[0], fa[0]
        assert float(got[i]) == pytest.approx(float(S), rel=5e-4)
        np.testing.assert_allclose(ag.grad[i].cpu().numpy(), fa, rtol=5e-3, atol=5e-3 * np.abs(fa).max())
        np.testing.assert_allclose(xg.grad[i].cpu().numpy(), gx, rtol=1e-2, atol=1e-2 * np.abs(gx).max())


def test_kd_loss_2d_matches_per_image_loop(gpu_device):
    """kd_loss_2d(pred_xy, target_xy, pred_cls, target_cls, w, h, level, kd_loss, dim, pos_per_img, pos_per_img_t):
    one packed launch for all images == the reference's per-image loop over the oracle OT (an image with an empty
    set is skipped, the inputs are normalised in place)."""
    from kd6d.losses import SamplesLoss, kd_loss_2d
    from oracle import kd_step_ref as O
    g = torch.Generator().manual_seed(9)
    pos, pos_t = [10, 0, 7, 9], [9, 8, 0, 10]
    P, M = sum(pos), sum(pos_t)
    pred = torch.rand(P * 8, 2, generator=g) * torch.tensor([640.0, 480.0])
    targ = pred[torch.randint(0, P * 8, (M * 8,), generator=g)] + torch.randn(M * 8, 2, generator=g) * 3.0
    a = torch.rand(P, 1, generator=g).expand(P, 8).contiguous() * 0.9 + 0.05
    b = torch.rand(M, 8, generator=g) * 0.9 + 0.05
    dev = gpu_device

    def run(kd, device):
        leaf = pred.clone().to(device).requires_grad_(True)
        al = a.clone().to(device).requires_grad_(True)
        pxy = leaf * 1.0                                        # non-leaf, as in the reference (decode output)
        txy = targ.clone().to(device)
        out = kd_loss_2d(pxy, txy, al, b.to(device), 640, 480, "point", kd, 2, pos_per_img=pos, pos_per_img_t=pos_t)
        (sum(out) / len(out)).backward()
        return out, leaf.grad.cpu(), al.grad.cpu(), pxy.detach().cpu(), txy.cpu()

    ref_fn = lambda al, x, be, y: O.sinkhorn_divergence_torch(al, x, be, y, 0.001, 0.5, 0.5)   # noqa: E731
    out_r, gx_r, ga_r, pn_r, tn_r = run(ref_fn, "cpu")
    kd = SamplesLoss("sinkhorn", p=2, blur=0.001, scaling=0.5, reach=0.5)
    out_g, gx_g, ga_g, pn_g, tn_g = run(kd, dev)
    assert len(out_g) == len(out_r) == 2
    torch.testing.assert_close(torch.stack([o.detach().cpu() for o in out_g]), torch.stack([o.detach() for o in out_r]),
                               rtol=2e-4, atol=1e-7)
    torch.testing.assert_close(pn_g, pn_r)                      # normalised in place on both sides
    torch.testing.assert_close(tn_g, tn_r)
    torch.testing.assert_close(ga_g, ga_r, rtol=2e-3, atol=1e-6)
    torch.testing.assert_close(gx_g, gx_r, rtol=5e-3, atol=5e-3 * float(gx_r.abs().max()))
    # any other callable is applied image by image (here: the oracle on CPU tensors through the same function)
    assert float(gx_r.abs().max()) > 0


@pytest.mark.parametrize("mixed", [False, True])
def test_kd_pose_loss_call_signature_vs_oracle(gpu_device, mixed):
    """KDPoseLoss(gamma, alpha, anchor_sizes, anchor_strides, positive_type, positive_num, positive_lambda, top_k,
    internal_K, diameters, target_coder, cfg_kd)(pred_cls, pred_reg, targets, anchors, pred_t) on per-level NCHW
    head outputs that carry autograd, with pred_t in the reference's dict layout -> [cls, reg, kd] and the gradients
    of their weighted sum w.r.t. every level, vs oracle.kd_pose_loss."""
    from kd6d.losses import KDPoseLoss
    from kd6d.synthetic import INTERNAL_K, MESH_DIAMETERS, make_batch
    from oracle import kd_step_ref as O
    dev = gpu_device
    B, crop = 4, 128
    levels = [(crop // 8 // (2 ** i),) * 2 for i in range(4)]
    cells = sum(h * w for h, w in levels)
    counts = [h * w for h, w in levels]
    images, targets = make_batch(B, 77, crop=crop, mixed_classes=mixed)
    tds = [t.as_dict() for t in targets]
    g = torch.Generator().manual_seed(1)
    cls = [torch.randn(B, 15, h, w, generator=g) * 1.5 - 2.0 for h, w in levels]
    reg = [torch.randn(B, 240, h, w, generator=g) * 0.3 for h, w in levels]
    # teacher knowledge in the reference layout: 9 cells per image around the student's decode range
    t_scores = [torch.rand(9, 1, generator=g).expand(9, 8).contiguous() * 0.6 + 0.3 for _ in range(B)]
    t_kps = [torch.rand(9, 8, 2, generator=g) * torch.tensor([200.0, 150.0]) + torch.tensor([220.0, 160.0]) for _ in range(B)]
    t_kps[2] = t_kps[2][:0]; t_scores[2] = t_scores[2][:0]           # one image without teacher cells: skipped
    pred_t = {"post_kp_2d": torch.cat(t_kps), "post_kp_cls": torch.cat(t_scores), "post_pos_per_img": [len(s) for s in t_scores]}
    keys_ref = torch.rand(B * cells, generator=torch.Generator().manual_seed(4))

    def choose(vp, n, im, l, gt):
        off = im * cells + sum(counts[:l])
        return torch.argsort(keys_ref[off + vp], stable=True)[:n]

    # oracle
    cls_r = [c.clone().requires_grad_(True) for c in cls]
    reg_r = [r.clone().requires_grad_(True) for r in reg]
    labels, gt_idx, aux = O.ssc_assign(tds, levels, choose=choose)
    out = O.kd_pose_loss(cls_r, reg_r, tds, (t_scores, t_kps), INTERNAL_K, MESH_DIAMETERS, labels, gt_idx, aux)
    (out["loss_cls"] * 0.1 + out["loss_reg"] + out["loss_kd"] * 5.0).backward()
    # kd6d
    cfg_kd = dict(LOSS_WEIGHT_KD=5.0, LEVEL="pred", GLEVEL="point", GTYPE="sinkhorn", GP=2.0, GBLUR=0.001, GnD=2,
                  WEIGHTED_OT=True, DETACH=False, SCALING=0.5, REACH=0.5)

    class Coder:
        target_type = "3D"

    crit = KDPoseLoss(2.0, 0.25, [32, 64, 128, 256, 512], [8, 16, 32, 64, 128], "SSC", 10, 1.0, 9, INTERNAL_K,
                      MESH_DIAMETERS, Coder(), cfg_kd)
    crit.keys = keys_ref[ref_to_packed_rows(B, levels)].to(dev)
    cls_g = [c.clone().to(dev).requires_grad_(True) for c in cls]
    reg_g = [r.clone().to(dev).requires_grad_(True) for r in reg]
    tg = [t.to(dev) for t in targets]
    l_cls, l_reg, l_kd = crit(cls_g, reg_g, tg, None, {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in pred_t.items()})
    (l_cls * 0.1 + l_reg + l_kd * 5.0).backward()
    torch.cuda.synchronize()
    assert crit.pos_per_img == out["pos_per_img"] and crit.step == 1
    assert float(out["loss_kd"]) > 0
    assert float(l_cls) == pytest.approx(float(out["loss_cls"]), rel=1e-4)
    assert float(l_reg) == pytest.approx(float(out["loss_reg"]), rel=1e-4)
    assert float(l_kd) == pytest.approx(float(out["loss_kd"]), rel=5e-4)
    for a_, b_ in zip(cls_g + reg_g, cls_r + reg_r):
        scale = max(float(b_.grad.abs().max()), 1e-12)
        torch.testing.assert_close(a_.grad.cpu(), b_.grad, rtol=5e-3, atol=5e-3 * scale)
