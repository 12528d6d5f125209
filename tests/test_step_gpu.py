"""End-to-end parity of the HIP KD step (through the C ABI) against
  (a) the golden vectors captured from the imported reference (tests/golden/*.npz), and
  (b) the CPU oracle (oracle/kd_step_ref.py) on the same seeded inputs.

fp32 mode (exact-fp32 MFMA) is held to fp32 tolerances: logits 2e-3 of the tensor scale,
losses rtol 1e-3, gradients 1e-2 relative per parameter tensor, KD loss rtol 1e-3.
bf16 mode is held to bf16 tolerances (logits 6e-2 of scale, losses 5e-2).
"""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# bf16 step vs the oracle's bf16-storage emulation, worst per-tensor gradient-norm deviation at B = 2 (by crop size);
# measured on MI355X in round 3, bounds <= 2x the measurement (profiles/README.md)
BF16_EMU_WORST = {128: 0.4, 256: 0.3}      # measured 0.12 / 0.18 (128x128), 0.04 ... 0.13 (256x256)


def make_cfg(arch, precision):
    from kd6d.arguments.argument import custom_cfg
    with open(os.path.join(ROOT, "configs", "ape.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["RUNTIME"] = {"PRECISION": precision}
    cfg["MODEL"]["BACKBONE"] = arch
    cfg = custom_cfg(cfg)
    cfg["KD"] = dict(LOSS_WEIGHT_KD=5.0, LEVEL="pred", GLEVEL="point", GTYPE="sinkhorn", GP=2.0, GBLUR=0.001, GnD=2,
                     WEIGHTED_OT=True, DETACH=False, SCALING=0.5, REACH=0.5)
    return cfg


def build(arch, precision, seed, dev, cls_bias=None):
    from kd6d import backbone as BB
    from kd6d.models.model_kd import PoseModuleKD
    from oracle import kd_step_ref as O
    m = PoseModuleKD(make_cfg(arch, precision), getattr(BB, arch)())
    sd = O.seeded_state_dict(O.PoseNetRef(arch), seed)
    if cls_bias is not None:
        sd["head.cls_logits.bias"] = torch.as_tensor(cls_bias, dtype=torch.float32)
    m.load_state_dict(sd)
    return m.to(dev)


def packed_to_ref(packed, batch, levels, c):
    """(rows, Cpad) packed -> (B, cells, c) in the reference's image-major order."""
    outs, r = [], 0
    for (h, w) in levels:
        n = batch * h * w
        outs.append(packed[r:r + n].reshape(batch, h * w, -1)[..., :c])
        r += n
    return torch.cat(outs, 1)


def ref_to_packed_rows(batch, levels):
    """index array: packed row -> (image, cell) flat index b*cells + cell."""
    cells = sum(h * w for h, w in levels)
    idx, off = [], 0
    for (h, w) in levels:
        for b in range(batch):
            idx.append(torch.arange(h * w) + b * cells + off)
        off += h * w
    return torch.cat(idx)


def logits_close(got, ref, precision, scale=None):
    """fp32: max |diff| <= 2e-3 * tensor scale.  bf16: relative RMS error <= 8e-2 (bf16 rounding
    of every activation through ~25 conv/norm layers with batch statistics over B=2)."""
    d = np.abs(got - ref)
    if precision == "fp32":
        return d.max() <= 2e-3 * max(scale or 0.0, np.abs(ref).max())
    return float(np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean())) <= 8e-2


def sample(flat_levels_tensor):
    flat = flat_levels_tensor.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, 4096).long()
    return flat[idx].numpy()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["step_tinyh_b2_128", "step_tiny_b2_128", "step_tinyh_b2_256"])
def test_step_against_reference_golden(gpu_device, name, precision):
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.synthetic import make_batch
    z = np.load(os.path.join(G, name + ".npz"))
    dev = gpu_device
    B, crop, arch = int(z["batch"]), int(z["crop"]), str(z["student_arch"])
    images, targets = make_batch(B, int(z["seed"]), crop=crop)
    teacher = build("darknet53", precision, 2, dev, z["teacher_cls_bias"]).eval()
    student = build(arch, precision, 1, dev).train()
    img = ImageList(images.tensors.to(dev), images.sizes)
    tgt = PackedTargets(targets, dev)

    with torch.no_grad():
        pred_t = teacher(img, targets=tgt, is_teacher=True)
    tnet = teacher.net
    cls_t = packed_to_ref(tnet.buf("cls.logits", (tnet.rows, 16), torch.float32).cpu(), B, tnet.levels, 15)
    reg_t = packed_to_ref(tnet.buf("pose.logits", (tnet.rows, 240), torch.float32).cpu(), B, tnet.levels, 240)
    for got, key in ((cls_t, "t_cls"), (reg_t, "t_reg")):
        ref = z[key + "_sample"]
        scale = float(z[key + "_absmean"])
        assert logits_close(sample(got), ref, precision, scale), key
    cnt = pred_t["post_pos_per_img"]
    if precision == "fp32":
        assert cnt == z["teacher_counts"].tolist()
        # order inside an image is score-descending per level in both; compare as sorted sets
        a = pred_t["post_kp_2d"].cpu().numpy().reshape(-1, 16); b = z["teacher_kp"].reshape(-1, 16)
        np.testing.assert_allclose(a[np.lexsort(a.T)], b[np.lexsort(b.T)], rtol=1e-3, atol=0.5)
        np.testing.assert_allclose(np.sort(pred_t["post_kp_cls"].cpu().numpy()[:, 0]), np.sort(z["teacher_cls"][:, 0]),
                                   rtol=1e-3, atol=1e-4)

    # keys that make the SSC kernel pick exactly the reference's random positives
    levels_s = [(crop // 8 // (2 ** i), crop // 8 // (2 ** i)) for i in range(4)]
    perm = ref_to_packed_rows(B, levels_s)
    lab_ref = torch.from_numpy(z["labels"].astype(np.int64)).reshape(-1)[perm]
    student._debug_keys = torch.where(lab_ref > 0, 0.0, 1.0).to(torch.float32).to(dev)
    student.zero_grad()
    _, loss_dict = student(img, targets=tgt, pred_t=pred_t)
    snet = student.net
    assert snet.levels == levels_s
    labels = student.loss_evaluator.ctx["labels"].cpu()
    assert torch.equal(labels.long(), lab_ref), "SSC labels differ from the reference capture"
    cls_s = packed_to_ref(snet.buf("cls.logits", (snet.rows, 16), torch.float32).cpu(), B, snet.levels, 15)
    reg_s = packed_to_ref(snet.buf("pose.logits", (snet.rows, 240), torch.float32).cpu(), B, snet.levels, 240)
    for got, key in ((cls_s, "s_cls_sample"), (reg_s, "s_reg_sample")):
        ref = z[key]
        assert logits_close(sample(got), ref, precision), key
    loss = loss_dict["loss_cls"] * 0.1 + loss_dict["loss_reg"] * 1.0 + loss_dict["loss_kd"] * 5.0
    loss.backward()
    torch.cuda.synchronize()
    rt = 1e-3 if precision == "fp32" else 5e-2
    assert float(loss_dict["loss_cls"]) == pytest.approx(float(z["loss_cls"]), rel=rt)
    assert float(loss_dict["loss_reg"]) == pytest.approx(float(z["loss_reg"]), rel=rt)
    if precision == "fp32":
        assert float(loss_dict["loss_kd"]) == pytest.approx(float(z["loss_kd"]), rel=2e-3)
    else:
        assert float(loss_dict["loss_kd"]) == pytest.approx(float(z["loss_kd"]), rel=0.25)
    # per-parameter gradient norms of the reference backward pass.  The gradient of this
    # random-weight network is ill-conditioned (a 1e-5 relative parameter perturbation moves single
    # parameter-tensor gradients of the ORACLE by up to 30 %), so bf16 is held to
    # aggregate bounds: every tensor norm within 50 %, the norm-weighted mean deviation within 6 %,
    # the global norm within 6 %.
    names = [str(n) for n in z["grad_names"]]
    ref_norms = dict(zip(names, z["grad_norms"]))
    got = {k: p.grad for k, p in student.named_parameters() if p.grad is not None}
    gn_ref = float(z["grad_norm"])
    dev_ = {k: abs(float(got[k].float().norm()) - float(ref_norms[k])) / max(float(ref_norms[k]), 1e-6 * gn_ref)
            for k in names}
    worst = max(dev_.values())
    wmean = sum(dev_[k] * float(ref_norms[k]) ** 2 for k in names) / sum(float(ref_norms[k]) ** 2 for k in names)
    total = float(torch.sqrt(sum((g.float() ** 2).sum() for g in got.values())))
    print("[%s %s] grad-norm deviation: worst %.4f (%s) weighted-mean %.5f total %.5f" % (
        name, precision, worst, max(dev_, key=dev_.get), wmean, abs(total - gn_ref) / gn_ref))
    if precision == "fp32":
        # per-tensor worst case: a tiny-norm BN gain whose gradient is a cancelling sum; its value moves with
        # the order of the fp32 atomics (observed 0.1 % .. 1.3 % run to run), the norm-weighted mean does not
        assert worst <= 2e-2 and wmean <= 1e-3, "per-parameter grad-norm deviation worst %.4f mean %.5f" % (worst, wmean)
        assert total == pytest.approx(gn_ref, rel=1e-2)
    else:
        # bf16 vs the fp32 reference capture: the number format dominates the per-tensor figure (rounding only the
        # input image to bf16 moves the small BatchNorm tensors of the backbone by tens of percent in the ORACLE,
        # profiles/r03_bf16_sensitivity.md), so the golden bounds the aggregates ...
        assert wmean <= 8e-3, (worst, wmean)
        assert total == pytest.approx(gn_ref, rel=6e-3)
        # ... and the per-tensor norms are held against the oracle under bf16-storage emulation (same roundings, fp32
        # arithmetic; the oracle itself is pinned by this golden in fp32), with the reference's SSC labels
        from kd6d.synthetic import INTERNAL_K, MESH_DIAMETERS
        from oracle import kd_step_ref as O
        lab_img = torch.from_numpy(z["labels"].astype(np.int64)).reshape(B, -1)
        cells = lab_img.shape[1]
        counts = [h * w for h, w in levels_s]

        def choose(vp, n, im, l, g):
            off = sum(counts[:l])
            picked = torch.nonzero(lab_img[im, off + vp] > 0).reshape(-1)
            assert len(picked) == n, (len(picked), n)
            return picked

        emu = O.KDStepRef(arch, "darknet53", K=INTERNAL_K, diameters=MESH_DIAMETERS, kd_weight=5.0,
                          teacher_cls_bias=[float(v) for v in z["teacher_cls_bias"]], emulate_bf16=True)
        out, _ = emu.forward_backward(images.tensors, [t.as_dict() for t in targets], choose=choose)
        eg = {k: p.grad for k, p in emu.student.named_parameters() if p.grad is not None}
        egn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in eg.values())))
        # tensors whose gradient the HIP path itself does not reproduce between two runs on the same inputs (float
        # atomics retire in a different order, a bf16 store turns 1e-7 into occasional 2^-9 flips, the small BatchNorm
        # tensors of the backbone amplify them) are reported, not bounded per tensor: see tests/test_fullsize_gpu.py
        twins = []
        for _ in range(2):
            s_ = build(arch, precision, 1, dev).train()
            s_._debug_keys = student._debug_keys
            s_.zero_grad()
            _, ld_ = s_(img, targets=tgt, pred_t=pred_t)
            (ld_["loss_cls"] * 0.1 + ld_["loss_reg"] * 1.0 + ld_["loss_kd"] * 5.0).backward()
            torch.cuda.synchronize()
            twins.append({k: float(p.grad.float().norm()) for k, p in s_.named_parameters() if p.grad is not None})
        noisy = {k: abs(a - twins[1][k]) / max(a, twins[1][k], 1e-30) for k, a in twins[0].items()}
        noisy = {k: v for k, v in noisy.items() if v > 0.03}
        # what is set aside must be a small part of the update: < 10 % of the squared gradient norm at B = 2 (measured
        # 3.4 % / 3.2 % at 128x128, < 1 % at 256x256; at B = 16 the bound is 0.1 %, tests/test_fullsize_gpu.py)
        assert sum(float(eg[k].norm()) ** 2 for k in noisy) <= 1e-1 * egn ** 2, noisy
        dev_e = {k: abs(float(got[k].float().norm()) - float(eg[k].norm())) / max(float(eg[k].norm()), 1e-6 * egn)
                 for k in eg if k not in noisy}
        worst_e = max(dev_e.values())
        print("[%s bf16] vs bf16-storage emulation: worst %.4f (%s), %d tensors not reproducible between two runs %s, losses %s" % (
            name, worst_e, max(dev_e, key=dev_e.get), len(noisy), {k.replace("backbone.features.", ""): round(v, 3) for k, v in noisy.items()},
            {k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}))
        for k in ("loss_cls", "loss_reg"):
            assert float(loss_dict[k]) == pytest.approx(float(out[k]), rel=1e-2)
        assert float(loss_dict["loss_kd"]) == pytest.approx(float(out["loss_kd"]), rel=0.1)
        assert worst_e <= BF16_EMU_WORST[crop], (worst_e, max(dev_e, key=dev_e.get))
        assert total == pytest.approx(egn, rel=3e-3)


def test_step_against_oracle_fp32_with_optimizer(gpu_device):
    """Full step incl. fused clip+AdamW+OneCycle vs the CPU oracle, element-wise on every tensor.
    Iteration 2 restarts from the oracle's parameters: Adam's first update is lr*sign(g), so rounding
    noise in near-zero gradients becomes O(lr) parameter differences, and the gradient of this network
    moves by percents under 1e-5 parameter perturbations (measured on the oracle itself)."""
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.optim import FusedClipAdamW
    from kd6d.synthetic import INTERNAL_K, MESH_DIAMETERS, make_batch
    from oracle import kd_step_ref as O
    dev = gpu_device
    B, crop, arch = 2, 128, "darknet_tiny_h"
    bias = [1.0] + [-6.0] * 14
    images, targets = make_batch(B, 21, crop=crop)
    teacher = build("darknet53", "fp32", 2, dev, bias).eval()
    student = build(arch, "fp32", 1, dev).train()
    opt = FusedClipAdamW(student, lr=1e-3, weight_decay=1e-4, eps=1e-8, max_norm=1.0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, 1e-3, 10100, pct_start=0.05, cycle_momentum=False,
                                                anneal_strategy="linear")
    step = O.KDStepRef(arch, "darknet53", K=INTERNAL_K, diameters=MESH_DIAMETERS, kd_weight=5.0, teacher_cls_bias=bias)
    img = ImageList(images.tensors.to(dev), images.sizes)
    tgt = PackedTargets(targets, dev)
    levels = [(crop // 8 // (2 ** i),) * 2 for i in range(4)]
    perm = ref_to_packed_rows(B, levels)
    cells = sum(h * w for h, w in levels)
    for it in range(2):
        keys_ref = torch.rand(B * cells, generator=torch.Generator().manual_seed(100 + it))
        student._debug_keys = keys_ref[perm].to(dev)
        counts = [h * w for h, w in levels]

        def choose(vp, n, im, l, g, keys_ref=keys_ref):
            off = im * cells + sum(counts[:l])
            k = keys_ref[off + vp]
            return torch.argsort(k, stable=True)[:n]

        if it > 0:
            student.load_state_dict(step.student.state_dict())
        pre = {k: v.clone() for k, v in step.student.state_dict().items()}
        res, ex = step.step(images.tensors, [t.as_dict() for t in targets], choose=choose, return_extras=True)
        ref_grads = {k: p.grad.clone() for k, p in step.student.named_parameters() if p.grad is not None}
        with torch.no_grad():
            pred_t = teacher(img, targets=tgt, is_teacher=True)
        student.zero_grad()
        _, ld = student(img, targets=tgt, pred_t=pred_t)
        loss = ld["loss_cls"] * 0.1 + ld["loss_reg"] * 1.0 + ld["loss_kd"] * 5.0
        loss.backward()
        got_grads = {k: p.grad.detach().clone().cpu() for k, p in student.named_parameters() if p.grad is not None}
        opt.step(); sched.step()
        torch.cuda.synchronize()
        assert float(ld["loss_cls"]) == pytest.approx(res["loss_cls"], rel=1e-3)
        assert float(ld["loss_reg"]) == pytest.approx(res["loss_reg"], rel=1e-3)
        assert float(ld["loss_kd"]) == pytest.approx(res["loss_kd"], rel=2e-3)
        assert float(opt.grad_norm()) == pytest.approx(res["grad_norm"], rel=5e-3)
        clip = min(1.0, 1.0 / (res["grad_norm"] + 1e-6))
        for k, g in ref_grads.items():
            r = g / clip
            tol = 2e-2 * float(r.abs().max()) + 1e-6 * res["grad_norm"]
            assert float((got_grads[k] - r).abs().max()) <= tol, (it, k)
        sd = student.state_dict()
        if it == 0:      # Adam state is identical only on the first step (see docstring)
            for k, v in step.student.state_dict().items():
                if k.endswith("num_batches_tracked"):
                    assert int(sd[k]) == int(v)
                    continue
                torch.testing.assert_close(sd[k].cpu(), v, rtol=2e-3, atol=2e-4,
                                           msg=lambda m, k=k: "%s (iter %d): %s" % (k, it, m))


def test_mixed_class_batch_against_oracle(gpu_device):
    """BASELINE config 4 semantics: the LINEMOD classes mixed in one batch, darknet53 -> darknet_tiny.
    The reference broadcasts pred_cls[..., unique(cls)] (kd_loss.py:43,83), which is only defined for a
    single-class batch; the HIP path (and the oracle) gather the OT weight per cell, pred_cls[i, cls_i], and
    use each cell's own mesh diameter and 3D box in the object-space loss.  fp32: losses 1e-3, global
    gradient norm 5e-3; per parameter tensor the gradient NORM within 5 % and the norm-weighted mean deviation
    within 1e-3 (single elements of this random-weight, batch-normalised network move by several percent with
    the order of the fp32 atomics -- which tensor is worst changes from run to run)."""
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.synthetic import INTERNAL_K, LINEMOD_CLASSES, MESH_DIAMETERS, make_batch
    from oracle import kd_step_ref as O
    dev = gpu_device
    B, crop, arch = 5, 128, "darknet_tiny"
    bias = [1.0] + [-6.0] * 14
    images, targets = make_batch(B, 33, crop=crop, mixed_classes=True)
    assert len({int(t.class_ids[0]) for t in targets}) == B and LINEMOD_CLASSES[2] == 3   # 5 different classes, id gap
    teacher = build("darknet53", "fp32", 2, dev, bias).eval()
    student = build(arch, "fp32", 1, dev).train()
    step = O.KDStepRef(arch, "darknet53", K=INTERNAL_K, diameters=MESH_DIAMETERS, kd_weight=5.0, teacher_cls_bias=bias)
    img = ImageList(images.tensors.to(dev), images.sizes)
    tgt = PackedTargets(targets, dev)
    levels = [(crop // 8 // (2 ** i),) * 2 for i in range(4)]
    perm = ref_to_packed_rows(B, levels)
    cells = sum(h * w for h, w in levels)
    counts = [h * w for h, w in levels]
    keys_ref = torch.rand(B * cells, generator=torch.Generator().manual_seed(7))
    student._debug_keys = keys_ref[perm].to(dev)

    def choose(vp, n, im, l, g):
        off = im * cells + sum(counts[:l])
        return torch.argsort(keys_ref[off + vp], stable=True)[:n]

    res = step.step(images.tensors, [t.as_dict() for t in targets], choose=choose)
    ref_grads = {k: p.grad.clone() for k, p in step.student.named_parameters() if p.grad is not None}
    with torch.no_grad():
        pred_t = teacher(img, targets=tgt, is_teacher=True)
    student.zero_grad()
    _, ld = student(img, targets=tgt, pred_t=pred_t)
    (ld["loss_cls"] * 0.1 + ld["loss_reg"] * 1.0 + ld["loss_kd"] * 5.0).backward()
    torch.cuda.synchronize()
    assert res["loss_kd"] > 0, "the KD term must be active"
    assert float(ld["loss_cls"]) == pytest.approx(res["loss_cls"], rel=1e-3)
    assert float(ld["loss_reg"]) == pytest.approx(res["loss_reg"], rel=1e-3)
    assert float(ld["loss_kd"]) == pytest.approx(res["loss_kd"], rel=2e-3)
    got = {k: p.grad.detach().cpu() for k, p in student.named_parameters() if p.grad is not None}
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in got.values())))
    assert gn == pytest.approx(res["grad_norm"], rel=5e-3)
    clip = min(1.0, 1.0 / (res["grad_norm"] + 1e-6))
    num = den = 0.0
    for k, g in ref_grads.items():
        rn = float((g / clip).norm())
        dev_k = abs(float(got[k].norm()) - rn) / max(rn, 1e-6 * res["grad_norm"])
        assert dev_k <= 5e-2, (k, dev_k)
        num += dev_k * rn ** 2
        den += rn ** 2
    assert num / den <= 1e-3


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graph_replay_matches_eager_steps(gpu_device, precision):
    """hipGraph replay (kd6d/graph.py: 2 captured graphs + device-resident lr/bias corrections) trains
    exactly like the eager launch sequence: same losses and same parameters after 4 steps with a
    moving OneCycle learning rate and changing batches.  (The capture warm-up must not train.)"""
    from kd6d.graph import GraphedKDStep
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.optim import FusedClipAdamW
    from kd6d.synthetic import make_batch
    dev = gpu_device
    B, crop, arch = 2, 64, "darknet_tiny_h"
    bias = [1.0] + [-6.0] * 14
    teacher = build("darknet53", precision, 2, dev, bias).eval()
    batches = []
    for i in range(2):
        images, targets = make_batch(B, 10 + i, crop=crop)
        batches.append((ImageList(images.tensors.to(dev), images.sizes), PackedTargets(targets, dev)))
    rows = B * sum((crop // 8 // 2 ** i) ** 2 for i in range(4))
    keys = torch.rand(rows, generator=torch.Generator().manual_seed(3)).to(dev)

    def run(graph, pipeline=False):
        student = build(arch, precision, 1, dev).train()
        student._debug_keys = keys
        opt = FusedClipAdamW(student, lr=1e-3)
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, 1e-3, 40, pct_start=0.25, cycle_momentum=False,
                                                    anneal_strategy="linear")
        gs = GraphedKDStep(teacher, student, opt, (0.1, 1.0, 5.0), pipeline=pipeline) if graph else None
        hist = []
        if pipeline:
            assert gs(*batches[0]) is None            # priming call: teacher only
        for it in range(4):
            img, tgt = batches[it % 2]
            if pipeline:                              # call k: teacher on batch k+1, student step on batch k
                ld = gs(*batches[(it + 1) % 2]) if it < 3 else gs.flush()
            elif gs is not None:
                ld = gs(img, tgt)
            else:
                student.zero_grad()
                with torch.no_grad():
                    pred_t = teacher(img, targets=tgt, is_teacher=True)
                _, ld = student(img, targets=tgt, pred_t=pred_t)
                (ld["loss_cls"] * 0.1 + ld["loss_reg"] + ld["loss_kd"] * 5.0).backward()
                opt.step()
            sched.step()
            hist.append([float(ld[k]) for k in ("loss_cls", "loss_reg", "loss_kd")] + [float(opt.grad_norm())])
        torch.cuda.synchronize()
        return np.array(hist), student.net.store.params.cpu().numpy().copy(), opt.steps

    h_e, p_e, n_e = run(False)
    h_g, p_g, n_g = run(True)
    p0 = build(arch, precision, 1, dev).net.store.params.cpu().numpy()
    assert n_e == n_g == 4
    assert np.isfinite(h_e).all() and h_e[:, 2].max() > 0, "the KD term must be active in this test"
    # step 1 sees identical weights in both runs (the capture warm-up is rolled back): only the order of
    # the fp32 atomics differs.  Later steps drift apart the way two eager runs do: AdamW's first
    # updates are +-lr * sign(g), which amplifies that noise on near-zero gradients.
    tol = 1e-3 if precision == "fp32" else 2e-2
    # the global grad norm sums thousands of atomically accumulated terms: fp32 moves ~1e-3 run to run, bf16 ~2e-2
    gtol = np.array([tol, tol, tol, 5e-3 if precision == "fp32" else 6e-2])
    assert (np.abs(h_g[0] - h_e[0]) <= np.abs(h_e[0]) * gtol).all(), (h_g[0], h_e[0])
    np.testing.assert_allclose(h_g[1:, :3], h_e[1:, :3], rtol=0.1)
    d_e, d_g = np.abs(p_e - p0), np.abs(p_g - p0)
    assert d_e.max() > 1e-4, "the eager run did not train"
    # same schedule: the per-parameter travel after 4 steps (~ sum of the 4 learning rates) agrees
    assert abs(d_g.mean() / d_e.mean() - 1.0) < 0.02, (d_g.mean(), d_e.mean())
    assert np.mean(np.abs(p_g - p_e) > 1e-3) < 0.15
    # cross-step pipeline (teacher of batch k+1 beside the student step of batch k): same training sequence
    h_p, p_p, n_p = run(True, pipeline=True)
    assert n_p == 4
    assert (np.abs(h_p[0] - h_e[0]) <= np.abs(h_e[0]) * gtol).all(), (h_p[0], h_e[0])
    np.testing.assert_allclose(h_p[1:, :3], h_e[1:, :3], rtol=0.1)
    assert abs(np.abs(p_p - p0).mean() / d_e.mean() - 1.0) < 0.02


def test_pipeline_restarts_after_flush(gpu_device):
    """flush() trains on the batch that was still waiting; a call after it starts a NEW pipeline (teacher only)
    instead of training on that batch a second time: 3 batches + flush = 3 optimiser steps, 2 more + flush = 5, and
    the first loss of the second pipeline is the loss of ITS first batch at the weights of that moment (recomputed
    eagerly on a copy of the student)."""
    from kd6d.graph import GraphedKDStep
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.optim import FusedClipAdamW
    from kd6d.synthetic import make_batch
    dev = gpu_device
    B, crop = 2, 64
    teacher = build("darknet53", "fp32", 2, dev, [1.0] + [-6.0] * 14).eval()
    batches = []
    for i in range(3):
        images, targets = make_batch(B, 20 + i, crop=crop)
        batches.append((ImageList(images.tensors.to(dev), images.sizes), PackedTargets(targets, dev)))
    rows = B * sum((crop // 8 // 2 ** i) ** 2 for i in range(4))
    student = build("darknet_tiny_h", "fp32", 1, dev).train()
    student._debug_keys = torch.rand(rows, generator=torch.Generator().manual_seed(3)).to(dev)
    opt = FusedClipAdamW(student, lr=1e-4)
    gs = GraphedKDStep(teacher, student, opt, (0.1, 1.0, 5.0), pipeline=True)
    assert gs(*batches[0]) is None
    gs(*batches[1]); gs(*batches[2])
    gs.flush()
    assert opt.steps == 3 and gs.flush() is None
    assert gs(*batches[1]) is None and opt.steps == 3          # new pipeline: teacher only
    # what the next student step must report: the losses of batches[1] at the current weights, computed eagerly on a copy
    twin = build("darknet_tiny_h", "fp32", 1, dev).train()
    twin.load_state_dict(student.state_dict())
    twin._debug_keys = student._debug_keys
    with torch.no_grad():
        pred_t = teacher(batches[1][0], targets=batches[1][1], is_teacher=True)
    _, want = twin(batches[1][0], targets=batches[1][1], pred_t=pred_t)
    want = [float(want[k]) for k in ("loss_cls", "loss_reg", "loss_kd")]
    ld = gs(*batches[0])                                       # student step on batches[1]
    got = [float(ld[k]) for k in ("loss_cls", "loss_reg", "loss_kd")]
    gs.flush()
    assert opt.steps == 5
    np.testing.assert_allclose(got, want, rtol=1e-4)
    assert got[2] > 0, "the KD term must be active in this test"


@pytest.mark.parametrize("group", [2, 3])
def test_grouped_teacher_matches_sequential_steps(gpu_device, group):
    """GroupedTeacherKDStep (the teacher over the batches of `group` steps in one pass, cut into `group` graph segments,
    kd6d/graph.py) trains exactly like the strictly sequential replayed step: 10 batches in, 2 * group priming calls that
    return None, then the loss of batch k - 2 * group per call, flush() once per batch still pending (the last period is
    a partial one: 10 is not a multiple of 3; group 2 drains from a period boundary) and None afterwards; every step's three losses and the final parameters agree with
    GraphedKDStep(pipeline=False) over the same batches (fp32; the teacher's layers tile 2-3x the rows differently, so
    its cells agree to rounding, not bitwise).  A call after the drain starts a new pipeline."""
    from kd6d.graph import GraphedKDStep, GroupedTeacherKDStep
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.optim import FusedClipAdamW
    from kd6d.synthetic import make_batch
    dev = gpu_device
    B, crop, n = 2, 64, 10
    batches = []
    for i in range(n):
        images, targets = make_batch(B, 40 + i, crop=crop)
        batches.append((ImageList(images.tensors.to(dev), images.sizes), PackedTargets(targets, dev)))
    rows = B * sum((crop // 8 // 2 ** i) ** 2 for i in range(4))
    keys = torch.rand(rows, generator=torch.Generator().manual_seed(3)).to(dev)
    names = ("loss_cls", "loss_reg", "loss_kd")

    def make(grouped):
        teacher = build("darknet53", "fp32", 2, dev, [1.0] + [-6.0] * 14).eval()
        student = build("darknet_tiny_h", "fp32", 1, dev).train()
        student._debug_keys = keys
        opt = FusedClipAdamW(student, lr=1e-4)
        gs = (GroupedTeacherKDStep(teacher, student, opt, (0.1, 1.0, 5.0), group=group) if grouped
              else GraphedKDStep(teacher, student, opt, (0.1, 1.0, 5.0), pipeline=False))
        return gs, student, opt

    gs, student, opt = make(False)
    want = []
    for b in batches:
        ld = gs(*b)
        want.append([float(ld[k]) for k in names])
    p_want = student.net.store.params.detach().clone()
    assert opt.steps == n

    gs, student, opt = make(True)
    got = []
    for i, b in enumerate(batches):
        ld = gs(*b)
        assert (ld is None) == (i < 2 * group)
        if ld is not None:
            got.append([float(ld[k]) for k in names])
    assert opt.steps == n - 2 * group and gs.pending_steps == 2 * group
    with pytest.raises(RuntimeError):
        # a partly drained pipeline takes no new batches
        gs.flush(); got.append([float(gs.losses[k]) for k in names]); gs(*batches[0])
    while True:
        ld = gs.flush()
        if ld is None:
            break
        got.append([float(ld[k]) for k in names])
    assert opt.steps == n and len(got) == n and gs.pending_steps == 0 and not gs.pending
    # the first steps agree to rounding; ten AdamW steps on, the atomic-order noise of either run has grown to 0.5-1 %
    # of a loss (two runs of the SEQUENTIAL step differ by as much: lr * sign(g) flips where g is near zero)
    np.testing.assert_allclose(np.array(got[:3]), np.array(want[:3]), rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(np.array(got), np.array(want), rtol=3e-2, atol=1e-6)
    assert np.array(want)[:, 2].min() > 0, "the KD term must be active in this test"
    d = (student.net.store.params.detach() - p_want).abs()
    assert float(d.max()) <= 2.1e-3 and float(d.mean()) <= 5e-5, (float(d.max()), float(d.mean()))   # 10 steps of lr 1e-4: a
    # parameter moves by <= 1e-3, so two runs are at most 2e-3 apart -- where the sign of a near-zero gradient differs
    assert gs.teacher_passes >= n // group and len(gs.segment_ms) == group
    assert gs(*batches[0]) is None and opt.steps == n      # a new pipeline: loading only
    assert gs.pending_steps == 1


def test_grouped_teacher_prepare_leaves_training_state_untouched(gpu_device):
    """GroupedTeacherKDStep.prepare(sample batch) records every graph up front (what a data-parallel run does before its
    first RCCL communicator exists): afterwards no optimiser step has been taken, parameters / optimiser moments / BatchNorm
    buffers are bitwise what they were, the pipeline is empty, and the first trained batch reports the loss the eager step
    reports for it at the initial weights."""
    from kd6d.graph import GroupedTeacherKDStep
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.optim import FusedClipAdamW
    from kd6d.synthetic import make_batch
    dev = gpu_device
    B, crop, group = 2, 64, 2
    batches = []
    for i in range(5):
        images, targets = make_batch(B, 60 + i, crop=crop)
        batches.append((ImageList(images.tensors.to(dev), images.sizes), PackedTargets(targets, dev)))
    rows = B * sum((crop // 8 // 2 ** i) ** 2 for i in range(4))
    keys = torch.rand(rows, generator=torch.Generator().manual_seed(3)).to(dev)
    teacher = build("darknet53", "fp32", 2, dev, [1.0] + [-6.0] * 14).eval()
    student = build("darknet_tiny_h", "fp32", 1, dev).train()
    student._debug_keys = keys
    opt = FusedClipAdamW(student, lr=1e-4)
    st = student.net.store
    before = (st.params.clone(), st.bufs.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), student._nbt.clone())
    gs = GroupedTeacherKDStep(teacher, student, opt, (0.1, 1.0, 5.0), group=group)
    gs.prepare(*batches[4])                                   # a sample that is NOT the first training batch
    torch.cuda.synchronize()
    assert opt.steps == 0 and gs.pending_steps == 0 and gs.teacher_passes == 0
    assert gs.g_student is not None and gs.g_teacher is not None and len(gs.segment_ms) == group
    after = (st.params, st.bufs, opt.exp_avg, opt.exp_avg_sq, student._nbt)
    for a, b in zip(before, after):
        assert torch.equal(a, b)
    gs.prepare(*batches[4])                                   # idempotent
    # the eager step's losses for batch 0 at the (untouched) initial weights, on a copy of the student
    twin = build("darknet_tiny_h", "fp32", 1, dev).train()
    twin.load_state_dict(student.state_dict())
    twin._debug_keys = keys
    with torch.no_grad():
        pred_t = teacher(batches[0][0], targets=batches[0][1], is_teacher=True)
    _, want = twin(batches[0][0], targets=batches[0][1], pred_t=pred_t)
    want = [float(want[k].detach()) for k in ("loss_cls", "loss_reg", "loss_kd")]
    out = [gs(*batches[i % 4]) for i in range(2 * group + 1)]
    assert all(o is None for o in out[:2 * group]) and out[-1] is not None
    got = [float(out[-1][k]) for k in ("loss_cls", "loss_reg", "loss_kd")]
    np.testing.assert_allclose(got, want, rtol=1e-3)
    assert opt.steps == 1


def test_replayed_step_draws_new_sampling_keys(gpu_device):
    """Without pinned keys the replayed step draws its SSC sampling keys on the device from (seed, step counter, cell):
    every replay sees new keys in [0, 1), the same seed reproduces the sequence, and the step stays finite."""
    from kd6d.graph import GraphedKDStep
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.optim import FusedClipAdamW
    from kd6d.synthetic import make_batch
    dev = gpu_device
    teacher = build("darknet53", "bf16", 2, dev, [1.0] + [-6.0] * 14).eval()
    images, targets = make_batch(2, 31, crop=64)
    batch = (ImageList(images.tensors.to(dev), images.sizes), PackedTargets(targets, dev))

    def run():
        torch.manual_seed(1234)
        student = build("darknet_tiny_h", "bf16", 1, dev).train()
        opt = FusedClipAdamW(student, lr=1e-4)
        gs = GraphedKDStep(teacher, student, opt, (0.1, 1.0, 5.0), pipeline=False)
        keys, losses = [], []
        for _ in range(4):
            ld = gs(*batch)
            torch.cuda.synchronize()
            (kb,) = list(student.loss_evaluator._keys.values())
            keys.append(kb.clone())
            losses.append([float(ld[k]) for k in ("loss_cls", "loss_reg", "loss_kd")])
        return keys, np.array(losses)

    keys, losses = run()
    assert np.isfinite(losses).all()
    for k in keys:
        assert float(k.min()) >= 0.0 and float(k.max()) < 1.0 and 0.3 < float(k.mean()) < 0.7
    for a, b in zip(keys[:-1], keys[1:]):
        assert float((a == b).float().mean()) < 1e-2
    keys2, _ = run()
    for a, b in zip(keys, keys2):
        assert torch.equal(a, b)


def test_eval_between_graph_replays_sees_current_weights(gpu_device):
    """A replayed optimiser graph changes the weights without passing through Python: the eval-mode BatchNorm
    scale/shift cached by the previous validation must not survive it (every VAL_FREQ steps train_kd.py validates
    between replays).  The eval forward after further replays equals the eval forward of a fresh module loaded with
    the current state_dict."""
    from kd6d.graph import GraphedKDStep
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.optim import FusedClipAdamW
    from kd6d.synthetic import make_batch
    dev = gpu_device
    B, crop, arch = 2, 64, "darknet_tiny_h"
    teacher = build("darknet53", "fp32", 2, dev, [1.0] + [-6.0] * 14).eval()
    student = build(arch, "fp32", 1, dev).train()
    opt = FusedClipAdamW(student, lr=1e-2)
    images, targets = make_batch(B, 10, crop=crop)
    img, tgt = ImageList(images.tensors.to(dev), images.sizes), PackedTargets(targets, dev)
    gs = GraphedKDStep(teacher, student, opt, (0.1, 1.0, 5.0))

    def eval_logits(m):
        m.eval()
        cls, reg = m.net.forward(img.tensors)
        out = (cls.clone(), reg.clone())
        m.train()
        return out

    gs(img, tgt)
    first = eval_logits(student)
    for _ in range(3):
        gs(img, tgt)
    second = eval_logits(student)
    fresh = build(arch, "fp32", 1, dev)
    fresh.load_state_dict(student.state_dict())
    want = eval_logits(fresh)
    torch.cuda.synchronize()
    # (GroupNorm statistics are accumulated with float atomics: equal up to their summation order)
    assert float((first[1] - second[1]).abs().max()) > 1e-2, "the replayed steps did not train"
    torch.testing.assert_close(second[0], want[0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(second[1], want[1], rtol=1e-4, atol=1e-4)


def test_data_parallel_step_semantics_two_shards(gpu_device):
    """What N = 2 ranks compute (libs/train_libs.py:117,272): each rank runs the step on its shard of
    the global batch (per-rank loss sums / KD mean, local BatchNorm statistics), the flat gradient buckets are
    averaged, and every rank applies clip + AdamW with lr = BASE_LR / N.  Emulated on one GPU -- the HIP step on the
    two half-batches one after the other, bucket mean, fused optimiser -- against the oracle doing the same."""
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.optim import FusedClipAdamW
    from kd6d.synthetic import INTERNAL_K, MESH_DIAMETERS, make_batch
    from oracle import kd_step_ref as O
    dev = gpu_device
    N, Bs, crop, arch = 2, 2, 128, "darknet_tiny_h"
    bias = [1.0] + [-6.0] * 14
    teacher = build("darknet53", "fp32", 2, dev, bias).eval()
    student = build(arch, "fp32", 1, dev).train()
    base_lr = 1e-3 / N
    opt = FusedClipAdamW(student, lr=base_lr, weight_decay=1e-4, eps=1e-8, max_norm=1.0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, base_lr, 10100, pct_start=0.05, cycle_momentum=False,
                                                anneal_strategy="linear")
    ref = O.KDStepRef(arch, "darknet53", K=INTERNAL_K, diameters=MESH_DIAMETERS, kd_weight=5.0, teacher_cls_bias=bias,
                      n_gpu=N)
    levels = [(crop // 8 // (2 ** i),) * 2 for i in range(4)]
    cells = sum(h * w for h, w in levels)
    counts = [h * w for h, w in levels]
    st = student.net.store
    buckets, ref_buckets, losses, ref_losses = [], [], [], []
    for r in range(N):                                   # seed = 1000 * rank + step, as bench.py / train_kd.py shard
        images, targets = make_batch(Bs, 1000 * r, crop=crop)
        keys_ref = torch.rand(Bs * cells, generator=torch.Generator().manual_seed(50 + r))
        student._debug_keys = keys_ref[ref_to_packed_rows(Bs, levels)].to(dev)

        def choose(vp, n, im, l, g, keys_ref=keys_ref):
            off = im * cells + sum(counts[:l])
            return torch.argsort(keys_ref[off + vp], stable=True)[:n]

        out, _ = ref.forward_backward(images.tensors, [t.as_dict() for t in targets], choose=choose)
        ref_buckets.append({k: p.grad.clone() for k, p in ref.student.named_parameters() if p.grad is not None})
        ref_losses.append([float(out[k]) for k in ("loss_cls", "loss_reg", "loss_kd")])
        img, tgt = ImageList(images.tensors.to(dev), images.sizes), PackedTargets(targets, dev)
        student.zero_grad()
        with torch.no_grad():
            pred_t = teacher(img, targets=tgt, is_teacher=True)
        _, ld = student(img, targets=tgt, pred_t=pred_t)
        (ld["loss_cls"] * 0.1 + ld["loss_reg"] * 1.0 + ld["loss_kd"] * 5.0).backward()
        buckets.append(st.grads[:st.n_train].clone())
        losses.append([float(ld[k]) for k in ("loss_cls", "loss_reg", "loss_kd")])
    np.testing.assert_allclose(np.array(losses), np.array(ref_losses), rtol=2e-3)
    assert abs(losses[0][0] - losses[1][0]) > 1e-3 * abs(losses[0][0]), "the two shards must differ"
    # the collective: mean of the rank buckets (kd6d_comm_allreduce(mean) / ReduceOp.AVG)
    st.grads[:st.n_train].copy_((buckets[0] + buckets[1]) / N)
    for k, p in ref.student.named_parameters():
        if k in ref_buckets[0]:
            p.grad = (ref_buckets[0][k] + ref_buckets[1][k]) / N
    ref_mean = {k: p.grad.clone() for k, p in ref.student.named_parameters() if p.grad is not None}
    got_mean = {k: p.grad.detach().clone().cpu() for k, p in student.named_parameters() if p.grad is not None}
    gn_ref = ref.optimizer_step()
    opt.step(); sched.step()
    torch.cuda.synchronize()
    assert opt.param_groups[0]["lr"] == pytest.approx(ref.opt.param_groups[0]["lr"], rel=1e-12)
    assert float(opt.grad_norm()) == pytest.approx(gn_ref, rel=5e-3)
    # B = 2 per shard on 16x16 ... 2x2 maps: batch statistics over a few hundred samples make single tensors of this
    # random-weight network ill-conditioned, so element-wise bounds are loose and the
    # direction / norm bounds carry the check
    num = den = dot = 0.0
    for k, r in ref_mean.items():
        tol = 0.15 * float(r.abs().max()) + 1e-6 * gn_ref
        assert float((got_mean[k] - r).abs().max()) <= tol, k
        rn = float(r.norm())
        num += abs(float(got_mean[k].norm()) - rn) * rn
        den += rn ** 2
        dot += float((got_mean[k].double() * r.double()).sum())
    got_norm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in got_mean.values())))
    assert num / den <= 2e-3, num / den
    assert dot / (got_norm * gn_ref) >= 1.0 - 1e-3
    sd = student.state_dict()
    for k, v in ref.student.state_dict().items():
        if k.endswith("num_batches_tracked") or "running_" in k:
            continue                                      # BN statistics are per rank: not part of the exchange
        torch.testing.assert_close(sd[k].cpu(), v, rtol=2e-3, atol=2e-4, msg=lambda m, k=k: "%s: %s" % (k, m))
