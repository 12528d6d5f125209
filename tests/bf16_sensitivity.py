"""How much of a bf16 step's per-tensor gradient deviation from the fp32 oracle is the NUMBER FORMAT?

CPU only, test infrastructure (run by hand: `python tests/bf16_sensitivity.py > profiles/r03_bf16_sensitivity.md`).
The fp32 oracle (oracle/kd_step_ref.py) runs BASELINE config 2's step (B = 16, 256x256, darknet53 -> darknet_tiny_h)
several times with ONE class of values passed through bf16 and everything else in fp32, and the gradients of the
student are compared with the unperturbed run.  Rounding points are switched on with torch hooks, the arithmetic stays
torch's fp32: no kernel of the HIP path is involved, so what the table shows is a property of the network at its
(seeded) initialisation, not of an implementation.
"""
import os
import sys

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kd-6d-pose-adlp_amd"))

from kd6d.synthetic import INTERNAL_K, MESH_DIAMETERS, make_batch  # noqa: E402
from oracle import kd_step_ref as O  # noqa: E402

B = 16
BIAS = [1.0] + [-6.0] * 14
WATCH = ["backbone.features.stage1.unit1.bn.weight", "backbone.features.stage1.unit1.bn.bias",
         "backbone.features.stage2.unit1.bn.weight", "backbone.features.stage2.unit1.conv.weight",
         "backbone.features.stage3.unit2.bn.weight", "backbone.features.stage4.unit2.conv.weight",
         "fpn.out_convs.2.weight", "head.pose_tower.0.weight", "head.pose_pred.weight"]


def main():
    torch.manual_seed(0)
    ref = O.KDStepRef("darknet_tiny_h", "darknet53", K=INTERNAL_K, diameters=MESH_DIAMETERS, kd_weight=5.0,
                      teacher_cls_bias=BIAS)
    images, targets = make_batch(B, 41)
    td = [t.as_dict() for t in targets]
    levels = [(32, 32), (16, 16), (8, 8), (4, 4)]
    counts = [h * w for h, w in levels]
    cells = sum(counts)
    keys = torch.rand(B * cells, generator=torch.Generator().manual_seed(17))

    def choose(vp, n, im, l, g):
        off = im * cells + sum(counts[:l])
        return torch.argsort(keys[off + vp], stable=True)[:n]

    tk = ref.teacher_knowledge(images.tensors, td)          # the teacher is not perturbed: same cells in every run
    ref.teacher_knowledge = lambda im, t: tk
    student = ref.student
    cfg = {}

    def rb(g):
        return g.bfloat16().float()

    def ste(x):
        return x + (x.bfloat16().float() - x).detach()

    def in_scope(name):
        return cfg["scope"] == "all" or any(name.startswith(p) for p in cfg["scope"])

    for name, m in student.named_modules():
        if isinstance(m, nn.Conv2d):
            def conv_hook(mod, inp, out, name=name):
                if cfg["bwd"] and in_scope(name) and out.requires_grad:
                    out.register_hook(rb)                   # gradient w.r.t. a conv output
            m.register_forward_hook(conv_hook)
        if isinstance(m, (O.ConvBlockRef, nn.ReLU, nn.MaxPool2d)):
            def act_hook(mod, inp, out, name=name):
                if not in_scope(name):
                    return None
                o = ste(out) if cfg["fwd"] else out         # a stored activation
                if cfg["bwd"] and o.requires_grad:
                    o.register_hook(rb)                     # gradient w.r.t. a stored activation
                return o
            m.register_forward_hook(act_hook)
    orig = {k: v.detach().clone() for k, v in student.state_dict().items()}

    def run(fwd=False, bwd=False, scope="all", wts=False, inp=False, noise=0.0):
        cfg.update(fwd=fwd, bwd=bwd, scope=scope)
        student.load_state_dict({k: (v.bfloat16().float() if (wts and v.dim() == 4) else v) for k, v in orig.items()})
        im = images.tensors.bfloat16().float() if inp else images.tensors
        if noise:
            im = im * (1.0 + noise * torch.randn(im.shape, generator=torch.Generator().manual_seed(5)))
        ref.forward_backward(im, td, choose=choose)
        return {k: p.grad.clone() for k, p in student.named_parameters() if p.grad is not None}

    g0 = run()
    total0 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g0.values())))
    rows = []

    def rep(tag, g):
        el = {k: float((g[k] - g0[k]).norm() / g0[k].norm()) for k in g0 if float(g0[k].norm()) > 0}
        nm = {k: abs(float(g[k].norm()) - float(g0[k].norm())) / float(g0[k].norm()) for k in el}
        total = float(torch.sqrt(sum((g[k].double() ** 2).sum() for k in g)))
        cos = float(sum((g[k].double() * g0[k].double()).sum() for k in g0)) / (total * total0)
        rows.append("| %s | %s | %.3f (%s) | %.3f | %.1e | %.1e |" % (
            tag, " ".join("%.3f" % el[k] for k in WATCH), max(nm.values()), max(nm, key=nm.get).replace("backbone.features.", ""),
            max(el.values()), abs(total - total0) / total0, 1.0 - cos))

    rep("gradients (activation + conv-output gradients) -> bf16", run(bwd=True))
    rep("input image -> bf16", run(inp=True))
    rep("input image x (1 + 1e-6 N(0,1))", run(noise=1e-6))
    rep("input image x (1 + 1e-4 N(0,1))", run(noise=1e-4))
    rep("conv weights -> bf16 (the shadow)", run(wts=True))
    rep("stored activations -> bf16, backbone stages 1-2 only", run(fwd=True, scope=("backbone.features.stage1", "backbone.features.stage2")))
    rep("stored activations -> bf16, whole backbone", run(fwd=True, scope=("backbone",)))
    rep("stored activations -> bf16, FPN + head only", run(fwd=True, scope=("head", "fpn")))
    rep("all of the above", run(fwd=True, bwd=True, wts=True, inp=True))
    print("# bf16 sensitivity of the student's gradients (oracle only, CPU)\n")
    print("`python tests/bf16_sensitivity.py`: BASELINE config 2 (B = 16, 256x256, darknet53 -> darknet_tiny_h, seeded "
          "weights), fp32 oracle with ONE class of values rounded to bf16; deviations from the unperturbed fp32 run.  "
          "Columns: element-wise relative error |g - g0| / |g0| of nine tensors (%s); worst per-tensor NORM deviation "
          "(the test's `worst_norm`) and its tensor; worst element-wise error; global gradient-norm deviation; 1 - cosine "
          "of the whole gradient.\n" % ", ".join("`%s`" % w.replace("backbone.features.", "") for w in WATCH))
    print("| what is rounded | element-wise error of the watched tensors | worst norm dev. | worst elem. | global norm | 1 - cos |")
    print("|---|---|---|---|---|---|")
    print("\n".join(rows))
    print("\nReading: rounding the GRADIENT tensors of the reverse sweep to bf16 -- the only thing a kernel-side change such "
          "as fp32 incoming gradients for the first BatchNorm layers could remove -- costs 1-3 % on the worst tensor.  Rounding "
          "what the FORWARD pass stores or reads (the input image, the weights, the activations) moves the small BatchNorm "
          "tensors of the backbone by tens of percent: max-pool / LeakyReLU decisions flip under a 2^-9 perturbation and the "
          "gradients of these layers are cancelling sums at this initialisation.  The whole-gradient direction (1 - cos) and the "
          "global norm stay at the 1e-3 level.  A bf16 step is therefore compared with the oracle's bf16-storage emulation "
          "(`oracle.kd_step_ref.bf16_storage`), which applies the same roundings in fp32 arithmetic.")


if __name__ == "__main__":
    main()
