"""Fused clip + AdamW over the flat parameter buffer (train_kd.py:138-139, train_libs.py:119).

Subclasses torch.optim.Optimizer only for bookkeeping (param_groups / lr schedulers such as the
reference's OneCycleLR work unchanged); step() is two kd6d launches and never touches torch math.
"""
import torch

from . import ops
from .ops import check, lib


class FusedClipAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, max_norm=1.0):
        self.model = model
        self.net = model.net
        super().__init__([p for p in model.parameters()], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_norm = max_norm
        st = self.net.store
        st.ensure_grads()
        dev = st.params.device
        self.exp_avg = torch.zeros(st.n_train, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(st.n_train, dtype=torch.float32, device=dev)
        # lr, 1-b1^t, sqrt(1-b2^t) in graph-replay form and the squared gradient norm (written by kd6d_clip_adamw from
        # kd6d_sumsq's partial sums, added in a fixed order: reproducible); kd6d_set_hyper (advance(), before every launch)
        # writes the three and clears the fourth in one launch
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self.gnorm_sq = self.hyper[3:4]
        self.gnorm_parts = torch.zeros(128, dtype=torch.float32, device=dev)        # KD6D_SUMSQ_PARTS
        self.steps = 0

    @torch.no_grad()
    def step(self, closure=None):
        self.advance()
        self.launch(device_schedule=False)

    def advance(self):
        """Host half of a step: count it, publish (lr, bias corrections) to the device and clear the gradient-norm
        accumulator.  In graph mode this runs eagerly before every replay of the captured
        `launch(device_schedule=True)`; launch() without a preceding advance() would add to the old norm."""
        g = self.param_groups[0]
        self.steps += 1
        check(lib.kd6d_set_hyper(ops._ptr(self.hyper), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                 self.steps, ops._stream()), "kd6d_set_hyper")

    @torch.no_grad()
    def launch(self, device_schedule=True):
        """Device half: sum of squares + fused clip/AdamW (+ bf16 shadow refresh).  With
        device_schedule the launch carries no per-step host scalar and can be captured in a hipGraph."""
        st = self.net.store
        g = self.param_groups[0]
        n = st.n_train
        P = ops._ptr
        s = ops._stream()
        check(lib.kd6d_sumsq(P(st.grads), n, P(self.gnorm_parts), s), "kd6d_sumsq")
        shadow = st.ensure_shadow() if self.net.dtype == torch.bfloat16 else None
        check(lib.kd6d_clip_adamw(P(st.params), P(st.grads), P(self.exp_avg), P(self.exp_avg_sq), n, P(self.gnorm_parts),
                                  P(self.gnorm_sq), float(self.max_norm or 0.0), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                  float(g["eps"]), float(g["weight_decay"]), max(self.steps, 1),
                                  P(self.hyper) if device_schedule else None, P(shadow), s), "kd6d_clip_adamw")
        self.net._weights_dirty = shadow is None and self.net.dtype == torch.bfloat16
        for _, bn in self.net.bns:
            bn.fold = None

    def grad_norm(self):
        """Pre-clip global gradient norm of the last step (device scalar)."""
        return self.gnorm_sq.sqrt()

    def state_dict(self):
        return dict(steps=self.steps, exp_avg=self.exp_avg.cpu(), exp_avg_sq=self.exp_avg_sq.cpu(),
                    param_groups=[{k: v for k, v in g.items() if k != "params"} for g in self.param_groups])

    def load_state_dict(self, sd):
        if "steps" not in sd or "exp_avg" not in sd:
            # a torch.optim.AdamW state dict ('state' per parameter index + 'param_groups'), e.g. the `optim` entry of a
            # latest.pth the reference wrote: per-parameter moments in the reference's registration order do not map
            # onto the flat buffers without the reference's module tree
            raise ValueError("FusedClipAdamW.load_state_dict: this is not a kd6d optimiser state (keys %s); a "
                             "torch.optim.AdamW state cannot be resumed here -- load only the 'model' entry of the "
                             "checkpoint (Adam moments restart from zero)" % sorted(sd.keys()))
        self.steps = int(sd["steps"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
