"""PoseModuleKD -- drop-in for models/model_kd.py:14-95 of the reference, on the kd6d engine.

Same constructor (`PoseModuleKD(cfg, backbone)`), same forward signature and return values
(train: `(None, {'loss_cls','loss_reg','loss_kd'})`; eval + is_teacher: the `pred_t` dict), same
state_dict keys/shapes.  Internally nothing is a torch op: the forward runs
HIP kernels over packed NHWC buffers, the three returned loss scalars carry an autograd node
whose backward() launches the hand-written reverse sweep, and every nn.Parameter is a view into
one flat fp32 buffer (its .grad a view into one flat gradient buffer).
"""
import torch
from torch import nn

from .. import engine, kd_losses, ops
from ..kd_losses import KDLoss, PackedTargets, TeacherKnowledge


class _Shell(nn.Module):
    """Container used only to reproduce the reference's parameter hierarchy."""


def _get_shell(root, dotted):
    m = root
    for part in dotted:
        nxt = m._modules.get(part)
        if nxt is None:
            nxt = _Shell()
            m.add_module(part, nxt)
        m = nxt
    return m


class _StepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, losses, module):
        ctx.module = module
        return losses[0].clone(), losses[1].clone(), losses[2].clone()

    @staticmethod
    def backward(ctx, g_cls, g_reg, g_kd):
        m = ctx.module
        dev = m.net.device
        z = torch.zeros((), dtype=torch.float32, device=dev)
        w = torch.stack([g if g is not None else z for g in (g_cls, g_reg, g_kd)]).to(torch.float32).contiguous()
        m._run_backward(w)
        return torch.zeros_like(m._anchor), None, None


class PoseModuleKD(nn.Module):
    def __init__(self, cfg, backbone):
        super().__init__()
        arch = getattr(backbone, "arch", None) or cfg["MODEL"]["BACKBONE"]
        prec = cfg.get("RUNTIME", {}).get("PRECISION", "bf16")
        dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[prec]
        self.cfg = cfg
        self.net = engine.PoseNet(arch, dtype, n_class=cfg["DATASETS"]["N_CLASS"], n_conv=cfg["MODEL"]["N_CONV"],
                                  prior=cfg["MODEL"]["PRIOR"])
        feat, oc = engine.BACKBONE_CFG[arch]
        assert list(cfg["MODEL"]["FEAT_CHANNELS"]) == feat and cfg["MODEL"]["OUT_CHANNEL"] == oc, \
            "cfg FEAT_CHANNELS/OUT_CHANNEL do not match backbone %s" % arch
        self.inference_th = cfg["TEST"]["CONFIDENCE_TH"]
        from ..postprocess import PostProcessor
        self.post_processor = PostProcessor(cfg["TEST"]["CONFIDENCE_TH"], cfg["SOLVER"]["POSITIVE_NUM"],
                                            cfg["SOLVER"]["POSITIVE_LAMBDA"], cfg["DATASETS"].get("SYMMETRY_TYPES", {}))
        self.positive_num = cfg["SOLVER"]["POSITIVE_NUM"]
        self.positive_lambda = cfg["SOLVER"]["POSITIVE_LAMBDA"]
        if cfg["SOLVER"]["POSITIVE_TYPE"] != "SSC" or cfg["SOLVER"]["LOSS_REG_TYPE"] != "3D" or \
                cfg["SOLVER"]["REGRESSION_TYPE"] != "POINT":
            raise NotImplementedError("the HIP path implements POSITIVE_TYPE=SSC, LOSS_REG_TYPE=3D, REGRESSION_TYPE=POINT")
        kd = cfg.get("KD", {})
        if "LEVEL" in kd and kd["LEVEL"] != "pred":
            raise KeyError("Ooops, KD from %s is not defined." % kd["LEVEL"])
        # postprocess_kd.py:187-202: the reference runs RANSAC-EPnP on a teacher's selected cells and keeps the image's
        # cells only if the solver succeeds.  Off by default (it needs the cells on the host: one synchronisation per
        # step, which a replayed hipGraph cannot contain); cfg['RUNTIME']['TEACHER_PNP_GATE'] (--teacher_pnp_gate) turns
        # it on for eager launches, with kd6d/libs/pnp.py as the solver
        self.teacher_pnp_gate = bool(cfg.get("RUNTIME", {}).get("TEACHER_PNP_GATE", False))
        self.loss_evaluator = KDLoss(cfg["INPUT"]["INTERNAL_K"], cfg["DATASETS"]["MESH_DIAMETERS"],
                                     cfg["SOLVER"]["FOCAL_GAMMA"], cfg["SOLVER"]["FOCAL_ALPHA"], self.positive_num,
                                     self.positive_lambda, kd if "GTYPE" in kd else None)
        # ---- parameters / buffers under the reference names, as views of the flat store ----
        self._names = []
        for name, view, is_param in self.net.named_logical():
            parts = name.split(".")
            shell = _get_shell(self, parts[:-1])
            if is_param:
                shell.register_parameter(parts[-1], nn.Parameter(view, requires_grad=True))
            else:
                shell.register_buffer(parts[-1], view)
            self._names.append((name, is_param))
        self._nbt = torch.zeros(max(len(self.net.bns), 1), dtype=torch.long)
        for i, (bn_name, _) in enumerate(self.net.bns):
            _get_shell(self, bn_name.split(".")).register_buffer("num_batches_tracked", self._nbt[i])
        ag = _get_shell(self, ["anchor_generator", "cell_anchors"])
        for i, (s, st) in enumerate(zip(engine.ANCHOR_SIZES, engine.ANCHOR_STRIDES)):
            c = st / 2.0
            ag.register_buffer(str(i), torch.tensor([[c - 0.5 * (s - 1)] * 2 + [c + 0.5 * (s - 1)] * 2], dtype=torch.float32))
        self._anchor = torch.zeros(1, requires_grad=True)
        if getattr(backbone, "pretrained_file", None):
            sd = torch.load(backbone.pretrained_file, map_location="cpu")
            own = self.state_dict()
            self.load_state_dict({k: v for k, v in {("backbone." + k): v for k, v in sd.items()}.items()
                                  if k in own and own[k].shape == v.shape}, strict=False)

    # ---- nn.Module plumbing ---------------------------------------------------------------
    def _bind(self):
        st = self.net.store
        views = {n: v for n, v, _ in self.net.named_logical()}
        for name, is_param in self._names:
            parts = name.split(".")
            shell = _get_shell(self, parts[:-1])
            if is_param:
                shell._parameters[parts[-1]].data = views[name]
            else:
                shell._buffers[parts[-1]] = views[name]
        self._nbt = self._nbt.to(self.net.device)
        for i, (bn_name, _) in enumerate(self.net.bns):
            _get_shell(self, bn_name.split("."))._buffers["num_batches_tracked"] = self._nbt[i]
        ag = _get_shell(self, ["anchor_generator", "cell_anchors"])
        for k in list(ag._buffers.keys()):
            ag._buffers[k] = ag._buffers[k].to(self.net.device)
        self._anchor = torch.zeros(1, device=self.net.device, requires_grad=True)
        self._bind_grads()

    def _bind_grads(self):
        st = self.net.store
        if st.grads is None:
            return
        for e in st.order:
            if e.region != "train":
                continue
            if e.name == "head.scales":
                for l in range(self.net.n_levels):
                    p = _get_shell(self, ["head", "scales", str(l)])._parameters["scale"]
                    p.grad = st.storage(e, "grads")[l:l + 1]
                continue
            parts = e.name.split(".")
            p = _get_shell(self, parts[:-1])._parameters[parts[-1]]
            p.grad = st.logical_view(e, "grads")

    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, dtype=torch.float32, device=self.net.device))
        if probe.dtype != torch.float32:
            raise TypeError("PoseModuleKD keeps fp32 master parameters; choose bf16 compute with "
                            "cfg['RUNTIME']['PRECISION'], not module.half()/bfloat16()")
        if probe.device != self.net.device:
            self.net.to(probe.device)
            self._bind()
        return self

    def zero_grad(self, set_to_none=False):
        st = self.net.store
        st.ensure_grads()
        first = _get_shell(self, ["head", "cls_logits"])._parameters["weight"]
        if first.grad is None:
            self._bind_grads()
        st.grads.zero_()

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        return type(sd)((k, v.detach().clone().contiguous()) for k, v in sd.items())

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        self.net.invalidate()
        return out

    def train(self, mode=True):
        if bool(mode) != bool(self.net.training):
            # eval-mode BatchNorm scale/shift are cached per weight load; a replayed optimiser graph changes the
            # weights without passing through Python, so the cache is dropped at every mode switch
            for _, bn in self.net.bns:
                bn.fold = None
        super().train(mode)
        self.net.training = mode
        return self

    # ---- the hot path ---------------------------------------------------------------------
    def forward(self, images, targets, is_teacher=False, pred_t=None, cfg_kd=None):
        x = images.tensors if hasattr(images, "tensors") else images
        if x.device != self.net.device:
            raise RuntimeError("images are on %s but the model is on %s" % (x.device, self.net.device))
        B = x.shape[0]
        net = self.net
        if self.training:
            losses = self._forward_losses(x, targets, pred_t)
            l_cls, l_reg, l_kd = _StepFn.apply(self._anchor, losses, self)
            return None, {"loss_cls": l_cls, "loss_reg": l_reg, "loss_kd": l_kd}
        if is_teacher:
            # caller-owned output buffers (GraphedKDStep): cleared together with the statistics arena in one launch
            flats = getattr(self, "_teacher_flats", None)
            if flats is not None and flats[0].numel() != B * kd_losses.CAP * 48:
                flats = None                  # another batch size than the one the buffers were made for
            pre = flats is not None and net.scratch_region() is not None
            if pre:
                ops.zero_many([net.scratch_region(), flats[0], flats[1]])
            cls, reg = net.forward(x, scratch_zeroed=pre)
            tgt = targets if isinstance(targets, PackedTargets) else PackedTargets(targets, net.device)
            tk = kd_losses.teacher_select(cls, reg, net.levels, B, tgt.bbox_trans, self.inference_th,
                                          self.positive_num, self.positive_lambda, frame_wh=tgt.frame_wh,
                                          flats=flats, zeroed=pre)
            if self.teacher_pnp_gate:
                self._apply_pnp_gate(tk, cls, tgt)
            return tk
        # evaluation: candidate cells per ground-truth class on the GPU, PnP-RANSAC on the host (models/model_kd.py:94-95)
        cls, reg = net.forward(x)
        tgt = targets if isinstance(targets, PackedTargets) else PackedTargets(targets, net.device)
        return self.post_processor(cls, reg, net.levels, B, tgt), {}

    def _apply_pnp_gate(self, tk, cls, tgt):
        """Drop the teacher cells of every image whose pose cannot be solved from them (postprocess_kd.py:187-202).
        Host round trip: eager launches only."""
        import numpy as np
        from ..libs.pnp import solve_pnp_ransac
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the teacher PnP gate synchronises with the host: use --launch eager with it")
        cnt = tk.t_cnt.cpu().tolist()
        kp = tk.t_kp.cpu().numpy()
        rows = tk.t_row.cpu().numpy()
        K, kp3d = tgt.K.cpu().numpy(), tgt.kp3d.cpu().numpy()
        keep = []
        for b, n in enumerate(cnt):
            ok = False
            if n > 0:
                r0 = int(rows[b * tk.cap])
                prob = torch.sigmoid(cls[r0, :self.net.n_cls]).cpu().numpy()
                cand = np.nonzero(prob > self.inference_th)[0]
                c = int(cand[0]) if len(cand) else int(prob.argmax())   # labels are visited in ascending order, first result kept
                uv = kp[b * tk.cap:b * tk.cap + n].reshape(-1, 2)
                xyz = np.tile(kp3d[b, c], (n, 1))
                ok = solve_pnp_ransac(xyz, uv, K[b], reproj_err=5.0)[0]
            keep.append(1 if ok else 0)
        tk.t_cnt.mul_(torch.tensor(keep, dtype=tk.t_cnt.dtype, device=tk.t_cnt.device))

    def _begin_step(self, x):
        """Step prologue of the fused training step: ONE launch zeroes the gradient bucket (the reference's
        optimizer.zero_grad(), train_kd.py:104), the statistics arena, the dense head gradient and the loss-side slot
        arrays, and counts the step in num_batches_tracked.  Returns True when everything could be covered (the level
        grid of this input shape is known from an earlier forward); otherwise only the bucket and the counters are
        done here and the forward / loss code zero their own buffers as in the eager path."""
        net, st = self.net, self.net.store
        st.ensure_grads()
        first = _get_shell(self, ["head", "cls_logits"])._parameters["weight"]
        if first.grad is None:
            self._bind_grads()
        B = x.shape[0]
        known = (net.levels is not None and net.batch == B and net.in_hw == tuple(x.shape[-2:])
                 and net.scratch_region() is not None)
        regions = [st.grads]
        if known:
            wf, wi = self.loss_evaluator.workspaces(B, net.device)
            regions += [net.scratch_region(), net.buf("dreg", (net.rows, net.pose_pred.cout_p)), wf, wi]
        ops.zero_many(regions, counter=self._nbt if known else None)     # else: _forward_losses counts the step
        return known

    def _forward_losses(self, x, targets, pred_t, prezeroed=False):
        """Student forward + the three loss sums -> fp32[3] device tensor {cls, reg, kd} (unweighted).
        prezeroed: _begin_step() covered the scratch arena, the loss workspaces and the step counters."""
        net = self.net
        B = x.shape[0]
        st = net.store
        st.ensure_grads()
        ops.mark("student.fwd.start")
        tgt = targets if isinstance(targets, PackedTargets) else PackedTargets(targets, net.device)
        # the SSC assignment reads the targets only: fork it beside the forward when a side stream exists and the
        # shapes are those of the previous step (the level grid is known only after a first forward)
        pre, side = None, net.side_stream
        keys = getattr(self, "_debug_keys", None)
        if side is not None and net.levels is not None and net.batch == B and net.in_hw == tuple(x.shape[-2:]):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pre = self.loss_evaluator.assign(net.levels, B, tgt, keys, prezeroed=prezeroed,
                                                 step_counter=self._nbt if prezeroed else None)
        cls, reg = net.forward(x, scratch_zeroed=prezeroed)
        if pre is not None:
            torch.cuda.current_stream().wait_stream(side)
        ops.mark("student.fwd.end")
        if isinstance(pred_t, kd_losses.DeferredTeacher):      # teacher ran concurrently on another stream
            pred_t = pred_t.join()
        teacher = pred_t if isinstance(pred_t, TeacherKnowledge) else None
        if pred_t is not None and teacher is None:
            raise TypeError("pred_t must come from a kd6d teacher forward (TeacherKnowledge)")
        losses = self.loss_evaluator.forward(cls, reg, net.levels, B, tgt, teacher, keys=keys,
                                             seg_scale=st.storage(net.scales), pre=pre)
        ops.mark("student.loss.end")
        if not prezeroed:
            self._nbt += 1
        return losses

    def step_losses(self, images, targets, pred_t, weights):
        """forward + backward of d(sum_i weights[i] * loss_i) without the autograd detour (the ~16 one-element
        torch kernels that `(l_cls * w + ...).backward()` puts between the loss and the reverse sweep).
        weights: fp32[3] device tensor.  Returns the fp32[3] loss tensor (static storage: the next call overwrites
        it).  The call opens with its own zero_grad() -- the gradients of THIS batch land in the flat bucket."""
        if not self.training:
            raise RuntimeError("step_losses() is the training step")
        x = images.tensors if hasattr(images, "tensors") else images
        if x.device != self.net.device:
            raise RuntimeError("images are on %s but the model is on %s" % (x.device, self.net.device))
        pre = self._begin_step(x)
        losses = self._forward_losses(x, targets, pred_t, prezeroed=pre)
        self._run_backward(weights, dreg_zeroed=pre)
        return losses

    def _run_backward(self, weights, dreg_zeroed=False):
        net, st = self.net, self.net.store
        first = _get_shell(self, ["head", "cls_logits"])._parameters["weight"]
        if first.grad is None:
            self._bind_grads()
        rows = net.rows
        dcls = net.buf("dcls", (rows, 16))
        dreg = net.buf("dreg", (rows, self.net.pose_pred.cout_p))
        if not dreg_zeroed:
            dreg.zero_()
        self.loss_evaluator.backward(weights, net.dtype, dcls, dreg, dseg_scale=st.acc(net.scales),
                                     acc_stride=st.acc_stride)
        ops.mark("student.bwd.start")
        from ..libs import distributed as D
        own = not getattr(self, "_defer_allreduce", False)      # GraphedKDStep schedules the exchange itself
        split = D.bucket_split(st) if (own and D.exchange_active() and D.EXCHANGE_MODE == "overlap") else None
        if split is not None:
            comm = self._comm_stream = getattr(self, "_comm_stream", None) or torch.cuda.Stream()

            def early():
                for side in [torch.cuda.current_stream()] + list(net.side_streams or ([net.side_stream] if net.side_stream else [])):
                    comm.wait_stream(side)
                with torch.cuda.stream(comm):
                    st.resolve_grads(split, st.n_train)      # the accumulated FPN + head gradients -> fp32 first
                    D.exchange_slice(st, split, st.n_train)
            net.grad_hook = early
            net.resolve_hi = split
        try:
            net.backward(dcls, dreg)
        finally:
            if split is not None:
                net.grad_hook = None
                net.resolve_hi = None
        ops.mark("student.bwd.end")
        if split is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
            D.exchange_slice(st, 0, split)
        elif own:
            D.exchange_gradients(st)
