"""hipGraph replay of the whole KD step (train_kd.py:104-140 of the reference).

One step is ~400 kernel launches with static shapes, static buffer addresses and no host
synchronisation, so launching it from Python costs more host time (~6 ms) than the kernels take.
`GraphedKDStep` captures it once into two hipGraphs and replays them:

    G1: zero grads -> teacher forward -> teacher cell selection -> student forward -> SSC
        assignment + focal / object-space / Sinkhorn-OT losses -> backward sweep
    (eager) RCCL mean all-reduce of the flat gradient bucket, world size > 1 only
    G2: sum of squares -> fused clip + AdamW (+ bf16 shadow refresh)

Everything that changes from step to step enters through device memory: the batch is copied into
static input buffers, the OneCycle learning rate and Adam bias corrections are written by the tiny
`kd6d_set_hyper` launch (`FusedClipAdamW.advance`), the random keys of the SSC positive sampling
come from torch's graph-safe Philox generator.
"""
import torch

from .kd_losses import DeferredTeacher, PackedTargets
from .libs import distributed as D
from .libs.poses import ImageList

class GraphedKDStep:
    def __init__(self, teacher, student, optimizer, loss_weights=(0.1, 1.0, 5.0), cfg_kd=None, warmup=3,
                 concurrent=True):
        self.teacher, self.student, self.opt = teacher, student, optimizer
        # fork/join inside the captured graph: the teacher's forward runs beside the student's, and the weight
        # gradients beside the dgrad / normalisation chain (many of these kernels fill < 256 CUs on their own)
        self.teacher_stream = torch.cuda.Stream() if concurrent else None
        student.net.side_stream = torch.cuda.Stream() if concurrent else None
        self.w_cls, self.w_reg, self.w_kd = (float(w) for w in loss_weights)
        self.cfg_kd = cfg_kd
        self.warmup = warmup
        self.g_step = self.g_opt = None
        self.images = self.tgt = self.losses = None

    # the body the reference's loop runs per iteration (train_kd.py:104-137)
    def _forward_backward(self):
        self.student.zero_grad()
        with torch.no_grad():
            if self.teacher_stream is None:
                pred_t = self.teacher(self.images, targets=self.tgt, is_teacher=True, cfg_kd=self.cfg_kd)
            else:
                self.teacher_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.teacher_stream):
                    pred_t = self.teacher(self.images, targets=self.tgt, is_teacher=True, cfg_kd=self.cfg_kd)
                pred_t = DeferredTeacher(pred_t, self.teacher_stream)
        _, ld = self.student(self.images, targets=self.tgt, pred_t=pred_t, cfg_kd=self.cfg_kd)
        loss = ld["loss_cls"] * self.w_cls + ld["loss_reg"] * self.w_reg
        if self.w_kd > 0.0:
            loss = loss + ld["loss_kd"] * self.w_kd
        loss.backward()
        return {k: v.detach() for k, v in ld.items()}

    def _load(self, images, tgt):
        x = images.tensors if hasattr(images, "tensors") else images
        if self.images is None:
            self.images = ImageList(torch.empty_like(x), getattr(images, "sizes", None))
            self.tgt = tgt.clone_static()
        assert x.shape == self.images.tensors.shape, "the captured step has a static batch shape"
        assert (tgt.mask_h, tgt.mask_w) == (self.tgt.mask_h, self.tgt.mask_w)
        self.images.tensors.copy_(x, non_blocking=True)
        self.tgt.copy_from(tgt)

    def _snapshot(self):
        st, opt = self.student.net.store, self.opt
        return dict(params=st.params.clone(), bufs=st.bufs.clone(), m=opt.exp_avg.clone(), v=opt.exp_avg_sq.clone(),
                    nbt=self.student._nbt.clone(), steps=opt.steps, sc=getattr(opt, "_step_count", None))

    def _restore(self, snap):
        st, opt = self.student.net.store, self.opt
        st.params.copy_(snap["params"]); st.bufs.copy_(snap["bufs"])
        opt.exp_avg.copy_(snap["m"]); opt.exp_avg_sq.copy_(snap["v"])
        self.student._nbt.copy_(snap["nbt"])
        if st.shadow is not None:
            st.refresh_shadow()                 # bf16 copy of the restored master weights
        opt.steps = snap["steps"]
        if snap["sc"] is not None:
            opt._step_count = snap["sc"]

    def _capture(self):
        self.student._defer_allreduce = True
        snap = self._snapshot()                              # the warm-up steps below must not train
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                        # eager warm-up: allocates every static buffer
            for _ in range(self.warmup):
                self._forward_backward()
                self._exchange()
                self.opt.advance()
                self.opt.launch(device_schedule=True)
                self._count_opt_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.g_step = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_step):
            self.losses = self._forward_backward()
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, pool=self.g_step.pool()):
            self.opt.launch(device_schedule=True)
        self._restore(snap)

    def _exchange(self):
        if D.get_world_size() > 1:
            st = self.student.net.store
            D.allreduce_mean_(st.grads[:st.n_train])

    def _count_opt_step(self):
        # torch's lr schedulers count optimizer.step() calls to warn about ordering
        if hasattr(self.opt, "_opt_called"):
            self.opt._opt_called = True
        if hasattr(self.opt, "_step_count"):
            self.opt._step_count += 1

    def __call__(self, images, tgt):
        """One KD step on (images, PackedTargets).  Returns the dict of (device, static) loss scalars."""
        if not isinstance(tgt, PackedTargets):
            tgt = PackedTargets(tgt, self.student.net.device)
        first = self.g_step is None
        self._load(images, tgt)
        if first:
            self._capture()
        self.g_step.replay()
        self._exchange()
        self.opt.advance()
        self.g_opt.replay()
        self._count_opt_step()
        return self.losses
