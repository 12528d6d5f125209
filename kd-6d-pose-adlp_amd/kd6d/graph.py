"""hipGraph replay of the whole KD step (train_kd.py:104-140 of the reference).

One step is ~350 kernel launches with static shapes, static buffer addresses and no host
synchronisation, so launching it from Python costs more host time (~6 ms) than the kernels take.
`GraphedKDStep` captures it once into two hipGraphs and replays them:

    G1: zero grads -> teacher forward + cell selection  ||  student forward -> SSC assignment +
        focal / object-space / Sinkhorn-OT losses -> backward sweep (weight gradients on a third stream)
    (eager) RCCL mean all-reduce of the flat gradient bucket, world size > 1 only
    G2: sum of squares -> fused clip + AdamW (+ bf16 shadow refresh)

Everything that changes from step to step enters through device memory: the batch is copied into
static input buffers, the OneCycle learning rate and Adam bias corrections are written by the tiny
`kd6d_set_hyper` launch (`FusedClipAdamW.advance`), the random keys of the SSC positive sampling
come from torch's graph-safe Philox generator.

`pipeline=True` additionally software-pipelines ACROSS steps: the frozen teacher does not depend on
the student's weights, so call k runs the teacher on batch k beside the student's forward/backward
on batch k-1 (whose teacher cells were produced by call k-1 and are double-buffered).  Every batch
still gets exactly one teacher forward and one student step; the losses returned by call k belong
to batch k-1 and `flush()` trains on the last pending batch.  The critical path of a call drops from
teacher + student to max(teacher, student), and the two halves fill each other's idle CUs.
"""

import torch

from . import ops
from .kd_losses import DeferredTeacher, PackedTargets
from .libs import distributed as D
from .libs.poses import ImageList


class GraphedKDStep:
    WGRAD_STREAMS = 4          # weight-gradient launches kept in flight beside the dgrad / normalisation chain
    # ... each sized (split-K count) for CUs / this many.  Pipelined steps (a teacher forward shares the device for
    # the first 60 % of the step): CUs / 2; strictly sequential steps: CUs / 4 (4242 against 4168 images/s).
    WGRAD_BUDGET_DIV = {True: 2.0, False: 4.0}

    def __init__(self, teacher, student, optimizer, loss_weights=(0.1, 1.0, 5.0), cfg_kd=None, warmup=3,
                 concurrent=True, pipeline=False, exchange=None):
        self.teacher, self.student, self.opt = teacher, student, optimizer
        # fork/join inside the captured graph: the teacher's forward runs beside the student's, and the weight
        # gradients beside the dgrad / normalisation chain (many of these kernels fill < 256 CUs on their own)
        self.teacher_stream = torch.cuda.Stream() if (concurrent or pipeline) else None
        # ... with several weight gradients in flight, each sized for a share of the CUs: the same k-loop work
        # with proportionally fewer atomic dW-tile flushes (measured: 1 -> 4 streams = +8 % on the step)
        # (WGRAD_STREAMS / WGRAD_BUDGET_DIV.  Swept on the closing state of round 1, medians of
        #  3 runs: 4 streams at CUs/4 4539 images/s, CUs/3 4566, CUs/2 4595-4640, CUs/1.5 4597, whole device 4529;
        #  3 streams 4330-4430, 5 streams 4140-4200, 6-8 streams 4340)
        nside = self.WGRAD_STREAMS
        budget_div = self.WGRAD_BUDGET_DIV[bool(pipeline)]
        snet = student.net
        snet.side_stream = torch.cuda.Stream() if concurrent else None
        snet.side_streams = ([snet.side_stream] + [torch.cuda.Stream() for _ in range(nside - 1)]
                             if concurrent and nside > 1 else None)
        snet.wgrad_cu_budget = int(ops.device_cu_count() / budget_div) if concurrent and nside > 1 else 0
        # grouped head weight gradients: one workgroup per two CUs beside a teacher forward (5284-5292 images/s against
        # 5215-5234 at one per CU and 5208-5214 at two), two per CU when the step is strictly sequential (4553 against 4467)
        snet.wgrad_group_wgs = ops.device_cu_count() // 2 if pipeline else 2 * ops.device_cu_count()
        snet.wgrad_group_flush = "head_end" if pipeline else "fpn_end"      # measured, see PoseNet.backward
        self.w_cls, self.w_reg, self.w_kd = (float(w) for w in loss_weights)
        self._w = None                                     # the same weights as a device tensor
        self.cfg_kd = cfg_kd
        self.warmup = warmup
        self.pipeline = pipeline
        self.g_step = self.g_opt = None
        self.graphs_per_step = 2
        self.images = self.tgt = self.losses = None        # the batch the student trains on
        self.images_nxt = self.tgt_nxt = None              # pipeline: the batch the teacher looks at
        self.t_cur = None                                  # pipeline: teacher cells of `images`
        self._blocks = None                                # pipeline: [current, next] hand-over blocks
        self._nhwc = None                                  # pipeline: (current, next) packed NHWC inputs inside them
        self.pending = False
        self.primed = False                                # pipeline: the teacher has seen a batch the student has not
        # data-parallel exchange schedule (kd6d/libs/distributed.py EXCHANGE_MODE): "between" the two graphs, or
        # "overlap" = two slices captured inside the single step graph, the big one beside the backbone sweep
        self.exchange = exchange or D.EXCHANGE_MODE
        assert self.exchange in ("between", "overlap")
        self.comm_stream = None
        self._split = None

    # ---- the body the reference's loop runs per iteration (train_kd.py:104-137) ----------
    def _overlap(self):
        return self.exchange == "overlap" and D.exchange_active()

    def _early_exchange(self):
        """PoseNet.backward calls this when the FPN + head gradients have been issued: their all-reduce goes out on the
        communication stream, behind everything the weight-gradient streams and the sweep's stream hold so far."""
        snet = self.student.net
        comm = self.comm_stream
        for side in [torch.cuda.current_stream()] + list(snet.side_streams or ([snet.side_stream] if snet.side_stream else [])):
            comm.wait_stream(side)
        with torch.cuda.stream(comm):
            D.exchange_slice(snet.store, self._split, snet.store.n_train)

    def _student_step(self, pred_t):
        if self._w is None:
            self._w = torch.tensor([self.w_cls, self.w_reg, self.w_kd], dtype=torch.float32,
                                   device=self.student.net.device)
        snet = self.student.net
        snet.nhwc_in = self._nhwc[0] if self._nhwc is not None else None
        overlap = self._overlap()
        if overlap:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream()
                self._split = D.bucket_split(snet.store)
            snet.grad_hook = self._early_exchange
        try:
            losses = self.student.step_losses(self.images, self.tgt, pred_t, self._w)
        finally:
            snet.nhwc_in = None
            snet.grad_hook = None
        if overlap:        # the sweep has joined its side streams: the backbone's slice, behind the big one
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            D.exchange_slice(snet.store, 0, self._split)
        return {"loss_cls": losses[0], "loss_reg": losses[1], "loss_kd": losses[2]}

    def _teacher(self, images, tgt):
        tnet = self.teacher.net
        tnet.nhwc_out = self._nhwc[1] if (self._nhwc is not None and images is self.images_nxt) else None
        try:
            return self._teacher_launch(images, tgt)
        finally:
            tnet.nhwc_out = None

    def _teacher_launch(self, images, tgt):
        with torch.no_grad():
            if self.teacher_stream is None:
                return self.teacher(images, targets=tgt, is_teacher=True, cfg_kd=self.cfg_kd)
            self.teacher_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.teacher_stream):
                ops.mark("teacher.start")
                pred = self.teacher(images, targets=tgt, is_teacher=True, cfg_kd=self.cfg_kd)
                ops.mark("teacher.end")
            return DeferredTeacher(pred, self.teacher_stream)

    def _forward_backward(self):
        ops.mark("step.start")              # (step_losses opens with the step's zero_grad)
        if not self.pipeline:
            return self._student_step(self._teacher(self.images, self.tgt))
        nxt = self._teacher(self.images_nxt, self.tgt_nxt)       # batch k, beside ...
        losses = self._student_step(self.t_cur)                  # ... the student step on batch k-1
        nxt = nxt.join() if isinstance(nxt, DeferredTeacher) else nxt
        # (running these hand-over copies on the teacher's stream behind the reverse sweep removes the 50 us they
        # trail the step by, but the extra mid-step dependency costs 120 us: measured, reverted)
        self._advance(nxt)
        ops.mark("step.end")
        return losses

    def _advance(self, pred_nxt):
        """k -> k+1: what the teacher just saw becomes the student's next batch."""
        if self._blocks is not None:         # everything handed over lives in one block per side: ONE copy
            self._blocks[0].copy_(self._blocks[1], non_blocking=True)
            for key in ("post_kp_2d", "post_kp_cls", "post_pos_per_img"):
                self.t_cur.pop(key, None)
            return
        self.t_cur.copy_from(pred_nxt)
        self.images.tensors.copy_(self.images_nxt.tensors, non_blocking=True)
        self.tgt.copy_from(self.tgt_nxt)

    def _make_blocks(self):
        """Re-home images, targets and teacher cells of the current / next batch in two byte blocks of identical
        layout, so that the hand-over at the end of a step is one device copy instead of six."""
        from .kd_losses import TeacherKnowledge
        dev = self.images.tensors.device
        # the image travels in the layout both networks consume (packed NHWC, 8 channels, the networks' dtype): the
        # teacher's conversion of batch k is what the student reads one replay later instead of converting it again
        snet, tnet = self.student.net, self.teacher.net
        assert snet.dtype == tnet.dtype, "teacher and student share the converted input: same precision"
        nhwc_cur = ops.image_to_nhwc(self.images.tensors, snet.dtype, 8)
        tgt_bytes = torch.empty(self.tgt.block_bytes(), dtype=torch.uint8, device=dev)      # shape donor only
        parts = [nhwc_cur, tgt_bytes, self.t_cur.flats[0], self.t_cur.flats[1]]
        offs, total = [], 0
        for t in parts:
            offs.append(total)
            total += (t.numel() * t.element_size() + 255) // 256 * 256
        blocks = [torch.zeros(total, dtype=torch.uint8, device=dev) for _ in range(2)]

        def views(block):
            return [block[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for o, t in zip(offs, parts)]

        cur, nxt = views(blocks[0]), views(blocks[1])
        cur[0].copy_(nhwc_cur)
        ops.image_to_nhwc(self.images_nxt.tensors, snet.dtype, 8, out=nxt[0])
        self._nhwc = (cur[0], nxt[0])
        for side, tgt in ((cur, self.tgt), (nxt, self.tgt_nxt)):
            tgt.rebind_block(side[1])
        cur[2].copy_(self.t_cur.flats[0]); cur[3].copy_(self.t_cur.flats[1])
        self.t_cur = TeacherKnowledge.from_flats(cur[2], cur[3], self.t_cur.batch, self.t_cur.cap)
        self.teacher._teacher_flats = (nxt[2], nxt[3])     # the teacher's selection writes straight into the next block
        self._blocks = blocks

    # ---- static inputs ------------------------------------------------------------------------
    def _load(self, images, tgt):
        x = images.tensors if hasattr(images, "tensors") else images
        if self.images is None:
            self.images = ImageList(torch.empty_like(x), getattr(images, "sizes", None))
            self.tgt = tgt.clone_static()
            if self.pipeline:
                self.images_nxt = ImageList(torch.empty_like(x), getattr(images, "sizes", None))
                self.tgt_nxt = tgt.clone_static()
        dst_i, dst_t = (self.images_nxt, self.tgt_nxt) if self.pipeline else (self.images, self.tgt)
        assert x.shape == dst_i.tensors.shape, "the captured step has a static batch shape"
        assert (tgt.mask_h, tgt.mask_w) == (dst_t.mask_h, dst_t.mask_w)
        dst_i.tensors.copy_(x, non_blocking=True)
        dst_t.copy_from(tgt)

    # ---- capture ------------------------------------------------------------------------------
    def _snapshot(self):
        st, opt = self.student.net.store, self.opt
        snap = dict(params=st.params.clone(), bufs=st.bufs.clone(), m=opt.exp_avg.clone(), v=opt.exp_avg_sq.clone(),
                    nbt=self.student._nbt.clone(), steps=opt.steps, sc=getattr(opt, "_step_count", None))
        if self.pipeline:
            snap.update(img=(self._nhwc[0] if self._nhwc is not None else self.images.tensors).clone(),
                        mask=self.tgt.mask.clone(), ff=self.tgt.flat_f.clone(),
                        fi=self.tgt.flat_i.clone(), tf=self.t_cur.flats[0].clone(), ti=self.t_cur.flats[1].clone())
        return snap

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
        if self.pipeline:                       # the warm-up iterations advanced the pipeline: rewind it
            (self._nhwc[0] if self._nhwc is not None else self.images.tensors).copy_(snap["img"])
            self.tgt.mask.copy_(snap["mask"])
            self.tgt.flat_f.copy_(snap["ff"]); self.tgt.flat_i.copy_(snap["fi"])
            self.t_cur.flats[0].copy_(snap["tf"]); self.t_cur.flats[1].copy_(snap["ti"])

    def _capture(self):
        self.student._defer_allreduce = True
        if self.pipeline and self._blocks is None:
            self._make_blocks()
        elif not self.pipeline and getattr(self.teacher, "_teacher_flats", None) is None:
            from .kd_losses import teacher_flats     # caller-owned teacher outputs: cleared by the teacher's prologue
            self.teacher._teacher_flats = teacher_flats(self.images.tensors.shape[0], self.images.tensors.device)
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
        # Without a gradient exchange (one rank) nothing has to happen between the reverse sweep and the optimiser:
        # ONE graph per step, and the eager kd6d_set_hyper launch moves in front of it.  With an exchange the RCCL
        # all-reduce sits between two graphs.
        self.graphs_per_step = 2 if (D.exchange_active() and not self._overlap()) else 1
        self.g_step = torch.cuda.CUDAGraph()
        # thread-local capture mode: with a process group alive, RCCL's watchdog thread polls events concurrently
        with torch.cuda.graph(self.g_step, capture_error_mode="thread_local"):
            self.losses = self._forward_backward()
            if self.graphs_per_step == 1:
                self.opt.launch(device_schedule=True)
        if self.graphs_per_step == 2:
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=self.g_step.pool(), capture_error_mode="thread_local"):
                self.opt.launch(device_schedule=True)
        self._restore(snap)

    def _exchange(self):
        if not self._overlap():            # (overlap: both slices were issued inside the step already)
            D.exchange_gradients(self.student.net.store)

    def _count_opt_step(self):
        # torch's lr schedulers count optimizer.step() calls to warn about ordering
        if hasattr(self.opt, "_opt_called"):
            self.opt._opt_called = True
        if hasattr(self.opt, "_step_count"):
            self.opt._step_count += 1

    def _replay(self):
        if self.graphs_per_step == 1:
            self.opt.advance()                 # lr / bias corrections of THIS step, consumed at the graph's end
            self.g_step.replay()
        else:
            self.g_step.replay()
            self._exchange()
            self.opt.advance()
            self.g_opt.replay()
        self._count_opt_step()
        for _, bn in self.student.net.bns:     # what FusedClipAdamW.launch() does when it runs from Python: the
            bn.fold = None                     # cached eval-mode BN scale/shift belong to the previous weights
        return self.losses

    # ---- public -------------------------------------------------------------------------------
    def __call__(self, images, tgt):
        """One KD step.  Returns the dict of (device, static) loss scalars -- of THIS batch, or with
        pipeline=True of the previous one (None on the priming call)."""
        if not isinstance(tgt, PackedTargets):
            tgt = PackedTargets(tgt, self.student.net.device)
        self._load(images, tgt)
        if not self.pipeline:
            if self.g_step is None:
                self._capture()
            return self._replay()
        if not self.primed:                     # priming call: teacher only, the student starts next call
            flats = getattr(self.teacher, "_teacher_flats", None)
            self.teacher._teacher_flats = None  # (eager call: fresh output buffers, not the captured step's)
            with torch.no_grad():
                pred = self.teacher(self.images_nxt, targets=self.tgt_nxt, is_teacher=True, cfg_kd=self.cfg_kd)
            self.teacher._teacher_flats = flats
            if self.t_cur is None:
                self.t_cur = pred.clone_static()
            else:                               # a new pipeline after flush(): the captured step's buffers stay
                self.t_cur.copy_from(pred)
            self.images.tensors.copy_(self.images_nxt.tensors)
            if self._nhwc is not None:
                ops.image_to_nhwc(self.images_nxt.tensors, self.student.net.dtype, 8, out=self._nhwc[0])
            self.tgt.copy_from(self.tgt_nxt)
            self.primed = self.pending = True
            return None
        if self.g_step is None:
            self._capture()
        self.pending = True
        return self._replay()

    def flush(self):
        """pipeline=True: train on the batch that is still waiting for its student step."""
        if not (self.pipeline and self.pending and self.primed):
            return None
        if self.g_step is None:
            self._capture()
        self.pending = False
        out = self._replay()       # the teacher re-reads the same (last) batch: harmless, results unused
        self.primed = False        # a later call starts a new pipeline (teacher only) instead of repeating this batch
        return out
