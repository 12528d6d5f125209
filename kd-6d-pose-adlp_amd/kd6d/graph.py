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

    @property
    def pending_steps(self):
        """Batches the pipeline holds that have not had their student step yet."""
        return int(bool(self.pipeline and self.pending and self.primed))

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
            snet.store.resolve_grads(self._split, snet.store.n_train)      # accumulated FPN + head gradients -> fp32
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
            snet.resolve_hi = self._split
        try:
            losses = self.student.step_losses(self.images, self.tgt, pred_t, self._w)
        finally:
            snet.nhwc_in = None
            snet.grad_hook = None
            snet.resolve_hi = None
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


class _GroupSide:
    """One block of GroupedTeacherKDStep: `group` batches side by side (packed NHWC images, targets, teacher cells),
    everything a view of ONE byte buffer."""
    __slots__ = ("block", "nhwc", "tgts", "wf", "wi", "tk")


class GroupedTeacherKDStep(GraphedKDStep):
    """The pipelined step with the frozen teacher run over the batches of `group` consecutive steps AT ONCE, its pass
    cut into `group` segments of equal device time, one beside every student step.

    The teacher does not depend on the student's weights, so nothing forces it to see one batch per launch: at B = 16
    crops most of its layers are a single partial round of tiles on 256 CUs (342 tiles of 128 x 128 on 512 slots in
    the head towers, 8-32 tiles in stages 4-5), and the same forward over 32 / 48 images costs 83 / 72 us per image
    instead of 104 (tools/teacher_batch_probe.py).

    Three blocks of `group` batches each: LOAD (being filled, one batch per call), PASS (the teacher's input and, once
    its last segment has run, its cells) and CURRENT (what the student trains on).  Call k = m * group + s
        converts batch k into slot s of LOAD,
        replays teacher segment s (its own hipGraph, on the teacher's stream) over PASS = the batches of period m - 1,
        replays student step s (its own hipGraph [+ the optimiser]) on slot s of CURRENT = batch k - 2 * group,
    and the last call of a period ends with CURRENT <- PASS <- LOAD (two device copies, both streams joined).  Every
    batch still gets exactly one teacher forward and one student step.  The losses a call returns belong to batch
    k - 2 * group (None for the first 2 * group calls); `flush()` trains one pending batch per call without consuming a
    new one.  The teacher's segments are cut where a timed trial replay (device timestamps at every layer-group
    boundary, PoseNet.cut_hook) says the pass has spent s / group of its time; the cut graphs are captured in ONE walk
    through the forward, ending one capture and beginning the next at those boundaries.

    (Measured on the way, 60-step runs of bench.py: the whole group pass inside the last student step's graph of a
    period, 5110-5230 images/s at group 2-4 against 5400 for one teacher forward per step -- the steps without a teacher
    beside them leave the device half empty and the one with it is no shorter than teacher + student back to back;
    segment s forked and JOINED inside the graph of student step s, 5540-5660 -- every step then ends with whichever
    branch is longer running alone; segments as graphs of their own on the teacher's stream, joined with the student's
    stream only at the period's end, as built here: 5780-5810.)

    What it needs from the caller is look-ahead only: 2 * group batches in flight instead of one."""

    def __init__(self, teacher, student, optimizer, loss_weights=(0.1, 1.0, 5.0), cfg_kd=None, warmup=3, group=2,
                 exchange=None):
        if int(group) < 2:
            raise ValueError("GroupedTeacherKDStep: group >= 2 (GraphedKDStep(pipeline=True) is the group of one)")
        super().__init__(teacher, student, optimizer, loss_weights, cfg_kd=cfg_kd, warmup=warmup, concurrent=True,
                         pipeline=True, exchange=exchange)
        self.group = int(group)
        # the student's graphs have no teacher branch of their own and the teacher's segments run beside ALL of a step,
        # not its first 60 %: the weight-gradient settings of the strictly sequential step win here too (100-step runs,
        # interleaved, group 2: 5950-5980 images/s with the pipelined settings, 6200-6220 with two workgroups per CU for
        # the grouped launch and its flush behind the FPN sweep; CUs / 4 per side-stream launch: +-0)
        snet = student.net
        snet.wgrad_cu_budget = int(ops.device_cu_count() / self.WGRAD_BUDGET_DIV[False])
        snet.wgrad_group_wgs = 2 * ops.device_cu_count()
        snet.wgrad_group_flush = "fpn_end"
        self.g_student = self.g_teacher = None
        self._debug_skip_teacher = 0      # timing experiments only (bench.py --debug-skip-teacher): 1 = no teacher segment
        # is replayed, 2 = every second one
        self.sides = None                 # [current, pass, load]
        self.n_loaded = 0                 # batches in the load block
        self.p_valid = 0                  # batches of the pass block (the teacher's input)
        self.t_pos = 0                    # next teacher segment of the pass block
        self.c_valid = 0                  # batches of the current block (they carry teacher cells)
        self.c_pos = 0                    # next slot of the current block to train on
        self.draining = False
        self._img_big = self._tgt_big = None
        self._geom = None
        self.teacher_passes = 0           # completed group passes (bench.py reports them)
        self.segment_ms = None            # the trial replay's time per segment

    # `pending` of the base class, as a count
    @property
    def pending_steps(self):
        return (self.c_valid - self.c_pos) + self.p_valid + self.n_loaded

    @property
    def pending(self):
        return self.pending_steps > 0

    @pending.setter
    def pending(self, value):
        pass

    # ---- blocks ---------------------------------------------------------------------------------
    def _build_sides(self, x, tgt):
        from .kd_losses import CAP, TeacherKnowledge
        T = self.group
        B, _, H, W = x.shape
        if T * B * H * W >= 1 << 24:
            # the convolution entry points take < 2^24 rows per launch (kd6d_conv2d_fwd: "grid too large"); the group
            # pass's first layer has group * B * H * W of them
            raise ValueError("GroupedTeacherKDStep: group %d x %d images of %dx%d = %d input pixels per teacher pass, the "
                             "convolution launches take < 2^24 rows: use group <= %d"
                             % (T, B, H, W, T * B * H * W, ((1 << 24) - 1) // (B * H * W)))
        dev = x.device
        snet, tnet = self.student.net, self.teacher.net
        assert snet.dtype == tnet.dtype, "teacher and student share the converted input: same precision"
        esz = torch.empty((), dtype=snet.dtype).element_size()
        img_b = B * H * W * 8 * esz
        tb = tgt.block_bytes()
        n, nt = B * CAP, T * B * CAP
        wf_b, wi_n = nt * 48 * 4, nt + (T * B + 3) // 4 * 4

        def al(v):
            return (v + 255) // 256 * 256

        o_tgt = al(T * img_b)
        o_wf = o_tgt + T * al(tb)
        o_wi = o_wf + al(wf_b)
        total = o_wi + al(wi_n * 4)
        sides = []
        for _ in range(3):
            sd = _GroupSide()
            blk = sd.block = torch.zeros(total, dtype=torch.uint8, device=dev)
            sd.nhwc = blk[0:T * img_b].view(snet.dtype).view(T * B * H * W, 8)
            sd.tgts = []
            for s in range(T):
                t = tgt.clone_static()
                t.rebind_block(blk[o_tgt + s * al(tb):o_tgt + s * al(tb) + tb])
                sd.tgts.append(t)
            wf = sd.wf = blk[o_wf:o_wf + wf_b].view(torch.float32)
            wi = sd.wi = blk[o_wi:o_wi + wi_n * 4].view(torch.int32)
            kp, kpn = wf[0:nt * 16].view(nt, 8, 2), wf[nt * 16:nt * 32].view(nt, 8, 2)
            sc, beta = wf[nt * 32:nt * 40].view(nt, 8), wf[nt * 40:nt * 48].view(nt, 8)
            # the cells of slot s: a contiguous range of every slot array (they are image-major), no copies
            sd.tk = [TeacherKnowledge(wi[nt + s * B:nt + (s + 1) * B], kp[s * n:(s + 1) * n], sc[s * n:(s + 1) * n],
                                      wi[s * n:(s + 1) * n], kpn[s * n:(s + 1) * n], beta[s * n:(s + 1) * n], CAP, B)
                     for s in range(T)]
            sides.append(sd)
        self.sides = sides
        self._geom = (B, H, W)
        self.images = ImageList(torch.empty(1, dtype=x.dtype, device=dev).expand(B, 3, H, W),
                                getattr(self, "_sizes", None))                    # shape donor: the step reads `nhwc`
        self._img_big = ImageList(torch.empty(1, dtype=x.dtype, device=dev).expand(T * B, 3, H, W), None)
        big = object.__new__(PackedTargets)                                       # what the teacher reads of the targets
        big.bbox_trans = torch.zeros((T * B,) + tuple(tgt.bbox_trans.shape[1:]), dtype=torch.float32, device=dev)
        big.frame_wh = tgt.frame_wh
        self._tgt_big = big

    def _slot_rows(self, s):
        B, H, W = self._geom
        return slice(s * B * H * W, (s + 1) * B * H * W)

    def _load_group(self, images, tgt):
        assert self.n_loaded < self.group
        self._load_device(images, tgt, self.n_loaded)
        self.n_loaded += 1

    def _load_device(self, images, tgt, s):
        x = images.tensors if hasattr(images, "tensors") else images
        B, H, W = self._geom
        assert tuple(x.shape) == (B, 3, H, W), "the captured step has a static batch shape"
        L = self.sides[2]
        assert (tgt.mask_h, tgt.mask_w) == (L.tgts[s].mask_h, L.tgts[s].mask_w)
        # (on the student's stream, between two replays.  On a stream of its own -- the conversion is 15 us of a chain
        #  that bounds the step -- the step got 10-25 % SLOWER: 4640-5280 against 5780-5810 images/s at group 2-6)
        ops.image_to_nhwc(x.contiguous(), self.student.net.dtype, 8, out=L.nhwc[self._slot_rows(s)])
        L.tgts[s].copy_from(tgt)

    def _fill_unloaded(self):
        """A partial load block (a drain before `group` batches arrived): the free slots repeat slot 0, so that the
        group pass and a capture's warm-up steps see valid inputs everywhere."""
        L = self.sides[2]
        for s in range(self.n_loaded, self.group):
            L.nhwc[self._slot_rows(s)].copy_(L.nhwc[self._slot_rows(0)], non_blocking=True)
            L.tgts[s].copy_from(L.tgts[0])

    def _rotate(self):
        """Period end: CURRENT <- PASS <- LOAD.  The caller has issued every teacher segment of the pass block."""
        self._rotate_device(bool(self.p_valid), self.n_loaded)
        self.c_valid, self.p_valid, self.n_loaded = self.p_valid, self.n_loaded, 0
        self.c_pos = self.t_pos = 0

    def _rotate_device(self, pass_valid, n_loaded):
        main, ts = torch.cuda.current_stream(), self.teacher_stream
        C, P, L = self.sides
        if 0 < n_loaded < self.group:
            self._fill_unloaded()
        main.wait_stream(ts)
        if pass_valid:
            C.block.copy_(P.block, non_blocking=True)
        if n_loaded:
            P.block.copy_(L.block, non_blocking=True)
        ts.wait_stream(main)

    # ---- the teacher's pass over the pass block ----------------------------------------------------------
    def _teacher_forward(self):
        """Python walk through the group pass (current stream = the teacher's); its cells land in the block's flats."""
        T = self.group
        B = self._geom[0]
        P, tnet = self.sides[1], self.teacher.net
        ops.mark("teacher.start")
        for s in range(T):
            self._tgt_big.bbox_trans[s * B:(s + 1) * B].copy_(P.tgts[s].bbox_trans, non_blocking=True)
        tnet.nhwc_in = P.nhwc
        keep = getattr(self.teacher, "_teacher_flats", None)
        self.teacher._teacher_flats = (P.wf, P.wi)
        try:
            self.teacher(self._img_big, targets=self._tgt_big, is_teacher=True, cfg_kd=self.cfg_kd)
        finally:
            tnet.nhwc_in = None
            self.teacher._teacher_flats = keep
        ops.mark("teacher.end")

    def _time_cuts(self):
        """Where to cut the pass: a trial graph of the whole pass with a device timestamp at every layer-group boundary
        (PoseNet.cut_hook), replayed twice; cut k goes to the boundary closest to k / group of the total time."""
        import ctypes
        from ._lib import check, lib
        T, ts, tnet = self.group, self.teacher_stream, self.teacher.net
        tbuf = torch.zeros(1024, dtype=torch.int64, device=self.sides[1].block.device)
        count = [0]

        def stamp():
            assert count[0] < tbuf.numel()
            check(lib.kd6d_mark(ctypes.c_void_p(tbuf.data_ptr() + 8 * count[0]), ops._stream()), "kd6d_mark")
            count[0] += 1

        with torch.no_grad(), torch.cuda.stream(ts):
            self._teacher_forward()                       # eager: allocates every static buffer of this batch size
            ts.synchronize()
            trial = torch.cuda.CUDAGraph()
            with torch.cuda.graph(trial, stream=ts, capture_error_mode="thread_local"):
                stamp()
                tnet.cut_hook = stamp
                try:
                    self._teacher_forward()
                finally:
                    tnet.cut_hook = None
                stamp()
            trial.replay(); trial.replay()
            ts.synchronize()
        t = tbuf[:count[0]].cpu().tolist()
        rel = [(v - t[0]) * 1e-5 for v in t[1:]]          # ms after the start; rel[i] = end of layer group i, rel[-1] = end
        total, nhook = rel[-1], len(rel) - 1
        assert nhook >= T - 1, "the teacher's forward passes fewer cut points than segments"
        cuts, last = [], -1
        for k in range(1, T):
            cand = range(last + 1, nhook - (T - 1 - k))
            i = min(cand, key=lambda j: abs(rel[j] - k * total / T))
            cuts.append(i); last = i
        edges = [0.0] + [rel[i] for i in cuts] + [total]
        self.segment_ms = [edges[i + 1] - edges[i] for i in range(T)]
        del trial
        return cuts

    def _capture_teacher(self):
        """The pass over the pass block as `group` graphs, captured in ONE walk through the forward: the hook ends the
        running capture at a cut and begins the next graph's."""
        T, ts, tnet = self.group, self.teacher_stream, self.teacher.net
        torch.cuda.synchronize()
        cuts = self._time_cuts()
        graphs = [torch.cuda.CUDAGraph() for _ in range(T)]
        pool = torch.cuda.graph_pool_handle()
        state = {"hook": 0, "g": 0}

        def switch():
            g = state["g"]
            if g < T - 1 and state["hook"] == cuts[g]:
                graphs[g].capture_end()
                state["g"] = g + 1
                graphs[g + 1].capture_begin(pool=pool, capture_error_mode="thread_local")
            state["hook"] += 1

        torch.cuda.synchronize()
        with torch.no_grad(), torch.cuda.stream(ts):
            graphs[0].capture_begin(pool=pool, capture_error_mode="thread_local")
            tnet.cut_hook = switch
            try:
                self._teacher_forward()
            finally:
                tnet.cut_hook = None
                graphs[state["g"]].capture_end()
        assert state["g"] == T - 1, "the teacher's forward passed fewer cut points than segments"
        torch.cuda.synchronize()
        self.g_teacher = graphs

    def _teacher_segments(self, upto):
        """Issue the pass block's segments t_pos .. upto - 1 on the teacher's stream."""
        while self.p_valid and self.t_pos < upto:
            self._replay_teacher_segment(self.t_pos)
            self.t_pos += 1
            if self.t_pos == self.group:
                self.teacher_passes += 1

    def _replay_teacher_segment(self, i):
        if self.g_teacher is None:
            self._capture_teacher()
        if self._debug_skip_teacher == 1 or (self._debug_skip_teacher == 2 and i % 2 == 1):
            return
        with torch.cuda.stream(self.teacher_stream):
            self.g_teacher[i].replay()

    # ---- the student's step on one slot of the current block ----------------------------------------------
    def _student_body(self, s):
        C = self.sides[0]
        ops.mark("step.start")
        self.tgt, self.t_cur = C.tgts[s], C.tk[s]
        self._nhwc = (C.nhwc[self._slot_rows(s)], None)
        losses = self._student_step(self.t_cur)
        ops.mark("step.end")
        return losses

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
            st.refresh_shadow()
        opt.steps = snap["steps"]
        if snap["sc"] is not None:
            opt._step_count = snap["sc"]

    def _capture(self):
        """The `group` student graphs (one per slot of the current block; the blocks are only read)."""
        T = self.group
        self.student._defer_allreduce = True
        torch.cuda.synchronize()
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                        # eager warm-up: allocates every static buffer
            for _ in range(self.warmup):
                for s in range(T):
                    self._student_body(s)
                    # (no gradient exchange here: these steps are thrown away with the snapshot, and no RCCL communicator
                    #  may come to life before the graphs exist -- see prepare())
                    self.opt.advance()
                    self.opt.launch(device_schedule=True)
                    self._count_opt_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graphs_per_step = 2 if (D.exchange_active() and not self._overlap()) else 1
        self.g_student, pool = [], None
        for s in range(T):
            g = torch.cuda.CUDAGraph()
            kw = {} if pool is None else {"pool": pool}
            with torch.cuda.graph(g, capture_error_mode="thread_local", **kw):
                self.losses = self._student_body(s)
                if self.graphs_per_step == 1:
                    self.opt.launch(device_schedule=True)
            pool = g.pool()
            self.g_student.append(g)
        self.g_step = self.g_student[0]
        if self.graphs_per_step == 2:
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=pool, capture_error_mode="thread_local"):
                self.opt.launch(device_schedule=True)
        self._restore(snap)

    def _ensure_captured(self):
        if self.g_student is None:
            self._capture()

    def _replay_student(self, s):
        if self.graphs_per_step == 1:
            self.opt.advance()
            self.g_student[s].replay()
        else:
            self.g_student[s].replay()
            self._exchange()
            self.opt.advance()
            self.g_opt.replay()
        self._count_opt_step()
        for _, bn in self.student.net.bns:
            bn.fold = None
        return self.losses

    # ---- one tick of the pipeline -------------------------------------------------------------------------
    def _tick(self, drain):
        T = self.group
        if drain:
            while self.c_pos >= self.c_valid:      # nothing trainable: finish the teacher on the pass block, rotate
                self._teacher_segments(T)
                self._rotate()
        out = None
        if self.c_pos < self.c_valid:
            s = self.c_pos
            self._ensure_captured()                # (before this step's teacher segment goes out)
            self._teacher_segments(s + 1)          # segment s beside student step s
            out = self._replay_student(s)
            self.c_pos += 1
        if not drain and self.n_loaded == T:       # period end
            self._teacher_segments(T)
            self._rotate()
        return out

    # ---- public -------------------------------------------------------------------------------------
    def __call__(self, images, tgt):
        """One call = one batch in.  Returns the (device, static) loss scalars of batch k - 2 * group, None while the
        pipeline fills (the first 2 * group calls)."""
        tgt = self._prepare(images, tgt)
        if self.draining:
            if self.pending_steps > 0:
                raise RuntimeError("GroupedTeacherKDStep: flush() the %d pending batches before feeding new ones"
                                   % self.pending_steps)
            self.draining = False
        self._load_group(images, tgt)
        return self._tick(False)

    def _prepare(self, images, tgt):
        if not isinstance(tgt, PackedTargets):
            tgt = PackedTargets(tgt, self.student.net.device)
        if self.sides is None:
            self._sizes = getattr(images, "sizes", None)
            self._build_sides(images.tensors if hasattr(images, "tensors") else images, tgt)
        return tgt

    def prepare(self, images, tgt):
        """Capture every graph NOW from a sample batch, leaving the pipeline empty and the training state untouched.

        For data-parallel runs: call it BEFORE the first RCCL communicator of the process is created (before
        kd6d.libs.distributed.init_exchange() and before any torch.distributed collective on the device).  Measured in the
        one-rank rehearsal: graphs instantiated after ncclCommInitRank has run replay 18 % slower in this launch mode
        (5115-5180 against 6216-6290 images/s; destroying the communicator again does not bring it back, creating it after
        the capture costs nothing; which stream the teacher replays on, the number of weight-gradient streams or the
        communicator's channel count make no difference).  Without prepare() the graphs are captured by the first calls
        that need them, as before."""
        tgt = self._prepare(images, tgt)
        if self.g_student is not None and self.g_teacher is not None:
            return
        if self.pending_steps:
            raise RuntimeError("GroupedTeacherKDStep.prepare(): call it before the first batch is fed")
        for _ in range(self.group):
            self._load_group(images, tgt)
        self._rotate()                          # PASS <- LOAD
        self._teacher_segments(self.group)      # records the teacher's graphs (and runs them once: cells of the sample)
        self._rotate()                          # CURRENT <- PASS
        self._ensure_captured()                 # the student's graphs; snapshot / restore around their warm-up steps
        torch.cuda.synchronize()
        self.n_loaded = self.p_valid = self.t_pos = self.c_valid = self.c_pos = 0
        self.teacher_passes = 0

    def flush(self):
        """Train on ONE batch that is still waiting for its student step, without consuming a new one; None when
        nothing is pending.  (Call until it returns None to drain the pipeline; a later __call__ starts a new one.)"""
        if self.pending_steps == 0:
            return None
        self.draining = True
        return self._tick(True)
