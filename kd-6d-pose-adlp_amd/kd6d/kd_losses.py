"""Host orchestration of the loss side of the KD step (no arithmetic: kernel launches only).

Replaces the Python of losses/kd_loss.py:111-160 (KDPoseLoss.__call__), :40-109
(KDObjectSpaceLoss), losses/loss_libs.py:1-51 (kd_loss_2d), losses/loss.py:164-268
(prepare_targets) and postprocess/postprocess_kd.py (teacher knowledge) with one launch each and
no device->host synchronisation anywhere in the step.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops
from ._lib import Levels, check, lib

ANCHOR_SIZES = [32, 64, 128, 256, 512]
ANCHOR_STRIDES = [8, 16, 32, 64, 128]
CAP = 32          # slots per image for positives / teacher cells (reference: ~10, POSITIVE_NUM)
MAX_GT = _lib.MAX_GT


def make_levels(batch, shapes, sizes=ANCHOR_SIZES, strides=ANCHOR_STRIDES):
    lv = Levels()
    lv.n, lv.batch = len(shapes), batch
    for i in range(_lib.MAX_SEG):
        lv.anchor_size[i] = float(sizes[i])
        lv.anchor_stride[i] = float(strides[i])
        if i < len(shapes):
            lv.h[i], lv.w[i] = shapes[i]
    return lv


class PackedTargets:
    """list[PoseAnnot] -> dense device tensors, padded to MAX_GT instances per image."""

    def __init__(self, targets, device):
        B = len(targets)
        self.batch = B
        f32 = dict(dtype=torch.float32, device=device)
        self.mask = torch.stack([t.mask.to(device=device, dtype=torch.float32) for t in targets]).contiguous()
        self.mask_h, self.mask_w = int(self.mask.shape[1]), int(self.mask.shape[2])
        self.kp3d = torch.stack([t.keypoints_3d.to(**f32) for t in targets]).contiguous()     # (B,15,8,3)
        self.K = torch.stack([t.K.to(**f32) for t in targets]).contiguous()
        self.bbox_trans = torch.stack([t.bbox_trans.to(**f32) for t in targets]).contiguous()
        cls = torch.full((B, MAX_GT), 0, dtype=torch.int32)
        ngt = torch.zeros(B, dtype=torch.int32)
        rot = torch.zeros(B, MAX_GT, 3, 3)
        tr = torch.zeros(B, MAX_GT, 3)
        same_dev = all(t.class_ids.device == torch.device(device) for t in targets) and torch.device(device).type != "cpu"
        if same_dev:
            # inputs already resident on the device: assemble there (no host round trip)
            cls, ngt, rot, tr = cls.to(device), ngt.to(device), rot.to(device), tr.to(device)
        for i, t in enumerate(targets):
            g = min(len(t.class_ids), MAX_GT)
            if len(t.class_ids) > MAX_GT:
                raise ValueError("kd6d_ssc_assign handles at most %d instances per image" % MAX_GT)
            cls[i, :g] = t.class_ids[:g].to(torch.int32)
            ngt[i] = g
            rot[i, :g] = t.rotations[:g].to(torch.float32)
            tr[i, :g] = t.translations[:g].reshape(g, 3).to(torch.float32)
        self.class_ids = cls.to(device).contiguous()
        self.n_gt = ngt.to(device).contiguous()
        self.rot = rot.to(device).contiguous()
        self.trans = tr.to(device).contiguous()
        self.frame_wh = (640.0, 480.0)        # kd_loss.py:116-117 ("not 256")
        self._pack_small()

    _SMALL_F = ("kp3d", "K", "bbox_trans", "rot", "trans")
    _SMALL_I = ("class_ids", "n_gt")

    def _pack_small(self):
        """Re-home the small per-image fields as views of one fp32 and one int32 buffer (8 tensors -> 2)."""
        for names, attr in ((self._SMALL_F, "flat_f"), (self._SMALL_I, "flat_i")):
            parts = [getattr(self, n) for n in names]
            sizes = [(p.numel() + 3) // 4 * 4 for p in parts]          # keep every view 16-byte aligned
            flat = torch.zeros(sum(sizes), dtype=parts[0].dtype, device=parts[0].device)
            off = 0
            for n, p_, sz in zip(names, parts, sizes):
                v = flat[off:off + p_.numel()].view(p_.shape)
                v.copy_(p_)
                setattr(self, n, v)
                off += sz
            setattr(self, attr, flat)
        # ... and mask, floats, ints in ONE block, so that a static copy of a batch is one device copy
        self.block = None
        self._into_block(torch.zeros(self.block_bytes(), dtype=torch.uint8, device=self.mask.device))

    def _block_layout(self):
        """Byte offsets of (mask, flat_f, flat_i) inside one block, each segment 256-byte aligned, and the total."""
        sizes = [t.numel() * t.element_size() for t in (self.mask, self.flat_f, self.flat_i)]
        offs, total = [], 0
        for sz in sizes:
            offs.append(total)
            total += (sz + 255) // 256 * 256
        return offs, sizes, total

    def _into_block(self, block):
        """Re-home mask / flat_f / flat_i as views of `block` (uint8, _block_layout()'s size); contents are copied."""
        offs, sizes, total = self._block_layout()
        assert block.dtype == torch.uint8 and block.numel() == total
        new = [block[o:o + sz].view(t.dtype).view(t.shape)
               for o, sz, t in zip(offs, sizes, (self.mask, self.flat_f, self.flat_i))]
        for n, t in zip(new, (self.mask, self.flat_f, self.flat_i)):
            n.copy_(t)
        self.mask, self.flat_f, self.flat_i = new
        self.block = block
        self._rebind()

    def block_bytes(self):
        return self._block_layout()[2]

    def rebind_block(self, block):
        """Move the batch into caller-owned storage (GraphedKDStep's hand-over blocks); contents are copied."""
        self._into_block(block)

    def clone_static(self):
        out = object.__new__(PackedTargets)
        out.__dict__.update(self.__dict__)
        out.mask = self.mask.clone()
        for names, attr in ((self._SMALL_F, "flat_f"), (self._SMALL_I, "flat_i")):
            setattr(out, attr, getattr(self, attr).clone())
        out._rebind()
        out.block = None
        out._into_block(torch.zeros(out.block_bytes(), dtype=torch.uint8, device=out.mask.device))
        return out

    def _rebind(self):
        for names, attr in ((self._SMALL_F, "flat_f"), (self._SMALL_I, "flat_i")):
            flat, off = getattr(self, attr), 0
            for n in names:
                old = getattr(self, n)
                setattr(self, n, flat[off:off + old.numel()].view(old.shape))
                off += (old.numel() + 3) // 4 * 4

    def copy_from(self, other):
        """One device copy when both sides keep their batch in one block of the same layout (every PackedTargets
        built by __init__ / clone_static does), three otherwise."""
        mine, theirs = getattr(self, "block", None), getattr(other, "block", None)
        if mine is not None and theirs is not None and mine.numel() == theirs.numel():
            mine.copy_(theirs, non_blocking=True)
            return
        self.mask.copy_(other.mask, non_blocking=True)
        self.flat_f.copy_(other.flat_f, non_blocking=True)
        self.flat_i.copy_(other.flat_i, non_blocking=True)

    def rebind_storage(self, mask, flat_f, flat_i):
        """Move the three device buffers into caller-owned storage of the same shapes (contents are copied)."""
        for new, old in ((mask, self.mask), (flat_f, self.flat_f), (flat_i, self.flat_i)):
            assert new.shape == old.shape and new.dtype == old.dtype
            new.copy_(old)
        self.mask, self.flat_f, self.flat_i = mask, flat_f, flat_i
        self.block = None
        self._rebind()


_SLOT_STARTS = {}


def _slot_starts(batch, cap, device):
    """int32 [0, cap, 2*cap, ...]: constant per (batch, cap, device), built once (it used to cost two torch launches
    inside every replayed step)."""
    key = (batch, cap, str(device))
    t = _SLOT_STARTS.get(key)
    if t is None:
        t = _SLOT_STARTS[key] = torch.arange(batch, dtype=torch.int32, device=device) * cap
    return t


def teacher_flats(batch, device, cap=None):
    """Caller-owned output buffers of teacher_select: fp32 (n*48) and int32 (n + batch rounded up to 4)."""
    n = batch * (cap or CAP)
    return (torch.zeros(n * 48, dtype=torch.float32, device=device),
            torch.zeros(n + (batch + 3) // 4 * 4, dtype=torch.int32, device=device))


class TeacherKnowledge(dict):
    """pred_t of the reference (models/model_kd.py:83-92).  The device-side slot arrays are what the
    student step consumes; the reference-named entries are materialised (with a sync) on demand."""

    def __init__(self, t_cnt, t_kp, t_score, t_row, t_kp_norm, t_beta, cap, batch, flats=None):
        super().__init__()
        self.t_cnt, self.t_kp, self.t_score, self.t_row = t_cnt, t_kp, t_score, t_row
        self.t_kp_norm, self.t_beta = t_kp_norm, t_beta
        self.cap, self.batch = cap, batch
        self.t_start = _slot_starts(batch, cap, t_cnt.device)
        self.flats = flats            # (fp32, int32) buffers all the slot arrays are views of

    @staticmethod
    def from_flats(wf, wi, batch, cap):
        """Slot arrays as views of one fp32 (n*48) and one int32 (n + batch rounded up to 4) buffer."""
        n, b = batch * cap, batch
        return TeacherKnowledge(wi[n:n + b], wf[0:n * 16].view(n, 8, 2), wf[n * 32:n * 40].view(n, 8), wi[0:n],
                                wf[n * 16:n * 32].view(n, 8, 2), wf[n * 40:n * 48].view(n, 8), cap, b, (wf, wi))

    def clone_static(self):
        """Persistent copy (own storage) that `copy_from` refreshes: the double buffer of the step pipeline."""
        wf, wi = (t.clone() for t in self.flats)
        return TeacherKnowledge.from_flats(wf, wi, self.batch, self.cap)

    def copy_from(self, other):
        for mine, theirs in zip(self.flats, other.flats):
            mine.copy_(theirs, non_blocking=True)
        for key in ("post_kp_2d", "post_kp_cls", "post_pos_per_img"):
            self.pop(key, None)

    def __missing__(self, key):
        if key not in ("post_kp_2d", "post_kp_cls", "post_pos_per_img"):
            raise KeyError(key)
        cnt = self.t_cnt.cpu().tolist()
        kp = torch.cat([self.t_kp[b * self.cap:b * self.cap + n] for b, n in enumerate(cnt)]) if self.batch else self.t_kp[:0]
        sc = torch.cat([self.t_score[b * self.cap:b * self.cap + n] for b, n in enumerate(cnt)]) if self.batch else self.t_score[:0]
        self["post_kp_2d"], self["post_kp_cls"], self["post_pos_per_img"] = kp, sc, cnt
        return dict.__getitem__(self, key)


class DeferredTeacher:
    """pred_t produced on another HIP stream (the frozen teacher's forward overlaps the student's).
    The student's loss joins that stream right before it first reads the teacher's cells."""

    def __init__(self, value, stream):
        self.value, self.stream = value, stream

    def join(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        return self.value


def teacher_select(cls_t, reg_t, levels, batch, bbox_trans, th=0.1, positive_num=10, positive_lambda=1.0, cap=CAP,
                   frame_wh=(640.0, 480.0), flats=None, zeroed=False):
    """flats: optional caller-owned (fp32 n*48, int32 n + batch rounded up to 4) output buffers (the step pipeline
    keeps them inside its hand-over block); zeroed here unless the caller already did (zeroed=True: the teacher's
    step prologue, ops.zero_many)."""
    dev = cls_t.device
    lv = make_levels(batch, levels)
    n = batch * cap
    if flats is None:
        wf = torch.zeros(n * 48, dtype=torch.float32, device=dev)        # one fill for all fp32 outputs
        wi = torch.zeros(n + (batch + 3) // 4 * 4, dtype=torch.int32, device=dev)
    else:
        wf, wi = flats
        assert wf.numel() == n * 48 and wi.numel() == n + (batch + 3) // 4 * 4
        if not zeroed:
            wf.zero_(); wi.zero_()
    t_kp, t_kp_n = wf[0:n * 16].view(n, 8, 2), wf[n * 16:n * 32].view(n, 8, 2)
    t_score, t_beta = wf[n * 32:n * 40].view(n, 8), wf[n * 40:n * 48].view(n, 8)
    t_row, t_cnt = wi[0:n], wi[n:n + batch]
    check(lib.kd6d_teacher_select(ctypes.byref(lv), ops._ptr(cls_t), ops._ptr(reg_t), ops._ptr(bbox_trans),
                                  th, float(positive_num), float(positive_lambda), cap, frame_wh[0], frame_wh[1],
                                  ops._ptr(t_cnt), ops._ptr(t_kp), ops._ptr(t_score), ops._ptr(t_row),
                                  ops._ptr(t_kp_n), ops._ptr(t_beta), ops._stream()),
          "kd6d_teacher_select")
    return TeacherKnowledge(t_cnt, t_kp, t_score, t_row, t_kp_n, t_beta, cap, batch, (wf, wi))


class KDLoss:
    """Student-side losses of one step.  forward() launches; backward(weights) fills dcls/dreg."""

    def __init__(self, internal_K, diameters, gamma=2.0, alpha=0.25, positive_num=10, positive_lambda=1.0,
                 kd_cfg=None, cap=CAP):
        K = np.asarray(internal_K, np.float64).reshape(3, 3)
        self.kinv = (ctypes.c_float * 9)(*np.linalg.inv(K).reshape(-1).astype(np.float32).tolist())
        self.diameters_host = [float(d) for d in diameters]
        self.diameters = None
        self.gamma, self.alpha = float(gamma), float(alpha)
        self.positive_num, self.positive_lambda = float(positive_num), float(positive_lambda)
        kd_cfg = kd_cfg or {}
        if kd_cfg.get("GTYPE", "sinkhorn") != "sinkhorn":
            raise NotImplementedError("only --gtype sinkhorn is implemented on the HIP path (got %r)" % kd_cfg.get("GTYPE"))
        if kd_cfg.get("GLEVEL", "point") != "point" or int(kd_cfg.get("GnD", 2)) != 2:
            raise NotImplementedError("only --glevel point / --gnD 2 are implemented")
        self.p = float(kd_cfg.get("GP", 2.0))
        self.blur = float(kd_cfg.get("GBLUR", 0.001))
        self.scaling = float(kd_cfg.get("SCALING", 0.5))
        reach = kd_cfg.get("REACH", 0.5)
        self.reach = -1.0 if reach is None else float(reach)
        self.weighted = bool(kd_cfg.get("WEIGHTED_OT", True))
        self.detach = bool(kd_cfg.get("DETACH", False))
        if not self.weighted:
            raise NotImplementedError("unweighted OT (--weightedOT false) is not implemented on the HIP path")
        self.cap = cap
        self.ctx = None
        self._ws = {}
        self._keys = {}
        # device-side sampling keys: reproducible from torch.manual_seed.  The key of a cell is a hash of (seed, step
        # counter, cell index) and train_kd.py seeds every rank alike, so the rank is mixed in: data-parallel ranks
        # must not draw the same key field every step (the reference's torch.randperm runs from one seed on every rank
        # too, but consumes the generator per ground truth of ITS shard, so its ranks decorrelate); the constant also
        # keeps (seed 0, step 0, cell 0) from hashing the all-zero state
        from .libs.distributed import get_rank
        self.seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (get_rank() + 1)) & 0xFFFFFFFFFFFFFFFF
        self.anchor_sizes, self.anchor_strides = ANCHOR_SIZES, ANCHOR_STRIDES      # configs/ape.yaml:3-4

    def workspaces(self, batch, device):
        """The per-step accumulators / slot arrays of the loss side, (fp32, int32), persistent per batch size: the
        step prologue (PoseModuleKD._begin_step) zeroes them together with the gradient bucket."""
        key = (batch, str(device))
        ws = self._ws.get(key)
        if ws is None:
            n, bp = batch * self.cap, (batch + 3) // 4 * 4
            ws = self._ws[key] = (torch.zeros(8 + bp + n * 64 + 16, dtype=torch.float32, device=device),
                                  torch.zeros(3 * bp + 4 + 2 * n, dtype=torch.int32, device=device))
        return ws

    def assign(self, levels, batch, tgt, keys=None, prezeroed=False, step_counter=None):
        """SSC target assignment + the zeroed per-step workspaces.  Depends on the targets only, not on the student's
        output: the graphed step runs it on a side stream beside the student's forward.  prezeroed: the workspaces
        are the persistent pair of workspaces(), already zeroed by the step prologue.  keys: the sampling keys (tests
        pin them); otherwise drawn here -- on the device from (seed, step_counter[0], cell) when the caller has a
        device step counter (graph-replayable without torch's generator), else by torch.rand."""
        dev = tgt.mask.device
        rows = batch * sum(h * w for h, w in levels)
        cap = self.cap
        lv = make_levels(batch, levels, self.anchor_sizes, self.anchor_strides)
        if keys is None and step_counter is not None:
            kb = self._keys.get((rows, str(dev)))
            if kb is None:
                kb = self._keys[(rows, str(dev))] = torch.empty(rows, dtype=torch.float32, device=dev)
            keys = ops.uniform_keys(kb, step_counter, self.seed)
        elif keys is None:
            keys = torch.rand(rows, dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        labels = torch.empty(rows, **i32)
        n = batch * cap
        bp = (batch + 3) // 4 * 4
        # one zero fill per dtype for every per-step accumulator / slot array of the loss side
        if prezeroed:
            wf, wi = self.workspaces(batch, dev)
        else:
            wf = torch.zeros(8 + bp + n * 64 + 16, **f32)     # + the two loss sums' fixed-point workspaces
            wi = torch.zeros(3 * bp + 4 + 2 * n, **i32)
        pos_cnt = wi[0:batch]
        pos_row, pos_gt = wi[3 * bp + 4:3 * bp + 4 + n], wi[3 * bp + 4 + n:3 * bp + 4 + 2 * n]
        P = ops._ptr
        check(lib.kd6d_ssc_assign(ctypes.byref(lv), P(tgt.mask), tgt.mask_h, tgt.mask_w, P(tgt.kp3d), P(tgt.K),
                                  P(tgt.class_ids), P(tgt.n_gt), P(tgt.rot), P(tgt.trans), P(tgt.bbox_trans),
                                  P(keys), self.positive_num, self.positive_lambda, cap, P(labels), P(pos_cnt),
                                  P(pos_row), P(pos_gt), ops._stream()), "kd6d_ssc_assign")
        return dict(rows=rows, levels=tuple(levels), batch=batch, labels=labels, wf=wf, wi=wi, keys=keys)

    def forward(self, cls_s, reg_s, levels, batch, tgt, teacher, keys=None, seg_scale=None, pre=None):
        dev = cls_s.device
        rows = cls_s.shape[0]
        cap = self.cap
        lv = make_levels(batch, levels, self.anchor_sizes, self.anchor_strides)
        if self.diameters is None or self.diameters.device != dev:
            self.diameters = torch.tensor(self.diameters_host, dtype=torch.float32, device=dev)
        if pre is None or pre["rows"] != rows or pre["levels"] != tuple(levels) or pre["batch"] != batch:
            pre = self.assign(levels, batch, tgt, keys)
        labels, wf, wi = pre["labels"], pre["wf"], pre["wi"]
        n = batch * cap
        bp = (batch + 3) // 4 * 4
        pos_cnt, valid, s_start = wi[0:batch], wi[bp:bp + batch], wi[2 * bp:2 * bp + batch]
        n_valid = wi[3 * bp:3 * bp + 1]
        pos_row, pos_gt = wi[3 * bp + 4:3 * bp + 4 + n], wi[3 * bp + 4 + n:3 * bp + 4 + 2 * n]
        st = ops._stream()
        P = ops._ptr
        losses = wf[0:4]                       # cls, reg, kd, (pad)
        loss_img = wf[8:8 + batch]
        o = 8 + bp
        xs, g_reg, g_xs = (wf[o + k * n * 16:o + (k + 1) * n * 16].view(n, 8, 2) for k in range(3))
        alpha, g_alpha = (wf[o + n * 48 + k * n * 8:o + n * 48 + (k + 1) * n * 8].view(n, 8) for k in range(2))
        ws_cls, ws_reg = wf[o + n * 64:o + n * 64 + 8], wf[o + n * 64 + 8:o + n * 64 + 16]     # kd6d_scalar_ws each
        check(lib.kd6d_focal_fwd(P(cls_s), P(labels), rows, self.gamma, self.alpha, P(losses[0:1]), P(ws_cls), st),
              "kd6d_focal_fwd")
        fw, fh = tgt.frame_wh
        check(lib.kd6d_student_points(ctypes.byref(lv), P(cls_s), P(reg_s), P(pos_cnt), P(pos_row), P(pos_gt),
                                      P(tgt.class_ids), P(tgt.kp3d), P(tgt.rot), P(tgt.trans), P(tgt.bbox_trans),
                                      P(self.diameters), self.kinv, fw, fh, cap, P(xs), P(alpha), P(g_reg),
                                      P(losses[1:2]), P(ws_reg), P(s_start), st), "kd6d_student_points")
        if teacher is not None:
            check(lib.kd6d_sinkhorn_div_fwd_bwd(P(xs), P(alpha), P(s_start), P(pos_cnt), P(teacher.t_kp_norm),
                                                P(teacher.t_beta), P(teacher.t_start), P(teacher.t_cnt), batch, self.p,
                                                self.blur, self.scaling, self.reach, P(loss_img), P(valid), None,
                                                P(g_xs), P(g_alpha), st), "kd6d_sinkhorn_div_fwd_bwd")
            check(lib.kd6d_kd_mean(P(loss_img), P(valid), batch, P(losses[2:3]), P(n_valid), st), "kd6d_kd_mean")
        self.ctx = dict(lv=lv, cls=cls_s, reg=reg_s, labels=labels, pos_cnt=pos_cnt, pos_row=pos_row, pos_gt=pos_gt,
                        tgt=tgt, g_reg=g_reg, g_xs=g_xs, g_alpha=g_alpha, n_valid=n_valid, valid=valid,
                        rows=rows, batch=batch, xs=xs, alpha=alpha, seg_scale=seg_scale, losses=losses)
        return losses

    def backward(self, weights, dtype, dcls, dreg, dseg_scale=None, acc_stride=0):
        """weights: device fp32 tensor {d/d loss_cls, d/d loss_reg, d/d loss_kd}.  dreg must be zeroed.  dseg_scale:
        lo-plane view of the PLANAR gradient accumulators of the head's scales (int64, stride acc_stride), or None."""
        assert dseg_scale is None or (dseg_scale.dtype == torch.int64 and acc_stride > 0)
        c = self.ctx
        P = ops._ptr
        st = ops._stream()
        tgt = c["tgt"]
        fw, fh = tgt.frame_wh
        check(lib.kd6d_focal_bwd(ops.dt_code(dtype), P(c["cls"]), P(c["labels"]), c["rows"], self.gamma, self.alpha,
                                 P(weights[0:1]), P(dcls), st), "kd6d_focal_bwd")
        check(lib.kd6d_loss_backward(ctypes.byref(c["lv"]), ops.dt_code(dtype), P(c["cls"]), P(c["reg"]),
                                     P(c["pos_cnt"]), P(c["pos_row"]), P(c["pos_gt"]), P(tgt.class_ids),
                                     P(tgt.bbox_trans), P(c["g_reg"]), P(c["g_xs"]), P(c["g_alpha"]),
                                     P(c["n_valid"]), P(c["valid"]), P(weights), P(c["seg_scale"]),
                                     P(dseg_scale), int(acc_stride), fw, fh, self.cap, int(self.detach), P(dcls), P(dreg), st),
              "kd6d_loss_backward")
