"""Static executor of the pose network on the kd6d C ABI.

One `PoseNet` = backbone (darknet53 | darknet_tiny | darknet_tiny_h) + FPN + PoseHead laid out
for MI355X: packed NHWC activations, the head run over ALL pyramid levels in one launch per layer,
parameters / gradients / Adam state in single flat fp32 buffers (one fused optimiser launch, one
RCCL all-reduce bucket), a bf16 shadow of the parameters for the bf16 MFMA path, and a hand-written
reverse sweep instead of an autograd tape (the graph is static).

Reference being replaced: models/model.py:455-487 (PoseModule.__init__), backbone/darknet*.py,
models/model.py:40-103 (FPN), :370-451 (PoseHead).  Parameter NAMES and logical shapes follow the
reference state_dict; storage is KRSC with channels padded to multiples of 8.
"""
import math

import torch

from . import ops
from .ops import ACT_LEAKY, ACT_NONE, ACT_RELU

ANCHOR_SIZES = [32, 64, 128, 256, 512]
ANCHOR_STRIDES = [8, 16, 32, 64, 128]
TINY_CHANNELS = {
    "darknet_tiny": [[16], [32], [16, 128, 16, 128], [32, 256, 32, 256], [64, 512, 64, 512, 128]],
    "darknet_tiny_h": [[8], [16], [8, 64, 8, 64], [16, 128, 16, 128], [32, 256, 32, 256, 64]],
}
BACKBONE_CFG = {  # arguments/argument.py:59-68
    "darknet53": ([0, 0, 256, 512, 1024], 256),
    "darknet_tiny": ([0, 0, 128, 128], 256),
    "darknet_tiny_h": ([0, 0, 64, 64], 128),
}


def _pad8(c):
    return (c + 7) // 8 * 8


# ----------------------------------------------------------------------------------------
# flat parameter store
# ----------------------------------------------------------------------------------------
class Entry:
    __slots__ = ("name", "kind", "shape", "store_shape", "offset", "numel", "trainable", "region", "det", "slab", "parts")

    def __init__(self, name, kind, shape, store_shape, trainable):
        self.name, self.kind, self.shape, self.store_shape = name, kind, tuple(shape), tuple(store_shape)
        self.numel = int(math.prod(store_shape))
        self.trainable = trainable
        self.offset = -1
        self.region = None
        self.det = False        # its gradient is summed in the fixed-point accumulator image (ParamStore.gacc)
        self.slab = None        # ... or arrives as partial images in this fp32 (parts, numel) slab (per-layer weight gradients)
        self.parts = 0


class ParamStore:
    """Flat fp32 buffers.  Regions: 'train' (optimised), 'frozen' (registered-but-unused parameters
    such as backbone.output.* and head.scales.4 of a 4-level student: the reference never gives them
    a gradient, so AdamW never touches them), 'buf' (BN running stats)."""

    ALIGN = 8

    def __init__(self):
        self.entries = {}
        self.order = []
        self.device = torch.device("cpu")
        self.finalized = False

    def add(self, name, kind, shape, store_shape=None, trainable=True, region=None):
        assert not self.finalized and name not in self.entries
        e = Entry(name, kind, shape, store_shape or shape, trainable)
        e.region = region or ("train" if trainable else "frozen")
        self.entries[name] = e
        self.order.append(e)
        return e

    def finalize(self):
        sizes = {"train": 0, "frozen": 0, "buf": 0}
        for e in self.order:
            e.offset = sizes[e.region]
            sizes[e.region] += (e.numel + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.sizes = sizes
        self.n_train = sizes["train"]
        n_par = sizes["train"] + sizes["frozen"]
        self.params = torch.zeros(max(n_par, 8), dtype=torch.float32)
        self.bufs = torch.zeros(max(sizes["buf"], 8), dtype=torch.float32)
        self.grads = None
        self.gacc = None
        self._resolve_plans = {}
        self.shadow = None
        self.finalized = True

    def base(self, e):
        return e.offset + (self.n_train if e.region == "frozen" else 0)

    def storage(self, e, which="params"):
        """1-D view of the entry inside a flat buffer."""
        if e.region == "buf":
            return self.bufs[e.offset:e.offset + e.numel]
        flat = getattr(self, which)
        b = self.base(e)
        return flat[b:b + e.numel]

    def logical_view(self, e, which="params"):
        """View with the reference's logical shape (OIHW for convs) sharing the flat storage."""
        s = self.storage(e, which)
        if e.kind == "conv":
            co, ci, kh, kw = e.shape
            cop, _, _, cip = e.store_shape
            return s.view(cop, kh, kw, cip)[:co, :, :, :ci].permute(0, 3, 1, 2)
        return s.view(e.store_shape)[tuple(slice(0, d) for d in e.shape)]

    def to(self, device):
        self.device = torch.device(device)
        self.params = self.params.to(device)
        self.bufs = self.bufs.to(device)
        if self.grads is not None:
            self.grads = self.grads.to(device)
        if self.gacc is not None:
            self.gacc = self.gacc.to(device)
        for e in self.order:
            e.slab = None
        self._resolve_plans = {}
        if self.shadow is not None:
            self.shadow = self.shadow.to(device)

    def ensure_grads(self):
        if self.grads is None:
            self.grads = torch.zeros(max(self.n_train, 8), dtype=torch.float32, device=self.params.device)
        if self.gacc is None:
            # Fixed-point accumulator image of the gradient bucket (csrc/kd6d_det.h, include/kd6d.h "reproducible
            # reductions"): PLANAR, the lo words of all elements then the hi words.  Every gradient that is a sum over
            # workgroups (per-layer weight gradients, bias gradients, GroupNorm gains / shifts, the head's scales) is
            # added here with integer atomics -- bitwise reproducible -- and resolve_grads() turns the touched regions
            # into fp32 gradients once per step, leaving them zero for the next one.
            self.gacc = torch.zeros(2 * self.acc_stride, dtype=torch.int64, device=self.params.device)

    @property
    def acc_stride(self):
        """Distance (int64 words) between the lo and the hi word of a gradient accumulator."""
        return max(self.n_train, 8)

    def acc(self, e):
        """lo-plane view of the entry's gradient accumulators (pass with acc_stride to the kd6d entry points); marks
        the entry for resolve_grads()."""
        assert e.region == "train"
        self.ensure_grads()
        if not e.det:
            e.det = True
            self._resolve_plans = {}
        b = self.base(e)
        return self.gacc[b:b + e.numel]

    def slab(self, e, parts):
        """fp32 (parts, numel) slab the per-layer weight-gradient launch of entry e stores its partial images into; marks
        the entry for resolve_grads().  One slab per entry, kept for the life of the store (recorded graphs hold it)."""
        assert e.region == "train" and not e.det
        if e.slab is None or e.slab.shape[0] < parts:
            assert not torch.cuda.is_current_stream_capturing(), "ParamStore.slab: allocation inside a graph capture"
            e.slab = torch.empty(parts, e.numel, dtype=torch.float32, device=self.params.device)
            e.parts = 0
        if e.parts != parts:            # another geometry / budget than the last launch: the region table changes
            assert not torch.cuda.is_current_stream_capturing(), \
                "ParamStore.slab: a new split count inside a graph capture (run one eager warm-up step first)"
            e.parts = parts
            self._resolve_plans = {}
        return e.slab[:parts]

    def resolve_grads(self, lo=0, hi=None):
        """grads[lo:hi] += the gradients that were summed across workgroups, for every marked entry in that range:
        fixed-point accumulators (cleared) and partial-image slabs (added in part order).  One launch
        (kd6d_grad_acc_resolve); the region table is built on first use -- outside any stream capture."""
        hi = self.n_train if hi is None else hi
        plan = self._resolve_plans.get((lo, hi))
        if plan is None:
            assert not torch.cuda.is_current_stream_capturing(), \
                "ParamStore.resolve_grads: first use of a new set of accumulated gradients inside a graph capture " \
                "(run one eager warm-up step first)"
            regs = []           # [first, count, parts, slab address]
            for e in self.order:
                if e.region != "train" or not (lo <= self.base(e) < hi):
                    continue
                b, n = self.base(e), e.numel
                if e.slab is not None and e.parts > 0:
                    regs.append([b, n, e.parts, e.slab.data_ptr()])
                elif e.det:
                    if regs and regs[-1][2] == 0 and regs[-1][0] + regs[-1][1] == b:
                        regs[-1][1] += n
                    else:
                        regs.append([b, n, 0, 0])
            desc, blk = [], 0
            for b, n, parts, ptr in regs:
                desc += [b, n, blk, parts, ptr]
                per = 1024 // (ops.lib.kd6d_grad_acc_resolve_part_groups(parts) if parts else 1)     # elements per workgroup
                blk += (n + per - 1) // per
            plan = (torch.tensor(desc or [0] * 5, dtype=torch.int64, device=self.params.device), len(regs), blk)
            self._resolve_plans[(lo, hi)] = plan
        desc, n_regions, blocks = plan
        if n_regions == 0:
            return
        ops.check(ops.lib.kd6d_grad_acc_resolve(ops._ptr(desc), n_regions, blocks, ops._ptr(self.gacc), self.acc_stride,
                                                ops._ptr(self.grads), ops._stream()), "kd6d_grad_acc_resolve")

    def ensure_shadow(self):
        if self.shadow is None:
            self.shadow = torch.empty(self.params.numel(), dtype=torch.bfloat16, device=self.params.device)
        return self.shadow

    def refresh_shadow(self):
        sh = self.ensure_shadow()
        ops.check(ops.lib.kd6d_cast_f32_to_bf16(ops._ptr(self.params), ops._ptr(sh), self.params.numel(),
                                                ops._stream()), "kd6d_cast_f32_to_bf16")


# ----------------------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------------------
class Conv:
    """conv2d (+ optional bias) over 1..5 packed levels.  Weight entry is KRSC-stored."""

    def __init__(self, net, name, cin, cout, k, stride=1, bias=True, trainable=True):
        self.net, self.name = net, name
        self.cin, self.cout, self.k, self.stride = cin, cout, k, stride
        self.cin_p, self.cout_p = _pad8(cin), _pad8(cout)
        st = net.store
        self.w = st.add(name + ".weight", "conv", (cout, cin, k, k), (self.cout_p, k, k, self.cin_p), trainable)
        self.b = st.add(name + ".bias", "vec", (cout,), (self.cout_p,), trainable) if bias else None
        self.geoms = {}
        self._fusable = {}
        self._parts = {}
        self.wt_off = None     # offset inside the dgrad-packed weight buffer
        net.convs.append(self)

    def geom(self, batch, levels):
        key = (batch, tuple(levels))
        g = self.geoms.get(key)
        if g is None:
            g = ops.Geom(batch, self.cin_p, self.cout_p, self.k, self.stride, self.k // 2, list(levels))
            self.geoms[key] = g
        return g

    # pointers into the flat buffers ------------------------------------------------------
    def weight(self):
        st = self.net.store
        src = st.params if self.net.dtype == torch.float32 else st.shadow
        b = st.base(self.w)
        return src[b:b + self.w.numel]

    def bias(self):
        return None if self.b is None else self.net.store.storage(self.b)

    def weight_t(self):
        return self.net.wt[self.wt_off:self.wt_off + self.w.numel]

    def flops(self, g):
        """Algorithmic FLOPs of one pass (2*MAC on the reference's logical channel counts)."""
        return 2 * g.rows_out * self.cout * self.k * self.k * self.cin

    def fwd(self, x, batch, levels, out=None, scale=None, shift=None, act=ACT_NONE, residual=None,
            seg_scale=None, out_f32=False, stats=None, stats_groups=0):
        g = self.geom(batch, levels)
        if shift is None:
            shift = self.bias()
        return ops.conv2d_fwd(g, x, self.weight(), out=out, ch_scale=scale, ch_shift=shift, act=act,
                              residual=residual, seg_scale=seg_scale, out_f32=out_f32, flops=self.flops(g),
                              stats=stats, stats_groups=stats_groups, workspace=self.net.workspace()), g

    def norm_fusable(self, batch, levels, kind, groups=0):
        """Does this layer take the conv + normalisation + activation launch (kd6d_conv2d_fwd_norm)?  Cached per
        geometry and value of the `conv.fuse_norm` option."""
        key = (batch, tuple(levels), kind, groups, self.net.dtype, ops.get_option("conv.fuse_norm"))
        ok = self._fusable.get(key)
        if ok is None:
            ok = self._fusable[key] = ops.conv_norm_fusable(self.geom(batch, levels), self.net.dtype, kind, groups)
        return ok

    def fwd_norm(self, x, batch, levels, y, kind, gamma, beta, act, raw_out=None, groups=0, stats=None, **bn):
        """conv (+ bias) -> normalisation -> activation in one launch; stats / counters live in the per-step zeroed arena."""
        g = self.geom(batch, levels)
        net = self.net
        if stats is None:
            stats = net.scratch(self.name + ".fstats", ops.conv_norm_stats_floats(g, kind, groups))
        counters = net.scratch(self.name + ".fctr", ops.conv_norm_counter_words(g, kind))
        return ops.conv2d_fwd_norm(g, x, self.weight(), y, kind, gamma, beta, stats, counters, act, raw_out=raw_out,
                                   bias=self.bias(), groups=groups, flops=self.flops(g), **bn), g

    def wgrad_slab(self, g, dtype, cu_budget):
        """The slab this layer's weight-gradient launch stores its partial images into (ParamStore.slab)."""
        key = (id(g), dtype, cu_budget, ops.get_option("wgrad.small"))
        parts = self._parts.get(key)
        if parts is None:
            parts = self._parts[key] = ops.conv2d_wgrad_parts(g, dtype, self.b is not None, cu_budget)
        return self.net.store.slab(self.w, parts)

    def bwd(self, x, dy, batch, levels, need_dx=True, dx=None, accumulate=False, need_dw=True):
        """wgrad (+ bias grad) into the flat grad buffer, then dgrad."""
        st = self.net.store
        g = self.geom(batch, levels)
        grp = self.net.wgrad_group if self.net.grouping else None
        if not need_dw:
            pass
        elif grp is not None and x.dtype == torch.bfloat16 and ops.wgrad_group_supported(g, x.dtype):
            # collected: one launch pair for the whole network section (PoseNet.flush_wgrad_group); x and dy are
            # per-layer buffers that stay untouched until the sweep has joined its side streams
            grp.add(g, x, dy, st.storage(self.w, "grads"), None if self.b is None else st.storage(self.b, "grads"),
                    flops=self.flops(g))
        elif (side := self.net.next_side_stream()) is None:
            ops.conv2d_wgrad(g, x, dy, self.wgrad_slab(g, x.dtype, 0), flops=self.flops(g),
                             dbias=None if self.b is None else st.acc(self.b), acc_stride=st.acc_stride)
        else:
            # the weight gradient feeds nothing in the reverse sweep: fork it onto the side stream so it
            # overlaps the dgrad / normalisation chain (both under-fill 256 CUs at these layer sizes);
            # PoseNet.backward joins before the gradient exchange.  x and dy are per-layer buffers.
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ops.conv2d_wgrad(g, x, dy, self.wgrad_slab(g, x.dtype, self.net.wgrad_cu_budget), flops=self.flops(g),
                                 dbias=None if self.b is None else st.acc(self.b), acc_stride=st.acc_stride,
                                 cu_budget=self.net.wgrad_cu_budget)
        if not need_dx:
            return None
        return ops.conv2d_dgrad(g, dy, self.weight_t(), dx=dx, accumulate=accumulate, flops=self.flops(g))


class BatchNorm:
    def __init__(self, net, name, c, trainable=True):
        st = net.store
        self.net, self.c = net, c
        self.gamma = st.add(name + ".weight", "vec", (c,), (c,), trainable)
        self.beta = st.add(name + ".bias", "vec", (c,), (c,), trainable)
        self.rm = st.add(name + ".running_mean", "vec", (c,), (c,), False, region="buf")
        self.rv = st.add(name + ".running_var", "vec", (c,), (c,), False, region="buf")
        self.nbt_index = len(net.bns)
        net.bns.append((name, self))
        self.fold = None

    def folded(self):
        """eval-mode scale/shift (computed once per weight load; torch ops = plumbing)."""
        if self.fold is None:
            st = self.net.store
            scale = st.storage(self.gamma) * torch.rsqrt(st.storage(self.rv) + 1e-5)
            shift = st.storage(self.beta) - st.storage(self.rm) * scale
            self.fold = (scale.contiguous(), shift.contiguous())
        return self.fold


class ConvBlock:
    """conv(no bias) + BN + LeakyReLU(0.1)  (backbone/common.py:250-324)."""
    # up to 256 workgroups adding into one channel; beyond that a separate pass wins
    FUSE_STATS_MAX_ROWS = 1 << 15

    @staticmethod
    def bwd_replicas(rows):
        """Replica rows of the backward reduction's accumulators (kd6d.h): from 64 workgroups up."""
        return 8 if rows >= (1 << 14) else 1

    def scratch_floats(self, rows):
        """fp32 slots of the block's slice of the per-step zeroed arena: {sum, sumsq} accumulators of the batch statistics,
        saved mean / invstd, the backward's R replica rows of two accumulator sums, its barrier words."""
        c, A, R = self.conv.cout_p, ops.ACC_FLOATS, self.bwd_replicas(rows)
        return (2 * A + 2 + 2 * A * R) * c + ops.BARRIER_WORDS

    def __init__(self, net, name, cin, cout, k, stride=1):
        self.net, self.name = net, name
        self.conv = Conv(net, name + ".conv", cin, cout, k, stride, bias=False)
        self.bn = BatchNorm(net, name + ".bn", self.conv.cout_p)
        assert self.conv.cout_p == cout, "BN channels must already be a multiple of 8"

    def fwd_eval(self, x, batch, levels, residual=None):
        sc, sh = self.bn.folded()
        y, g = self.conv.fwd(x, batch, levels, scale=sc, shift=sh, act=ACT_LEAKY, residual=residual)
        return y, g.levels_out

    def fwd_train(self, x, batch, levels, tape, pool=False, pending=None, defer=False):
        """pool=True: the block is followed by MaxPool2d(2,2) (darknet.py:94-97); normalisation, activation and
        pooling run as one kernel and only the pooled tensor is stored (csrc/norm_ops.hip, bn_pool_*).
        defer=True: this block's BatchNorm + LeakyReLU is NOT launched here -- the next block of the stage applies it
        while it loads its input (kd6d_conv2d_fwd_block); returns (None, levels, pending record for that block).
        pending: the record of the previous block when ITS normalisation was deferred to this convolution (x is None)."""
        net, st = self.net, self.net.store
        # the pre-BN tensor stays fp32 (also in bf16 mode): (x - mean) must not cancel bf16 rounding
        c = self.conv.cout_p
        geom = self.conv.geom(batch, levels)
        A = ops.ACC_FLOATS         # fp32 slots of one accumulator: the sums are kd6d_acc (reproducible reductions, kd6d.h)
        s = net.scratch(self.name, self.scratch_floats(geom.rows_out))
        ssum, ssq = s[0:A * c], s[A * c:2 * A * c]
        mean, invstd = s[2 * A * c:(2 * A + 1) * c], s[(2 * A + 1) * c:(2 * A + 2) * c]
        if pending is not None or defer:
            raw = net.buf(self.name + ".raw", (geom.rows_out, c), torch.float32)
            bn_in = z_in = None
            if pending is not None:
                prev, raw_prev, sums_prev, mean_prev, invstd_prev = pending
                pst = prev.bn
                bn_in = dict(sums=sums_prev, replicas=ops.BN_REPLICAS, gamma=st.storage(pst.gamma), beta=st.storage(pst.beta),
                             act=ACT_LEAKY, eps=1e-5, momentum=0.1, running_mean=st.storage(pst.rm),
                             running_var=st.storage(pst.rv), save_mean=mean_prev, save_invstd=invstd_prev)
                z_in = net.buf(prev.name + ".z", raw_prev.shape, net.dtype)     # written by this launch: what wgrad reads
                x = raw_prev
            if defer:          # the consumer adds the replica rows
                stats, reps = net.scratch(self.name + ".sums", ops.BN_REPLICAS * 2 * c * A), ops.BN_REPLICAS
            else:              # bn_train_fwd / bn_pool_train_fwd below read plain {sum, sumsq}
                stats, reps = s[0:2 * A * c], 1
            fused = defer or geom.rows_out <= self.FUSE_STATS_MAX_ROWS
            ops.conv2d_fwd_block(geom, x, self.conv.weight(), raw, stats=stats if fused else None, stats_replicas=reps,
                                 bn_in=bn_in, z_out=z_in, flops=self.conv.flops(geom))
            g = geom
            x_saved = z_in if pending is not None else x
            if defer:
                tape.append((self, x_saved, raw, batch, tuple(levels), None))
                return None, g.levels_out, (self, raw, stats, mean, invstd)
            if not fused:
                ops.colstats(raw, ssum, ssq)
            x = x_saved
        else:
            if not pool and net.fuse_norm_on() and self.conv.norm_fusable(batch, levels, ops.NORM_BATCH):
                # conv -> batch statistics -> normalise + LeakyReLU as ONE launch (grid barrier in the epilogue)
                raw = net.buf(self.name + ".raw", (geom.rows_out, c), torch.float32)
                z = net.buf(self.name + ".z", raw.shape, net.dtype)
                _, g = self.conv.fwd_norm(x, batch, levels, z, ops.NORM_BATCH, st.storage(self.bn.gamma), st.storage(self.bn.beta),
                                          ACT_LEAKY, raw_out=raw, eps=1e-5, momentum=0.1, running_mean=st.storage(self.bn.rm),
                                          running_var=st.storage(self.bn.rv), save_mean=mean, save_invstd=invstd)
                tape.append((self, x, raw, batch, tuple(levels), None))
                return z, g.levels_out, None
            # batch statistics come out of the conv epilogue; for the long, narrow first layers (hundreds of
            # workgroups would add into the same 8..64 addresses) a separate reduction pass is cheaper
            fused = geom.rows_out <= self.FUSE_STATS_MAX_ROWS
            raw, g = self.conv.fwd(x, batch, levels, out_f32=True,
                                   out=net.buf(self.name + ".raw", (geom.rows_out, c), torch.float32),
                                   stats=s[0:2 * A * c] if fused else None, stats_groups=0)
            if not fused:
                ops.colstats(raw, ssum, ssq)
        if pool:
            (h, w), = g.levels_out
            z = net.buf(self.name + ".zpool", (batch * (h // 2) * (w // 2), c), net.dtype)
            ops.bn_pool_train_fwd(raw, z, batch, h, w, ssum, ssq, st.storage(self.bn.gamma), st.storage(self.bn.beta),
                                  1e-5, 0.1, st.storage(self.bn.rm), st.storage(self.bn.rv), mean, invstd, ACT_LEAKY)
            tape.append((self, x, raw, batch, tuple(levels), (h, w)))
            return z, [(h // 2, w // 2)], None
        z = net.buf(self.name + ".z", raw.shape, net.dtype)
        ops.bn_train_fwd(raw, z, ssum, ssq, st.storage(self.bn.gamma), st.storage(self.bn.beta), 1e-5, 0.1,
                         st.storage(self.bn.rm), st.storage(self.bn.rv), mean, invstd, ACT_LEAKY)
        tape.append((self, x, raw, batch, tuple(levels), None))
        return z, g.levels_out, None

    def bwd(self, rec, dz, need_dx=True, dx=None, accumulate=False):
        _, x, raw, batch, levels, pooled = rec
        net, st = self.net, self.net.store
        c = self.conv.cout_p
        R, A = self.bwd_replicas(raw.shape[0]), ops.ACC_FLOATS
        s = net.scratch(self.name, self.scratch_floats(raw.shape[0]))
        mean, invstd = s[2 * A * c:(2 * A + 1) * c], s[(2 * A + 1) * c:(2 * A + 2) * c]
        o = (2 * A + 2) * c
        w1, w2 = s[o:o + A * R * c], s[o + A * R * c:o + 2 * A * R * c]
        counter = s[o + 2 * A * R * c:o + 2 * A * R * c + ops.BARRIER_WORDS]      # zeroed with the arena at the start of the step
        draw = net.buf(self.name + ".draw", raw.shape, net.dtype)
        if pooled is not None:                  # dz is the gradient of the pooled output
            ops.bn_pool_train_bwd(raw, dz, draw, batch, pooled[0], pooled[1], mean, invstd, st.storage(self.bn.gamma),
                                  st.storage(self.bn.beta), ACT_LEAKY, w1, w2, st.storage(self.bn.gamma, "grads"),
                                  st.storage(self.bn.beta, "grads"), replicas=R, counter=counter)
        else:
            ops.bn_train_bwd(raw, dz, draw, mean, invstd, st.storage(self.bn.gamma), st.storage(self.bn.beta),
                             ACT_LEAKY, w1, w2, st.storage(self.bn.gamma, "grads"), st.storage(self.bn.beta, "grads"),
                             replicas=R, counter=counter)
        return self.conv.bwd(x, draw, batch, levels, need_dx=need_dx, dx=dx, accumulate=accumulate)


class GroupNormReLU:
    def __init__(self, net, name, c, groups=32):
        st = net.store
        self.net, self.name, self.c, self.groups = net, name, c, groups
        self.gamma = st.add(name + ".weight", "vec", (c,), (c,), True)
        self.beta = st.add(name + ".bias", "vec", (c,), (c,), True)

    def stats(self, batch, levels):
        """{sum, sumsq} per (level, image, group) in the per-step zeroed scratch arena; filled by the
        epilogue of the conv that produces this layer's input."""
        return self.net.scratch(self.name + ".stats", len(levels) * batch * self.groups * 2 * ops.ACC_FLOATS)

    def fwd(self, x, batch, levels, out=None):
        net, st = self.net, self.net.store
        hw = [h * w for (h, w) in levels]
        if out is None:
            out = net.buf(self.name + ".y", x.shape)
        ops.gn_relu_fwd(x, out, hw, batch, self.groups, st.storage(self.gamma), st.storage(self.beta), 1e-5,
                        self.stats(batch, levels), flags=ops.GN_STATS_READY)
        return out

    def fwd_fused(self, conv, x, batch, levels, out, raw_out=None):
        """conv -> GroupNorm -> ReLU as one launch (the conv's epilogue); the statistics land where bwd expects them."""
        st = self.net.store
        conv.fwd_norm(x, batch, levels, out, ops.NORM_GROUP, st.storage(self.gamma), st.storage(self.beta), ACT_RELU,
                      raw_out=raw_out, groups=self.groups, eps=1e-5, stats=self.stats(batch, levels))
        return out

    def bwd_item(self, x, dz, batch, levels, dx):
        """The argument tuple of ops.gn_relu_bwd / gn_relu_bwd_pair for this layer."""
        net, st = self.net, self.net.store
        gsum = net.scratch(self.name + ".gsum", ops.gn_bwd_workspace_floats(len(levels), batch, self.groups))
        return (x, dz, dx, st.storage(self.gamma), st.storage(self.beta), self.stats(batch, levels), gsum,
                st.acc(self.gamma), st.acc(self.beta))

    def bwd(self, x, dz, batch, levels, dx):
        hw = [h * w for (h, w) in levels]
        x, dz, dx, gamma, beta, stats, gsum, dgamma, dbeta = self.bwd_item(x, dz, batch, levels, dx)
        ops.gn_relu_bwd(x, dz, dx, hw, batch, self.groups, gamma, beta, stats, gsum, dgamma, dbeta,
                        self.net.store.acc_stride, flags=ops.GN_WS_ZEROED)
        return dx


# ----------------------------------------------------------------------------------------
# the network
# ----------------------------------------------------------------------------------------
class PoseNet:
    def __init__(self, arch, dtype=torch.bfloat16, n_class=16, n_conv=4, prior=0.01):
        assert arch in BACKBONE_CFG, "Unsupported backbone %r" % (arch,)
        self.arch, self.dtype = arch, dtype
        self.store = ParamStore()
        self.convs, self.bns = [], []
        self._bufs, self._scratch_off, self._scratch_size, self.scratch_buf = {}, {}, 0, None
        self.wt = None
        self.training = True
        self.in_hw = None
        self.levels, self.batch = None, None
        self.side_stream = None        # set (e.g. by GraphedKDStep) to run weight gradients concurrently
        self.side_streams = None       # optional list: consecutive weight gradients rotate over these streams
        self.wgrad_cu_budget = 0       # CUs each forked weight gradient aims to fill (0 = the device)
        # the weight gradients of the head and FPN output convolutions as ONE grouped launch per step
        # (csrc/conv_wgrad_group.hip); created on first use, bf16 only.  wgrad_group_wgs: workgroups of that launch
        # (0 = one per two CUs).  Measured on the step (images/s, 64 / 128 / 256 workgroups): pipelined 4831 / 4828 /
        # 4788, strictly sequential 4050 / 4164 / 4241 -- GraphedKDStep asks for one per CU in sequential mode
        # set by GraphedKDStep around its own calls only: the packed NHWC input to use instead of converting `images`
        # (nhwc_in), or where to put the conversion (nhwc_out)
        self.nhwc_in = self.nhwc_out = None
        self.wgrad_group = None
        self.grouping = False          # True while the reverse sweep is inside the section whose dW are collected
        self.use_wgrad_group = True
        self.wgrad_group_wgs = 0
        # GroupedTeacherKDStep: called at layer-group boundaries of an eval-mode forward (the places where the frozen
        # teacher's pass over several steps' batches may be cut into per-step graph segments)
        self.cut_hook = None
        self.wgrad_group_flush = "head_end"     # or "fpn_end" (GraphedKDStep picks by launch mode)
        self.grad_hook = None           # called by backward() when the FPN + head gradients have been issued
        self.resolve_hi = None          # backward() resolves the accumulated gradients of [0, resolve_hi) at its end (None: all)
        self.fuse_pool = True           # BN + act + maxpool as one kernel (training)
        # conv + normalisation + activation as one launch (kd6d_conv2d_fwd_norm: in-kernel barrier in the conv epilogue).
        # Measured on the step, interleaved runs on one box (profiles/README.md round 3): the frozen teacher's towers
        # fused (the fp32 pre-normalisation tensor is then not even stored) +0.4 ... +1.1 % on config 2 but -1.8 % on
        # config 4's shard and -1.2 % on full frames; the student's towers fused -3 %, its BatchNorm blocks -1.2 %.  A
        # workgroup that waits at a barrier keeps its LDS and wave slots while the other stream's kernels could use them:
        # a kernel boundary is the cheaper barrier when two streams share the device.  Off (None = off); the entry point,
        # its tests and the residency argument (csrc/kd6d_barrier.h) stay for single-stream callers.
        self.fuse_norm = None
        # inside a stage of the student's backbone a block's BatchNorm + LeakyReLU can be applied by the NEXT block's
        # convolution while it loads its input (kd6d_conv2d_fwd_block): up to 10 of the 15 normalise launches (and the 3
        # separate statistics passes of stage 3) leave the forward chain without any wait inside a kernel.  Measured
        # (interleaved runs on one box, round 3): 0 -> 5192 / 5212 images/s, 1 -> 5151 / 5172, 2 -> 5187: the
        # register-staged kernel that can transform on load gives back what the removed launches save (the 3x3 layers
        # leave the LDS-DMA kernels and re-read the fp32 tensor once per tap).  Off by default; kept, tested, for
        # layer mixes where the normalise launches weigh more.
        self.bn_on_load = 0            # 0 off | 1 where the next block is a 1x1 convolution | 2 every in-stage transition
        # cls / pose tower layers as one launch (training): 0 = off, 1 = forward and data gradients, 2 = forward only
        self.pair_towers = 1
        self._side_rr = 0
        feat, oc = BACKBONE_CFG[arch]
        self.out_channel = oc
        self.n_levels = 5 if arch == "darknet53" else 4
        self.n_cls = n_class - 1
        st = self.store
        # ---- backbone ----
        if arch == "darknet53":
            self.init_block = ConvBlock(self, "backbone.features.init_block", 3, 32, 3)
            self.stages = []
            cin = 32
            for i, (c, n) in enumerate(zip([64, 128, 256, 512, 1024], [2, 3, 9, 9, 5])):
                units = []
                for j in range(n):
                    nm = "backbone.features.stage%d.unit%d" % (i + 1, j + 1)
                    if j == 0:
                        units.append(("down", ConvBlock(self, nm, cin, c, 3, 2)))
                    else:
                        units.append(("res", ConvBlock(self, nm + ".conv1", cin, c // 2, 1),
                                      ConvBlock(self, nm + ".conv2", c // 2, c, 3)))
                    cin = c
                self.stages.append(units)
            st.add("backbone.output.weight", "vec", (1000, 1024), (1000, 1024), False)
            st.add("backbone.output.bias", "vec", (1000,), (1000,), False)
        else:
            self.stages = []
            cin = 3
            chans = TINY_CHANNELS[arch]
            for i, per_stage in enumerate(chans):
                units = []
                for j, c in enumerate(per_stage):
                    pointwise = len(per_stage) > 1 and (j % 2 == 0)     # darknet.py:92 with odd_pointwise
                    units.append(ConvBlock(self, "backbone.features.stage%d.unit%d" % (i + 1, j + 1), cin, c,
                                           1 if pointwise else 3))
                    cin = c
                self.stages.append(units)
            st.add("backbone.output.final_conv.weight", "vec", (1000, cin, 1, 1), (1000, cin, 1, 1), False)
            st.add("backbone.output.final_conv.bias", "vec", (1000,), (1000,), False)
        # ---- FPN ----
        self.inner, self.outc = {}, {}
        for i, c in enumerate(feat):
            if c == 0:
                continue
            self.inner[i] = Conv(self, "fpn.inner_convs.%d" % i, c, oc, 1)
            self.outc[i] = Conv(self, "fpn.out_convs.%d" % i, oc, oc, 3)
        self.p6 = Conv(self, "fpn.top_blocks.p6", feat[-1], oc, 3, 2)
        self.p7 = Conv(self, "fpn.top_blocks.p7", oc, oc, 3, 2)
        # ---- head ----
        self.cls_tower, self.pose_tower = [], []
        for t, tower in (("cls_tower", self.cls_tower), ("pose_tower", self.pose_tower)):
            for i in range(n_conv):
                tower.append((Conv(self, "head.%s.%d" % (t, 3 * i), oc, oc, 3),
                              GroupNormReLU(self, "head.%s.%d" % (t, 3 * i + 1), oc)))
        self.cls_logits = Conv(self, "head.cls_logits", oc, self.n_cls, 3)
        self.pose_pred = Conv(self, "head.pose_pred", oc, self.n_cls * 16, 3)
        self.scales = st.add("head.scales", "vec", (8,), (8,), True)          # slots 0..n_levels-1 used
        for l in range(5):
            if l >= self.n_levels:
                st.add("head.scales.%d.scale" % l, "vec", (1,), (1,), False)
        self.prior = prior
        st.finalize()
        self.reset_parameters()

    # ---- parameters ----------------------------------------------------------------------
    def named_logical(self):
        """(reference name, logical-shape view) for every state_dict tensor except the anchors."""
        st = self.store
        out = []
        for e in st.order:
            if e.name == "head.scales":
                for l in range(self.n_levels):
                    out.append(("head.scales.%d.scale" % l, st.storage(e)[l:l + 1], True))
                continue
            out.append((e.name, st.logical_view(e), e.region != "buf"))
        return out

    def reset_parameters(self, seed=None):
        """Reference initialisation (darknet*.py:_init_params, model.py:22-37,420-433)."""
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        st = self.store
        for e in st.order:
            v = st.logical_view(e) if e.name != "head.scales" else st.storage(e)
            if e.name == "head.scales" or e.name.endswith(".scale"):
                v.fill_(1.0)
            elif e.name.endswith("running_var"):
                v.fill_(1.0)
            elif e.name.endswith("running_mean"):
                v.zero_()
            elif e.kind == "conv":
                fan_in = e.shape[1] * e.shape[2] * e.shape[3]
                if e.name.startswith("head."):
                    v.copy_(torch.randn(e.shape, generator=g) * 0.01)
                else:
                    a = 1.0 if e.name.startswith("fpn.") else 0.0
                    bound = math.sqrt(2.0 / (1 + a * a)) * math.sqrt(3.0 / fan_in)
                    v.copy_((torch.rand(e.shape, generator=g) * 2 - 1) * bound)
            elif e.name.endswith("bn.weight") or (e.name.startswith("head.") and e.name.endswith(".weight")):
                v.fill_(1.0)          # BN / GN gamma
            elif e.name == "head.cls_logits.bias":
                v.fill_(-math.log((1 - self.prior) / self.prior))
            elif e.name.startswith("backbone.output"):
                v.copy_(torch.randn(e.shape, generator=g) * 0.01 if len(e.shape) > 1 else torch.zeros(e.shape))
            else:
                v.zero_()
        self.invalidate()

    def invalidate(self):
        for _, bn in self.bns:
            bn.fold = None
        self._weights_dirty = True

    def refresh_derived_in_place(self, need_dgrad=False):
        """The master weights / buffers changed BEHIND recorded hipGraphs (a parameter broadcast after the step was
        captured: train_kd.py's graphs-first start of a data-parallel run): recompute everything the recorded kernels read
        that is derived from them INTO THE BUFFERS THEY WERE RECORDED WITH -- the bf16 shadow, the dgrad packing and the
        eval-mode BatchNorm folds.  (invalidate() would free the fold tensors whose addresses the graphs hold.)"""
        st = self.store
        for _, bn in self.bns:
            if bn.fold is not None:
                scale = st.storage(bn.gamma) * torch.rsqrt(st.storage(bn.rv) + 1e-5)
                bn.fold[0].copy_(scale)
                bn.fold[1].copy_(st.storage(bn.beta) - st.storage(bn.rm) * scale)
        self._weights_dirty = True
        self.prepare_weights(need_dgrad=need_dgrad and self.wt is not None)

    def to(self, device):
        self.store.to(device)
        self._bufs.clear()
        self.scratch_buf = None
        self.wt = None
        self.invalidate()
        return self

    @property
    def device(self):
        return self.store.params.device

    # ---- workspaces ----------------------------------------------------------------------
    def buf(self, name, shape, dtype=None):
        dtype = dtype or self.dtype
        key = (name, tuple(shape), dtype)
        b = self._bufs.get(key)
        if b is None:
            b = torch.empty(tuple(shape), dtype=dtype, device=self.device)
            self._bufs[key] = b
        return b

    SCRATCH_FLOATS = 1 << 22
    WORKSPACE_BYTES = 64 << 20      # split-K partial slabs (fp32) of the few-tile / long-K layers

    def fuse_norm_on(self):
        return False if self.fuse_norm is None else bool(self.fuse_norm)

    def next_side_stream(self):
        if self.side_streams:
            self._side_rr = (self._side_rr + 1) % len(self.side_streams)
            return self.side_streams[self._side_rr]
        return self.side_stream

    def flush_wgrad_group(self):
        """Launch the collected weight gradients (on a side stream when there is one: nothing in the sweep reads dW)."""
        grp = self.wgrad_group
        if grp is None or len(grp) == 0:
            return
        side = self.next_side_stream()
        if side is None:
            grp.launch()
            return
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            grp.launch()

    def workspace(self):
        ws = self._bufs.get("__workspace__")
        if ws is None:
            ws = torch.empty(self.WORKSPACE_BYTES // 4, dtype=torch.float32, device=self.device)
            self._bufs["__workspace__"] = ws
        return ws

    def scratch(self, name, n):
        """fp32 slice of the per-step scratch arena (zeroed once per step by the training forward)."""
        off = self._scratch_off.get((name, n))          # per (name, size): the same net may run several batch sizes
        if off is None:
            off = self._scratch_size
            self._scratch_off[(name, n)] = off
            self._scratch_size += (n + 7) // 8 * 8
            assert self._scratch_size <= self.SCRATCH_FLOATS, "scratch arena too small"
        if self.scratch_buf is None:
            self.scratch_buf = torch.zeros(self.SCRATCH_FLOATS, dtype=torch.float32, device=self.device)
        return self.scratch_buf[off:off + n]

    def prepare_weights(self, need_dgrad):
        """bf16 shadow + dgrad packing (one launch each) whenever the master weights changed."""
        st = self.store
        if self.dtype == torch.bfloat16 and (self._weights_dirty or st.shadow is None):
            st.refresh_shadow()
        if need_dgrad:
            if self.wt is None:
                off, blk, desc = 0, 0, []
                self._dgrad_layers = [c for c in self.convs if c.w.trainable]
                for c in self._dgrad_layers:
                    c.wt_off = off
                    desc += [st.base(c.w), off, c.cout_p, c.cin_p, c.k, blk]
                    off += c.w.numel
                    blk += (c.w.numel + 2047) // 2048
                self.wt = torch.empty(off, dtype=self.dtype, device=self.device)
                self._dgrad_desc = torch.tensor(desc, dtype=torch.int32, device=self.device)
                self._dgrad_blocks = blk
            src = st.params if self.dtype == torch.float32 else st.shadow
            ops.pack_dgrad_weights(src, self.wt, self._dgrad_desc, len(self._dgrad_layers), self._dgrad_blocks)
        self._weights_dirty = False

    # ---- forward ---------------------------------------------------------------------------
    def _cut(self):
        if self.cut_hook is not None and not self.training:
            self.cut_hook()

    def _backbone53(self, x, B, lv):
        x, lv = self.init_block.fwd_eval(x, B, lv) if not self.training else self.init_block.fwd_train(x, B, lv, self.tape)[:2]
        self._cut()
        feats = []
        for units in self.stages:
            for u in units:
                if u[0] == "down":
                    x, lv = u[1].fwd_eval(x, B, lv) if not self.training else u[1].fwd_train(x, B, lv, self.tape)[:2]
                else:
                    if self.training:
                        raise NotImplementedError("darknet53 is only run as the frozen teacher (eval mode) "
                                                  "in the KD step; training it is outside the hot path")
                    h, _ = u[1].fwd_eval(x, B, lv)
                    x, lv = u[2].fwd_eval(h, B, lv, residual=x)
                self._cut()
            feats.append((x, lv))
        return feats

    def _backbone_tiny(self, x, B, lv):
        feats = []
        n = len(self.stages)
        for i, units in enumerate(self.stages):
            if self.training and self.fuse_pool:
                pending = None
                for j, u in enumerate(units):
                    last = j == len(units) - 1
                    # inside a stage the next block applies this block's BatchNorm + LeakyReLU while loading its input
                    # (one launch per block instead of conv [+ statistics] + normalise): stages 3-5; the first two
                    # stages are single blocks
                    nxt_pointwise = (not last) and units[j + 1].conv.k == 1
                    x, lv, pending = u.fwd_train(x, B, lv, self.tape, pool=(i != n - 1 and last), pending=pending,
                                                 defer=self.bn_on_load == 2 and not last or (self.bn_on_load == 1 and nxt_pointwise))
                if i != n - 1:
                    self.tape.append(("pooled", i))
                feats.append((x, lv))
                continue
            for u in units:
                x, lv = u.fwd_train(x, B, lv, self.tape)[:2] if self.training else u.fwd_eval(x, B, lv)
            if i != n - 1:
                (h, w) = lv[0]
                y = self.buf("pool%d" % i, (B * (h // 2) * (w // 2), x.shape[1]))
                ops.maxpool2_fwd(x, y, B, h, w)
                if self.training:
                    self.tape.append(("pool", x, B, h, w))
                x, lv = y, [(h // 2, w // 2)]
            feats.append((x, lv))
        # darknet.py:125-135: out4 = stage5(stage4(out3))
        return [feats[0], feats[1], feats[2], feats[4]]

    def scratch_region(self):
        """The part of the statistics arena a step has to zero (None until a first forward has sized it)."""
        if self.scratch_buf is None or self._scratch_size == 0:
            return None
        return self.scratch_buf[:max(self._scratch_size, 8)]

    def forward(self, images, scratch_zeroed=False):
        """images (B,3,H,W) fp32 NCHW on device -> packed logits cls (rows,16) fp32, reg (rows,240) fp32.
        scratch_zeroed: the caller's step prologue (ops.zero_many) already cleared scratch_region()."""
        B, _, H, W = images.shape
        self.prepare_weights(need_dgrad=self.training)
        self.tape = []
        if self.scratch_buf is not None and not scratch_zeroed:
            # one memset per step: every atomically accumulated statistic lives here (only the part in use)
            self.scratch_buf[:max(self._scratch_size, 8)].zero_()
        if self.nhwc_in is not None:        # GraphedKDStep: the frozen teacher converted this batch one replay earlier
            x = self.nhwc_in
            assert x.shape == (B * H * W, 8) and x.dtype == self.dtype
        else:
            out = self.nhwc_out if self.nhwc_out is not None else self.buf("input", (B * H * W, 8))
            assert out.shape == (B * H * W, 8) and out.dtype == self.dtype
            x = ops.image_to_nhwc(images.contiguous(), self.dtype, 8, out=out)
        feats = self._backbone53(x, B, [(H, W)]) if self.arch == "darknet53" else self._backbone_tiny(x, B, [(H, W)])
        ops.mark("%s.fwd.backbone.end" % ("teacher" if not self.training else "student"))
        oc = self.out_channel
        idxs = sorted(self.inner.keys())
        top = idxs[-1]
        levels = [feats[i][1][0] for i in idxs]
        h6 = ((levels[-1][0] + 1) // 2, (levels[-1][1] + 1) // 2)
        h7 = ((h6[0] + 1) // 2, (h6[1] + 1) // 2)
        levels_all = levels + [h6, h7]
        row0, r = [], 0
        for (h, w) in levels_all:
            row0.append(r)
            r += B * h * w
        head_in = self.buf("head_in", (r, oc))
        self.levels, self.rows, self.level_row0, self.batch = levels_all, r, row0, B
        self.in_hw = (H, W)

        def slot(li):
            h, w = levels_all[li]
            return head_in[row0[li]:row0[li] + B * h * w]

        inner_prev = None
        self.fpn_ctx = {}
        for pos in range(len(idxs) - 1, -1, -1):
            i = idxs[pos]
            f, lv = feats[i]
            h, w = lv[0]
            if inner_prev is None:
                inner, _ = self.inner[i].fwd(f, B, lv, out=self.buf("inner%d" % i, (B * h * w, oc)))
            else:
                lat, _ = self.inner[i].fwd(f, B, lv, out=self.buf("lat%d" % i, (B * h * w, oc)))
                inner = ops.upsample2_add(lat, inner_prev, self.buf("inner%d" % i, (B * h * w, oc)), B, h, w)
            self.outc[i].fwd(inner, B, lv, out=slot(pos))
            self.fpn_ctx[i] = (f, lv, inner)
            inner_prev = inner
            self._cut()
        ftop, lvtop = feats[top]
        p6, _ = self.p6.fwd(ftop, B, lvtop, out=slot(len(idxs)))
        p6r = ops.eltwise(ops.ELT_RELU, p6, None, self.buf("p6_relu", p6.shape))
        self.p7.fwd(p6r, B, [h6], out=slot(len(idxs) + 1))
        self.p6_ctx = (ftop, lvtop, p6, p6r, h6)
        ops.mark("%s.fwd.fpn.end" % ("teacher" if not self.training else "student"))
        # ---- head over all levels at once ----
        self.head_ctx = {}
        outs = {}
        towers = (("cls", self.cls_tower, self.cls_logits), ("pose", self.pose_tower, self.pose_pred))
        if self.training and self.pair_towers:
            # layer by layer through BOTH towers: the two convolutions of a layer have identical geometry and go
            # out as one launch (ops.conv_pair: 228 tiles of 192x128 instead of 2 x 170 tiles of 128x128)
            xs = {t: head_in for t, _, _ in towers}
            saved = {t: [] for t, _, _ in towers}
            for li in range(len(self.cls_tower)):
                raws = {}
                fuse = self.fuse_norm_on() and all(tower[li][0].norm_fusable(B, levels_all, ops.NORM_GROUP, tower[li][1].groups)
                                              for _, tower, _ in towers)
                with ops.conv_pair():
                    for tname, tower, _ in towers:
                        conv, gn = tower[li]
                        raws[tname] = self.buf("%s.raw%d" % (tname, li), (r, oc), torch.float32)
                        if fuse:
                            gn.fwd_fused(conv, xs[tname], B, levels_all, self.buf("%s.act%d" % (tname, li), (r, oc)),
                                         raw_out=raws[tname])
                        else:
                            conv.fwd(xs[tname], B, levels_all, out_f32=True, out=raws[tname],
                                     stats=gn.stats(B, levels_all), stats_groups=gn.groups)
                for tname, tower, _ in towers:
                    gn = tower[li][1]
                    y = self.buf("%s.act%d" % (tname, li), (r, oc))
                    if not fuse:
                        gn.fwd(raws[tname], B, levels_all, out=y)
                    saved[tname].append((xs[tname], raws[tname]))
                    xs[tname] = y
            for tname, tower, final in towers:
                seg = self.store.storage(self.scales) if tname == "pose" else None
                out, _ = final.fwd(xs[tname], B, levels_all, seg_scale=seg, out_f32=True,
                                   out=self.buf("%s.logits" % tname, (r, final.cout_p), torch.float32))
                self.head_ctx[tname] = (saved[tname], xs[tname])
                outs[tname] = out
            return outs["cls"], outs["pose"]
        for tname, tower, final in towers:
            x = head_in
            saved = []
            for li, (conv, gn) in enumerate(tower):
                y = self.buf("%s.act%d" % (tname, li), (r, oc))
                if self.fuse_norm_on() and conv.norm_fusable(B, levels_all, ops.NORM_GROUP, gn.groups):
                    # eval mode (the frozen teacher): the fp32 pre-normalisation tensor is not even stored
                    raw = self.buf("%s.raw%d" % (tname, li), (r, oc), torch.float32) if self.training else None
                    gn.fwd_fused(conv, x, B, levels_all, y, raw_out=raw)
                else:
                    raw, _ = conv.fwd(x, B, levels_all, out_f32=True,
                                      out=self.buf("%s.raw%d" % (tname, li), (r, oc), torch.float32),
                                      stats=gn.stats(B, levels_all), stats_groups=gn.groups)
                    gn.fwd(raw, B, levels_all, out=y)
                saved.append((x, raw))
                x = y
                self._cut()
            seg = self.store.storage(self.scales) if tname == "pose" else None
            out, _ = final.fwd(x, B, levels_all, seg_scale=seg, out_f32=True,
                               out=self.buf("%s.logits" % tname, (r, final.cout_p), torch.float32))
            self.head_ctx[tname] = (saved, x)
            outs[tname] = out
        return outs["cls"], outs["pose"]

    # ---- backward (student only) -------------------------------------------------------------
    def backward(self, dcls, dreg):
        """dcls (rows,16), dreg (rows,240) in self.dtype: gradients w.r.t. the UNSCALED head outputs
        (the Scale module's factor is already folded into dreg by kd6d_loss_backward)."""
        assert self.training and self.arch != "darknet53"
        B, lv_all, r, oc = self.batch, self.levels, self.rows, self.out_channel
        if self.use_wgrad_group and self.dtype == torch.bfloat16 and self.wgrad_group is None:
            self.wgrad_group = ops.WgradGroup(self.wgrad_group_wgs or max(ops.device_cu_count() // 2, 1))
        self.grouping = self.wgrad_group is not None
        d_head_in = self.buf("d_head_in", (r, oc))
        first = True
        towers = (("cls", self.cls_tower, self.cls_logits, dcls), ("pose", self.pose_tower, self.pose_pred, dreg))
        if self.pair_towers:
            dxs = {}
            for tname, tower, final, dlog in towers:
                saved, last = self.head_ctx[tname]
                dxs[tname] = final.bwd(last, dlog, B, lv_all, dx=self.buf("%s.dact" % tname, (r, oc)))
            for li in range(len(self.cls_tower) - 1, -1, -1):
                # the two GroupNorm backwards of the layer as one launch (170 four-wave workgroups each: alone they
                # are latency-bound and, back to back on one stream, the second waited for the first)
                draws, items = {}, []
                for tname, tower, _, _ in towers:
                    gn = tower[li][1]
                    draws[tname] = self.buf("%s.draw%d" % (tname, li), (r, oc))
                    items.append(gn.bwd_item(self.head_ctx[tname][0][li][1], dxs[tname], B, lv_all, draws[tname]))
                ops.gn_relu_bwd_pair(items, [h * w for (h, w) in lv_all], B, self.cls_tower[li][1].groups,
                                     self.store.acc_stride, flags=ops.GN_WS_ZEROED)
                if li > 0:
                    # the two data gradients as one launch; the weight gradients fork onto the side streams as usual
                    with ops.conv_pair(enabled=self.pair_towers == 1):
                        for tname, tower, _, _ in towers:
                            conv = tower[li][0]
                            dxs[tname] = conv.bwd(self.head_ctx[tname][0][li][0], draws[tname], B, lv_all,
                                                  dx=self.buf("%s.dact" % tname, (r, oc)))
                else:   # both towers add into d_head_in: one after the other
                    for k, (tname, tower, _, _) in enumerate(towers):
                        tower[0][0].bwd(self.head_ctx[tname][0][0][0], draws[tname], B, lv_all, dx=d_head_in,
                                        accumulate=k > 0)
            towers = ()
        for tname, tower, final, dlog in towers:
            saved, last = self.head_ctx[tname]
            dx = final.bwd(last, dlog, B, lv_all, dx=self.buf("%s.dact" % tname, (r, oc)))
            for li in range(len(tower) - 1, -1, -1):
                conv, gn = tower[li]
                x_in, raw = saved[li]
                draw = gn.bwd(raw, dx, B, lv_all, self.buf("%s.draw%d" % (tname, li), (r, oc)))
                if li > 0:
                    dx = conv.bwd(x_in, draw, B, lv_all, dx=self.buf("%s.dact" % tname, (r, oc)))
                else:
                    conv.bwd(x_in, draw, B, lv_all, dx=d_head_in, accumulate=not first)
            first = False
        row0 = self.level_row0
        ops.mark("student.bwd.head.end")

        def dslot(li):
            h, w = lv_all[li]
            return d_head_in[row0[li]:row0[li] + B * h * w]

        idxs = sorted(self.inner.keys())
        n_l = len(idxs)
        # d_head_in is complete: the FPN output convolutions' weight gradients (x = the top-down sums kept by the
        # forward, dy = the level slices of d_head_in) join the head's group, which then goes out as one launch pair
        grouped_out = set()
        if self.wgrad_group is not None:
            for pos in range(n_l):
                i = idxs[pos]
                f, lv, inner = self.fpn_ctx[i]
                conv = self.outc[i]
                g = conv.geom(B, lv)
                if ops.wgrad_group_supported(g, inner.dtype):
                    self.wgrad_group.add(g, inner, dslot(pos), self.store.storage(conv.w, "grads"),
                                         self.store.storage(conv.b, "grads"), flops=conv.flops(g))
                    grouped_out.add(i)
            # launched here, beside the FPN / backbone sweep, when a teacher forward shares the device (pipelined
            # steps: 5363-5382 images/s against 5107-5135 with the launch behind the FPN sweep); strictly sequential
            # steps launch it behind the FPN sweep, beside the backbone sweep's chain of small launches (4836 against
            # 4655).  At the very end of the sweep it costs 2-7 % of the step (it then runs with nothing beside it).
            if self.wgrad_group_flush != "fpn_end":
                self.flush_wgrad_group()
        self.grouping = False
        # P7 = conv(relu(P6)); P6 = conv(top feature)
        ftop, lvtop, p6, p6r, h6 = self.p6_ctx
        d_p6r = self.p7.bwd(p6r, dslot(n_l + 1), B, [h6], dx=self.buf("d_p6r", p6.shape))
        d_p6 = ops.eltwise(ops.ELT_RELU_BWD, p6, d_p6r, self.buf("d_p6a", p6.shape))
        d_p6 = ops.eltwise(ops.ELT_ADD, d_p6, dslot(n_l), self.buf("d_p6", p6.shape))
        dfeat = {}
        top = idxs[-1]
        dfeat[top] = self.p6.bwd(ftop, d_p6, B, lvtop, dx=self.buf("dfeat%d" % top, ftop.shape))
        d_inner_up = None
        for pos in range(n_l):
            i = idxs[pos]
            f, lv, inner = self.fpn_ctx[i]
            h, w = lv[0]
            d_inner = self.buf("d_inner%d" % i, inner.shape)
            if d_inner_up is not None:
                ops.sumpool2(d_inner_up[0], d_inner, B, d_inner_up[1], d_inner_up[2])
                self.outc[i].bwd(inner, dslot(pos), B, lv, dx=d_inner, accumulate=True, need_dw=i not in grouped_out)
            else:
                self.outc[i].bwd(inner, dslot(pos), B, lv, dx=d_inner, need_dw=i not in grouped_out)
            if i == top:
                self.inner[i].bwd(f, d_inner, B, lv, dx=dfeat[top], accumulate=True)
            else:
                dfeat[i] = self.inner[i].bwd(f, d_inner, B, lv, dx=self.buf("dfeat%d" % i, f.shape))
            d_inner_up = (d_inner, h, w)
        ops.mark("student.bwd.fpn.end")
        if self.wgrad_group is not None and self.wgrad_group_flush == "fpn_end":
            self.flush_wgrad_group()
        if self.grad_hook is not None:
            # every gradient outside the backbone has been issued (on this stream or on the weight-gradient streams):
            # a data-parallel caller starts their exchange here, beside the backbone sweep
            self.grad_hook()
        # ---- backbone (tiny): walk the tape in reverse ----
        # feats index -> gradient arriving from the FPN; out4 = index 3 (after stage 5), out3 = index 2
        grad = dfeat[top]
        pending = {2: dfeat.get(2)}            # added when the sweep reaches out3 (after pool3)
        stage_of_pool = {}
        i_rec = len(self.tape) - 1
        pools_seen = 0
        n_pools = sum(1 for t in self.tape if t[0] == "pool")
        while i_rec >= 0:
            rec = self.tape[i_rec]
            if rec[0] == "pooled":              # the pool lives inside the block's BN kernels; out3 = after pool 2
                if rec[1] == 2 and pending[2] is not None:
                    grad = ops.eltwise(ops.ELT_ADD, grad, pending[2], self.buf("d_out3", grad.shape))
            elif rec[0] == "pool":
                _, x, b_, h, w = rec
                pool_idx = n_pools - 1 - pools_seen    # 3,2,1,0
                pools_seen += 1
                if pool_idx == 2 and pending[2] is not None:
                    grad = ops.eltwise(ops.ELT_ADD, grad, pending[2], self.buf("d_out3", grad.shape))
                dx = self.buf("dpool%d" % pool_idx, x.shape)
                grad = ops.maxpool2_bwd(x, grad, dx, b_, h, w)
            else:
                blk = rec[0]
                need_dx = i_rec > 0
                grad = blk.bwd(rec, grad, need_dx=need_dx,
                               dx=self.buf(blk.name + ".dx", rec[1].shape) if need_dx else None)
            i_rec -= 1
        ops.mark("student.bwd.main.end")
        for side in (self.side_streams or ([self.side_stream] if self.side_stream is not None else [])):
            torch.cuda.current_stream().wait_stream(side)
        # every accumulated gradient (per-layer weight / bias gradients, GroupNorm gains and shifts, the head's scales)
        # -> fp32, accumulators cleared for the next step: one launch.  A caller that resolved the FPN + head part
        # early (grad_hook, the overlapped exchange) sets resolve_hi to where that part begins.
        self.store.resolve_grads(0, self.resolve_hi)
        return None
