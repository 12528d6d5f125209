"""Thin Python wrappers over the C ABI: tensors in, tensors out, no arithmetic here.

All activations are "packed NHWC": a 2-D tensor (rows, C) whose rows are the pixels of
one or several pyramid levels laid out level-major, then image, then row-major (y, x).
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, GN_STATS_READY, GN_WS_ZEROED, ConvGeom, check, lib  # noqa: F401


def dt_code(dtype):
    if dtype == torch.bfloat16:
        return _lib.KD6D_BF16
    if dtype == torch.float32:
        return _lib.KD6D_F32
    raise TypeError("kd6d supports bfloat16 and float32 activations, got %s" % dtype)


def granule(dtype):
    return 8 if dtype == torch.bfloat16 else 4


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "kd6d ops need contiguous device tensors"
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- reproducible reductions (include/kd6d.h): statistics and workspaces are kd6d_acc, 16 bytes each ----------
ACC_FLOATS = 4                       # fp32 slots one accumulator takes in an fp32 arena
ACC_ACT, ACC_GRAD = _lib.ACC_ACT, _lib.ACC_GRAD


def acc_zeros(n, device):
    """n zeroed accumulators as an fp32 tensor (the engine keeps them inside its per-step zeroed fp32 arena)."""
    return torch.zeros(n * ACC_FLOATS, dtype=torch.float32, device=device)


def acc_read(acc, n, kind, out=None, accumulate=False, clear=False):
    """fp32 values of n interleaved accumulators (kd6d_acc_read): tests, debugging, stand-alone callers."""
    assert acc.is_contiguous() and acc.numel() * acc.element_size() >= n * 16
    if out is None:
        assert not accumulate
        out = torch.empty(n, dtype=torch.float32, device=acc.device)
    check(lib.kd6d_acc_read(_ptr(acc), n, kind, _ptr(out), int(accumulate), int(clear), _stream()), "kd6d_acc_read")
    return out


def planar_acc(n, device):
    """A stand-alone PLANAR gradient accumulator array for n elements: int64 (2 * n), lo plane then hi plane; pass
    acc[:n] with stride n to conv2d_wgrad / gn_relu_bwd / loss_backward."""
    return torch.zeros(2 * n, dtype=torch.int64, device=device)


def planar_acc_value(acc):
    """fp32 value of a stand-alone planar accumulator array made by planar_acc (torch arithmetic: test helper; the
    engine resolves its gradient bucket with kd6d_grad_acc_resolve)."""
    n = acc.numel() // 2
    lo, hi = acc[:n].double(), acc[n:].double()
    return (hi * 2.0 ** (47 - ACC_GRAD) + lo * 2.0 ** (-ACC_GRAD)).float()


# ---- optional per-launch timing (bench.py roofline leg): HIP events on the launch stream ----------
_prof = None


def profile_begin():
    global _prof
    _prof = []


def profile_end():
    """-> list of (kind, algorithmic_flops, milliseconds, tag) for every bracketed launch."""
    global _prof
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    return [(k, f, e0.elapsed_time(e1), t) for (k, f, e0, e1, t) in rec]


_pair_deferred = None      # inside a conv_pair bracket while profiling: (kind, flops, tag) of the recorded launches


class _Timed:
    __slots__ = ("kind", "flops", "e0", "tag", "n0")

    def __init__(self, kind, flops, tag=None):
        self.kind, self.flops, self.e0, self.tag, self.n0 = kind, flops, None, tag, 0

    def __enter__(self):
        if _prof is not None:
            if _pair_deferred is not None:
                self.n0 = lib.kd6d_conv2d_pair_pending()
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.e0 is not None:
            if _pair_deferred is not None and lib.kd6d_conv2d_pair_pending() > self.n0:
                _pair_deferred.append((self.kind, self.flops, self.tag))     # launched (and timed) when the bracket closes
                return
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _prof.append((self.kind, self.flops, self.e0, e1, self.tag))


class Geom:
    """Forward-sense geometry of one conv layer over 1..5 pyramid levels."""

    def __init__(self, batch, cin, cout, ksize, stride, pad, levels):
        assert 1 <= len(levels) <= _lib.MAX_SEG
        self.batch, self.cin, self.cout = batch, cin, cout
        self.ksize, self.stride, self.pad = ksize, stride, pad
        self.levels_in = [tuple(l) for l in levels]
        self.levels_out = [((h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1)
                           for (h, w) in self.levels_in]
        g = ConvGeom()
        g.nseg, g.batch, g.cin, g.cout = len(levels), batch, cin, cout
        g.ksize, g.stride, g.pad = ksize, stride, pad
        rin = rout = 0
        for s, ((h, w), (ho, wo)) in enumerate(zip(self.levels_in, self.levels_out)):
            g.seg[s].in_h, g.seg[s].in_w, g.seg[s].out_h, g.seg[s].out_w = h, w, ho, wo
            g.seg[s].in_row0, g.seg[s].out_row0 = rin, rout
            rin += batch * h * w
            rout += batch * ho * wo
        self.rows_in, self.rows_out = rin, rout
        self.c = g

    @property
    def ref(self):
        return ctypes.byref(self.c)


def conv2d_fwd(geom, x, w, out=None, ch_scale=None, ch_shift=None, act=ACT_NONE, residual=None,
               seg_scale=None, out_f32=False, flops=0, stats=None, stats_groups=0, workspace=None):
    assert x.shape == (geom.rows_in, geom.cin), (x.shape, geom.rows_in, geom.cin)
    assert w.dtype == x.dtype and w.numel() == geom.cout * geom.ksize * geom.ksize * geom.cin
    odt = torch.float32 if out_f32 else x.dtype
    if out is None:
        out = torch.empty((geom.rows_out, geom.cout), dtype=odt, device=x.device)
    assert out.shape == (geom.rows_out, geom.cout) and out.dtype == odt
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == odt
    for v in (ch_scale, ch_shift):
        assert v is None or (v.dtype == torch.float32 and v.numel() >= geom.cout)
    assert seg_scale is None or (seg_scale.dtype == torch.float32 and seg_scale.numel() >= len(geom.levels_in))
    with _Timed("conv_fwd", flops, geom):
        check(lib.kd6d_conv2d_fwd(geom.ref, dt_code(x.dtype), _ptr(x), _ptr(w), _ptr(out), _ptr(ch_scale),
                                  _ptr(ch_shift), act, _ptr(residual), _ptr(seg_scale), int(out_f32),
                                  _ptr(stats), int(stats_groups), _ptr(workspace),
                                  0 if workspace is None else workspace.numel() * workspace.element_size(),
                                  _stream()), "kd6d_conv2d_fwd")
    return out


NORM_GROUP, NORM_BATCH = _lib.NORM_GROUP, _lib.NORM_BATCH


def conv_norm_fusable(geom, dtype, kind, groups=0):
    """kd6d_conv2d_fwd_norm_fusable: does this layer take the conv + normalisation + activation launch?"""
    return bool(lib.kd6d_conv2d_fwd_norm_fusable(geom.ref, dt_code(dtype), int(kind), int(groups)))


def conv_norm_counter_words(geom, kind):
    """32-bit words of the pre-zeroed barrier counters of conv2d_fwd_norm (kd6d.h)."""
    return BARRIER_WORDS if kind == NORM_BATCH else len(geom.levels_in) * geom.batch * _lib.NORM_MAX_CTILES


def conv_norm_stats_floats(geom, kind, groups=0):
    """fp32 slots of the fused launch's statistics accumulators."""
    if kind == NORM_BATCH:
        return _lib.BN_FUSED_REPLICAS * 2 * geom.cout * ACC_FLOATS
    return len(geom.levels_in) * geom.batch * groups * 2 * ACC_FLOATS


def conv2d_fwd_norm(geom, x, w, y, kind, gamma, beta, stats, counters, act, raw_out=None, bias=None, groups=0, eps=1e-5,
                    momentum=0.1, running_mean=None, running_var=None, save_mean=None, save_invstd=None, flops=0):
    """conv (+ bias) -> GroupNorm / train-mode BatchNorm -> activation as ONE launch (kd6d_conv2d_fwd_norm).
    y: (rows_out, cout) in x.dtype; raw_out: optional fp32 pre-normalisation tensor for the backward pass;
    stats / counters: pre-zeroed (conv_norm_stats_floats / conv_norm_counter_words)."""
    assert x.shape == (geom.rows_in, geom.cin) and w.dtype == x.dtype
    assert y.shape == (geom.rows_out, geom.cout) and y.dtype == x.dtype
    assert raw_out is None or (raw_out.shape == y.shape and raw_out.dtype == torch.float32)
    assert stats.dtype == torch.float32 and stats.numel() >= conv_norm_stats_floats(geom, kind, groups)   # accumulators
    assert counters.numel() >= conv_norm_counter_words(geom, kind) and counters.element_size() == 4
    n = _lib.ConvNorm()
    n.kind, n.groups, n.act, n.eps, n.momentum = int(kind), int(groups), int(act), float(eps), float(momentum)
    for name, t in (("gamma", gamma), ("beta", beta), ("y", y), ("stats", stats), ("counters", counters),
                    ("running_mean", running_mean), ("running_var", running_var), ("save_mean", save_mean),
                    ("save_invstd", save_invstd)):
        assert t is None or (t.is_cuda and t.is_contiguous())
        setattr(n, name, t.data_ptr() if t is not None else None)
    with _Timed("conv_fwd", flops, geom):
        check(lib.kd6d_conv2d_fwd_norm(geom.ref, dt_code(x.dtype), _ptr(x), _ptr(w), _ptr(raw_out), _ptr(bias),
                                       ctypes.byref(n), _stream()), "kd6d_conv2d_fwd_norm")
    return y


BN_REPLICAS = _lib.BN_FUSED_REPLICAS


def conv2d_fwd_block(geom, x, w, y_raw, stats=None, stats_replicas=1, bn_in=None, z_out=None, flops=0):
    """kd6d_conv2d_fwd_block: the convolution of a train-mode ConvBlock -> fp32 y_raw + its batch sums (stats:
    stats_replicas rows of {sum, sumsq}, pre-zeroed).  bn_in: dict(sums, replicas, gamma, beta, act, eps, momentum,
    running_mean, running_var, save_mean, save_invstd) -- then x is the PREVIOUS block's fp32 conv output, normalised and
    activated while loaded, and z_out receives that activation (what this layer's weight gradient reads)."""
    assert y_raw.shape == (geom.rows_out, geom.cout) and y_raw.dtype == torch.float32
    assert stats is None or (stats.dtype == torch.float32 and stats.numel() >= stats_replicas * 2 * geom.cout * ACC_FLOATS)
    bn = None
    if bn_in is not None:
        assert x.shape == (geom.rows_in, geom.cin) and x.dtype == torch.float32
        assert z_out is None or (z_out.shape == x.shape and z_out.dtype == w.dtype)
        bn = _lib.BnIn()
        bn.replicas, bn.act = int(bn_in["replicas"]), int(bn_in["act"])
        bn.eps, bn.momentum = float(bn_in.get("eps", 1e-5)), float(bn_in.get("momentum", 0.1))
        assert bn_in["sums"].numel() >= bn.replicas * 2 * geom.cin * ACC_FLOATS
        for name in ("sums", "gamma", "beta", "running_mean", "running_var", "save_mean", "save_invstd"):
            t = bn_in.get(name)
            assert t is None or (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32)
            setattr(bn, name, t.data_ptr() if t is not None else None)
    else:
        assert x.shape == (geom.rows_in, geom.cin) and x.dtype == w.dtype and z_out is None
    with _Timed("conv_fwd", flops, geom):
        check(lib.kd6d_conv2d_fwd_block(geom.ref, dt_code(w.dtype), _ptr(x), ctypes.byref(bn) if bn is not None else None,
                                        _ptr(z_out), _ptr(w), _ptr(y_raw), _ptr(stats), int(stats_replicas), _stream()),
              "kd6d_conv2d_fwd_block")
    return y_raw


def conv2d_dgrad(geom, dy, wt, dx=None, accumulate=False, flops=0):
    assert dy.shape == (geom.rows_out, geom.cout), (dy.shape, geom.rows_out, geom.cout)
    assert wt.dtype == dy.dtype and wt.numel() == geom.cout * geom.ksize * geom.ksize * geom.cin
    if dx is None:
        assert not accumulate
        dx = torch.empty((geom.rows_in, geom.cin), dtype=dy.dtype, device=dy.device)
    assert dx.shape == (geom.rows_in, geom.cin) and dx.dtype == dy.dtype
    with _Timed("conv_dgrad", flops, geom):
        check(lib.kd6d_conv2d_dgrad(geom.ref, dt_code(dy.dtype), _ptr(dy), _ptr(wt), _ptr(dx),
                                    int(accumulate), _stream()), "kd6d_conv2d_dgrad")
    return dx


def conv2d_wgrad_parts(geom, dtype, with_bias=False, cu_budget=0):
    """Number of partial dW images kd6d_conv2d_wgrad writes for this geometry / budget (kd6d_conv2d_wgrad_parts)."""
    n = int(lib.kd6d_conv2d_wgrad_parts(geom.ref, dt_code(dtype), int(with_bias), int(cu_budget)))
    if n < 1:
        check(n, "kd6d_conv2d_wgrad_parts")
    return n


def conv2d_wgrad(geom, x, dy, dw_slab, flops=0, dbias=None, acc_stride=0, cu_budget=0):
    """dw_slab: fp32 (parts, cout*k*k*cin) -- every pixel split stores its partial dW image (conv2d_wgrad_parts of them);
    the caller adds them in order (ParamStore.resolve_grads / kd6d_grad_acc_resolve).  dbias: lo-plane view of PLANAR
    gradient accumulators (int64; the hi word of element i sits acc_stride words behind its lo word)."""
    assert x.shape == (geom.rows_in, geom.cin) and dy.shape == (geom.rows_out, geom.cout)
    assert x.dtype == dy.dtype and dw_slab.dtype == torch.float32 and dw_slab.is_contiguous()
    assert dw_slab.numel() % (geom.cout * geom.ksize * geom.ksize * geom.cin) == 0
    with _Timed("conv_wgrad", flops, geom):
        assert dbias is None or (dbias.dtype == torch.int64 and dbias.numel() >= geom.cout and acc_stride > 0)
        check(lib.kd6d_conv2d_wgrad(geom.ref, dt_code(x.dtype), _ptr(x), _ptr(dy), _ptr(dw_slab), dw_slab.numel(),
                                    _ptr(dbias), int(acc_stride), int(cu_budget), _stream()), "kd6d_conv2d_wgrad")
    return dw_slab


def conv2d_wgrad_f32(geom, x, dy, with_bias=False, cu_budget=0):
    """Stand-alone weight gradient -> fp32 tensors (dw, dbias | None): slab and bias accumulators made, filled and
    reduced here (torch arithmetic on the device: tests and tools; the engine resolves with kd6d_grad_acc_resolve)."""
    nw = geom.cout * geom.ksize * geom.ksize * geom.cin
    parts = conv2d_wgrad_parts(geom, x.dtype, with_bias, cu_budget)
    slab = torch.empty(parts, nw, dtype=torch.float32, device=x.device)
    acc = planar_acc(geom.cout, x.device) if with_bias else None
    conv2d_wgrad(geom, x, dy, slab, dbias=acc[:geom.cout] if with_bias else None, acc_stride=geom.cout,
                 cu_budget=cu_budget)
    dw = slab[0].clone()
    for k in range(1, parts):              # the order kd6d_grad_acc_resolve adds them in
        dw += slab[k]
    return dw, (planar_acc_value(acc) if with_bias else None)


def set_option(name, value):
    """kd6d_set_option (include/kd6d.h): pin a kernel family for the calls that follow; tests and benches only."""
    check(lib.kd6d_set_option(name.encode(), int(value)))


def get_option(name):
    v = ctypes.c_longlong(0)
    check(lib.kd6d_get_option(name.encode(), ctypes.byref(v)))
    return int(v.value)


class option:
    """`with ops.option("conv.smallc", 1): ...` -- the option is put back on exit."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def zero_many(tensors, counter=None):
    """kd6d_zero_regions: zero every tensor of `tensors` (contiguous device tensors, None entries skipped) and add 1
    to each element of the int64 tensor `counter`, as one launch on the current stream."""
    lst = _lib.ZeroList()
    n = 0
    for t in tensors:
        if t is None or t.numel() == 0:
            continue
        assert t.is_contiguous() and n < _lib.MAX_ZERO
        lst.ptr[n] = t.data_ptr()
        lst.bytes[n] = t.numel() * t.element_size()
        n += 1
    lst.n = n
    if counter is not None:
        assert counter.dtype == torch.int64 and counter.is_contiguous()
    check(lib.kd6d_zero_regions(ctypes.byref(lst), _ptr(counter) if counter is not None else None,
                                counter.numel() if counter is not None else 0, _stream()), "kd6d_zero_regions")


def uniform_keys(out, counter, seed):
    """kd6d_uniform_keys: out (fp32, device) <- uniform [0, 1) keys of (seed, counter[0], index)."""
    assert out.dtype == torch.float32 and counter.dtype == torch.int64
    check(lib.kd6d_uniform_keys(_ptr(out), out.numel(), _ptr(counter), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream()),
          "kd6d_uniform_keys")
    return out


def device_cu_count():
    n = lib.kd6d_device_cu_count()
    if n <= 0:
        raise RuntimeError("kd6d_device_cu_count failed: %s" % lib.kd6d_last_error().decode())
    return n


_marks = None          # (int64 tensor, {name: slot}) while a timeline is being recorded (tools/step_timeline.py)


def mark(name):
    """Device timestamp into the slot registered for `name`; no-op unless marks_begin() was called."""
    if _marks is None:
        return
    buf, slots = _marks
    idx = slots.setdefault(name, len(slots))
    assert idx < buf.numel(), "too many marks"
    check(lib.kd6d_mark(ctypes.c_void_p(buf.data_ptr() + 8 * idx), _stream()), "kd6d_mark")


def marks_begin(device, n=64):
    global _marks
    _marks = (torch.zeros(n, dtype=torch.int64, device=device), {})
    return _marks


def marks_end():
    global _marks
    m, _marks = _marks, None
    return m


def pack_dgrad_weights(w_base, wt_base, desc_dev, n_layers, total_blocks):
    check(lib.kd6d_pack_dgrad_weights(dt_code(w_base.dtype), _ptr(w_base), _ptr(wt_base), _ptr(desc_dev),
                                      n_layers, total_blocks, _stream()), "kd6d_pack_dgrad_weights")


def colstats(x, sum_, sumsq=None):
    """sum_ / sumsq: C accumulators each (acc_zeros(C) or slices of the zeroed arena), class ACC_ACT."""
    rows, c = x.shape
    assert sum_.numel() >= c * ACC_FLOATS and (sumsq is None or sumsq.numel() >= c * ACC_FLOATS)
    check(lib.kd6d_colstats(dt_code(x.dtype), _ptr(x), rows, c, _ptr(sum_), _ptr(sumsq), _stream()),
          "kd6d_colstats")


def _xf32(x, act_dtype):
    """1 when the pre-normalisation tensor is fp32 but activations are bf16."""
    return int(x.dtype == torch.float32 and act_dtype == torch.bfloat16)


def bn_train_fwd(x, y, sum_, sumsq, gamma, beta, eps, momentum, running_mean, running_var,
                 save_mean, save_invstd, act):
    rows, c = x.shape
    check(lib.kd6d_bn_train_fwd(dt_code(y.dtype), _xf32(x, y.dtype), _ptr(x), _ptr(y), rows, c, _ptr(sum_), _ptr(sumsq),
                                _ptr(gamma), _ptr(beta), eps, momentum, _ptr(running_mean),
                                _ptr(running_var), _ptr(save_mean), _ptr(save_invstd), act, _stream()),
          "kd6d_bn_train_fwd")
    return y


BARRIER_WORDS = 32          # KD6D_BARRIER_WORDS: pre-zeroed 32-bit words of an in-kernel barrier (kd6d.h)


def bn_train_bwd(x, dz, dx, mean, invstd, gamma, beta, act, ws_sum_dy, ws_sum_dy_xhat, dgamma, dbeta, replicas=1,
                 counter=None):
    """ws_sum_dy / ws_sum_dy_xhat: `replicas` rows of C accumulators each, pre-zeroed (kd6d.h).  counter: BARRIER_WORDS
    pre-zeroed 32-bit words -> the one-launch backward (in-kernel barrier) when the tensor fits; None -> reduce + apply."""
    rows, c = x.shape
    assert ws_sum_dy.numel() >= replicas * c * ACC_FLOATS and ws_sum_dy_xhat.numel() >= replicas * c * ACC_FLOATS
    assert counter is None or counter.numel() >= BARRIER_WORDS
    check(lib.kd6d_bn_train_bwd(dt_code(dz.dtype), _xf32(x, dz.dtype), _ptr(x), _ptr(dz), _ptr(dx), rows, c, _ptr(mean),
                                _ptr(invstd), _ptr(gamma), _ptr(beta), act, _ptr(ws_sum_dy), _ptr(ws_sum_dy_xhat),
                                _ptr(counter), _ptr(dgamma), _ptr(dbeta), replicas, _stream()), "kd6d_bn_train_bwd")
    return dx


class conv_pair:
    """`with ops.conv_pair(): convA(...); convB(...)` -- two independent, identically shaped 3x3 convolutions
    (forward or data gradient) as one launch (kd6d_conv2d_pair_begin/_end in kd6d.h).  The two convolutions run
    when the block closes: nothing inside it may consume their results."""

    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        global _pair_deferred
        if self.enabled:
            check(lib.kd6d_conv2d_pair_begin(), "kd6d_conv2d_pair_begin")
            if _prof is not None:
                _pair_deferred = []
        return self

    def __exit__(self, *exc):
        global _pair_deferred
        if self.enabled:
            deferred, _pair_deferred = _pair_deferred, None
            if deferred:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            check(lib.kd6d_conv2d_pair_end(), "kd6d_conv2d_pair_end")
            if deferred:       # one launch: its duration against the work of both convolutions
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                _prof.append((deferred[0][0], sum(d[1] for d in deferred), e0, e1, deferred[0][2]))
        return False


def bn_pool_train_fwd(x, y, batch, h, w, sum_, sumsq, gamma, beta, eps, momentum, running_mean, running_var,
                      save_mean, save_invstd, act):
    """BN(train) + act + MaxPool2d(2,2): x (batch*h*w, C) conv output -> y (batch*h/2*w/2, C) pooled."""
    rows, c = x.shape
    assert rows == batch * h * w and tuple(y.shape) == (batch * (h // 2) * (w // 2), c)
    check(lib.kd6d_bn_pool_train_fwd(dt_code(y.dtype), _xf32(x, y.dtype), _ptr(x), _ptr(y), batch, h, w, c,
                                     _ptr(sum_), _ptr(sumsq), _ptr(gamma), _ptr(beta), eps, momentum,
                                     _ptr(running_mean), _ptr(running_var), _ptr(save_mean), _ptr(save_invstd), act,
                                     _stream()), "kd6d_bn_pool_train_fwd")
    return y


def bn_pool_train_bwd(x, dy, dx, batch, h, w, mean, invstd, gamma, beta, act, ws_sum_dy, ws_sum_dy_xhat, dgamma,
                      dbeta, replicas=1, counter=None):
    """dy: gradient of the POOLED output; dx: gradient of the conv output x (same rows as x, dtype of dy);
    counter as in bn_train_bwd."""
    rows, c = x.shape
    assert rows == batch * h * w and tuple(dy.shape) == (batch * (h // 2) * (w // 2), c) and dx.shape == x.shape
    assert ws_sum_dy.numel() >= replicas * c * ACC_FLOATS and ws_sum_dy_xhat.numel() >= replicas * c * ACC_FLOATS
    assert counter is None or counter.numel() >= BARRIER_WORDS
    check(lib.kd6d_bn_pool_train_bwd(dt_code(dy.dtype), _xf32(x, dy.dtype), _ptr(x), _ptr(dy), _ptr(dx), batch, h, w,
                                     c, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), act, _ptr(ws_sum_dy),
                                     _ptr(ws_sum_dy_xhat), _ptr(counter), _ptr(dgamma), _ptr(dbeta), replicas,
                                     _stream()), "kd6d_bn_pool_train_bwd")
    return dx


def _hw_array(level_hw):
    arr = (ctypes.c_int32 * len(level_hw))(*[int(v) for v in level_hw])
    return arr


def gn_relu_fwd(x, y, level_hw, batch, groups, gamma, beta, eps, stats, flags=0):
    rows, c = x.shape
    assert rows == batch * sum(level_hw)
    assert stats.numel() >= len(level_hw) * batch * groups * 2 * ACC_FLOATS
    check(lib.kd6d_gn_relu_fwd(dt_code(y.dtype), _xf32(x, y.dtype), _ptr(x), _ptr(y), _hw_array(level_hw), len(level_hw),
                               batch, c, groups, _ptr(gamma), _ptr(beta), eps, _ptr(stats), flags, _stream()),
          "kd6d_gn_relu_fwd")
    return y


def gn_bwd_workspace_floats(n_levels, batch, groups):
    """fp32 slots of kd6d_gn_relu_bwd's workspace: the group-sum accumulators plus one barrier counter per (level, image)."""
    return 2 * n_levels * batch * groups * ACC_FLOATS + n_levels * batch


def gn_relu_bwd(x, dz, dx, level_hw, batch, groups, gamma, beta, stats, gsum_ws, dgamma, dbeta, acc_stride=0, eps=1e-5,
                flags=0):
    """dgamma / dbeta: lo-plane views of PLANAR gradient accumulators with stride acc_stride (ParamStore.acc /
    planar_acc), or None."""
    rows, c = x.shape
    assert rows == batch * sum(level_hw)
    assert gsum_ws.numel() >= gn_bwd_workspace_floats(len(level_hw), batch, groups), "gsum_ws too small (kd6d.h)"
    assert all(t is None or t.dtype == torch.int64 for t in (dgamma, dbeta))
    check(lib.kd6d_gn_relu_bwd(dt_code(dz.dtype), _xf32(x, dz.dtype), _ptr(x), _ptr(dz), _ptr(dx), _hw_array(level_hw),
                               len(level_hw), batch, c, groups, _ptr(gamma), _ptr(beta), eps, _ptr(stats),
                               _ptr(gsum_ws), _ptr(dgamma), _ptr(dbeta), int(acc_stride), flags, _stream()),
          "kd6d_gn_relu_bwd")
    return dx


def gn_relu_bwd_pair(items, level_hw, batch, groups, acc_stride, eps=1e-5, flags=0):
    """Two gn_relu_bwd of identical geometry as one launch (kd6d_gn_relu_bwd_pair).  items: two tuples
    (x, dz, dx, gamma, beta, stats, gsum_ws, dgamma, dbeta); dgamma / dbeta as in gn_relu_bwd."""
    assert len(items) == 2
    packed = []
    for (x, dz, dx, gamma, beta, stats, gsum_ws, dgamma, dbeta) in items:
        rows, c = x.shape
        assert rows == batch * sum(level_hw) and x.shape == items[0][0].shape and dz.dtype == items[0][1].dtype
        assert gsum_ws.numel() >= gn_bwd_workspace_floats(len(level_hw), batch, groups), "gsum_ws too small (kd6d.h)"
        it = _lib.GnItem()
        for name, t in zip(("x", "dz", "dx", "gamma", "beta", "stats", "gsum_ws", "dgamma", "dbeta"),
                           (x, dz, dx, gamma, beta, stats, gsum_ws, dgamma, dbeta)):
            assert t is None or (t.is_cuda and t.is_contiguous())
            setattr(it, name, t.data_ptr() if t is not None else None)
        packed.append(it)
    x0, dz0 = items[0][0], items[0][1]
    check(lib.kd6d_gn_relu_bwd_pair(dt_code(dz0.dtype), _xf32(x0, dz0.dtype), ctypes.byref(packed[0]),
                                    ctypes.byref(packed[1]), _hw_array(level_hw), len(level_hw), batch, x0.shape[1],
                                    groups, eps, int(acc_stride), flags, _stream()), "kd6d_gn_relu_bwd_pair")


def maxpool2_fwd(x, y, b, h, w):
    c = x.shape[-1]
    check(lib.kd6d_maxpool2_fwd(dt_code(x.dtype), _ptr(x), _ptr(y), b, h, w, c, _stream()),
          "kd6d_maxpool2_fwd")
    return y


def maxpool2_bwd(x, dy, dx, b, h, w, accumulate=False):
    c = x.shape[-1]
    check(lib.kd6d_maxpool2_bwd(dt_code(x.dtype), _ptr(x), _ptr(dy), _ptr(dx), b, h, w, c,
                                int(accumulate), _stream()), "kd6d_maxpool2_bwd")
    return dx


def upsample2_add(fine, coarse, out, b, h, w):
    c = fine.shape[-1]
    check(lib.kd6d_upsample2_add(dt_code(fine.dtype), _ptr(fine), _ptr(coarse), _ptr(out), b, h, w, c,
                                 _stream()), "kd6d_upsample2_add")
    return out


def sumpool2(dfine, dcoarse, b, h, w, accumulate=False):
    c = dfine.shape[-1]
    check(lib.kd6d_sumpool2(dt_code(dfine.dtype), _ptr(dfine), _ptr(dcoarse), b, h, w, c,
                            int(accumulate), _stream()), "kd6d_sumpool2")
    return dcoarse


ELT_RELU, ELT_RELU_BWD, ELT_ADD = 0, 1, 2


def eltwise(mode, x, dy, y):
    check(lib.kd6d_eltwise(dt_code(x.dtype), mode, _ptr(x), _ptr(dy), _ptr(y), x.numel(), _stream()),
          "kd6d_eltwise")
    return y


def image_to_nhwc(img, dtype, cpad=8, out=None):
    b, c, h, w = img.shape
    assert img.dtype == torch.float32
    if out is None:
        out = torch.empty((b * h * w, cpad), dtype=dtype, device=img.device)
    check(lib.kd6d_image_to_nhwc(dt_code(dtype), _ptr(img), _ptr(out), b, c, h, w, cpad, _stream()),
          "kd6d_image_to_nhwc")
    return out


def sinkhorn_div(xs, alpha, s_start, s_cnt, yt, beta, t_start, t_cnt, n_images, p, blur, scaling, reach, loss_kp=None):
    """xs (P,8,2), alpha (P,8), yt (M,8,2), beta (M,8); image b owns student rows
    [s_start[b], s_start[b]+s_cnt[b]) and teacher rows likewise (int32 device arrays).
    Returns loss_img (B), valid_img (B) int32, grad_xs (P,8,2), grad_alpha (P,8); loss_kp: optional (B,8) fp32
    output of the eight per-keypoint divergences."""
    dev = xs.device
    loss = torch.empty(n_images, dtype=torch.float32, device=dev)
    valid = torch.empty(n_images, dtype=torch.int32, device=dev)
    gx = torch.zeros_like(xs)
    ga = torch.zeros_like(alpha)
    check(lib.kd6d_sinkhorn_div_fwd_bwd(_ptr(xs), _ptr(alpha), _ptr(s_start), _ptr(s_cnt), _ptr(yt), _ptr(beta),
                                        _ptr(t_start), _ptr(t_cnt), n_images, p, blur, scaling,
                                        reach if reach is not None else -1.0, _ptr(loss), _ptr(valid), _ptr(loss_kp),
                                        _ptr(gx), _ptr(ga), _stream()), "kd6d_sinkhorn_div_fwd_bwd")
    return loss, valid, gx, ga


def sinkhorn_dense(x, alpha, y, beta, blur=0.05, scaling=0.5, reach=0.5, diameter=None, p=2.0):
    """Debiased (unbalanced) Sinkhorn divergence between two LARGE weighted point sets: x (N,D), alpha (N),
    y (M,D), beta (M), D in {2,4,8,16} -- losses/kd_loss.py:26-30 / loss_libs.py:47 with a dense grid of local
    predictions (BASELINE config 5).  Returns loss (1,), dS/dx (N,D), dS/dalpha (N).
    diameter=None reproduces geomloss' default: it is measured on the device and read back (one sync, like
    the reference's `.item()`); pass a float (geomloss' diameter= argument) to stay asynchronous."""
    N, D = x.shape
    M = y.shape[0]
    dev = x.device
    assert x.dtype == torch.float32 and y.shape == (M, D) and alpha.shape == (N,) and beta.shape == (M,)
    nws = int(lib.kd6d_sinkhorn_dense_workspace_floats(N, M, D))
    ws = torch.empty(nws, dtype=torch.float32, device=dev)
    if diameter is None:
        d = torch.empty(1, dtype=torch.float32, device=dev)
        check(lib.kd6d_sinkhorn_dense_diameter(_ptr(x), _ptr(y), N, M, D, _ptr(ws), _ptr(d), _stream()),
              "kd6d_sinkhorn_dense_diameter")
        diameter = float(d.item())
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    gx = torch.empty_like(x)
    ga = torch.empty_like(alpha)
    check(lib.kd6d_sinkhorn_dense_fwd_bwd(_ptr(x), _ptr(alpha), _ptr(y), _ptr(beta), N, M, D, p, blur, scaling,
                                          reach if reach is not None else -1.0, float(max(diameter, 1e-12)), _ptr(ws), nws,
                                          _ptr(loss), _ptr(gx), _ptr(ga), _stream()), "kd6d_sinkhorn_dense_fwd_bwd")
    return loss, gx, ga


# ---- grouped weight gradient (csrc/conv_wgrad_group.hip) -----------------------------------------------------
def wgrad_group_supported(geom, dtype):
    return bool(lib.kd6d_wgrad_group_supported(geom.ref, dt_code(dtype)))


class WgradGroup:
    """The weight gradients of several layers as one launch pair.  add() collects (geometry, x, dy, dw, dbias);
    launch() plans the work list on first use (or when the set of tensors changed), keeps the plan and the
    partial-tile slab on the device, and enqueues the two kernels on the current stream.  The collected tensors
    must stay alive and unchanged until the launch has run (the engine's per-layer buffers are static)."""

    def __init__(self, n_workgroups):
        self.n_workgroups = int(n_workgroups)
        self.items, self.flops = [], 0
        # one (plan, slab, info, tensors) per distinct set of tensors, kept for the life of the group: a captured
        # hipGraph replays the device pointers of the plan it was captured with, so a later eager launch with other
        # buffers (another batch size on the same network) must not free or overwrite it
        self._plans = {}

    def add(self, geom, x, dy, dw, dbias=None, flops=0):
        assert x.shape == (geom.rows_in, geom.cin) and dy.shape == (geom.rows_out, geom.cout)
        assert x.dtype == dy.dtype == torch.bfloat16 and dw.dtype == torch.float32
        assert dw.numel() == geom.cout * geom.ksize * geom.ksize * geom.cin
        assert dbias is None or (dbias.dtype == torch.float32 and dbias.numel() >= geom.cout)
        self.items.append((geom, x, dy, dw, dbias))
        self.flops += flops

    def __len__(self):
        return len(self.items)

    def _build(self, device):
        # the plan travels host -> device with a pageable copy and the slab is allocated: neither may happen inside a
        # stream capture (the engine's eager warm-up steps run the same launch first, so the capture finds the plan)
        assert not torch.cuda.is_current_stream_capturing(), \
            "WgradGroup: first launch of a new tensor set inside a graph capture (run one eager warm-up step first)"
        n = len(self.items)
        arr = (_lib.WgradItem * n)()
        for i, (g, x, dy, dw, db) in enumerate(self.items):
            ctypes.memmove(ctypes.byref(arr[i].geom), ctypes.byref(g.c), ctypes.sizeof(_lib.ConvGeom))
            arr[i].x, arr[i].dy, arr[i].dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
            arr[i].dbias = db.data_ptr() if db is not None else None
        info = (ctypes.c_int32 * 4)()
        nbytes = lib.kd6d_wgrad_group_plan(arr, n, _lib.KD6D_BF16, self.n_workgroups, None, 0, info)
        if nbytes < 0:
            check(int(nbytes), "kd6d_wgrad_group_plan")
        host = torch.zeros(int(nbytes), dtype=torch.uint8)
        rc = lib.kd6d_wgrad_group_plan(arr, n, _lib.KD6D_BF16, self.n_workgroups, ctypes.c_void_p(host.data_ptr()),
                                       int(nbytes), info)
        if rc < 0:
            check(int(rc), "kd6d_wgrad_group_plan")
        slab_floats = int(info[2]) + (int(info[3]) << 31)
        # allocated on the default stream: the caching allocator then never hands the blocks to another stream's
        # tensors while a side stream (where the launch runs) still uses them
        with torch.cuda.stream(torch.cuda.default_stream(device)):
            plan = host.to(device)
            slab = torch.empty(slab_floats, dtype=torch.float32, device=device)
        torch.cuda.current_stream().wait_stream(torch.cuda.default_stream(device))
        return dict(plan=plan, slab=slab, info=(int(info[0]), int(info[1])), keep=list(self.items))

    def launch(self):
        """Enqueue on the current stream; clears the collected items."""
        if not self.items:
            return
        key = (self.n_workgroups,) + tuple(
            (x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None if db is None else db.data_ptr(), g.rows_out)
            for g, x, dy, dw, db in self.items)
        ent = self._plans.get(key)
        if ent is None:
            ent = self._plans[key] = self._build(self.items[0][1].device)
        with _Timed("conv_wgrad", self.flops, self.items[0][0]):
            check(lib.kd6d_wgrad_group_launch(_ptr(ent["plan"]), ent["info"][0], ent["info"][1], _ptr(ent["slab"]),
                                              _stream()), "kd6d_wgrad_group_launch")
        self.items, self.flops = [], 0
