"""Per-kernel parity: every HIP kernel through the C ABI vs. the same op computed on the
host CPU in fp32/fp64 (torch CPU ops are the oracle here: the reference itself is plain
torch conv2d / batch_norm / group_norm / max_pool2d / interpolate).

bf16 kernels are checked on bf16-representable inputs against an fp32 CPU result, so the
only differences are fp32 accumulation order (tolerance 2e-5 * sum|a*b| bound, stated per
test) and, where the output is stored in bf16, one bf16 rounding (2^-8 relative).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util_pack import pack_levels, round_to, unpack_levels, w_to_dgrad, w_to_krsc

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float32]


def _ops():
    from kd6d import ops
    return ops


def _accs(n, dev, fill_zero=True):
    """n accumulators (kd6d_acc, include/kd6d.h "reproducible reductions") as an fp32 tensor: zeroed, or -- for the
    entry points that clear their workspace themselves -- uninitialised."""
    return (torch.zeros if fill_zero else torch.empty)(n * 4, device=dev)


def _acc_val(acc, n, grad=False):
    """fp32 values of n accumulators (kd6d_acc_read)."""
    ops = _ops()
    return ops.acc_read(acc, n, ops.ACC_GRAD if grad else ops.ACC_ACT)


class _GradAcc:
    """Stand-alone PLANAR gradient accumulators for tensors of the given sizes (what ParamStore.gacc is for the engine):
    .views[i] goes to the entry point together with .stride; .value(i) is the fp32 result."""

    def __init__(self, sizes, dev):
        ops = _ops()
        self.sizes, self.stride = list(sizes), sum(sizes)
        self.acc = ops.planar_acc(self.stride, dev)
        offs = [sum(self.sizes[:k]) for k in range(len(self.sizes))]
        self.views = [self.acc[o:o + n] for o, n in zip(offs, self.sizes)]
        self.offs = offs

    def value(self, i):
        v = _ops().planar_acc_value(self.acc)
        return v[self.offs[i]:self.offs[i] + self.sizes[i]]



def _option(name, value):
    """Pin a kernel family through kd6d_set_option for the rest of this test (put back by the fixture below)."""
    ops = _ops()
    _RESTORE.append((name, ops.get_option(name)))
    ops.set_option(name, value)


_RESTORE = []


@pytest.fixture(autouse=True)
def _restore_options():
    yield
    if _RESTORE:
        ops = _ops()
        while _RESTORE:
            name, old = _RESTORE.pop()
            ops.set_option(name, old)


def _tol(dtype, stored):
    # stored=True: result was rounded to `dtype` on the way out
    if dtype == torch.float32:
        return dict(rtol=2e-4, atol=2e-4)
    return dict(rtol=1.2e-2, atol=1.2e-2) if stored else dict(rtol=2e-4, atol=2e-4)


CONV_CASES = [
    # (B, Cin, Cout, k, stride, levels)
    (2, 8, 16, 3, 1, [(12, 10)]),
    (2, 16, 8, 1, 1, [(9, 7)]),
    (2, 32, 64, 3, 2, [(16, 16)]),
    (1, 64, 128, 3, 1, [(8, 8)]),
    (2, 128, 256, 1, 1, [(4, 4)]),
    (3, 24, 40, 3, 1, [(8, 8), (4, 4), (2, 2)]),       # multi-level, odd channel counts
    (2, 128, 16, 3, 1, [(8, 8), (4, 4), (2, 2), (1, 1)]),
    (1, 256, 512, 3, 2, [(6, 6)]),
    (2, 8, 32, 3, 1, [(40, 40)]),                      # exercises the 256-pixel tiles
    # 3x3/s1 with C % 64 == 0: the halo-patch kernel (several tiles, levels sharing a tile, N tail, 3 chunks)
    (2, 128, 256, 3, 1, [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]),
    (1, 64, 240, 3, 1, [(64, 64)]),
    (3, 192, 64, 3, 1, [(5, 7), (3, 2)]),
    (2, 16, 64, 3, 1, [(32, 32)]),                     # dgrad: 64 -> 16 channels, the 128x32 halo tile
    (2, 256, 24, 3, 1, [(16, 16), (8, 8)]),            # fwd: narrow result with an N tail
    # 3x3/s1 with C in {8, 16, 32}: the resident-patch kernel (wide maps, K padding, N tails, two channel tiles)
    (2, 8, 8, 3, 1, [(256, 256)]),
    (2, 32, 64, 3, 1, [(256, 256)]),
    (1, 16, 128, 3, 1, [(64, 64)]),
    (2, 32, 64, 3, 1, [(128, 128)]),
    (2, 32, 256, 3, 1, [(16, 16), (8, 8)]),
    (3, 16, 24, 3, 1, [(9, 7)]),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_plain(gpu_device, dtype, case):
    ops = _ops()
    B, Cin, Cout, k, stride, levels = case
    g = torch.Generator().manual_seed(B * 1000 + Cin * 7 + Cout)
    pad = k // 2
    xs = [round_to(torch.randn(B, Cin, h, w, generator=g), dtype) for (h, w) in levels]
    w = round_to(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5, dtype)
    geom = ops.Geom(B, Cin, Cout, k, stride, pad, levels)
    y = ops.conv2d_fwd(geom, pack_levels(xs, dtype).to(gpu_device), w_to_krsc(w, dtype).to(gpu_device),
                       out_f32=True)
    torch.cuda.synchronize()
    got = unpack_levels(y.cpu(), B, geom.levels_out)
    for x, gl in zip(xs, got):
        ref = F.conv2d(x, w, stride=stride, padding=pad)
        torch.testing.assert_close(gl, ref, **_tol(dtype, stored=False))


@pytest.mark.parametrize("case", [
    # few output tiles, long K: the split-K path (fp32 slabs in the workspace + finalize launch)
    (2, 256, 128, 3, 1, [(8, 8)]),
    (1, 512, 200, 3, 2, [(16, 16)]),          # stride 2, N tail (200 = 3 tiles + 8 channels)
    (2, 1024, 64, 1, 1, [(6, 6), (3, 3)]),    # 1x1, two levels
])
def test_conv_fwd_splitk(gpu_device, case):
    ops = _ops()
    dtype = torch.bfloat16
    B, Cin, Cout, k, stride, levels = case
    g = torch.Generator().manual_seed(5 + Cout)
    pad = k // 2
    xs = [round_to(torch.randn(B, Cin, h, w, generator=g), dtype) for (h, w) in levels]
    w = round_to(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5, dtype)
    geom = ops.Geom(B, Cin, Cout, k, stride, pad, levels)
    res = [round_to(torch.randn(B, Cout, h, w_, generator=g), dtype) for (h, w_) in geom.levels_out]
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    dev = gpu_device
    ws = torch.full((8 << 20,), float("nan"), device=dev)          # any contents
    xp, wp, rp = pack_levels(xs, dtype).to(dev), w_to_krsc(w, dtype).to(dev), pack_levels(res, dtype).to(dev)
    y_ws = ops.conv2d_fwd(geom, xp, wp, ch_scale=scale.to(dev), ch_shift=shift.to(dev), act=1, residual=rp, workspace=ws)
    y_1p = ops.conv2d_fwd(geom, xp, wp, ch_scale=scale.to(dev), ch_shift=shift.to(dev), act=1, residual=rp)
    yf = ops.conv2d_fwd(geom, xp, wp, out_f32=True, workspace=ws)
    torch.cuda.synchronize()
    assert not torch.isnan(ws[:geom.rows_out * Cout * 2]).any(), "the split-K path did not run (slabs untouched)"
    for x, r, a, b_, f in zip(xs, res, unpack_levels(y_ws.cpu(), B, geom.levels_out),
                              unpack_levels(y_1p.cpu(), B, geom.levels_out), unpack_levels(yf.cpu(), B, geom.levels_out)):
        conv = F.conv2d(x, w, stride=stride, padding=pad)
        ref = F.leaky_relu(conv * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), 0.1) + r
        torch.testing.assert_close(a, ref, **_tol(dtype, stored=True))
        torch.testing.assert_close(b_, ref, **_tol(dtype, stored=True))
        torch.testing.assert_close(f, conv, **_tol(dtype, stored=False))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_fwd_epilogue(gpu_device, dtype):
    ops = _ops()
    B, Cin, Cout, k, levels = 2, 32, 48, 3, [(8, 8), (4, 4)]
    g = torch.Generator().manual_seed(7)
    xs = [round_to(torch.randn(B, Cin, h, w, generator=g), dtype) for (h, w) in levels]
    res = [round_to(torch.randn(B, Cout, h, w, generator=g), dtype) for (h, w) in levels]
    w = round_to(torch.randn(Cout, Cin, k, k, generator=g) * 0.05, dtype)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    segs = torch.tensor([1.5, 0.25])
    geom = ops.Geom(B, Cin, Cout, k, 1, 1, levels)
    dev = gpu_device
    for act in (0, 1, 2):
        y = ops.conv2d_fwd(geom, pack_levels(xs, dtype).to(dev), w_to_krsc(w, dtype).to(dev),
                           ch_scale=scale.to(dev), ch_shift=shift.to(dev), act=act,
                           residual=pack_levels(res, dtype).to(dev), seg_scale=segs.to(dev))
        torch.cuda.synchronize()
        got = unpack_levels(y.cpu(), B, levels)
        for li, (x, r, gl) in enumerate(zip(xs, res, got)):
            ref = F.conv2d(x, w, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
            ref = ref * segs[li]
            if act == 1:
                ref = F.leaky_relu(ref, 0.1)
            elif act == 2:
                ref = F.relu(ref)
            ref = ref + r
            torch.testing.assert_close(gl, ref, **_tol(dtype, stored=True))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad(gpu_device, dtype, case):
    ops = _ops()
    B, Cin, Cout, k, stride, levels = case
    eg = 8 if dtype == torch.bfloat16 else 4
    if Cout % eg:
        pytest.skip("dgrad needs cout %% %d == 0" % eg)
    g = torch.Generator().manual_seed(11)
    pad = k // 2
    geom = ops.Geom(B, Cin, Cout, k, stride, pad, levels)
    w = round_to(torch.randn(Cout, Cin, k, k, generator=g) / (Cout * k * k) ** 0.5, dtype)
    dys = [round_to(torch.randn(B, Cout, h, w_, generator=g), dtype) for (h, w_) in geom.levels_out]
    dx0 = [round_to(torch.randn(B, Cin, h, w_, generator=g), dtype) for (h, w_) in levels]
    dev = gpu_device
    wt = w_to_dgrad(w, dtype).to(dev)
    dx = ops.conv2d_dgrad(geom, pack_levels(dys, dtype).to(dev), wt)
    dx_acc = pack_levels(dx0, dtype).to(dev)
    ops.conv2d_dgrad(geom, pack_levels(dys, dtype).to(dev), wt, dx=dx_acc, accumulate=True)
    torch.cuda.synchronize()
    got = unpack_levels(dx.cpu(), B, levels)
    got_acc = unpack_levels(dx_acc.cpu(), B, levels)
    for (h, w_), dy, gl, ga, d0 in zip(levels, dys, got, got_acc, dx0):
        ref = torch.nn.grad.conv2d_input((B, Cin, h, w_), w, dy, stride=stride, padding=pad)
        torch.testing.assert_close(gl, ref, **_tol(dtype, stored=True))
        torch.testing.assert_close(ga, ref + d0, **_tol(dtype, stored=True))


@pytest.mark.parametrize("B,levels", [(16, [(32, 32), (16, 16), (8, 8), (4, 4)]), (2, [(32, 32), (16, 16), (8, 8), (4, 4)]),
                                      (3, [(20, 12), (7, 5)])])
def test_conv_pair_bracket_matches_separate_launches(gpu_device, B, levels):
    """ops.conv_pair: the cls- and pose-tower convolutions of a head layer (same geometry, different tensors,
    weights, biases and GroupNorm statistics) as ONE launch -- forward with fused bias + statistics, then the two
    data gradients.  Same values as two separate launches (the per-output summation order does not depend on the
    tile), and the separate launches are checked against torch elsewhere in this file.  B = 16 is the student
    head of the benchmark (228 tiles of 192x128 for the pair)."""
    ops = _ops()
    dev = gpu_device
    dtype = torch.bfloat16
    C, G = 128, 32
    g = torch.Generator().manual_seed(B)
    geom = ops.Geom(B, C, C, 3, 1, 1, levels)
    rows = geom.rows_out

    def mk():
        x = round_to(torch.randn(rows, C, generator=g), dtype).to(dtype).to(dev)
        w = round_to(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5, dtype)
        bias = torch.randn(C, generator=g).to(dev)
        return x, w_to_krsc(w, dtype).to(dev), w_to_dgrad(w, dtype).to(dev), bias

    A, Bt = mk(), mk()
    n_stats = len(levels) * B * G * 2

    def run(paired):
        outs = []
        with ops.conv_pair(enabled=paired):
            for x, wk, wt, bias in (A, Bt):
                stats = _accs(n_stats, dev)
                y = ops.conv2d_fwd(geom, x, wk, ch_shift=bias, out_f32=True, stats=stats, stats_groups=G)
                outs.append((y, stats))
        dxs = []
        with ops.conv_pair(enabled=paired):
            for (x, wk, wt, bias), (y, _) in zip((A, Bt), outs):
                dxs.append(ops.conv2d_dgrad(geom, x, wt))          # any (rows, C) tensor serves as dy
        torch.cuda.synchronize()
        return [(y.cpu(), _acc_val(st, n_stats).cpu()) for y, st in outs], [d.cpu() for d in dxs]

    (fa, fb), (da, db) = run(True)
    (ra, rb), (ea, eb) = run(False)
    for (y, st), (yr, sr) in ((fa, ra), (fb, rb)):
        torch.testing.assert_close(y, yr, rtol=1e-6, atol=1e-6)
        assert torch.equal(st, sr)              # same tiles, fixed-point accumulators: the statistics agree bit for bit
    torch.testing.assert_close(da.float(), ea.float(), rtol=0, atol=0)
    torch.testing.assert_close(db.float(), eb.float(), rtol=0, atol=0)
    assert not torch.equal(fa[0], fb[0])
    assert ops.lib.kd6d_conv2d_pair_pending() == 0


WGRAD_EXTRA = [
    # 128-wide dW tiles (the LDS-DMA ring kernel): several taps per 128-column row, partial j-tile, odd sizes
    (2, 16, 128, 3, 1, [(32, 32)]),
    (3, 32, 256, 3, 1, [(16, 16), (7, 5)]),
    (2, 8, 72, 3, 2, [(33, 31)]),
    (16, 128, 128, 3, 1, [(32, 32), (16, 16), (8, 8), (4, 4)]),     # the student head shape (many splits)
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES + WGRAD_EXTRA)
def test_conv_wgrad(gpu_device, dtype, case):
    ops = _ops()
    B, Cin, Cout, k, stride, levels = case
    eg = 8 if dtype == torch.bfloat16 else 4
    if Cout % eg:
        pytest.skip("wgrad needs cout %% %d == 0" % eg)
    g = torch.Generator().manual_seed(13)
    pad = k // 2
    geom = ops.Geom(B, Cin, Cout, k, stride, pad, levels)
    xs = [round_to(torch.randn(B, Cin, h, w_, generator=g), dtype) for (h, w_) in levels]
    dys = [round_to(torch.randn(B, Cout, h, w_, generator=g), dtype) for (h, w_) in geom.levels_out]
    dev = gpu_device
    dw, db = ops.conv2d_wgrad_f32(geom, pack_levels(xs, dtype).to(dev), pack_levels(dys, dtype).to(dev), with_bias=True)
    torch.cuda.synchronize()
    dw = dw.view(Cout, k, k, Cin)
    ref = torch.zeros(Cout, Cin, k, k)
    ref_b = torch.zeros(Cout, dtype=torch.float64)
    for x, dy in zip(xs, dys):
        ref += torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, stride=stride, padding=pad)
        ref_b += dy.double().sum(dim=(0, 2, 3))
    got = dw.cpu().permute(0, 3, 1, 2)
    scale = float(ref.abs().max())
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4 * max(scale, 1.0))
    torch.testing.assert_close(db.cpu().double(), ref_b, rtol=2e-4, atol=2e-4 * max(float(ref_b.abs().max()), 1.0))


@pytest.mark.parametrize("case", [
    # (B, Cin, Cout, k, levels): the persistent padded-patch kernel for small dW (Cin in {8,16,32}, Cout <= 64, no bias)
    (2, 8, 8, 3, [(256, 256)]),
    (2, 8, 16, 3, [(128, 128)]),
    (4, 16, 8, 1, [(64, 64)]),
    (4, 8, 64, 3, [(64, 64)]),
    (2, 32, 64, 3, [(64, 64)]),
    (2, 32, 40, 1, [(64, 64)]),
    (3, 16, 24, 3, [(9, 7)]),          # ragged: tiles shorter than the image, odd width
    (2, 8, 8, 3, [(3, 300)]),          # wider than 256: stays on the general kernel
])
def test_conv_wgrad_small_layers(gpu_device, case):
    ops = _ops()
    _option("wgrad.small", 1)
    dtype = torch.bfloat16
    B, Cin, Cout, k, levels = case
    g = torch.Generator().manual_seed(3 + Cout)
    pad = k // 2
    geom = ops.Geom(B, Cin, Cout, k, 1, pad, levels)
    xs = [round_to(torch.randn(B, Cin, h, w_, generator=g), dtype) for (h, w_) in levels]
    dys = [round_to(torch.randn(B, Cout, h, w_, generator=g), dtype) for (h, w_) in geom.levels_out]
    dev = gpu_device
    dw = sum(ops.conv2d_wgrad_f32(geom, pack_levels(xs, dtype).to(dev), pack_levels(dys, dtype).to(dev), cu_budget=budget)[0]
             for budget in (0, 16)).view(Cout, k, k, Cin)
    torch.cuda.synchronize()
    ref = torch.zeros(Cout, Cin, k, k)
    for x, dy in zip(xs, dys):
        ref += torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, stride=1, padding=pad)
    got = dw.cpu().permute(0, 3, 1, 2) / 2
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4 * max(float(ref.abs().max()), 1.0))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # (B, Cin, Cout, k, stride, levels, groups)   groups 0 = per-channel (BatchNorm) statistics
    (2, 8, 16, 3, 1, [(12, 10)], 0),
    (3, 32, 64, 3, 2, [(16, 16)], 0),
    (2, 16, 256, 1, 1, [(9, 9)], 0),
    (3, 128, 128, 3, 1, [(8, 8), (4, 4), (2, 2), (1, 1)], 32),      # 4 channels per group, tiny levels
    (2, 64, 256, 3, 1, [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)], 32),   # 8 per group, halo kernel
    (5, 128, 256, 3, 1, [(3, 3)], 32),                              # fragments straddle images
])
def test_conv_fwd_fused_statistics(gpu_device, dtype, case):
    """Statistics accumulated by the conv epilogue == sums over the stored fp32 output
    (what kd6d_colstats / the GroupNorm reduction pass would produce)."""
    ops = _ops()
    B, Cin, Cout, k, stride, levels, groups = case
    g = torch.Generator().manual_seed(17 + Cout)
    pad = k // 2
    xs = [round_to(torch.randn(B, Cin, h, w, generator=g), dtype) for (h, w) in levels]
    w = round_to(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5, dtype)
    bias = torch.randn(Cout, generator=g)
    geom = ops.Geom(B, Cin, Cout, k, stride, pad, levels)
    dev = gpu_device
    n_stats = 2 * Cout if groups == 0 else len(levels) * B * groups * 2
    stats = _accs(n_stats, dev)
    y = ops.conv2d_fwd(geom, pack_levels(xs, dtype).to(dev), w_to_krsc(w, dtype).to(dev), ch_shift=bias.to(dev),
                       out_f32=True, stats=stats, stats_groups=groups)
    torch.cuda.synchronize()
    got_y = unpack_levels(y.cpu(), B, geom.levels_out)
    st = _acc_val(stats, n_stats).cpu().double()
    if groups == 0:
        s1 = sum(t.double().sum(dim=(0, 2, 3)) for t in got_y)
        s2 = sum((t.double() ** 2).sum(dim=(0, 2, 3)) for t in got_y)
        ref = torch.cat([s1, s2])
    else:
        parts = []
        for t in got_y:                       # (B, C, h, w) -> (B, G, cpg*h*w)
            tg = t.double().reshape(B, groups, -1)
            parts.append(torch.stack([tg.sum(-1), (tg ** 2).sum(-1)], dim=-1).reshape(-1))
        ref = torch.cat(parts)
    torch.testing.assert_close(st, ref, rtol=1e-4, atol=1e-4 * max(float(ref.abs().max()), 1.0))
    for x, gl in zip(xs, got_y):              # and the output itself is unchanged by the statistics pass
        torch.testing.assert_close(gl, F.conv2d(x, w, bias, stride=stride, padding=pad), **_tol(dtype, stored=False))


@pytest.mark.parametrize("dtype", [torch.bfloat16])
@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[3] == 3 and c[4] == 1])
def test_conv_resident_patch_kernel_on_every_small_c_case(gpu_device, dtype, case):
    """The resident-patch kernel (C in {8,16,32}) is dispatched from 2^17 pixels up; option conv.smallc = 1 lifts the size rule so the small / ragged 3x3 CONV_CASES (multi-level, N tails, two channel tiles) run
    through it as well -- forward, dgrad (where its gather source is narrow) and the fused BatchNorm statistics."""
    _option("conv.smallc", 1)
    test_conv_fwd_plain(gpu_device, dtype, case)
    test_conv_dgrad(gpu_device, dtype, case)
    B, Cin, Cout, k, stride, levels = case
    if Cin in (8, 16, 32) and Cout % 4 == 0:
        test_conv_fwd_fused_statistics(gpu_device, dtype, (B, Cin, Cout, k, stride, levels, 0))


@pytest.mark.parametrize("variant", [11, 12, 13, 14, 15])
@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[3] == 3 and c[4] == 1 and c[1] % 64 == 0
                                  and max(w for _, w in c[5]) <= 32])
def test_conv_halo_two_per_cu_variants(gpu_device, case, variant):
    """The single-patch-buffer halo variants built for two workgroups per CU (option conv.halo = 11: 128x128 on 4
    waves, 12: on 8 waves, 13: 128x64, 14: 64x64, 15: 128x32; maps up to 32 wide): forward, data gradient and the
    fused statistics on the multi-level, N-tail and 3-chunk cases -- the chunk-boundary reload is what these exercise
    (Cin 128, 192, 256).  The default dispatch uses them too (conv.halo_pairing = 1); the double-buffered originals
    stay covered by the same cases with the pairing switched off."""
    _option("conv.halo", variant)
    test_conv_fwd_plain(gpu_device, torch.bfloat16, case)
    test_conv_dgrad(gpu_device, torch.bfloat16, case)
    B, Cin, Cout, k, stride, levels = case
    if Cout % 4 == 0:
        test_conv_fwd_fused_statistics(gpu_device, torch.bfloat16, (B, Cin, Cout, k, stride, levels, 0))


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[3] == 3 and c[4] == 1 and c[1] % 64 == 0])
def test_conv_halo_one_per_cu_originals(gpu_device, case):
    _option("conv.halo_pairing", 0)
    test_conv_fwd_plain(gpu_device, torch.bfloat16, case)
    test_conv_dgrad(gpu_device, torch.bfloat16, case)
    B, Cin, Cout, k, stride, levels = case
    if Cout % 4 == 0:
        test_conv_fwd_fused_statistics(gpu_device, torch.bfloat16, (B, Cin, Cout, k, stride, levels, 0))


def test_pack_dgrad_weights(gpu_device):
    ops = _ops()
    dev = gpu_device
    g = torch.Generator().manual_seed(3)
    # (cout, cin, k): narrow layers (element-wise path) and cin % 64 == 0, cout % 16 == 0 layers (the LDS-tiled path: 16 x 64
    # tiles, more tiles than blocks for 240 x 128 x 3 x 3 and fewer for 16 x 64 x 1 x 1)
    shapes = [(16, 8, 3), (8, 16, 1), (40, 24, 3), (128, 128, 3), (240, 128, 3), (16, 64, 1), (64, 256, 1), (48, 192, 3)]
    for dtype in DTYPES:
        ws = [torch.randn(o, k, k, i, generator=g).to(dtype) for (o, i, k) in shapes]
        flat = torch.cat([w.reshape(-1) for w in ws]).to(dev)
        out = torch.zeros_like(flat)
        desc, off, blk = [], 0, 0
        for (o, i, k) in shapes:
            n = o * i * k * k
            desc += [off, off, o, i, k, blk]
            off += n
            blk += (n + 2047) // 2048
        ops.pack_dgrad_weights(flat, out, torch.tensor(desc, dtype=torch.int32, device=dev), len(shapes), blk)
        torch.cuda.synchronize()
        off = 0
        for w, (o, i, k) in zip(ws, shapes):
            n = o * i * k * k
            got = out[off:off + n].cpu().reshape(i, k, k, o)
            assert torch.equal(got, w.permute(3, 1, 2, 0).contiguous())
            off += n


MIXED = [(torch.bfloat16, False), (torch.bfloat16, True), (torch.float32, False)]   # (activation dtype, x kept fp32)


@pytest.mark.parametrize("dtype,xf32", MIXED)
@pytest.mark.parametrize("C,rows", [(8, 1000), (16, 4096), (64, 777), (256, 300), (512, 64), (1024, 40), (8, 300000), (16, 70001)])
@pytest.mark.parametrize("onepass", [False, True])
def test_batchnorm_train_fwd_bwd(gpu_device, dtype, xf32, C, rows, onepass):
    """onepass: the backward as one launch whose workgroups meet at an in-kernel barrier (kd6d_bn_train_bwd with a
    counter; the size limit is lifted so that the long tensors run 512-workgroup grids) against the reduce + apply
    pair."""
    ops = _ops()
    _option("bn.onepass_max", 1 << 40)
    dev = gpu_device
    g = torch.Generator().manual_seed(C + rows)
    x = round_to(torch.randn(rows, C, generator=g) * 2 + 0.5, torch.float32 if xf32 else dtype)
    dz = round_to(torch.randn(rows, C, generator=g), dtype)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1
    rm, rv = torch.zeros(C), torch.ones(C)
    # CPU oracle (double)
    xr = x.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = beta.double().requires_grad_(True)
    rmr, rvr = rm.double().clone(), rv.double().clone()
    yr = F.leaky_relu(F.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5), 0.1)
    yr.backward(dz.double())
    xd, dzd = x.to(torch.float32 if xf32 else dtype).to(dev), dz.to(dtype).to(dev)
    s1 = _accs(C, dev); s2 = _accs(C, dev)
    ops.colstats(xd, s1, s2)
    y = torch.empty_like(dzd)
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    mean = torch.empty(C, device=dev); invstd = torch.empty(C, device=dev)
    ops.bn_train_fwd(xd, y, s1, s2, gamma.to(dev), beta.to(dev), 1e-5, 0.1, rm_d, rv_d, mean, invstd, 1)
    dx = torch.empty_like(dzd)
    R = 1 if rows < 1000 else 8            # replica rows of the backward accumulators (kd6d.h)
    w1 = _accs(R * C, dev); w2 = _accs(R * C, dev)
    dgam = torch.zeros(C, device=dev); dbet = torch.zeros(C, device=dev)
    counter = torch.zeros(32, dtype=torch.int32, device=dev) if onepass else None
    ops.bn_train_bwd(xd, dzd, dx, mean, invstd, gamma.to(dev), beta.to(dev), 1, w1, w2, dgam, dbet, replicas=R,
                     counter=counter)
    torch.cuda.synchronize()
    assert ops.lib.kd6d_barrier_timeouts() == 0
    if onepass:
        eg = 4 if dtype == torch.float32 else 8
        fits = rows * (C // eg) <= 512 * 256 * 4        # 512 resident workgroups x 4 granules per thread (kd6d.h)
        assert (int(counter[0].item()) > 0) == fits, "one-launch path taken / not taken against the documented rule"
    tol = _tol(dtype, stored=True)
    torch.testing.assert_close(y.cpu().double(), yr.detach(), **tol)
    torch.testing.assert_close(rm_d.cpu().double(), rmr, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv_d.cpu().double(), rvr, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dx.cpu().double(), xr.grad, **tol)
    gs = max(1.0, float(gr.grad.abs().max()))
    torch.testing.assert_close(dgam.cpu().double(), gr.grad, rtol=1e-3, atol=1e-3 * gs)
    torch.testing.assert_close(dbet.cpu().double(), br.grad, rtol=1e-3, atol=1e-3 * gs)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,xf32", [(torch.float32, True), (torch.bfloat16, True), (torch.bfloat16, False)])
@pytest.mark.parametrize("B,H,W,C", [(2, 8, 12, 8), (3, 16, 16, 64), (2, 64, 64, 16), (16, 64, 64, 8), (1, 2, 2, 256)])
def test_bn_pool_fused_pair_equals_bn_then_maxpool(gpu_device, dtype, xf32, B, H, W, C):
    """BN(train)+LeakyReLU+MaxPool2d(2,2) as one kernel (darknet.py:94-97 behind a ConvBlock): the pooled output is
    bit-identical to bn_train_fwd -> maxpool2_fwd, the backward matches bn_train_bwd(maxpool2_bwd(.)) up to the
    order of the fp32 reductions, and both agree with torch autograd in float64.  Ties inside a window (common in
    bf16) exercise the first-maximum rule."""
    ops = _ops()
    dev = gpu_device
    _option("bn.onepass_max", 1 << 40)      # the one-launch backward at every size
    if dtype == torch.float32 and C % 4:
        pytest.skip("granule")
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + C)
    rows = B * H * W
    xt = torch.float32 if xf32 else dtype
    x = torch.randn(rows, C, generator=g) * 2 + 0.5
    n_tie = x[1::3].shape[0]
    x[::3][:n_tie] = x[1::3]                     # equal neighbours -> ties inside windows
    x = round_to(x, xt)
    dy = round_to(torch.randn(B * (H // 2) * (W // 2), C, generator=g), dtype)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    xd, dyd = x.to(xt).to(dev), dy.to(dtype).to(dev)
    s1 = _accs(C, dev); s2 = _accs(C, dev)
    ops.colstats(xd, s1, s2)
    R = 1 if rows < 1000 else 8

    def run(fused, onepass=False):
        counter = torch.zeros(32, dtype=torch.int32, device=dev) if onepass else None
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        mean = torch.empty(C, device=dev); invstd = torch.empty(C, device=dev)
        w1 = _accs(R * C, dev); w2 = _accs(R * C, dev)
        dgam = torch.zeros(C, device=dev); dbet = torch.zeros(C, device=dev)
        yp = torch.empty_like(dyd)
        dx = torch.empty(rows, C, dtype=dtype, device=dev)
        if fused:
            ops.bn_pool_train_fwd(xd, yp, B, H, W, s1, s2, gamma, beta, 1e-5, 0.1, rm, rv, mean, invstd, 1)
            ops.bn_pool_train_bwd(xd, dyd, dx, B, H, W, mean, invstd, gamma, beta, 1, w1, w2, dgam, dbet, replicas=R,
                                  counter=counter)
        else:
            z = torch.empty(rows, C, dtype=dtype, device=dev)
            ops.bn_train_fwd(xd, z, s1, s2, gamma, beta, 1e-5, 0.1, rm, rv, mean, invstd, 1)
            ops.maxpool2_fwd(z, yp, B, H, W)
            dz = torch.empty_like(z)
            ops.maxpool2_bwd(z, dyd, dz, B, H, W)
            ops.bn_train_bwd(xd, dz, dx, mean, invstd, gamma, beta, 1, w1, w2, dgam, dbet, replicas=R)
        torch.cuda.synchronize()
        return [t.cpu() for t in (yp, dx, dgam, dbet, rm, rv)]

    fy, fdx, fdg, fdb, frm, frv = run(True)
    uy, udx, udg, udb, urm, urv = run(False)
    oy, odx, odg, odb, _, _ = run(True, onepass=True)       # backward as one launch (in-kernel barrier)
    assert ops.lib.kd6d_barrier_timeouts() == 0
    torch.testing.assert_close(odx.double(), fdx.double(), **_tol(dtype, stored=True))
    torch.testing.assert_close(odg, fdg, rtol=1e-4, atol=1e-4 * max(1.0, float(fdg.abs().max())))
    torch.testing.assert_close(odb, fdb, rtol=1e-4, atol=1e-4 * max(1.0, float(fdg.abs().max())))
    assert torch.equal(fy, uy)
    assert torch.equal(frm, urm) and torch.equal(frv, urv)
    tol = _tol(dtype, stored=True)
    torch.testing.assert_close(fdx.double(), udx.double(), **tol)
    gs = max(1.0, float(udg.abs().max()))
    torch.testing.assert_close(fdg, udg, rtol=1e-4, atol=1e-4 * gs)
    torch.testing.assert_close(fdb, udb, rtol=1e-4, atol=1e-4 * gs)
    if dtype == torch.float32:                  # float64 autograd (no ties to speak of after fp32 rounding)
        xr = x.double().requires_grad_(True)
        gr, br = gamma.cpu().double().requires_grad_(True), beta.cpu().double().requires_grad_(True)
        a = F.leaky_relu(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5), 0.1)
        p = F.max_pool2d(a.view(B, H, W, C).permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).reshape(-1, C)
        p.backward(dy.double())
        torch.testing.assert_close(fy.double(), p.detach(), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(fdx.double(), xr.grad, rtol=2e-3, atol=2e-3)
        torch.testing.assert_close(fdg.double(), gr.grad, rtol=1e-3, atol=1e-3 * gs)


def test_in_kernel_barriers_under_repetition(gpu_device):
    """The one-launch BN / GN backward kernels exchange partial sums across workgroups inside the launch.  300
    back-to-back launches each (512- and 274-workgroup BN grids, 8-workgroup GN sibling groups) must all reproduce
    the two-launch result: a workgroup that left the barrier before a sibling's partial sum was visible would miss
    ~1/274 of a total, far outside the tolerance.  No barrier may time out."""
    ops = _ops()
    dev = gpu_device
    _option("bn.onepass_max", 1 << 40)
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(99)
    for C, rows in [(16, 70001), (64, 65536)]:
        x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(dev)
        dz = (torch.rand(rows, C, generator=g) + 0.5).to(bf).to(dev)         # positive: large, stable sums
        gamma = (torch.rand(C, generator=g) + 0.5).to(dev); beta = torch.zeros(C, device=dev)
        s1 = _accs(C, dev); s2 = _accs(C, dev)
        ops.colstats(x, s1, s2)
        mean = torch.empty(C, device=dev); invstd = torch.empty(C, device=dev)
        y = torch.empty(rows, C, dtype=bf, device=dev)
        ops.bn_train_fwd(x, y, s1, s2, gamma, beta, 1e-5, 0.1, torch.zeros(C, device=dev), torch.ones(C, device=dev),
                         mean, invstd, 1)

        def bwd(counter):
            w1 = _accs(8 * C, dev); w2 = _accs(8 * C, dev)
            dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
            dx = torch.empty(rows, C, dtype=bf, device=dev)
            ops.bn_train_bwd(x, dz, dx, mean, invstd, gamma, beta, 1, w1, w2, dg, db, replicas=8, counter=counter)
            return dx, dg, db

        rdx, rdg, rdb = bwd(None)
        first = None
        for it in range(300):
            counter = torch.zeros(32, dtype=torch.int32, device=dev)
            dx, dg, db = bwd(counter)
            assert float((dg - rdg).abs().max()) <= 2e-5 * float(rdg.abs().max()) + 1e-3, (C, rows, it)
            assert float((db - rdb).abs().max()) <= 2e-5 * float(rdb.abs().max()) + 1e-3, (C, rows, it)
            assert bool(((dx.float() - rdx.float()).abs() <= 2e-2 * rdx.float().abs().clamp(min=1.0)).all()), (C, rows, it)
            # fixed-point partial sums: every repetition of the launch gives the SAME bits
            if first is None:
                first = (dx.clone(), dg.clone(), db.clone())
            else:
                assert torch.equal(dx, first[0]) and torch.equal(dg, first[1]) and torch.equal(db, first[2]), (C, rows, it)
    # GroupNorm: 32x32 level -> 8 sibling workgroups per image
    C, G, B, levels = 128, 32, 4, [(32, 32), (16, 16)]
    hw = [h * w for h, w in levels]
    rows = B * sum(hw)
    x = torch.randn(rows, C, generator=g).to(dev)
    dz = (torch.rand(rows, C, generator=g) + 0.5).to(bf).to(dev)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev); beta = (torch.rand(C, generator=g)).to(dev)
    stats = _accs(len(levels) * B * G * 2, dev, fill_zero=False)
    y = torch.empty(rows, C, dtype=bf, device=dev)
    ops.gn_relu_fwd(x, y, hw, B, G, gamma, beta, 1e-5, stats)
    ref = first = None
    for it in range(301):
        ops.set_option("gn.onepass", 0 if it == 0 else 1)
        try:
            gsum = torch.empty(ops.gn_bwd_workspace_floats(len(levels), B, G), device=dev)
            ga = _GradAcc([C, C], dev)
            dx = torch.empty(rows, C, dtype=bf, device=dev)
            ops.gn_relu_bwd(x, dz, dx, hw, B, G, gamma, beta, stats, gsum, ga.views[0], ga.views[1], ga.stride)
            dg = ga.value(0)
        finally:
            ops.set_option("gn.onepass", 1)
        if it == 0:
            ref = (dx.float(), dg.clone())
        else:
            assert bool(((dx.float() - ref[0]).abs() <= 2e-2 * ref[0].abs().clamp(min=1.0)).all()), it
            assert float((dg - ref[1]).abs().max()) <= 2e-5 * float(ref[1].abs().max()) + 1e-3, it
            if first is None:
                first = (dx.clone(), ga.acc.clone())
            else:               # bitwise: the same launch, the same bits (accumulator words included)
                assert torch.equal(dx, first[0]) and torch.equal(ga.acc, first[1]), it
    torch.cuda.synchronize()
    assert ops.lib.kd6d_barrier_timeouts() == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,rows", [(240, 1003), (16, 50), (40, 333)])
def test_colstats_any_channel_count(gpu_device, dtype, C, rows):
    ops = _ops()
    g = torch.Generator().manual_seed(C)
    x = round_to(torch.randn(rows, C, generator=g), dtype)
    s1 = _accs(C, gpu_device); s2 = _accs(C, gpu_device)
    ops.colstats(x.to(dtype).to(gpu_device), s1, s2)
    torch.cuda.synchronize()
    torch.testing.assert_close(_acc_val(s1, C).cpu(), x.sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(_acc_val(s2, C).cpu(), (x * x).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype,xf32", MIXED)
@pytest.mark.parametrize("C", [128, 256])
@pytest.mark.parametrize("levels,onepass", [([(6, 6), (3, 3), (2, 2), (1, 1)], "1"), ([(6, 6), (3, 3), (2, 2), (1, 1)], "0"),
                                            ([(32, 32), (16, 12), (5, 5)], "1"), ([(32, 32), (16, 12), (5, 5)], "0")])
def test_groupnorm_relu_fwd_bwd(gpu_device, dtype, xf32, C, levels, onepass):
    """onepass "1": the backward is one kernel whose workgroups of a (level, image) meet at an in-kernel barrier
    (32x32 levels span 8..32 workgroups); "0": the reduce + apply pair."""
    ops = _ops()
    dev = gpu_device
    _option("gn.onepass", int(onepass))
    B, G = 3, 32
    g = torch.Generator().manual_seed(C)
    xdt = torch.float32 if xf32 else dtype
    xs = [round_to(torch.randn(B, C, h, w, generator=g) * 1.5 + 0.2, xdt) for (h, w) in levels]
    dzs = [round_to(torch.randn(B, C, h, w, generator=g), dtype) for (h, w) in levels]
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    gr = gamma.double().requires_grad_(True)
    br = beta.double().requires_grad_(True)
    refs, xrs = [], []
    for x, dz in zip(xs, dzs):
        xr = x.double().requires_grad_(True)
        yr = F.relu(F.group_norm(xr, G, gr, br, 1e-5))
        yr.backward(dz.double())
        refs.append(yr.detach()); xrs.append(xr)
    hw = [h * w for (h, w) in levels]
    xp = pack_levels(xs, xdt).to(dev); dzp = pack_levels(dzs, dtype).to(dev)
    y = torch.empty_like(dzp); dx = torch.empty_like(dzp)
    stats = _accs(len(levels) * B * G * 2, dev, fill_zero=False)
    gsum = torch.empty(ops.gn_bwd_workspace_floats(len(levels), B, G), device=dev)
    ga = _GradAcc([C, C], dev)
    ops.gn_relu_fwd(xp, y, hw, B, G, gamma.to(dev), beta.to(dev), 1e-5, stats)     # flags=0: reduces + zeroes itself
    ops.gn_relu_bwd(xp, dzp, dx, hw, B, G, gamma.to(dev), beta.to(dev), stats, gsum, ga.views[0], ga.views[1], ga.stride)
    torch.cuda.synchronize()
    dgam, dbet = ga.value(0), ga.value(1)
    assert ops.lib.kd6d_barrier_timeouts() == 0
    tol = _tol(dtype, stored=True)
    for gl, ref in zip(unpack_levels(y.cpu(), B, levels), refs):
        torch.testing.assert_close(gl.double(), ref, **tol)
    for gl, xr in zip(unpack_levels(dx.cpu(), B, levels), xrs):
        torch.testing.assert_close(gl.double(), xr.grad, rtol=tol["rtol"], atol=3 * tol["atol"])
    gs = max(1.0, float(gr.grad.abs().max()))
    torch.testing.assert_close(dgam.cpu().double(), gr.grad, rtol=1e-3, atol=1e-3 * gs)
    torch.testing.assert_close(dbet.cpu().double(), br.grad, rtol=1e-3, atol=1e-3 * gs)


@pytest.mark.parametrize("onepass", [1, 0])
@pytest.mark.parametrize("levels", [[(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)], [(6, 6), (3, 3), (1, 1)]])
def test_groupnorm_bwd_pair_equals_two_launches(gpu_device, levels, onepass):
    """kd6d_gn_relu_bwd_pair: the two tower layers' GroupNorm backwards in one launch give what two launches give
    (dx to one bf16 ulp on a few elements -- same code, group sums differ by atomic order; dgamma / dbeta to
    atomic-order noise); with gn.onepass = 0 the entry falls back to two launch pairs."""
    ops = _ops()
    dev = gpu_device
    _option("gn.onepass", onepass)
    B, G, C = 4, 32, 128
    hw = [h * w for (h, w) in levels]
    rows = B * sum(hw)
    g = torch.Generator().manual_seed(11)
    bf = torch.bfloat16

    def make():
        x = (torch.randn(rows, C, generator=g) * 1.5 + 0.2).to(dev)                     # fp32 raw conv output
        dz = torch.randn(rows, C, generator=g).to(bf).to(dev)
        gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
        beta = (torch.randn(C, generator=g) * 0.2).to(dev)
        stats = _accs(len(levels) * B * G * 2, dev, fill_zero=False)
        y = torch.empty(rows, C, dtype=bf, device=dev)
        ops.gn_relu_fwd(x, y, hw, B, G, gamma, beta, 1e-5, stats)
        return x, dz, gamma, beta, stats

    sets = [make(), make()]
    nws = ops.gn_bwd_workspace_floats(len(levels), B, G)

    def run(pair):
        outs, items = [], []
        ga = _GradAcc([C, C, C, C], dev)                 # one accumulator image for both items, as in the engine
        for k, (x, dz, gamma, beta, stats) in enumerate(sets):
            dx = torch.empty(rows, C, dtype=bf, device=dev)
            gsum = torch.empty(nws, device=dev)
            outs.append(dx)
            items.append((x, dz, dx, gamma, beta, stats, gsum, ga.views[2 * k], ga.views[2 * k + 1]))
        if pair:
            ops.gn_relu_bwd_pair(items, hw, B, G, ga.stride)
        else:
            for (x, dz, dx, gamma, beta, stats, gsum, dgam, dbet) in items:
                ops.gn_relu_bwd(x, dz, dx, hw, B, G, gamma, beta, stats, gsum, dgam, dbet, ga.stride)
        torch.cuda.synchronize()
        return [(dx, ga.value(2 * k), ga.value(2 * k + 1)) for k, dx in enumerate(outs)]

    want, got = run(False), run(True)
    assert ops.lib.kd6d_barrier_timeouts() == 0
    for (dx_w, dg_w, db_w), (dx_g, dg_g, db_g) in zip(want, got):
        # same kernel body, same row chunks, fixed-point group sums: the pair launch reproduces the two launches bit for bit
        assert torch.equal(dx_g, dx_w)
        assert torch.equal(dg_g, dg_w) and torch.equal(db_g, db_w)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pool_upsample_eltwise(gpu_device, dtype):
    ops = _ops()
    dev = gpu_device
    B, C, H, W = 2, 16, 8, 6
    g = torch.Generator().manual_seed(5)
    x = round_to(torch.randn(B, C, H, W, generator=g), dtype)
    x[0, :, 0:2, 0:2] = 1.0   # ties: gradient must go to the first max (torch order)
    dy = round_to(torch.randn(B, C, H // 2, W // 2, generator=g), dtype)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    yr.backward(dy)
    xp = pack_levels([x], dtype).to(dev)
    y = torch.empty(B * (H // 2) * (W // 2), C, dtype=dtype, device=dev)
    ops.maxpool2_fwd(xp, y, B, H, W)
    dx = torch.empty_like(xp)
    ops.maxpool2_bwd(xp, pack_levels([dy], dtype).to(dev), dx, B, H, W)
    torch.cuda.synchronize()
    assert torch.equal(unpack_levels(y.cpu(), B, [(H // 2, W // 2)])[0], yr.detach())
    assert torch.equal(unpack_levels(dx.cpu(), B, [(H, W)])[0], xr.grad)
    # upsample + add and its adjoint
    coarse = round_to(torch.randn(B, C, H // 2, W // 2, generator=g), dtype)
    out = torch.empty_like(xp)
    ops.upsample2_add(xp, pack_levels([coarse], dtype).to(dev), out, B, H, W)
    ref = x + F.interpolate(coarse, scale_factor=2, mode="nearest")
    dco = torch.empty(B * (H // 2) * (W // 2), C, dtype=dtype, device=dev)
    ops.sumpool2(xp, dco, B, H, W)
    torch.cuda.synchronize()
    tol = _tol(dtype, stored=True)
    torch.testing.assert_close(unpack_levels(out.cpu(), B, [(H, W)])[0], ref, **tol)
    torch.testing.assert_close(unpack_levels(dco.cpu(), B, [(H // 2, W // 2)])[0],
                               F.avg_pool2d(x, 2) * 4, **tol)
    # relu / relu-bwd / add
    r = torch.empty_like(xp)
    ops.eltwise(ops.ELT_RELU, xp, None, r)
    rb = torch.empty_like(xp)
    ops.eltwise(ops.ELT_RELU_BWD, xp, out, rb)
    torch.cuda.synchronize()
    assert torch.equal(r.cpu().float(), F.relu(xp.cpu().float()))
    assert torch.equal(rb.cpu().float(), torch.where(xp.cpu().float() > 0, out.cpu().float(), torch.zeros(())))
    img = torch.randn(2, 3, 5, 7, generator=g)
    nh = ops.image_to_nhwc(img.to(dev), dtype)
    torch.cuda.synchronize()
    got = nh.cpu().float().reshape(2, 5, 7, 8)
    torch.testing.assert_close(got[..., :3], round_to(img, dtype).permute(0, 2, 3, 1))
    assert float(got[..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize("reach", [0.5, None])
@pytest.mark.parametrize("lanes", ["1", "0"])
def test_sinkhorn_kernel_vs_oracle(gpu_device, reach, lanes):
    """fp32 kernel vs fp64 oracle: loss rtol 1e-4, grads rtol 2e-3.  lanes "1": sets of up to 16
    points run their four softmins side by side on the wave's four 16-lane rows (the 20 / 17-point images still take
    the general path); "0": the general path for every image."""
    ops = _ops()
    _option("sinkhorn.lanes", int(lanes))
    from oracle.sinkhorn_ref import kd_loss_images
    r = np.random.default_rng(0)
    counts_s = [10, 0, 9, 12, 1, 10, 20, 16, 3]
    counts_t = [10, 7, 0, 9, 3, 10, 11, 16, 17]
    P, M = sum(counts_s), sum(counts_t)
    # LINEMOD-like: clustered votes around 8 keypoints, normalised coordinates
    centres = r.uniform(0.3, 0.7, (8, 2))
    xs = (centres[None] + r.normal(0, 0.02, (P, 8, 2))).astype(np.float32)
    yt = (centres[None] + r.normal(0, 0.01, (M, 8, 2))).astype(np.float32)
    al = np.repeat(r.uniform(0.05, 0.95, (P, 1)), 8, 1).astype(np.float32)
    be = np.repeat(r.uniform(0.2, 0.99, (M, 1)), 8, 1).astype(np.float32)
    if reach is None:
        al /= al.sum(0, keepdims=True); be /= be.sum(0, keepdims=True)
    s_off = np.concatenate([[0], np.cumsum(counts_s)]).astype(np.int32)
    t_off = np.concatenate([[0], np.cumsum(counts_t)]).astype(np.int32)
    blur = 0.001 if reach else 0.01
    if reach is None:
        # balanced problems need equal masses per image
        for i in range(len(counts_s)):
            a = slice(s_off[i], s_off[i + 1]); b = slice(t_off[i], t_off[i + 1])
            if counts_s[i] and counts_t[i]:
                al[a] /= al[a].sum(0, keepdims=True); be[b] /= be[b].sum(0, keepdims=True)
    loss_r, valid_r, gx_r, ga_r = kd_loss_images(xs.astype(np.float64), al.astype(np.float64), s_off,
                                                 yt.astype(np.float64), be.astype(np.float64), t_off,
                                                 blur=blur, scaling=0.5, reach=reach)
    dev = gpu_device
    t = lambda a: torch.from_numpy(a).to(dev)
    loss, valid, gx, ga = ops.sinkhorn_div(t(xs), t(al), t(s_off[:-1].copy()), t(np.asarray(counts_s, np.int32)),
                                           t(yt), t(be), t(t_off[:-1].copy()), t(np.asarray(counts_t, np.int32)),
                                           len(counts_s), 2.0, blur, 0.5, reach)
    torch.cuda.synchronize()
    assert valid.cpu().tolist() == valid_r.tolist()
    np.testing.assert_allclose(loss.cpu().numpy(), loss_r, rtol=1e-4, atol=1e-7)
    gscale = np.abs(gx_r).max()
    np.testing.assert_allclose(gx.cpu().numpy(), gx_r, rtol=2e-3, atol=2e-3 * gscale)
    np.testing.assert_allclose(ga.cpu().numpy(), ga_r, rtol=2e-3, atol=2e-3 * np.abs(ga_r).max())


def _dense_problem(N, M, D, seed, reach):
    """ZebraPose-like dense local predictions: D-dim code probabilities sigmoid(N(0,2)) per cell, weighted by a
    segmentation score sigmoid(N(0,2)); a few cells carry zero weight (background)."""
    r = np.random.default_rng(seed)
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    x = sig(r.normal(0, 2, (N, D))).astype(np.float32)
    y = (sig(r.normal(0, 2, (M, D))) * 0.9 + 0.05).astype(np.float32)
    a = sig(r.normal(0, 2, N)).astype(np.float32)
    b = sig(r.normal(0, 2, M)).astype(np.float32)
    a[17::17] = 0.0
    b[13::13] = 0.0
    if reach is None:
        a /= a.sum(); b /= b.sum()
    return x, a, y, b


@pytest.mark.parametrize("case", [
    # (N, M, D, blur, reach)
    (300, 257, 16, 0.05, 0.5),        # ragged sizes (not multiples of the 64-column tile / 256-row workgroup)
    (130, 520, 16, 0.001, 0.5),       # the reference's blur: eps = 1e-6
    (200, 200, 2, 0.01, None),        # balanced
    (1000, 700, 8, 0.05, 0.5),
    (64, 1, 4, 0.05, 0.5),            # a single target point
])
def test_sinkhorn_dense_kernel_vs_oracle(gpu_device, case):
    """Dense (online-logsumexp) Sinkhorn kernel, fp32, vs the fp64 oracle: loss rtol 2e-4, gradients 5e-3 of
    their scale (the fp32-vs-fp64 spread of the algorithm itself is 2e-6 / 4e-4)."""
    ops = _ops()
    from oracle.sinkhorn_ref import sinkhorn_divergence
    N, M, D, blur, reach = case
    x, a, y, b = _dense_problem(N, M, D, N + M, reach)
    S_r, gx_r, ga_r = sinkhorn_divergence(a[None].astype(np.float64), x[None].astype(np.float64),
                                          b[None].astype(np.float64), y[None].astype(np.float64), blur=blur,
                                          scaling=0.5, reach=reach, with_grad=True)
    t = lambda v: torch.from_numpy(v).to(gpu_device)
    loss, gx, ga = ops.sinkhorn_dense(t(x), t(a), t(y), t(b), blur=blur, scaling=0.5, reach=reach)
    torch.cuda.synchronize()
    np.testing.assert_allclose(loss.cpu().numpy(), S_r, rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(gx.cpu().numpy(), gx_r[0], rtol=5e-3, atol=5e-3 * np.abs(gx_r).max())
    np.testing.assert_allclose(ga.cpu().numpy(), ga_r[0], rtol=5e-3, atol=5e-3 * np.abs(ga_r).max())
    # the caller-supplied diameter (geomloss' diameter= argument) gives the same schedule as the measured one
    from oracle.sinkhorn_ref import max_diameter
    loss2, _, _ = ops.sinkhorn_dense(t(x), t(a), t(y), t(b), blur=blur, scaling=0.5, reach=reach,
                                     diameter=max_diameter(x[None], y[None]))
    assert float(loss2) == pytest.approx(float(loss), rel=1e-5)
    # a second execution is bitwise equal: per-workgroup partials are added in block order, no float atomics
    loss3, gx3, ga3 = ops.sinkhorn_dense(t(x), t(a), t(y), t(b), blur=blur, scaling=0.5, reach=reach)
    assert torch.equal(loss3, loss) and torch.equal(gx3, gx) and torch.equal(ga3, ga)


def test_sinkhorn_dense_agrees_with_small_set_kernel(gpu_device):
    """The two independent HIP implementations (per-image 8-wave kernel, dense tiled kernel) agree on a
    problem both can run."""
    ops = _ops()
    N, M = 90, 70
    x, a, y, b = _dense_problem(N, M, 2, 5, 0.5)
    t = lambda v: torch.from_numpy(v).to(gpu_device)
    loss_d, gx_d, ga_d = ops.sinkhorn_dense(t(x), t(a), t(y), t(b), blur=0.001, scaling=0.5, reach=0.5)
    xs = np.repeat(x[:, None, :], 8, 1).copy(); ys = np.repeat(y[:, None, :], 8, 1).copy()
    al = np.repeat(a[:, None], 8, 1).copy(); be = np.repeat(b[:, None], 8, 1).copy()
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=gpu_device)
    loss_s, valid, gx_s, ga_s = ops.sinkhorn_div(t(xs), t(al), i32([0]), i32([N]), t(ys), t(be), i32([0]), i32([M]),
                                                 1, 2.0, 0.001, 0.5, 0.5)
    torch.cuda.synchronize()
    assert int(valid[0]) == 1
    assert float(loss_s[0]) == pytest.approx(8.0 * float(loss_d), rel=2e-4)
    np.testing.assert_allclose(gx_s[:, 0].cpu().numpy(), gx_d.cpu().numpy(), rtol=5e-3,
                               atol=5e-3 * float(gx_d.abs().max()))
    np.testing.assert_allclose(ga_s[:, 0].cpu().numpy(), ga_d.cpu().numpy(), rtol=2e-3,
                               atol=2e-3 * float(ga_d.abs().max()))


@pytest.mark.parametrize("blur", [0.05, 0.001])
def test_sinkhorn_dense_full_grid_properties(gpu_device, blur):
    """BASELINE config 5 at full size -- N = M = 128*128 cells, 16-D codes -- where no CPU oracle (nor the
    reference: geomloss would need KeOps) can go: size-independent properties of the divergence.
    S(a,a) = 0 with zero gradient, S(a,b) = S(b,a), S > 0 for different clouds, finite gradients."""
    ops = _ops()
    N = M = 128 * 128
    x, a, y, b = _dense_problem(N, M, 16, 1, 0.5)
    t = lambda v: torch.from_numpy(v).to(gpu_device)
    S_ab, gx, ga = ops.sinkhorn_dense(t(x), t(a), t(y), t(b), blur=blur, scaling=0.5, reach=0.5)
    S_ba, _, _ = ops.sinkhorn_dense(t(y), t(b), t(x), t(a), blur=blur, scaling=0.5, reach=0.5)
    S_aa, gx0, _ = ops.sinkhorn_dense(t(x), t(a), t(x), t(a), blur=blur, scaling=0.5, reach=0.5)
    torch.cuda.synchronize()
    s_ab, s_ba, s_aa = float(S_ab), float(S_ba), float(S_aa)
    assert np.isfinite([s_ab, s_ba, s_aa]).all() and bool(torch.isfinite(gx).all()) and bool(torch.isfinite(ga).all())
    assert s_ab > 0
    assert s_ba == pytest.approx(s_ab, rel=2e-4)
    assert abs(s_aa) <= 1e-4 * s_ab
    assert float(gx0.abs().max()) <= 1e-3 * float(gx.abs().max())


def test_fused_clip_adamw_matches_torch(gpu_device):
    """kd6d_sumsq + kd6d_clip_adamw vs clip_grad_norm_ + torch.optim.AdamW (train_libs.py:119 settings)
    over several steps on identical gradients, and vs the reference capture tests/golden/optim.npz."""
    import ctypes
    import os
    ops = _ops()
    lib, P = ops.lib, ops._ptr
    dev = gpu_device
    n = 10007
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, weight_decay=1e-4, eps=1e-8)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, 1e-3, 10100, pct_start=0.05, cycle_momentum=False,
                                              anneal_strategy="linear")
    p = p0.clone().to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    shadow = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 if step % 2 else 0.001)    # clipped and unclipped steps
        ref_p.grad = grad.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        lr = opt.param_groups[0]["lr"]
        opt.step(); sch.step()
        gd = grad.to(dev)
        ss = torch.zeros(1, device=dev)
        parts = torch.empty(128, device=dev)             # KD6D_SUMSQ_PARTS partial sums, added by kd6d_clip_adamw
        ops.check(lib.kd6d_sumsq(P(gd), n, P(parts), ops._stream()))
        if step % 2:       # host-scalar form
            ops.check(lib.kd6d_clip_adamw(P(p), P(gd), P(m), P(v), n, P(parts), P(ss), 1.0, lr, 0.9, 0.999, 1e-8, 1e-4, step,
                                          None, P(shadow), ops._stream()))
        else:              # device-resident schedule (the hipGraph replay form); host lr/step args are ignored
            hyper = torch.zeros(4, device=dev)
            ops.check(lib.kd6d_set_hyper(P(hyper), lr, 0.9, 0.999, step, ops._stream()))
            ops.check(lib.kd6d_clip_adamw(P(p), P(gd), P(m), P(v), n, P(parts), P(ss), 1.0, 123.0, 0.9, 0.999, 1e-8, 1e-4, 0,
                                          P(hyper), P(shadow), ops._stream()))
        torch.cuda.synchronize()
        assert float(ss) == pytest.approx(float((grad.double() ** 2).sum()), rel=1e-5)
        torch.testing.assert_close(p.cpu(), ref_p.detach(), rtol=1e-5, atol=1e-6)
        assert torch.equal(shadow.cpu(), p.cpu().to(torch.bfloat16))
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim.npz"))
    p = torch.linspace(-1, 1, 16).to(dev); m = torch.zeros(16, device=dev); v = torch.zeros(16, device=dev)
    g = torch.Generator().manual_seed(1)
    for it in range(3):
        gd = torch.randn(16, generator=g).to(dev)
        ops.check(lib.kd6d_clip_adamw(P(p), P(gd), P(m), P(v), 16, None, None, 0.0, float(z["lrs"][it]), 0.9, 0.999, 1e-8,
                                      1e-4, it + 1, None, None, ops._stream()))
        torch.cuda.synchronize()
        np.testing.assert_allclose(p.cpu().numpy(), z["params"][it], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("B,C,levels,n_wg", [
    (2, 128, [(32, 32), (16, 16), (8, 8), (4, 4)], 64),      # the student head's level set, short splits
    (16, 128, [(32, 32), (16, 16), (8, 8), (4, 4)], 256),    # ... at the benchmark batch, one workgroup per CU
    (3, 256, [(9, 7), (5, 3), (2, 1)], 40),                  # odd maps, two 128-channel chunks, ragged steps
])
def test_wgrad_group_matches_torch(gpu_device, B, C, levels, n_wg):
    """ops.WgradGroup: the weight (and bias) gradients of a tower layer, the 240-channel pose layer, the 16-channel
    cls layer and a 1x1 layer in ONE launch pair, vs torch's conv2d_weight per layer.  bf16-representable inputs,
    fp32 accumulation: only the summation order differs (2e-4 of the tensor scale).  The call accumulates (+=)."""
    ops = _ops()
    dev = gpu_device
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(B * 7 + C)
    group = ops.WgradGroup(n_wg)
    cases = []
    for cout, k, bias in ((C, 3, True), (240, 3, True), (16, 3, True), (C, 1, False), (24, 1, True)):
        geom = ops.Geom(B, C, cout, k, 1, k // 2, levels)
        assert ops.wgrad_group_supported(geom, dtype)
        xs = [round_to(torch.randn(B, C, h, w, generator=g), dtype) for (h, w) in levels]
        dys = [round_to(torch.randn(B, cout, h, w, generator=g), dtype) for (h, w) in levels]
        dw = torch.full((cout, k, k, C), 0.25, dtype=torch.float32, device=dev)
        db = torch.full((cout,), 0.5, dtype=torch.float32, device=dev) if bias else None
        xp, dyp = pack_levels(xs, dtype).to(dev), pack_levels(dys, dtype).to(dev)
        group.add(geom, xp, dyp, dw, db)
        cases.append((cout, k, xs, dys, dw, db))
    assert not ops.wgrad_group_supported(ops.Geom(B, C, C, 3, 2, 1, levels), dtype)          # stride 2
    assert not ops.wgrad_group_supported(ops.Geom(B, 64, C, 3, 1, 1, levels), dtype)         # cin % 128
    group.launch()
    assert len(group) == 0
    torch.cuda.synchronize()
    for cout, k, xs, dys, dw, db in cases:
        ref = torch.zeros(cout, C, k, k)
        ref_b = torch.zeros(cout, dtype=torch.float64)
        for x, dy in zip(xs, dys):
            ref += torch.nn.grad.conv2d_weight(x, (cout, C, k, k), dy, stride=1, padding=k // 2)
            ref_b += dy.double().sum(dim=(0, 2, 3))
        got = dw.cpu().permute(0, 3, 1, 2) - 0.25
        scale = max(float(ref.abs().max()), 1.0)
        torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4 * scale, msg=lambda m: "cout %d k %d: %s" % (cout, k, m))
        if db is not None:
            torch.testing.assert_close(db.cpu().double() - 0.5, ref_b, rtol=2e-4,
                                       atol=2e-4 * max(float(ref_b.abs().max()), 1.0))


def test_zero_regions_step_prologue(gpu_device):
    """kd6d_zero_regions: several regions of different dtypes and ragged sizes (a partial last 16-B granule, a
    4-byte region, an empty entry) are cleared by one launch, nothing beyond a region's end is touched, and the int64
    counters go up by one."""
    ops = _ops()
    dev = gpu_device
    sizes = [(torch.float32, 1 << 20), (torch.float32, 4099), (torch.bfloat16, 2 * 7 + 16 * 5), (torch.int32, 1),
             (torch.float32, 0), (torch.int32, 37)]
    guards, views = [], []
    for dt, n in sizes:
        big = torch.full((n + 64,), 3, dtype=dt, device=dev)      # 32 guard elements either side (16-B aligned start)
        guards.append(big)
        views.append(big[32:32 + n])
    counter = torch.arange(11, dtype=torch.int64, device=dev)
    ops.zero_many(views + [None], counter=counter)
    torch.cuda.synchronize()
    for (dt, n), big in zip(sizes, guards):
        assert float(big[32:32 + n].float().abs().sum()) == 0.0
        assert bool((big[:32] == 3).all()) and bool((big[32 + n:] == 3).all()), (dt, n)
    assert counter.tolist() == list(range(1, 12))
    ops.zero_many([], counter=counter)
    torch.cuda.synchronize()
    assert counter.tolist() == list(range(2, 13))
    with pytest.raises(RuntimeError, match="16-B aligned"):
        ops.zero_many([guards[0][1:9]])


def test_uniform_keys_counter_based(gpu_device):
    """kd6d_uniform_keys: values in [0, 1), uniform (mean, variance, decile counts), reproducible from (seed, counter),
    different for another counter or seed, no repeats between neighbouring cells."""
    ops = _ops()
    dev = gpu_device
    n = 1 << 18
    c3 = torch.tensor([3, 99], dtype=torch.int64, device=dev)
    c4 = torch.tensor([4], dtype=torch.int64, device=dev)
    a = ops.uniform_keys(torch.empty(n, device=dev), c3, 1234).clone()
    b = ops.uniform_keys(torch.empty(n, device=dev), c3, 1234)
    c = ops.uniform_keys(torch.empty(n, device=dev), c4, 1234)
    d = ops.uniform_keys(torch.empty(n, device=dev), c3, 1235)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert float(a.min()) >= 0.0 and float(a.max()) < 1.0
    assert abs(float(a.mean()) - 0.5) < 5e-3 and abs(float(a.var()) - 1.0 / 12.0) < 2e-3
    hist = torch.histc(a, bins=10, min=0.0, max=1.0)
    assert float((hist - n / 10).abs().max()) < 6 * (n / 10) ** 0.5
    for other in (c, d):
        assert float((a == other).float().mean()) < 1e-3
        assert abs(float(((a - 0.5) * (other - 0.5)).mean())) < 1e-3          # uncorrelated streams
    assert float((a[1:] == a[:-1]).float().mean()) < 1e-3


# ---------------------------------------------------------------------------------------------------------
# convolution + normalisation + activation as ONE launch (kd6d_conv2d_fwd_norm)
# ---------------------------------------------------------------------------------------------------------
FUSED_GN_CASES = [
    # (B, C, levels, halo-kernel option: -1 = the dispatcher's choice, 0 = generic / LDS-DMA kernels)
    (2, 128, [(32, 32), (16, 16), (8, 8), (4, 4)], -1),          # student towers, small batch: keys span several tiles
    (16, 128, [(32, 32), (16, 16), (8, 8), (4, 4)], -1),         # ... at the benchmark batch (128x128 twin tiles)
    (16, 256, [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)], -1),  # teacher towers: two channel tiles, 2x2 level
    (3, 128, [(15, 20), (8, 10), (4, 5)], -1),                   # odd maps (full-frame levels)
    (2, 256, [(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], -1),  # maps 80 wide: not the halo kernel
    (3, 128, [(12, 12), (6, 6), (3, 3)], 0),                     # halo kernel off
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", FUSED_GN_CASES, ids=[str(i) for i in range(len(FUSED_GN_CASES))])
def test_conv_fwd_norm_group(gpu_device, dtype, case):
    """models/model.py:395-417,438-451: tower Conv2d(3x3, bias) -> GroupNorm(32) -> ReLU over all pyramid levels, one
    launch (window barrier per (level, image) in the conv epilogue).  Against torch on the CPU, and against the two-launch
    path (kd6d_conv2d_fwd with fused statistics + kd6d_gn_relu_fwd): the same statistics up to atomic order."""
    ops = _ops()
    dev = gpu_device
    B, C, levels, halo = case
    if dtype == torch.float32 and B * sum(h * w for h, w in levels) > 8000:
        pytest.skip("fp32 mode runs the exact-fp32 MFMA chain: covered by the small cases")
    _option("conv.halo", halo)
    G = 32
    gen = torch.Generator().manual_seed(C + B)
    geom = ops.Geom(B, C, C, 3, 1, 1, levels)
    xs = [round_to(torch.randn(B, C, h, w, generator=gen), dtype) for (h, w) in levels]
    w = round_to(torch.randn(C, C, 3, 3, generator=gen) / (C * 9) ** 0.5, dtype)
    bias = torch.randn(C, generator=gen) * 0.1
    gamma = torch.rand(C, generator=gen) + 0.5
    beta = torch.randn(C, generator=gen) * 0.2
    # a level whose rows do not fall into whole 16-row fragments of one image (H*W or its first row not a multiple of 16)
    # gets its statistics from a follow-up launch behind the convolution: nothing to wait for inside the kernel, so such
    # geometries do not take the fused launch (round 4) -- the two-launch path below is then checked against torch alone
    row, whole = 0, True
    for (h, w_) in levels:
        whole = whole and (h * w_) % 16 == 0 and row % 16 == 0
        row += B * h * w_
    assert bool(ops.conv_norm_fusable(geom, dtype, ops.NORM_GROUP, G)) == whole
    xp = pack_levels(xs, dtype).to(dev)
    wk = w_to_krsc(w, dtype).to(dev)
    raws, ys = [], []
    for x in xs:
        raw = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
        raws.append(raw)
        ys.append(F.relu(F.group_norm(raw, G, gamma.double(), beta.double(), 1e-5)))
    y = torch.empty(geom.rows_out, C, dtype=dtype, device=dev)
    raw_out = torch.empty(geom.rows_out, C, dtype=torch.float32, device=dev)
    stats = None
    for with_raw in ((True, False) if whole else ()):          # eval-mode callers do not store the pre-normalisation tensor
        stats = torch.zeros(ops.conv_norm_stats_floats(geom, ops.NORM_GROUP, G), device=dev)
        ctr = torch.zeros(ops.conv_norm_counter_words(geom, ops.NORM_GROUP), dtype=torch.int32, device=dev)
        y.fill_(7.0)
        ops.conv2d_fwd_norm(geom, xp, wk, y, ops.NORM_GROUP, gamma.to(dev), beta.to(dev), stats, ctr, ops.ACT_RELU,
                            raw_out=raw_out if with_raw else None, bias=bias.to(dev), groups=G)
        torch.cuda.synchronize()
        assert ops.lib.kd6d_barrier_timeouts() == 0
        for gl, ref in zip(unpack_levels(y.float().cpu(), B, levels), ys):
            torch.testing.assert_close(gl.double(), ref, **_tol(dtype, stored=True))
    if whole:
        for gl, ref in zip(unpack_levels(raw_out.cpu(), B, levels), raws):
            torch.testing.assert_close(gl.double(), ref, rtol=2e-4, atol=2e-4)
    # the two-launch path on the same inputs
    n_st = len(levels) * B * G * 2
    stats2 = _accs(n_st, dev)
    raw2 = ops.conv2d_fwd(geom, xp, wk, ch_shift=bias.to(dev), out_f32=True, stats=stats2, stats_groups=G)
    y2 = torch.empty_like(y)
    ops.gn_relu_fwd(raw2, y2, [h * w_ for (h, w_) in levels], B, G, gamma.to(dev), beta.to(dev), 1e-5, stats2,
                    flags=ops.GN_STATS_READY)
    torch.cuda.synchronize()
    for gl, ref in zip(unpack_levels(y2.float().cpu(), B, levels), ys):
        torch.testing.assert_close(gl.double(), ref, **_tol(dtype, stored=True))
    if not whole:
        return
    torch.testing.assert_close(_acc_val(stats, n_st).cpu(), _acc_val(stats2, n_st).cpu(), rtol=1e-6, atol=1e-5)   # what kd6d_gn_relu_bwd reads
    d = (y.float() - y2.float()).abs()
    ulp = y2.float().abs() * 2.0 ** -7 + 1e-6 if dtype == torch.bfloat16 else y2.float().abs() * 1e-5 + 1e-5
    assert bool((d <= ulp).all()), float(d.max())
    if dtype == torch.bfloat16:
        assert float((d > 0).float().mean()) < 1e-2      # a few elements one rounding step apart


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # (B, Cin, Cout, k, (H, W)): the student's non-pooled ConvBlocks of stages 3-5 (tiny_h / tiny) + odd sizes
    (16, 64, 16, 1, (32, 32)), (16, 128, 32, 1, (16, 16)), (16, 32, 256, 3, (16, 16)), (16, 256, 64, 1, (16, 16)),
    (16, 64, 512, 3, (16, 16)), (2, 16, 128, 3, (15, 20)), (3, 128, 16, 1, (7, 9)),
])
def test_conv_fwd_norm_batch(gpu_device, dtype, case):
    """backbone/common.py:316-324 in train mode: Conv2d(no bias) -> BatchNorm2d (batch statistics, running-stat update)
    -> LeakyReLU(0.1) as one launch (grid barrier in the conv epilogue), against torch on the CPU."""
    ops = _ops()
    dev = gpu_device
    B, Cin, Cout, k, (H, W) = case
    if dtype == torch.float32 and B * H * W * Cout * Cin * k * k > 3e9:
        pytest.skip("fp32 mode: covered by the smaller cases")
    gen = torch.Generator().manual_seed(Cin * 7 + Cout)
    geom = ops.Geom(B, Cin, Cout, k, 1, k // 2, [(H, W)])
    if not ops.conv_norm_fusable(geom, dtype, ops.NORM_BATCH):
        pytest.skip("not a fused geometry for this precision (the engine then runs conv + bn_train_fwd)")
    x = round_to(torch.randn(B, Cin, H, W, generator=gen), dtype)
    w = round_to(torch.randn(Cout, Cin, k, k, generator=gen) / (Cin * k * k) ** 0.5, dtype)
    gamma = torch.rand(Cout, generator=gen) + 0.5
    beta = torch.randn(Cout, generator=gen) * 0.2
    rm0 = torch.randn(Cout, generator=gen) * 0.1
    rv0 = torch.rand(Cout, generator=gen) + 0.5
    raw = F.conv2d(x.double(), w.double(), padding=k // 2)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    yr = F.leaky_relu(F.batch_norm(raw, rm, rv, gamma.double(), beta.double(), True, 0.1, 1e-5), 0.1)
    xp = pack_levels([x], dtype).to(dev)
    y = torch.empty(geom.rows_out, Cout, dtype=dtype, device=dev)
    raw_out = torch.empty(geom.rows_out, Cout, dtype=torch.float32, device=dev)
    stats = torch.zeros(ops.conv_norm_stats_floats(geom, ops.NORM_BATCH), device=dev)
    ctr = torch.zeros(ops.conv_norm_counter_words(geom, ops.NORM_BATCH), dtype=torch.int32, device=dev)
    rmd, rvd = rm0.to(dev), rv0.to(dev)
    mean = torch.empty(Cout, device=dev); invstd = torch.empty(Cout, device=dev)
    ops.conv2d_fwd_norm(geom, xp, w_to_krsc(w, dtype).to(dev), y, ops.NORM_BATCH, gamma.to(dev), beta.to(dev), stats, ctr,
                        ops.ACT_LEAKY, raw_out=raw_out, running_mean=rmd, running_var=rvd, save_mean=mean, save_invstd=invstd)
    torch.cuda.synchronize()
    assert ops.lib.kd6d_barrier_timeouts() == 0
    torch.testing.assert_close(unpack_levels(raw_out.cpu(), B, [(H, W)])[0].double(), raw, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(unpack_levels(y.float().cpu(), B, [(H, W)])[0].double(), yr, **_tol(dtype, stored=True))
    m_ref = raw.mean((0, 2, 3)); v_ref = raw.var((0, 2, 3), unbiased=False)
    torch.testing.assert_close(mean.cpu().double(), m_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(invstd.cpu().double(), torch.rsqrt(v_ref + 1e-5), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(rmd.cpu().double(), rm, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(rvd.cpu().double(), rv, rtol=1e-3, atol=1e-4)


def test_conv_fwd_norm_under_repetition(gpu_device):
    """300 back-to-back launches of each fused form with fresh counters, the GroupNorm one on two streams at once
    (a window-barrier launch beside a grid-barrier launch is what the step runs): no wait may give up and every
    launch must give the first one's result."""
    ops = _ops()
    dev = gpu_device
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(3)
    B = 16
    levels = [(32, 32), (16, 16), (8, 8), (4, 4)]
    gg = ops.Geom(B, 128, 128, 3, 1, 1, levels)
    gb = ops.Geom(B, 32, 256, 3, 1, 1, [(16, 16)])
    xg = torch.randn(gg.rows_in, 128, generator=gen).to(dtype).to(dev)
    wg = (torch.randn(128 * 9 * 128, generator=gen) / 34.0).to(dtype).to(dev)
    xb = torch.randn(gb.rows_in, 32, generator=gen).to(dtype).to(dev)
    wb = (torch.randn(256 * 9 * 32, generator=gen) / 17.0).to(dtype).to(dev)
    ones = {c: torch.ones(c, device=dev) for c in (128, 256)}
    zeros = {c: torch.zeros(c, device=dev) for c in (128, 256)}
    n = 300
    sg = torch.zeros(n, ops.conv_norm_stats_floats(gg, ops.NORM_GROUP, 32), device=dev)
    cg = torch.zeros(n, ops.conv_norm_counter_words(gg, ops.NORM_GROUP), dtype=torch.int32, device=dev)
    sb = torch.zeros(n, ops.conv_norm_stats_floats(gb, ops.NORM_BATCH), device=dev)
    cb = torch.zeros(n, ops.conv_norm_counter_words(gb, ops.NORM_BATCH), dtype=torch.int32, device=dev)
    yg = [torch.empty(gg.rows_out, 128, dtype=dtype, device=dev) for _ in range(2)]
    yb = [torch.empty(gb.rows_out, 256, dtype=dtype, device=dev) for _ in range(2)]
    mean = torch.empty(256, device=dev); invstd = torch.empty(256, device=dev)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    bad = 0
    for i in range(n):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.conv2d_fwd_norm(gg, xg, wg, yg[min(i, 1)], ops.NORM_GROUP, ones[128], zeros[128], sg[i], cg[i], ops.ACT_RELU,
                                groups=32)
        ops.conv2d_fwd_norm(gb, xb, wb, yb[min(i, 1)], ops.NORM_BATCH, ones[256], zeros[256], sb[i], cb[i], ops.ACT_LEAKY,
                            save_mean=mean, save_invstd=invstd)
        torch.cuda.current_stream().wait_stream(side)
        if i >= 1 and i % 50 == 0:
            torch.cuda.synchronize()
            for a, b_ in ((yg[0], yg[1]), (yb[0], yb[1])):
                d = (a.float() - b_.float()).abs()
                bad += int((d > a.float().abs() * 2.0 ** -6 + 1e-3).sum())
    torch.cuda.synchronize()
    assert ops.lib.kd6d_barrier_timeouts() == 0
    assert bad == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # (B, C0, C1, k1, C2, k2, (H, W)): block A = conv(C0 -> C1, k1), block B = conv(C1 -> C2, k2): the in-stage
    # transitions of darknet_tiny_h / darknet_tiny stages 3-5 and odd sizes
    (16, 16, 8, 1, 64, 3, (64, 64)), (16, 8, 64, 3, 8, 1, (64, 64)), (16, 64, 16, 1, 128, 3, (32, 32)),
    (16, 16, 128, 3, 16, 1, (32, 32)), (16, 128, 32, 1, 256, 3, (16, 16)), (16, 32, 256, 3, 64, 1, (16, 16)),
    (4, 32, 256, 3, 32, 1, (15, 20)), (3, 64, 512, 3, 64, 1, (7, 9)),
])
def test_conv_block_bn_on_load(gpu_device, dtype, case):
    """kd6d_conv2d_fwd_block: block A's train-mode BatchNorm + LeakyReLU (backbone/common.py:316-324) applied by block
    B's convolution while it loads A's fp32 conv output.  Against the separate launches (kd6d_bn_train_fwd on the same
    batch sums, then kd6d_conv2d_fwd): the activation B writes on the way must be what bn_train_fwd stores, B's conv
    output and batch sums what the convolution of that tensor gives, and A's save_mean / save_invstd / running statistics
    must be the ones the separate launch leaves."""
    ops = _ops()
    dev = gpu_device
    B, C0, C1, k1, C2, k2, (H, W) = case
    if dtype == torch.float32 and B * H * W > 20000:
        pytest.skip("fp32 mode: covered by the smaller cases")
    gen = torch.Generator().manual_seed(C0 + 3 * C1 + 7 * C2)
    R = ops.BN_REPLICAS
    ga = ops.Geom(B, C0, C1, k1, 1, k1 // 2, [(H, W)])
    gb = ops.Geom(B, C1, C2, k2, 1, k2 // 2, [(H, W)])
    x = pack_levels([round_to(torch.randn(B, C0, H, W, generator=gen), dtype)], dtype).to(dev)
    wa = w_to_krsc(round_to(torch.randn(C1, C0, k1, k1, generator=gen) / (C0 * k1 * k1) ** 0.5, dtype), dtype).to(dev)
    wb = w_to_krsc(round_to(torch.randn(C2, C1, k2, k2, generator=gen) / (C1 * k2 * k2) ** 0.5, dtype), dtype).to(dev)
    gamma = (torch.rand(C1, generator=gen) + 0.5).to(dev)
    beta = (torch.randn(C1, generator=gen) * 0.2).to(dev)
    rows = ga.rows_out
    raw_a = torch.empty(rows, C1, device=dev)
    sums_a = _accs(R * 2 * C1, dev)                      # R replica rows of {sum[C1], sumsq[C1]} accumulators
    ops.conv2d_fwd_block(ga, x, wa, raw_a, stats=sums_a, stats_replicas=R)
    torch.cuda.synchronize()
    ref_sum = raw_a.double().sum(0).cpu(); ref_sq = (raw_a.double() ** 2).sum(0).cpu()
    tot = _acc_val(sums_a, R * 2 * C1).view(R, 2, C1).double().sum(0).cpu()
    torch.testing.assert_close(tot[0], ref_sum, rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(tot[1], ref_sq, rtol=1e-4, atol=1e-2)
    # separate launches on the summed replica rows: the accumulators' int64 words add exactly
    plain = sums_a.view(torch.int64).view(R, 2, C1, 2).sum(0).contiguous().view(torch.float32).view(2, C1 * 4)
    rm1, rv1 = torch.zeros(C1, device=dev), torch.ones(C1, device=dev)
    m1, is1 = torch.empty(C1, device=dev), torch.empty(C1, device=dev)
    z_ref = torch.empty(rows, C1, dtype=dtype, device=dev)
    ops.bn_train_fwd(raw_a, z_ref, plain[0], plain[1], gamma, beta, 1e-5, 0.1, rm1, rv1, m1, is1, ops.ACT_LEAKY)
    sums_ref = _accs(2 * C2, dev)
    raw_ref = ops.conv2d_fwd(gb, z_ref, wb, out_f32=True, stats=sums_ref, stats_groups=0)
    # the one launch
    rm2, rv2 = torch.zeros(C1, device=dev), torch.ones(C1, device=dev)
    m2, is2 = torch.empty(C1, device=dev), torch.empty(C1, device=dev)
    z = torch.full((rows, C1), 7.0, dtype=dtype, device=dev)
    raw_b = torch.empty(gb.rows_out, C2, device=dev)
    sums_b = _accs(R * 2 * C2, dev)
    ops.conv2d_fwd_block(gb, raw_a, wb, raw_b, stats=sums_b, stats_replicas=R, z_out=z,
                         bn_in=dict(sums=sums_a, replicas=R, gamma=gamma, beta=beta, act=ops.ACT_LEAKY, eps=1e-5,
                                    momentum=0.1, running_mean=rm2, running_var=rv2, save_mean=m2, save_invstd=is2))
    torch.cuda.synchronize()
    for a, b_ in ((m2, m1), (is2, is1), (rm2, rm1), (rv2, rv1)):
        torch.testing.assert_close(a, b_, rtol=1e-5, atol=1e-6)
    d = (z.float() - z_ref.float()).abs()
    ulp = z_ref.float().abs() * (2.0 ** -7 if dtype == torch.bfloat16 else 1e-5) + 1e-6
    assert bool((d <= ulp).all()), float(d.max())               # same statistics (integer sums); fma contraction may differ
    if dtype == torch.bfloat16:
        assert float((d > 0).float().mean()) < 5e-2             # a few values one rounding step apart
    torch.testing.assert_close(raw_b, raw_ref, rtol=5e-3, atol=5e-3)       # a few inputs one bf16 rounding step apart
    torch.testing.assert_close(_acc_val(sums_b, R * 2 * C2).view(R, 2 * C2).sum(0), _acc_val(sums_ref, 2 * C2),
                               rtol=1e-3, atol=0.5)


@pytest.mark.parametrize("blur", [0.05, 0.001])
@pytest.mark.parametrize("mode", [0, 1, 3])
def test_sinkhorn_dense_matrix_pipe_softmin(gpu_device, blur, mode):
    """D = 16 (BASELINE config 5's code dimension): the gradient-free softmin passes on the fp32 matrix pipe
    (dense_softmin_mfma_kernel: r.c from v_mfma_f32_32x32x2_f32, |r|^2 + |c|^2 - 2 r.c form on centred points, taken
    while eps >= 1.5e-4 diameter^2) against the fp64 oracle at the same tolerances as the difference-form kernel
    (option sinkhorn.dense_mfma = 0), ragged sizes included."""
    ops = _ops()
    from oracle.sinkhorn_ref import sinkhorn_divergence
    _option("sinkhorn.dense_mfma", mode)
    N, M, D, reach = 900, 777, 16, 0.5
    x, a, y, b = _dense_problem(N, M, D, 11, reach)
    S_r, gx_r, ga_r = sinkhorn_divergence(a[None].astype(np.float64), x[None].astype(np.float64),
                                          b[None].astype(np.float64), y[None].astype(np.float64), blur=blur,
                                          scaling=0.5, reach=reach, with_grad=True)
    t = lambda v: torch.from_numpy(v).to(gpu_device)
    loss, gx, ga = ops.sinkhorn_dense(t(x), t(a), t(y), t(b), blur=blur, scaling=0.5, reach=reach)
    torch.cuda.synchronize()
    err = abs(float(loss) - float(S_r[0])) / abs(float(S_r[0]))
    egx = float(np.abs(gx.cpu().numpy() - gx_r[0]).max() / np.abs(gx_r).max())
    ega = float(np.abs(ga.cpu().numpy() - ga_r[0]).max() / np.abs(ga_r).max())
    print("[dense OT D=16 blur %g mfma=%d] loss rel err %.2e, grad_x %.2e, grad_alpha %.2e of scale" % (blur, mode, err, egx, ega))
    np.testing.assert_allclose(loss.cpu().numpy(), S_r, rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(gx.cpu().numpy(), gx_r[0], rtol=5e-3, atol=5e-3 * np.abs(gx_r).max())
    np.testing.assert_allclose(ga.cpu().numpy(), ga_r[0], rtol=5e-3, atol=5e-3 * np.abs(ga_r).max())


@pytest.mark.parametrize("N,M", [(900, 777), (64, 130), (1000, 333), (129, 64)])
def test_sinkhorn_dense_gradient_as_second_product(gpu_device, N, M):
    """The two gradient-carrying softmins of the last extrapolation on the matrix pipe (dense_softmin_mfma_grad_kernel: the
    softmax-weighted sums as a second product whose B operand is the first product's accumulators, fp16 pieces hi +
    lo / 2^11 on both sides) against the difference form on the SAME inputs (option 3 = the same passes, gradient form on
    the VALU): values to 1e-6, gradients to 2e-4 of scale -- and not bitwise equal, i.e. the kernel under test did run.
    Ragged row counts (not multiples of 64) and column counts (not multiples of 128: the padded A-operand tiles)."""
    ops = _ops()
    D, reach, blur = 16, 0.5, 0.05
    x, a, y, b = _dense_problem(N, M, D, 23, reach)
    t = lambda v: torch.from_numpy(v).to(gpu_device)
    res = {}
    for mode in (1, 3):
        _option("sinkhorn.dense_mfma", mode)
        loss, gx, ga = ops.sinkhorn_dense(t(x), t(a), t(y), t(b), blur=blur, scaling=0.5, reach=reach)
        torch.cuda.synchronize()
        res[mode] = (float(loss), gx.cpu().numpy(), ga.cpu().numpy())
    assert np.isfinite(res[1][1]).all() and np.isfinite(res[1][2]).all()
    assert res[1][0] == pytest.approx(res[3][0], rel=2e-6)
    sg, sa = np.abs(res[3][1]).max(), np.abs(res[3][2]).max()
    egx = float(np.abs(res[1][1] - res[3][1]).max() / sg)
    ega = float(np.abs(res[1][2] - res[3][2]).max() / sa)
    print("[dense OT second product %dx%d] grad_x %.2e, grad_alpha %.2e of scale" % (N, M, egx, ega))
    assert egx <= 2e-4 and ega <= 2e-5, (egx, ega)
    assert not np.array_equal(res[1][1], res[3][1]), "option 1 did not take the matrix-pipe gradient kernel"


@pytest.mark.parametrize("blur", [0.001, 0.01])
@pytest.mark.parametrize("N,M", [(900, 777), (64, 130), (1000, 333), (129, 64)])
def test_sinkhorn_dense_screened_small_epsilon_passes(gpu_device, N, M, blur):
    """Passes with eps below the matrix-pipe rule (dense_softmin_screen_kernel): approximate exponents from the matrix
    pipe decide WHICH pairs can contribute (within 40 + the error bound of the row's running maximum), those pairs are
    evaluated exactly in the difference form.  Against the difference form over ALL pairs on the same inputs (option
    sinkhorn.dense_screen = 0): the neglected terms are < 2^-40 each, so the loss agrees to fp32 rounding and the
    gradients to 1e-4 of their scale; ragged row / column counts; gradient-free and gradient-carrying passes."""
    ops = _ops()
    D, reach = 16, 0.5
    x, a, y, b = _dense_problem(N, M, D, 31, reach)
    t = lambda v: torch.from_numpy(v).to(gpu_device)
    res = {}
    for screen in (1, 0):
        _option("sinkhorn.dense_screen", screen)
        loss, gx, ga = ops.sinkhorn_dense(t(x), t(a), t(y), t(b), blur=blur, scaling=0.5, reach=reach)
        torch.cuda.synchronize()
        res[screen] = (float(loss), gx.cpu().numpy(), ga.cpu().numpy())
    assert np.isfinite(res[1][0]) and np.isfinite(res[1][1]).all() and np.isfinite(res[1][2]).all()
    sg, sa = np.abs(res[0][1]).max(), np.abs(res[0][2]).max()
    el = abs(res[1][0] - res[0][0]) / abs(res[0][0])
    egx = float(np.abs(res[1][1] - res[0][1]).max() / sg)
    ega = float(np.abs(res[1][2] - res[0][2]).max() / sa)
    print("[dense OT screened %dx%d blur %g] loss %.2e, grad_x %.2e, grad_alpha %.2e of scale" % (N, M, blur, el, egx, ega))
    # at eps = blur^2 = 1e-6 the softmax weights of near-tied columns move by ~10 % under the fp32 rounding of the potentials
    # alone (1e-7 / eps = 0.1 in the exponent): two correct fp32 evaluations that add in another order differ by ~5e-4 of the
    # gradient scale there (the difference form itself is 7e-4 from the fp64 oracle); at blur 0.01 that effect is 100x smaller
    assert el <= 5e-6 and egx <= (2e-3 if blur < 0.005 else 1e-4) and ega <= 2e-5, (el, egx, ega)


@pytest.mark.parametrize("blur", [0.05, 0.001])
@pytest.mark.parametrize("N,M", [(900, 777), (130, 64), (257, 1000)])
def test_sinkhorn_dense_rows_per_workgroup_do_not_change_the_result(gpu_device, N, M, blur):
    """The matrix-pipe softmins with 64 or 128 rows per workgroup (option sinkhorn.dense_rows; 128 halves what a launch
    pulls from L2 and is taken from 8192 rows up): a row's arithmetic does not depend on which workgroup holds it, so the
    two give BITWISE the same loss and gradients -- matrix-pipe passes, screened passes and the gradient-carrying ones."""
    ops = _ops()
    x, a, y, b = _dense_problem(N, M, 16, 41, 0.5)
    t = lambda v: torch.from_numpy(v).to(gpu_device)
    res = {}
    for rows in (64, 128):
        _option("sinkhorn.dense_rows", rows)
        res[rows] = ops.sinkhorn_dense(t(x), t(a), t(y), t(b), blur=blur, scaling=0.5, reach=0.5)
        torch.cuda.synchronize()
    for u, v in zip(res[64], res[128]):
        assert bool(torch.isfinite(u).all()) and torch.equal(u, v)


def test_context_isolates_options_pair_bracket_and_timeouts(gpu_device):
    """kd6d_ctx (include/kd6d.h): options, the pair bracket and the barrier-timeout counter belong to a context; entry
    points act on the calling thread's current one.  A second context must not see the first one's options or open
    bracket, a launch issued under it counts its barrier give-ups in its own word, and the default context is what a
    host that never creates one gets."""
    import ctypes
    ops = _ops()
    lib = ops.lib
    dev = gpu_device
    h = ctypes.c_void_p()
    ops.check(lib.kd6d_ctx_create(ctypes.byref(h)), "kd6d_ctx_create")
    default = lib.kd6d_ctx_current()
    try:
        assert ops.get_option("conv.halo") == -1
        ops.check(lib.kd6d_ctx_set_option(h, b"conv.halo", 0))
        assert ops.get_option("conv.halo") == -1                       # the current (default) context is untouched
        v = ctypes.c_longlong(7)
        ops.check(lib.kd6d_ctx_get_option(h, b"conv.halo", ctypes.byref(v)))
        assert v.value == 0
        ops.check(lib.kd6d_conv2d_pair_begin())                        # bracket open in the default context ...
        assert lib.kd6d_ctx_conv2d_pair_pending(h) == 0
        ops.check(lib.kd6d_ctx_conv2d_pair_begin(h))                   # ... does not block one in the other
        ops.check(lib.kd6d_ctx_conv2d_pair_end(h))
        ops.check(lib.kd6d_conv2d_pair_end())
        # a convolution under the new context takes ITS options (halo kernel off -> the generic kernels), same numbers
        B, C, levels = 2, 64, [(16, 16)]
        gen = torch.Generator().manual_seed(5)
        geom = ops.Geom(B, C, C, 3, 1, 1, levels)
        x = torch.randn(geom.rows_in, C, generator=gen).to(torch.bfloat16).to(dev)
        w = (torch.randn(C * 9 * C, generator=gen) / 24.0).to(torch.bfloat16).to(dev)
        y0 = ops.conv2d_fwd(geom, x, w, out_f32=True)
        ops.check(lib.kd6d_ctx_make_current(h))
        assert ops.get_option("conv.halo") == 0 and lib.kd6d_ctx_current() == h.value
        y1 = ops.conv2d_fwd(geom, x, w, out_f32=True)
        # a barrier kernel under the new context: its counter is its own word
        gg = ops.Geom(B, 128, 128, 3, 1, 1, [(8, 8), (4, 4)])
        xg = torch.randn(gg.rows_in, 128, generator=gen).to(torch.bfloat16).to(dev)
        wg = (torch.randn(128 * 9 * 128, generator=gen) / 34.0).to(torch.bfloat16).to(dev)
        yg = torch.empty(gg.rows_out, 128, dtype=torch.bfloat16, device=dev)
        st = torch.zeros(ops.conv_norm_stats_floats(gg, ops.NORM_GROUP, 32), device=dev)
        ct = torch.zeros(ops.conv_norm_counter_words(gg, ops.NORM_GROUP), dtype=torch.int32, device=dev)
        ops.conv2d_fwd_norm(gg, xg, wg, yg, ops.NORM_GROUP, torch.ones(128, device=dev), torch.zeros(128, device=dev), st, ct,
                            ops.ACT_RELU, groups=32)
        torch.cuda.synchronize()
        assert lib.kd6d_barrier_timeouts() == 0 and lib.kd6d_ctx_barrier_timeouts(h) == 0
        ops.check(lib.kd6d_ctx_make_current(None))
        assert lib.kd6d_ctx_current() == default and ops.get_option("conv.halo") == -1
        torch.testing.assert_close(y1, y0, rtol=2e-4, atol=2e-4)
    finally:
        lib.kd6d_ctx_make_current(None)
        ops.check(lib.kd6d_ctx_destroy(h), "kd6d_ctx_destroy")
    assert lib.kd6d_ctx_destroy(ctypes.c_void_p(default)) != 0        # the default context cannot be destroyed
