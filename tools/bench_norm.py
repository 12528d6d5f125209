"""Micro-benchmark of the normalisation kernels on the student's layer shapes (run on the GPU box).

    python tools/bench_norm.py [--iters 50]

BatchNorm(train)+LeakyReLU: colstats, apply, backward reduce, backward apply on the fp32 pre-normalisation
tensor of every student ConvBlock; GroupNorm+ReLU forward / backward on the head-tower tensor.  Each launch is
captured `iters` times in one hipGraph (see tools/bench_conv.py) and the HBM bytes it must move are printed
beside the time, so the distance to the ~4 TB/s a streaming kernel reaches is visible per launch.
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "kd-6d-pose-adlp_amd"))
sys.path.insert(0, os.path.join(HERE, "tools"))
import torch  # noqa: E402

from kd6d import ops  # noqa: E402
from bench_conv import timeit_graph  # noqa: E402

B = 16
# (name, rows per image, channels) of the darknet_tiny_h ConvBlocks at 256x256 crops
BN_LAYERS = [("u1", 256 * 256, 8), ("u2", 128 * 128, 16), ("s3.1x1", 64 * 64, 8), ("s3.3x3", 64 * 64, 64),
             ("s4.1x1", 32 * 32, 16), ("s4.3x3", 32 * 32, 128), ("s5.1x1", 16 * 16, 32), ("s5.3x3", 16 * 16, 256),
             ("s5.last", 16 * 16, 64)]
GN_LEVELS = [32 * 32, 16 * 16, 8 * 8, 4 * 4]


def bn_scratch(c, R, dev):
    """{sum, sumsq} accumulators, mean, invstd, R replica rows of the backward's two accumulator sums (engine layout)."""
    A = ops.ACC_FLOATS
    s = torch.zeros((2 * A + 2 + 2 * A * R) * c, device=dev)
    o = (2 * A + 2) * c
    return dict(sum=s[0:A * c], sumsq=s[A * c:2 * A * c], mean=s[2 * A * c:(2 * A + 1) * c],
                invstd=s[(2 * A + 1) * c:o], w1=s[o:o + A * R * c], w2=s[o + A * R * c:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    print("| layer | kernel | rows | C | MB | us | TB/s |")
    print("|---|---|---|---|---|---|---|")

    def row(name, kern, rows, c, mb, us):
        print("| %s | %s | %d | %d | %.1f | %.1f | %.2f |" % (name, kern, rows, c, mb, us, mb / max(us, 1e-9)))

    for name, hw, c in BN_LAYERS:
        rows = B * hw
        x = torch.randn(rows, c, device=dev)
        dz = torch.randn(rows, c, device=dev).to(bf)
        y = torch.empty(rows, c, dtype=bf, device=dev)
        dx = torch.empty(rows, c, dtype=bf, device=dev)
        R = 8 if rows >= (1 << 14) else 1          # as kd6d/engine.py ConvBlock.bwd_replicas
        q = bn_scratch(c, R, dev)
        gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        ops.colstats(x, q['sum'], q['sumsq'])
        mb_x, mb_a = rows * c * 4 / 1e6, rows * c * 2 / 1e6
        row(name, "colstats", rows, c, mb_x, timeit_graph(lambda: ops.colstats(x, q['sum'], q['sumsq']), a.iters))
        row(name, "bn_apply_fwd", rows, c, mb_x + mb_a, timeit_graph(
            lambda: ops.bn_train_fwd(x, y, q['sum'], q['sumsq'], gamma, beta, 1e-5, 0.1, rm, rv, q['mean'],
                                     q['invstd'], 1), a.iters))
        c_ = ops.lib

        def reduce_only():
            ops.check(c_.kd6d_bn_train_bwd_reduce(ops.dt_code(bf), 1, ops._ptr(x), ops._ptr(dz), rows, c,
                                                  ops._ptr(q['mean']), ops._ptr(q['invstd']), ops._ptr(gamma),
                                                  ops._ptr(beta), 1, ops._ptr(q['w1']), ops._ptr(q['w2']),
                                                  R, ops._stream()), "reduce")

        def apply_only():
            ops.check(c_.kd6d_bn_train_bwd_apply(ops.dt_code(bf), 1, ops._ptr(x), ops._ptr(dz), ops._ptr(dx), rows, c,
                                                 ops._ptr(q['mean']), ops._ptr(q['invstd']), ops._ptr(gamma),
                                                 ops._ptr(beta), 1, ops._ptr(q['w1']), ops._ptr(q['w2']),
                                                 ops._ptr(dg), ops._ptr(db), R, ops._stream()), "apply")

        row(name, "bn_bwd_reduce", rows, c, mb_x + mb_a, timeit_graph(reduce_only, a.iters))
        row(name, "bn_bwd_apply", rows, c, mb_x + 2 * mb_a, timeit_graph(apply_only, a.iters))
        ctr = torch.zeros(32, dtype=torch.int32, device=dev)

        def onepass():      # the counter (and the sums) are NOT re-zeroed between the timed launches: the barrier is
            ctr.zero_()     # passed at once after the first -- so zero it inside the timed region (one small memset)
            ops.bn_train_bwd(x, dz, dx, q['mean'], q['invstd'], gamma, beta, 1, q['w1'],
                             q['w2'], dg, db, replicas=R, counter=ctr)

        def pair():
            ctr.zero_()
            ops.bn_train_bwd(x, dz, dx, q['mean'], q['invstd'], gamma, beta, 1, q['w1'],
                             q['w2'], dg, db, replicas=R, counter=None)

        row(name, "bn_bwd pair + memset", rows, c, 2 * mb_x + 3 * mb_a, timeit_graph(pair, a.iters))
        row(name, "bn_bwd one launch + memset", rows, c, mb_x + 2 * mb_a, timeit_graph(onepass, a.iters))

    # the four ConvBlocks that sit in front of a MaxPool2d(2,2): fused BN + act + pool against the separate launches
    for name, side, c in [("u1", 256, 8), ("u2", 128, 16), ("s3.last", 64, 64), ("s4.last", 32, 128)]:
        rows = B * side * side
        x = torch.randn(rows, c, device=dev)
        z = torch.empty(rows, c, dtype=bf, device=dev)
        dz = torch.empty(rows, c, dtype=bf, device=dev)
        yp = torch.empty(rows // 4, c, dtype=bf, device=dev)
        dyp = torch.randn(rows // 4, c, device=dev).to(bf)
        dx = torch.empty(rows, c, dtype=bf, device=dev)
        R = 8 if rows >= (1 << 14) else 1
        q = bn_scratch(c, R, dev)
        gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        ops.colstats(x, q['sum'], q['sumsq'])
        mean, invstd, w1, w2 = q['mean'], q['invstd'], q['w1'], q['w2']
        mb_x, mb_a = rows * c * 4 / 1e6, rows * c * 2 / 1e6

        def fwd_sep():
            ops.bn_train_fwd(x, z, q['sum'], q['sumsq'], gamma, beta, 1e-5, 0.1, rm, rv, mean, invstd, 1)
            ops.maxpool2_fwd(z, yp, B, side, side)

        def bwd_sep():
            ops.maxpool2_bwd(z, dyp, dz, B, side, side)
            ops.bn_train_bwd(x, dz, dx, mean, invstd, gamma, beta, 1, w1, w2, dg, db, replicas=R)

        row(name, "bn_apply + maxpool (2 launches)", rows, c, mb_x + 2.25 * mb_a, timeit_graph(fwd_sep, a.iters))
        row(name, "bn_pool_fwd (fused)", rows, c, mb_x + 0.25 * mb_a, timeit_graph(
            lambda: ops.bn_pool_train_fwd(x, yp, B, side, side, q['sum'], q['sumsq'], gamma, beta, 1e-5, 0.1, rm, rv,
                                          mean, invstd, 1), a.iters))
        row(name, "maxpool_bwd + bn_bwd (3 launches)", rows, c, 2 * mb_x + 5.25 * mb_a, timeit_graph(bwd_sep, a.iters))
        row(name, "bn_pool_bwd (fused, 2 launches)", rows, c, 2 * mb_x + 1.5 * mb_a, timeit_graph(
            lambda: ops.bn_pool_train_bwd(x, dyp, dx, B, side, side, mean, invstd, gamma, beta, 1, w1, w2, dg, db,
                                          replicas=R), a.iters))

    c, groups = 128, 32
    rows = B * sum(GN_LEVELS)
    x = torch.randn(rows, c, device=dev)
    dz = torch.randn(rows, c, device=dev).to(bf)
    y = torch.empty(rows, c, dtype=bf, device=dev)
    dx = torch.empty(rows, c, dtype=bf, device=dev)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    ga = ops.planar_acc(2 * c, dev)
    dg, db = ga[0:c], ga[c:2 * c]
    stats = ops.acc_zeros(len(GN_LEVELS) * B * groups * 2, dev)
    gsum = torch.zeros(ops.gn_bwd_workspace_floats(len(GN_LEVELS), B, groups), device=dev)
    ops.gn_relu_fwd(x, y, GN_LEVELS, B, groups, gamma, beta, 1e-5, stats)
    mb_x, mb_a = rows * c * 4 / 1e6, rows * c * 2 / 1e6
    row("head", "gn_fwd (stats ready)", rows, c, mb_x + mb_a, timeit_graph(
        lambda: ops.gn_relu_fwd(x, y, GN_LEVELS, B, groups, gamma, beta, 1e-5, stats, flags=ops.GN_STATS_READY), a.iters))
    row("head", "gn_bwd (reduce + apply)", rows, c, 2 * mb_x + 3 * mb_a, timeit_graph(
        lambda: ops.gn_relu_bwd(x, dz, dx, GN_LEVELS, B, groups, gamma, beta, stats, gsum, dg, db, 2 * c), a.iters))


if __name__ == "__main__":
    main()
