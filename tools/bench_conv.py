"""Micro-benchmark of the conv kernels on the shapes of the KD step (run on the GPU box).

    python tools/bench_conv.py [--kind fwd|dgrad|wgrad|all] [--set teacher|student|all] [--iters 50]

Each shape is captured `iters` times in one hipGraph and replayed between two HIP events, so the figure
is the steady-state kernel time incl. the ~1.5 us launch boundary, without host launch or event cost.
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "kd-6d-pose-adlp_amd"))
import torch  # noqa: E402

from kd6d import ops  # noqa: E402

B = 16
# (name, cin, cout, k, stride, [levels (h,w) of the INPUT])
TEACHER = [
    ("t.init", 8, 32, 3, 1, [(256, 256)]),
    ("t.s1.down", 32, 64, 3, 2, [(256, 256)]),
    ("t.s1.1x1", 64, 32, 1, 1, [(128, 128)]),
    ("t.s1.3x3", 32, 64, 3, 1, [(128, 128)]),
    ("t.s2.down", 64, 128, 3, 2, [(128, 128)]),
    ("t.s2.1x1", 128, 64, 1, 1, [(64, 64)]),
    ("t.s2.3x3", 64, 128, 3, 1, [(64, 64)]),
    ("t.s3.down", 128, 256, 3, 2, [(64, 64)]),
    ("t.s3.1x1", 256, 128, 1, 1, [(32, 32)]),
    ("t.s3.3x3", 128, 256, 3, 1, [(32, 32)]),
    ("t.s4.down", 256, 512, 3, 2, [(32, 32)]),
    ("t.s4.1x1", 512, 256, 1, 1, [(16, 16)]),
    ("t.s4.3x3", 256, 512, 3, 1, [(16, 16)]),
    ("t.s5.down", 512, 1024, 3, 2, [(16, 16)]),
    ("t.s5.1x1", 1024, 512, 1, 1, [(8, 8)]),
    ("t.s5.3x3", 512, 1024, 3, 1, [(8, 8)]),
    ("t.fpn.in5", 1024, 256, 1, 1, [(8, 8)]),
    ("t.fpn.out5", 256, 256, 3, 1, [(8, 8)]),
    ("t.fpn.out4", 256, 256, 3, 1, [(16, 16)]),
    ("t.fpn.out3", 256, 256, 3, 1, [(32, 32)]),
    ("t.fpn.p6", 1024, 256, 3, 2, [(8, 8)]),
    ("t.fpn.p7", 256, 256, 3, 2, [(4, 4)]),
    ("t.head.tower", 256, 256, 3, 1, [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]),
    ("t.head.cls", 256, 16, 3, 1, [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]),
    ("t.head.pose", 256, 240, 3, 1, [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]),
]
STUDENT = [
    ("s.u1", 8, 8, 3, 1, [(256, 256)]),
    ("s.u2", 8, 16, 3, 1, [(128, 128)]),
    ("s.s3.1x1", 16, 8, 1, 1, [(64, 64)]),
    ("s.s3.3x3", 8, 64, 3, 1, [(64, 64)]),
    ("s.s4.1x1", 64, 16, 1, 1, [(32, 32)]),
    ("s.s4.3x3", 16, 128, 3, 1, [(32, 32)]),
    ("s.s5.1x1", 128, 32, 1, 1, [(16, 16)]),
    ("s.s5.3x3", 32, 256, 3, 1, [(16, 16)]),
    ("s.s5.last", 256, 64, 1, 1, [(16, 16)]),
    ("s.fpn.in", 64, 128, 1, 1, [(32, 32)]),
    ("s.fpn.out3", 128, 128, 3, 1, [(32, 32)]),
    ("s.fpn.out4", 128, 128, 3, 1, [(16, 16)]),
    ("s.fpn.p6", 64, 128, 3, 2, [(16, 16)]),
    ("s.head.tower", 128, 128, 3, 1, [(32, 32), (16, 16), (8, 8), (4, 4)]),
    ("s.head.cls", 128, 16, 3, 1, [(32, 32), (16, 16), (8, 8), (4, 4)]),
    ("s.head.pose", 128, 240, 3, 1, [(32, 32), (16, 16), (8, 8), (4, 4)]),
]


EAGER = False


def timeit(fn, iters):
    if EAGER:           # for rocprofv3 counter runs: plain launches
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return 0.0
    return timeit_graph(fn, iters)


def timeit_graph(fn, iters):
    """`iters` launches captured in one hipGraph (no host launch cost between them), replayed 3x."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="all")
    ap.add_argument("--set", default="all")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=B)
    ap.add_argument("--only", default="")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--stats", type=int, default=-1, help="fwd: fused statistics (0 = per channel, G = groups), fp32 output")
    a = ap.parse_args()
    global EAGER
    EAGER = a.eager
    dev = torch.device("cuda:0")
    shapes = (TEACHER if a.set in ("teacher", "all") else []) + (STUDENT if a.set in ("student", "all") else [])
    kinds = ["fwd", "dgrad", "wgrad"] if a.kind == "all" else [a.kind]
    print("| layer | kind | M | Cout | K | GFLOP | us | TFLOP/s |")
    print("|---|---|---|---|---|---|---|---|")
    g = torch.Generator(device="cpu").manual_seed(0)
    ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)      # split-K workspace (64 MB)
    for name, cin, cout, k, stride, levels in shapes:
        if a.only and a.only not in name:
            continue
        geom = ops.Geom(a.batch, cin, cout, k, stride, k // 2, levels)
        x = (torch.randn(geom.rows_in, cin, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        w = (torch.randn(cout * k * k * cin, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        dy = (torch.randn(geom.rows_out, cout, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        y = torch.empty(geom.rows_out, cout, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(geom.rows_in, cin, dtype=torch.bfloat16, device=dev)
        dw = torch.zeros(cout * k * k * cin, dtype=torch.float32, device=dev)
        flop = 2.0 * geom.rows_out * cout * k * k * cin
        for kind in kinds:
            if name.startswith("t.") and kind != "fwd":
                continue
            if kind == "fwd" and a.stats >= 0:
                yf = torch.empty(geom.rows_out, cout, dtype=torch.float32, device=dev)
                st = torch.zeros(max(2 * cout, len(levels) * a.batch * max(a.stats, 1) * 2), device=dev)
                us0 = timeit(lambda: ops.conv2d_fwd(geom, x, w, out=yf, out_f32=True), a.iters)
                us = timeit(lambda: ops.conv2d_fwd(geom, x, w, out=yf, out_f32=True, stats=st, stats_groups=a.stats), a.iters)
                name = name + " (f32 out %.1f us; +stats)" % us0
            elif kind == "fwd":
                us = timeit(lambda: ops.conv2d_fwd(geom, x, w, out=y, workspace=ws), a.iters)
            elif kind == "dgrad":
                us = timeit(lambda: ops.conv2d_dgrad(geom, dy, w, dx=dx), a.iters)
            else:
                us = timeit(lambda: ops.conv2d_wgrad(geom, x, dy, dw), a.iters)
            print("| %s | %s | %d | %d | %d | %.2f | %.1f | %.0f |" % (name, kind, geom.rows_out, cout, k * k * cin,
                                                                      flop / 1e9, us, flop / max(us, 1e-9) / 1e6))


if __name__ == "__main__":
    main()
