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
sys.path.insert(0, os.path.join(HERE, "tools"))
sys.path.insert(0, os.path.join(HERE, "kd-6d-pose-adlp_amd"))
import torch  # noqa: E402

from kd6d import ops  # noqa: E402

from step_layers import B, STUDENT, STUDENT_640, TEACHER, TEACHER_640  # noqa: E402


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
    ap.add_argument("--opt", action="append", default=[], help="kernel-selection option name=value (kd6d_set_option)")
    a = ap.parse_args()
    for kv in a.opt:
        k_, v_ = kv.split("=")
        ops.set_option(k_, int(v_))
    global EAGER
    EAGER = a.eager
    dev = torch.device("cuda:0")
    shapes = (TEACHER if a.set in ("teacher", "all") else []) + (STUDENT if a.set in ("student", "all") else [])
    if a.set in ("teacher640", "student640"):          # the full-frame variant (480 x 640) of the same step
        shapes = TEACHER_640 if a.set == "teacher640" else STUDENT_640
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
        nw = cout * k * k * cin
        dw = torch.empty(ops.conv2d_wgrad_parts(geom, x.dtype), nw, device=dev)          # partial dW images (kd6d.h)
        flop = 2.0 * geom.rows_out * cout * k * k * cin
        for kind in kinds:
            if name.startswith(("t.", "t640.")) and kind != "fwd":
                continue
            if kind == "fwd" and a.stats >= 0:
                yf = torch.empty(geom.rows_out, cout, dtype=torch.float32, device=dev)
                st = ops.acc_zeros(max(2 * cout, len(levels) * a.batch * max(a.stats, 1) * 2), dev)
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
