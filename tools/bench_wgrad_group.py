"""Micro-benchmark of the grouped weight gradient (ops.WgradGroup) on the student head of the KD step
(B = 16, 4 levels, 8 tower layers + cls_logits + pose_pred), against the per-layer launches it replaces.

    python tools/bench_wgrad_group.py [--channels 128] [--wgs 128,192,256,384,512]
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
from bench_conv import timeit_graph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--wgs", type=str, default="128,192,256,320,384,512")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    C, B = a.channels, a.batch
    levels = [(32, 32), (16, 16), (8, 8), (4, 4)]
    g = torch.Generator().manual_seed(0)
    layers = []
    for cout in [C] * 8 + [16, 240]:
        geom = ops.Geom(B, C, cout, 3, 1, 1, levels)
        x = (torch.randn(geom.rows_in, C, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        dy = (torch.randn(geom.rows_out, cout, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        dw = torch.zeros(cout * 9 * C, dtype=torch.float32, device=dev)
        db = torch.zeros(cout, dtype=torch.float32, device=dev)
        layers.append((geom, x, dy, dw, db))
    flop = sum(2.0 * ge.rows_out * ge.cout * 9 * C for ge, *_ in layers)

    def per_layer(budget):
        for (ge, x, dy, dw, db) in layers:
            key = (id(ge), budget)
            if key not in slabs:             # partial dW images + planar bias accumulators (kd6d.h)
                slabs[key] = (torch.empty(ops.conv2d_wgrad_parts(ge, x.dtype, True, budget), dw.numel(), device=dev),
                              ops.planar_acc(db.numel(), dev))
            slab, acc = slabs[key]
            ops.conv2d_wgrad(ge, x, dy, slab, dbias=acc[:db.numel()], acc_stride=db.numel(), cu_budget=budget)

    slabs = {}
    per_layer(0); per_layer(128)             # allocate outside the timed graph

    for budget in (0, 128):
        us = timeit_graph(lambda: per_layer(budget), a.iters)
        print("per-layer launches (cu_budget %3d): %8.1f us  %6.0f TFLOP/s" % (budget, us, flop / us / 1e6))
    for n_wg in [int(v) for v in a.wgs.split(",")]:
        grp = ops.WgradGroup(n_wg)

        def run():
            for ge, x, dy, dw, db in layers:
                grp.add(ge, x, dy, dw, db)
            grp.launch()

        us = timeit_graph(run, a.iters)
        print("grouped, %4d workgroups asked (%4d planned): %8.1f us  %6.0f TFLOP/s" % (n_wg, grp._info[0], us, flop / us / 1e6))


if __name__ == "__main__":
    main()
