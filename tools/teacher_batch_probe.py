"""Probe: the frozen teacher's forward (+ cell selection) alone, replayed as a hipGraph, at several batch sizes --
what a teacher pass over the images of TWO steps at once would save per image (tile quantisation, launch count).
Usage: python tools/teacher_batch_probe.py [--batches 16 32] [--frame crop256|full640]"""
import argparse
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "kd-6d-pose-adlp_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[16, 32])
    ap.add_argument("--frame", default="crop256")
    ap.add_argument("--reps", type=int, default=100)
    args = ap.parse_args()
    import bench
    from kd6d import backbone as BB
    from kd6d.kd_losses import PackedTargets, teacher_flats
    from kd6d.models.model_kd import PoseModuleKD
    from kd6d.synthetic import make_batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for B in args.batches:
        teacher = PoseModuleKD(bench.make_cfg("darknet53", "bf16"), BB.darknet53())
        teacher.net.reset_parameters(seed=2)
        with torch.no_grad():
            sd = teacher.state_dict()
            sd["head.cls_logits.bias"] = torch.tensor(bench.TEACHER_CLS_BIAS)
            teacher.load_state_dict(sd)
        teacher = teacher.to(dev).eval()
        images, targets = make_batch(B, 7, full_frame=args.frame == "full640")
        images = images.to(dev)
        tgt = PackedTargets(targets, dev)
        teacher._teacher_flats = teacher_flats(B, dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):
                teacher(images, targets=tgt, is_teacher=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), torch.no_grad():
            teacher(images, targets=tgt, is_teacher=True)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            g.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.reps * 1e3
        print("teacher forward B=%d: %.3f ms per pass, %.1f us per image" % (B, ms, ms / B * 1e3), flush=True)
        del g, teacher


if __name__ == "__main__":
    main()
