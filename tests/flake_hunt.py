"""Replay one graphed KD step many times on FIXED weights and report every buffer that is not bitwise the same as in the
first replay -- the tool that localises a run-to-run difference to the first kernel that produced it.  (Round 4: it found
that packed-fp32 instructions in the loss kernels occasionally return a wrong half beside the other network's
convolutions: DESIGN.md section 6, kd-6d-pose-adlp_amd/build.py EXTRA_FLAGS.)

    python tests/flake_hunt.py [--arch darknet_tiny] [--mixed] [--precision bf16] [--iters 300] [--group 1]

The optimiser runs with lr = 0 and weight_decay = 0, so the parameters (and everything derived from them) do not move;
the same batch is fed every call, so from the third call on every replay computes the same thing.  Compared per replay:
the three losses, the gradient norm, the flat gradient bucket and every named activation / scratch buffer of both
networks (PoseNet._bufs).  A buffer that differs is printed with the number of differing elements and where they lie
(first / last differing row), in the order the forward and reverse sweeps allocate them.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kd-6d-pose-adlp_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))     # test infrastructure: builds its networks like tests/test_step_gpu.py (seeded weights from oracle/)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="darknet_tiny")
    ap.add_argument("--mixed", action="store_true")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--group", type=int, default=1)
    ap.add_argument("--stop", type=int, default=3, help="stop after this many differing replays")
    ap.add_argument("--tcut", type=int, default=0, help="variant tcut: the teacher's forward stops at its N-th layer-group "
                    "boundary (of ~40) -- which part of it disturbs the student's loss kernels")
    ap.add_argument("--opt", action="append", default=[], help="kernel-selection option name=value (kd6d_set_option)")
    ap.add_argument("--dot", default="", help="write the captured step graph as DOT here (hipGraphDebugDotPrint)")
    ap.add_argument("--variant", default="", help="pipeline-mode experiments: join (the loss kernels wait for the teacher's "
                    "stream), noblocks (no hand-over blocks: separate copies), noside (no side stream for the SSC assignment)")
    ap.add_argument("--mode", default="pipeline", choices=["pipeline", "sequential", "eager", "eager1"],
                    help="pipeline / sequential: GraphedKDStep with / without the cross-step pipeline; eager: launches from "
                         "Python with the side streams; eager1: launches from Python on ONE stream")
    args = ap.parse_args()
    from test_step_gpu import build, ref_to_packed_rows
    from kd6d.graph import GraphedKDStep, GroupedTeacherKDStep
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.optim import FusedClipAdamW
    from kd6d.synthetic import make_batch
    from kd6d import ops
    dev = torch.device("cuda:0")
    for kv in args.opt:
        k, v = kv.split("=")
        ops.set_option(k, int(v))
    B, crop = 16, 256
    BIAS = [1.0] + [-6.0] * 14
    teacher = build("darknet53", args.precision, 2, dev, BIAS).eval()
    student = build(args.arch, args.precision, 1, dev).train()
    with_opt = "opt" in args.variant       # a real update every replay, state rewound before the next one
    opt = FusedClipAdamW(student, lr=1e-3 if with_opt else 0.0, weight_decay=1e-4 if with_opt else 0.0, eps=1e-8, max_norm=1.0)
    levels = [(crop // 8 // (2 ** i),) * 2 for i in range(4)]
    cells = sum(h * w for h, w in levels)
    keys_ref = torch.rand(B * cells, generator=torch.Generator().manual_seed(17))
    student._debug_keys = keys_ref[ref_to_packed_rows(B, levels)].to(dev)
    images, targets = make_batch(B, 41, crop=crop, mixed_classes=args.mixed, full_frame=False)
    batch = (ImageList(images.tensors.to(dev), images.sizes), PackedTargets(targets, dev))
    if args.mode in ("eager", "eager1"):
        if args.mode == "eager":
            student.net.side_stream = torch.cuda.Stream()
        w = torch.tensor([0.1, 1.0, 5.0], device=dev)

        def g(images, tgt):
            with torch.no_grad():
                pred_t = teacher(images, targets=tgt, is_teacher=True)
            losses = student.step_losses(images, tgt, pred_t, w)
            opt.step()
            return {"loss_cls": losses[0:1], "loss_reg": losses[1:2], "loss_kd": losses[2:3]}
        warm = 3
    elif args.group == 1:
        g = GraphedKDStep(teacher, student, opt, (0.1, 1.0, 5.0), pipeline=args.mode == "pipeline")
        warm = 3
        if "noblocks" in args.variant:
            g._make_blocks = lambda: None
        if "noside" in args.variant:
            student.net.side_stream = None
        if "poison" in args.variant:
            # every replay starts by filling the student's logits with NaN: a consumer that runs before the producing
            # convolution of THIS replay has finished shows up as NaN instead of as last replay's (identical) values
            b0 = student._begin_step

            def b1(x):
                for k, buf in student.net._bufs.items():
                    if isinstance(k, tuple) and isinstance(k[0], str) and k[0].endswith(".logits"):
                        buf.fill_(float("nan"))
                return b0(x)
            student._begin_step = b1
        if "fake" in args.variant:
            # the teacher's forward replaced by vendor GEMMs of about the same duration on the teacher's stream: is it
            # OUR teacher kernels or any concurrent work that disturbs the student's loss kernels?
            from kd6d.kd_losses import DeferredTeacher
            A = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
            Bm = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
            C = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
            torch.mm(A, Bm, out=C)

            def fake_launch(images, tgt):
                g.teacher_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(g.teacher_stream):
                    for _ in range(12):
                        torch.mm(A, Bm, out=C)
                return DeferredTeacher(g.t_cur, g.teacher_stream)
            g._teacher_launch = fake_launch
        if "tcut" in args.variant:
            from kd6d.kd_losses import DeferredTeacher

            class Stop(Exception):
                pass
            st_ = {"n": 0}

            def hook():
                st_["n"] += 1
                if st_["n"] == args.tcut:
                    raise Stop()

            def cut_launch(images, tgt):
                g.teacher_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(g.teacher_stream):
                    st_["n"] = 0
                    teacher.net.cut_hook = hook
                    try:
                        with torch.no_grad():
                            g.teacher(images, targets=tgt, is_teacher=True, cfg_kd=g.cfg_kd)
                    except Stop:
                        pass
                    finally:
                        teacher.net.cut_hook = None
                return DeferredTeacher(g.t_cur, g.teacher_stream)
            g._teacher_launch = cut_launch
        if "noheal" in args.variant:
            # the end-of-step hand-over copy (block[1] -> block[0]) is dropped after the pipeline has filled: the same batch
            # is fed every call, so block[0] must never change again -- any byte of it that does was written by a kernel
            # that has no business there
            g._advance_real = g._advance
            g._advance = lambda pred: (g._advance_real(pred) if not getattr(g, "_frozen", False) else None)
        if "join" in args.variant:
            fwd0 = student.loss_evaluator.forward

            def fwd1(*a, **k):
                if g.teacher_stream is not None:
                    torch.cuda.current_stream().wait_stream(g.teacher_stream)
                return fwd0(*a, **k)
            student.loss_evaluator.forward = fwd1
    else:
        g = GroupedTeacherKDStep(teacher, student, opt, (0.1, 1.0, 5.0), group=args.group)
        warm = 3 * args.group
    if args.dot:
        real = torch.cuda.CUDAGraph

        class Dbg(real):
            def __new__(cls, *a, **k):
                o = real.__new__(cls, *a, **k)
                return o

            def __init__(self, *a, **k):
                super().__init__(*a, **k)
                self.enable_debug_mode()
        torch.cuda.CUDAGraph = Dbg
    for _ in range(warm):
        g(*batch)
    torch.cuda.synchronize()
    if args.dot:
        g.g_step.debug_dump(args.dot)
        print("graph written to", args.dot)
        return

    st_ = student.net.store
    state0 = None
    if with_opt:
        state0 = dict(params=st_.params.clone(), bufs=st_.bufs.clone(), m=opt.exp_avg.clone(), v=opt.exp_avg_sq.clone(),
                      nbt=student._nbt.clone(), steps=opt.steps)

    def snapshot():
        out = {}
        if with_opt:        # rewind: the same update from the same state every replay
            st_.params.copy_(state0["params"]); st_.bufs.copy_(state0["bufs"])
            opt.exp_avg.copy_(state0["m"]); opt.exp_avg_sq.copy_(state0["v"])
            student._nbt.copy_(state0["nbt"]); opt.steps = state0["steps"]
            student.net.refresh_derived_in_place(need_dgrad=True)
        ld = g(*batch)
        torch.cuda.synchronize()
        if with_opt:
            out["opt.params_after"] = st_.params.clone()
            out["opt.exp_avg_after"] = opt.exp_avg.clone()
            out["opt.exp_avg_sq_after"] = opt.exp_avg_sq.clone()
            if st_.shadow is not None:
                out["opt.shadow_after"] = st_.shadow.clone()
            out["opt.bn_buffers_after"] = st_.bufs.clone()
        for k, v in ld.items():
            out["loss." + k] = v.detach().clone().reshape(-1)
        out["grad_norm"] = torch.tensor([float(opt.grad_norm())])
        out["student.grads"] = student.net.store.grads.clone()
        if getattr(g, "_blocks", None) is not None and "blocks" in args.variant:
            out["block0.after"] = g._blocks[0].clone()     # what the NEXT step's student will read
            out["block1.after"] = g._blocks[1].clone()
        ctx = student.loss_evaluator.ctx or {}
        for k in ("cls", "reg", "labels", "pos_cnt", "pos_row", "pos_gt", "xs", "alpha", "g_reg", "g_xs", "g_alpha", "valid",
                  "n_valid"):
            if isinstance(ctx.get(k), torch.Tensor):
                out["lossctx." + k] = ctx[k].clone()
        for tag, net in (("teacher", teacher.net), ("student", student.net)):
            for key, buf in net._bufs.items():
                if key == "__workspace__":
                    continue
                out["%s.%s" % (tag, key if isinstance(key, str) else "/".join(map(str, key)))] = buf.clone()
        return out

    if "noheal" in args.variant:
        # NOTE: the graph was captured with the copy inside; re-capture without it
        g._frozen = True
        g.g_step = None
        g(*batch)
        torch.cuda.synchronize()
        pristine = g._blocks[0].clone()
        offs_note = "block bytes %d" % pristine.numel()
        print(offs_note)
        for it in range(args.iters):
            g(*batch)
            torch.cuda.synchronize()
            ne = torch.nonzero(g._blocks[0] != pristine).reshape(-1)
            if ne.numel():
                print("replay %d: %d bytes of block[0] changed, byte offsets %d..%d" % (it, ne.numel(), int(ne[0]), int(ne[-1])))
                w0 = (int(ne[0]) // 4) * 4
                a = pristine[w0:w0 + 256].view(torch.float32).cpu()
                b = g._blocks[0][w0:w0 + 256].view(torch.float32).cpu()
                print("   pristine:", [round(float(v), 4) for v in a[:24]])
                print("   now     :", [round(float(v), 4) for v in b[:24]])
                ai_, bi_ = pristine[w0:w0 + 256].view(torch.int32).cpu(), g._blocks[0][w0:w0 + 256].view(torch.int32).cpu()
                print("   as int32:", ai_[:12].tolist(), "->", bi_[:12].tolist())
                g._blocks[0].copy_(pristine)
                args.stop -= 1
                if args.stop <= 0:
                    break
        print("part offsets: nhwc 0, then targets, flats (see GraphedKDStep._make_blocks); tgt block bytes", g.tgt.block_bytes())
        print("... and the usual comparison on the frozen pipeline:")
    ref = snapshot()
    print("comparing %d buffers (%.1f MB) over %d replays" % (len(ref), sum(v.numel() * v.element_size() for v in ref.values()) / 1e6,
                                                              args.iters), flush=True)
    bad = 0
    for it in range(args.iters):
        cur = snapshot()
        diff = [k for k in ref if k in cur and not torch.equal(ref[k], cur[k])]
        if diff:
            bad += 1
            print("replay %d: %d buffers differ" % (it, len(diff)))
            for k in diff:
                a, b = ref[k], cur[k]
                ne = (a != b)
                if a.dim() == 2:
                    rows = torch.nonzero(ne.any(1)).reshape(-1)
                    where = "rows %d..%d of %d (%d rows)" % (int(rows[0]), int(rows[-1]), a.shape[0], rows.numel())
                else:
                    idx = torch.nonzero(ne.reshape(-1)).reshape(-1)
                    where = "elements %d..%d of %d" % (int(idx[0]), int(idx[-1]), a.numel())
                print("   %-60s %9d differing, %s, max |d| %.3g" % (k, int(ne.sum()), where,
                                                                    float((a.float() - b.float()).abs().max())))
            if "student.dreg/(21760, 240)/torch.bfloat16" in diff or any(k.startswith("student.dreg") for k in diff):
                k = [k for k in diff if k.startswith("student.dreg")][0]
                a, b = ref[k].float().cpu(), cur[k].float().cpu()
                idx = torch.nonzero(a != b)
                for r, c in idx[:40].tolist():
                    print("      dreg[%d, %d]: %.6g vs %.6g" % (r, c, float(a[r, c]), float(b[r, c])))
                pr = cur["lossctx.pos_row"].cpu().reshape(16, -1)
                pc = cur["lossctx.pos_cnt"].cpu()
                pg = cur["lossctx.pos_gt"].cpu().reshape(16, -1)
                rows = sorted(set(idx[:, 0].tolist()))
                for bimg in range(16):
                    lst = pr[bimg, :int(pc[bimg])].tolist()
                    if any(r in lst for r in rows):
                        print("      image %d: pos_cnt %d pos_row %s pos_gt %s dup=%s" % (bimg, int(pc[bimg]), lst,
                              pg[bimg, :int(pc[bimg])].tolist(), len(set(lst)) != len(lst)))
            for k in diff:
                if k.startswith("block"):
                    a, b = ref[k].cpu(), cur[k].cpu()
                    idx = torch.nonzero(a != b).reshape(-1)
                    w0 = (int(idx[0]) // 4) * 4
                    print("      %s: %d bytes differ, offsets %d..%d; floats at %d: %s -> %s" % (
                        k, idx.numel(), int(idx[0]), int(idx[-1]), w0,
                        [round(float(v), 5) for v in a[w0:w0 + 64].view(torch.float32)],
                        [round(float(v), 5) for v in b[w0:w0 + 64].view(torch.float32)]))
                if k.startswith("lossctx.") or k.startswith("loss."):
                    a, b = ref[k].reshape(-1).cpu(), cur[k].reshape(-1).cpu()
                    idx = torch.nonzero(a != b).reshape(-1)[:48].tolist()
                    print("      %s: %s" % (k, ", ".join("[%d] %.7g vs %.7g" % (i, float(a[i]), float(b[i])) for i in idx)))
            sys.stdout.flush()
            if bad >= args.stop:
                break
    print("barrier timeouts:", int(ops.lib.kd6d_barrier_timeouts()))
    print("replays that differed: %d of %d" % (bad, it + 1))


if __name__ == "__main__":
    main()
