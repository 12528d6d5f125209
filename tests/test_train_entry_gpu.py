"""The callers either side of the hot path: the BOP reader + GPU Dynamic-Zoom-In front-end as
train_kd.py consumes it (kd6d.libs.train_libs.build_dataset), resuming from latest.pth, and the optional teacher
PnP gate of postprocess_kd.py:187-202."""
import os
import sys

import numpy as np
import pytest
import torch

from test_step_gpu import build, make_cfg

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, G)


def test_build_dataset_yields_dzi_batches(gpu_device, tmp_path):
    """build_dataset(cfg): BOP image list -> frames -> ONE kd6d_dzi_crop launch per batch -> (ImageList of 256x256
    crops, PackedTargets with cropped masks / bbox_trans / bbox_scale, metas); the batch then goes through a teacher
    forward.  bbox_trans must map the projected 3D box of instance 0 into the crop (the DZI box is 1.5 x the box,
    jittered by <= 25 % of it, dzi_libs.py:14-53)."""
    from bop_fixture import write_tree
    from kd6d.libs.dataset import projected_box
    from kd6d.libs.train_libs import build_dataset, dataset_meshes
    tree = write_tree(str(tmp_path))
    cfg = make_cfg("darknet_tiny_h", "fp32")
    cfg["DATASETS"].update(TRAIN=tree["list_file"], VALID=tree["list_file"], MESH_DIR=tree["models"], BBOX_FILE=tree["bbox"],
                           N_CLASS=3)
    cfg["INPUT"].update(INTERNAL_WIDTH=tree["W"], INTERNAL_HEIGHT=tree["H"])
    cfg["SOLVER"]["IMS_PER_BATCH"] = 2
    cfg["RUNTIME"].update(N_GPU=1, DISTRIBUTED=False, NUM_WORKERS=0)
    np.random.seed(0)
    train_loader, valid_loader = build_dataset(cfg, gpu_device)
    assert len(dataset_meshes(valid_loader)) == 2
    images, tgt, metas = next(iter(train_loader))
    assert images.tensors.shape == (2, 3, 256, 256) and images.tensors.is_cuda and torch.isfinite(images.tensors).all()
    assert tgt.mask.shape == (2, 256, 256) and tgt.bbox_trans.shape == (2, 2, 3) and len(metas) == 2
    assert set(torch.unique(tgt.mask).tolist()) <= {-1.0, 0.0, 1.0, 2.0}
    ds = train_loader.loader.dataset
    by_path = {ds.img_files[i]: i for i in range(len(ds))}
    for b, m in enumerate(metas):
        frame, target, _ = ds[by_path[m["path"]]]
        box = projected_box(target, 0)
        A = tgt.bbox_trans[b].cpu().numpy().astype(np.float64)
        ctr = A[:, :2] @ np.array([0.5 * (box[0] + box[2]), 0.5 * (box[1] + box[3])]) + A[:, 2]
        side = max(box[2] - box[0], box[3] - box[1]) * A[0, 0]          # box side in crop pixels
        assert abs(A[0, 0] - A[1, 1]) < 1e-6 and A[0, 1] == 0 and A[1, 0] == 0
        assert np.all(np.abs(ctr - 128.0) <= 0.25 * side + 1.0), (ctr, side)
    # a whole frame list runs through the front-end
    assert sum(1 for _ in valid_loader) == 2


def test_resume_from_latest_pth(gpu_device, tmp_path):
    """train_kd.py:149-160 / libs/train_libs.py:144-166: {steps, model, optim, sched} written every VAL_FREQ steps;
    a rebuilt model + FusedClipAdamW + OneCycleLR continue with the same next step (fp32 eager; float atomics in the
    normalisation statistics leave ~1e-6 of summation-order noise)."""
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs.poses import ImageList
    from kd6d.libs.train_libs import build_model
    from kd6d.models.model_kd import PoseModuleKD
    from kd6d.synthetic import make_batch
    dev = gpu_device
    wd = str(tmp_path) + "/"
    cfg = make_cfg("darknet_tiny_h", "fp32")
    cfg["RUNTIME"].update(WORKING_DIR=wd, WEIGHT_FILE="", N_GPU=1, DISTRIBUTED=False)
    cfg["SOLVER"]["MAX_ITER"] = 40
    teacher = build("darknet53", "fp32", 2, dev, [1.0] + [-6.0] * 14).eval()
    images, targets = make_batch(2, 3, crop=64)
    img, tgt = ImageList(images.tensors.to(dev), images.sizes), PackedTargets(targets, dev)
    keys = torch.rand(2 * 85, generator=torch.Generator().manual_seed(1)).to(dev)

    def one_step(model, opt, sched):
        model._debug_keys = keys
        model.zero_grad()
        with torch.no_grad():
            pred_t = teacher(img, targets=tgt, is_teacher=True)
        _, ld = model(img, targets=tgt, pred_t=pred_t)
        (ld["loss_cls"] * 0.1 + ld["loss_reg"] + ld["loss_kd"] * 5.0).backward()
        opt.step(); sched.step()
        return [float(ld[k]) for k in ("loss_cls", "loss_reg", "loss_kd")]

    torch.manual_seed(0)
    model, opt, sched, steps = build_model(cfg, PoseModuleKD, dev)
    assert steps == 0
    model.train()
    for _ in range(3):
        one_step(model, opt, sched)
    torch.save({"steps": 3, "model": model.state_dict(), "optim": opt.state_dict(), "sched": sched.state_dict()},
               os.path.join(wd, "latest.pth"))
    want = one_step(model, opt, sched)
    model2, opt2, sched2, steps2 = build_model(cfg, PoseModuleKD, dev)
    model2.train()
    assert steps2 == 3 and opt2.steps == 3 and sched2.last_epoch == 3
    got = one_step(model2, opt2, sched2)
    torch.cuda.synchronize()
    np.testing.assert_allclose(got, want, rtol=1e-5)
    # AdamW moves a parameter whose gradient is ~0 by +-lr * (noise / |noise|): float-atomic summation order shows up
    # as a fraction of lr (2.5e-4 here) on a few hundred of 2.3 M elements
    lr = opt.param_groups[0]["lr"]
    d = (model2.net.store.params - model.net.store.params).abs()
    # (measured: max 4.5e-5 = 0.07 lr; 0.1-0.8 % of the elements above 1e-6, run to run; a lost optimiser state moves
    #  every element by ~lr)
    assert float(d.max()) <= 0.2 * lr and float((d > 1e-6).float().mean()) < 2e-2, (float(d.max()), lr)
    # the moments after step 4: a lost or stale state would differ by 0.9 * m3 (relative O(1)); what is left between two
    # runs of the same step is summation-order noise (measured: 2 of 2.24 M elements off by 1.3e-6 at |m| < 3e-4)
    for a, b in ((opt2.exp_avg, opt.exp_avg), (opt2.exp_avg_sq, opt.exp_avg_sq)):
        assert float((a - b).norm() / b.norm()) < 1e-4
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())
    assert opt2.param_groups[0]["lr"] == pytest.approx(opt.param_groups[0]["lr"], rel=1e-12)
    # an `optim` entry written by torch.optim.AdamW (the reference's) is refused with a clear message, the weights load
    torch.save({"steps": 3, "model": model.state_dict(), "optim": {"state": {}, "param_groups": [{}]}, "sched": sched.state_dict()},
               os.path.join(wd, "latest.pth"))
    model3, opt3, sched3, steps3 = build_model(cfg, PoseModuleKD, dev)
    assert steps3 == 3 and opt3.steps == 0
    assert torch.equal(model3.state_dict()["head.cls_logits.weight"], model.state_dict()["head.cls_logits.weight"])
    # ... and the OneCycle schedule still continues from step 3 (the stale `sched` entry above is at step 4: it is
    # replaced by a fresh schedule fast-forwarded to the checkpoint's step), only the Adam moments restart
    ref_p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.AdamW([ref_p], lr=cfg["SOLVER"]["BASE_LR"], weight_decay=1e-4, eps=1e-8)
    ref_sched = torch.optim.lr_scheduler.OneCycleLR(ref_opt, cfg["SOLVER"]["BASE_LR"], cfg["SOLVER"]["MAX_ITER"] + 100,
                                                    pct_start=0.05, cycle_momentum=False, anneal_strategy="linear")
    lrs = [ref_opt.param_groups[0]["lr"]]
    for _ in range(5):
        ref_opt.step(); ref_sched.step()
        lrs.append(ref_opt.param_groups[0]["lr"])
    assert sched3.last_epoch == 3 and opt3.param_groups[0]["lr"] == pytest.approx(lrs[3], rel=1e-9)
    model3.train()
    one_step(model3, opt3, sched3)
    assert opt3.param_groups[0]["lr"] == pytest.approx(lrs[4], rel=1e-9)
    # a latest.pth exactly as the reference writes it (train_kd.py:153-160): torch AdamW `optim`, torch OneCycleLR `sched`
    ref_opt2 = torch.optim.AdamW([ref_p], lr=cfg["SOLVER"]["BASE_LR"], weight_decay=1e-4, eps=1e-8)
    ref_sched2 = torch.optim.lr_scheduler.OneCycleLR(ref_opt2, cfg["SOLVER"]["BASE_LR"], cfg["SOLVER"]["MAX_ITER"] + 100,
                                                     pct_start=0.05, cycle_momentum=False, anneal_strategy="linear")
    for _ in range(3):
        ref_opt2.step(); ref_sched2.step()
    torch.save({"steps": 3, "model": model.state_dict(), "optim": ref_opt2.state_dict(), "sched": ref_sched2.state_dict()},
               os.path.join(wd, "latest.pth"))
    model4, opt4, sched4, steps4 = build_model(cfg, PoseModuleKD, dev)
    assert steps4 == 3 and opt4.steps == 0 and sched4.last_epoch == 3
    assert opt4.param_groups[0]["lr"] == pytest.approx(lrs[3], rel=1e-9)
    model4.train()
    one_step(model4, opt4, sched4)
    assert opt4.param_groups[0]["lr"] == pytest.approx(lrs[4], rel=1e-9)


def _encoded_pose_logits(targets, levels, B, noise, rng):
    """cls / reg logits (packed rows) whose confident cells vote the projected 3D-box corners of the targets' poses."""
    from kd6d import engine
    rows = B * sum(h * w for h, w in levels)
    cls = torch.full((rows, 16), -10.0)
    reg = torch.zeros(rows, 240)
    row0 = 0
    for li, (h, w) in enumerate(levels):
        st, sz = float(engine.ANCHOR_STRIDES[li]), float(engine.ANCHOR_SIZES[li])
        for b in range(B):
            t = targets[b]
            c = int(t.class_ids[0])
            Kb = t.K.numpy().astype(np.float64)
            cam = t.rotations[0].numpy().astype(np.float64) @ t.keypoints_3d[c].numpy().T.astype(np.float64) + t.translations[0].numpy().reshape(3, 1)
            uv = (Kb @ cam)[:2] / (Kb @ cam)[2]
            bt = t.bbox_trans.numpy().astype(np.float64)
            p = bt[:, :2] @ uv + bt[:, 2:3]
            ctr = p.mean(1)
            x, y = int(np.clip(ctr[0] // st, 0, w - 1)), int(np.clip(ctr[1] // st, 0, h - 1))
            r = row0 + b * h * w + y * w + x
            q = p + rng.normal(0, noise[b], p.shape)
            cls[r, c] = 3.0
            reg[r, c * 16:c * 16 + 8] = torch.from_numpy((q[0] - (x * st + st * 0.5)) / sz)
            reg[r, c * 16 + 8:c * 16 + 16] = torch.from_numpy((q[1] - (y * st + st * 0.5)) / sz)
        row0 += B * h * w
    return cls, reg


@pytest.mark.parametrize("group", [1, 2])
def test_train_entry_pipelined_takes_exactly_max_iter_steps(gpu_device, tmp_path, group):
    """train_kd.py --launch pipeline [--teacher_group 2] on synthetic batches: the pipeline holds 1 (2 * group) batches that
    have not had their student step when the loop approaches MAX_ITER; the loop stops feeding and drains them with flush(),
    so the checkpoint written at step MAX_ITER says exactly MAX_ITER optimiser steps -- 7 is neither a multiple of the
    group nor of the period -- and training ends normally."""
    import subprocess
    wd = str(tmp_path) + "/"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "train_kd.py"), "--config_file", "configs/ape.yaml", "--config_file_t",
           "configs/ape.yaml", "--backbone", "darknet_tiny_h", "--backbone_t", "darknet53", "--kd_weight", "5.",
           "--working_dir", wd, "--synthetic", "--skip_teacher_eval", "--launch", "pipeline", "--teacher_group", str(group),
           "--max_iters", "7", "--val_freq", "7", "--batch_size", "2", "--image_size", "64"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "Training finished" in r.stdout
    outs = [os.path.join(d, f) for d, _, fs in os.walk(wd) for f in fs]
    latest = [f for f in outs if f.endswith("latest.pth")]
    assert latest, outs
    ck = torch.load(latest[0], map_location="cpu", weights_only=False)
    assert ck["steps"] == 7
    assert ck["optim"]["steps"] == 7
    assert ck["sched"]["last_epoch"] == 7


def _train_cmd(root, wd, extra):
    return [sys.executable, os.path.join(root, "train_kd.py"), "--config_file", "configs/ape.yaml", "--config_file_t",
            "configs/ape.yaml", "--backbone", "darknet_tiny_h", "--backbone_t", "darknet53", "--kd_weight", "5.",
            "--working_dir", wd, "--synthetic", "--skip_teacher_eval", "--batch_size", "2", "--image_size", "64"] + extra


def test_train_entry_graphs_first_data_parallel_start_one_rank_rehearsal(gpu_device, tmp_path):
    """train_kd.py's data-parallel start in the grouped launch mode -- graphs recorded BEFORE the communicator, then
    barrier, kd6d_comm_init, parameter broadcast, in-place refresh of what the recorded kernels read (teacher included),
    the all-reduce of every step between the two graphs, the collective barrier-timeout check, shutdown -- rehearsed on
    ONE GPU with --rccl_single_rank (a one-rank process group; bench.py --rccl-single-rank does the same for the bench
    path).  The run must reach MAX_ITER with the same losses as the run without any process group: at world size 1 the
    broadcast and the mean all-reduce are identities, and the step is bitwise reproducible."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, extra in (("dp", ["--rccl_single_rank"]), ("plain", [])):
        wd = str(tmp_path / tag) + "/"
        cmd = _train_cmd(root, wd, ["--launch", "pipeline", "--teacher_group", "2", "--max_iters", "50", "--val_freq", "50"] + extra)
        r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
        assert "Training finished" in r.stdout
        outs[tag] = r.stdout
    dp = outs["dp"]
    i_graphs, i_comm = dp.find("step graphs recorded before the communicator"), dp.find("gradient exchange: ")
    assert 0 <= i_graphs < i_comm, dp[-2000:]                    # graphs first, communicator second
    assert re.search(r"gradient exchange: kd6d_comm \(librccl \d+\.\d+\.\d+\)", dp), dp[-2000:]
    assert "gradient exchange" not in outs["plain"]
    line = lambda text: [l for l in text.splitlines() if l.startswith("steps: 50/50")]
    assert line(dp) and line(dp)[0].split("(")[0] == line(outs["plain"])[0].split("(")[0], (line(dp), line(outs["plain"]))


def test_train_entry_two_ranks_when_two_gpus_are_visible(gpu_device, tmp_path):
    """The same start on TWO ranks over RCCL (the first multi-GPU box runs this; it skips itself on a one-GPU box):
    python -m torch.distributed.run --nproc-per-node 2 train_kd.py --launch pipeline --teacher_group 2.  Both ranks
    finish and rank 0 reports the kd6d route (global batch sharded, lr = BASE_LR / 2: libs/train_libs.py:117,272)."""
    import subprocess
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    wd = str(tmp_path) + "/"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577"] + _train_cmd(root, wd, ["--launch", "pipeline", "--teacher_group", "2", "--max_iters",
                                                             "20", "--val_freq", "20", "--batch_size", "4"])[1:]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=1200,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "step graphs recorded before the communicator" in r.stdout and "gradient exchange: kd6d_comm" in r.stdout
    assert "Training finished" in r.stdout


def test_teacher_pnp_gate(gpu_device):
    """postprocess_kd.py:187-202: an image's teacher cells are kept only if RANSAC-PnP recovers a pose from them.
    Image 0's cells vote a consistent pose (0.5 px noise) and stay; image 1's votes are scrambled (60 px) and go."""
    from kd6d import engine, kd_losses
    from kd6d.kd_losses import PackedTargets
    from kd6d.synthetic import make_batch
    dev = gpu_device
    B, crop = 2, 256
    _, targets = make_batch(B, 5, crop=crop)
    levels = [(crop // s, crop // s) for s in engine.ANCHOR_STRIDES]
    cls, reg = _encoded_pose_logits(targets, levels, B, noise=[0.5, 60.0], rng=np.random.default_rng(1))
    tgt = PackedTargets(targets, dev)
    teacher = build("darknet53", "fp32", 2, dev).eval()
    tk = kd_losses.teacher_select(cls.to(dev), reg.to(dev), levels, B, tgt.bbox_trans, 0.1, 10, 1.0, frame_wh=tgt.frame_wh)
    before = tk.t_cnt.cpu().tolist()
    assert min(before) >= 3
    teacher._apply_pnp_gate(tk, cls.to(dev), tgt)
    after = tk.t_cnt.cpu().tolist()
    assert after[0] == before[0] and after[1] == 0, (before, after)
    assert tk["post_pos_per_img"] == after
