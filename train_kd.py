"""KD training entry point -- same command line, config surface and loop as the reference's
train_kd.py:34-171, on the MI355X-native kd6d step.

    python train_kd.py --config_file ./configs/ape.yaml --config_file_t ./configs/ape.yaml \
        --backbone darknet_tiny_h --backbone_t darknet53 --weight_file_t model_t_ape/final.pth \
        --kd_weight 5. --max_iters 10000 --working_dir outputs/ape/kd/ --synthetic

Multi-GPU: python -m torch.distributed.run --nproc-per-node N train_kd.py ...  (one process per
GPU; the gradient exchange goes through kd6d_comm_* = librccl over xGMI).  Without --synthetic the BOP / LINEMOD
image lists of the yaml are read (kd6d/libs/train_libs.build_dataset: frames at the internal resolution, the
Dynamic-Zoom-In crop + normalisation run on the GPU); with it, seeded LINEMOD-shaped batches.  As in the
reference the teacher is validated once before training (skip with --skip_teacher_eval) and every VAL_FREQ steps
rank 0 validates the student (kd6d/libs/eval_libs.valid: eval forward -> pose candidates -> PnP-RANSAC -> ADI /
REP) and writes latest.pth.  Scalars go to tensorboardX under the reference's tags when that package is
installed, else to <working_dir>/scalars.jsonl with the same tags.
"""
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "kd-6d-pose-adlp_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from kd6d.arguments.argument_kd import get_args  # noqa: E402
from kd6d._lib import lib  # noqa: E402
from kd6d.kd_losses import PackedTargets  # noqa: E402
from kd6d.libs.distributed import get_rank, init_exchange, shard_batch, shutdown_exchange, synchronize  # noqa: E402
from kd6d.libs.eval_libs import valid  # noqa: E402
from kd6d.libs.train_libs import (build_dataset, build_model, build_model_teacher, dataset_meshes,  # noqa: E402
                                  start_exchange_after_graphs, stop_if_barrier_timeouts)
from kd6d.models.model_kd import PoseModuleKD as PoseModule  # noqa: E402
from kd6d.synthetic import make_batch  # noqa: E402


def synthetic_loader(cfg, device, n_batches=8):
    per_gpu = shard_batch(cfg["SOLVER"]["IMS_PER_BATCH"])
    size = cfg["RUNTIME"].get("IMAGE_SIZE", 256)
    batches = []
    for i in range(n_batches):
        images, targets = make_batch(per_gpu, 1000 * get_rank() + i, crop=size,
                                     mixed_classes=cfg["DATASETS"].get("MIXED_CLASSES", False),
                                     class_offset=get_rank() * per_gpu)
        batches.append((images.to(device), PackedTargets(targets, device), None))
    while True:
        for b in batches:
            yield b


def synthetic_valid_loader(cfg, device, n_batches=2):
    """(images, PackedTargets, meta_infos) of held-out seeded batches + surrogate meshes (the 8 box corners)."""
    per_gpu = shard_batch(cfg["SOLVER"]["IMS_PER_BATCH"])
    size = cfg["RUNTIME"].get("IMAGE_SIZE", 256)
    loader, meshes = [], None
    for i in range(n_batches):
        images, targets = make_batch(per_gpu, 900000 + i, crop=size,
                                     mixed_classes=cfg["DATASETS"].get("MIXED_CLASSES", False))
        metas = [{"path": "val%d_%d" % (i, j), "K": t.K.numpy(), "class_ids": [int(c) for c in t.class_ids],
                  "rotations": [r.numpy() for r in t.rotations],
                  "translations": [x.numpy().reshape(3, 1) for x in t.translations]} for j, t in enumerate(targets)]
        meshes = [targets[0].keypoints_3d[c].numpy() for c in range(cfg["DATASETS"]["N_CLASS"] - 1)]
        loader.append((images.to(device), PackedTargets(targets, device), metas))
    return loader, meshes


class ScalarWriter:
    """tensorboardX.SummaryWriter when importable (train_kd.py:81 of the reference), else one JSON line per scalar."""

    def __init__(self, log_dir):
        self.tb = self.fh = None
        try:
            from tensorboardX import SummaryWriter
            self.tb = SummaryWriter(log_dir)
        except ImportError:
            self.fh = open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag, value, step):
        if self.tb is not None:
            self.tb.add_scalar(tag, value, step)
        else:
            self.fh.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
            self.fh.flush()

    def add_scalars(self, main_tag, values, step):
        for k, v in values.items():
            self.add_scalar("%s/%s" % (main_tag, k), v, step)


if __name__ == "__main__":
    torch.manual_seed(0)
    np.random.seed(0)
    random.seed(0)
    cfg, cfg_t = get_args()

    n_gpu = int(os.environ["WORLD_SIZE"]) if "WORLD_SIZE" in os.environ else 1
    local_rank = int(os.environ.get("LOCAL_RANK", cfg["RUNTIME"]["LOCAL_RANK"]))
    cfg["RUNTIME"]["N_GPU"] = n_gpu
    # --rccl_single_rank: a one-rank process group that still runs the whole data-parallel path (communicator, parameter
    # broadcast, the all-reduce of every step, the collective stop) -- the rehearsal a one-GPU box allows
    rehearsal = n_gpu == 1 and cfg["RUNTIME"].get("RCCL_SINGLE_RANK", False)
    if rehearsal:
        import socket
        from kd6d.libs import distributed as _D0
        _D0.SINGLE_RANK_EXCHANGE = True
        with socket.socket() as _s:
            _s.bind(("127.0.0.1", 0))
            _port = _s.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_port))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    cfg["RUNTIME"]["DISTRIBUTED"] = n_gpu > 1 or rehearsal
    cfg_t["RUNTIME"]["DISTRIBUTED"] = n_gpu > 1 or rehearsal
    device = cfg["RUNTIME"]["RUNNING_DEVICE"]
    if device != "cuda":
        raise SystemExit("the kd6d step runs on MI355X only (--running_device cuda); the CPU restatement "
                         "lives in oracle/ and is test infrastructure")
    if n_gpu > 1 and hasattr(os, "sched_setaffinity") and int(cfg["RUNTIME"].get("NUM_WORKERS", 0)) == 0:
        # one host thread per rank issues ~1 ms of launches per step: keep it on its own slice of the cores of this NODE
        # (LOCAL_WORLD_SIZE ranks share it).  Only without DataLoader workers: they would inherit the narrowed mask.
        cpus = sorted(os.sched_getaffinity(0))
        local_n = int(os.environ.get("LOCAL_WORLD_SIZE", n_gpu))
        per = max(len(cpus) // max(local_n, 1), 1)
        os.sched_setaffinity(0, cpus[local_rank * per:(local_rank + 1) * per] or cpus)
    torch.cuda.set_device(local_rank)
    if cfg["RUNTIME"].get("TWO_LAUNCH_NORM_BWD"):
        from kd6d import ops
        ops.set_option("bn.onepass", 0)
        ops.set_option("gn.onepass", 0)
    # Data-parallel + grouped teacher pass: the step's hipGraphs are recorded BEFORE the process's first RCCL communicator
    # exists (graphs instantiated after ncclCommInitRank replay ~18 % slower in that launch mode) --
    # no barrier, no communicator and no parameter broadcast until GroupedTeacherKDStep.prepare() has run, further down.
    # (The overlapped exchange captures its collectives inside the step graph and needs the communicator first.)
    graphs_first = (cfg["RUNTIME"]["DISTRIBUTED"] and cfg["RUNTIME"].get("LAUNCH") == "pipeline"
                    and int(cfg["RUNTIME"].get("TEACHER_GROUP", 1)) > 1 and cfg["RUNTIME"].get("EXCHANGE", "between") == "between")
    cfg["RUNTIME"]["DEFER_BROADCAST"] = cfg_t["RUNTIME"]["DEFER_BROADCAST"] = graphs_first
    if cfg["RUNTIME"]["DISTRIBUTED"]:
        torch.distributed.init_process_group(backend="nccl", init_method="env://")
        from kd6d.libs import distributed as _D
        _D.EXCHANGE_MODE = cfg["RUNTIME"].get("EXCHANGE", "between")
        if not graphs_first:
            synchronize()
            print("gradient exchange: " + init_exchange())      # kd6d_comm_* over librccl (include/kd6d.h)

    if cfg["RUNTIME"]["SYNTHETIC"]:
        train_loader = synthetic_loader(cfg, device)
        valid_loader, valid_meshes = synthetic_valid_loader(cfg, device)
    else:
        train_loader, valid_loader = build_dataset(cfg, device)               # train_kd.py:57 of the reference
        valid_meshes = dataset_meshes(valid_loader)

        def epochs(loader):                                                    # the reference loops `while True` over epochs
            while True:
                for item in loader:
                    yield item
        train_loader = epochs(train_loader)
    cfg["KD"]["vis_dir"] = cfg["RUNTIME"]["WORKING_DIR"]

    print("Building teacher ......")
    model_t = build_model_teacher(cfg_t, PoseModule, device)
    print("Building student ......")
    model, optimizer, scheduler, total_steps = build_model(cfg, PoseModule, device)
    VAL_FREQ = cfg["SOLVER"]["VAL_FREQ"]
    wd = cfg["RUNTIME"]["WORKING_DIR"]
    print("working directory: " + wd)
    if get_rank() == 0:
        os.makedirs(wd, exist_ok=True)
        print("Model size: Student VS Teacher: %d  vs %d" % (sum(p.numel() for p in model.parameters()),
                                                           sum(p.numel() for p in model_t.parameters())))
        with open(os.path.join(wd, "cfg.json"), "w") as f:
            json.dump(cfg, f, indent=4, sort_keys=True, default=str)
    logger = ScalarWriter(wd) if get_rank() == 0 else None

    model_t.eval()
    if get_rank() == 0 and not cfg["RUNTIME"]["SKIP_TEACHER_EVAL"]:
        # train_kd.py:85-86 of the reference: the teacher's own accuracy before distilling from it
        acc_t = valid(cfg_t, 0, valid_loader, model_t, device, valid_meshes)
        seen = [a for a in acc_t[0] if a]
        print("teacher valid: %s" % ({k: round(float(sum(a[k] for a in seen)) / len(seen), 2) for k in seen[0]}
                                      if seen else "no objects"))
    model.train()
    cfg_kd = cfg["KD"]
    w_cls, w_reg, w_kd = cfg["SOLVER"]["LOSS_WEIGHT_CLS"], cfg["SOLVER"]["LOSS_WEIGHT_REG"], cfg["KD"]["LOSS_WEIGHT_KD"]
    t0 = time.time()
    launch = cfg["RUNTIME"].get("LAUNCH", "graph")
    if launch != "pipeline" and int(cfg["RUNTIME"].get("TEACHER_GROUP", 1)) > 1 and get_rank() == 0:
        print("note: --teacher_group %d is ignored without --launch pipeline" % int(cfg["RUNTIME"]["TEACHER_GROUP"]))
    gstep = None
    if launch != "eager":
        from kd6d.graph import GraphedKDStep, GroupedTeacherKDStep
        group = int(cfg["RUNTIME"].get("TEACHER_GROUP", 1))
        if launch == "pipeline" and group > 1:
            if cfg["RUNTIME"]["DISTRIBUTED"] and not graphs_first and get_rank() == 0:
                print("note: --teacher_group %d with --exchange overlap: the communicator has to exist before the step is "
                      "recorded, and hipGraphs recorded after it replay ~18 %% slower in this launch mode (DESIGN.md "
                      "section 7); --exchange between records the graphs first" % group)
            # the teacher over the batches of `group` steps in one pass (2 * group batches in flight)
            gstep = GroupedTeacherKDStep(model_t, model, optimizer, (w_cls, w_reg, w_kd), cfg_kd=cfg_kd, group=group)
        else:
            gstep = GraphedKDStep(model_t, model, optimizer, (w_cls, w_reg, w_kd), cfg_kd=cfg_kd,
                                  pipeline=(launch == "pipeline"))
    if graphs_first:
        # graphs first (from the first training batch, which is then fed as usual), communicator second, then what the
        # model builders deferred: rank 0's weights and buffers to every rank, and an eager refresh of what the recorded
        # kernels read of them (bf16 shadow, dgrad packing)
        import itertools
        train_iter = iter(train_loader)
        first = next(train_iter)
        print("step graphs recorded before the communicator (graphs first)")
        start_exchange_after_graphs(gstep, model_t, model, first)        # kd6d/libs/train_libs.py: the order and why
        train_loader = itertools.chain([first], train_iter)
    MAX_ITER = cfg["SOLVER"]["MAX_ITER"]
    pipelined = gstep is not None and gstep.pipeline
    for idx, (images, targets, _) in enumerate(train_loader):
        if total_steps >= MAX_ITER:
            if get_rank() == 0:
                valid(cfg, total_steps, valid_loader, model, device, valid_meshes, logger=logger)      # train_kd.py:95-99
                torch.save(model.state_dict(), os.path.join(wd, "final.pth"))
            print("Training finished")
            break
        if gstep is not None:
            # the same iteration body (train_kd.py:104-140 of the reference), captured once and replayed.  Pipelined:
            # call k runs the teacher on batch k beside the student step on batch k-1 (grouped: k - 2 * group), so the
            # LAST steps are flushes of the pending batches (no new batch is consumed for them): exactly MAX_ITER
            # optimiser steps
            if pipelined and gstep.pending_steps > 0 and total_steps + gstep.pending_steps >= MAX_ITER:
                loss_dict = gstep.flush()
            else:
                loss_dict = gstep(images, targets)
            if loss_dict is None:          # priming call of the pipeline: teacher only
                continue
            total_steps += 1
            loss_cls, loss_reg, loss_kd = (loss_dict["loss_cls"] * w_cls, loss_dict["loss_reg"] * w_reg,
                                           loss_dict["loss_kd"] * w_kd)
        else:
            total_steps += 1
            model.zero_grad()
            with torch.no_grad():
                pred_t = model_t(images, targets=targets, is_teacher=True, cfg_kd=cfg_kd)
            _, loss_dict = model(images, targets=targets, pred_t=pred_t, cfg_kd=cfg_kd)
            loss_cls = (loss_dict["loss_cls"] * w_cls).mean()
            loss_reg = (loss_dict["loss_reg"] * w_reg).mean()
            loss = loss_cls + loss_reg
            loss_kd = (loss_dict["loss_kd"] * w_kd).mean()
            if w_kd > 0.0:
                loss = loss + loss_kd
            loss.backward()
            optimizer.step()          # clip_grad_norm_(GRAD_CLIP) is fused into the optimiser kernel
        scheduler.step()
        if logger is not None and total_steps % 10 == 0:                    # train_kd.py:113-122: unweighted values
            logger.add_scalar("training/learning_rate", optimizer.param_groups[0]["lr"], total_steps)
            logger.add_scalar("training/loss_cls", float(loss_dict["loss_cls"]), total_steps)
            logger.add_scalar("training/loss_reg", float(loss_dict["loss_reg"]), total_steps)
            logger.add_scalar("training/loss_cls_reg", float(loss_dict["loss_cls"]) + float(loss_dict["loss_reg"]), total_steps)
            logger.add_scalar("training/loss_kd", float(loss_dict["loss_kd"]), total_steps)
        if get_rank() == 0 and (total_steps % 50 == 0 or total_steps == 1):
            dt = time.time() - t0
            print("steps: %d/%d, lr:%.6f, cls:%.4f, reg:%.4f, kd:%.4f  (%.1f img/s)" % (
                total_steps, cfg["SOLVER"]["MAX_ITER"], optimizer.param_groups[0]["lr"], float(loss_cls),
                float(loss_reg), float(loss_kd), idx * cfg["SOLVER"]["IMS_PER_BATCH"] / max(dt, 1e-9)))
        if total_steps % 50 == 0 or total_steps % VAL_FREQ == 0:
            # in-kernel barriers of the one-launch BN / GN backward give up after a bounded spin instead of hanging
            # the GPU; a wait that gave up means wrong gradients, so training stops
            # The decision is collective (MAX over ranks of the per-device counter): a rank that stopped alone would
            # leave the others waiting in the next gradient all-reduce.
            stop_if_barrier_timeouts(lib.kd6d_barrier_timeouts(), cfg["RUNTIME"]["DISTRIBUTED"])
        if get_rank() == 0 and total_steps % VAL_FREQ == 0:
            acc = valid(cfg, total_steps, valid_loader, model, device, valid_meshes, logger=logger)     # train_kd.py:148-150
            model.train()
            seen = [a for a in acc[0] if a]
            print("valid @ %d: %s" % (total_steps, {k: round(float(sum(a[k] for a in seen)) / len(seen), 2) for k in seen[0]}
                                      if seen else "no objects"))
            torch.save({"steps": total_steps, "model": model.state_dict(), "optim": optimizer.state_dict(),
                        "sched": scheduler.state_dict()}, os.path.join(wd, "latest.pth"))
    if get_rank() == 0:
        with open(os.path.join(wd, "info.txt"), "w") as f:
            f.write("finished at: %s\nworking_dir: %s\ncommands:%s" % (
                time.strftime("%Y%m%d_%H%M%S"), wd, " ".join(sys.argv)))
    if cfg["RUNTIME"]["DISTRIBUTED"]:
        synchronize()
        shutdown_exchange()                       # ncclCommDestroy of the kd6d communicator
        torch.distributed.destroy_process_group()
