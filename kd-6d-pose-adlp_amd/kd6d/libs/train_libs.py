"""Model / optimiser builders with the reference's names (libs/train_libs.py:80-206).

build_model   -> (model, optimizer, scheduler, total_steps): AdamW(lr=BASE_LR/N_GPU, wd 1e-4,
                 eps 1e-8) + OneCycleLR(linear, pct_start .05, MAX_ITER+100) exactly as the
                 reference, with the optimiser fused into kd6d_clip_adamw (clip folded in).
build_model_teacher -> frozen teacher.
Data-parallel: parameters are broadcast from rank 0 once (the reference's DDP constructor does the
same before being discarded) and gradients are mean-all-reduced every step over RCCL.
"""
import os

import torch
from torch import optim

from .. import backbone as _bb
from ..optim import FusedClipAdamW
from . import distributed as D


def build_backbone(arch):
    if arch == "darknet_tiny":
        return _bb.darknet_tiny(pretrained=True)
    if arch == "darknet_tiny_h":
        return _bb.darknet_tiny_h(pretrained=False)     # no pretrained weights exist for it
    if arch == "darknet53":
        return _bb.darknet53(pretrained=True)
    raise ValueError("unsupported backbone %r" % (arch,))


def _load_weights(model, path):
    if not path or not os.path.exists(path):
        print("-- Random initialized weights.")
        return False
    chkpt = torch.load(path, map_location="cpu")
    if "model" in chkpt:
        chkpt = chkpt["model"]
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in chkpt.items() if k in own}, strict=False)
    print("Weights are loaded from " + path)
    return True


def build_model(cfg, posemodule, device="cuda"):
    model = posemodule(cfg, build_backbone(cfg["MODEL"]["BACKBONE"]))
    _load_weights(model, cfg["RUNTIME"].get("WEIGHT_FILE", ""))
    model = model.to(device)
    n_gpu = cfg["RUNTIME"].get("N_GPU", 1)
    base_lr = cfg["SOLVER"]["BASE_LR"] / n_gpu
    optimizer = FusedClipAdamW(model, lr=base_lr, weight_decay=0.0001, eps=1e-8,
                               max_norm=cfg["SOLVER"].get("GRAD_CLIP", 1.0))
    scheduler = optim.lr_scheduler.OneCycleLR(optimizer, base_lr, cfg["SOLVER"]["MAX_ITER"] + 100, pct_start=0.05,
                                              cycle_momentum=False, anneal_strategy="linear")
    if cfg["RUNTIME"].get("DISTRIBUTED", False):
        D.broadcast_(model.net.store.params, 0)
        D.broadcast_(model.net.store.bufs, 0)
        model.net.invalidate()
    total_steps = 0
    wd = cfg["RUNTIME"].get("WORKING_DIR", "")
    latest = os.path.join(wd, "latest.pth") if wd else ""
    if latest and os.path.exists(latest):
        chkpt = torch.load(latest, map_location="cpu")
        total_steps = chkpt["steps"]
        model.load_state_dict(chkpt["model"])
        optimizer.load_state_dict(chkpt["optim"])
        scheduler.load_state_dict(chkpt["sched"])
        print("Weights, optimzer, scheduler are loaded from %s, starting from step %d" % (latest, total_steps))
    return model, optimizer, scheduler, total_steps


def build_model_teacher(cfg, posemodule, device):
    model = posemodule(cfg, build_backbone(cfg["MODEL"]["BACKBONE"]))
    _load_weights(model, cfg["RUNTIME"].get("WEIGHT_FILE", ""))
    model = model.to(device)
    if cfg["RUNTIME"].get("DISTRIBUTED", False):
        D.broadcast_(model.net.store.params, 0)
        D.broadcast_(model.net.store.bufs, 0)
        model.net.invalidate()
    return model
