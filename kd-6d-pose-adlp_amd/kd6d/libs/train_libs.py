"""Model / optimiser builders with the reference's names (libs/train_libs.py:80-206).

build_model   -> (model, optimizer, scheduler, total_steps): AdamW(lr=BASE_LR/N_GPU, wd 1e-4,
                 eps 1e-8) + OneCycleLR(linear, pct_start .05, MAX_ITER+100) exactly as the
                 reference, with the optimiser fused into kd6d_clip_adamw (clip folded in).
build_model_teacher -> frozen teacher.
Data-parallel: parameters are broadcast from rank 0 once (the reference's DDP constructor does the
same before being discarded) and gradients are mean-all-reduced every step over RCCL.
"""
import os

import numpy as np
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
    if cfg["RUNTIME"].get("DISTRIBUTED", False) and not cfg["RUNTIME"].get("DEFER_BROADCAST", False):
        # (DEFER_BROADCAST: train_kd.py broadcasts after the step's graphs have been recorded, see there)
        D.broadcast_(model.net.store.params, 0)
        D.broadcast_(model.net.store.bufs, 0)
        model.net.invalidate()
    total_steps = 0
    wd = cfg["RUNTIME"].get("WORKING_DIR", "")
    latest = os.path.join(wd, "latest.pth") if wd else ""
    if latest and os.path.exists(latest):
        chkpt = torch.load(latest, map_location="cpu", weights_only=False)
        total_steps = chkpt["steps"]
        model.load_state_dict(chkpt["model"])
        # The optimiser and the scheduler are resumed independently.  An `optim` entry this optimiser cannot take
        # (e.g. one written by the reference's torch.optim.AdamW: per-tensor moments, not the flat buffers) only
        # restarts the Adam moments; the OneCycle schedule still continues from step N -- the reference's `sched`
        # entry is a plain torch scheduler state and loads as is, and a missing / unusable one is replaced by
        # fast-forwarding the fresh scheduler to `total_steps`.
        resumed = []
        try:
            optimizer.load_state_dict(chkpt["optim"])
            resumed.append("optimizer")
        except (ValueError, KeyError, TypeError) as e:
            print("optimiser state of %s NOT resumed (Adam moments restart): %s" % (latest, e))
        try:
            scheduler.load_state_dict(chkpt["sched"])
            if int(scheduler.last_epoch) != int(total_steps):
                raise ValueError("scheduler state is at step %d, checkpoint at %d" % (scheduler.last_epoch, total_steps))
            resumed.append("scheduler")
        except (ValueError, KeyError, TypeError, AttributeError) as e:
            print("scheduler state of %s NOT loaded (%s): fast-forwarding OneCycle to step %d" % (latest, e, total_steps))
            scheduler = _fast_forward_scheduler(optimizer, base_lr, cfg["SOLVER"]["MAX_ITER"] + 100, total_steps)
        lr_now = scheduler.get_last_lr()[0]
        for g in optimizer.param_groups:
            g["lr"] = lr_now
        print("Weights%s are loaded from %s, starting from step %d (lr %.6g)" % (
            "".join(", " + r for r in resumed), latest, total_steps, lr_now))
    return model, optimizer, scheduler, total_steps


def _fast_forward_scheduler(optimizer, base_lr, total, steps):
    """A fresh OneCycleLR (libs/train_libs.py:118-120) advanced to `steps` without touching the weights."""
    import warnings
    for g in optimizer.param_groups:          # OneCycleLR's constructor derives initial_lr / max_lr / min_lr again
        for k in ("initial_lr", "max_lr", "min_lr"):
            g.pop(k, None)
    sched = optim.lr_scheduler.OneCycleLR(optimizer, base_lr, total, pct_start=0.05, cycle_momentum=False,
                                          anneal_strategy="linear")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")       # "lr_scheduler.step() before optimizer.step()": intended here
        for _ in range(int(steps)):
            sched.step()
    return sched


def build_model_teacher(cfg, posemodule, device):
    model = posemodule(cfg, build_backbone(cfg["MODEL"]["BACKBONE"]))
    _load_weights(model, cfg["RUNTIME"].get("WEIGHT_FILE", ""))
    model = model.to(device)
    if cfg["RUNTIME"].get("DISTRIBUTED", False) and not cfg["RUNTIME"].get("DEFER_BROADCAST", False):
        # (DEFER_BROADCAST: train_kd.py broadcasts after the step's graphs have been recorded, see there)
        D.broadcast_(model.net.store.params, 0)
        D.broadcast_(model.net.store.bufs, 0)
        model.net.invalidate()
    return model


class DziLoader:
    """DataLoader of full frames -> what the reference's loaders yield after `dzi_train` / `dzi_test`
    (libs/dzi_libs.py:55-140) and `collate_fn`: (ImageList of 256x256 normalised crops, targets with the cropped mask,
    bbox_trans and bbox_scale, meta_infos).  Normalize + the affine crop of image and mask run as ONE kd6d_dzi_crop
    launch per batch on the GPU; the three jitter numbers per image come from numpy's global RNG like the reference.
    The CPU augmentation chain of libs/transform.py is not rebuilt: frames must already have the internal resolution."""

    def __init__(self, loader, cfg, device, training):
        from .dzi_libs import normalize_lut
        self.loader, self.cfg, self.device, self.training = loader, cfg, device, training
        self.lut = normalize_lut(cfg["INPUT"]["PIXEL_MEAN"], cfg["INPUT"]["PIXEL_STD"], device)
        self.size = (cfg["INPUT"]["INTERNAL_HEIGHT"], cfg["INPUT"]["INTERNAL_WIDTH"])

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        from ..kd_losses import PackedTargets
        from .dataset import projected_box
        from .dzi_libs import aug_bbox_DZI, dzi_batch, test_bbox_DZI
        from .poses import ImageList, PoseAnnot
        for frames, masks, targets, metas in self.loader:
            B, H, W, _ = frames.shape
            if (H, W) != tuple(self.size):
                raise ValueError("frames are %dx%d but INPUT.INTERNAL_* is %dx%d: the CPU Resize transform of the reference "
                                 "is not rebuilt, store the frames at the internal resolution" % (W, H, self.size[1], self.size[0]))
            centers, scales = [], []
            for t in targets:
                # to_object_boxlist().bbox[0]; a frame without a known object (the reference would fail on it) is
                # cropped around the whole frame
                box = projected_box(t, 0) if len(t.class_ids) else np.array([0.0, 0.0, float(W), float(H)])
                c, s = aug_bbox_DZI(box, H, W) if self.training else test_bbox_DZI(box, H, W)
                centers.append(c); scales.append(s)
            images, crop_masks, trans, bscale = dzi_batch(frames.to(self.device, non_blocking=True).contiguous(),
                                                          masks.to(self.device, non_blocking=True).contiguous(),
                                                          np.stack(centers), np.asarray(scales), self.lut)
            R = images.shape[-1]
            dev = self.device
            out = [PoseAnnot(t.keypoints_3d.to(dev), t.K.to(dev), crop_masks[i], t.class_ids.to(dev), t.rotations.to(dev),
                             t.translations.to(dev), R, R, bscale[i], trans[i]) for i, t in enumerate(targets)]
            yield ImageList(images, [(R, R)] * B), PackedTargets(out, dev), metas


def build_dataset(cfg, device="cuda"):
    """libs/train_libs.py:209-291: (train_loader, valid_loader) over the BOP image lists of cfg['DATASETS'], batch
    = IMS_PER_BATCH / N_GPU per rank, DistributedSampler semantics of libs/distributed.py.  DATASETS.TRAIN may be one
    list file or several (configs/linemod13.yaml): the datasets are concatenated."""
    from torch.utils.data import ConcatDataset, DataLoader
    from .dataset import BOP_Dataset, collate_frames
    ds = cfg["DATASETS"]

    def make(files, training):
        files = [files] if isinstance(files, str) else list(files)
        sets = [BOP_Dataset(f, ds["MESH_DIR"], ds["BBOX_FILE"], ds.get("SYMMETRY_TYPES"), training=training) for f in files]
        return sets[0] if len(sets) == 1 else ConcatDataset(sets)

    train_set, valid_set = make(ds["TRAIN"], True), make(ds["VALID"], False)
    per_gpu = D.shard_batch(cfg["SOLVER"]["IMS_PER_BATCH"])
    dist_on = cfg["RUNTIME"].get("DISTRIBUTED", False)

    def loader(dset, shuffle):
        if dist_on:
            smp = D.DistributedSampler(dset, shuffle=shuffle)
        elif shuffle:
            smp = torch.utils.data.RandomSampler(dset)
        else:
            smp = torch.utils.data.SequentialSampler(dset)
        return DataLoader(dset, batch_size=per_gpu, sampler=smp, num_workers=cfg["RUNTIME"].get("NUM_WORKERS", 0),
                          collate_fn=collate_frames, drop_last=shuffle)

    return DziLoader(loader(train_set, True), cfg, device, True), DziLoader(loader(valid_set, False), cfg, device, False)


def dataset_meshes(loader):
    """Mesh vertex arrays per class id of a DziLoader's dataset (what valid() measures ADI / REP on)."""
    dset = loader.loader.dataset
    dset = dset.datasets[0] if hasattr(dset, "datasets") else dset
    return [m.vertices for m in dset.meshes]


def start_exchange_after_graphs(gstep, model_t, model, first_batch, log=print):
    """Data-parallel run in the grouped launch mode (train_kd.py --launch pipeline --teacher_group > 1, `between` exchange):
    the step's hipGraphs are recorded BEFORE the process's first RCCL communicator exists (graphs instantiated after
    ncclCommInitRank replay ~18 % slower in that mode).  Order, every rank alike:
      1. gstep.prepare(first batch)      every graph recorded, pipeline left empty, training state untouched
      2. barrier, init_exchange()        the ranks agree, then the kd6d communicator (kd6d_comm_init is collective)
      3. broadcast rank 0's parameters and buffers of teacher and student (what build_model* deferred:
         libs/train_libs.py:123-130 of the reference, the DDP constructor's broadcast)
      4. refresh IN PLACE what the recorded kernels read of them: bf16 shadows, the student's dgrad packing and the
         teacher's eval-mode BatchNorm folds (PoseNet.refresh_derived_in_place; the teacher's segments were recorded on
         its pre-broadcast folds and shadow -- a rank whose initial teacher differed from rank 0's would otherwise distil
         from a different teacher)
    The first batch is then fed as usual by the caller.  Returns the exchange route string."""
    gstep.prepare(first_batch[0], first_batch[1])
    D.synchronize()
    route = D.init_exchange()
    log("gradient exchange: " + route)
    for m in (model_t, model):
        D.broadcast_(m.net.store.params, 0)
        D.broadcast_(m.net.store.bufs, 0)
    model_t.net.refresh_derived_in_place(need_dgrad=False)
    model.net.refresh_derived_in_place(need_dgrad=True)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return route


def stop_if_barrier_timeouts(n_local, distributed):
    """In-kernel barriers of the one-launch BN / GN backward give up after a bounded spin instead of hanging the GPU; a
    wait that gave up means wrong gradients, so training stops.  The decision is COLLECTIVE (MAX over ranks of the
    per-device counter): a rank that stopped alone would leave the others waiting in the next gradient all-reduce.
    Every rank shuts its exchange down and leaves the process group before raising."""
    n_to = D.max_over_ranks(n_local)
    if n_to != 0:
        D.shutdown_exchange()
        if distributed and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        raise SystemExit("kd6d: %d in-kernel barrier waits timed out (gradients of a step are wrong); "
                         "re-run with --two_launch_norm_bwd" % n_to)
    return 0

