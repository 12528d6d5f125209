"""Command line + yaml -> (cfg, cfg_t): the flag surface of arguments/argument_kd.py:15-106.

Every reference flag keeps its name, type and default.  Additive flags of this build:
--precision {bf16,fp32}, --synthetic, --skip_teacher_eval, --batch_size (per-step GLOBAL batch,
overrides SOLVER.IMS_PER_BATCH), --image_size, --mixed_classes; yaml files may name a `_BASE_` file.
"""
import argparse
import os

import yaml

from .argument import custom_cfg


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = v.lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def get_argparser():
    p = argparse.ArgumentParser()
    p.add_argument("--local_rank", type=int, default=0)
    p.add_argument("--config_file", type=str, default="./configs/ape.yaml")
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--working_dir", type=str, default="./outputs/")
    p.add_argument("--test_file", type=str, default="")
    p.add_argument("--weight_file", type=str, default="")
    p.add_argument("--running_device", type=str, default="cuda")
    p.add_argument("--backbone", type=str, default="darknet_tiny_h")
    p.add_argument("--max_iters", type=int, default=20000, help="max iteration")
    p.add_argument("--base_lr", type=float, default=0.001, help="base learning rate")
    # teacher
    p.add_argument("--config_file_t", type=str, default="./configs/occ_linemod.yaml")
    p.add_argument("--backbone_t", type=str, default="darknet53")
    p.add_argument("--weight_file_t", type=str, default="")
    # distillation
    p.add_argument("--kd_weight", type=float, default=5, help="weight of loss_kd_loss")
    p.add_argument("--kd_level", type=str, default="pred", help="level to be distilled")
    p.add_argument("--gtype", type=str, default="sinkhorn", help="function of kd loss",
                   choices=["l1", "l2", "sinkhorn", "gaussian", "laplacian", "energy"])
    p.add_argument("--glevel", type=str, default="point", help="level of kd loss", choices=["point"])
    p.add_argument("--p", type=float, default=2.0, help="p of loss_kd_loss")
    p.add_argument("--blur", type=float, default=0.001, help="blur of loss_kd_loss")
    p.add_argument("--gnD", type=int, default=2, help="dimensions of loss_kd_loss")
    p.add_argument("--weightedOT", type=str2bool, nargs="?", const=True, default=True, help="weighted OT of loss_kd_loss")
    p.add_argument("--wot_detach", type=str2bool, nargs="?", const=True, default=False, help="weighted ot with detached cls")
    p.add_argument("--scaling", type=float, default=0.5, help="param for sinkhorn loss")
    p.add_argument("--reach", type=float, default=0.5, help="param for sinkhorn loss")
    # additive (this build)
    p.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--synthetic", action="store_true", help="seeded LINEMOD-shaped synthetic batches (no dataset)")
    p.add_argument("--skip_teacher_eval", action="store_true")
    p.add_argument("--launch", type=str, default="graph", choices=["graph", "pipeline", "eager"],
                   help="graph: replay the captured step (hipGraph); pipeline: also overlap the teacher forward of "
                        "the next batch with the student step of the current one; eager: launch kernel by kernel")
    p.add_argument("--teacher_group", type=int, default=1,
                   help="--launch pipeline only: run the frozen teacher over the batches of this many consecutive steps in "
                        "one pass, cut into as many graph segments (kd6d.graph.GroupedTeacherKDStep; 2 * group batches "
                        "of look-ahead).  1 = one teacher forward per step")
    p.add_argument("--batch_size", type=int, default=0, help="global batch; 0 = SOLVER.IMS_PER_BATCH")
    p.add_argument("--image_size", type=int, default=256, help="synthetic crop size")
    p.add_argument("--val_freq", type=int, default=0, help="validate every N steps; 0 = the backbone's default")
    p.add_argument("--teacher_pnp_gate", action="store_true",
                   help="keep a teacher's cells only if RANSAC-PnP solves a pose from them (postprocess_kd.py:187-202); "
                        "needs --launch eager (one host synchronisation per step)")
    p.add_argument("--two_launch_norm_bwd", action="store_true",
                   help="BatchNorm / GroupNorm backward as reduce + apply launches instead of one launch with an "
                        "in-kernel barrier (the remedy train_kd.py names when a barrier wait timed out)")
    p.add_argument("--exchange", type=str, default="between", choices=["between", "overlap"],
                   help="data-parallel gradient exchange: one all-reduce between the step's two graphs, or two slices inside "
                        "the step (FPN + head beside the backbone sweep; kd6d/libs/distributed.py EXCHANGE_MODE)")
    p.add_argument("--rccl_single_rank", action="store_true",
                   help="one process, one GPU, but the whole data-parallel path: a one-rank process group, the kd6d "
                        "communicator, the parameter broadcast and every step's all-reduce (rehearsal on a one-GPU box)")
    p.add_argument("--mixed_classes", type=str2bool, nargs="?", const=True, default=None,
                   help="synthetic batches mix the 13 LINEMOD classes (default: DATASETS.MIXED_CLASSES of the yaml)")
    return p


def load_yaml(path):
    """yaml -> dict.  A top-level `_BASE_: other.yaml` (path relative to the file; additive key of this build) is
    loaded first and the file's own sections are merged over it key by key."""
    with open(path, "r") as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader) or {}
    base = cfg.pop("_BASE_", None)
    if base is None:
        return cfg
    out = load_yaml(os.path.join(os.path.dirname(os.path.abspath(path)), base))

    def merge(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                merge(dst[k], v)
            else:
                dst[k] = v
    merge(out, cfg)
    return out


def _runtime(args, config_file, weight_file):
    return dict(LOCAL_RANK=args.local_rank, CONFIG_FILE=config_file, NUM_WORKERS=args.num_workers,
                WEIGHT_FILE=weight_file, RUNNING_DEVICE=args.running_device, PRECISION=args.precision,
                TEACHER_PNP_GATE=bool(args.teacher_pnp_gate),
                TWO_LAUNCH_NORM_BWD=bool(args.two_launch_norm_bwd), EXCHANGE=args.exchange,
                RCCL_SINGLE_RANK=bool(getattr(args, "rccl_single_rank", False)))


def build_cfgs(args):
    cfg = load_yaml(args.config_file)
    cfg["RUNTIME"] = _runtime(args, args.config_file, args.weight_file)
    cfg["RUNTIME"]["WORKING_DIR"] = args.working_dir
    cfg["RUNTIME"]["SYNTHETIC"] = bool(args.synthetic)
    cfg["RUNTIME"]["SKIP_TEACHER_EVAL"] = bool(args.skip_teacher_eval)
    cfg["RUNTIME"]["LAUNCH"] = args.launch
    cfg["RUNTIME"]["TEACHER_GROUP"] = max(1, int(args.teacher_group))
    cfg["RUNTIME"]["IMAGE_SIZE"] = int(args.image_size)
    if args.mixed_classes is not None:
        cfg["DATASETS"]["MIXED_CLASSES"] = bool(args.mixed_classes)
    cfg["DATASETS"].setdefault("MIXED_CLASSES", False)
    if len(args.test_file) > 0:
        cfg["DATASETS"]["TEST"] = args.test_file
    cfg["MODEL"]["BACKBONE"] = args.backbone
    cfg = custom_cfg(cfg)
    cfg["SOLVER"]["MAX_ITER"] = args.max_iters
    cfg["SOLVER"]["BASE_LR"] = args.base_lr
    if args.batch_size > 0:
        cfg["SOLVER"]["IMS_PER_BATCH"] = args.batch_size
    if args.val_freq > 0:
        cfg["SOLVER"]["VAL_FREQ"] = args.val_freq
    cfg.setdefault("KD", {})
    cfg["KD"]["LOSS_WEIGHT_KD"] = args.kd_weight
    cfg["KD"]["LEVEL"] = args.kd_level
    if cfg["KD"]["LEVEL"] == "pred":
        cfg["KD"].update(GLEVEL=args.glevel, GTYPE=args.gtype, GP=args.p, GBLUR=args.blur, GnD=args.gnD,
                         WEIGHTED_OT=args.weightedOT, DETACH=args.wot_detach, SCALING=args.scaling, REACH=args.reach)
    cfg_t = load_yaml(args.config_file_t)
    cfg_t["RUNTIME"] = _runtime(args, args.config_file_t, args.weight_file_t)
    cfg_t["MODEL"]["BACKBONE"] = args.backbone_t
    cfg_t = custom_cfg(cfg_t)
    cfg_t.setdefault("KD", {})
    return cfg, cfg_t


def get_args(argv=None):
    args = get_argparser().parse_args(argv)
    return build_cfgs(args)
