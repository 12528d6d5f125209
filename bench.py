"""KD train-step throughput on MI355X (BASELINE.json metric) -- driver contract in the task brief.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = teacher (Darknet-53, frozen, eval) forward + teacher cell selection + student
(Darknet-tiny-H, train) forward + focal / object-space / Sinkhorn-OT losses + backward +
(RCCL mean all-reduce of the flat gradient bucket) + fused clip + AdamW + OneCycle, on a seeded
synthetic LINEMOD-shaped batch (640x480 frame geometry, 256x256 DZI crops: what the reference
actually feeds the network), B = 16 images per GPU, inputs resident in HBM.

Default launch mode on one rank (--teacher-group 3, kd6d.graph.GroupedTeacherKDStep): the frozen teacher runs over the 48
images of three consecutive steps in one pass, cut into three hipGraph segments of equal device time, one replayed on the
teacher's stream beside each student step; every batch still gets exactly one teacher forward and one student step.  The timed
region is aligned so that it ENDS with a completed pass; every timed call replays exactly one teacher segment (1 / group
of a pass), so the teacher work inside the region is K segments = K * B images (`config.teacher_images_in_timed_region`;
`teacher_passes_completed_in_timed_region` counts the passes that END inside it -- the first of them began before it).
--teacher-group 1 = one teacher forward per step (rounds 1-2).
Under a process group (N > 1, or the one-rank rehearsal --rccl-single-rank) every graph is recorded from a sample batch
BEFORE the RCCL communicator is created (GroupedTeacherKDStep.prepare), then the parameters are
broadcast and the timed steps exchange their gradients between the step's two graphs.

`python bench.py --gpus N` without WORLD_SIZE in the environment launches its own N ranks (one child process per
GPU, started before this process makes any GPU call; rank 0's line is passed through, a failing rank fails the run).

The single JSON line carries, besides the contract fields:
  roofline     -- dominant kernel family = the implicit-GEMM convolutions (kd6d_conv2d_{fwd,dgrad,wgrad} and the
                  grouped weight gradient): `achieved` = algorithmic conv FLOP of a step (SURVEY.md 8(d), 2*MAC on the
                  reference's channel counts) / WALL ms_per_step of the timed, hipGraph-replayed steps, per GPU;
                  peak = 2.5 PFLOP/s dense bf16 MFMA.  `eager_launch_events` keeps the per-launch HIP-event figures of
                  a few eagerly launched single-stream steps after the timed region (a different schedule: per-family
                  kernel efficiency, not step time).  `traffic` = HBM bytes of the conv family per step from the
                  rocprofv3 PMC passes committed under profiles/ (tools/pmc_traffic.sh), null when that file does not
                  describe the configuration being run.
  cpu_baseline -- oracle/kd_step_ref.py (pure-torch fp32 port of the reference step, validated
                  against the imported reference) timed on this host's cores, rank 0, N=1 only.
  secondary    -- (N=1, default workload only) 20-step timings of the other BASELINE configurations, each run by this
                  same script in a child process AFTER the headline's timed region: config 4's per-GPU shard
                  (--workload linemod13), full 480x640 frames (--frame full640), one teacher forward per step
                  (--teacher-group 1), strictly sequential steps (--no-pipeline), the one-rank rehearsal of the data-parallel
                  path (--rccl-single-rank) and the dense 16-D OT of config 5 (--workload dense16d).  The headline does not
                  depend on them: a failing child is reported inside its entry.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "kd-6d-pose-adlp_amd"))

import torch  # noqa: E402

FLOP_PER_IMG = {  # teacher fwd + 3 x student fwd, 2*MAC of all convs
    ("darknet_tiny_h", 256): 49.2e9, ("darknet_tiny", 256): 86.9e9,
    ("darknet_tiny_h", 640): 230.6e9, ("darknet_tiny", 640): 407.4e9,
}
PEAK_BF16 = 2.5e15
PEAK_F32 = 157.3e12


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=16, help="images per GPU")
    p.add_argument("--workload", type=str, default="ape", choices=["ape", "linemod13", "dense16d"],
                   help="ape: BASELINE configs 2/3 (configs/ape.yaml, the metric's configuration, default); linemod13: "
                        "config 4 (configs/linemod13.yaml: the 13 LINEMOD classes mixed in a batch, darknet_tiny "
                        "student); dense16d: config 5 (configs/dense16d.yaml: the OT loss alone on a 128x128 grid of "
                        "16-D codes, one problem per image)")
    p.add_argument("--student", type=str, default="")
    p.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--frame", type=str, default="crop256", choices=["crop256", "full640"])
    p.add_argument("--exchange", type=str, default="between", choices=["between", "overlap"],
                   help="data-parallel exchange schedule: one all-reduce between the step's two graphs (default), or two "
                        "slices captured inside the single step graph, the FPN + head slice beside the backbone sweep "
                        "(kd6d/libs/distributed.py EXCHANGE_MODE)")
    p.add_argument("--rccl-single-rank", action="store_true",
                   help="under `torch.distributed.run --nproc-per-node 1`: run the N > 1 code path (RCCL communicator, "
                        "the all-reduce between the two graphs, barriers) on one GPU")
    p.add_argument("--opt", action="append", default=[],
                   help="kernel-selection option name=value (kd6d_set_option, include/kd6d.h): tuning aid for A/B runs")
    p.add_argument("--fuse-norm", type=str, default="", choices=["", "none", "teacher", "student", "both"],
                   help="A/B aid: which network's convolutions take the conv + normalisation launch (kd6d_conv2d_fwd_norm); "
                        "default = the engine's measured choice")
    p.add_argument("--bn-on-load", type=int, default=-1,
                   help="A/B aid: the student's in-stage BatchNorm + LeakyReLU applied by the next block's convolution while it "
                        "loads (kd6d_conv2d_fwd_block): 0 never, 1 where the next block is a 1x1 convolution, 2 every in-stage "
                        "transition; default = the engine's measured choice")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true", help="skip the `secondary` timings (child runs of the other configs)")
    p.add_argument("--no-launch-events", action="store_true",
                   help="skip the eagerly launched, HIP-event-instrumented steps behind roofline.eager_launch_events")
    p.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying hipGraphs")
    p.add_argument("--teacher-group", type=int, default=0,
                   help="pipelined launch only: run the frozen teacher over the batches of this many consecutive steps at "
                        "once (kd6d.graph.GroupedTeacherKDStep); 1 = one teacher forward per step; 0 (default) = 3, "
                        "except with --exchange overlap under a process group (1): collectives captured inside the step "
                        "graph need the communicator before the capture, and graphs recorded after an RCCL communicator "
                        "was created replay 18 %% slower in the grouped mode (DESIGN.md section 7)")
    p.add_argument("--debug-skip-teacher", type=int, default=0,
                   help="timing experiment, INVALID as a result (reported as such): 1 = replay no teacher segment, 2 = every "
                        "second one -- what the student's steps cost without / with half of the teacher beside them")
    p.add_argument("--tune", action="append", default=[],
                   help="schedule experiment, name=value: budget_div (CUs / this = the weight gradients' split-K budget), "
                        "group_wgs (workgroups of the grouped weight gradient), group_flush (head_end|fpn_end), streams "
                        "(weight-gradient side streams), fuse_stats_rows (largest layer whose BatchNorm sums come out of the "
                        "convolution's epilogue instead of a colstats launch)")
    p.add_argument("--no-pipeline", action="store_true",
                   help="do not overlap the teacher forward of batch k+1 with the student step of batch k")
    p.add_argument("--cpu-steps", type=int, default=4)
    p.add_argument("--layer-table", type=str, default="", help="write the per-launch conv table (instrumented steps) here")
    p.add_argument("--timeline", action="store_true",
                   help="analysis aid: capture device timestamp markers into the graph and print the phase boundaries of the "
                        "last replayed step to stderr (adds ~15 one-thread launches per step)")
    return p.parse_args()


WORKLOAD_YAML = {"ape": "ape.yaml", "linemod13": "linemod13.yaml", "dense16d": "dense16d.yaml"}


def make_cfg(arch, precision, workload="ape"):
    from kd6d.arguments.argument import custom_cfg
    from kd6d.arguments.argument_kd import load_yaml
    cfg = load_yaml(os.path.join(HERE, "configs", WORKLOAD_YAML[workload]))
    cfg["RUNTIME"] = {"PRECISION": precision}
    cfg["MODEL"]["BACKBONE"] = arch
    cfg = custom_cfg(cfg)
    cfg["KD"] = dict(LOSS_WEIGHT_KD=5.0, LEVEL="pred", GLEVEL="point", GTYPE="sinkhorn", GP=2.0, GBLUR=0.001, GnD=2,
                     WEIGHTED_OT=True, DETACH=False, SCALING=0.5, REACH=0.5)
    return cfg


TEACHER_CLS_BIAS = [1.0] + [-6.0] * 14


class _StdoutToStderr:
    """The contract is ONE JSON line on stdout; librccl prints a version banner there when a communicator is created
    (torch's and ours alike).  File descriptor 1 points at stderr until the JSON line is written."""

    def __init__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        print(line, flush=True)


def _gpu_numa_cpus(local_rank, world):
    """CPUs next to GPU `local_rank`, found WITHOUT touching the GPU: the kfd topology in sysfs lists the GPU nodes in
    HIP's device order and, per node, an io_link to its host NUMA node.  Falls back to an even split of the CPUs this
    process may run on."""
    allowed = sorted(os.sched_getaffinity(0))
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        gpus = []
        for n in sorted(os.listdir(base), key=int):
            with open(os.path.join(base, n, "properties")) as f:
                props = dict(l.split()[:2] for l in f if len(l.split()) >= 2)
            if int(props.get("simd_count", 0)) > 0:
                gpus.append(n)
        node = gpus[local_rank]
        cpu_node = None
        links = os.path.join(base, node, "io_links")
        for l in sorted(os.listdir(links), key=int):
            with open(os.path.join(links, l, "properties")) as f:
                lp = dict(x.split()[:2] for x in f if len(x.split()) >= 2)
            with open(os.path.join(base, lp["node_to"], "properties")) as f:
                tp = dict(x.split()[:2] for x in f if len(x.split()) >= 2)
            if int(tp.get("cpu_cores_count", 0)) > 0:
                cpu_node = int(lp["node_to"])
                break
        with open("/sys/devices/system/node/node%d/cpulist" % cpu_node) as f:
            cpus = set()
            for part in f.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        near = [c for c in allowed if c in cpus]
        if near:
            return near
    except (OSError, ValueError, KeyError, IndexError, TypeError):
        pass
    per = max(len(allowed) // max(world, 1), 1)
    return allowed[local_rank * per:(local_rank + 1) * per] or allowed


def pin_host_thread(local_rank, world):
    """Eight ranks on one host each issue ~1 ms of launches per 3-ms step: keep every rank's host thread on the
    cores of its GPU's NUMA node.  Called before any GPU call; a single rank is left alone."""
    if world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = _gpu_numa_cpus(local_rank, world)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return len(cpus)


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as child processes (env RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_*, what torch.distributed.run would set) BEFORE this process makes any GPU call,
    pass rank 0's JSON line through, and exit with the first non-zero return code.  The reference does the same through
    `torch.distributed.launch` (train.sh; train_kd.py:43-51 reads WORLD_SIZE / --local_rank)."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(n))
    out0 = b""
    import select
    while pending:
        # drain rank 0's stdout so that it never blocks on a full pipe
        if procs[0].stdout is not None and 0 in pending:
            ready, _, _ = select.select([procs[0].stdout], [], [], 0.2)
            if ready:
                chunk = os.read(procs[0].stdout.fileno(), 1 << 16)
                out0 += chunk
        else:
            time.sleep(0.2)
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:                 # the others would wait for the dead rank in a collective
                    procs[q].terminate()
    out0 += procs[0].stdout.read() or b""
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    raise SystemExit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    world_env = int(os.environ.get("WORLD_SIZE", 1))
    if world_env > 1:
        n_vis = torch.cuda.device_count()          # (does not initialise the GPU)
        if n_vis < world_env:
            raise SystemExit("bench.py: %d GPUs needed, %d visible (rank %s)" % (world_env, n_vis, os.environ.get("RANK", "0")))
    pinned = pin_host_thread(int(os.environ.get("LOCAL_RANK", 0)), world_env)
    out_fd = _StdoutToStderr()
    if args.workload == "dense16d":
        return bench_dense(args, out_fd)
    from kd6d.arguments.argument_kd import load_yaml
    cfg_w = load_yaml(os.path.join(HERE, "configs", WORKLOAD_YAML[args.workload]))
    mixed = bool(cfg_w["DATASETS"].get("MIXED_CLASSES", False))
    if not args.student:
        args.student = cfg_w["MODEL"]["BACKBONE"]
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the kd6d step has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_pg = world > 1 or (args.rccl_single_rank and "MASTER_ADDR" in os.environ)
    if use_pg:
        dist.init_process_group(backend="nccl", init_method="env://")

    from kd6d import backbone as BB, ops
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs import distributed as D
    for kv in args.opt:
        ops.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    from kd6d.models.model_kd import PoseModuleKD
    from kd6d.optim import FusedClipAdamW
    from kd6d.synthetic import make_batch

    teacher = PoseModuleKD(make_cfg("darknet53", args.precision), BB.darknet53())
    teacher.net.reset_parameters(seed=2)
    with torch.no_grad():
        sd = teacher.state_dict()
        sd["head.cls_logits.bias"] = torch.tensor(TEACHER_CLS_BIAS)
        # random-init BN statistics: make the frozen teacher's activations well-scaled
        teacher.load_state_dict(sd)
    teacher = teacher.to(dev).eval()
    student = PoseModuleKD(make_cfg(args.student, args.precision), getattr(BB, args.student)())
    student.net.reset_parameters(seed=1)
    student = student.to(dev).train()
    if args.bn_on_load >= 0:
        student.net.bn_on_load = args.bn_on_load
    if args.fuse_norm:
        teacher.net.fuse_norm = args.fuse_norm in ("teacher", "both")
        student.net.fuse_norm = args.fuse_norm in ("student", "both")
    D.SINGLE_RANK_EXCHANGE = bool(args.rccl_single_rank)
    D.EXCHANGE_MODE = args.exchange
    if args.teacher_group == 0:
        # default: the grouped teacher pass; beside a gradient exchange only in its "between" schedule (a collective
        # captured INSIDE the step's graph needs the communicator before the capture, which this launch mode pays 18 % for)
        args.teacher_group = 1 if (use_pg and args.exchange == "overlap") else 3
    grouped = args.teacher_group > 1 and not args.no_pipeline and not args.no_graph

    def start_exchange():
        r = D.init_exchange()                   # kd6d_comm_* over librccl (include/kd6d.h)
        D.broadcast_(student.net.store.params, 0)
        student.net.invalidate()
        return r

    route = "none"
    if use_pg and not grouped:
        route = start_exchange()
    base_lr = 1e-3 / world                      # libs/train_libs.py:117
    opt = FusedClipAdamW(student, lr=base_lr, weight_decay=1e-4, eps=1e-8, max_norm=1.0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, base_lr, 10100, pct_start=0.05, cycle_momentum=False,
                                                anneal_strategy="linear")
    full = args.frame == "full640"
    B = args.batch
    batches = []
    for i in range(4):
        images, targets = make_batch(B, 1000 * rank + i, full_frame=full, mixed_classes=mixed, class_offset=rank * B)
        batches.append((images.to(dev), PackedTargets(targets, dev)))

    from kd6d.graph import GraphedKDStep
    tune = dict(kv.split("=") for kv in args.tune)
    if "streams" in tune:
        GraphedKDStep.WGRAD_STREAMS = int(tune["streams"])
    if "fuse_stats_rows" in tune:
        from kd6d import engine as _engine
        _engine.ConvBlock.FUSE_STATS_MAX_ROWS = int(tune["fuse_stats_rows"])
    if args.no_graph:
        gstep = None
    else:
        if args.teacher_group > 1 and not args.no_pipeline:
            from kd6d.graph import GroupedTeacherKDStep
            gstep = GroupedTeacherKDStep(teacher, student, opt, (0.1, 1.0, 5.0), group=args.teacher_group)
        else:
            gstep = GraphedKDStep(teacher, student, opt, (0.1, 1.0, 5.0), pipeline=not args.no_pipeline)
    group = getattr(gstep, "group", 1)
    if use_pg and grouped:
        # graphs first, communicator second (GroupedTeacherKDStep.prepare): every graph is recorded from a sample batch
        # before the first RCCL communicator of the process exists; the parameters are broadcast afterwards and the bf16
        # shadow / dgrad packing the recorded kernels read are refreshed eagerly
        # the SAME function train_kd.py runs (and tests/test_distributed_cpu.py drives at world size 2 on gloo): prepare ->
        # communicator -> broadcast of both networks' parameters and buffers -> in-place refresh of everything the recorded
        # kernels read that is derived from them
        from kd6d.libs.train_libs import start_exchange_after_graphs
        route = start_exchange_after_graphs(gstep, teacher, student, batches[0], log=lambda m: print(m, file=sys.stderr))
    if gstep is not None:
        if "budget_div" in tune:
            student.net.wgrad_cu_budget = int(ops.device_cu_count() / float(tune["budget_div"]))
        if "group_wgs" in tune:
            student.net.wgrad_group_wgs = int(tune["group_wgs"])
        if "group_flush" in tune:
            student.net.wgrad_group_flush = tune["group_flush"]
    if args.debug_skip_teacher and group > 1:
        gstep._debug_skip_teacher = args.debug_skip_teacher
    if args.timeline and gstep is not None:
        ops.marks_begin(dev)                     # before the capture: the markers become graph nodes
    n_prime = 0
    if gstep is not None and gstep.pipeline:
        # priming call(s): teacher only, no student step yet
        n_prime = 1 if group == 1 else 2 * group      # (grouped: one period loading, one with the teacher on it)
        for j in range(n_prime):
            assert gstep(*batches[j % len(batches)]) is None

    def step(i, eager=False):
        images, tgt = batches[i % len(batches)]
        if gstep is not None and not eager:
            # pipelined: this call runs the teacher on batch i+1 beside the student step on batch i
            # (one teacher forward and one student step per call either way)
            ld = gstep(*batches[(i + n_prime) % len(batches)]) if gstep.pipeline else gstep(images, tgt)
            sched.step()
            return ld
        student._defer_allreduce = False
        student.net.side_stream = None        # per-launch HIP-event timing wants one kernel at a time
        student.net.side_streams = None
        student.zero_grad()
        with torch.no_grad():
            pred_t = teacher(images, targets=tgt, is_teacher=True)
        _, ld = student(images, targets=tgt, pred_t=pred_t)
        loss = ld["loss_cls"] * 0.1 + ld["loss_reg"] * 1.0 + ld["loss_kd"] * 5.0
        loss.backward()
        opt.step()
        sched.step()
        return ld

    # grouped teacher: every `group`-th call carries the teacher pass over `group` batches.  The timed region must not
    # undercount it: extra untimed calls shift the phase so that the LAST timed call is one with a pass, i.e. the region
    # holds ceil(steps / group) passes (>= steps * B images through the teacher)
    n_align = (-(args.warmup + args.steps)) % group
    for i in range(args.warmup + n_align):
        step(i)
    step_base = args.warmup + n_align
    passes0 = getattr(gstep, "teacher_passes", 0)
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ld = step(step_base + i)
    t_enqueued = time.perf_counter() - t0          # host-side launch time (the GPU runs behind)
    passes_timed = getattr(gstep, "teacher_passes", 0) - passes0
    torch.cuda.synchronize()
    if use_pg:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [elapsed / args.steps * 1e3]
    rccl_ranks = 1
    if use_pg:
        rccl_ranks = dist.get_world_size()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(t) for _ in range(rccl_ranks)]
        dist.all_gather(every, t)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]
        elapsed = max(float(x.item()) for x in every)          # the contract's MAX over ranks
    losses = {k: float(v) for k, v in ld.items()}
    finite = all(v == v and abs(v) != float("inf") for v in losses.values())
    # the one-launch BN / GN backward kernels wait at in-kernel barriers with a bounded spin: a wait that gave up
    # means wrong gradients in the timed steps, so the run is invalid (reported in the JSON line, non-zero exit)
    barrier_timeouts = int(ops.lib.kd6d_barrier_timeouts())
    if args.timeline and gstep is not None and rank == 0:
        buf, slots = ops.marks_end()
        t = buf.cpu().tolist()
        t0 = t[slots["step.start"]]
        for name, idx in sorted(slots.items(), key=lambda kv: t[kv[1]]):
            print("timeline %-28s %9.1f us" % (name, (t[idx] - t0) * 0.01), file=sys.stderr)
    # host cost of issuing one step onto an idle GPU (in the timed loop the host mostly waits for queue space)
    t_issue = []
    for j in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(args.warmup + args.steps + j)
        t_issue.append(time.perf_counter() - t1)
    torch.cuda.synchronize()

    # ---- roofline leg: per-launch HIP-event timing of every conv launch, in instrumented steps that
    # every rank runs (the step contains the gradient all-reduce) but only rank 0 records ----
    n_instr = 0 if args.no_launch_events else 3
    if rank == 0:
        ops.profile_begin()
    for i in range(n_instr):
        # eager launches cost the host ~15 us each, more than most of these kernels run: park the GPU behind a spin
        # kernel while the host queues the step, so that the event pairs bracket back-to-back kernels, not launch gaps
        torch.cuda._sleep(int(2.0e9 * 0.010))
        step(args.warmup + args.steps + i, eager=True)
    rec = ops.profile_end() if rank == 0 else []

    out = None
    if rank == 0 and args.layer_table and n_instr:
        write_layer_table(args.layer_table, rec, n_instr)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        size = 640 if full else 256
        flop_img = FLOP_PER_IMG.get((args.student, size))
        conv = [(k, f, ms_) for (k, f, ms_, _t) in rec if k.startswith("conv")]
        tot_flop = sum(f for _, f, _ in conv)
        tot_ms = sum(m for _, _, m in conv)
        by_kind = {}
        for k, f, m in conv:
            a = by_kind.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1; a[1] += f; a[2] += m
        peak = PEAK_BF16 if args.precision == "bf16" else PEAK_F32
        achieved = tot_flop / (tot_ms * 1e-3) if tot_ms > 0 else 0.0
        traffic = None
        import glob
        tfiles = sorted(glob.glob(os.path.join(HERE, "profiles", "r*_conv_hbm_traffic.json")))
        tpath = tfiles[-1] if tfiles else ""          # the latest round's PMC passes (tools/pmc_traffic.sh on the closing state)
        if (tpath and args.student == "darknet_tiny_h" and not full and B == 16 and args.precision == "bf16" and not args.opt
                and group == 3):      # (the file describes the default schedule: grouped teacher pass, group 3)
            # HBM bytes of the conv family per step from rocprofv3 PMC passes (tools/pmc_traffic.sh, FETCH_SIZE x2
            # per the gfx950 correction + WRITE_SIZE), committed with the profile it was taken from
            with open(tpath) as f:
                traffic = json.load(f)["conv_family_hbm_MB_per_step"] * 1e6
        # Headline: the algorithmic conv FLOPs of a step (2*MAC of every convolution, teacher forward +
        # 3 x student forward) over the WALL time of the timed, replayed steps -- the schedule `value` is measured on.
        # `eager_launch_events` is a different schedule (each kernel alone on one stream, weight gradients sized for the
        # whole device, HIP events around every launch): per-family kernel efficiency, not step time.  In the replayed
        # step up to 6 streams overlap, so summed kernel time may exceed ms_per_step; the per-family table of the
        # REPLAYED step is the rocprofv3 summary under profiles/ (README there).
        wall_tflops = (value * flop_img / world) if flop_img else None          # per GPU
        roof = {"bound": "mfma", "kernel": "implicit-GEMM convolutions of the step (fwd + dgrad + wgrad)",
                "basis": "algorithmic conv FLOP per step / wall ms_per_step of the timed hipGraph-replayed steps, per GPU",
                "achieved": wall_tflops / 1e12 if wall_tflops else achieved / 1e12, "peak": peak / 1e12,
                "unit": "TFLOP/s", "frac": (wall_tflops if wall_tflops else achieved) / peak,
                "flop_per_step": flop_img * B if flop_img else tot_flop / max(n_instr, 1),
                "traffic": traffic,
                "traffic_unit": "HBM bytes per step, conv family (rocprofv3 PMC passes of the round's closing state: %s)" % (
                    os.path.join("profiles", os.path.basename(tpath)) if tpath else "none"),
                "eager_launch_events": {
                    "note": "HIP events around every conv launch of %d eagerly launched single-stream steps after the "
                            "timed region; NOT the timed schedule" % n_instr,
                    "tflops": achieved / 1e12, "frac": achieved / peak, "launches_per_step": len(conv) // max(n_instr, 1),
                    "avg_launch_us": 1e3 * tot_ms / max(len(conv), 1), "kernel_ms_per_step_serial": tot_ms / max(n_instr, 1),
                    "by_kind": {k: {"launches_per_step": v[0] // max(n_instr, 1),
                                    "tflops": v[1] / (v[2] * 1e-3) / 1e12 if v[2] else 0,
                                    "kernel_ms_per_step_serial": v[2] / max(n_instr, 1)} for k, v in by_kind.items()}}}
        if n_instr == 0:
            roof.pop("eager_launch_events")
        out = {"metric": "KD train-step images/sec (teacher+student fwd + OT loss + bwd + AdamW)", "value": value,
               "unit": "images/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "ms_per_step_per_rank": per_rank_ms, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
               "data": "synthetic",
               "config": {"workload": "%s KD: darknet53 teacher -> %s student, kd_weight=5, %s, batch=%d/GPU, "
                                      "%s" % ("13 LINEMOD classes mixed per batch," if mixed else "Ape",
                                              args.student, args.precision, B,
                                              "480x640 full frames" if full else "640x480 frames, 256x256 DZI crops"),
                          "global_batch": B * world, "parallelism": "dp%d" % world, "exchange": route,
                          "exchange_schedule": (D.EXCHANGE_MODE if use_pg else "none"),
                          "launch": "eager" if gstep is None else ("hipGraph replay (%d graph%s/step)" % (
                              gstep.graphs_per_step, "" if gstep.graphs_per_step == 1 else "s") + (
                              "" if not gstep.pipeline else (", teacher(k+1) overlapped with student step(k)" if group == 1 else
                                                             ", teacher over the %d batches of steps k+%d..k+%d in one pass every "
                                                             "%d steps, beside a student step" % (group, 1, group, group)))),
                          "teacher_group": group,
                          "teacher_passes_completed_in_timed_region": passes_timed if group > 1 else args.steps,
                          "teacher_segments_in_timed_region": args.steps,          # one per call: 1 / group of a pass
                          "teacher_images_in_timed_region": args.steps * B,
                          "weights": "random-init (seeded), teacher cls bias set so ~10 cells/img pass 0.1"},
               "losses_last_step": losses, "finite": finite and not args.debug_skip_teacher, "barrier_timeouts": barrier_timeouts,
               "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
               "host_issue_ms_idle_gpu": min(t_issue) * 1e3,
               "roofline": roof}
        # ---- CPU baseline leg (oracle = port of the reference step), rank 0, N=1 only ----
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, B, full)
        if pinned:
            out["config"]["host_cpus_per_rank"] = pinned
        default_run = (args.workload == "ape" and not full and not args.no_pipeline and not args.no_graph and args.teacher_group == 3
                       and args.student == "darknet_tiny_h" and args.precision == "bf16" and not args.opt and not args.tune)
        if world == 1 and default_run and not args.no_secondary:
            out["secondary"] = secondary_runs(args)
    if use_pg:
        dist.barrier()
        D.shutdown_exchange()
        dist.destroy_process_group()
    barrier_timeouts = max(barrier_timeouts, int(ops.lib.kd6d_barrier_timeouts()))
    if rank == 0:
        out["barrier_timeouts"] = barrier_timeouts
        out_fd.emit(json.dumps(out))
    if barrier_timeouts != 0 or not finite:
        raise SystemExit("bench.py: INVALID run (barrier_timeouts=%d, finite=%s)" % (barrier_timeouts, finite))


def bench_dense(args, out_fd):
    """BASELINE config 5 (configs/dense16d.yaml): the OT distribution-alignment loss on a dense grid of local
    predictions, one problem per image -- loss value + d/dx + d/dalpha per step, N = M = GRID_H * GRID_W points of
    CODE_DIM dimensions.  One rank per GPU runs its own images (no collective: the problems are independent)."""
    import numpy as np
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the kd6d step has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="nccl", init_method="env://")
    from kd6d import ops
    from kd6d.arguments.argument_kd import load_yaml
    for kv in args.opt:
        ops.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    kd = load_yaml(os.path.join(HERE, "configs", "dense16d.yaml"))["KD_DENSE"]
    N = M = int(kd["GRID"][0]) * int(kd["GRID"][1])
    D = int(kd["CODE_DIM"])
    r = np.random.default_rng(1000 * rank)
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))          # noqa: E731  (codes / scores = sigmoid(N(0,2)))
    mk = lambda *shape: torch.from_numpy(sig(r.normal(0, 2, shape)).astype(np.float32)).to(dev)     # noqa: E731
    x, y, a, b = mk(N, D), mk(M, D), mk(N), mk(M)
    results = []
    steps = max(1, min(args.steps, 20))
    for blur in kd["BLUR"]:
        diam = float(torch.sqrt(((torch.maximum(x.max(0).values, y.max(0).values)
                                  - torch.minimum(x.min(0).values, y.min(0).values)) ** 2).sum()))
        n_eps = 2 + int(np.ceil((2 * np.log(blur) - 2 * np.log(diam)) / (2 * np.log(kd["SCALING"]))))
        passes = 4 * (1 + n_eps + 1)
        run = lambda: ops.sinkhorn_dense(x, a, y, b, blur=blur, scaling=kd["SCALING"], reach=kd["REACH"],   # noqa: E731
                                         diameter=diam, p=kd["P"])
        for _ in range(max(1, min(args.warmup, 3))):
            run()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, gx, ga = run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        ms = el / steps * 1e3
        pairs = passes * float(N) * float(M)
        # which passes run where (csrc/sinkhorn_dense.hip): with D = 16 the gradient-free softmin passes go to the matrix
        # pipe (inner products as six bf16 MFMAs on three-way split operands: 6 x 2 x 16 FLOP per pair) while
        # eps >= 1.5e-4 diameter^2; the small-eps steps and the last, gradient-carrying extrapolation keep the
        # difference form on the fp32 vector pipe ((2D + 8) lane-ops per pair, an FMA counted once)
        eps_list = [diam * diam] + [float(np.exp(2 * np.log(diam) + i * 2 * np.log(kd["SCALING"]))) for i in range(n_eps - 2)] + [blur * blur]
        on_mfma = D == 16 and ops.get_option("sinkhorn.dense_mfma") != 0
        thr = 0.0 if ops.get_option("sinkhorn.dense_mfma") == 2 else 1.5e-4 * diam * diam
        # (the last extrapolation's four softmins go there too: the two without a gradient as they are, the two with one
        #  with the weighted sums as a second product -- four more MFMAs of the same shape per 32 x 32 pairs, 128 FLOP per
        #  pair; option 3 keeps those two in the difference form)
        last_on = eps_list[-1] >= thr
        grad_on = last_on and ops.get_option("sinkhorn.dense_mfma") != 3
        n_mfma = (4 * (1 + sum(1 for e in eps_list if e >= thr)) + (2 if last_on else 0) + (2 if grad_on else 0)) if on_mfma else 0
        # below the rule (round 4): SCREENED passes -- the same six products give approximate exponents, one maximum chain per
        # tile decides which pairs can contribute at all, only those are evaluated in the difference form
        # (dense_softmin_screen_kernel; option sinkhorn.dense_screen = 0: every pair in the difference form)
        screened = on_mfma and ops.get_option("sinkhorn.dense_screen") != 0
        n_screen = (4 * sum(1 for e in eps_list if e < thr) + (0 if last_on else 4)) if screened else 0
        n_diff = passes - n_mfma - n_screen
        mfma_flop = ((n_mfma + n_screen) * 6 * 2 * D + (2 * 4 * 2 * D if (on_mfma and grad_on) else 0)) * float(N) * float(M)
        laneops = n_diff * float(N) * float(M) * (2 * D + 8)
        peak = PEAK_F32 / 2                         # lane-ops/s of the fp32 vector pipes (157.3 TFLOP/s counts FMA twice)
        # ALGORITHMIC work: 2 D flops of the inner product + ~8 of the exp / running logsumexp per pair
        algo_flop = pairs * (2 * D + 8)
        results.append({"blur": blur, "diameter": diam, "eps_steps": n_eps, "softmin_passes": passes,
                        "passes_matrix_pipe": n_mfma, "passes_screened": n_screen, "passes_difference_form": n_diff,
                        "ms_per_image": ms, "images_per_s": world * 1e3 / ms, "pairs_per_s": pairs / (ms * 1e-3),
                        "algorithmic_tflops": algo_flop / (ms * 1e-3) / 1e12,
                        "issued_mfma_tflops_over_whole_time": mfma_flop / (ms * 1e-3) / 1e12,
                        "frac_of_fp32_vector_peak_difference_form_passes": laneops / (ms * 1e-3) / peak, "loss": float(loss),
                        "finite": bool(torch.isfinite(loss).all() and torch.isfinite(gx).all() and torch.isfinite(ga).all())})
    if world > 1:
        dist.destroy_process_group()
    if rank != 0:
        return
    head = results[0]
    out_fd.emit(json.dumps({
        "metric": "dense-OT KD loss images/sec (Sinkhorn divergence value + gradients, one 128x128x16-D problem per image)",
        "value": head["images_per_s"], "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": min(args.warmup, 3),
        "ms_per_step": head["ms_per_image"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 5 (configs/dense16d.yaml): N = M = %d cells, D = %d, p=2, blur %s, scaling %s, "
                               "reach %s; the reference cannot run this size (geomloss needs KeOps above 5000^2 pairs)"
                               % (N, D, head["blur"], kd["SCALING"], kd["REACH"]), "parallelism": "replicas x%d" % world},
        # roofline.frac counts ALGORITHMIC flops (40 per pair at D = 16) against the roof SURVEY.md 8(d) names for this
        # path, the fp32 vector pipe (157.3 TFLOP/s): a fraction above what the difference form can reach means the bound was
        # sidestepped -- the inner products moved to the matrix pipe and, in the screened passes, the exact evaluation of the
        # pairs that cannot contribute skipped -- not beaten.  What the matrix pipe ISSUES for it (192 FLOP per pair: six bf16
        # MFMAs on three-way split fp32 operands) is kept under its own key, against its own peak.
        "roofline": {"bound": "valu-fp32 (SURVEY.md 8(d): online logsumexp, costs never stored)",
                     "kernel": "sinkhorn_dense: dense_softmin_mfma_kernel / dense_softmin_mfma_grad_kernel (matrix-pipe passes), "
                               "dense_softmin_screen_kernel (small-eps passes: matrix-pipe screen + exact difference form for the "
                               "pairs that pass), dense_softmin_kernel (difference form for every pair)",
                     "achieved": head["algorithmic_tflops"], "peak": PEAK_F32 / 1e12, "unit": "TFLOP/s",
                     "frac": head["algorithmic_tflops"] * 1e12 / PEAK_F32, "traffic": None,
                     "basis": "(2 D + 8) = 40 FLOP per (row, column) pair x %d softmin passes x N M pairs / WALL time of one image; "
                              "%d passes on the matrix pipe, %d screened on it (exact terms only for pairs within 40 + the error "
                              "bound of a row's running maximum), %d with every pair in the difference form"
                              % (head["softmin_passes"], head["passes_matrix_pipe"], head["passes_screened"], head["passes_difference_form"]),
                     "issued_mfma": {"achieved": head["issued_mfma_tflops_over_whole_time"], "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                                     "frac": head["issued_mfma_tflops_over_whole_time"] * 1e12 / PEAK_BF16,
                                     "basis": "192 FLOP per pair of the matrix-pipe and screened passes (+ 128 of the two gradient-carrying ones on the matrix pipe) / wall time"},
                     "algorithmic_frac_of_bf16_mfma_peak": head["algorithmic_tflops"] * 1e12 / PEAK_BF16},
        "all_blurs": results, "finite": all(r_["finite"] for r_ in results)}))


SECONDARY = [          # (name, BASELINE.json config it stands for, extra flags)
    ("linemod13", "config 4 per-GPU shard: 13 LINEMOD classes mixed per batch, darknet53 -> darknet_tiny", ["--workload", "linemod13"]),
    ("full640", "S640 variant of config 2: (16,3,480,640) full frames", ["--frame", "full640"]),
    ("teacher_group1", "config 2, one teacher forward per step beside the previous batch's student step (the launch mode of "
                       "rounds 1-2)", ["--teacher-group", "1"]),
    ("no_pipeline", "config 2, strictly sequential steps (teacher and student of the SAME batch in one step)", ["--no-pipeline"]),
    ("dense16d", "config 5: dense 16-D OT over a 128x128 cell grid", ["--workload", "dense16d"]),
    ("rccl_rehearsal_1rank", "config 3's code path on ONE rank: process group, graphs recorded before the RCCL communicator, "
                             "parameter broadcast, the all-reduce between the two graphs of every step (no second GPU: what "
                             "the exchange costs beside the grouped pass, not scaling)", ["--rccl-single-rank"]),
]


def secondary_runs(args):
    """The other BASELINE configurations, 20 timed steps each, run by this script in child processes (one at a time,
    after the headline's timed region; this process keeps its GPU memory but launches nothing meanwhile)."""
    import subprocess
    res = []
    for name, what, flags in SECONDARY:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "20", "--warmup", "5", "--batch",
               str(args.batch), "--no-cpu-baseline", "--no-secondary", "--no-launch-events"] + flags
        entry = {"name": name, "what": what, "flags": " ".join(flags)}
        t0 = time.perf_counter()
        env = None
        if "--rccl-single-rank" in flags:          # a one-rank process group of its own
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                entry.update(ok=False, rc=r.returncode, error=(r.stderr or "")[-400:])
            else:
                d = json.loads(line[-1])
                entry.update(ok=True, value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"], steps=d["steps"],
                             dtype=d["dtype"], workload=d["config"]["workload"], finite=d.get("finite"),
                             roofline={k: d["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac")})
                if "all_blurs" in d:
                    entry["all_blurs"] = [{k: b[k] for k in ("blur", "ms_per_image", "images_per_s", "passes_matrix_pipe",
                                                               "passes_screened", "passes_difference_form")} for b in d["all_blurs"]]
        except (subprocess.TimeoutExpired, ValueError, KeyError) as e:
            entry.update(ok=False, error=repr(e)[:400])
        entry["wall_s"] = time.perf_counter() - t0
        res.append(entry)
    return res


def write_layer_table(path, rec, n_instr):
    """Per-launch conv table of one step (mean over the instrumented steps), in launch order."""
    conv = [r for r in rec if r[0].startswith("conv")]
    per = len(conv) // n_instr
    lines = ["| # | kind | M | N(cout) | K | k | s | levels | GFLOP | us | TFLOP/s |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for i in range(per):
        k, f, _, g = conv[i]
        us = 1e3 * sum(conv[i + j * per][2] for j in range(n_instr)) / n_instr
        m = g.rows_in if k == "conv_dgrad" else g.rows_out
        lines.append("| %d | %s | %d | %d | %d | %d | %d | %d | %.2f | %.1f | %.0f |" %
                     (i, k[5:], m, g.cout, g.ksize * g.ksize * g.cin, g.ksize, g.stride, len(g.levels_in), f / 1e9, us,
                      f / (us * 1e-6) / 1e12 if us > 0 else 0))
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def cpu_baseline(args, B, full):
    from kd6d.synthetic import INTERNAL_K, MESH_DIAMETERS, make_batch
    from oracle import kd_step_ref as O
    cores = max(1, min(os.cpu_count() or 1, 64))
    torch.set_num_threads(cores)
    stepper = O.KDStepRef(args.student, "darknet53", K=INTERNAL_K, diameters=MESH_DIAMETERS, kd_weight=5.0,
                          teacher_cls_bias=TEACHER_CLS_BIAS)
    images, targets = make_batch(B, 0, full_frame=full, mixed_classes=args.workload == "linemod13")
    td = [t.as_dict() for t in targets]
    stepper.step(images.tensors, td)                      # warm-up
    n = max(1, args.cpu_steps)
    t0 = time.perf_counter()
    for _ in range(n):
        stepper.step(images.tensors, td)
    dt = time.perf_counter() - t0
    return {"value": B * n / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d steps of the same B=%d step after 1 warm-up (oracle/kd_step_ref.py, torch fp32, %d threads)"
                      % (n, B, cores), "ms_per_step": dt / n * 1e3}


if __name__ == "__main__":
    main()
