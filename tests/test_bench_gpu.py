"""bench.py as the driver runs it (one JSON line on stdout, the contract's keys) and its self-launch for --gpus N."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_line_contract_default_launch_mode(gpu_device):
    """`python bench.py --gpus 1 --steps 7 --warmup 2` (no secondary runs, no CPU leg): exactly one line on stdout, JSON,
    with the contract's fields, the roofline object, and the grouped teacher pass accounted for -- the timed region ends
    with a completed pass, so at least steps * B images went through the teacher inside it."""
    r = _run("--gpus", "1", "--steps", "7", "--warmup", "2", "--no-cpu-baseline", "--no-secondary", "--no-launch-events")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 7 and d["warmup"] == 2 and d["unit"] == "images/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert d["finite"] is True and d["barrier_timeouts"] == 0
    assert d["value"] == pytest.approx(16 * 1e3 / d["ms_per_step"], rel=1e-6)
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-6)
    assert rf["traffic"] is None or rf["traffic"] > 1e9
    cfg = d["config"]
    assert cfg["teacher_group"] == 3 and "workload" in cfg and "model" not in cfg
    assert cfg["teacher_passes_completed_in_timed_region"] == 3       # ceil(7 / 3): the region ends with a completed pass
    assert cfg["teacher_segments_in_timed_region"] == 7 and cfg["teacher_images_in_timed_region"] == 7 * 16


def test_bench_self_launch_reports_missing_gpus(gpu_device):
    """`python bench.py --gpus N` with no launcher around it starts its own N ranks before touching the GPU; with fewer
    GPUs than ranks the children say so and the run fails -- no usage message, no hang."""
    have = torch.cuda.device_count()
    want = have + 1
    r = _run("--gpus", str(want), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", timeout=300)
    assert r.returncode != 0
    assert ("%d GPUs needed, %d visible" % (want, have)) in (r.stderr + r.stdout), (r.stdout[-1000:], r.stderr[-1000:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_one_rank_rccl_rehearsal_takes_the_grouped_mode(gpu_device):
    """The N > 1 code path on one rank (process group, kd6d communicator, parameter broadcast, the all-reduce between the
    two graphs of every step): bench.py records every graph from a sample batch BEFORE the communicator exists
    (GroupedTeacherKDStep.prepare) and keeps the grouped teacher pass; --exchange overlap (collectives captured inside
    the step graph) falls back to one teacher forward per step."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    for extra, group, graphs in (((), 3, 2), (("--exchange", "overlap"), 1, 1)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                            "--rccl-single-rank", "--no-cpu-baseline", "--no-secondary", "--no-launch-events"] + list(extra),
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        cfg = d["config"]
        assert d["finite"] is True and d["barrier_timeouts"] == 0 and d["rccl_ranks"] == 1
        assert cfg["exchange"].startswith("kd6d_comm") and cfg["teacher_group"] == group, cfg
        assert ("%d graph" % graphs) in cfg["launch"], cfg["launch"]
