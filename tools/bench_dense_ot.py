"""BASELINE config 5: dense optimal transport over a 128x128 cell grid of 16-D local predictions, 1 MI355X.

Times kd6d_sinkhorn_dense_fwd_bwd (forward value + both gradients) per image and prices it against the
fp32 vector roofline: the online-logsumexp form streams only (N+M)(D+2) floats per pass, so the bound is the
VALU: per (row, column) pair D subtracts + D FMAs + ~8 ops of running-max / exp2 bookkeeping."""
import json
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "kd-6d-pose-adlp_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kd6d import ops  # noqa: E402

dev = torch.device("cuda:0")
N = M = 128 * 128
D = 16
r = np.random.default_rng(0)
sig = lambda z: 1.0 / (1.0 + np.exp(-z))
x = torch.from_numpy(sig(r.normal(0, 2, (N, D))).astype(np.float32)).to(dev)
y = torch.from_numpy(sig(r.normal(0, 2, (M, D))).astype(np.float32)).to(dev)
a = torch.from_numpy(sig(r.normal(0, 2, N)).astype(np.float32)).to(dev)
b = torch.from_numpy(sig(r.normal(0, 2, M)).astype(np.float32)).to(dev)
out = []
for blur in (0.05, 0.001):
    diam = float(torch.sqrt(((torch.maximum(x.max(0).values, y.max(0).values)
                              - torch.minimum(x.min(0).values, y.min(0).values)) ** 2).sum()))
    n_eps = 2 + int(np.ceil((2 * np.log(blur) - 2 * np.log(diam)) / (2 * np.log(0.5))))
    passes = 4 * (1 + n_eps + 1)
    for _ in range(2):
        ops.sinkhorn_dense(x, a, y, b, blur=blur, scaling=0.5, reach=0.5, diameter=diam)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 5
    e0.record()
    for _ in range(iters):
        loss, gx, ga = ops.sinkhorn_dense(x, a, y, b, blur=blur, scaling=0.5, reach=0.5, diameter=diam)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    pairs = passes * float(N) * float(M)
    laneops = pairs * (2 * D + 8)                       # fp32 lane-operations (an FMA counted once)
    peak = 256 * 4 * 32 * 2.4e9                         # lanes per clock x clock = fp32 lane-ops/s (157 TFLOP/s / 2)
    out.append({"blur": blur, "diameter": diam, "eps_steps": n_eps, "softmin_passes": passes, "ms_per_image": ms,
                "images_per_s": 1e3 / ms, "pairs_per_s": pairs / (ms * 1e-3),
                "valu_lane_ops_per_s": laneops / (ms * 1e-3), "frac_of_fp32_vector_peak": laneops / (ms * 1e-3) / peak,
                "loss": float(loss)})
print(json.dumps({"workload": "dense OT, 128x128 cells, 16-D codes, p=2, scaling .5, reach .5 (BASELINE config 5)",
                  "N": N, "M": M, "D": D, "results": out}))
