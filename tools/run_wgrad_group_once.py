"""Launch the grouped head weight gradient a few times eagerly (for rocprofv3 --pmc runs)."""
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(HERE, "kd-6d-pose-adlp_amd"))
import torch  # noqa: E402

from kd6d import ops  # noqa: E402

dev = torch.device("cuda:0")
C, B = 128, 16
levels = [(32, 32), (16, 16), (8, 8), (4, 4)]
g = torch.Generator().manual_seed(0)
layers = []
for cout in [C] * 8 + [16, 240]:
    geom = ops.Geom(B, C, cout, 3, 1, 1, levels)
    x = (torch.randn(geom.rows_in, C, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dy = (torch.randn(geom.rows_out, cout, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    layers.append((geom, x, dy, torch.zeros(cout * 9 * C, dtype=torch.float32, device=dev),
                   torch.zeros(cout, dtype=torch.float32, device=dev)))
grp = ops.WgradGroup(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
for _ in range(4):
    for ge, x, dy, dw, db in layers:
        grp.add(ge, x, dy, dw, db)
    grp.launch()
    torch.cuda.synchronize()
print("done")
