"""Yardstick only (never on the product path): what the vendor GEMM (torch.matmul -> hipBLASLt) reaches
on the plain-GEMM equivalents of the step's conv layers, bf16, same M/N/K.  An implicit GEMM cannot beat
this by much; it tells how far the hand-written kernels are from what the hardware gives at these sizes."""
import torch

SHAPES = [("t.head.tower", 21824, 256, 2304), ("t.s3.3x3", 16384, 256, 1152), ("t.s4.3x3", 4096, 512, 2304),
          ("t.s5.3x3", 1024, 1024, 4608), ("t.s3.1x1", 16384, 128, 256), ("t.s4.1x1", 4096, 256, 512),
          ("t.s2.3x3", 65536, 128, 576), ("t.s1.3x3", 262144, 64, 288), ("t.fpn.p6", 256, 256, 9216),
          ("s.head.tower", 21760, 128, 1152), ("big", 8192, 8192, 8192)]
dev = torch.device("cuda:0")
print("| layer | M | N | K | us | TFLOP/s |\n|---|---|---|---|---|---|")
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, b.t(), out=c)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.matmul(a, b.t(), out=c)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(20):
            torch.matmul(a, b.t(), out=c)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("| %s | %d | %d | %d | %.1f | %.0f |" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6))
