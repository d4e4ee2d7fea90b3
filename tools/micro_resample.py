"""Times the bilinear adjoint and the PReLU backward at the step's shapes (dev tool).  Bytes = dout read once + din written once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# (B, C, Hi, Wi, Ho, Wo): din is (Hi, Wi), dout is (Ho, Wo)
shapes = [(8, 36, 200, 300, 400, 600), (8, 36, 400, 600, 200, 300), (8, 36, 100, 150, 200, 300), (8, 36, 200, 300, 100, 150),
          (8, 72, 50, 75, 100, 150), (8, 72, 100, 150, 50, 75)]
for B, C, Hi, Wi, Ho, Wo in shapes:
    dout = torch.randn(B, C, Ho, Wo, device=dev); din = torch.empty(B, C, Hi, Wi, device=dev)
    us = timeit(lambda: ops.bilinear_bwd(dout, din, B, C, Hi, Wi, Ho, Wo))
    by = (dout.numel() + din.numel()) * 4
    print(f"bilinear_bwd {B}x{C} ({Ho}x{Wo}) -> ({Hi}x{Wi}): {us:7.1f} us  {by / us / 1e3:6.0f} GB/s")
