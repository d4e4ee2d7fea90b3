import os, sys
sys.path.insert(0, "/root/repo")
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
B = 8
for HW in (60000, 3750):
  for M in (16, 48, 96, 192):
    for K in (32, 64, 128, 256, 512):
        x = torch.randn(B, K, HW, device=dev); w = torch.randn(M, K, device=dev)
        y = torch.empty(B, M, HW, device=dev)
        t = timeit(lambda: ops.pw_conv_bf16x3(x, 0, K * HW, w, 0, 0, K, 1, y, 0, M * HW, B, M, K, HW))
        print(f"HW={HW} M={M:3d} K={K:3d}: {t:7.1f} us")
