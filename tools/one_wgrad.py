"""Run one conv3 weight-gradient shape a few times (for rocprofv3 counter passes; dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
B, M, N, H, W = (int(a) for a in (sys.argv[1:6] or (8, 36, 36, 200, 300)))
dev = torch.device("cuda:0")
x = torch.rand(B, N, H, W, device=dev); gy = torch.rand(B, M, H, W, device=dev); gw = torch.empty(M, N, 3, 3, device=dev)
for _ in range(5): ops.conv3x3_wgrad(gy, x, gw, B, M, N, H, W)
torch.cuda.synchronize()
