"""How long does the host take to ENQUEUE one training step (no sync) vs the step's wall time? (dev tool)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from hvi_cidnet_amd.dp import DataParallelTrainer

dev = torch.device("cuda:0")
B, H, W = (int(a) for a in (sys.argv[1:4] or (8, 400, 600)))
m = P.CIDNet().to(dev)
tr = DataParallelTrainer(m, lr=1e-4)
x = torch.rand(B, 3, H, W, device=dev); gt = torch.rand(B, 3, H, W, device=dev)
for _ in range(3): tr.step(x, gt)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); tr.step(x, gt); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0):7.2f} ms   step wall {1e3 * (t2 - t0):7.2f} ms", flush=True)
