"""Throughput of the MSSA / TNSM variants (BASELINE.json configs[4]): fwd + L1 + bwd + fused Adam, bs=16, 400x600, one MI355X (dev tool)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from hvi_cidnet_amd.dp import DataParallelTrainer
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
out = {"batch": B, "shape": "3x400x600", "step": "fwd + L1 + bwd + fused Adam"}
for name, ctor, loss in (("CIDNet_MSSA", P.CIDNet_MSSA, None), ("CIDNet_TNSM", P.CIDNet_TNSM, "tnsm")):
    torch.manual_seed(0)
    m = ctor().to(dev)
    lf = (lambda y, gt: (y[0] - gt).abs().mean() + 0.1 * y[1].mean()) if loss == "tnsm" else None
    tr = DataParallelTrainer(m, lr=1e-4, loss_fn=lf, use_hip_kernels=True) if lf else DataParallelTrainer(m, lr=1e-4)
    x = torch.rand(B, 3, 400, 600, device=dev); gt = torch.rand(B, 3, 400, 600, device=dev)
    for _ in range(3): tr.step(x, gt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 6
    for _ in range(n): tr.step(x, gt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    out[name] = {"ms_per_step": round(dt * 1e3, 2), "images/s": round(B / dt, 1)}
    del tr, m
    torch.cuda.empty_cache()
print(json.dumps(out))
