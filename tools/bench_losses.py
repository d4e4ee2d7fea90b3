"""Cost of the device-side training objective (L1 + SSIM + Edge on RGB and on HVIT(out)) at the benchmark's image size (dev tool)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
dev = torch.device("cuda:0")
m = P.CIDNet().to(dev)
out = torch.rand(8, 3, 400, 600, device=dev, requires_grad=True); gt = torch.rand(8, 3, 400, 600, device=dev)

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

ssim = P.SSIM(weight=0.5); crit = P.CIDNetLoss(m)
def f_ssim():
    out.grad = None; ssim(out, gt).backward()
def f_all():
    out.grad = None; crit(out, gt).backward()
print(json.dumps({"shape": "8x3x400x600", "ssim_fwd_bwd_ms": round(timeit(f_ssim), 3), "l1+ssim+edge_rgb_and_hvi_fwd_bwd_ms": round(timeit(f_all), 3)}))
