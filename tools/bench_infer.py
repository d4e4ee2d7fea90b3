"""Config 4 of BASELINE.json: HBM roofline of the HVIT / PHVIT kernels on 32x3x1024x1024 and CIDNet inference
throughput at that shape (dev tool; prints one JSON object)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from hvi_cidnet_amd import ops

dev = torch.device("cuda:0")
B, H, W = 32, 1024, 1024
x = torch.rand(B, 3, H, W, device=dev)
k = torch.full([1], 0.2, device=dev)

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

out = {}
px = B * H * W
with torch.no_grad():
    hvi = ops.HVITFn.apply(x, k)
    ms = timeit(lambda: ops.HVITFn.apply(x, k))
    out["hvit_fwd"] = {"ms": round(ms, 4), "GB/s": round(24.0 * px / (ms * 1e-3) / 1e9, 1), "alg_bytes_per_px": 24}
    ms = timeit(lambda: ops.PHVITFn.apply(hvi, None, None, 0.2, False, 1.3, False, 1.0))
    out["phvit_fwd"] = {"ms": round(ms, 4), "GB/s": round(24.0 * px / (ms * 1e-3) / 1e9, 1), "alg_bytes_per_px": 24}
g = torch.rand_like(x)
xr = x.clone().requires_grad_(True)
kr = k.clone().requires_grad_(True)
def bwd():
    y = ops.HVITFn.apply(xr, kr)
    y.backward(g)
    xr.grad = None; kr.grad = None
ms_fb = timeit(bwd, 10)
out["hvit_fwd+bwd"] = {"ms": round(ms_fb, 4), "GB/s": round((24.0 + 36.0) * px / (ms_fb * 1e-3) / 1e9, 1), "alg_bytes_per_px": 60}
m = P.CIDNet().to(dev).eval()
with torch.no_grad():
    ms = timeit(lambda: m(x), 3)
out["cidnet_inference_1024x1024_bs32"] = {"ms": round(ms, 2), "images/s": round(B / (ms * 1e-3), 1)}
out["hbm_peak_GBs"] = 8000
print(json.dumps(out))
