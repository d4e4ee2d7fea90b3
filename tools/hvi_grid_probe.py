import os, sys, torch
sys.path.insert(0, os.getcwd())
from hvi_cidnet_amd import ops
dev = torch.device("cuda:0")
B, H, W = 32, 1024, 1024
x = torch.rand(B, 3, H, W, device=dev); k = torch.full([1], 0.2, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    hvi = ops.HVITFn.apply(x, k)
    a = timeit(lambda: ops.HVITFn.apply(x, k)); b = timeit(lambda: ops.PHVITFn.apply(hvi, None, None, 0.2, False, 1.3, False, 1.0))
px = B * H * W
print(os.environ.get("CIDNET_LIB_PATH", "default"), f"HVIT {24.0 * px / a / 1e6:.0f} GB/s  PHVIT {24.0 * px / b / 1e6:.0f} GB/s")
